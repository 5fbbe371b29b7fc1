#!/usr/bin/env python3
"""Phase times of the chain kernel K9 on one benchmark batch (CMDP_CHAIN_DEBUG=1 prints them per call):
    CMDP_CHAIN_DEBUG=1 python tools/dbg_k9.py [suite class scope n_instances steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from colosseum_amd import _lib as L  # noqa: E402
from colosseum_amd import benchmark as bm  # noqa: E402
from colosseum_amd.mdp import make_model  # noqa: E402

suite = sys.argv[1] if len(sys.argv) > 1 else "benchmark_continuous_ergodic"
cls = sys.argv[2] if len(sys.argv) > 2 else "MiniGridEmptyContinuous"
scope = sys.argv[3] if len(sys.argv) > 3 else "prms_3"
n = int(sys.argv[4]) if len(sys.argv) > 4 else 20
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 3000
cfg = json.load(open(os.path.join(ROOT, "tests/golden/G11_benchmark_configs.json")))[suite]["mdp_configs"][cls][scope]
models = [make_model(cls, seed=s, **cfg) for s in range(n)]
t0 = time.time()
rows = bm._run_group(models, list(range(n)), "QLearningContinuous", bm.DEFAULT_AGENT_CONFIGS["QLearningContinuous"], steps, 100,
                     L.RNG_MT_COMPAT, 0, beta_rewards="philox")
dt = time.time() - t0
print(f"{cls} {scope}: {n} instances, S={models[0].n_states}, {steps} steps, {len(rows[0])} rows in {dt:.2f} s = {dt / len(rows[0]) * 1e3:.2f} ms per row")
# a second, longer run on fresh copies: (t2 - t1) / extra rows = the steady-state cost of a log row
if len(sys.argv) > 6:
    steps2 = int(sys.argv[6])
    t0 = time.time()
    rows2 = bm._run_group(models, list(range(n)), "QLearningContinuous", bm.DEFAULT_AGENT_CONFIGS["QLearningContinuous"], steps2, 100,
                          L.RNG_MT_COMPAT, 0, beta_rewards="philox")
    dt2 = time.time() - t0
    print(f"steady state: {(dt2 - dt) / (len(rows2[0]) - len(rows[0])) * 1e3:.3f} ms per row ({steps2} steps in {dt2:.2f} s)")
