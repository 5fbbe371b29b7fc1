#!/usr/bin/env python3
"""Where the set-up of one benchmark batch goes (tables, handle, baselines, agent): cProfile of `_run_group` with a short run.
    python tools/prof_setup.py [suite class scope n_instances]"""
import cProfile
import json
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from colosseum_amd import _lib as L  # noqa: E402
from colosseum_amd import benchmark as bm  # noqa: E402
from colosseum_amd.mdp import make_model  # noqa: E402

suite = sys.argv[1] if len(sys.argv) > 1 else "benchmark_continuous_ergodic"
cls = sys.argv[2] if len(sys.argv) > 2 else "MiniGridEmptyContinuous"
scope = sys.argv[3] if len(sys.argv) > 3 else "prms_0"
n = int(sys.argv[4]) if len(sys.argv) > 4 else 90
cfg = json.load(open(os.path.join(ROOT, "tests/golden/G11_benchmark_configs.json")))[suite]["mdp_configs"][cls][scope]
models = [make_model(cls, seed=s, **cfg) for s in range(n)]
agent = "QLearningContinuous" if "Continuous" in cls else "QLearningEpisodic"
for rep in range(2):  # the first pass pays the library load and the HIP context
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    rows = bm._run_group(models, list(range(n)), agent, bm.DEFAULT_AGENT_CONFIGS[agent], 1000, 100, L.RNG_MT_COMPAT, 0, beta_rewards="philox")
    pr.disable()
    print("pass %d: %.2f s, phases %s" % (rep, time.time() - t0, rows[0].phase_seconds))
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
