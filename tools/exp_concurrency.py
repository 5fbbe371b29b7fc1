#!/usr/bin/env python3
"""N copies of one benchmark batch run concurrently (one host thread and stream each): time per log row against N.
Separates GPU-side contention from everything else a full C4 run mixes in.
    python tools/exp_concurrency.py [class scope n_instances steps]"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CMDP_SYNC_MODE", "block")
from colosseum_amd import _lib as L  # noqa: E402
from colosseum_amd import benchmark as bm  # noqa: E402
from colosseum_amd.mdp import make_model  # noqa: E402

cls = sys.argv[1] if len(sys.argv) > 1 else "MiniGridEmptyContinuous"
scope = sys.argv[2] if len(sys.argv) > 2 else "prms_0"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 90
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 100000
suite = "benchmark_continuous_ergodic" if "Continuous" in cls else "benchmark_episodic_ergodic"
cfg = json.load(open(os.path.join(ROOT, "tests/golden/G11_benchmark_configs.json")))[suite]["mdp_configs"][cls][scope]
models = [make_model(cls, seed=s, **cfg) for s in range(n)]
agent = "QLearningContinuous" if "Continuous" in cls else "QLearningEpisodic"
sys.setswitchinterval(2e-4)


def one(out, k):
    rows = bm._run_group(models, list(range(n)), agent, bm.DEFAULT_AGENT_CONFIGS[agent], steps, 100, L.RNG_MT_COMPAT, 0, beta_rewards="philox")
    out[k] = rows[0].phase_seconds


one({}, 0)   # warm-up: library, context, plans
for N in (1, 2, 4, 8, 16):
    out = {}
    th = [threading.Thread(target=one, args=(out, k)) for k in range(N)]
    t0 = time.time()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.time() - t0
    inter = sorted(v[1] for v in out.values())
    print("%2d copies: wall %.2f s; interaction %.2f .. %.2f s = %.2f .. %.2f ms per row; set-up %.2f .. %.2f s" % (
        N, wall, inter[0], inter[-1], inter[0] / (steps / 100) * 1e3, inter[-1] / (steps / 100) * 1e3,
        min(v[0] for v in out.values()), max(v[0] for v in out.values())), flush=True)
