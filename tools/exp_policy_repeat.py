#!/usr/bin/env python3
"""How often is the greedy policy of the continuous Q-learning agent the SAME at two consecutive log rows (100 steps apart)?
(If often, the stationary-distribution solve of a row could be skipped for instances whose policy did not change.)
    python tools/exp_policy_repeat.py [class scope n_instances steps]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from colosseum_amd import _lib as L  # noqa: E402
from colosseum_amd import benchmark as bm  # noqa: E402
from colosseum_amd.agents import BatchedQLearningContinuous  # noqa: E402
from colosseum_amd.batched import BatchedMDP  # noqa: E402
from colosseum_amd.mdp import make_model  # noqa: E402

cls = sys.argv[1] if len(sys.argv) > 1 else "MiniGridEmptyContinuous"
scope = sys.argv[2] if len(sys.argv) > 2 else "prms_0"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 500000
cfg = json.load(open(os.path.join(ROOT, "tests/golden/G11_benchmark_configs.json")))["benchmark_continuous_ergodic"]["mdp_configs"][cls][scope]
models = [make_model(cls, seed=s, **cfg) for s in range(n)]
env = BatchedMDP(models, rng_mode=L.RNG_PHILOX, philox_keys=np.arange(n, dtype=np.uint64) + 17)
agent = BatchedQLearningContinuous(env, list(range(n)), optimization_horizon=steps, **bm.DEFAULT_AGENT_CONFIGS["QLearningContinuous"])
env.reset()
prev = None
same = np.zeros((steps // 100, n), bool)
changed_states = np.zeros((steps // 100, n), np.int32)
for i in range(steps // 100):
    agent.run(100)
    pi = [p.argmax(1) for p in agent.policy()]
    if prev is not None:
        for b in range(n):
            d = int((pi[b] != prev[b]).sum())
            same[i, b] = d == 0
            changed_states[i, b] = d
    prev = pi
q = steps // 100 // 5
for k in range(5):
    sl = slice(k * q, (k + 1) * q)
    print("rows %5d..%5d: policy unchanged in %.1f %% of (row, instance) pairs; mean states whose action changed %.2f" % (
        k * q, (k + 1) * q, 100 * same[sl].mean(), changed_states[sl].mean()))
