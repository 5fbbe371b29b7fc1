#!/usr/bin/env python3
"""Debug aid: baselines (optimal / worst / random average reward) of one benchmark instance through the batched loop and
through GpuMDP, side by side.   python tools/dbg_c4_baselines.py SUITE CLASS SCOPE SEED"""
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
from colosseum_amd.experiment.batched_loop import BatchedContinuousLoop  # noqa: E402
from colosseum_amd.mdp import gpu_mdp, make_model  # noqa: E402

suite, cls, scope, seed = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
cfg = json.load(open(os.path.join(ROOT, "tests/golden/G11_benchmark_configs.json")))[suite]["mdp_configs"][cls][scope]
m = make_model(cls, seed=seed, **cfg)
env = BatchedMDP([m], rng_mode=L.RNG_MT_COMPAT, flags=L.FLAG_REWARD_CACHE)
ag = BatchedQLearningContinuous(env, [seed], optimization_horizon=1500, **bm.DEFAULT_AGENT_CONFIGS["QLearningContinuous"])
loop = BatchedContinuousLoop(env, ag)
vt = loop.vt
print("batched  opt/worst/rand:", [repr(x.scalar(0)) for x in (vt.opt, vt.worst, vt.rand)])
mdp = getattr(gpu_mdp, cls)(seed=seed, **cfg)
print("GpuMDP   opt/worst/rand:", repr(mdp.optimal_average_reward), repr(mdp.worst_average_reward), repr(mdp.random_average_reward))
from colosseum_amd.hardness import _vi_rule
print("S", m.n_states, "A", m.n_actions, "nnz", len(m.csr()[1]), "vi rule", _vi_rule(m.n_states, m.n_actions, len(m.csr()[1])))
