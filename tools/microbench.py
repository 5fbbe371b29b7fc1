#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md (not the headline bench): stochastic-dynamics rollouts through the
HBM-table kernel, the per-call step API (PCIe inclusive), hardness batches."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colosseum_amd import _lib as L  # noqa: E402
from colosseum_amd.batched import BatchedMDP, tables_from_models  # noqa: E402
from colosseum_amd.mdp import make_model  # noqa: E402
from colosseum_amd.mdp.fast_batch import deepsea_episodic_tables  # noqa: E402


def timed(f, n=5):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n


def replicate(tables_one, models, copies):
    ms = [models[i % len(models)] for i in range(copies)]
    return tables_from_models(ms, with_dp=False)


out = {}
# stochastic dynamics, Philox, on-device random policy
for name, cls, kws, B in (
    ("frozenlake20", "FrozenLakeContinuous", [dict(seed=s, size=20, p_frozen=0.9, p_rand=0.1) for s in range(8)], 4096),
    ("minigrid_empty8", "MiniGridEmptyContinuous", [dict(seed=s, size=8, p_rand=0.1, p_lazy=0.05) for s in range(8)], 8192),
    ("deepsea20_prand", "DeepSeaEpisodic", [dict(seed=s, size=20, p_rand=0.2) for s in range(8)], 16384),
):
    models = [make_model(cls, **kw) for kw in kws]
    env = BatchedMDP(tables=replicate(None, models, B), rng_mode=L.RNG_PHILOX, philox_keys=np.arange(B, dtype=np.uint64))
    env.reset()
    n = 2000
    out[name] = dict(instances=B, states=int(models[0].n_states), lds_plan=env.lds_plan())
    for label, kernel in (("hbm_tables_k1", L.ROLLOUT_GLOBAL), ("lds_resident_k1s", L.ROLLOUT_LDS_STOCHASTIC)):
        try:
            env.set_rollout_kernel(kernel)
            dt = timed(lambda: (env.rollout_async(n), env.synchronize()))
            out[name][label + "_steps_per_s"] = B * n / dt
        except L.CmdpError as e:
            out[name][label + "_steps_per_s"] = None
            out[name][label + "_error"] = str(e)[:120]
    env.close()

# per-call step API: H2D actions + kernel + D2H obs/reward/type every call
B = 65536
env = BatchedMDP(tables=deepsea_episodic_tables(np.arange(B), 30), rng_mode=L.RNG_PHILOX)
env.reset()
acts = np.random.RandomState(0).randint(0, 2, B).astype(np.int32)
dt = timed(lambda: env.step(acts, auto_reset=True), n=50)
out["step_api_deepsea30"] = dict(instances=B, ms_per_call=dt * 1e3, steps_per_s=B / dt)
env.close()
# f1: on-device Q-learning agents (Bernstein UCB) fused with the environment step, DeepSea-30
from colosseum_amd.agents import BatchedQLearningEpisodic  # noqa: E402

for B in (4096, 32768):
    env = BatchedMDP(tables=deepsea_episodic_tables(np.arange(B), 30), rng_mode=L.RNG_PHILOX)
    env.reset()
    ag = BatchedQLearningEpisodic(env, np.arange(B), optimization_horizon=500_000, p=0.05, c_1=0.9415278732894797,
                                  c_2=0.013873778519317169, min_at=0.07263563483119442, UCB_type="bernstein")
    n = 2000
    dt = timed(lambda: ag.run(n), n=3)
    out[f"qlearning_deepsea30_B{B}"] = dict(instances=B, agent_steps_per_s=B * n / dt)
    ag.close()
    env.close()
print(json.dumps(out, indent=1))
