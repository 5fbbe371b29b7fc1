#!/usr/bin/env python3
"""One stochastic-dynamics rollout workload for profiling: FrozenLake 20x20 (p_rand 0.1), Philox,
on-device random policy, B instances, `--kernel k1|k1s|auto`.  Prints one JSON line with the HIP-event launch time, the
transitions per launch and -- from SURVEY 8(d)'s CSR accounting, 8 + 8*nnz(s,a) + 28 bytes per transition with the mean
row length of the batch -- the algorithmic bytes per launch.      python tools/prof_stoch.py --instances 4096 --kernel k1s"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colosseum_amd import _lib as L  # noqa: E402
from colosseum_amd.batched import BatchedMDP, tables_from_models  # noqa: E402
from colosseum_amd.mdp import make_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--instances", type=int, default=4096)
ap.add_argument("--kernel", default="auto")
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("--launches", type=int, default=5)
ap.add_argument("--family", default="frozenlake20")
a = ap.parse_args()

FAMILIES = {
    "frozenlake20": ("FrozenLakeContinuous", [dict(seed=s, size=20, p_frozen=0.9, p_rand=0.1) for s in range(8)]),
    "minigrid_empty8": ("MiniGridEmptyContinuous", [dict(seed=s, size=8, p_rand=0.1, p_lazy=0.05) for s in range(8)]),
    "deepsea20_prand": ("DeepSeaEpisodic", [dict(seed=s, size=20, p_rand=0.2) for s in range(8)]),
}
cls, kws = FAMILIES[a.family]
models = [make_model(cls, **kw) for kw in kws]
B = a.instances
ms = [models[i % len(models)] for i in range(B)]
env = BatchedMDP(tables=tables_from_models(ms, with_dp=False), rng_mode=L.RNG_PHILOX, philox_keys=np.arange(B, dtype=np.uint64))
env.reset()
env.set_rollout_kernel({"auto": L.ROLLOUT_AUTO, "k1": L.ROLLOUT_GLOBAL, "k1s": L.ROLLOUT_LDS_STOCHASTIC}[a.kernel])
env.rollout_async(a.steps)
env.synchronize()
t0 = time.perf_counter()
for _ in range(a.launches):
    env.rollout_async(a.steps)
env.synchronize()
dt = (time.perf_counter() - t0) / a.launches
nnz = float(np.mean([len(m.sp_next) / (m.n_states * m.n_actions) for m in models]))
print(json.dumps(dict(family=a.family, instances=B, kernel=a.kernel, lds_plan=env.lds_plan(), steps_per_launch=a.steps,
                      launch_ms=dt * 1e3, transitions_per_s=B * a.steps / dt, mean_entries_per_row=nnz,
                      algorithmic_bytes_per_transition=8 + 8 * nnz + 28,
                      algorithmic_bytes_per_launch=(8 + 8 * nnz + 28) * B * a.steps, build_id=L.load().cmdp_build_id().decode()[:16])))
env.close()
