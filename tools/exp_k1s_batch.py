"""Experiment: K1 (tables in HBM, lane per instance) against K1S (tables in LDS, few instances per CU) as the batch grows.
K1's rate grows with the batch (more wavefronts hide its latency), K1S's does not (LDS capacity)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colosseum_amd import _lib as L  # noqa: E402
from colosseum_amd.batched import BatchedMDP, tables_from_models  # noqa: E402
from colosseum_amd.mdp import make_model  # noqa: E402


def timed(f, n=3):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n


out = {}
for name, cls, kws in (
    ("frozenlake20", "FrozenLakeContinuous", [dict(seed=s, size=20, p_frozen=0.9, p_rand=0.1) for s in range(8)]),
    ("minigrid_empty8", "MiniGridEmptyContinuous", [dict(seed=s, size=8, p_rand=0.1, p_lazy=0.05) for s in range(8)]),
    ("deepsea20_prand", "DeepSeaEpisodic", [dict(seed=s, size=20, p_rand=0.2) for s in range(8)]),
):
    models = [make_model(cls, **kw) for kw in kws]
    for B in (2048, 8192, 32768, 131072):
        ms = [models[i % len(models)] for i in range(B)]
        env = BatchedMDP(tables=tables_from_models(ms, with_dp=False), rng_mode=L.RNG_PHILOX, philox_keys=np.arange(B, dtype=np.uint64))
        env.reset()
        n = 1000
        row = dict(instances_per_workgroup=env.lds_plan().get("instances_per_workgroup"))
        for label, kernel in (("k1", L.ROLLOUT_GLOBAL), ("k1s", L.ROLLOUT_LDS_STOCHASTIC)):
            env.set_rollout_kernel(kernel)
            dt = timed(lambda: (env.rollout_async(n), env.synchronize()))
            row[label] = B * n / dt
        out[f"{name}_B{B}"] = row
        print(name, B, row, flush=True)
        env.close()
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "exp_k1s_batch.json"), "w"), indent=1)
