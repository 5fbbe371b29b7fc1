import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from colosseum_amd import _lib as L
from colosseum_amd.batched import BatchedMDP
from colosseum_amd.mdp.fast_batch import deepsea_episodic_tables
B, size = 75, 10
tables = deepsea_episodic_tables(np.arange(B), size)
keys = (np.arange(B, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(12345)
for n in (10, 20, 37, 128, 1280, 1281, 5000):
    out = {}
    for which in (L.ROLLOUT_GLOBAL, L.ROLLOUT_EPISODE_PARALLEL):
        env = BatchedMDP(tables=tables, rng_mode=L.RNG_PHILOX, philox_keys=keys)
        env.set_rollout_kernel(which)
        env.reset()
        a = env.rollout(n)
        vs, vsa = env.visits()
        out[which] = (a["last_obs"].copy(), a["reward_sum"].copy(), vs.copy(), vsa.copy())
        env.close()
    g, e = out[L.ROLLOUT_GLOBAL], out[L.ROLLOUT_EPISODE_PARALLEL]
    print(n, [int((x != y).sum()) for x, y in zip(g, e)], "first sums", g[1][:4], e[1][:4], flush=True)
