"""Time the C2 rollout (65 536 x DeepSeaEpisodic-30, random policy) kernel by kernel: HIP events inside the library
(cmdp_stat) and the wall clock of back-to-back steps.  usage: python tools/time_rollout.py [B] [size] [steps] [kernel]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colosseum_amd import _lib as L
from colosseum_amd.batched import BatchedMDP
from colosseum_amd.mdp.fast_batch import deepsea_episodic_tables

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
size = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30000
which = int(sys.argv[4]) if len(sys.argv) > 4 else 0
t0 = time.time()
tables = deepsea_episodic_tables(np.arange(B), size)
keys = (np.arange(B, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(12345)
env = BatchedMDP(tables=tables, rng_mode=L.RNG_PHILOX, philox_keys=keys)
if which:
    env.set_rollout_kernel(which)
print("build %.1f s, plan %s" % (time.time() - t0, env.lds_plan()), flush=True)
env.reset()
for _ in range(3):
    env.rollout_async(n)
env.synchronize()
K = 10
t0 = time.time()
for _ in range(K):
    env.rollout_async(n)
env.synchronize()
dt = (time.time() - t0) / K
ms = {}
for name, st in (("rollout", L.STAT_ROLLOUT_KERNEL_MS), ("second", L.STAT_HIST_KERNEL_MS)):
    import ctypes as C
    v = C.c_double(0.0)
    rc = L.load().cmdp_stat(env.handle, st, C.byref(v))
    ms[name] = round(v.value, 4) if rc == 0 else None   # kernels without the split
print("B=%d size=%d n=%d: %.3f ms per step = %.3e transitions/s; kernel ms %s" % (B, size, n, dt * 1e3, B * n / dt, ms), flush=True)
env.close()
