import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from colosseum_amd import _lib as L
from colosseum_amd.batched import BatchedMDP
from colosseum_amd.mdp.fast_batch import deepsea_episodic_tables
B = 65536
tables = deepsea_episodic_tables(np.arange(B), 30)
env = BatchedMDP(tables=tables, rng_mode=L.RNG_PHILOX, philox_keys=np.arange(B).astype(np.uint64))
env.reset()
ev = bench.HipEvents()
for trial in range(2):
    for _ in range(3):
        env.rollout_async(30000)
    env.synchronize()
    K = 40
    marks = [ev.create() for _ in range(K + 1)]
    ev.record(marks[0], env.stream)
    for k in range(K):
        env.rollout_async(30000)
        ev.record(marks[k + 1], env.stream)
    env.synchronize()
    print(" ".join("%.3f" % ev.elapsed_ms(marks[k], marks[k + 1]) for k in range(K)), flush=True)
