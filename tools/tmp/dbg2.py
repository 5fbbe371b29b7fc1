import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ["CMDP_K1E_VERBOSE"] = "1"
from colosseum_amd import _lib as L
from colosseum_amd.batched import BatchedMDP
from colosseum_amd.mdp.fast_batch import deepsea_episodic_tables
B, size = 40, 31
tables = deepsea_episodic_tables(np.arange(1000, 1000 + B), size)
keys = (np.arange(1000, 1000 + B) * 7919).astype(np.uint64)
env = BatchedMDP(tables=tables, rng_mode=L.RNG_PHILOX, philox_keys=keys)
print(env.lds_plan(), flush=True)
env.set_rollout_kernel(L.ROLLOUT_EPISODE_PARALLEL)
env.reset()
try:
    a = env.rollout(77)
    print("ok", a["reward_sum"][:3])
except Exception as e:
    print("ERR", e)
