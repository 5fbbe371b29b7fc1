import sys, os, json, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from colosseum_amd import _lib as L
from colosseum_amd.batched import BatchedMDP
from colosseum_amd.mdp import make_model
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
cfg = json.load(open(os.path.join(ROOT, "tests/golden/G11_benchmark_configs.json")))["benchmark_continuous_ergodic"]["mdp_configs"]
cls, scope = "MiniGridEmptyContinuous", "prms_3"
print(cfg[cls][scope])
ms = [make_model(cls, seed=s, **cfg[cls][scope]) for s in range(3)]
S, A = ms[0].n_states, ms[0].n_actions
rng = np.random.default_rng(S)
acts = [rng.integers(0, A, m.n_states).astype(np.int32) for m in ms]
acts += [np.full(m.n_states, k % A, np.int32) for k, m in enumerate(ms)]
starts = [int(rng.integers(0, m.n_states)) for m in ms] * 2
env = BatchedMDP(ms + ms, rng_mode=L.RNG_PHILOX, with_env=False)
fast, ncls_fast = env.average_reward(acts, starts)
print([type(x).__name__ for x in fast], ncls_fast)
for i in range(6):
    mask = np.zeros(6, bool); mask[i] = True
    env.average_reward(acts, starts, mask=mask)
    v = C.c_double(); L.check(L.load().cmdp_stat(env.handle, L.STAT_CHAIN_FAST_INSTANCES, C.byref(v)))
    print(i, "fast (incl. 5 masked):", v.value)
