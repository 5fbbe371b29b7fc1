import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colosseum_amd import _lib as L
from colosseum_amd.batched import BatchedMDP, tables_from_models
from colosseum_amd.mdp import make_model
B = 70
models = [make_model("RiverSwimEpisodic", seed=1000 + i, size=40) for i in range(B)]
m = models[0]
print("S", m.n_states, "A", m.n_actions, "H", m.H, "rewards", sorted(set(np.round(m.sp_rp0, 6).tolist()))[:10], "nsucc", int(np.diff(m.sp_ptr).max()), flush=True)
tables = tables_from_models(models, True, False)
keys = (np.arange(1000, 1000 + B) * 7919).astype(np.uint64)
res = {}
for which in (L.ROLLOUT_GLOBAL, L.ROLLOUT_EPISODE_PARALLEL):
    env = BatchedMDP(tables=tables, rng_mode=L.RNG_PHILOX, philox_keys=keys)
    env.set_rollout_kernel(which)
    env.reset()
    a = env.rollout(13)
    b = env.rollout(9001)
    vs, vsa = env.visits()
    res[which] = (a["last_obs"], a["reward_sum"], b["last_obs"], b["reward_sum"], vs, vsa)
    env.close()
g, e = res[L.ROLLOUT_GLOBAL], res[L.ROLLOUT_EPISODE_PARALLEL]
for name, x, y in zip(("a.last", "a.rsum", "b.last", "b.rsum", "vs", "vsa"), g, e):
    bad = np.nonzero(x != y)[0]
    print(name, "mismatches", len(bad), bad[:10], x[bad[:5]], y[bad[:5]])
from oracle import oracle as O
last, rsum, ovs, ovsa = O.batch_rollout(tables, 0, B, 13 + 9001, rng_mode=1, philox_keys=keys, want_visits=True)
tot = g[1] + g[3]
bad = np.nonzero(np.abs(tot - rsum) > 1e-9)[0]
print("oracle vs K1 rsum mismatches", len(bad), bad[:8], tot[bad[:5]], rsum[bad[:5]])
print("visits equal", np.array_equal(g[4], ovs), np.array_equal(g[5], ovsa))
# per-instance oracle env, launch by launch
for b in bad[:3]:
    e = O.OracleEnv(models[b], rng_mode=1, philox_key=int(keys[b]))
    e.reset()
    r1 = e.rollout(13, trace=False)
    r2 = e.rollout(9001, trace=False)
    print(b, "oracle env partial sums", r1["reward_sum"], r2["reward_sum"], "gpu", g[1][b], g[3][b])
