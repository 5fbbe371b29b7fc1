"""Bulk construction of large batches without one Python object graph per instance.

DeepSea (reference colosseum/mdp/deep_sea/base.py): `does_seed_change_MDP_structure()` is False -- the
graph, its DFS order and the rewards by raw action do not depend on the seed; only the per-state action
permutation does (one `_rng.rand(2).argsort()` per state in first-touch order, mdp/base.py:505-518).
So a batch over seeds is the template built once by the generic builder plus one
`RandomState(seed).rand(S, 2)` per instance.  tests/test_builder.py checks the tables against the generic
builder instance by instance."""
from typing import Sequence

import numpy as np

from .registry import make_model


def deepsea_episodic_tables(seeds: Sequence[int], size: int, with_dp: bool = False) -> dict:
    """Tables (BatchedMDP(tables=...)) of DeepSeaEpisodic(seed=s, size=size) for every s in seeds: default
    arguments, i.e. randomize_actions=True, deterministic rewards, no p_rand."""
    seeds = np.asarray(seeds, np.int64)
    B = len(seeds)
    tpl = make_model("DeepSeaEpisodic", seed=0, size=size, randomize_actions=False)
    S, A, H = tpl.n_states, 2, tpl.H
    assert (np.diff(tpl.sp_ptr) == 1).all()
    next_raw = tpl.sp_next.reshape(S, A)  # identity mapping: agent action == raw action
    rew_raw = tpl.sp_rp0.reshape(S, A)
    # action permutation of every state, in first-touch (== state index) order
    draws = np.empty((B, S, A), np.float64)
    rs = np.random.RandomState(0)
    for i, s in enumerate(seeds.tolist()):
        rs.seed(s)  # == np.random.RandomState(s), the reference's mdp._rng (mdp/base.py:408)
        draws[i] = rs.rand(S, A)
    # rand(2).argsort() is [0, 1] unless the first draw is the larger one (A == 2: identity or swap)
    swap = (draws[:, :, 0] > draws[:, :, 1])[:, :, None]
    del draws
    # transitions of raw action a are filed under agent action map[a] (mdp_creation.py:226) and the reward
    # of agent action m is that of raw action map[m] (mdp/base.py:1179-1185); for a swap both are the reversal
    nxt = np.where(swap, next_raw[None, :, ::-1], next_raw[None, :, :]).astype(np.int32)
    rew = np.where(swap, rew_raw[None, :, ::-1], rew_raw[None, :, :])
    R = B * S * A
    t = dict(
        B=B, A=A, H=H, rewards_range=(0.0, 1.0),
        state_off=np.arange(B + 1, dtype=np.int64) * S,
        sp_ptr=np.arange(R + 1, dtype=np.int64),
        sp_next=nxt.reshape(-1),
        sp_cum=np.ones(R, np.float64),
        sp_reward=np.ascontiguousarray(rew, np.float64).reshape(-1),
        sp_rkind=np.zeros(R, np.uint8),
        sp_rp0=np.ascontiguousarray(rew, np.float64).reshape(-1),
        sp_rp1=np.zeros(R, np.float64),
        sp_seed=np.zeros(R, np.int32),  # never read: every row is deterministic (no sampler stream)
        start_off=np.arange(B + 1, dtype=np.int64),
        start_state=np.full(B, tpl.start_states[0], np.int32),
        start_cum=np.ones(B, np.float64),
        start_seed=np.zeros(B, np.int32),
    )
    if with_dp:
        t["csr_ptr"] = np.arange(R + 1, dtype=np.int64)
        t["csr_col"] = t["sp_next"]
        t["csr_val"] = np.ones(R, np.float32)
        t["R"] = t["sp_reward"].astype(np.float32)
    return t


def _build_fl(args):
    from .registry import make_model

    seed, size, kw = args
    m = make_model("FrozenLakeContinuous", seed=seed, size=size, **kw)
    ptr, col, val = m.csr()
    return m.n_states, ptr, col, val, m.reward_matrix().ravel()


def frozenlake_dp_tables(seeds, size, workers=1, context="fork", **kw):
    """DP half of the tables (`tables_from_models(..., with_env=False)`) for many FrozenLakeContinuous(seed, size, **kw)
    instances (config C3 = p_frozen 0.9, slippery, p_rand 0.1), built by a process pool -- the builder is
    host Python.  `context="fork"` must only be used before the HIP runtime is initialised in this process; "spawn" is
    safe at any time."""
    import multiprocessing as mp

    kw = dict(dict(p_frozen=0.9, is_slippery=True, p_rand=0.1), **kw)
    jobs = [(int(s), size, kw) for s in seeds]
    if workers > 1:
        with mp.get_context(context).Pool(workers) as pool:
            res = pool.map(_build_fl, jobs, chunksize=max(1, len(jobs) // (workers * 8)))
    else:
        res = [_build_fl(j) for j in jobs]
    S = np.array([r[0] for r in res], np.int64)
    nz = np.array([len(r[2]) for r in res], np.int64)
    nz_off = np.concatenate([[0], np.cumsum(nz)])
    return dict(
        B=len(res), A=4, H=0, rewards_range=(0.0, 1.0),
        state_off=np.concatenate([[0], np.cumsum(S)]).astype(np.int64),
        csr_ptr=np.concatenate([r[1][:-1].astype(np.int64) + nz_off[i] for i, r in enumerate(res)] + [nz_off[-1:]]),
        csr_col=np.concatenate([r[2] for r in res]),
        csr_val=np.concatenate([r[3] for r in res]),
        R=np.concatenate([r[4] for r in res]),
    )
