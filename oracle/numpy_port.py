"""numpy restatement of the hot path -- TEST / BENCH INFRASTRUCTURE ONLY (never imported by colosseum_amd/).

SURVEY 8(d) asks for a pure-numpy CPU figure beside the C oracle: numpy is how the reference itself executes
(`BaseMDP.step` is per-step Python over numpy / dict objects, colosseum/mdp/base.py:1279-1317; the Jacobi sweep is
`R + gamma * (T @ V)` on a `sparse.COO`, colosseum/dynamic_programming/infinite_horizon.py:145-164).  Two forms:

* `rollout_vectorised`: the C2 job (deterministic dynamics, Philox random policy -- the SAME counter-based streams as the
  GPU kernels and the C oracle, so visit counts are comparable bit for bit) with one numpy operation per step over a
  whole slice of instances: what a numpy user would write to batch the reference.
* `step_loop_python`: one instance, one Python iteration per step, table look-ups per step -- the reference's execution
  model (its own per-step cost in this container is 5.8e4 steps/s; this loop has none of its
  dm_env / dict / sampler-object overhead and is an upper bound for that style).
* `jacobi_vi`: discounted value iteration, float32, in-order row sums, `diff < eps` in float64 -- the oracle's and the
  kernels' arithmetic (infinite_horizon.py:145-164), vectorised over the rows of ONE instance.

Each is checked against the C oracle in tests/test_numpy_port.py."""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_LO = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox-4x32-10 on uint32 arrays (Random123's rounds; csrc/cmdp_device.h:15-30)."""
    c0, c1, c2, c3 = (np.asarray(x, np.uint32) for x in (c0, c1, c2, c3))
    k0, k1 = np.asarray(k0, np.uint32).copy(), np.asarray(k1, np.uint32).copy()
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            h0, l0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _LO).astype(np.uint32)
            h1, l1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _LO).astype(np.uint32)
            c0, c1, c2, c3 = h1 ^ c1 ^ k0, l1, h0 ^ c3 ^ k1, l0
            k0 = k0 + _W0
            k1 = k1 + _W1
    return c0, c1, c2, c3


def random_actions(keys, n0, n_steps, A):
    """[n_steps, B] actions of transitions n0 .. n0+n_steps-1 (Philox domain 2, include/cmdp.h CMDP_RNG_PHILOX): for A in
    {2, 4, 16, 256} the PACKED stream -- block n // apb with apb = 128 / log2 A actions per block, action k = n % apb in bits
    [lg * (k % apw), +lg) of word k // apw (apw = 32 / lg); otherwise word (n & 3) of block n >> 2, (word * A) >> 32."""
    keys = np.asarray(keys, np.uint64)
    k0, k1 = (keys & _LO).astype(np.uint32), (keys >> np.uint64(32)).astype(np.uint32)
    n = np.arange(n0, n0 + n_steps, dtype=np.uint64)
    lg = {2: 1, 4: 2, 16: 4, 256: 8}.get(int(A), 0)
    apb = np.uint64(128 // lg if lg else 4)
    blocks = np.unique(n // apb)
    B, nb = len(keys), len(blocks)
    blk = np.repeat(blocks[:, None], B, 1)
    w = philox4x32_10((blk & _LO).astype(np.uint32), (blk >> np.uint64(32)).astype(np.uint32), np.full((nb, B), 2, np.uint32),
                      np.zeros((nb, B), np.uint32), np.broadcast_to(k0, (nb, B)), np.broadcast_to(k1, (nb, B)))
    words = np.stack(w, 1)  # [nb, 4, B]
    bi = (n // apb - blocks[0]).astype(np.int64)
    k = n % apb
    if lg:
        apw = np.uint64(32 // lg)
        sel = words[bi, (k // apw).astype(np.int64)].astype(np.uint64)  # [n_steps, B]
        sh = (np.uint64(lg) * (k % apw))[:, None]
        return ((sel >> sh) & np.uint64(A - 1)).astype(np.int64)
    sel = words[bi, k.astype(np.int64)]
    return ((sel.astype(np.uint64) * np.uint64(A)) >> np.uint64(32)).astype(np.int64)


def _det_tables(tables, b0, b1):
    t = tables
    A, H = int(t["A"]), int(t["H"])
    so = np.asarray(t["state_off"], np.int64)
    S = int(so[b0 + 1] - so[b0])
    assert (np.diff(so[b0:b1 + 1]) == S).all(), "equal state counts"
    r0, r1 = so[b0] * A, so[b1] * A
    ptr = np.asarray(t["sp_ptr"], np.int64)
    assert (np.diff(ptr[r0:r1 + 1]) == 1).all(), "deterministic dynamics (one successor per row)"
    e0 = ptr[r0]
    nxt = np.asarray(t["sp_next"], np.int64)[e0:e0 + (r1 - r0)].reshape(b1 - b0, S * A)
    rew = np.asarray(t["sp_reward"], np.float64)[e0:e0 + (r1 - r0)].reshape(b1 - b0, S * A)
    st = np.asarray(t["start_off"], np.int64)
    assert (np.diff(st[b0:b1 + 1]) == 1).all(), "one start state"
    start = np.asarray(t["start_state"], np.int64)[st[b0]:st[b1]]
    lo, hi = (float(x) for x in t["rewards_range"])
    return A, H, S, nxt, rew * (hi - lo) - lo, start


def rollout_vectorised(tables, b0, b1, n_steps, philox_keys, chunk=4096):
    """reset() + n_steps transitions of instances [b0, b1) under the Philox random policy, episodic auto-reset:
    (last_obs, reward_sum, visits_s [B*S], visits_sa [B*S*A])."""
    A, H, S, nxt, rew, start = _det_tables(tables, b0, b1)
    B = b1 - b0
    keys = np.asarray(philox_keys, np.uint64)[b0:b1]
    ar = np.arange(B)
    cur = start.copy()
    h = np.zeros(B, np.int64)
    vs = np.zeros((B, S), np.int64)
    vsa = np.zeros((B, S * A), np.int64)
    rsum = np.zeros(B)
    vs[ar, cur] += 1
    for c0 in range(0, n_steps, chunk):
        acts = random_actions(keys, c0, min(chunk, n_steps - c0), A)
        for a in acts:
            row = cur * A + a
            cur = nxt[ar, row]
            rsum += rew[ar, row]
            h += 1
            vs[ar, cur] += 1
            vsa[ar, cur * A + a] += 1  # arrival state, action taken at the departure state (base.py:1302-1303)
            if H > 0:
                end = h >= H
                if end.any():
                    cur = np.where(end, start, cur)
                    h[end] = 0
                    vs[ar[end], cur[end]] += 1
    return cur.astype(np.int32), rsum, vs.ravel(), vsa.ravel()  # after a terminating step the fused loop has already reset


def step_loop_python(tables, b, n_steps, philox_key):
    """One instance, one Python iteration per step (the reference's execution model).  Returns (visits_s, reward_sum)."""
    A, H, S, nxt, rew, start = _det_tables(tables, b, b + 1)
    nxt, rew, s0 = nxt[0].tolist(), rew[0].tolist(), int(start[0])
    acts = random_actions(np.array([philox_key], np.uint64), 0, n_steps, A)[:, 0].tolist()
    vs = [0] * S
    cur, h, rsum = s0, 0, 0.0
    vs[cur] += 1
    for a in acts:
        row = cur * A + a
        cur = nxt[row]
        rsum += rew[row]
        h += 1
        vs[cur] += 1
        if H > 0 and h >= H:
            cur, h = s0, 0
            vs[cur] += 1
    return np.array(vs, np.int64), rsum


def jacobi_vi(ptr, col, val, R, S, A, gamma=0.99, eps=1e-6, max_sweeps=1_000_000):
    """`_discounted_value_iteration_sparse` (infinite_horizon.py:145-164) on the CSR of one instance: float32 products
    added one at a time in column order (`np.add.reduceat` would add a0 + (a1 + a2 + ...): numpy reduces the tail first),
    so the rows are padded to the longest one with (column 0, coefficient +0.0) -- `acc + 0 * v == acc` exactly -- and
    accumulated entry by entry; `diff < eps` in float64.  Returns (Q [S*A], V [S], sweeps)."""
    ptr = np.asarray(ptr, np.int64)
    n = np.diff(ptr)
    K = int(n.max())
    rows = np.repeat(np.arange(len(n)), n)
    pos = np.arange(len(col)) - np.repeat(ptr[:-1], n)
    colp = np.zeros((len(n), K), np.int64)
    valp = np.zeros((len(n), K), np.float32)
    colp[rows, pos] = np.asarray(col, np.int64)
    valp[rows, pos] = np.asarray(val, np.float32)
    R = np.asarray(R, np.float32).ravel()
    g = np.float32(gamma)
    V = np.zeros(S, np.float32)
    for sweep in range(1, max_sweeps + 1):
        tv = np.zeros(len(n), np.float32)
        for k in range(K):
            tv = tv + valp[:, k] * V[colp[:, k]]              # (T @ V), float32, entry by entry
        Q = R + g * tv                                        # gamma * (T @ V): the reference's precedence
        Vn = Q.reshape(S, A).max(1)
        diff = np.abs(V - Vn).max()
        V = Vn
        if float(diff) < eps:
            return Q, V, sweep
    raise RuntimeError("max sweeps")
