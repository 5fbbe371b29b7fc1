"""K9 (cmdp_average_reward): average reward of deterministic policies, fully on the device, against the host
restatement of colosseum/mdp/utils/markov_chain.py:12-136 (recurrent classes by scipy + networkx-order DFS, GTH by the
dense kernel K7): same value, same numpy type, same number of recurrent classes."""
import os

import numpy as np
import pytest

from colosseum_amd import _lib as L
from colosseum_amd.batched import BatchedMDP
from colosseum_amd.markov_chain import (get_average_reward, get_transition_probabilities, recurrent_classes)
from colosseum_amd.mdp import make_model

pytestmark = pytest.mark.gpu

CASES = [
    ("FrozenLakeContinuous", dict(seed=3, size=5, p_frozen=0.8)),
    ("DeepSeaContinuous", dict(seed=1, size=8)),
    ("DeepSeaContinuous", dict(seed=2, size=12, p_rand=0.1)),
    ("MiniGridEmptyContinuous", dict(seed=4, size=6, n_starting_states=2)),
    ("MiniGridRoomsContinuous", dict(seed=5, room_size=3, n_rooms=4)),
    ("FrozenLakeContinuous", dict(seed=6, size=12, p_frozen=0.9)),
]


@pytest.mark.parametrize("cls,kw", CASES)
def test_average_reward_kernel_matches_host_restatement(need_gpu, cls, kw):
    m = make_model(cls, **kw)
    S, A = m.n_states, m.n_actions
    T, R = m.dense()
    rng = np.random.default_rng(S)
    n_pol = 24
    acts, starts = [], []
    for i in range(n_pol):
        if i % 3 == 0:   # fully random action per state: usually several recurrent classes
            a = rng.integers(0, A, S)
        elif i % 3 == 1:  # one action nearly everywhere
            a = np.full(S, rng.integers(0, A))
            a[rng.integers(0, S, max(1, S // 10))] = rng.integers(0, A)
        else:            # greedy w.r.t. a noisy reward table
            a = (R + rng.normal(size=R.shape) * 0.3).argmax(1)
        acts.append(a.astype(np.int32))
        starts.append(int(rng.integers(0, S)))
    env = BatchedMDP([m] * n_pol, rng_mode=L.RNG_PHILOX, with_env=False)
    fast, ncls_fast = env.average_reward(acts, starts)      # default: wave butterfly sums inside the elimination
    env.set_option(L.OPT_CHAIN_EXACT_ORDER, 1)              # the reference's summation order: bit-equal below
    vals, ncls = env.average_reward(acts, starts)
    np.testing.assert_array_equal(ncls_fast, ncls)
    for x, y in zip(fast, vals):
        assert type(x) is type(y) and x == pytest.approx(y, rel=1e-6 if isinstance(y, np.float32) else 1e-12, abs=1e-15)
    multi = 0
    for i in range(n_pol):
        pol = np.zeros((S, A), np.float32)
        pol[np.arange(S), acts[i]] = 1
        want = get_average_reward(T, R, pol, [(starts[i], 1.0)])
        nc = len(recurrent_classes(get_transition_probabilities(T, pol)))
        multi += nc > 1
        assert ncls[i] == nc, (cls, i)
        assert type(vals[i]) is type(want), (cls, i, type(vals[i]), type(want), nc)
        assert vals[i] == want, (cls, i, vals[i], want, nc)
    # a masked call leaves the other instances alone and returns the same values
    mask = np.zeros(n_pol, bool)
    mask[::2] = True
    vals2, _ = env.average_reward(acts, starts, mask=mask)
    assert [vals2[i] for i in range(0, n_pol, 2)] == [vals[i] for i in range(0, n_pol, 2)]
    env.close()
    if cls.startswith("MiniGrid"):  # turning on the spot for ever: random policies split the grid into many classes
        assert multi >= 4


def test_fast_chain_kernel_on_irreducible_chains(need_gpu):
    """K9F (fill-reducing elimination order, rounds of independent pivots) on the chains it exists for -- the benchmark's
    continuous MiniGrid settings under greedy-like policies, where p_rand makes every policy's chain irreducible --
    against K9 in the reference's order (itself bit-equal to the host restatement, test above): same numpy type (float64:
    one class == the whole chain), same class count, values within 1e-11 relative (the SURVEY's tolerance for this row is
    1e-6; GTH is subtraction-free, so another elimination order moves the result by a few ulp only).  The statistic says
    the fast kernel really took them; a reducible chain in the same batch goes to K9."""
    import ctypes as C
    import json

    from conftest import GOLDEN

    cfg = json.load(open(os.path.join(GOLDEN, "G11_benchmark_configs.json")))["benchmark_continuous_ergodic"]["mdp_configs"]
    for cls, scope in (("MiniGridEmptyContinuous", "prms_3"), ("MiniGridEmptyContinuous", "prms_0"), ("MiniGridRoomsContinuous", "prms_0"),
                       ("FrozenLakeContinuous", "prms_0"), ("DeepSeaContinuous", "prms_0")):
        ms = [make_model(cls, seed=s, **cfg[cls][scope]) for s in range(3)]
        S, A = ms[0].n_states, ms[0].n_actions
        rng = np.random.default_rng(S)
        acts = [rng.integers(0, A, m.n_states).astype(np.int32) for m in ms]
        acts += [np.full(m.n_states, k % A, np.int32) for k, m in enumerate(ms)]
        starts = [int(rng.integers(0, m.n_states)) for m in ms] * 2
        env = BatchedMDP(ms + ms, rng_mode=L.RNG_PHILOX, with_env=False)
        fast, ncls_fast = env.average_reward(acts, starts)
        n_fast = C.c_double()
        L.check(L.load().cmdp_stat(env.handle, L.STAT_CHAIN_FAST_INSTANCES, C.byref(n_fast)))
        env.set_option(L.OPT_CHAIN_EXACT_ORDER, 1)
        vals, ncls = env.average_reward(acts, starts)
        np.testing.assert_array_equal(ncls_fast, ncls)
        # irreducible == one recurrent class that is the whole chain == the reference's float64 branch with one class
        irreducible = sum(1 for v, n in zip(vals, ncls) if n == 1 and type(v) is np.float64)
        assert int(n_fast.value) == irreducible >= 3, (cls, scope, n_fast.value, ncls, [type(v).__name__ for v in vals])
        for x, y in zip(fast, vals):
            assert type(x) is type(y) and x == pytest.approx(y, rel=1e-11, abs=1e-15), (cls, scope, x, y)
        # masked calls leave the others alone
        mask = np.array([1, 0, 1, 0, 1, 0], bool)
        env.set_option(L.OPT_CHAIN_EXACT_ORDER, 0)
        part, _ = env.average_reward(acts, starts, mask=mask)
        assert [part[i] for i in (0, 2, 4)] == [fast[i] for i in (0, 2, 4)]
        env.close()


def test_mixing_time_against_dense_float64_powers(need_gpu):
    """Build-defined measure (no reference counterpart): both device paths -- one sparse step at a time with the row in
    LDS, and matrix powers with the binary search on t (what config C5 runs) -- must give the t and the total variation
    of a plain numpy float64 evaluation; a periodic chain reports -1."""
    from colosseum_amd.hardness import mixing_time
    from colosseum_amd.markov_chain import gth_batch

    ms = [make_model("FrozenLakeContinuous", seed=3, size=5, p_frozen=0.8),
          make_model("MiniGridEmptyContinuous", seed=4, size=5, n_starting_states=2, p_lazy=0.2),
          make_model("FrozenLakeContinuous", seed=6, size=8, p_frozen=0.9),
          make_model("DeepSeaContinuous", seed=1, size=6),       # period = size under any policy
          make_model("MiniGridRoomsContinuous", seed=0, room_size=4, n_rooms=4, p_lazy=0.1)]   # t_mix in the hundreds: eight squarings, then the search
    t, tv = mixing_time(ms, threshold=0.25, max_steps=20000)
    wants = []
    for i, m in enumerate(ms):
        T, _ = m.dense()
        P = T.astype(np.float64).mean(1)
        P /= P.sum(1, keepdims=True)
        if i == 3:
            assert t[i] == -1
            wants.append(None)
            continue
        sd = gth_batch([P])[0]
        X = np.eye(m.n_states)
        want = None
        for step in range(1, 20001):
            X = X @ P
            d = 0.5 * np.abs(X - sd).sum(1).max()
            if d <= 0.25:
                want = (step, d)
                break
        assert want is not None and t[i] == want[0], (i, t[i], want)
        assert tv[i] == pytest.approx(want[1], rel=1e-10)
        wants.append(want)
    assert wants[4][0] > 200
    # the matrix-power path, forced, with as few S x S buffers as it can work with and with plenty
    for i in (0, 2, 3, 4):
        m = ms[i]
        T, _ = m.dense()
        P = T.astype(np.float64).mean(1)
        P /= P.sum(1, keepdims=True)
        env = BatchedMDP([m], with_env=False)
        env.set_option(L.OPT_MIXING_PATH, 1)
        sd = gth_batch([P])[0] if i != 3 else np.full(m.n_states, 1.0 / m.n_states)
        for buffers in ("3", "24"):  # 3: only two powers are ever held, the search ends in a long run of sparse steps
            os.environ["CMDP_MIX_MAX_BUFFERS"] = buffers
            tt, tvv = env.mixing_time([sd], threshold=0.25, max_steps=20000)
            del os.environ["CMDP_MIX_MAX_BUFFERS"]
            if i == 3:
                assert tt[0] == -1
            else:
                assert tt[0] == wants[i][0] and tvv[0] == pytest.approx(wants[i][1], rel=1e-9), (i, buffers, tt, wants[i])
        env.close()


def test_mixing_time_large_chain_path_at_scale(need_gpu):
    """Config C5's second half -- the mixing time of a chain too large for the row-in-LDS stepping path -- in the driver's
    GPU suite at reduced size (the full S = 50 272 run takes 103 s: tools/run_c5.py --mixing, profiles/).
    (i) MiniGridRooms 14 x 14 x 4 rooms (S = 3 152, above the 1 024-state switch to dense float64 matrix powers in HBM):
    t_mix and the total variation at t_mix against a plain numpy float64 evaluation by repeated squaring + binary search.
    (ii) room size 28 as in C5, 4 rooms (S = 12 560, 1.26 GB per power): properties the definition implies -- the total
    variation at t_mix is at most the threshold, and t_mix does not grow when the threshold is relaxed."""
    from colosseum_amd.hardness import mixing_time
    from colosseum_amd.markov_chain import gth_batch

    m = make_model("MiniGridRoomsContinuous", seed=0, room_size=14, n_rooms=4, n_starting_states=2, p_lazy=0.1)
    assert m.n_states == 3152
    t, tv = mixing_time([m], threshold=0.25, max_steps=1 << 20)
    T, _ = m.dense()
    P = T.astype(np.float64).mean(1)
    P /= P.sum(1, keepdims=True)
    sd = gth_batch([P])[0]

    def tvd(X):
        return 0.5 * np.abs(X - sd).sum(1).max()

    powers = [P]
    while tvd(powers[-1]) > 0.25:          # A_k = P^(2^k) until 2^K steps are mixed
        powers.append(powers[-1] @ powers[-1])
    K = len(powers) - 1
    assert K >= 8
    X, steps = None, 0                     # largest t (bit by bit) whose power is NOT yet mixed; t_mix = t + 1
    for k in range(K - 1, -1, -1):
        Y = powers[k] if X is None else X @ powers[k]
        if tvd(Y) > 0.25:
            X, steps = Y, steps + (1 << k)
    want_t = steps + 1
    want_tv = tvd(P if X is None else X @ P)
    assert int(t[0]) == want_t, (t, want_t)
    assert tv[0] == pytest.approx(want_tv, rel=1e-8)

    big = make_model("MiniGridRoomsContinuous", seed=0, room_size=28, n_rooms=4, n_starting_states=2, p_lazy=0.1)
    assert big.n_states == 12560
    t25, tv25 = mixing_time([big], threshold=0.25, max_steps=1 << 22)
    t40, tv40 = mixing_time([big], threshold=0.40, max_steps=1 << 22)
    assert 0 < tv25[0] <= 0.25 and 0 < tv40[0] <= 0.40
    assert 1000 < t40[0] <= t25[0]


def test_chain_api_argument_checks(need_gpu):
    """Error paths of the new entry points: they fail loudly with a library error, never with a GPU fault."""
    m = make_model("FrozenLakeContinuous", seed=3, size=5, p_frozen=0.8)
    S, A = m.n_states, m.n_actions
    env = BatchedMDP([m, m], rng_mode=L.RNG_PHILOX, with_env=False)
    good = [np.zeros(S, np.int32)] * 2
    with pytest.raises(L.CmdpError):
        env.average_reward([np.full(S, A, np.int32)] * 2, [0, 0])       # action out of range
    with pytest.raises(L.CmdpError):
        env.average_reward(good, [0, S])                               # start state out of range
    vals, ncls = env.average_reward(good, [0, S - 1])
    assert all(np.isfinite(v) for v in vals) and (ncls >= 1).all()
    with pytest.raises(L.CmdpError):
        env.diameter_range(5, 3)                                       # empty / reversed range
    with pytest.raises(L.CmdpError):
        env.diameter_range(0, 2 * S + 1)                               # beyond the flat state space
    assert env.diameter_range(4, 4).size == 0
    with pytest.raises(L.CmdpError):
        env.mixing_time([np.full(S, 1.0 / S)] * 2, threshold=0.0)      # threshold must be positive
    env.close()
    ep = make_model("DeepSeaEpisodic", seed=0, size=4)
    env = BatchedMDP([ep], rng_mode=L.RNG_PHILOX, with_env=False)
    with pytest.raises(L.CmdpError):                                   # average rewards are a continuous-setting notion
        env.average_reward([np.zeros(ep.n_states, np.int32)], [0])
    env.close()
