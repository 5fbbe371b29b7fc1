"""The numpy restatement bench.py times beside the C oracle (oracle/numpy_port.py, SURVEY 8(d)'s "pure-numpy
restatement of a1 / a6") against the C oracle: same Philox streams, bit-equal visit counts, reward sums and values."""
import numpy as np

from colosseum_amd.mdp import make_model
from colosseum_amd.mdp.fast_batch import deepsea_episodic_tables
from oracle import numpy_port as NP
from oracle import oracle as O


def test_philox_and_action_stream_equal_the_oracle():
    for c, k in (((0, 0, 0, 0), (0, 0)), ((0xffffffff,) * 4, (0xffffffff, 0xffffffff)), ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0))):
        got = NP.philox4x32_10(*[np.array([x], np.uint32) for x in c], *[np.array([x], np.uint32) for x in k])
        np.testing.assert_array_equal(np.array([g[0] for g in got]), O.philox(c, k))


def test_vectorised_rollout_equals_the_c_oracle():
    seeds = np.arange(24)
    tables = deepsea_episodic_tables(seeds, 9)
    keys = (seeds * 7919 + 3).astype(np.uint64)
    for n in (1, 8, 9, 2001):
        last, rsum, vs, vsa = NP.rollout_vectorised(tables, 3, 21, n, keys)
        olast, orsum, ovs, ovsa = O.batch_rollout(tables, 3, 21, n, rng_mode=1, philox_keys=keys, want_visits=True)
        np.testing.assert_array_equal(last, olast)
        np.testing.assert_array_equal(vs, ovs)
        np.testing.assert_array_equal(vsa, ovsa)
        np.testing.assert_array_equal(rsum, orsum)
    vs1, r1 = NP.step_loop_python(tables, 5, 3000, int(keys[5]))
    _, orsum, ovs, _ = O.batch_rollout(tables, 5, 6, 3000, rng_mode=1, philox_keys=keys, want_visits=True)
    np.testing.assert_array_equal(vs1, ovs)
    assert r1 == orsum[0]


def test_numpy_jacobi_vi_equals_the_c_oracle():
    for seed in (0, 3):
        m = make_model("FrozenLakeContinuous", seed=seed, size=8, p_frozen=0.9, p_rand=0.1)
        ptr, col, val = m.csr()
        Q, V, sw = NP.jacobi_vi(ptr, col, val, m.reward_matrix(), m.n_states, m.n_actions, 0.99, 1e-6)
        oQ, oV, oit, _ = O.vi_discounted(m.n_states, m.n_actions, m.csr(), m.reward_matrix(), 0.99, 1e-6, 1)
        assert sw == oit
        np.testing.assert_array_equal(V, oV)
        np.testing.assert_array_equal(Q, oQ.ravel())
