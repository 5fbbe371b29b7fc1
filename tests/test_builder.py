"""Host model builder vs the reference's own structures (golden fixtures G1-G4): state order, action
permutation, successor lists, probabilities, reward means, start distribution, dense T and R -- all exact."""
import numpy as np
import pytest

from conftest import load_golden
from colosseum_amd.batched import tables_from_models
from colosseum_amd.mdp import make_model
from colosseum_amd.mdp.fast_batch import deepsea_episodic_tables


@pytest.mark.parametrize("name", ["G1_deepsea8", "G2_deepsea30", "G3_stochastic", "G4_frozenlake20_vi", "G12_families"])
def test_structure_matches_reference(name):
    z, cases = load_golden(name)
    for i, c in enumerate(cases):
        m = make_model(c["cls"], **c["kwargs"])
        k = f"c{i}_"
        S, A, H = z[k + "SAH"]
        assert (m.n_states, m.n_actions, m.H) == (S, A, H), c
        np.testing.assert_array_equal(m.nodes, z[k + "nodes"])
        np.testing.assert_array_equal(m.sp_ptr, z[k + "sp_ptr"])
        np.testing.assert_array_equal(m.sp_next, z[k + "sp_next"])
        np.testing.assert_array_equal(m.sp_prob, z[k + "sp_prob"])
        np.testing.assert_array_equal(m.sp_rmean, z[k + "sp_rmean"])
        if k + "start_states" in z:
            np.testing.assert_array_equal(m.start_states, z[k + "start_states"])
            np.testing.assert_array_equal(m.start_probs, z[k + "start_probs"])
        T, R = m.dense()
        np.testing.assert_array_equal(R, z[k + "R"])
        if k + "T_idx" in z:
            nz = np.nonzero(T)
            np.testing.assert_array_equal(np.stack(nz), z[k + "T_idx"])
            np.testing.assert_array_equal(T[nz], z[k + "T_val"])
        extra = c.get("extra", {})
        fam = m.extra["family"]
        if "lake" in extra:
            assert ["".join(r) for r in fam.lake] == extra["lake"]
        if "goal_position" in extra:
            assert list(fam.goal_position) == extra["goal_position"]


def test_first_samples_of_every_sampler_match_reference():
    """The per-(s,a) sampler seeds are right: the first 8 draws of random.Random(seed).choices(...) equal the
    reference's cached_states (custom_samplers.py:55-57)."""
    import random

    for name in ("G3_stochastic", "G12_families"):
        _check_first_samples(*load_golden(name))


def _check_first_samples(z, cases):
    import random

    for i, c in enumerate(cases):
        m = make_model(c["cls"], **c["kwargs"])
        first = z[f"c{i}_sp_first"]
        for r in range(m.n_states * m.n_actions):
            lo, hi = m.sp_ptr[r], m.sp_ptr[r + 1]
            if hi - lo == 1:
                assert first[r, 0] == -1
                continue
            got = random.Random(int(m.sp_seed[r])).choices(m.sp_next[lo:hi].tolist(), weights=m.sp_prob[lo:hi].tolist(), k=8)
            assert got == first[r].tolist(), (c, r)
        sf = z[f"c{i}_start_first"]
        if len(sf):
            got = random.Random(m.start_seed).choices(m.start_states.tolist(), weights=m.start_probs.tolist(), k=len(sf))
            assert got == sf.tolist()


def test_deepsea_fast_batch_equals_generic_builder():
    seeds = [0, 1, 5, 123, 40000]
    for size in (5, 12):
        fast = deepsea_episodic_tables(seeds, size, with_dp=True)
        ref = tables_from_models([make_model("DeepSeaEpisodic", seed=s, size=size) for s in seeds])
        for k in ref:
            if k == "sp_seed":  # never read for deterministic rows
                continue
            np.testing.assert_array_equal(np.asarray(ref[k]), np.asarray(fast[k]), err_msg=k)


def test_constructor_checks():
    with pytest.raises(AssertionError):
        make_model("DeepSeaEpisodic", seed=0, size=5, p_lazy=0.1)  # no lazy mechanic for DeepSea
    with pytest.raises(KeyError):
        make_model("RandomWalkEpisodic", seed=0, size=5)  # not a reference family
    with pytest.raises(NotImplementedError):
        make_model("DeepSeaEpisodic", seed=0)


def test_recurrent_class_order_is_networkx_attracting_components_order():
    """markov_chain.py:95 of the reference lists the recurrent classes in networkx's `attracting_components` order and
    keeps the FIRST one the start state reaches: the order is part of the result (networkx 3.4.2 is the checker here)."""
    nx = pytest.importorskip("networkx")
    from colosseum_amd.markov_chain import recurrent_classes

    rng = np.random.default_rng(5)
    seen_multi = 0
    for trial in range(300):
        n = int(rng.integers(3, 40))
        tps = np.zeros((n, n), np.float32)
        for s in range(n):  # sparse random chain: 1-3 successors, biased to nearby states so that several classes form
            k = int(rng.integers(1, 4))
            succ = np.clip(s + rng.integers(-3, 4, k), 0, n - 1)
            for j in succ:
                tps[s, j] += 1.0 / k
        want = [sorted(c) for c in nx.attracting_components(nx.DiGraph(tps))]
        got = [c.tolist() for c in recurrent_classes(tps)]
        assert got == want, trial
        seen_multi += len(want) > 1
    assert seen_multi > 50


def _custom_model(z, k, c):
    T0 = {int(a): float(b) for a, b in zip(z[k + "in_T0k"], z[k + "in_T0v"])}
    return make_model(c["cls"], T_0=T0, T=z[k + "in_T"], R=z[k + "in_R"], **c["kwargs"])


def test_custom_mdp_matches_reference():
    """CustomMDP (golden G13: the reference built from a dict of deterministic reward distributions; here from the mean
    matrix, which is what the reference reduces the dict to): structure, sampler seeds, and through the oracle the
    4 000-step reference trajectories."""
    import random

    from oracle import oracle as O

    z, cases = load_golden("G13_custom")
    for i, c in enumerate(cases):
        k = f"c{i}_"
        m = _custom_model(z, k, c)
        S, A, H = z[k + "SAH"]
        assert (m.n_states, m.n_actions, m.H) == (S, A, H), c
        for name in ("nodes", "sp_ptr", "sp_next", "sp_prob", "sp_rmean", "start_states", "start_probs"):
            np.testing.assert_array_equal(getattr(m, name), z[k + name], err_msg=name)
        # (the reference's `T`, `R` properties return the USER's arrays, indexed by node ID and raw action
        #  (custom_mdp.py:213), not the matrices of the graph it simulates -- state indices follow the DFS and actions the
        #  per-state permutation; the sampler tables above are the dynamics, and they are equal)
        np.testing.assert_array_equal(z[k + "R"], z[k + "in_R"].astype(np.float32))
        first = z[k + "sp_first"]
        for r in range(S * A):
            lo, hi = m.sp_ptr[r], m.sp_ptr[r + 1]
            if hi - lo > 1:
                got = random.Random(int(m.sp_seed[r])).choices(m.sp_next[lo:hi].tolist(), weights=m.sp_prob[lo:hi].tolist(), k=8)
                assert got == first[r].tolist()
        e = O.OracleEnv(m, rng_mode=0)
        assert e.reset() == z[k + "resets"][0]
        out = e.rollout(len(z[k + "actions"]), z[k + "actions"])
        np.testing.assert_array_equal(out["obs"], z[k + "obs"])
        np.testing.assert_array_equal(out["rew"], z[k + "rew"])
        vs, vsa = e.visits()
        np.testing.assert_array_equal(vs, z[k + "visits_s"])
