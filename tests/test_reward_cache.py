"""Reference-exact stochastic rewards for BATCHES (CMDP_FLAG_REWARD_CACHE, csrc/cmdp_reward_cache.h).

CPU part: the sampler the library fills its reward caches with (`cmdp_legacy_beta`: MT19937 + numpy's legacy Beta) against
numpy's own `RandomState.beta`, draw for draw -- the call `scipy.stats.beta(a, b).rvs(5000, random_state=mdp._rng)` of
BaseMDP.sample_reward (colosseum/mdp/base.py:1196-1203) ends in.

GPU part: the reference's own outputs through the batched path -- golden G8 (trajectories with Beta rewards incl. cache
refills) through `BatchedMDP.rollout` / `.step`, golden G17 (the reference's MDPLoop + Q-learning agents on benchmark
MDPs with Beta rewards: rows, action streams, Q / N tables) through the device agents and the one-call logged loop."""
import ctypes as C
import json

import numpy as np
import pytest

from conftest import load_golden
from colosseum_amd import _lib as L
from colosseum_amd.mdp import make_model

# every Beta parameterisation of the default benchmark suites (G11) plus shapes on all branches of the legacy sampler:
# Johnk (a, b <= 1), exponential (shape 1), Ahrens-Dieter (shape < 1), Marsaglia-Tsang (shape > 1)
PARAMS = [(1.0, 25.0), (1.0, 250.0), (12.0, 1.0), (1.0, 159.0), (15.0, 1.0), (1.0, 99.0), (99.0, 1.0), (35.0, 1.0),
          (1.0, 0.11), (1.0, 4.0), (0.5, 0.11), (1.5, 4.0), (0.3, 0.11), (2.0, 4.0), (1.0, 790.0), (39.0, 1.0), (224.0, 1.0),
          (0.7, 0.7), (0.2, 3.0), (3.0, 0.2), (1.0, 1.0), (24.0, 1.0), (1.0, 249.0), (0.05, 0.05), (7.5, 0.9)]


def _lib_beta(state, a, b, n):
    """(samples, new state) -- `state` as RandomState.get_state()."""
    key = np.ascontiguousarray(state[1], np.uint32).copy()
    pos, has, cached = C.c_int32(int(state[2])), C.c_int32(int(state[3])), C.c_double(float(state[4]))
    out = np.zeros(n)
    L.check(L.load().cmdp_legacy_beta(L.ptr(key), C.byref(pos), C.byref(has), C.byref(cached), float(a), float(b), n, L.ptr(out)))
    return out, ("MT19937", key, pos.value, has.value, cached.value)


def test_legacy_beta_equals_numpy_draw_for_draw():
    for i, (a, b) in enumerate(PARAMS):
        rs = np.random.RandomState(1000 + i)
        rs.rand(7 * i)  # an arbitrary position inside the first block
        want = rs.__class__(0)
        want.set_state(rs.get_state())
        ref = want.beta(a, b, 5000)
        got, st = _lib_beta(rs.get_state(), a, b, 5000)
        np.testing.assert_array_equal(got, ref, err_msg=f"Beta({a}, {b})")
        ws = want.get_state()
        assert st[2] == ws[2] and st[3] == ws[3] and st[4] == ws[4] and np.array_equal(st[1], ws[1]), (a, b)


def test_legacy_beta_blocks_interleaved_on_one_stream():
    """What an MDP's stream sees: blocks of 5000 draws of DIFFERENT distributions one after the other, the polar
    Gaussian's cached second variate carried from one block into the next."""
    rs = np.random.RandomState(77)
    state = rs.get_state()
    order = [PARAMS[k % len(PARAMS)] for k in (3, 11, 0, 13, 12, 5, 9, 2, 13, 13, 8, 21)]
    for a, b in order:
        ref = rs.beta(a, b, 5000)
        got, state = _lib_beta(state, a, b, 5000)
        np.testing.assert_array_equal(got, ref, err_msg=f"Beta({a}, {b})")
    ws = rs.get_state()
    assert state[2] == ws[2] and state[3] == ws[3] and state[4] == ws[4] and np.array_equal(state[1], ws[1])


def test_legacy_beta_rejects_bad_arguments():
    rs = np.random.RandomState(0).get_state()
    with pytest.raises(L.CmdpError):
        _lib_beta(rs, 0.0, 1.0, 4)
    with pytest.raises(L.CmdpError):
        _lib_beta(("MT19937", rs[1], 700, 0, 0.0), 1.0, 1.0, 4)


def test_builder_snapshots_the_reward_stream():
    m = make_model("DeepSeaEpisodic", seed=2, size=6, p_rand=0.3, make_reward_stochastic=True)
    st = m.extra["rng_state"]
    assert st[0] == "MT19937" and len(st[1]) == 624 and 0 <= st[2] <= 624
    a = np.random.RandomState(0)
    a.set_state(st)
    # the snapshot is where the live generator stood at the end of construction
    np.testing.assert_array_equal(a.rand(5), m.extra["rng"].rand(5))


# ---- GPU ---------------------------------------------------------------------------------------------------------------
def _kwargs(c):
    kw = dict(c["kwargs"])
    if "rewards_range" in kw:
        kw["rewards_range"] = tuple(kw["rewards_range"])
    return kw


@pytest.mark.gpu
def test_repositioned_reward_streams_start_from_empty_caches(need_gpu):
    """cmdp_set_reward_streams called again = a new run on a fresh copy of every MDP: the blocks already installed on the device
    (and their read positions) must not be served any more -- a fresh BaseMDP starts with empty caches
    (colosseum/mdp/base.py:1187-1207).  Deterministic dynamics, Beta rewards: a first leg of whole episodes leaves the
    instances in their reset state with half-used caches on the device; after repositioning the streams to their initial
    states a second leg must equal what a handle that never ran gives, reward for reward.  A handle built from bare tables
    with CMDP_FLAG_REWARD_CACHE and no streams refuses to step BEFORE anything is stepped."""
    from colosseum_amd.batched import BatchedMDP, tables_from_models

    ms = [make_model("DeepSeaEpisodic", seed=s, size=6, make_reward_stochastic=True) for s in (3, 4, 5)]
    H = ms[0].H
    rs = np.random.RandomState(11)
    n1, n2 = 1000 * H, 777 * H + 3
    a1 = rs.randint(0, 2, (n1, 3)).astype(np.int8)
    a2 = rs.randint(0, 2, (n2, 3)).astype(np.int8)
    states = [m.extra["rng_state"] for m in ms]
    env = BatchedMDP(ms, rng_mode=L.RNG_MT_COMPAT, flags=L.FLAG_REWARD_CACHE, with_dp=False)
    env.reset()
    env.rollout(n1, a1)
    assert env.reward_cache_stats()["fills"] > 0
    cur, h, _ = env.state()
    assert (h == 0).all() and (cur == ms[0].start_states[0]).all()   # whole episodes: back in the reset state
    env.set_reward_streams(states)                                    # a new run: streams at their initial states, EMPTY caches
    x = env.rollout(n2, a2, trace=True)
    fresh = BatchedMDP(ms, rng_mode=L.RNG_MT_COMPAT, flags=L.FLAG_REWARD_CACHE, with_dp=False)
    fresh.reset()
    y = fresh.rollout(n2, a2, trace=True)
    np.testing.assert_array_equal(x["obs"], y["obs"])
    np.testing.assert_array_equal(x["rew"], y["rew"])
    np.testing.assert_array_equal(x["reward_sum"], y["reward_sum"])
    env.close()
    fresh.close()
    # no streams: refused before anything moves
    t = tables_from_models(ms[:2], True, False)
    env = BatchedMDP(tables=t, rng_mode=L.RNG_MT_COMPAT, flags=L.FLAG_REWARD_CACHE)
    env.reset()
    before = [a.copy() for a in env.state()]
    with pytest.raises(L.CmdpError) as ei:
        env.rollout(50, np.zeros((50, 2), np.int8))
    assert ei.value.code == L.ERR_INVALID and "cmdp_set_reward_streams" in str(ei.value)
    for a, b in zip(before, env.state()):
        np.testing.assert_array_equal(a, b)
    vs, vsa = env.visits()
    assert vs.sum() == 2 and vsa.sum() == 0      # the two resets, no transition
    env.set_reward_streams([m.extra["rng_state"] for m in ms[:2]])
    env.rollout(50, np.zeros((50, 2), np.int8))   # and the handle is usable afterwards
    env.close()


@pytest.mark.gpu
def test_g8_through_batched_rollout_and_step(need_gpu):
    """The reference's Beta-reward trajectories (7 000 steps, caches refilled) through the BATCHED path: every case three
    times in one batch (instances park and resume independently), observations / rewards / step types bit-equal."""
    from colosseum_amd.batched import BatchedMDP

    z, cases = load_golden("G8_stochastic_rewards")
    for i, c in enumerate(cases):
        k = f"c{i}_"
        m = make_model(c["cls"], **_kwargs(c))
        acts = z[k + "actions"].astype(np.int8)
        n = len(acts)
        env = BatchedMDP([m, m, m], rng_mode=L.RNG_MT_COMPAT, flags=L.FLAG_REWARD_CACHE, with_dp=False)
        assert (env.reset() == z[k + "resets"][0]).all()
        out = env.rollout(n, np.repeat(acts[:, None], 3, 1), trace=True)
        for b in range(3):
            np.testing.assert_array_equal(out["obs"][:, b], z[k + "obs"], err_msg=str(c))
            np.testing.assert_array_equal(out["stype"][:, b], z[k + "stype"])
            np.testing.assert_array_equal(out["rew"][:, b], z[k + "rew"], err_msg=str(c))
        st = env.reward_cache_stats()
        assert st["fills"] % 3 == 0 and st["fills"] >= 3 and st["rounds"] >= st["fills"] // 3
        # the sum of the call accumulates across the relaunches in transition order
        acc = 0.0
        for r in z[k + "rew"].tolist():
            acc += r
        assert (out["reward_sum"] == acc).all()
        env.close()
        # per-call API: cmdp_step with explicit resets, two launch lengths mixed with a rollout in between
        env = BatchedMDP([m, m], rng_mode=L.RNG_MT_COMPAT, flags=L.FLAG_REWARD_CACHE, with_dp=False)
        env.reset()
        ri, t = 1, 0
        while t < 600:
            obs, rew, ty = env.step(np.full(2, acts[t]))
            assert (obs == z[k + "obs"][t]).all() and (ty == z[k + "stype"][t]).all(), (c, t)
            assert (rew == z[k + "rew"][t]).all(), (c, t, rew, z[k + "rew"][t])
            if ty[0] == 2:
                assert (env.reset() == z[k + "resets"][ri]).all()
                ri += 1
            t += 1
        env.close()


@pytest.mark.gpu
def test_philox_transitions_with_reference_reward_caches(need_gpu):
    """The flag is independent of the transition streams: Philox dynamics + reward caches (random policy on the device);
    two copies with the same key draw the same trajectory and the same rewards, with other keys other rewards."""
    from colosseum_amd.batched import BatchedMDP

    m = make_model("FrozenLakeContinuous", seed=1, size=4, p_frozen=0.9, p_lazy=0.05, make_reward_stochastic=True)
    env = BatchedMDP([m, m, m], rng_mode=L.RNG_PHILOX, philox_keys=[5, 5, 6], flags=L.FLAG_REWARD_CACHE, with_dp=False)
    env.reset()
    out = env.rollout(4000, None, trace=True)
    np.testing.assert_array_equal(out["rew"][:, 0], out["rew"][:, 1])
    np.testing.assert_array_equal(out["obs"][:, 0], out["obs"][:, 1])
    assert not np.array_equal(out["obs"][:, 0], out["obs"][:, 2])
    assert 0.0 < out["rew"].min() and out["rew"].max() < 1.0
    env.close()
    with pytest.raises(L.CmdpError):  # no streams handed over: the first parked instance makes the call fail loudly
        from colosseum_amd.batched import tables_from_models

        env = BatchedMDP(tables=tables_from_models([m], with_dp=False), rng_mode=L.RNG_MT_COMPAT, flags=L.FLAG_REWARD_CACHE)
        env.reset()
        env.rollout(10, np.zeros((10, 1), np.int8))


@pytest.mark.gpu
def test_g17_device_agents_on_beta_reward_mdps(need_gpu):
    """Golden G17: the reference's MDPLoop + Q-learning on benchmark MDPs with Beta rewards.  The device agents with the
    reference-exact reward caches reproduce the action stream, the rewards, the final Q / N tables (bit-equal) and the
    logger rows; the instance is replicated in the batch, with the replicas' agents seeded differently so that they park
    at different steps."""
    from colosseum_amd.agents import BatchedQLearningContinuous, BatchedQLearningEpisodic
    from colosseum_amd.batched import BatchedMDP
    from colosseum_amd.experiment.batched_loop import BatchedContinuousLoop, BatchedEpisodicLoop

    z, cases = load_golden("G17_mdploop_beta_rewards")
    for i, c in enumerate(cases):
        k = f"c{i}_"
        kw = dict(c["mdp_kwargs"])
        for key in ("optimal_distribution", "other_distribution", "sub_optimal_distribution"):
            if key in kw and kw[key] is not None:
                kw[key] = tuple(kw[key][:1]) + (tuple(kw[key][1]),)
        m = make_model(c["mdp_cls"], **kw)
        akw = dict(c["agent_kwargs"])
        seed = akw.pop("seed")
        episodic = c["agent"] == "QLearningEpisodic"
        Agent = BatchedQLearningEpisodic if episodic else BatchedQLearningContinuous
        Loop = BatchedEpisodicLoop if episodic else BatchedContinuousLoop
        T = c["T"]
        # (i) the logged loop as one library call: rows of the reference
        env = BatchedMDP([m, m, m], rng_mode=L.RNG_MT_COMPAT, flags=L.FLAG_REWARD_CACHE)
        ag = Agent(env, [seed, seed + 100, seed], **akw)
        loop = Loop(env, ag)
        rows = loop.run(T=T, log_every=c["log_every"])
        assert int(loop.last_training_step[0]) == c["last_training_step"]
        assert bool(loop.vt.is_training[0]) == c["is_training_at_end"]
        for b in (0, 2):
            assert len(rows[b]) == len(c["rows"])
            for got, ref in zip(rows[b], c["rows"]):
                for name, v in ref.items():
                    assert float(got[name]) == pytest.approx(v, rel=2e-6, abs=2e-5), (c["mdp_cls"], kw, name, got["steps"])
        st = env.reward_cache_stats()
        assert st["fills"] >= 2 * c["visited_triples"]
        ag.close()
        env.close()
        # (ii) plain runs with the training schedule of the reference loop: actions, rewards and tables bit-equal
        env = BatchedMDP([m, m], rng_mode=L.RNG_MT_COMPAT, flags=L.FLAG_REWARD_CACHE, with_dp=False)
        env.reset()
        ag = Agent(env, [seed + 7, seed], **akw)
        n_train = T if c["is_training_at_end"] else None
        if n_train is None:
            # MDPLoop froze training at a logging step: the updates stop right after the step it was logged at
            N = z[k + "N_final"]
            n_train = int(N.sum() - (N.size if episodic else 0))
        out1 = ag.run(n_train, train=True, trace_actions=True)
        Q, N = ag.tables()
        np.testing.assert_array_equal(N[1], z[k + "N_final"])
        np.testing.assert_array_equal(Q[1], z[k + "Q_final"])
        out2 = ag.run(T - n_train, train=False, trace_actions=True)
        acts = np.concatenate([out1["actions"][:, 1], out2["actions"][:, 1]])
        np.testing.assert_array_equal(acts, z[k + "actions"], err_msg=str(kw))
        assert out2["cumulative_reward"][1] == c["reward_sum"]
        ag.close()
        env.close()
