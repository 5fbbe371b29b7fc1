"""the batched on-device Q-learning agent against the reference.  Golden G7 holds the action stream and
the final Q / N tables of the reference's QLearningEpisodic driven by the reference's MDPLoop; the device agent (fused
select_action -> step -> step_update kernel) must reproduce them bit for bit, including the steps after MDPLoop froze
training.  A second test runs a batch of different instances and seeds against the numpy restatement of the agent
(tests/helpers_agents.py, itself pinned to G7) stepping the CPU oracle environment."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_csv_matches
from colosseum_amd import _lib as L
from colosseum_amd import timestep as ts_
from colosseum_amd.agents import BatchedQLearningEpisodic
from colosseum_amd.batched import BatchedMDP
from colosseum_amd.mdp import make_model
from helpers_agents import QLearningContinuous, QLearningEpisodic
from oracle import oracle as O

pytestmark = pytest.mark.gpu


class _Spec:
    def __init__(self, m):
        self.time_horizon = m.H
        self.observations = type("o", (), {"num_values": m.n_states})()
        self.actions = type("a", (), {"num_values": m.n_actions})()


class _SpecC(_Spec):
    def __init__(self, m):
        self.time_horizon = np.inf
        self.observations = type("o", (), {"num_values": m.n_states})()
        self.actions = type("a", (), {"num_values": m.n_actions})()


def _host_run(m, kw, n_steps, rng_mode=0, key=0):
    """helper agent + oracle env; returns (actions, Q, N)"""
    e = O.OracleEnv(m, rng_mode=rng_mode, philox_key=key)
    ag = QLearningEpisodic(mdp_specs=_Spec(m), **kw)
    ts, h, acts = ts_.restart(e.reset()), 0, []
    for _ in range(n_steps):
        a = int(ag.select_action(ts, h))
        acts.append(a)
        ty, o, r, _ = e.step(a)
        nts = ts_.termination(r, -1) if ty == 2 else ts_.transition(r, o)
        ag.step_update(ts, a, nts, h)
        h, ts = h + 1, nts
        if ty == 2:
            ts, h = ts_.restart(e.reset()), 0
    return np.array(acts), ag.Q, ag.N


def test_device_qlearning_reproduces_reference_run(need_gpu):
    cases = json.load(open(os.path.join(GOLDEN, "G7_mdploop_qlearning.json")))
    for c in cases:
        m = make_model(c["mdp_cls"], **c["mdp_kwargs"])
        env = BatchedMDP([m], rng_mode=L.RNG_MT_COMPAT, with_dp=False)
        env.reset()
        kw = dict(c["agent_kwargs"])
        seed = kw.pop("seed")
        ag = BatchedQLearningEpisodic(env, [seed], **kw)
        n_train = c["n_updates"]
        assert n_train <= c["T"]
        a1 = ag.run(n_train, train=True, trace_actions=True)["actions"][:, 0]
        Q, N = ag.tables()
        np.testing.assert_array_equal(N[0], np.asarray(c["N_final"], np.int32))
        np.testing.assert_array_equal(Q[0].astype(np.float64), np.asarray(c["Q_final"]))  # bit-equal float32 tables
        a2 = ag.run(c["T"] - n_train, train=False, trace_actions=True)["actions"][:, 0]
        np.testing.assert_array_equal(np.concatenate([a1, a2]), np.asarray(c["actions"], np.int8), err_msg=str(c["mdp_kwargs"]))
        ag.close()
        env.close()


def test_device_qlearning_batch_vs_numpy_agent(need_gpu):
    specs = [("DeepSeaEpisodic", dict(seed=s, size=sz, p_rand=pr)) for s, sz, pr in ((0, 5, None), (1, 7, 0.3), (2, 7, 0.05))]
    specs += [("FrozenLakeEpisodic", dict(seed=4, size=4, p_frozen=0.9, p_rand=0.1, H=12))]
    for ucb, hp in (("bernstein", dict(p=0.05, c_1=0.9415278732894797, c_2=0.013873778519317169, min_at=0.07263563483119442)),
                    ("hoeffding", dict(p=0.1, c_1=0.01, min_at=0.0)),
                    ("bernstein", dict(p=0.05, c_1=0.2, c_2=0.5, min_at=0.3))):
        for cls, kw in specs:
            base = make_model(cls, **kw)
            ms = [base, base, base]
            seeds = [11, 12, 13]
            keys = np.array([101, 102, 103], np.uint64)
            env = BatchedMDP(ms, rng_mode=L.RNG_PHILOX, philox_keys=keys, with_dp=False)
            env.reset()
            ag = BatchedQLearningEpisodic(env, seeds, optimization_horizon=3000, UCB_type=ucb, **hp)
            out = ag.run(3000, train=True, trace_actions=True)
            Q, N = ag.tables()
            for i in range(3):
                acts, hQ, hN = _host_run(base, dict(seed=seeds[i], optimization_horizon=3000, UCB_type=ucb, **hp), 3000,
                                         rng_mode=1, key=int(keys[i]))
                np.testing.assert_array_equal(out["actions"][:, i], acts, err_msg=f"{cls} {kw} {ucb} inst {i}")
                np.testing.assert_array_equal(N[i], hN)
                np.testing.assert_array_equal(Q[i], hQ)
            ag.close()
            env.close()


def _assert_same_rows(a, b):
    """Two runs of a batched loop: every logged value identical in value AND numpy type (steps_per_second is wall clock)."""
    assert len(a) == len(b)
    for ta, tb in zip(a, b):
        assert len(ta) == len(tb)
        for ra, rb in zip(ta, tb):
            assert set(ra) == set(rb)
            for k in ra:
                if k != "steps_per_second":
                    assert ra[k] == rb[k] and type(ra[k]) is type(rb[k]), (k, ra["steps"], ra[k], rb[k])


def test_batched_episodic_loop_matches_reference_logger_rows(need_gpu):
    """The whole MDPLoop.run on the device for a batch: golden G7 rows (17 deterministic indicators per logging step)
    for each reference run, here executed with the instance replicated three times in one batch."""
    from colosseum_amd.experiment.batched_loop import BatchedEpisodicLoop

    cases = json.load(open(os.path.join(GOLDEN, "G7_mdploop_qlearning.json")))
    for c in cases:
        m = make_model(c["mdp_cls"], **c["mdp_kwargs"])
        env = BatchedMDP([m, m, m], rng_mode=L.RNG_MT_COMPAT)
        kw = dict(c["agent_kwargs"])
        seed = kw.pop("seed")
        ag = BatchedQLearningEpisodic(env, [seed, seed, seed], **kw)
        loop = BatchedEpisodicLoop(env, ag)
        rows = loop.run(T=c["T"], log_every=c["log_every"])
        assert (loop.last_training_step == -1).all()
        for b in range(3):  # the batched runner's CSV text against the file the reference's own CSVLogger wrote
            assert_csv_matches(loop.vt.log.csv_text(b), c["csv_text"])
        for inst in rows:
            assert len(inst) == len(c["rows"])
            for got, ref in zip(inst, c["rows"]):
                for k, v in ref.items():
                    assert float(got[k]) == pytest.approx(v, rel=2e-6, abs=2e-5), (c["mdp_kwargs"], k, got["steps"])
        Q, N = ag.tables()
        np.testing.assert_array_equal(Q[2].astype(np.float64), np.asarray(c["Q_final"]))
        ag.close()
        env.close()
        # the default above is ONE library call for the whole run; with Python in the loop the rows must be identical
        env = BatchedMDP([m, m, m], rng_mode=L.RNG_MT_COMPAT)
        ag = BatchedQLearningEpisodic(env, [seed, seed, seed], **kw)
        loop2 = BatchedEpisodicLoop(env, ag)
        loop2.native = False
        _assert_same_rows(rows, loop2.run(T=c["T"], log_every=c["log_every"]))
        np.testing.assert_array_equal(loop2.vt.is_training, loop.vt.is_training)
        ag.close()
        env.close()


def test_device_continuous_qlearning_and_batched_loop(need_gpu):
    """Continuous setting: device QLearningContinuous (float64 tables, as numpy >= 2 makes them) against the reference
    run (golden G10): action streams, and the logger rows through BatchedContinuousLoop."""
    from colosseum_amd.agents import BatchedQLearningContinuous
    from colosseum_amd.experiment.batched_loop import BatchedContinuousLoop

    cases = json.load(open(os.path.join(GOLDEN, "G10_mdploop_continuous.json")))
    for c in cases:
        m = make_model(c["mdp_cls"], **c["mdp_kwargs"])
        kw = dict(c["agent_kwargs"])
        seed = kw.pop("seed")
        # (i) plain run: actions equal the reference's as long as its MDPLoop kept training
        env = BatchedMDP([m], rng_mode=L.RNG_MT_COMPAT, with_dp=False)
        env.reset()
        ag = BatchedQLearningContinuous(env, [seed], **kw)
        ref_rows = c["rows"]
        acts = ag.run(c["T"], trace_actions=True)["actions"][:, 0]
        host = QLearningContinuous(mdp_specs=_SpecC(m), seed=seed, **kw)
        e = O.OracleEnv(m, rng_mode=0)
        ts, hacts = ts_.restart(e.reset()), []
        for t in range(c["T"]):
            a = int(host.select_action(ts, t))
            hacts.append(a)
            ty, o, r, _ = e.step(a)
            nts = ts_.transition(r, o)
            host.step_update(ts, a, nts, t)
            ts = nts
        np.testing.assert_array_equal(acts, np.array(hacts, np.int8))
        Q, N = ag.tables()
        np.testing.assert_array_equal(Q[0], host.Q)
        np.testing.assert_array_equal(N[0], host.N)
        ag.close()
        env.close()
        # (ii) the full loop with logging, twice in one batch
        env = BatchedMDP([m, m], rng_mode=L.RNG_MT_COMPAT)
        ag = BatchedQLearningContinuous(env, [seed, seed], **kw)
        rows = BatchedContinuousLoop(env, ag).run(T=c["T"], log_every=c["log_every"])
        for inst in rows:
            assert len(inst) == len(ref_rows)
            for got, ref in zip(inst, ref_rows):
                for k, v in ref.items():
                    assert float(got[k]) == pytest.approx(v, rel=2e-6, abs=2e-5), (c["mdp_kwargs"], k, got["steps"])
        ag.close()
        env.close()
        # (iii) the same run with Python in the loop (one call per step of the schedule): identical rows, value and type
        env = BatchedMDP([m, m], rng_mode=L.RNG_MT_COMPAT)
        ag = BatchedQLearningContinuous(env, [seed, seed], **kw)
        loop = BatchedContinuousLoop(env, ag)
        loop.native = False
        _assert_same_rows(rows, loop.run(T=c["T"], log_every=c["log_every"]))
        ag.close()
        env.close()


def test_log_every_one_reads_the_reward_sum_of_the_previous_step(need_gpu):
    """A row at EVERY step (log_every = 1): no step lies between two rows, and the logged `cumulative_reward` must still be
    the sum through step t-1 (agent_mdp_interaction.py:291 adds the reward after the row is written) -- in the one-call
    loop, in the Python-driven loop, and equal to the per-instance MDPLoop with the numpy agent."""
    from colosseum_amd.experiment import MDPLoop, make_mdp_spec
    from colosseum_amd.experiment.batched_loop import BatchedEpisodicLoop
    from colosseum_amd.mdp import gpu_mdp

    kw = dict(seed=3, size=5, p_rand=0.2)
    hp = dict(p=0.05, UCB_type="bernstein", c_1=0.9415278732894797, c_2=0.013873778519317169, min_at=0.07263563483119442)
    m = make_model("DeepSeaEpisodic", **kw)
    T = 60
    runs = []
    for native in (True, False):
        env = BatchedMDP([m, m], rng_mode=L.RNG_MT_COMPAT)
        ag = BatchedQLearningEpisodic(env, [3, 3], optimization_horizon=T, **hp)
        loop = BatchedEpisodicLoop(env, ag)
        loop.native = native
        runs.append(loop.run(T=T, log_every=1))
        ag.close()
        env.close()
    _assert_same_rows(*runs)
    mdp = gpu_mdp.DeepSeaEpisodic(**kw)
    agent = QLearningEpisodic(seed=3, mdp_specs=make_mdp_spec(mdp), optimization_horizon=T, **hp)
    host = MDPLoop(mdp, agent)
    host.run(T=T, log_every=1)
    assert len(host.logger.data) == len(runs[0][0]) == T - 1 + 1
    cums = [float(r["cumulative_reward"]) for r in runs[0][0]]
    assert len(set(cums)) > 3  # the sum moves from row to row
    for got, ref in zip(runs[0][0], host.logger.data):
        for k in ref:
            if k != "steps_per_second":
                assert float(got[k]) == pytest.approx(float(ref[k]), rel=1e-6, abs=1e-5), (k, got["steps"])
    mdp.close()
