"""Config C1 (plumbing): a host agent with the reference's BaseAgent contract drives a GpuMDP through the MDPLoop;
the action stream and the 17 deterministic indicators of every logger row must equal the reference's own run
(golden G7: reference MDPLoop + reference QLearningEpisodic on the reference DeepSeaEpisodic)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_csv_matches
from colosseum_amd.experiment import MDPLoop, make_mdp_spec
from colosseum_amd.mdp import gpu_mdp
from helpers_agents import QLearningContinuous, QLearningEpisodic

pytestmark = pytest.mark.gpu


def test_gpu_mdp_basemdp_surface(need_gpu):
    mdp = gpu_mdp.DeepSeaEpisodic(seed=0, size=8)
    with pytest.raises(AttributeError):
        mdp.step(0)  # `necessary_reset` only exists after the first reset(), as in the reference
    ts = mdp.reset()
    assert ts.first() and ts.reward is None and ts.observation == mdp.node_to_index[mdp.cur_node]
    spec = make_mdp_spec(mdp)
    assert spec.observations.num_values == 36 and spec.actions.num_values == 2 and spec.time_horizon == 8
    for _ in range(8):
        ts = mdp.step(np.int64(1))
    assert ts.last() and ts.observation == -1 and mdp.necessary_reset and mdp.h == 8
    with pytest.raises(AssertionError):
        mdp.step(0)
    assert mdp.step(0, auto_reset=True).first()
    counts = mdp.get_visitation_counts()
    assert sum(counts.values()) == 8 + 2 and counts[mdp.starting_nodes[0]] >= 2
    mdp.reset_visitation_counts()
    assert sum(mdp.get_visitation_counts(False).values()) == 0
    Q, V = mdp.optimal_value_functions
    assert Q.shape == (9, 36, 2) and V[0, mdp.starting_states[0]] == pytest.approx(1.0)
    assert mdp.T.shape == (36, 2, 36) and mdp.R.dtype == np.float32
    mdp.close()


def test_mdploop_qlearning_matches_reference_logs(need_gpu, tmp_path):
    cases = json.load(open(os.path.join(GOLDEN, "G7_mdploop_qlearning.json")))
    for c in cases:
        mdp = getattr(gpu_mdp, c["mdp_cls"])(**c["mdp_kwargs"])
        agent = QLearningEpisodic(mdp_specs=make_mdp_spec(mdp), **c["agent_kwargs"])
        actions = []
        sel = agent.select_action
        agent.select_action = lambda ts, h, _s=sel: (actions.append(int(_s(ts, h))) or actions[-1])
        # rows into memory and, through the package's CSVLogger, into the wire format (the golden holds the file the
        # reference's own CSVLogger wrote for this run)
        from colosseum_amd.experiment import CSVLogger, InMemoryLogger

        class Tee:
            def __init__(self, *lg):
                self.lg = lg
                self.data = lg[0].data

            def write(self, d):
                [x.write(d) for x in self.lg]

            def reset(self):
                [x.reset() for x in self.lg]
                self.data = self.lg[0].data

            def close(self):
                [x.close() for x in self.lg]

        csvl = CSVLogger(str(tmp_path), label=f"case{len(actions)}-{c['mdp_cls']}", file_name=f"seed{c['mdp_kwargs']['seed']}_logs")
        loop = MDPLoop(mdp, agent, Tee(InMemoryLogger(), csvl))
        last, logs = loop.run(T=c["T"], log_every=c["log_every"])
        assert_csv_matches(open(csvl.file_path, newline="").read(), c["csv_text"])
        assert last == c["last_training_step"]
        assert actions == c["actions"]  # identical trajectories => identical agent decisions, step by step
        np.testing.assert_allclose(np.asarray(agent.Q, np.float64), np.asarray(c["Q_final"]), atol=1e-6)
        rows = loop.logger.data
        assert len(rows) == len(c["rows"])
        for got, ref in zip(rows, c["rows"]):
            assert set(ref) == set(got) - {"steps_per_second"}
            for k, v in ref.items():
                # values are rounded to 5 decimals by the loop; float32 DP => 1e-6 relative slack on top
                assert float(got[k]) == pytest.approx(v, rel=2e-6, abs=2e-5), (k, got["steps"])
        mdp.close()


def test_stationary_distributions_and_average_rewards(need_gpu):
    """GTH kernel + recurrent-class logic vs the reference (golden G9) and the oracle's sequential GTH."""
    import numpy as np

    from conftest import load_golden
    from colosseum_amd import markov_chain as mc
    from oracle import oracle as O

    z, cases = load_golden("G9_stationary")
    for i, c in enumerate(cases):
        mdp = getattr(gpu_mdp, c["cls"])(**c["kwargs"])
        k = f"c{i}_"
        for name in ("optimal", "worst", "random"):
            np.testing.assert_allclose(getattr(mdp, f"{name}_stationary_distribution"), z[k + f"sd_{name}"], atol=1e-12)
            assert getattr(mdp, f"{name}_average_reward") == pytest.approx(c[f"{name}_average_reward"], rel=1e-9, abs=1e-12)
        ar = mc.get_average_reward(mdp.T, mdp.R, z[k + "pi_rand"], [(mdp.starting_states[0], 1.0)])
        assert ar == pytest.approx(c["avg_reward_pi_rand"], rel=1e-9)
        tps = mc.get_transition_probabilities(mdp.T, z[k + "pi_rand"])
        cls = mc.recurrent_classes(tps)[0]
        sub = tps[np.ix_(cls, cls)]
        np.testing.assert_array_equal(mc.gth_batch([sub, sub])[1], O.gth(sub))  # device == oracle, bit for bit
        mdp.close()


def test_mdploop_continuous_matches_reference_logs(need_gpu):
    cases = json.load(open(os.path.join(GOLDEN, "G10_mdploop_continuous.json")))
    for c in cases:
        mdp = getattr(gpu_mdp, c["mdp_cls"])(**c["mdp_kwargs"])
        assert mdp.optimal_average_reward == pytest.approx(c["optimal_average_reward"], rel=1e-9)
        agent = QLearningContinuous(mdp_specs=make_mdp_spec(mdp), **c["agent_kwargs"])
        actions = []
        sel = agent.select_action
        agent.select_action = lambda ts, h, _s=sel: (actions.append(int(_s(ts, h))) or actions[-1])
        loop = MDPLoop(mdp, agent)
        last, logs = loop.run(T=c["T"], log_every=c["log_every"])
        assert last == c["last_training_step"]
        assert actions == c["actions"]
        rows = loop.logger.data
        assert len(rows) == len(c["rows"])
        for got, ref in zip(rows, c["rows"]):
            for k, v in ref.items():
                assert float(got[k]) == pytest.approx(v, rel=2e-6, abs=2e-5), (k, got["steps"])
        mdp.close()
