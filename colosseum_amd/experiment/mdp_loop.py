"""MDPLoop: one agent interacting with one MDP, the reference's `MDPLoop` call surface
(colosseum/experiment/agent_mdp_interaction.py:107-302: `MDPLoop(mdp, agent, logger).run(T, log_every, max_time)`
returning `(last_training_step, last_logs)`, one logger row of 18 indicators per logging step).

This is the B = 1 case of the batched machinery: the interaction goes through the C ABI (`mdp.step` / `mdp.reset`), and
ALL indicator bookkeeping -- cumulative regrets, the baselines' columns, the 5-decimal rounding with the reference's
numpy scalar types, the ring of recent regrets that freezes training once the policy is optimal -- is the vector tracker
of `vector_tracker.py` with one instance (pinned against the reference's own indicator code by golden G15).  What this
class adds is only what a single host agent needs: the step loop in the reference's call order, the evaluation of the
agent's current greedy policy (episodic: `V[0]` by policy evaluation on the HIP kernels through the MDP's persistent
device handle; continuous: average reward of the policy's chain from the current state), and the wall-clock limit."""
from time import time
from typing import Any, NamedTuple, Tuple

import numpy as np

from .vector_tracker import MP, ContinuousVectorTracker, EpisodicVectorTracker


class MDPSpec(NamedTuple):
    """colosseum/utils/acme/specs.py:16-26"""

    observations: Any
    actions: Any
    rewards: Any
    discounts: Any
    time_horizon: Any
    rewards_range: Tuple[float, float]
    emission_map: Any
    n_states: int


def make_mdp_spec(mdp) -> MDPSpec:
    """colosseum/utils/acme/specs.py:29-40"""
    return MDPSpec(
        observations=mdp.observation_spec(),
        actions=mdp.action_spec(),
        rewards=mdp.reward_spec(),
        discounts=mdp.discount_spec(),
        time_horizon=mdp.H if mdp.is_episodic() else np.inf,
        rewards_range=mdp.rewards_range,
        emission_map=mdp.emission_map,
        n_states=mdp.n_states,
    )


class InMemoryLogger:
    """colosseum/utils/acme/in_memory_logger.py"""

    def __init__(self):
        self.reset()

    def write(self, data):
        self._data.append(data)

    def close(self):
        pass

    def reset(self):
        self._data = []

    @property
    def data(self):
        return self._data


class MDPLoop:
    def __init__(self, mdp, agent, logger=None, n_log_intervals_to_check_for_agent_optimality: int = 10):
        self.logger = InMemoryLogger() if logger is None else logger
        self._mdp, self._agent = mdp, agent
        self._episodic = mdp.is_episodic()
        self._n_check = n_log_intervals_to_check_for_agent_optimality
        assert self._episodic == agent.is_episodic()
        self.actions_sequence = []
        self._tracker = None

    # -- the tracker of this (MDP, agent) pair ----------------------------------------------------------------------------
    def _new_tracker(self):
        m = self._mdp
        if self._episodic:
            # V[0] of the optimal / worst / uniform policies and the start distribution in the sampler's order
            # (`episodic_*_average_reward`, mdp/base_finite.py:339-375, sums over it in that order)
            idx = [m.node_to_index[n] for n in m.starting_nodes]
            probs = [float(m.starting_state_distribution[i]) for i in idx]
            v0 = [np.asarray(vf[1][0], np.float32) for vf in
                  (m.optimal_value_functions, m.worst_value_functions, m.random_value_functions)]
            return EpisodicVectorTracker(m.H, np.array([0, m.n_states]), *v0, [(idx, probs)], self._n_check)
        return ContinuousVectorTracker(*(MP.from_scalars([x]) for x in
                                         (m.optimal_average_reward, m.worst_average_reward, m.random_average_reward)),
                                       self._n_check)

    @property
    def _is_training(self) -> bool:
        return bool(self._tracker.is_training[0])

    @property
    def remaining_time(self) -> float:
        return self._max_time - (time() - self._timer)

    # -- evaluation of the agent's current greedy policy ------------------------------------------------------------------
    def _policy_value_at_time_zero(self) -> np.ndarray:
        """V[0] of `agent.current_optimal_stochastic_policy` (indicators.py:29-45 evaluates it with
        episodic_policy_evaluation); `mdp.get_value_functions` runs on the MDP's own device handle."""
        return np.asarray(self._mdp.get_value_functions(self._agent.current_optimal_stochastic_policy)[1][0], np.float32)

    def _policy_average_reward(self):
        """agent_mdp_interaction.py:518-524: average reward of the greedy policy's chain from the current state."""
        from ..markov_chain import get_average_reward

        m = self._mdp
        return get_average_reward(m.T, m.R, self._agent.current_optimal_stochastic_policy,
                                  [(m.node_to_index[m.cur_node], 1.0)])

    def _log(self, t: int, T: int, in_loop: bool):
        tr = self._tracker
        if self._episodic:
            # a frozen agent's policy no longer changes: the reference caches its regrets per start state (:536-556)
            if self._is_training or self._frozen_v0 is None:
                v0 = self._policy_value_at_time_zero()
                if not self._is_training:
                    self._frozen_v0 = v0
            else:
                v0 = self._frozen_v0
            start = self._mdp.node_to_index[self._mdp.last_starting_node]
            tr.update(t, T, v0, np.array([start]), np.array([self._cumulative_reward]), self._n_since_log, in_loop)
        else:
            tr.update(t, T, lambda need: [self._policy_average_reward()], np.array([self._cumulative_reward]),
                      self._n_since_log, in_loop)
        # the row from the columns just appended (what `tr.tables()[0][-1]` would materialise, without re-stacking every
        # earlier row of every column at every logging step)
        row = {"steps": t, **{k: v.round5().scalar(0) for k, v in tr.last_cols.items()}}
        self._last_logs = {"steps": t, **{k: v.scalar(0) for k, v in tr.last_cols.items()}}
        self.logger.write(row)

    def run(self, T: int, log_every: int = -1, max_time: float = np.inf):
        """agent_mdp_interaction.py:179-302.  The wall-clock limit freezes training once fewer than 0.5 s remain
        (checked at every step, as in the reference); per-call pre-emption of a running agent update
        (wrapt_timeout_decorator) is host orchestration and not reproduced."""
        assert type(log_every) == int, f"The log_every variable should be an integer, received value: {log_every}."
        log_every = -1 if log_every == 0 else log_every
        mdp, agent = self._mdp, self._agent
        mdp.reset_visitation_counts()
        self._tracker = self._new_tracker()
        self.logger.reset()
        self._cumulative_reward, self._n_since_log, self._n_episodes = 0.0, 0, 0
        self._last_training_step, self._last_logs, self._frozen_v0 = -1, None, None
        self._max_time, self._timer = max_time, time()
        ts = mdp.reset()
        agent.before_start_interacting()
        t = -1
        for t in range(T):
            if self._is_training and self.remaining_time < 0.5:
                self._tracker.is_training[0] = False
                self._last_training_step = t
            h = mdp.h
            action = agent.select_action(ts, h)
            new_ts = mdp.step(action)
            self.actions_sequence.append(new_ts.reward)
            if self._is_training:
                agent.step_update(ts, action, new_ts, h)
                if agent.is_episode_end(ts, action, new_ts, h):
                    agent.episode_end_update()
            if t > 0 and log_every > 0 and t % log_every == 0:
                self._log(t, T, in_loop=True)  # row, then the ring of recent regrets and the optimality freeze
                self._n_since_log = 0
                if hasattr(agent, "agent_logs"):
                    agent.agent_logs()
            self._n_since_log += 1
            self._cumulative_reward += new_ts.reward
            ts = new_ts
            if self._episodic and new_ts.last():
                assert mdp.necessary_reset or t == T - 2
                ts = mdp.reset()
                self._n_episodes += 1
        self._log(t, T, in_loop=False)
        self.logger.close()
        return self._last_training_step, self._last_logs
