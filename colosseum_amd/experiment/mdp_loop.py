"""MDPLoop: the agent/MDP interaction loop and its performance indicators, host side
(reference colosseum/experiment/agent_mdp_interaction.py:107-578, colosseum/experiment/indicators.py:29-45).

The loop body is the reference's; `mdp.step`/`mdp.reset` go through the C ABI and every policy evaluation behind
the regret indicators runs on the HIP dynamic-programming kernels.  Agents are host Python objects with the
reference's `BaseAgent` contract (`select_action`, `step_update`, `is_episode_end`, `episode_end_update`,
`before_start_interacting`, `agent_logs`, `current_optimal_stochastic_policy`, `is_episodic`).

Both settings are built.  Episodic regrets come from finite-horizon policy evaluation (config C1); continuous regrets
from the stationary distribution of the agent's current greedy policy (colosseum/mdp/utils/markov_chain.py:12-31),
computed by the GTH kernel through colosseum_amd.markov_chain."""
from time import time
from typing import Any, NamedTuple, Tuple

import numpy as np

from ..dynamic_programming import episodic_policy_evaluation, episodic_value_iteration
from ..markov_chain import get_average_reward


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


def get_episodic_regrets_and_average_reward_at_time_zero(H, T, R, policy, starting_state_distribution, optimal_value=None):
    """colosseum/experiment/indicators.py:29-45"""
    _, V = episodic_policy_evaluation(H, T, R, policy)
    episodic_agent_average_reward = sum(V[0] * starting_state_distribution)
    if optimal_value is None:
        _, optimal_value = episodic_value_iteration(H, T, R)
    regret_at_time_zero = np.maximum(optimal_value[0] - V[0], 0.0)
    return regret_at_time_zero, episodic_agent_average_reward


class MDPLoop:
    def __init__(self, mdp, agent, logger=None, n_log_intervals_to_check_for_agent_optimality: int = 10):
        self.logger = InMemoryLogger() if logger is None else logger
        self._mdp = mdp
        self._agent = agent
        self._episodic = mdp.is_episodic()
        self._n_steps_to_check_for_agent_optimality = n_log_intervals_to_check_for_agent_optimality
        assert self._episodic == agent.is_episodic()
        self.actions_sequence = []

    @property
    def remaining_time(self) -> float:
        return self._max_time - (time() - self._mdp_loop_timer)

    def _limit_exceeded(self, t):
        self._is_training = False
        self._last_training_step = t

    def run(self, T: int, log_every: int = -1, max_time: float = np.inf):
        """agent_mdp_interaction.py:179-302.  The wall-clock limit freezes training once fewer than 0.5 s remain
        (checked at every step, as in the reference); per-call pre-emption of a running agent update
        (wrapt_timeout_decorator) is host orchestration and not reproduced."""
        assert type(log_every) == int, f"The log_every variable should be an integer, received value: {log_every}."
        log_every = -1 if log_every == 0 else log_every
        mdp, agent = self._mdp, self._agent
        mdp.reset_visitation_counts()
        self._reset_run_variables()
        self._max_time = max_time
        ts = mdp.reset()
        agent.before_start_interacting()
        t = -1
        for t in range(T):
            if self._is_training and self.remaining_time < 0.5:
                self._limit_exceeded(t)
            h = mdp.h
            action = agent.select_action(ts, h)
            new_ts = mdp.step(action)
            self.actions_sequence.append(new_ts.reward)
            if self._is_training:
                agent.step_update(ts, action, new_ts, h)
            if self._is_training and agent.is_episode_end(ts, action, new_ts, h):
                agent.episode_end_update()
            if t > 0 and log_every > 0 and t % log_every == 0:
                self._update_performance_logs(t)
                self._n_steps_since_last_log = 0
                if hasattr(agent, "agent_logs"):
                    agent.agent_logs()
                self._latest_expected_regrets.append(self._normalized_regret)
                if len(self._latest_expected_regrets) > self._n_steps_to_check_for_agent_optimality:
                    self._latest_expected_regrets.pop(0)
                if self._is_training and t > 0.2 * T and self._is_policy_optimal():
                    self._is_training = False
            self._n_steps_since_last_log += 1
            self._cumulative_reward += new_ts.reward
            ts = new_ts
            if mdp.is_episodic() and new_ts.last():
                assert mdp.necessary_reset or t == T - 2
                ts = mdp.reset()
                self._n_episodes += 1
        self._update_performance_logs(t)
        self.logger.close()
        return self._last_training_step, self._last_logs

    # -- agent_mdp_interaction.py:304-390 -----------------------------------------------------------------------
    def _reset_run_variables(self):
        self._cumulative_reward = 0.0
        self._cumulative_regret = 0.0
        self._normalized_cumulative_regret = 0.0
        self._cumulative_expected_reward_agent = 0.0
        self._is_training = True
        self._n_steps_since_last_log = 0
        self._last_training_step = -1
        self._n_episodes = 0
        self._last_logs = None
        self._cached_episodic_regrets = None
        self._cached_continuous_regrets = None
        self._latest_expected_regrets = []
        m = self._mdp
        if self._episodic:
            span = m.episodic_optimal_average_reward - m.episodic_worst_average_reward
            self._episodic_regret_random_agent = m.episodic_optimal_average_reward - m.episodic_random_average_reward
            self._episodic_normalized_regret_random_agent = self._episodic_regret_random_agent / span
            self._episodic_regret_worst_agent = span
            self._episodic_normalized_regret_worst_agent = self._episodic_regret_worst_agent / span
            self._cumulative_reward_normalizer = lambda t, cr: (cr - t * m.episodic_worst_average_reward) / span
        else:
            span = m.optimal_average_reward - m.worst_average_reward
            self._regret_random_agent = m.optimal_average_reward - m.random_average_reward
            self._normalized_regret_random_agent = self._regret_random_agent / span
            self._regret_worst_agent = span
            self._normalized_regret_worst_agent = self._regret_worst_agent / span
            assert span > 0.0002, type(m).__name__ + str(m.parameters)  # agent_mdp_interaction.py:379-382
            self._cumulative_reward_normalizer = lambda t, cr: (cr - t * m.worst_average_reward) / span
        self.logger.reset()
        self._mdp_loop_timer = time()

    # -- agent_mdp_interaction.py:392-428 ----------------------------------------------------------------------------
    def _update_performance_logs(self, t: int):
        self._compute_performance_indicators(t + 1)
        n = self._cumulative_reward_normalizer
        self._last_logs = dict(
            steps=t,
            cumulative_regret=self._cumulative_regret,
            cumulative_reward=self._cumulative_reward,
            cumulative_expected_reward=self._cumulative_expected_reward_agent,
            normalized_cumulative_regret=self._normalized_cumulative_regret,
            normalized_cumulative_reward=n(t, self._cumulative_reward),
            normalized_cumulative_expected_reward=n(t, self._cumulative_expected_reward_agent),
            random_cumulative_regret=self._cumulative_regret_random_agent,
            random_cumulative_expected_reward=self._cumulative_reward_random_agent,
            random_normalized_cumulative_regret=self._normalized_cumulative_regret_random_agent,
            random_normalized_cumulative_expected_reward=n(t, self._cumulative_reward_random_agent),
            worst_cumulative_regret=self._cumulative_regret_worst_agent,
            worst_cumulative_expected_reward=self._cumulative_reward_worst_agent,
            worst_normalized_cumulative_regret=self._normalized_cumulative_regret_worst_agent,
            worst_normalized_cumulative_expected_reward=n(t, self._cumulative_reward_worst_agent),
            optimal_cumulative_expected_reward=self._cumulative_reward_optimal_agent,
            optimal_normalized_cumulative_expected_reward=n(t, self._cumulative_reward_optimal_agent),
            steps_per_second=t / (time() - self._mdp_loop_timer),
        )
        # "Communicate the indicators to the logger with a maximum of five digits" (:426-428)
        self.logger.write({k: np.round(v, 5) for k, v in self._last_logs.items()})

    # -- agent_mdp_interaction.py:435-502 -------------------------------------------------------------------------------
    def _compute_performance_indicators(self, t: int):
        m = self._mdp
        if self._episodic:
            self._compute_episodic_regret()
            self._cumulative_regret_random_agent = self._episodic_regret_random_agent * t
            self._normalized_cumulative_regret_random_agent = self._episodic_normalized_regret_random_agent * t
            self._cumulative_regret_worst_agent = self._episodic_regret_worst_agent * t
            self._normalized_cumulative_regret_worst_agent = self._episodic_normalized_regret_worst_agent * t
            self._cumulative_reward_random_agent = m.episodic_random_average_reward * t
            self._cumulative_reward_worst_agent = m.episodic_worst_average_reward * t
            self._cumulative_reward_optimal_agent = m.episodic_optimal_average_reward * t
            agent_average_reward = lambda: self._episodic_agent_average_reward / m.H  # noqa: E731
        else:
            self._compute_continuous_regret()
            self._cumulative_regret_random_agent = self._regret_random_agent * t
            self._normalized_cumulative_regret_random_agent = self._normalized_regret_random_agent * t
            self._cumulative_regret_worst_agent = self._regret_worst_agent * t
            self._normalized_cumulative_regret_worst_agent = self._normalized_regret_worst_agent * t
            self._cumulative_reward_random_agent = m.random_average_reward * t
            self._cumulative_reward_worst_agent = m.worst_average_reward * t
            self._cumulative_reward_optimal_agent = m.optimal_average_reward * t
            agent_average_reward = lambda: self._agent_continuous_average_reward  # noqa: E731
        assert self._regret >= 0.0, self._regret
        assert self._normalized_regret >= 0.0, self._normalized_regret
        self._cumulative_regret += self._regret * self._n_steps_since_last_log
        self._normalized_cumulative_regret += self._normalized_regret * self._n_steps_since_last_log
        self._cumulative_expected_reward_agent += agent_average_reward() * self._n_steps_since_last_log

    # -- agent_mdp_interaction.py:510-532 ---------------------------------------------------------------------------------
    def _compute_continuous_regret(self):
        if not self._is_training:
            if self._cached_continuous_regrets is None:
                self._cached_continuous_regrets = self._get_continuous_regrets()
            self._regret, self._normalized_regret = self._cached_continuous_regrets
        else:
            self._regret, self._normalized_regret = self._get_continuous_regrets()

    def _average_reward_of_agent_policy(self):
        """Average reward of the agent's current greedy policy from the current state's recurrent class; overridden by
        the batched loop (all instances' GTH eliminations in one device call)."""
        m = self._mdp
        return get_average_reward(m.T, m.R, self._agent.current_optimal_stochastic_policy,
                                  [(m.node_to_index[m.cur_node], 1.0)])

    def _get_continuous_regrets(self):
        m = self._mdp
        self._agent_continuous_average_reward = self._average_reward_of_agent_policy()
        r = m.optimal_average_reward - self._agent_continuous_average_reward
        if np.isclose(r, 0.0, atol=1e-3):
            r = 0.0
        if r < 0:
            r = 0
        nr = r / (m.optimal_average_reward - m.worst_average_reward)
        return r, nr

    # -- agent_mdp_interaction.py:534-578 ---------------------------------------------------------------------------------
    def _episodic_regrets_and_average_reward(self):
        """(regret at in-episode time zero for every state, average value at time zero) of the agent's current greedy
        policy; overridden by the batched loop, which evaluates all instances in one device call."""
        m = self._mdp
        return get_episodic_regrets_and_average_reward_at_time_zero(
            m.H, m.T, m.R, self._agent.current_optimal_stochastic_policy, m.starting_state_distribution,
            m.optimal_value_functions[1])

    def _compute_episodic_regret(self):
        m = self._mdp
        evaluate = self._episodic_regrets_and_average_reward

        if not self._is_training:
            if self._cached_episodic_regrets is None:
                Rs, epi = evaluate()
                self._episodic_agent_average_reward = epi
                self._cached_episodic_regrets = {
                    n: (Rs[m.node_to_index[n]] / m.H, Rs[m.node_to_index[n]] / m.get_minimal_regret_for_starting_node(n))
                    for n in m.starting_nodes
                }
            self._regret, self._normalized_regret = self._cached_episodic_regrets[m.last_starting_node]
        else:
            Rs, epi = evaluate()
            self._episodic_agent_average_reward = epi
            self._regret = Rs[m.node_to_index[m.last_starting_node]] / m.H
            self._normalized_regret = self._regret / m.get_minimal_regret_for_starting_node(m.last_starting_node) * m.H

    def _is_policy_optimal(self) -> bool:
        if (len(self._latest_expected_regrets) == self._n_steps_to_check_for_agent_optimality
                and np.isclose(0, self._latest_expected_regrets, atol=1e-4 if self._episodic else 1e-5).all()):
            if self._episodic:
                self._compute_episodic_regret()
            else:
                self._compute_continuous_regret()
            return bool(np.isclose(self._normalized_regret, 0).all())
        return False
