"""Test helper: a restatement of the reference's tabular episodic Q-learning agent with UCB exploration
(colosseum/agent/agents/episodic/q_learning.py:19-104, actor colosseum/agent/actors/Q_values_actor.py:67-88).
Agents are host Python and out of the product's scope; this copy of the *contract*
exists so that the GPU box -- where the reference is absent -- can replay the golden MDPLoop run (G7)."""
import numpy as np

from colosseum_amd.dynamic_programming import get_policy_from_q_values


class QLearningEpisodic:
    def __init__(self, seed, mdp_specs, optimization_horizon, p, c_1, c_2=None, min_at=0.0, UCB_type="hoeffding"):
        self._H = int(mdp_specs.time_horizon)
        self._n_states = mdp_specs.observations.num_values
        self._n_actions = mdp_specs.actions.num_values
        self._UCB_type, self._min_at, self._c_1, self._c_2, self._p = UCB_type.lower(), min_at, c_1, c_2, p
        H, S, A = self._H, self._n_states, self._n_actions
        self.i = np.log(S * A * optimization_horizon / p)
        self.N = np.ones((H, S, A), np.int32)
        self.Q = np.zeros((H, S, A), np.float32) + H
        self.V = np.zeros((H + 1, S), np.float32)
        self.mu = np.zeros((H, S, A), np.float32)
        self.sigma = np.zeros((H, S, A), np.float32)
        self.beta = np.zeros((H, S, A), np.float32)
        self._rng = np.random.RandomState(seed)  # the actor's stream (agent/actors/base.py:33)

    @staticmethod
    def is_episodic():
        return True

    @property
    def current_optimal_stochastic_policy(self):
        return get_policy_from_q_values(self.Q, True)

    def before_start_interacting(self):
        pass

    def episode_end_update(self):
        pass

    def agent_logs(self):
        pass

    def is_episode_end(self, ts_t, a_t, ts_tp1, time):
        return ts_tp1.last()

    def select_action(self, ts, time):
        q = self.Q[time, ts.observation]
        return self._rng.choice(np.where(q == q.max())[0])

    def step_update(self, ts_t, a_t, ts_tp1, time):
        H = self._H
        s_t, s_tp1 = ts_t.observation, ts_tp1.observation
        self.N[time, s_t, a_t] += 1
        t = self.N[time, s_t, a_t]
        alpha = max(self._min_at, (H + 1) / (H + t))
        if self._UCB_type == "hoeffding":
            b_t = self._c_1 * np.sqrt(H ** 3 * self.i / t)
        else:
            self.mu[time, s_t, a_t] += self.V[time + 1, s_tp1]
            self.sigma[time, s_t, a_t] += self.V[time + 1, s_tp1] ** 2
            old_beta = self.beta[time, s_t, a_t]
            self.beta[time, s_t, a_t] = min(
                self._c_1 * (np.sqrt((H * ((self.sigma[time, s_t, a_t] - self.mu[time, s_t, a_t]) ** 2) / t ** 2 + H) * self.i)
                             + np.sqrt(H ** 7 * self._n_states * self._n_actions) * self.i / t),
                self._c_2 * np.sqrt(H ** 3 * self.i / t),
            )
            b_t = (self.beta[time, s_t, a_t] - (1 - alpha) * old_beta) / 2 / alpha
        self.Q[time, s_t, a_t] = alpha * self.Q[time, s_t, a_t] + (1 - alpha) * (ts_tp1.reward + self.V[time + 1, s_tp1] + b_t)
        self.V[time, s_t] = min(H, self.Q[time, s_t].max())


class QLearningContinuous:
    """colosseum/agent/agents/infinite_horizon/q_learning.py:19-112 (optimistic Q-learning for the average-reward
    setting, Wei et al. 2020) with the greedy QValuesActor."""

    def __init__(self, seed, mdp_specs, optimization_horizon, min_at=0.0, confidence=0.95, span_approx_weight=1.0,
                 h_weight=1.0):
        S, A = mdp_specs.observations.num_values, mdp_specs.actions.num_values
        self.min_at = min_at if min_at > 0.009 else 0
        self.span_approx = span_approx_weight
        self.confidence = confidence
        self.optimization_horizon = optimization_horizon
        T = optimization_horizon
        self.H = h_weight * min(np.sqrt(self.span_approx * T / S / A), (T / S / A / np.log(4 * T / confidence)) ** 0.333)
        self.gamma = 1 - 1 / self.H
        self.N = np.zeros((S, A), np.int32)
        self.Q = np.zeros((S, A), np.float32) + self.H
        self.Q_main = np.zeros((S, A), np.float32) + self.H
        self.V = np.zeros((S,), np.float32) + self.H
        self._rng = np.random.RandomState(seed)

    @staticmethod
    def is_episodic():
        return False

    @property
    def current_optimal_stochastic_policy(self):
        return get_policy_from_q_values(self.Q, True)

    def before_start_interacting(self):
        pass

    def episode_end_update(self):
        pass

    def agent_logs(self):
        pass

    def is_episode_end(self, ts_t, a_t, ts_tp1, time):
        return ts_tp1.last()

    def select_action(self, ts, time):
        q = self.Q[ts.observation]
        return self._rng.choice(np.where(q == q.max())[0])

    def step_update(self, ts_t, a_t, ts_tp1, time):
        s_t, s_tp1 = ts_t.observation, ts_tp1.observation
        self.N[s_t, a_t] += 1
        alpha_t = max(self.min_at, (self.H + 1) / (self.H + self.N[s_t, a_t]))
        b_t = 4 * self.span_approx * np.sqrt(self.H / self.N[s_t, a_t] * np.log(2 * self.optimization_horizon / self.confidence))
        self.Q_main[s_t, a_t] = (1 - alpha_t) * self.Q[s_t, a_t] + alpha_t * (ts_tp1.reward + self.gamma * self.V[s_tp1] + b_t)
        self.Q[s_t, a_t] = min(self.Q[s_t, a_t], self.Q_main[s_t, a_t])
        self.V[s_tp1] = self.Q[s_tp1].max()
