"""BatchedEpisodicLoop: `MDPLoop.run` for B (environment, agent) pairs at once, agents on the device.

The reference runs one `MDPLoop` per OS process (colosseum/experiment/experiment_instances.py:160-166,178-223).  Here
the interaction (select_action -> step -> step_update, reset at episode ends) is the fused kernel of
colosseum_amd.agents; at every logging step the greedy policies of all agents are extracted and evaluated on the device
in one call, and the 18 indicators of every instance are then produced by the very indicator code of
`colosseum_amd.experiment.MDPLoop` (one light tracker object per instance), so rows are identical to a per-instance run:
same step at which the reward sum is read (the reference logs BEFORE adding the current step's reward), same
`_is_policy_optimal` freeze of training, same float32/float64 scalar arithmetic."""
from typing import Dict, List

import numpy as np

from ..agents import BatchedQLearningEpisodic
from .. import _lib as L
from ..batched import BatchedMDP
from .mdp_loop import InMemoryLogger, MDPLoop
from .vector_tracker import MP, ContinuousVectorTracker, EpisodicVectorTracker


class _InstanceView:
    """What MDPLoop's indicator code reads from `mdp`, for one instance of the batch (states are plain indices)."""

    def __init__(self, H, start_states, start_probs, v_opt0, v_worst0, v_rand0):
        self.H = H
        self.starting_nodes = [int(s) for s in start_states]
        self.node_to_index = {s: s for s in self.starting_nodes}
        self.last_starting_node = self.starting_nodes[0]
        self._probs = [float(p) for p in start_probs]
        self._v = dict(opt=v_opt0, worst=v_worst0, rand=v_rand0)
        self.parameters = {}

        def avg(v):
            acc = 0.0
            for sn, p in zip(self.starting_nodes, self._probs):
                acc += p * v[sn]
            return acc / H

        self.episodic_optimal_average_reward = avg(v_opt0)
        self.episodic_worst_average_reward = avg(v_worst0)
        self.episodic_random_average_reward = avg(v_rand0)

    @staticmethod
    def is_episodic():
        return True

    def get_minimal_regret_for_starting_node(self, n):
        return self._v["opt"][n] - self._v["worst"][n]


class _Tracker(MDPLoop):
    def __init__(self, view: _InstanceView, ssd: np.ndarray, n_check: int):
        self.logger = InMemoryLogger()
        self._mdp = view
        self._agent = None
        self._episodic = True
        self._n_steps_to_check_for_agent_optimality = n_check
        self._ssd = ssd
        self._eval = None
        self._max_time = np.inf

    def set_evaluation(self, v0: np.ndarray):
        """indicators.py:29-45 on the device-evaluated V[0]"""
        epi = sum(v0 * self._ssd)
        self._eval = (np.maximum(self._mdp._v["opt"] - v0, 0.0), epi)

    def _episodic_regrets_and_average_reward(self):
        return self._eval


class BatchedEpisodicLoop:
    """`vectorized=True` (default): indicators of all instances per numpy call (vector_tracker.py); False: one scalar
    tracker object per instance running MDPLoop's own indicator code (slow; kept as the cross-check)."""

    def __init__(self, env: BatchedMDP, agent: BatchedQLearningEpisodic,
                 n_log_intervals_to_check_for_agent_optimality: int = 10, vectorized: bool = True):
        assert env.H > 0 and env.models is not None
        self.env, self.agent = env, agent
        self.vectorized = vectorized
        H, A = env.H, env.A
        # baselines of every instance, batched: optimal values, worst policy values, uniform policy values
        Q, V = env.episodic_value_iteration()
        Qw, _ = env.episodic_value_iteration(R=[-m.reward_matrix() for m in env.models])
        pi_w = env.greedy_policy_episodic(Qw, q_layers=H + 1)
        _, Vw = env.episodic_policy_evaluation(pi_w)
        _, Vr = env.episodic_policy_evaluation([np.ones((H, m.n_states, A), np.float32) / A for m in env.models])
        self.trackers: List[_Tracker] = []
        if vectorized:
            flat0 = [np.concatenate([env.split_states(x, H + 1)[b][:m.n_states] for b, m in enumerate(env.models)])
                     for x in (V, Vw, Vr)]
            self.vt = EpisodicVectorTracker(H, env.state_off, *flat0, [(m.start_states, m.start_probs) for m in env.models],
                                            n_log_intervals_to_check_for_agent_optimality)
            return
        for b, m in enumerate(env.models):
            S = m.n_states
            v0 = [env.split_states(x, H + 1)[b][:S] for x in (V, Vw, Vr)]
            view = _InstanceView(H, m.start_states, m.start_probs, *v0)
            ssd = np.zeros(S)
            ssd[m.start_states] = m.start_probs
            self.trackers.append(_Tracker(view, ssd, n_log_intervals_to_check_for_agent_optimality))

    def _log(self, t: int, cum: np.ndarray, n_since: int, T: int, in_loop: bool):
        V0 = self.agent.evaluate()
        last_start = self.env.last_start()
        prev_start = self.env.previous_start
        hstep = self.env.state()[1]
        if self.vectorized:
            start = np.where((hstep == 0) & in_loop, prev_start, last_start)
            self.vt.update(t, T, V0, start, cum, n_since, in_loop)
            return
        for b, tr in enumerate(self.trackers):
            # the reference logs step t before the reset that follows a termination: if step t ended an episode
            # (in-episode time back at 0), its `last_starting_node` is still the start of the episode that ended
            ended = in_loop and hstep[b] == 0
            tr._mdp.last_starting_node = int(prev_start[b] if ended else last_start[b])
            tr.set_evaluation(self.env.split_states(V0)[b])
            tr._cumulative_reward = float(cum[b])
            tr._n_steps_since_last_log = n_since
            tr._update_performance_logs(t)
            if in_loop:  # agent_mdp_interaction.py:265-288
                tr._latest_expected_regrets.append(tr._normalized_regret)
                if len(tr._latest_expected_regrets) > tr._n_steps_to_check_for_agent_optimality:
                    tr._latest_expected_regrets.pop(0)
                if tr._is_training and t > 0.2 * T and tr._is_policy_optimal():
                    tr._is_training = False

    def run(self, T: int, log_every: int = -1) -> List[List[Dict[str, float]]]:
        env, agent = self.env, self.agent
        if self.vectorized:
            self.vt.reset()
        for tr in self.trackers:
            tr._reset_run_variables()
        env.reset_visits()
        env.reset()
        done, n_since = 0, 0
        mask = np.ones(env.B, bool)
        cum = np.zeros(env.B)
        log_ts = [t for t in range(log_every, T, log_every)] if log_every and log_every > 0 else []
        for tl in log_ts:
            # the reference reads `_cumulative_reward` at step tl BEFORE adding that step's reward: stop after step
            # tl-1 to read the sum, then execute step tl (whose update the logged policy already contains)
            if tl - done > 0:
                cum = agent.run(tl - done, train=mask)["cumulative_reward"]
                n_since += tl - done
            agent.run(1, train=mask)
            done = tl + 1
            self._log(tl, cum, n_since, T, in_loop=True)
            mask = self.vt.is_training.copy() if self.vectorized else np.array([tr._is_training for tr in self.trackers])
            n_since = 1
        if T - done > 0:
            n_since += T - done
        cum = agent.run(T - done, train=mask)["cumulative_reward"]
        self._log(T - 1, cum, n_since, T, in_loop=False)
        return self.vt.tables() if self.vectorized else [tr.logger.data for tr in self.trackers]


class _ContinuousView:
    def __init__(self, optimal, worst, random, parameters):
        self.optimal_average_reward, self.worst_average_reward, self.random_average_reward = optimal, worst, random
        self.parameters = parameters

    @staticmethod
    def is_episodic():
        return False


class _ContinuousTracker(MDPLoop):
    def __init__(self, view, n_check):
        self.logger = InMemoryLogger()
        self._mdp = view
        self._agent = None
        self._episodic = False
        self._n_steps_to_check_for_agent_optimality = n_check
        self._avg = None
        self._max_time = np.inf

    def _average_reward_of_agent_policy(self):
        return self._avg


class BatchedContinuousLoop:
    """`MDPLoop.run` for a batch of continuous (environment, QLearningContinuous) pairs: interaction on the device,
    regrets from the stationary distributions of the agents' greedy policies (host class bookkeeping, one batched GTH
    call per logging step)."""

    def __init__(self, env: BatchedMDP, agent, n_log_intervals_to_check_for_agent_optimality: int = 10,
                 vectorized: bool = True):
        from ..dynamic_programming import get_policy_from_q_values
        from ..markov_chain import AverageRewardCache, get_average_reward_batch

        assert env.H == 0 and env.models is not None
        self.env, self.agent = env, agent
        self.vectorized = vectorized
        self._device_chain = True
        self._batch = get_average_reward_batch
        A = env.A
        self._TR = [m.dense() for m in env.models]
        # baselines: optimal / worst (greedy w.r.t. VI on R / -R, gamma .99, eps 1e-3) and uniform policies
        Q, _, _ = env.value_iteration()
        Qw, _, _ = env.value_iteration(R=[-m.reward_matrix() for m in env.models])
        probs = []
        for b, m in enumerate(env.models):
            T, R = self._TR[b]
            starts = list(zip(m.start_states.tolist(), m.start_probs.tolist()))
            S = m.n_states
            pi_o = get_policy_from_q_values(env.split_rows(Q)[b].reshape(S, A), True)
            pi_w = get_policy_from_q_values(env.split_rows(Qw)[b].reshape(S, A), True)
            pi_r = np.ones((S, A), np.float32) / A
            probs += [(T, R, pi_o, starts), (T, R, pi_w, starts), (T, R, pi_r, None)]
        vals = self._batch(probs)
        self.trackers = []
        if vectorized:
            self.cache = AverageRewardCache(self._TR)
            self.vt = ContinuousVectorTracker(*(MP.from_scalars(vals[j::3]) for j in range(3)),
                                              n_log_intervals_to_check_for_agent_optimality)
            return
        for b, m in enumerate(env.models):
            # `sum(sd * ars)` of the reference is a left-to-right Python sum; the batched helper uses ndarray.sum --
            # the difference is below 1e-15 relative and far below the 5-decimal rounding of the logger
            view = _ContinuousView(vals[3 * b], vals[3 * b + 1], vals[3 * b + 2], m.extra.get("kwargs", {}))
            self.trackers.append(_ContinuousTracker(view, n_log_intervals_to_check_for_agent_optimality))

    def _log(self, t, cum, n_since, T, in_loop):
        if self.vectorized:
            def averages(need):
                if self._device_chain:
                    try:
                        return self.agent.average_reward(need)
                    except L.CmdpError as e:  # instance too large for the kernel's LDS budget: host bookkeeping + GTH kernel
                        if e.code != L.ERR_UNSUPPORTED:
                            raise
                        self._device_chain = False
                policies = self.agent.policy()
                cur, _, _ = self.env.state()
                return self.cache(need, policies, cur)

            self.vt.update(t, T, averages, cum, n_since, in_loop)
            return
        policies = self.agent.policy()
        cur, _, _ = self.env.state()
        avgs = self._batch([(self._TR[b][0], self._TR[b][1], policies[b], [(int(cur[b]), 1.0)]) for b in range(self.env.B)])
        for b, tr in enumerate(self.trackers):
            tr._avg = avgs[b]
            tr._cumulative_reward = float(cum[b])
            tr._n_steps_since_last_log = n_since
            tr._update_performance_logs(t)
            if in_loop:
                tr._latest_expected_regrets.append(tr._normalized_regret)
                if len(tr._latest_expected_regrets) > tr._n_steps_to_check_for_agent_optimality:
                    tr._latest_expected_regrets.pop(0)
                if tr._is_training and t > 0.2 * T and tr._is_policy_optimal():
                    tr._is_training = False

    run = BatchedEpisodicLoop.run
