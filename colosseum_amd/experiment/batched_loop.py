"""BatchedEpisodicLoop / BatchedContinuousLoop: `MDPLoop.run` for B (environment, agent) pairs at once, agents on the
device.

The reference runs one `MDPLoop` per OS process (colosseum/experiment/experiment_instances.py:160-166,178-223).  Here
the interaction (select_action -> step -> step_update, reset at episode ends) is the fused kernel of
colosseum_amd.agents; at every logging step the greedy policies of all agents are extracted and evaluated on the device
in one call, and the 18 indicators of every instance come from the vector tracker (vector_tracker.py, the same code
`MDPLoop` runs with B = 1), so rows are identical to a per-instance run: same step at which the reward sum is read (the
reference logs BEFORE adding the current step's reward), same `_is_policy_optimal` freeze of training, same
float32/float64 scalar arithmetic.

`run(T, log_every, max_time)`: the reference gives every instance `max_time` seconds of its own process
(agent_mdp_interaction.py:160-177,238-241: training is frozen once fewer than 0.5 s remain).  The instances of a batch
share one clock -- the batch's wall time -- so the limit freezes every instance still training at the first logging
step after `max_time - 0.5` seconds; `last_training_step[b]` records where (the reference's return value, -1 when the
limit was not hit)."""
from typing import Dict, List

import numpy as np

from ..agents import BatchedQLearningEpisodic
from .. import _lib as L
from ..batched import BatchedMDP
from .vector_tracker import MP, ContinuousVectorTracker, EpisodicVectorTracker


class _BatchedLoop:
    """What both settings share: the stepping schedule around the logging steps and the wall-clock limit."""

    env: BatchedMDP
    vt = None  # the vector tracker, built by the subclass

    def _log(self, t: int, cum: np.ndarray, n_since: int, T: int, in_loop: bool):
        raise NotImplementedError

    native = True  # the whole run in one library call (cmdp_qlearning_run_logged); False: the Python-driven loop below

    def _desc(self, T: int, log_every: int, max_time: float):
        raise NotImplementedError

    def run(self, T: int, log_every: int = -1, max_time: float = np.inf) -> List[List[Dict[str, float]]]:
        if self.native:
            try:
                return self._run_native(T, log_every, max_time)
            except L.CmdpError as e:  # e.g. a continuous instance beyond the LDS budget of the chain kernel K9
                # cmdp_qlearning_run_logged refuses such a batch BEFORE it resets or steps anything, so the agent the
                # Python-driven loop starts from is the untouched one
                if e.code != L.ERR_UNSUPPORTED:
                    raise
        return self._run_python(T, log_every, max_time)

    def _run_native(self, T: int, log_every: int, max_time: float):
        from .vector_tracker import n_log_rows, native_log

        self.vt.reset()
        desc, keep = self._desc(T, log_every, max_time)
        steps, values, kinds, last, training = self.agent.run_logged(desc, n_log_rows(T, log_every))
        self.vt.log = native_log(self.env.B, steps, values, kinds)
        self.vt.is_training = training
        self.last_training_step = last
        return self.vt.tables()

    def _run_python(self, T: int, log_every: int = -1, max_time: float = np.inf) -> List[List[Dict[str, float]]]:
        from time import time

        env, agent = self.env, self.agent
        self.vt.reset()
        self.last_training_step = np.full(env.B, -1, np.int64)
        timer = time()
        env.reset_visits()
        env.reset()
        done, n_since = 0, 0
        mask = np.ones(env.B, bool)
        cum = np.zeros(env.B)
        log_ts = [t for t in range(log_every, T, log_every)] if log_every and log_every > 0 else []
        for tl in log_ts:
            # the reference reads `_cumulative_reward` at step tl BEFORE adding that step's reward: stop after step
            # tl-1 to read the sum, then execute step tl (whose update the logged policy already contains); when no step
            # lies between two rows (log_every == 1) that sum is what the previous row's single step returned
            if tl - done > 0:
                cum = agent.run(tl - done, train=mask)["cumulative_reward"]
                n_since += tl - done
            elif done > 0:
                cum = cum_after
            cum_after = agent.run(1, train=mask)["cumulative_reward"]
            done = tl + 1
            self._log(tl, cum, n_since, T, in_loop=True)
            if max_time - (time() - timer) < 0.5:  # `_limit_exceeded` (agent_mdp_interaction.py:172-177) for the batch
                hit = self.vt.is_training.copy()
                self.last_training_step[hit] = tl
                self.vt.is_training[:] = False
            mask = self.vt.is_training.copy()
            n_since = 1
        if T - done > 0:
            n_since += T - done
        cum = agent.run(T - done, train=mask)["cumulative_reward"]
        self._log(T - 1, cum, n_since, T, in_loop=False)
        return self.vt.tables()


class BatchedEpisodicLoop(_BatchedLoop):
    def __init__(self, env: BatchedMDP, agent: BatchedQLearningEpisodic,
                 n_log_intervals_to_check_for_agent_optimality: int = 10):
        assert env.H > 0 and env.models is not None
        self.env, self.agent = env, agent
        H, A = env.H, env.A
        # baselines of every instance, batched: optimal values, worst policy values, uniform policy values
        Q, V = env.episodic_value_iteration()
        Qw, _ = env.episodic_value_iteration(R=[-m.reward_matrix() for m in env.models])
        pi_w = env.greedy_policy_episodic(Qw, q_layers=H + 1)
        _, Vw = env.episodic_policy_evaluation(pi_w)
        _, Vr = env.episodic_policy_evaluation([np.ones((H, m.n_states, A), np.float32) / A for m in env.models])
        flat0 = [np.concatenate([env.split_states(x, H + 1)[b][:m.n_states] for b, m in enumerate(env.models)])
                 for x in (V, Vw, Vr)]
        self.vt = EpisodicVectorTracker(H, env.state_off, *flat0, [(m.start_states, m.start_probs) for m in env.models],
                                        n_log_intervals_to_check_for_agent_optimality)

    def _desc(self, T, log_every, max_time):
        from .vector_tracker import loop_desc

        vt = self.vt
        return loop_desc(T, log_every, vt.n_check, (vt.opt, vt.worst, vt.rand), max_time, H=vt.H, opt0=vt.opt0, worst0=vt.worst0,
                         start_pos=vt._ss, start_prob=vt._sp)

    def _log(self, t: int, cum: np.ndarray, n_since: int, T: int, in_loop: bool):
        V0 = self.agent.evaluate()
        last_start = self.env.last_start()
        prev_start = self.env.previous_start
        hstep = self.env.state()[1]
        # the reference logs step t before the reset that follows a termination: if step t ended an episode (in-episode
        # time back at 0), its `last_starting_node` is still the start of the episode that ended
        start = np.where((hstep == 0) & in_loop, prev_start, last_start)
        self.vt.update(t, T, V0, start, cum, n_since, in_loop)


class BatchedContinuousLoop(_BatchedLoop):
    """`MDPLoop.run` for a batch of continuous (environment, QLearningContinuous) pairs: interaction on the device,
    regrets from the average rewards of the agents' greedy policies (kernel K9; host class bookkeeping + the GTH kernel
    above its LDS budget)."""

    def __init__(self, env: BatchedMDP, agent, n_log_intervals_to_check_for_agent_optimality: int = 10):
        from ..dynamic_programming import get_policy_from_q_values
        from ..markov_chain import AverageRewardCache, get_average_reward_batch

        assert env.H == 0 and env.models is not None
        self.env, self.agent = env, agent
        self._device_chain = True
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
        vals = get_average_reward_batch(probs, builtin_sum=True)  # the MDP's properties, not get_average_reward
        self.cache = AverageRewardCache(self._TR)
        self.vt = ContinuousVectorTracker(*(MP.from_scalars(vals[j::3]) for j in range(3)),
                                          n_log_intervals_to_check_for_agent_optimality)

    def _desc(self, T, log_every, max_time):
        from .vector_tracker import loop_desc

        vt = self.vt
        return loop_desc(T, log_every, vt.n_check, (vt.opt, vt.worst, vt.rand), max_time)

    def _log(self, t, cum, n_since, T, in_loop):
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
