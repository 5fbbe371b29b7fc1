"""Agents that run on the device next to their environments.

`BatchedQLearningEpisodic` = one reference `QLearningEpisodic` agent (colosseum/agent/agents/episodic/q_learning.py)
per instance of a `BatchedMDP`, advanced by a kernel that fuses select_action -> step -> step_update; Q tables and action
streams are bit-equal to the reference agent driven by the reference's MDPLoop (golden G7)."""
import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib as L
from .batched import BatchedMDP


class BatchedQLearningEpisodic:
    def __init__(self, env: BatchedMDP, seeds: Sequence[int], optimization_horizon: int, p: float, c_1: float,
                 c_2: Optional[float] = None, min_at: float = 0.0, UCB_type: str = "hoeffding"):
        ucb = {"hoeffding": 0, "bernstein": 1}[UCB_type.lower()]
        self._lib = L.load()
        self.env = env
        seeds = np.ascontiguousarray(seeds, np.int32)
        assert len(seeds) == env.B
        self._h = C.c_void_p()
        L.check(self._lib.cmdp_qlearning_create(C.byref(self._h), env._h, L.ptr(seeds), int(optimization_horizon), float(p),
                                                float(c_1), float(c_2 or 0.0), float(min_at), ucb))
        env._register_agent(self)

    def run(self, n_steps: int, train=True, trace_actions: bool = False):
        """n_steps of select_action -> step -> step_update per instance.  `train`: bool or per-instance mask.
        Returns the running cumulative reward (since creation) and, optionally, the actions [n_steps, B]."""
        n_steps = int(n_steps)
        acts = np.zeros((n_steps, self.env.B), np.int8) if trace_actions else None
        rsum = np.zeros(self.env.B, np.float64)
        mask = None
        if train is not True:
            mask = np.ascontiguousarray(np.broadcast_to(np.asarray(train, bool), (self.env.B,)), np.uint8)
        L.check(self._lib.cmdp_qlearning_run(self._h, n_steps, L.ptr(mask), L.ptr(acts), L.ptr(rsum)))
        return dict(cumulative_reward=rsum, actions=acts)

    def run_logged(self, desc, n_logs: int):
        """`MDPLoop.run` for the whole batch in one library call (cmdp_qlearning_run_logged): interaction, policy
        evaluation at every logging step and the reference's indicators, all without returning to Python.  `desc` from
        `vector_tracker.loop_desc`.  Returns (steps [n_logs], values, kinds [n_logs, 17, B], last_training_step, is_training)."""
        B = self.env.B
        steps = np.zeros(n_logs, np.int64)
        values = np.zeros((n_logs, len(L.LOG_COLUMNS), B), np.float64)
        kinds = np.zeros((n_logs, len(L.LOG_COLUMNS), B), np.uint8)
        last = np.zeros(B, np.int64)
        training = np.zeros(B, np.uint8)
        L.check(self._lib.cmdp_qlearning_run_logged(self._h, C.byref(desc), int(n_logs), L.ptr(steps), L.ptr(values), L.ptr(kinds),
                                                    L.ptr(last), L.ptr(training)))
        return steps, values, kinds, last, training.astype(bool)

    def evaluate(self) -> np.ndarray:
        """V[0, :] (concatenated over instances) of the current greedy policies, policy and evaluation on device."""
        V0 = np.zeros(int(self.env.state_off[-1]), np.float32)
        L.check(self._lib.cmdp_qlearning_evaluate(self._h, L.ptr(V0)))
        return V0

    def tables(self):
        """(Q, N): per instance arrays of shape [H, S_b, A]."""
        env = self.env
        n = int(env.H * env.row_off[-1])
        Q = np.zeros(n, np.float32)
        N = np.zeros(n, np.int32)
        L.check(self._lib.cmdp_qlearning_tables(self._h, L.ptr(Q), L.ptr(N)))
        qs = [x.reshape(env.H, -1, env.A) for x in env.split_rows(Q, env.H)]
        ns = [x.reshape(env.H, -1, env.A) for x in env.split_rows(N, env.H)]
        return qs, ns

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.cmdp_qlearning_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchedQLearningContinuous(BatchedQLearningEpisodic):
    """One reference `QLearningContinuous` (colosseum/agent/agents/infinite_horizon/q_learning.py) per instance."""

    def __init__(self, env: BatchedMDP, seeds: Sequence[int], optimization_horizon: int, min_at: float = 0.0,
                 confidence: float = 0.95, span_approx_weight: float = 1.0, h_weight: float = 1.0):
        self._lib = L.load()
        self.env = env
        seeds = np.ascontiguousarray(seeds, np.int32)
        assert len(seeds) == env.B
        self._h = C.c_void_p()
        L.check(self._lib.cmdp_qlearning_continuous_create(C.byref(self._h), env._h, L.ptr(seeds), int(optimization_horizon),
                                                           float(min_at), float(confidence), float(span_approx_weight),
                                                           float(h_weight)))
        env._register_agent(self)

    def tables(self):
        env = self.env
        Q = np.zeros(int(env.row_off[-1]), np.float64)  # float64 tables (NEP 50: float32 zeros + numpy float64 H)
        N = np.zeros(int(env.row_off[-1]), np.int32)
        L.check(self._lib.cmdp_qlearning_tables(self._h, L.ptr(Q), L.ptr(N)))
        return ([x.reshape(-1, env.A) for x in env.split_rows(Q)], [x.reshape(-1, env.A) for x in env.split_rows(N)])

    def evaluate(self):
        raise NotImplementedError("use policy(): continuous regrets come from the stationary distribution")

    def average_reward(self, mask=None) -> list:
        """`get_average_reward` of the current greedy policies from the current states, on the device (kernel K9): a
        list of numpy scalars (np.float32 where the reference's value is one) for the instances selected by `mask`."""
        B = self.env.B
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        avg = np.zeros(B, np.float64)
        kind = np.zeros(B, np.int32)
        L.check(self._lib.cmdp_qlearning_average_reward(self._h, L.ptr(m), L.ptr(avg), L.ptr(kind)))
        sel = range(B) if m is None else np.flatnonzero(m)
        return [np.float32(avg[b]) if kind[b] else np.float64(avg[b]) for b in sel]

    def policy(self):
        """argmax_2d greedy policies (RandomState(42) tie-break), per instance [S, A]."""
        env = self.env
        pi = np.zeros(int(env.row_off[-1]), np.float32)
        L.check(self._lib.cmdp_qlearning_policy(self._h, L.ptr(pi)))
        return [x.reshape(-1, env.A) for x in env.split_rows(pi)]
