"""Performance indicators of `MDPLoop` for a whole batch of instances at once (host side, numpy).

The reference computes its 18 indicators with Python/numpy *scalars* (agent_mdp_interaction.py:304-578): some are
`np.float32`, some `np.float64`, some plain Python floats, and under NEP 50 the type of every intermediate decides the
rounding of every operation (e.g. the cumulative regret of an episodic run is accumulated in float32, and in the
continuous setting the type even differs between instances, because the stationary distribution of a chain with one
recurrent class smaller than the state space is float32, markov_chain.py:98).  This module reproduces that for B
instances per numpy call (B = 1 is `MDPLoop`):

* `MP` ("mixed precision") is an array of B emulated numpy scalars, each carrying its kind (Python float = weak,
  float32, float64).  A binary operation promotes like NEP 50 and rounds to float32 where the scalar code would have
  produced a float32 (computing a +,-,*,/ of two float32 values in float64 and rounding once is exact).
* `EpisodicVectorTracker` / `ContinuousVectorTracker` are `_reset_run_variables`, `_compute_performance_indicators`,
  `_compute_*_regret`, `_is_policy_optimal` and `_update_performance_logs` written on `MP` values.
* `BatchLog` / `LogTable` keep the rows column-wise ([n_logs, B] per indicator); `LogTable` is the per-instance
  sequence-of-dicts view (`InMemoryLogger.data`), `BatchLog.csv_text` the CSVLogger text.

tests/test_vector_tracker.py drives these with the inputs of golden G15 and requires the rows the reference's own
indicator code produced from them: same values, same numpy types, same training freeze."""
from time import time
from typing import Dict, List, Sequence

import numpy as np

WEAK, F32, F64 = 0, 1, 2
_DT = {WEAK: np.float64, F32: np.float32, F64: np.float64}


def _r32(x):
    with np.errstate(all="ignore"):
        return x.astype(np.float32).astype(np.float64)


class MP:
    """B emulated numpy scalars.  `k` is an int when all share one kind (then `v` has that kind's native dtype and the
    arithmetic is numpy's own), otherwise an int8 array and `v` is float64 with float32 results rounded explicitly."""

    __slots__ = ("v", "k")

    def __init__(self, v, k):
        if isinstance(k, (int, np.integer)):
            self.k = int(k)
            self.v = np.asarray(v, _DT[self.k])
        else:
            self.k = np.asarray(k, np.int8)
            self.v = np.asarray(v, np.float64)

    # -- construction helpers ------------------------------------------------------------------------------------------
    @staticmethod
    def from_scalars(xs: Sequence) -> "MP":
        """From a list of Python floats / np.float32 / np.float64 scalars."""
        kinds = np.array([F32 if isinstance(x, np.float32) else F64 if isinstance(x, np.floating) else WEAK for x in xs],
                         np.int8)
        vals = np.array([float(x) for x in xs], np.float64)
        if (kinds == kinds[0]).all():
            return MP(vals.astype(_DT[int(kinds[0])]), int(kinds[0]))
        return MP(vals, kinds)

    def mixed(self) -> "MP":
        if isinstance(self.k, int):
            return MP(self.v.astype(np.float64), np.full(self.v.shape, self.k, np.int8))
        return self

    def kinds(self) -> np.ndarray:
        return np.full(self.v.shape, self.k, np.int8) if isinstance(self.k, int) else self.k

    def values64(self) -> np.ndarray:
        return self.v.astype(np.float64)

    def scalar(self, b: int):
        k = self.k if isinstance(self.k, int) else int(self.k[b])
        x = self.v[b]
        return float(x) if k == WEAK else np.float32(x) if k == F32 else np.float64(x)

    # -- arithmetic ----------------------------------------------------------------------------------------------------
    def _bin(self, o, fn, rev=False):
        if isinstance(o, (int, float)):
            if isinstance(self.k, int):
                c = _DT[self.k](o)
                with np.errstate(all="ignore"):
                    return MP(fn(c, self.v) if rev else fn(self.v, c), self.k)
            o = MP(np.full(self.v.shape, float(o)), np.zeros(self.v.shape, np.int8))
        if isinstance(self.k, int) and isinstance(o.k, int):
            k = max(self.k, o.k)
            dt = _DT[k]
            a, b = self.v.astype(dt, copy=False), o.v.astype(dt, copy=False)
            with np.errstate(all="ignore"):
                return MP(fn(b, a) if rev else fn(a, b), k)
        s, o = self.mixed(), o.mixed()
        k = np.maximum(s.k, o.k)
        f32 = k == F32
        a = np.where(f32 & (s.k == WEAK), _r32(s.v), s.v)
        b = np.where(f32 & (o.k == WEAK), _r32(o.v), o.v)
        with np.errstate(all="ignore"):
            r = fn(b, a) if rev else fn(a, b)
        return MP(np.where(f32, _r32(r), r), k)

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.divide)
    def __rtruediv__(self, o): return self._bin(o, np.divide, True)

    def round5(self) -> "MP":
        """np.round(x, 5): multiply by 1e5, rint, divide, each in the scalar's own type; a Python float comes back as
        np.float64."""
        if isinstance(self.k, int):
            return MP(np.round(self.v, 5), F64 if self.k == WEAK else self.k)
        f32 = self.k == F32
        with np.errstate(all="ignore"):
            m = self.v * 1e5
            m = np.rint(np.where(f32, _r32(m), m))
            r = m / 1e5
        return MP(np.where(f32, _r32(r), r), np.where(self.k == WEAK, F64, self.k).astype(np.int8))

    @staticmethod
    def where(cond: np.ndarray, a: "MP", b: "MP") -> "MP":
        if isinstance(a.k, int) and isinstance(b.k, int) and a.k == b.k:
            return MP(np.where(cond, a.v, b.v), a.k)
        a, b = a.mixed(), b.mixed()
        return MP(np.where(cond, a.v, b.v), np.where(cond, a.k, b.k))

    @staticmethod
    def weak(value: float, n: int) -> "MP":
        return MP(np.full(n, float(value)), WEAK)

    def isclose_to_zero(self, atol: float) -> np.ndarray:
        """np.isclose(x, 0.0, atol=atol) for a scalar x of each kind: |x| <= atol compared in x's own type."""
        if isinstance(self.k, int):
            return np.abs(self.v) <= _DT[self.k](atol)
        thr = np.where(self.k == F32, float(np.float32(atol)), atol)
        return np.abs(self.v) <= thr


# ----------------------------------------------------------------------------------------------------------------------
class BatchLog:
    """Column store of the logger rows of B instances: columns[name] = list over logs of (values, kinds)."""

    def __init__(self, B: int):
        self.B = B
        self.steps: List[int] = []
        self._cols: Dict[str, list] = {}
        self._final = None

    def append(self, t: int, cols: Dict[str, MP]):
        self.steps.append(int(t))
        for name, mp in cols.items():
            self._cols.setdefault(name, []).append((mp.v, mp.k))
        self._final = None

    def __len__(self):
        return len(self.steps)

    def finalize(self):
        if self._final is None:
            out = {}
            for name, rows in self._cols.items():
                vals = np.stack([np.asarray(v, np.float64) for v, _ in rows]) if rows else np.zeros((0, self.B))
                if all(isinstance(k, int) for _, k in rows) and len({k for _, k in rows}) <= 1:
                    kinds = rows[0][1] if rows else F64
                else:
                    kinds = np.stack([np.full(self.B, k, np.int8) if isinstance(k, int) else k for _, k in rows])
                out[name] = (vals, kinds)
            self._final = out
        return self._final

    def names(self) -> List[str]:
        return ["steps"] + list(self._cols)

    def value(self, name: str, i: int, b: int):
        if name == "steps":
            return self.steps[i]
        vals, kinds = self.finalize()[name]
        k = kinds if isinstance(kinds, int) else int(kinds[i, b])
        return np.float32(vals[i, b]) if k == F32 else np.float64(vals[i, b])

    def text_columns(self) -> Dict[str, np.ndarray]:
        """str() of every logged value, [n_logs, B] per column (what csv.DictWriter would print)."""
        out = {"steps": np.repeat(np.array([str(s) for s in self.steps], object)[:, None], self.B, 1)}
        for name, (vals, kinds) in self.finalize().items():
            t64 = vals.astype(str)
            if isinstance(kinds, int):
                out[name] = vals.astype(np.float32).astype(str) if kinds == F32 else t64
            else:
                out[name] = np.where(kinds == F32, vals.astype(np.float32).astype(str), t64)
        return out

    def csv_text(self, b: int, text_cols=None) -> str:
        cols = text_cols if text_cols is not None else self.text_columns()
        fields = sorted(cols)
        lines = [",".join(fields)]
        per = [cols[f][:, b] for f in fields]
        lines += [",".join(r) for r in zip(*per)]
        return "\r\n".join(lines) + "\r\n"


class LogTable:
    """`InMemoryLogger.data` of one instance of a `BatchLog` (a sequence of dicts, materialised on access)."""

    def __init__(self, log: BatchLog, b: int):
        self.log, self.b = log, b

    def __len__(self):
        return len(self.log)

    def _row(self, i: int):
        return {n: self.log.value(n, i, self.b) for n in self.log.names()}

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._row(j) for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self._row(i)

    def __iter__(self):
        return (self._row(i) for i in range(len(self)))


# ----------------------------------------------------------------------------------------------------------------------
class _VectorTracker:
    """State shared by both settings (agent_mdp_interaction.py:304-428); `optimal`, `worst`, `random` are the average
    rewards of the three baseline policies (MP, one entry per instance)."""

    episodic: bool

    def __init__(self, optimal: MP, worst: MP, random: MP, n_check: int):
        self.B = len(optimal.v)
        self.opt, self.worst, self.rand = optimal, worst, random
        self.n_check = n_check
        self.reset()

    def reset(self):
        B = self.B
        self.cum_regret = MP.weak(0.0, B)
        self.norm_cum_regret = MP.weak(0.0, B)
        self.cum_expected_reward = MP.weak(0.0, B)
        self.is_training = np.ones(B, bool)
        self.span = self.opt - self.worst
        self.regret_random = self.opt - self.rand
        self.norm_regret_random = self.regret_random / self.span
        self.regret_worst = self.span
        self.norm_regret_worst = self.regret_worst / self.span
        self._ring: List[MP] = []
        self.log = BatchLog(B)
        self.timer = time()

    def _normalizer(self, t: int, cr):
        return (cr - t * self.worst) / self.span

    def _accumulate_and_log(self, t: int, regret: MP, nregret: MP, agent_avg: MP, cum_reward: np.ndarray, n_since: int):
        t1 = t + 1
        self.cum_regret = self.cum_regret + regret * n_since
        self.norm_cum_regret = self.norm_cum_regret + nregret * n_since
        self.cum_expected_reward = self.cum_expected_reward + agent_avg * n_since
        cr = MP(np.asarray(cum_reward, np.float64), WEAK)
        rnd, wst, opt = self.rand * t1, self.worst * t1, self.opt * t1
        n = self._normalizer
        cols = dict(
            cumulative_regret=self.cum_regret,
            cumulative_reward=cr,
            cumulative_expected_reward=self.cum_expected_reward,
            normalized_cumulative_regret=self.norm_cum_regret,
            normalized_cumulative_reward=n(t, cr),
            normalized_cumulative_expected_reward=n(t, self.cum_expected_reward),
            random_cumulative_regret=self.regret_random * t1,
            random_cumulative_expected_reward=rnd,
            random_normalized_cumulative_regret=self.norm_regret_random * t1,
            random_normalized_cumulative_expected_reward=n(t, rnd),
            worst_cumulative_regret=self.regret_worst * t1,
            worst_cumulative_expected_reward=wst,
            worst_normalized_cumulative_regret=self.norm_regret_worst * t1,
            worst_normalized_cumulative_expected_reward=n(t, wst),
            optimal_cumulative_expected_reward=opt,
            optimal_normalized_cumulative_expected_reward=n(t, opt),
            steps_per_second=MP(np.full(self.B, t / max(time() - self.timer, 1e-9)), WEAK),
        )
        self.last_cols = cols  # un-rounded (MDPLoop.run returns them as `last_logs`)
        self.log.append(t, {k: v.round5() for k, v in cols.items()})

    def _after_log(self, t: int, T: int, nregret: MP, atol: float, recompute):
        """agent_mdp_interaction.py:265-288: ring of the latest normalised regrets and the optimality freeze."""
        self._ring.append(nregret)
        if len(self._ring) > self.n_check:
            self._ring.pop(0)
        if len(self._ring) == self.n_check and t > 0.2 * T and self.is_training.any():
            kinds = np.stack([r.kinds() for r in self._ring])
            vals = np.stack([r.values64() for r in self._ring])
            all32 = (kinds == F32).all(0)
            close = np.where(all32, np.isclose(0, vals.astype(np.float32), atol=atol).all(0),
                             np.isclose(0, vals, atol=atol).all(0))
            cand = close & self.is_training
            if cand.any():
                nr = recompute()
                k = nr.kinds()
                v = nr.values64()
                zero = np.where(k == F32, np.isclose(v.astype(np.float32), 0), np.isclose(v, 0))
                self.is_training &= ~(cand & zero)

    def tables(self) -> List[LogTable]:
        return [LogTable(self.log, b) for b in range(self.B)]


class EpisodicVectorTracker(_VectorTracker):
    """Episodic regrets (agent_mdp_interaction.py:534-578, indicators.py:29-45) for B instances.

    state_off [B+1]: offsets of the instances in the flat per-state arrays; opt0/worst0: V*[0], V_worst[0] (float32,
    flat); starts: per instance (start states, probabilities)."""

    episodic = True

    def __init__(self, H: int, state_off, opt0, worst0, rand0, starts, n_check: int = 10):
        self.H = H
        self.off = np.asarray(state_off, np.int64)[:-1]
        self.opt0 = np.asarray(opt0, np.float32)
        self.worst0 = np.asarray(worst0, np.float32)
        B = len(self.off)
        kmax = max(len(s) for s, _ in starts)
        self._ss = np.zeros((B, kmax), np.int64)
        self._sp = np.zeros((B, kmax), np.float64)
        avgs = []
        for b, (ss, pp) in enumerate(starts):
            ss, pp = [int(s) for s in ss], [float(p) for p in pp]
            order = np.argsort(ss, kind="stable")
            for j, o in enumerate(order):  # `sum(V[0] * ssd)` runs over the states in index order
                self._ss[b, j] = self.off[b] + ss[o]
                self._sp[b, j] = pp[o]
            self._ss[b, len(ss):] = self.off[b] + ss[order[0]]  # padding: probability 0
            row = []
            for v in (opt0, worst0, rand0):  # mdp/base.py episodic_*_average_reward: sum_sn p * V[0, sn] / H
                acc = 0.0
                for s, p in zip(ss, pp):
                    acc += p * v[self.off[b] + s]
                row.append(acc / H)
            avgs.append(row)
        super().__init__(*(MP.from_scalars([a[j] for a in avgs]) for j in range(3)), n_check)

    def _regrets(self, V0: np.ndarray, start_abs: np.ndarray):
        H = self.H
        Rs = np.maximum(self.opt0[start_abs] - V0[start_abs], np.float32(0.0))
        minimal = self.opt0[start_abs] - self.worst0[start_abs]
        with np.errstate(all="ignore"):
            regret = Rs / H
            nr = np.where(self.is_training, regret / minimal * H, Rs / minimal)  # cached form once frozen (:562-566)
        return MP(regret, F32), MP(nr.astype(np.float32), F32)

    def update(self, t: int, T: int, V0: np.ndarray, last_start: np.ndarray, cum_reward, n_since: int, in_loop: bool):
        """One `_update_performance_logs(t)` (+ the optimality check when `in_loop`).  V0: flat float32 values at
        in-episode time 0 of the agents' greedy policies; last_start: per-instance start state of the logged episode."""
        V0 = np.asarray(V0, np.float32)
        start_abs = self.off + np.asarray(last_start, np.int64)
        regret, nr = self._regrets(V0, start_abs)
        epi = np.zeros(self.B)
        for j in range(self._ss.shape[1]):
            epi = epi + V0[self._ss[:, j]].astype(np.float64) * self._sp[:, j]
        self._accumulate_and_log(t, regret, nr, MP(epi, F64) / self.H, cum_reward, n_since)
        if in_loop:
            self._after_log(t, T, nr, 1e-4, lambda: self._regrets(V0, start_abs)[1])


class ContinuousVectorTracker(_VectorTracker):
    """Continuous regrets (agent_mdp_interaction.py:510-532).  `average_rewards(need)` returns the average rewards of
    the current greedy policies from the current states for the instances flagged in `need` (a list of numpy scalars:
    their type matters), e.g. colosseum_amd.markov_chain.AverageRewardCache."""

    episodic = False

    def __init__(self, optimal: MP, worst: MP, random: MP, n_check: int = 10):
        super().__init__(optimal, worst, random, n_check)
        assert (self.span.values64() > 0.0002).all()  # agent_mdp_interaction.py:379-382

    def reset(self):
        super().reset()
        B = self.B
        self._cached = np.zeros(B, bool)
        self._c_r, self._c_nr, self._c_avg = MP.weak(0.0, B), MP.weak(0.0, B), MP.weak(0.0, B)

    def _regrets_from(self, avg: MP):
        r = self.opt - avg
        r = MP.where(r.isclose_to_zero(1e-3), MP.weak(0.0, self.B), r)
        r = MP.where(r.values64() < 0, MP.weak(0.0, self.B), r)
        return r, r / self.span

    def update(self, t: int, T: int, average_rewards, cum_reward, n_since: int, in_loop: bool):
        need = self.is_training | ~self._cached
        avg = self._c_avg
        if need.any():
            vals = average_rewards(need)
            full = [0.0] * self.B
            for b, x in zip(np.flatnonzero(need), vals):
                full[b] = x
            avg = MP.where(need, MP.from_scalars(full), self._c_avg)
        r_new, nr_new = self._regrets_from(avg)
        r, nr = MP.where(need, r_new, self._c_r), MP.where(need, nr_new, self._c_nr)
        newly = ~self.is_training & ~self._cached  # first evaluation after the freeze is kept for the rest of the run
        if newly.any():
            self._c_r, self._c_nr = MP.where(newly, r, self._c_r), MP.where(newly, nr, self._c_nr)
            self._c_avg = MP.where(newly, avg, self._c_avg)
            self._cached |= newly
        self._accumulate_and_log(t, r, nr, avg, cum_reward, n_since)
        if in_loop:
            self._after_log(t, T, nr, 1e-5, lambda: nr)


def _csv_texts_of_slice(args):
    """Worker of `csv_texts_parallel`: the CSV texts of a slice of instances of one batch (pure numpy / str work)."""
    steps, cols, B = args
    log = BatchLog(B)
    log.steps = list(steps)
    log._final = cols
    text = log.text_columns()
    return [log.csv_text(b, text) for b in range(B)]


def csv_texts_parallel(logs: Sequence["BatchLog"], workers: int, chunk: int = 16):
    """{id(log): [csv text per instance]} -- the float -> shortest-repr text conversion is Python-level work
    (~1 us per value, 90 k values per instance at the benchmark's cadence), so it is spread over a process pool
    ("spawn": safe after the HIP runtime has been initialised)."""
    tasks, where = [], []
    for log in logs:
        fin = log.finalize()
        for b0 in range(0, log.B, chunk):
            b1 = min(log.B, b0 + chunk)
            cols = {n: (v[:, b0:b1], k if isinstance(k, int) else k[:, b0:b1]) for n, (v, k) in fin.items()}
            tasks.append((log.steps, cols, b1 - b0))
            where.append((id(log), b0))
    if workers > 1 and len(tasks) > 1:
        import multiprocessing as mp

        with mp.get_context("spawn").Pool(min(workers, len(tasks))) as pool:
            parts = pool.map(_csv_texts_of_slice, tasks)
    else:
        parts = [_csv_texts_of_slice(t) for t in tasks]
    out: Dict[int, list] = {}
    for (key, b0), texts in zip(where, parts):
        out.setdefault(key, {})[b0] = texts
    return {key: [t for b0 in sorted(d) for t in d[b0]] for key, d in out.items()}


# ----------------------------------------------------------------------------------------------------------------------
# The same indicators computed by the library in C++ (csrc/cmdp_tracker.h): what `cmdp_qlearning_run_logged` runs between
# kernels.  `native_log` wraps its output as a `BatchLog`; `loop_desc` builds the description both entry points take.
def loop_desc(T: int, log_every: int, n_check: int, baselines: Sequence[MP], max_time: float = np.inf, H: int = 0,
              opt0=None, worst0=None, start_pos=None, start_prob=None):
    """(CmdpLoopDesc, keep-alive list) -- `baselines` = (optimal, worst, random) average rewards as MP."""
    from .. import _lib as L

    B = len(baselines[0].v)
    val = np.ascontiguousarray(np.stack([m.values64() for m in baselines], 1), np.float64)
    kind = np.ascontiguousarray(np.stack([m.kinds() for m in baselines], 1), np.int32)  # 0 Python float, 1 float32, 2 float64
    d = L.CmdpLoopDesc()
    d.n_steps, d.log_every, d.n_check, d.horizon = int(T), int(log_every if log_every and log_every > 0 else 0), int(n_check), int(H)
    d.max_time = float(max_time) if np.isfinite(max_time) else 1e300
    keep = [val, kind]
    d.base_val, d.base_kind = L.ptr(val), L.ptr(kind)
    if H:
        o = np.ascontiguousarray(opt0, np.float32)
        w = np.ascontiguousarray(worst0, np.float32)
        sp = np.ascontiguousarray(start_pos, np.int64)
        pp = np.ascontiguousarray(start_prob, np.float64)
        d.kmax = int(sp.shape[1])
        d.opt0, d.worst0, d.start_pos, d.start_prob = L.ptr(o), L.ptr(w), L.ptr(sp), L.ptr(pp)
        keep += [o, w, sp, pp]
    assert B == len(val)
    return d, keep


def n_log_rows(T: int, log_every: int) -> int:
    return (len(range(log_every, T, log_every)) if log_every and log_every > 0 else 0) + 1


def native_log(B: int, steps: np.ndarray, values: np.ndarray, kinds: np.ndarray) -> BatchLog:
    """BatchLog over the arrays cmdp_qlearning_run_logged / cmdp_tracker_replay fill: values, kinds
    [n_logs][CMDP_LOG_COLUMNS][B]."""
    from .._lib import LOG_COLUMNS

    log = BatchLog(B)
    log.steps = [int(s) for s in steps]
    log._final = {name: (np.ascontiguousarray(values[:, c, :]), np.ascontiguousarray(kinds[:, c, :]).astype(np.int8))
                  for c, name in enumerate(LOG_COLUMNS)}
    log._cols = {name: None for name in LOG_COLUMNS}  # names() reads the keys
    return log
