"""The batched (numpy-vectorised) indicator code must reproduce the scalar MDPLoop indicator code row by row, value by
value AND type by type (float32 / float64 decide the rounding of the reference's scalars under NEP 50).  CPU only."""
import csv
import io

import numpy as np
import pytest

from colosseum_amd.experiment.batched_loop import _ContinuousTracker, _ContinuousView, _InstanceView, _Tracker
from colosseum_amd.experiment.vector_tracker import (F32, F64, WEAK, MP, ContinuousVectorTracker, EpisodicVectorTracker)


def _same(a, b, key):
    assert type(a) is type(b) or (isinstance(a, (int, np.integer)) and isinstance(b, (int, np.integer))), (key, type(a), type(b))
    assert (a == b) or (np.isnan(a) and np.isnan(b)), (key, a, b)


def _ring(tr, t, T, in_loop):
    if in_loop:  # agent_mdp_interaction.py:265-288
        tr._latest_expected_regrets.append(tr._normalized_regret)
        if len(tr._latest_expected_regrets) > tr._n_steps_to_check_for_agent_optimality:
            tr._latest_expected_regrets.pop(0)
        if tr._is_training and t > 0.2 * T and tr._is_policy_optimal():
            tr._is_training = False


def test_mp_arithmetic_matches_numpy_scalars():
    rng = np.random.default_rng(0)
    n = 300
    def rand_scalars():
        out = []
        for _ in range(n):
            x = float(rng.normal() * 10.0 ** int(rng.integers(-3, 6)))
            k = rng.integers(0, 3)
            out.append(x if k == 0 else np.float32(x) if k == 1 else np.float64(x))
        return out
    a, b = rand_scalars(), rand_scalars()
    A, Bm = MP.from_scalars(a), MP.from_scalars(b)
    for op in (lambda x, y: x + y, lambda x, y: x - y, lambda x, y: x * y, lambda x, y: x / y,
               lambda x, y: 3 * x - y, lambda x, y: (x - 7 * y) / y, lambda x, y: 0.25 - x):
        got = op(A, Bm)
        for i in range(n):
            want = op(a[i], b[i])
            _same(got.scalar(i), want, i)
    r = A.round5()
    for i in range(n):
        want = np.round(a[i], 5)
        _same(r.scalar(i), want, i)
    # uniform kinds take the native path
    x32 = MP(np.float32(rng.normal(size=n)), F32)
    w = MP(rng.normal(size=n) * 1e3, WEAK)
    got = (w - 5 * x32) / x32
    for i in range(n):
        _same(got.scalar(i), (float(w.v[i]) - 5 * np.float32(x32.v[i])) / np.float32(x32.v[i]), i)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_episodic_vector_tracker_equals_scalar_trackers(seed):
    rng = np.random.default_rng(seed)
    B, H, n_check, T, log_every = 7, 6, 4, 4000, 100
    sizes = rng.integers(5, 12, B)
    off = np.concatenate([[0], np.cumsum(sizes)])
    worst0 = rng.random(off[-1]).astype(np.float32)
    rand0 = (worst0 + rng.random(off[-1]) * 2).astype(np.float32)
    opt0 = (rand0 + 0.1 + rng.random(off[-1]) * 3).astype(np.float32)
    starts = []
    for b in range(B):
        k = int(rng.integers(1, 4))
        ss = rng.choice(sizes[b], k, replace=False)
        pp = rng.dirichlet(np.ones(k))
        starts.append((ss.astype(np.int64), pp))
    vt = EpisodicVectorTracker(H, off, opt0, worst0, rand0, starts, n_check)
    scal = []
    for b in range(B):
        sl = slice(off[b], off[b + 1])
        view = _InstanceView(H, starts[b][0], starts[b][1], opt0[sl], worst0[sl], rand0[sl])
        ssd = np.zeros(sizes[b])
        ssd[starts[b][0]] = starts[b][1]
        tr = _Tracker(view, ssd, n_check)
        tr._reset_run_variables()
        scal.append(tr)
    vt.reset()
    V0 = rand0.copy()
    cum = np.zeros(B)
    logs = list(range(log_every, T, log_every)) + [T - 1]
    n_since = 0
    for li, t in enumerate(logs):
        in_loop = t != T - 1
        # agents improve; instances 0..3 become optimal (exactly, or within the tolerance of the optimality check)
        frozen = ~vt.is_training
        for b in range(B):
            if frozen[b]:
                continue
            sl = slice(off[b], off[b + 1])
            if b < 2 and li > 12:
                V0[sl] = opt0[sl]
            elif b < 4 and li > 15:
                V0[sl] = opt0[sl] - np.float32(1e-6) * rng.random(sizes[b]).astype(np.float32)
            else:
                V0[sl] = V0[sl] + (opt0[sl] - V0[sl]) * np.float32(0.1 * rng.random())
        cum = cum + rng.random(B) * log_every
        n_since = log_every if li else log_every + 0
        start = np.array([rng.choice(starts[b][0]) for b in range(B)])
        vt.update(t, T, V0.copy(), start, cum, n_since, in_loop)
        for b, tr in enumerate(scal):
            tr._mdp.last_starting_node = int(start[b])
            tr.set_evaluation(V0[off[b]:off[b + 1]].copy())
            tr._cumulative_reward = float(cum[b])
            tr._n_steps_since_last_log = n_since
            tr._update_performance_logs(t)
            _ring(tr, t, T, in_loop)
        assert [tr._is_training for tr in scal] == vt.is_training.tolist(), t
    assert not vt.is_training[:2].any() and vt.is_training[4:].all()
    tables = vt.tables()
    text = vt.log.text_columns()
    for b, tr in enumerate(scal):
        assert len(tables[b]) == len(tr.logger.data) == len(logs)
        for got, ref in zip(tables[b], tr.logger.data):
            assert sorted(got) == sorted(ref)
            for k in ref:
                if k != "steps_per_second":
                    _same(got[k], ref[k], (b, k, ref["steps"]))
        # CSV text of the instance == csv.DictWriter on the scalar rows
        buf = io.StringIO()
        w = csv.DictWriter(buf, fieldnames=sorted(tr.logger.data[0]))
        w.writeheader()
        for r in tr.logger.data:
            w.writerow({k: np.array(v) for k, v in r.items()})
        drop = lambda s: [",".join(c for i, c in enumerate(line.split(",")) if i != sorted(r).index("steps_per_second"))
                          for line in s.split("\r\n")]
        assert drop(vt.log.csv_text(b, text)) == drop(buf.getvalue())


@pytest.mark.parametrize("seed", [0, 1])
def test_continuous_vector_tracker_equals_scalar_trackers(seed):
    rng = np.random.default_rng(100 + seed)
    B, n_check, T, log_every = 9, 3, 3000, 100
    def scalar(x, k):
        return np.float32(x) if k == 1 else np.float64(x)
    opt = [scalar(0.6 + 0.3 * rng.random(), rng.integers(1, 3)) for _ in range(B)]
    worst = [scalar(0.05 * rng.random(), rng.integers(1, 3)) for _ in range(B)]
    rand = [scalar(0.2 + 0.1 * rng.random(), rng.integers(1, 3)) for _ in range(B)]
    vt = ContinuousVectorTracker(MP.from_scalars(opt), MP.from_scalars(worst), MP.from_scalars(rand), n_check)
    scal = []
    for b in range(B):
        tr = _ContinuousTracker(_ContinuousView(opt[b], worst[b], rand[b], {}), n_check)
        tr._reset_run_variables()
        scal.append(tr)
    vt.reset()
    cum = np.zeros(B)
    logs = list(range(log_every, T, log_every)) + [T - 1]
    calls = []
    for li, t in enumerate(logs):
        in_loop = t != T - 1
        avgs = []
        for b in range(B):
            k = rng.integers(1, 3)  # the type of the agent's average reward changes with the chain structure
            if b < 3 and li > 8:
                x = float(opt[b]) - (0.0 if b == 0 else 5e-4 * rng.random())  # within the 1e-3 snap of the regret
            elif b == 3 and li > 8:
                x = float(opt[b]) + 0.01  # negative regret -> clipped to the int 0
            else:
                x = float(rand[b]) + (float(opt[b]) - float(rand[b])) * min(1.0, li / 20) * rng.random()
            avgs.append(scalar(x, k))
        cum = cum + rng.random(B) * log_every
        n_since = log_every

        def averages(need, avgs=avgs):
            calls.append(need.copy())
            return [avgs[b] for b in np.flatnonzero(need)]

        was_training = vt.is_training.copy()
        vt.update(t, T, averages, cum, n_since, in_loop)
        for b, tr in enumerate(scal):
            tr._avg = avgs[b]
            tr._cumulative_reward = float(cum[b])
            tr._n_steps_since_last_log = n_since
            tr._update_performance_logs(t)
            _ring(tr, t, T, in_loop)
        assert [tr._is_training for tr in scal] == vt.is_training.tolist(), t
    assert not vt.is_training[:4].any() and vt.is_training[4:].all()
    assert not calls[-1][:4].any()  # frozen instances are no longer evaluated
    for b, tr in enumerate(scal):
        for got, ref in zip(vt.tables()[b], tr.logger.data):
            for k in ref:
                if k != "steps_per_second":
                    _same(got[k], ref[k], (b, k, ref["steps"]))


@pytest.mark.timeout(300)
def test_parallel_csv_formatting_equals_serial():
    from colosseum_amd.experiment.vector_tracker import BatchLog, csv_texts_parallel

    B, n = 24, 200
    log = BatchLog(B)
    rng = np.random.default_rng(0)
    for t in range(n):
        log.append(t, {f"c{i}": MP(np.round(rng.random(B) * 1000, 5).astype(np.float32 if i % 2 else np.float64),
                                  F32 if i % 2 else F64) for i in range(5)})
    tc = log.text_columns()
    ref = [log.csv_text(b, tc) for b in range(B)]
    assert csv_texts_parallel([log], 1)[id(log)] == ref
    assert csv_texts_parallel([log], 3, chunk=5)[id(log)] == ref
