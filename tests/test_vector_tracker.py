"""colosseum_amd.experiment.vector_tracker (the indicator code behind MDPLoop and the batched loops) against the
REFERENCE's own indicator code: golden G15 holds synthetic inputs and what agent_mdp_interaction.py:304-578 made of
them -- 17 logged columns per row, the numpy TYPE of every value (float32 / float64 decide the rounding of the
reference's scalars under NEP 50) and the training flag after every row, including runs that `_is_policy_optimal`
freezes.  CPU only."""
import csv
import io
import json

import numpy as np
import pytest

from conftest import load_golden
from colosseum_amd.experiment.vector_tracker import (F32, F64, WEAK, MP, ContinuousVectorTracker, EpisodicVectorTracker)


def _same(a, b, key):
    assert type(a) is type(b) or (isinstance(a, (int, np.integer)) and isinstance(b, (int, np.integer))), (key, type(a), type(b))
    assert (a == b) or (np.isnan(a) and np.isnan(b)), (key, a, b)


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


def _groups(setting):
    z, cases = load_golden("G15_indicators")
    by = {}
    for i, c in enumerate(cases):
        if c["setting"] == setting:
            by.setdefault(c["group"], []).append((("e" if setting == "episodic" else "c") + f"{i}_", c))
    return z, by


def _check_rows(z, members, tables, flags_seen):
    for b, (key, c) in enumerate(members):
        want, kinds = z[key + "rows"], z[key + "kinds"]
        assert len(tables[b]) == len(want)
        for i, row in enumerate(tables[b]):
            for j, name in enumerate(c["keys"]):
                got = row[name]
                if name == "steps":
                    assert int(got) == int(want[i, j])
                    continue
                assert float(got) == want[i, j] or (np.isnan(got) and np.isnan(want[i, j])), (key, i, name, got, want[i, j])
                assert (1 if isinstance(got, np.float32) else 2) == kinds[i, j], (key, i, name, type(got), kinds[i, j])
        np.testing.assert_array_equal(np.array([f[b] for f in flags_seen]), z[key + "is_training"], err_msg=key)


def test_episodic_vector_tracker_equals_reference_indicator_code():
    z, by = _groups("episodic")
    assert len(by) == 3
    n_frozen = 0
    for g, members in by.items():
        c0 = members[0][1]
        sizes = [len(z[k + "opt0"]) for k, _ in members]
        off = np.concatenate([[0], np.cumsum(sizes)])
        flat = [np.concatenate([z[k + n] for k, _ in members]) for n in ("opt0", "worst0", "rand0")]
        starts = [(z[k + "start_states"], z[k + "start_probs"]) for k, _ in members]
        vt = EpisodicVectorTracker(c0["H"], off, *flat, starts, c0["n_check"])
        ts = z[members[0][0] + "t"]
        flags = []
        for i, t in enumerate(ts):
            V0 = np.concatenate([z[k + "V0"][i] for k, _ in members])
            last = np.array([z[k + "last_start"][i] for k, _ in members])
            cum = np.array([z[k + "cum"][i] for k, _ in members])
            vt.update(int(t), c0["T"], V0, last, cum, int(z[members[0][0] + "n_since"][i]), bool(z[members[0][0] + "in_loop"][i]))
            flags.append(vt.is_training.copy())
        _check_rows(z, members, vt.tables(), flags)
        n_frozen += int((~vt.is_training).sum())
    assert n_frozen >= 4  # the freeze path of _is_policy_optimal is exercised


def test_continuous_vector_tracker_equals_reference_indicator_code():
    z, by = _groups("continuous")
    assert len(by) == 3
    n_frozen = 0
    for g, members in by.items():
        c0 = members[0][1]

        def mp(j):
            return MP.from_scalars([np.float32(z[k + "baselines"][j]) if z[k + "baseline_kinds"][j] == 1
                                    else np.float64(z[k + "baselines"][j]) for k, _ in members])

        vt = ContinuousVectorTracker(mp(0), mp(1), mp(2), c0["n_check"])
        ts = z[members[0][0] + "t"]
        flags = []
        for i, t in enumerate(ts):
            def averages(need, i=i):
                return [np.float32(z[k + "avg"][i]) if z[k + "avg_kinds"][i] == 1 else np.float64(z[k + "avg"][i])
                        for b, (k, _) in enumerate(members) if need[b]]
            cum = np.array([z[k + "cum"][i] for k, _ in members])
            vt.update(int(t), c0["T"], averages, cum, int(z[members[0][0] + "n_since"][i]), bool(z[members[0][0] + "in_loop"][i]))
            flags.append(vt.is_training.copy())
        _check_rows(z, members, vt.tables(), flags)
        n_frozen += int((~vt.is_training).sum())
    assert n_frozen >= 4


def test_csv_text_matches_dictwriter():
    """BatchLog.csv_text = what csv.DictWriter prints for the same rows (the reference's CSVLogger)."""
    z, by = _groups("episodic")
    members = by[0]
    c0 = members[0][1]
    sizes = [len(z[k + "opt0"]) for k, _ in members]
    off = np.concatenate([[0], np.cumsum(sizes)])
    flat = [np.concatenate([z[k + n] for k, _ in members]) for n in ("opt0", "worst0", "rand0")]
    vt = EpisodicVectorTracker(c0["H"], off, *flat, [(z[k + "start_states"], z[k + "start_probs"]) for k, _ in members], c0["n_check"])
    for i, t in enumerate(z[members[0][0] + "t"][:7]):
        vt.update(int(t), c0["T"], np.concatenate([z[k + "V0"][i] for k, _ in members]),
                  np.array([z[k + "last_start"][i] for k, _ in members]), np.array([z[k + "cum"][i] for k, _ in members]),
                  int(z[members[0][0] + "n_since"][i]), True)
    tables = vt.tables()
    for b in range(len(members)):
        rows = list(tables[b])
        buf = io.StringIO()
        w = csv.DictWriter(buf, fieldnames=sorted(rows[0].keys()))
        w.writeheader()
        for r in rows:
            w.writerow(r)
        assert vt.log.csv_text(b) == buf.getvalue()


# ---- the C++ tracker of the library (csrc/cmdp_tracker.h, what cmdp_qlearning_run_logged runs between kernels) ---------
def _native_rows(B, log, i_rows):
    return [[log.value(name, i, b) for name in log.names()] for b in range(B) for i in i_rows]


def _check_native(z, members, log, flags):
    from colosseum_amd._lib import LOG_COLUMNS

    for b, (key, c) in enumerate(members):
        want, kinds = z[key + "rows"], z[key + "kinds"]
        for i in range(len(want)):
            for j, name in enumerate(c["keys"]):
                if name == "steps":
                    assert log.steps[i] == int(want[i, j])
                    continue
                got = log.value(name, i, b)
                assert float(got) == want[i, j] or (np.isnan(got) and np.isnan(want[i, j])), (key, i, name, got, want[i, j])
                assert (1 if isinstance(got, np.float32) else 2) == kinds[i, j], (key, i, name, type(got), kinds[i, j])
        np.testing.assert_array_equal(flags[:, b].astype(bool), z[key + "is_training"], err_msg=key)
    assert set(LOG_COLUMNS) | {"steps"} == set(members[0][1]["keys"]) | {"steps_per_second"}


def test_native_tracker_equals_reference_indicator_code():
    """cmdp_tracker_replay (host-only entry point of libcmdp.so) on the inputs of golden G15: the rows, numpy types and
    training flags the reference's own indicator code produced."""
    import ctypes as C

    from colosseum_amd import _lib as L
    from colosseum_amd.experiment.vector_tracker import loop_desc, native_log

    lib = L.load()
    n_frozen = 0
    z, by = _groups("episodic")
    for g, members in by.items():
        c0 = members[0][1]
        B = len(members)
        sizes = [len(z[k + "opt0"]) for k, _ in members]
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        flat = [np.concatenate([z[k + n] for k, _ in members]) for n in ("opt0", "worst0", "rand0")]
        starts = [(z[k + "start_states"], z[k + "start_probs"]) for k, _ in members]
        vt = EpisodicVectorTracker(c0["H"], off, *flat, starts, c0["n_check"])  # prepares baselines and start tables
        d, keep = loop_desc(c0["T"], 100, c0["n_check"], (vt.opt, vt.worst, vt.rand), H=c0["H"], opt0=flat[0], worst0=flat[1],
                            start_pos=vt._ss, start_prob=vt._sp)
        ts = z[members[0][0] + "t"].astype(np.int64)
        n = len(ts)
        V0 = np.ascontiguousarray(np.stack([np.concatenate([z[k + "V0"][i] for k, _ in members]) for i in range(n)]), np.float32)
        start = np.ascontiguousarray(np.stack([[z[k + "last_start"][i] for k, _ in members] for i in range(n)]), np.int64)
        cum = np.ascontiguousarray(np.stack([[z[k + "cum"][i] for k, _ in members] for i in range(n)]), np.float64)
        in_loop = np.ascontiguousarray(z[members[0][0] + "in_loop"], np.uint8)
        n_since = np.ascontiguousarray(z[members[0][0] + "n_since"], np.int64)
        values = np.zeros((n, len(L.LOG_COLUMNS), B))
        kinds = np.zeros((n, len(L.LOG_COLUMNS), B), np.uint8)
        flags = np.zeros((n, B), np.uint8)
        L.check(lib.cmdp_tracker_replay(C.byref(d), B, 1, L.ptr(off), n, L.ptr(ts), L.ptr(in_loop), L.ptr(n_since), L.ptr(cum),
                                        L.ptr(V0), L.ptr(start), None, None, L.ptr(values), L.ptr(kinds), L.ptr(flags)))
        _check_native(z, members, native_log(B, ts, values, kinds), flags)
        n_frozen += int((flags[-1] == 0).sum())
    z, by = _groups("continuous")
    for g, members in by.items():
        c0 = members[0][1]
        B = len(members)

        def mp(j):
            return MP.from_scalars([np.float32(z[k + "baselines"][j]) if z[k + "baseline_kinds"][j] == 1
                                    else np.float64(z[k + "baselines"][j]) for k, _ in members])

        d, keep = loop_desc(c0["T"], 100, c0["n_check"], (mp(0), mp(1), mp(2)))
        ts = z[members[0][0] + "t"].astype(np.int64)
        n = len(ts)
        avg = np.ascontiguousarray(np.stack([[z[k + "avg"][i] for k, _ in members] for i in range(n)]), np.float64)
        akind = np.ascontiguousarray(np.stack([[1 if z[k + "avg_kinds"][i] == 1 else 0 for k, _ in members] for i in range(n)]), np.int32)
        cum = np.ascontiguousarray(np.stack([[z[k + "cum"][i] for k, _ in members] for i in range(n)]), np.float64)
        in_loop = np.ascontiguousarray(z[members[0][0] + "in_loop"], np.uint8)
        n_since = np.ascontiguousarray(z[members[0][0] + "n_since"], np.int64)
        values = np.zeros((n, len(L.LOG_COLUMNS), B))
        kinds = np.zeros((n, len(L.LOG_COLUMNS), B), np.uint8)
        flags = np.zeros((n, B), np.uint8)
        L.check(lib.cmdp_tracker_replay(C.byref(d), B, 0, None, n, L.ptr(ts), L.ptr(in_loop), L.ptr(n_since), L.ptr(cum),
                                        None, None, L.ptr(avg), L.ptr(akind), L.ptr(values), L.ptr(kinds), L.ptr(flags)))
        _check_native(z, members, native_log(B, ts, values, kinds), flags)
        n_frozen += int((flags[-1] == 0).sum())
    assert n_frozen >= 8
