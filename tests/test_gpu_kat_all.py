"""EVERY known answer the reference's authors left in `benchmark/cached_hardness_measures/` (golden G5: all fourteen
class folders, the four measures diameter / value_norm / suboptimal_gaps / n_states, every seed, no size filter)
through the HIP path: `colosseum_amd.hardness` on batches of all the MDPs the files name.

The cached files were written with the real numba / sparse / gym stack over several versions of the reference; G5 keeps
a file only if today's reference constructor reproduces its name from the parsed keywords (oracle/gen_golden.py g5).
A report with the row counts and the largest deviations per (class, measure) is written to gpurun_out/ when that
directory exists."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from colosseum_amd import hardness
from colosseum_amd.mdp import make_model

pytestmark = pytest.mark.gpu

# Tolerances.  The files print float32 values with 8 significant digits; both measures sit on value iterations that stop
# at max|dV| < 1e-3, so a row can agree to the printed digits only when the solver behind the file stopped at the very
# same sweep as ours.  TIGHT is that case and must hold for almost every row; LOOSE bounds what the stopping rule itself
# leaves open and must hold for EVERY row:
#  * diameter, continuous: every target runs to the stopping rule on both sides -> tight only.
#  * diameter, episodic: the authors' files come from two reference paths, per-target convergence (multi-process, what
#    the HIP kernels do) and the single-thread loop with the running-maximum early exit at 1e-2 (order dependent;
#    SURVEY 8e allows 0.05 for it); RiverSwim chains with p_lazy / p_rand >= 0.4 contract so slowly that float32
#    rounding order moves the stopping sweep of hitting times in the thousands (relative 3e-5).
#  * value_norm: max_{s,a} of a float32 standard deviation of V (|V| up to 100): it moves by up to 4e-4 relative when the
#    solve stops a few sweeps earlier or later (measured: SimpleGridEpisodic 16, eps 0.8e-3 .. 1.2e-3).
TIGHT = {
    "diameter": dict(rel=5e-6, abs=1e-3),
    "value_norm": dict(rel=5e-6, abs=2e-6),
    "suboptimal_gaps": dict(rel=2e-5, abs=0.0),
}
LOOSE = {
    "diameter": dict(rel=5e-5, abs=5e-2),
    "value_norm": dict(rel=5e-4, abs=2e-6),
    "suboptimal_gaps": dict(rel=2e-5, abs=0.0),
}
MIN_TIGHT_FRACTION = {"diameter": 0.985, "value_norm": 0.94, "suboptimal_gaps": 1.0}


def _load():
    rows = json.load(open(os.path.join(GOLDEN, "G5_hardness_kat.json")))
    models, index = [], {}
    for r in rows:
        k = (r["cls"], json.dumps(r["kwargs"], sort_keys=True))
        if k not in index:
            index[k] = len(models)
            models.append(make_model(r["cls"], **r["kwargs"]))
        r["model"] = index[k]
    return rows, models


def test_every_cached_hardness_value(need_gpu):
    rows, models = _load()
    assert len(rows) >= 2400 and len({r["cls"] for r in rows}) >= 12
    got = {}
    for measure, fn in (("diameter", hardness.diameter), ("value_norm", hardness.value_norm),
                        ("suboptimal_gaps", hardness.sum_reciprocals_suboptimality_gaps)):
        ids = sorted({r["model"] for r in rows if r["measure"] == measure})
        vals = fn([models[i] for i in ids])
        got[measure] = dict(zip(ids, vals.tolist()))
    got["n_states"] = {r["model"]: models[r["model"]].n_states for r in rows if r["measure"] == "n_states"}

    report, failures, tight, total = {}, [], {}, {}
    for r in rows:
        g, want, ms = got[r["measure"]][r["model"]], r["value"], r["measure"]
        key = "%s/%s" % (r["cls"], ms)
        rep = report.setdefault(key, dict(rows=0, tight=0, max_abs=0.0, max_rel=0.0, failed=0))
        rep["rows"] += 1
        total[ms] = total.get(ms, 0) + 1
        err = abs(g - want)
        rep["max_abs"] = max(rep["max_abs"], err)
        rep["max_rel"] = max(rep["max_rel"], err / max(abs(want), 1e-30))
        if ms == "n_states":
            ok = is_tight = g == want
        else:
            continuous_diameter = ms == "diameter" and "Continuous" in r["cls"]
            is_tight = err <= max(TIGHT[ms]["abs"], TIGHT[ms]["rel"] * abs(want))
            ok = is_tight or (not continuous_diameter and err <= max(LOOSE[ms]["abs"], LOOSE[ms]["rel"] * abs(want)))
        rep["tight"] += bool(is_tight)
        tight[ms] = tight.get(ms, 0) + bool(is_tight)
        if not ok:
            rep["failed"] += 1
            failures.append(dict(file=r["file"], got=g, want=want, hash_match=r["hash_match"]))
    summary = dict(rows=len(rows), models=len(models), failed=len(failures), tight=tight, total=total,
                   per_class_measure=report, failures=failures[:200])
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump(summary, open(os.path.join(out, "g5_all_rows_report.json"), "w"), indent=1)
    print("G5 through the HIP path: %d rows, %d MDPs; to the printed digits: %s; outside the stopping-rule bound: %d"
          % (len(rows), len(models), {m: "%d/%d" % (tight[m], total[m]) for m in total}, len(failures)))
    assert not failures, failures[:10]
    for ms, frac in MIN_TIGHT_FRACTION.items():
        assert tight[ms] >= frac * total[ms], (ms, tight[ms], total[ms])
    assert tight["n_states"] == total["n_states"]
