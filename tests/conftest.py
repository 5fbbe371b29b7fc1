import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cases = json.loads(str(z["cases"]))
    return z, cases


def gpu_available():
    try:
        from colosseum_amd import _lib

        return _lib.load().cmdp_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def need_gpu():
    """The product path has no CPU fallback: a -m gpu test on a box without a device must fail, not skip."""
    from colosseum_amd import _lib

    assert _lib.load().cmdp_device_count() > 0, "no HIP device visible to libcmdp.so"


def assert_csv_matches(got: str, ref: str, min_identical: float = 0.9):
    """A log file against one the reference's own CSVLogger wrote (golden G7/G10 `csv_text`): identical header, line
    count and line endings; every cell either the same TEXT or (float32 DP behind a 5-decimal rounding) the same number
    within 2e-6 relative / 2e-5 absolute; `steps_per_second` is wall clock.  At least `min_identical` of the cells must
    be byte-identical."""
    assert got.count("\r\n") == ref.count("\r\n") and got.endswith("\r\n") == ref.endswith("\r\n")
    g, r = got.split("\r\n"), ref.split("\r\n")
    assert g[0] == r[0], (g[0], r[0])
    assert len(g) == len(r)
    names = g[0].split(",")
    same = total = 0
    for lg, lr in zip(g[1:], r[1:]):
        if not lg and not lr:
            continue
        cg, cr = lg.split(","), lr.split(",")
        assert len(cg) == len(cr) == len(names)
        for name, a, b in zip(names, cg, cr):
            if name == "steps_per_second":
                continue
            total += 1
            if a == b:
                same += 1
            else:
                assert abs(float(a) - float(b)) <= max(2e-5, 2e-6 * abs(float(b))), (name, a, b)
    assert same >= min_identical * total, (same, total)
    return same, total
