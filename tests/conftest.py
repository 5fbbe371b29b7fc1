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
