"""The C-ABI library loads on a box without a GPU, exports every symbol include/cmdp.h declares, and fails
loudly (no CPU fallback) when asked to compute without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from colosseum_amd import _lib as L


def _declared():
    src = open(os.path.join(ROOT, "include", "cmdp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cmdp_[a-z0-9_]+)\s*\(", src)))


def test_header_and_library_agree():
    lib = L.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/cmdp.h but not exported by libcmdp.so"
    assert sorted(L.EXPORTS) == names
    assert lib.cmdp_version() == 1


def test_struct_layout_matches_header():
    assert C.sizeof(L.CmdpDesc) == 6 * 4 + 2 * 8 + 18 * 8


def test_no_silent_cpu_fallback():
    lib = L.load()
    if lib.cmdp_device_count() > 0:
        pytest.skip("a GPU is visible; the no-device path cannot be exercised")
    from colosseum_amd.batched import BatchedMDP
    from colosseum_amd.mdp import make_model

    with pytest.raises(L.CmdpError) as ei:
        BatchedMDP([make_model("DeepSeaEpisodic", seed=0, size=4)])
    assert ei.value.code == L.ERR_NO_DEVICE
    from colosseum_amd import dynamic_programming as dp

    m = make_model("DeepSeaEpisodic", seed=0, size=4)
    T, R = m.dense()
    with pytest.raises(L.CmdpError):
        dp.episodic_value_iteration(m.H, T, R)


def test_create_argument_validation():
    lib = L.load()
    h = C.c_void_p()
    d = L.CmdpDesc()
    assert lib.cmdp_create(C.byref(h), C.byref(d)) == L.ERR_INVALID
    assert b"n_instances" in lib.cmdp_last_error()
    assert lib.cmdp_create(None, None) == L.ERR_INVALID


def test_product_never_imports_the_oracle():
    """A product path that routes through the oracle would void every parity claim: no file of the package may import,
    load or execute anything under oracle/ (comments may mention "the oracle" as what the tests compare with)."""
    pattern = re.compile(r"^\s*(from|import)\s+oracle\b|libcmdp_oracle|oracle[/\\.](oracle|cmdp_oracle|ref_env|gen_golden)|"
                         r"['\"]oracle['\"]", re.M)
    checked = 0
    for base, _, files in os.walk(os.path.join(ROOT, "colosseum_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(base, f)).read()
                assert not pattern.search(src), f"{os.path.join(base, f)} reaches into oracle/"
                checked += 1
    assert checked >= 25
    for f in ("bench.py",):  # the bench may use it as the checker / CPU baseline only, after the timed region
        src = open(os.path.join(ROOT, f)).read()
        assert src.index("from oracle import oracle") > src.index("elapsed, launch_ms = timed_launches")
