"""A seeded slice of tools/fuzz_parity.py: 200 random MDPs over the seven generated families (random sizes, p_rand,
p_lazy, start-state counts, settings), every one checked GPU against oracle bit for bit -- trajectories in both RNG
modes, visit counts, discounted VI under both schemes, episodic VI, diameter through both kernels, the average-reward
kernel against the host restatement.  (The full sweep: `python tools/fuzz_parity.py 240` -- 5 276 MDPs, no mismatch.)"""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_random_mdps_gpu_equals_oracle(need_gpu):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(n_cases=200, seed=12345) == 200
