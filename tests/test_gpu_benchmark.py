"""Config C4 in miniature: the reference's quick-test benchmarks (golden G11 holds their MDP parameterisations) for the
four in-scope families, run with on-device Q-learning agents, CSV files in the reference's layout, and one instance
cross-checked against the per-instance host loop."""
import csv
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from colosseum_amd import benchmark as bm

pytestmark = pytest.mark.gpu


def _configs(name):
    return json.load(open(os.path.join(GOLDEN, "G11_benchmark_configs.json")))[name]["mdp_configs"]


def test_parse_gin_roundtrip():
    text = 'prms_0/DeepSeaEpisodic.size=10\nprms_0/DeepSeaEpisodic.p_rand = 0.05\nprms_1/DeepSeaEpisodic.p_lazy=None\n' \
           'prms_1/DeepSeaEpisodic.optimal_distribution = ("beta", (1.0, 2.5))\n# comment\n'
    cfg = bm.parse_gin(text)
    assert cfg == {"DeepSeaEpisodic": {"prms_0": {"size": 10, "p_rand": 0.05},
                                       "prms_1": {"p_lazy": None, "optimal_distribution": ("beta", (1.0, 2.5))}}}


@pytest.mark.parametrize("name", ["benchmark_episodic_quick_test", "benchmark_continuous_quick_test"])
def test_quick_benchmark_runs_and_writes_reference_csv_layout(need_gpu, tmp_path, name):
    instances = bm.enumerate_instances(_configs(name), n_seeds=2)
    assert len(instances) == 8 and [i.seed for i in instances] == [0] * 4 + [1] * 4  # seed-major
    results = bm.run_instances(instances, n_steps=3000, log_every=1000)
    assert sorted(results) == list(range(8))
    bm.write_csv_logs(str(tmp_path), instances, results)
    for i, ins in enumerate(instances):
        rows = results[i]
        assert [r["steps"] for r in rows] == [1000, 2000, 2999]
        assert all(r["cumulative_regret"] >= 0 and r["normalized_cumulative_regret"] >= 0 for r in rows)
        assert rows[-1]["worst_normalized_cumulative_regret"] == pytest.approx(3000, rel=1e-5)
        f = tmp_path / "logs" / ins.label / f"seed{ins.seed}_logs.csv"
        got = list(csv.DictReader(open(f)))
        assert len(got) == 3 and list(got[0].keys()) == sorted(rows[0].keys()) and len(got[0]) == 18


def test_batched_run_equals_per_instance_host_loop(need_gpu):
    """Deterministic-reward instances of the batch take the MT_COMPAT path: their rows must equal GpuMDP + MDPLoop +
    the numpy agent, i.e. what the reference produces for that (MDP, agent, seed)."""
    from colosseum_amd.experiment import MDPLoop, make_mdp_spec
    from colosseum_amd.mdp import gpu_mdp
    from helpers_agents import QLearningEpisodic

    instances = [i for i in bm.enumerate_instances(_configs("benchmark_episodic_quick_test"), n_seeds=2)
                 if not i.mdp_kwargs.get("make_reward_stochastic")]
    assert len(instances) == 4
    results = bm.run_instances(instances, n_steps=2500, log_every=500)
    for i, ins in enumerate(instances):
        mdp = getattr(gpu_mdp, ins.mdp_cls)(seed=ins.seed, **ins.mdp_kwargs)
        agent = QLearningEpisodic(seed=ins.seed, mdp_specs=make_mdp_spec(mdp), optimization_horizon=2500,
                                  **bm.DEFAULT_AGENT_CONFIGS["QLearningEpisodic"])
        loop = MDPLoop(mdp, agent)
        loop.run(T=2500, log_every=500)
        assert len(loop.logger.data) == len(results[i])
        for got, ref in zip(results[i], loop.logger.data):
            for k in ref:
                if k != "steps_per_second":
                    assert float(got[k]) == pytest.approx(float(ref[k]), rel=1e-6, abs=1e-5), (ins.label, ins.seed, k)
        mdp.close()
