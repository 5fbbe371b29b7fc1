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


def test_time_limit_ledger_and_resume(need_gpu, tmp_path):
    """Run control of a benchmark: an exhausted time budget freezes training and is recorded in
    `time_exceeded.txt` next to the log file (experiment_instances.py:218-222), frozen instances keep logging; a second
    invocation skips the instances whose log file exists (experiment_instance.py:67-82, folder_structuring.py:102)."""
    instances = bm.enumerate_instances(_configs("benchmark_episodic_quick_test"), n_seeds=1)
    assert bm.unfinished_instances(str(tmp_path), instances) == [0, 1, 2, 3]
    # max_time 0: fewer than 0.5 s remain at the first logging step -> training stops there for every instance
    results = bm.run_instances(instances, n_steps=2000, log_every=500, max_time=0.0)
    bm.write_csv_logs(str(tmp_path), instances, results)
    for i, ins in enumerate(instances):
        assert results[i].last_training_step == 500 and len(results[i]) == 4
        d = tmp_path / "logs" / ins.label
        ledger = (d / "time_exceeded.txt").read_text()
        assert ledger == f"last training step at (500) for {d / f'seed{ins.seed}_logs.csv'}\n"
    # without a limit nothing is recorded, and the rows differ from the frozen run's (training went on)
    free = bm.run_instances(instances[:1], n_steps=2000, log_every=500)
    assert free[0].last_training_step == -1
    assert free[0][-1]["cumulative_regret"] != results[0][-1]["cumulative_regret"]
    # resume: everything is on disk -> nothing left to do; remove one file -> exactly that instance is run again
    assert bm.unfinished_instances(str(tmp_path), instances) == []
    os.remove(bm.log_file(str(tmp_path), instances[2]))
    todo = bm.unfinished_instances(str(tmp_path), instances)
    assert todo == [2]
    again = bm.run_instances(instances, n_steps=2000, log_every=500, skip=[i for i in range(4) if i not in todo])
    assert sorted(again) == [2]
    bm.write_csv_logs(str(tmp_path), instances, again)
    assert bm.unfinished_instances(str(tmp_path), instances) == []
    # the file prints float32 values with their shortest repr
    np.testing.assert_allclose(bm.read_summary(str(tmp_path), instances[2]), bm.summary_vector(again[2]), rtol=1e-6)


def test_c4_full_enumeration_of_the_four_default_suites(need_gpu):
    """Config C4 with the FULL instance enumeration of the reference's four default benchmark suites (every family x gin
    setting x 20 seeds = 1 000 instances; `benchmark/experiment_config.yml`) at a reduced step count, and one instance of
    EVERY (suite, MDP class, setting) -- the 16 deterministic-reward and the 34 Beta-reward ones alike -- cross-checked
    against the per-instance path `GpuMDP` + `MDPLoop` + numpy agent (bit-equal to the reference loop,
    tests/test_gpu_mdploop.py; its Beta rewards come from numpy's own `RandomState.beta` continuing the MDP's stream,
    tests/test_rewards.py).  The batch runs the reference's streams throughout: MT19937 transition samplers and, for Beta
    rewards, the reference's per-triple caches of 5000 samples (CMDP_FLAG_REWARD_CACHE) -- no self-comparison."""
    from colosseum_amd.experiment import MDPLoop, make_mdp_spec
    from colosseum_amd.mdp import gpu_mdp
    from helpers_agents import QLearningContinuous, QLearningEpisodic

    suites = ["benchmark_episodic_ergodic", "benchmark_episodic_communicating",
              "benchmark_continuous_ergodic", "benchmark_continuous_communicating"]
    n_steps, log_every = 1500, 500
    total = checked = checked_beta = 0
    settings = set()
    for suite in suites:
        allcfg = json.load(open(os.path.join(GOLDEN, "G11_benchmark_configs.json")))[suite]
        assert allcfg["experiment_config"]["n_seeds"] == 20
        instances = bm.enumerate_instances(allcfg["mdp_configs"], n_seeds=20)
        total += len(instances)
        results = bm.run_instances(instances, n_steps=n_steps, log_every=log_every, beta_rewards="reference")
        assert sorted(results) == list(range(len(instances)))
        seen = set()
        for i, ins in enumerate(instances):
            rows = results[i]
            assert [r["steps"] for r in rows] == [500, 1000, 1499]
            last = rows[-1]
            assert last["cumulative_regret"] >= 0 and last["worst_normalized_cumulative_regret"] == pytest.approx(n_steps, rel=1e-4)
            assert last["optimal_normalized_cumulative_expected_reward"] >= n_steps - 1  # t + optimal / (optimal - worst)
            key = (suite, ins.mdp_cls, ins.mdp_scope)
            settings.add(key)
            # one seed per setting, a different one from setting to setting
            if key in seen or ins.seed != len(seen) % 20:
                continue
            seen.add(key)
            mdp = getattr(gpu_mdp, ins.mdp_cls)(seed=ins.seed, **ins.mdp_kwargs)
            agent_cls = QLearningEpisodic if ins.agent_cls == "QLearningEpisodic" else QLearningContinuous
            agent = agent_cls(seed=ins.seed, mdp_specs=make_mdp_spec(mdp), optimization_horizon=n_steps,
                              **bm.DEFAULT_AGENT_CONFIGS[ins.agent_cls])
            loop = MDPLoop(mdp, agent)
            loop.run(T=n_steps, log_every=log_every)
            for got, ref in zip(rows, loop.logger.data):
                for k in ref:
                    if k != "steps_per_second":
                        assert float(got[k]) == pytest.approx(float(ref[k]), rel=1e-6, abs=1e-5), (key, ins.seed, k)
            checked += 1
            checked_beta += bool(ins.mdp_kwargs.get("make_reward_stochastic"))
            mdp.close()
    print(f"C4: {total} instances over {len(settings)} (suite, class, setting) triples; {checked} cross-checked against "
          f"the per-instance host loop, {checked_beta} of them with Beta rewards")
    assert total == 1000 and len(settings) == 50 and checked == 50 and checked_beta == 34


def test_c4_philox_reward_mode_is_batch_independent(need_gpu):
    """The throughput mode of the runner (`beta_rewards="philox"`: Beta rewards sampled on the device from the instance's
    counter-based stream) is distribution-exact, not stream-exact; what it must guarantee is that an instance's rows do
    not depend on what else is in its device batch."""
    allcfg = json.load(open(os.path.join(GOLDEN, "G11_benchmark_configs.json")))["benchmark_continuous_communicating"]
    instances = [i for i in bm.enumerate_instances(allcfg["mdp_configs"], n_seeds=3)
                 if i.mdp_cls == "FrozenLakeContinuous" and i.mdp_kwargs.get("make_reward_stochastic")]
    assert len(instances) >= 3
    results = bm.run_instances(instances, n_steps=1500, log_every=500, beta_rewards="philox")
    alone = bm.run_instances(instances[1:2], n_steps=1500, log_every=500, beta_rewards="philox")[0]
    for got, ref in zip(results[1], alone):
        for k in ref:
            if k != "steps_per_second":
                assert got[k] == ref[k] and type(got[k]) is type(ref[k]), (k,)
    exact = bm.run_instances(instances[1:2], n_steps=1500, log_every=500, beta_rewards="reference")[0]
    assert exact[-1]["cumulative_reward"] != alone[-1]["cumulative_reward"]  # other streams, same distribution
