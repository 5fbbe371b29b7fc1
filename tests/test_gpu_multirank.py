"""The multi-rank entry points end to end on the one GPU of the test box (ranks share the device, gloo rendezvous on
127.0.0.1): `bench.py --gpus 2` spawning its own ranks and under torch.distributed.run, `tools/run_benchmark.py` (config
C4: contiguous shards of the instance list, CSV files identical to a single-process run) and `tools/run_c5.py` (config
C5: target ranges, one all_reduce(max))."""
import json
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

ENV = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
ENV.pop("RANK", None)
ENV.pop("WORLD_SIZE", None)


def _run(cmd, timeout=600):
    p = subprocess.run(cmd, cwd=ROOT, env=ENV, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-4000:]
    return p.stdout


def _torchrun(n, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port)]


SMALL = ["--instances", "2048", "--launch-steps", "600", "--steps", "3", "--warmup", "1", "--dense-instances", "0",
         "--vi-instances", "32", "--no-cpu", "--share-gpu", "--dist-backend", "gloo"]


def test_bench_two_ranks_without_and_with_a_launcher(need_gpu):
    one = json.loads(_run([sys.executable, "bench.py", "--gpus", "1"] + SMALL).strip().splitlines()[-1])
    assert one["n_gpus"] == 1 and one["scaling"] == "weak"
    # no WORLD_SIZE in the environment: bench.py starts its two ranks itself
    two = json.loads(_run([sys.executable, "bench.py", "--gpus", "2"] + SMALL).strip().splitlines()[-1])
    # the driver's form
    tr = json.loads(_run(_torchrun(2, 29731) + ["bench.py", "--gpus", "2"] + SMALL).strip().splitlines()[-1])
    for line in (two, tr):
        assert line["n_gpus"] == 2 and line["metric"] == one["metric"] and line["config"]["instances_per_gpu"] == 2048
        assert line["value"] > 0 and "gather_ms" in line and line["vi"]["total_sweeps"] > one["vi"]["total_sweeps"]
        assert line["roofline"]["frac"] <= 1.0


def test_run_benchmark_two_ranks_writes_the_single_process_files(need_gpu, tmp_path):
    args = ["tools/run_benchmark.py", "--configs-json", os.path.join(GOLDEN, "G11_benchmark_configs.json"), "--benchmark",
            "benchmark_episodic_quick_test", "--steps", "1500", "--seeds", "3", "--log-every", "500"]
    s1 = json.loads(_run([sys.executable] + args + ["--out", str(tmp_path / "one")]).strip().splitlines()[-1])
    s2 = json.loads(_run(_torchrun(2, 29732) + args + ["--out", str(tmp_path / "two"), "--share-gpu", "--dist-backend", "gloo"])
                    .strip().splitlines()[-1])
    assert s1["instances"] == s2["instances"] == 12
    assert s1["mean_normalized_cumulative_regret"] == s2["mean_normalized_cumulative_regret"]
    n = 0
    for d, _, files in os.walk(tmp_path / "one" / "logs"):
        for f in files:
            a = open(os.path.join(d, f)).read()
            b = open(os.path.join(str(d).replace(str(tmp_path / "one"), str(tmp_path / "two")), f)).read()
            ha, hb = a.split("\n", 1)[0].split(","), b.split("\n", 1)[0].split(",")
            assert ha == hb
            k = ha.index("steps_per_second")  # wall clock
            for la, lb in zip(a.strip().split("\n")[1:], b.strip().split("\n")[1:]):
                ca, cb = la.split(","), lb.split(",")
                assert ca[:k] + ca[k + 1:] == cb[:k] + cb[k + 1:], f
            n += 1
    assert n == 12
    # a second invocation finds every log file and runs nothing (the reference's resume)
    s3 = json.loads(_run([sys.executable] + args + ["--out", str(tmp_path / "one")]).strip().splitlines()[-1])
    assert s3["skipped_existing"] == 12 and s3["mean_normalized_cumulative_regret"] == pytest.approx(s1["mean_normalized_cumulative_regret"], rel=1e-6)


def test_run_c5_two_ranks_give_the_single_process_diameter(need_gpu):
    args = ["tools/run_c5.py", "--room-size", "6", "--n-rooms", "4"]
    one = json.loads(_run([sys.executable] + args).strip().splitlines()[-1])
    two = json.loads(_run(_torchrun(2, 29733) + args + ["--share-gpu", "--dist-backend", "gloo"]).strip().splitlines()[-1])
    assert two["world"] == 2 and two["targets_this_rank"] * 2 >= one["n_states"] - 1
    assert one["diameter"] == two["diameter"] and one["diameter"] > 0


def test_c4_at_its_real_length(need_gpu, tmp_path):
    """Config C4 as the reference defines it (benchmark/experiment_config.yml): the four default suites, 1 000 instances x
    500 000 steps, a log row every 100 steps -- 5 000 rows per instance, the reference's CSV files on disk.  Run in the
    throughput mode of the Beta rewards (device-sampled; the reference-exact mode is checked setting by setting in
    tests/test_gpu_benchmark.py and costs ~90 s here): the summary of that mode is deterministic and pinned."""
    import glob

    args = ["tools/run_benchmark.py", "--configs-json", os.path.join(GOLDEN, "G11_benchmark_configs.json"),
            "--benchmark", "benchmark_episodic_ergodic", "--benchmark", "benchmark_episodic_communicating",
            "--benchmark", "benchmark_continuous_ergodic", "--benchmark", "benchmark_continuous_communicating",
            "--out", str(tmp_path / "c4"), "--beta-rewards", "philox"]
    s = json.loads(_run([sys.executable] + args, timeout=500).strip().splitlines()[-1])
    assert s["instances"] == s["run"] == 1000 and s["steps_each"] == 500000 and s["skipped_existing"] == 0
    assert s["mean_normalized_cumulative_regret"] == pytest.approx(379778.62248119974, rel=1e-12)
    files = glob.glob(str(tmp_path / "c4" / "logs" / "*" / "seed*_logs.csv"))
    assert len(files) == 1000
    for f in files[::97]:
        lines = open(f, newline="").read().split("\r\n")
        assert len(lines) == 5002 and lines[-1] == "" and lines[0].startswith("cumulative_expected_reward,")
        assert lines[1].split(",")[lines[0].split(",").index("steps")] == "100"
        assert lines[5000].split(",")[lines[0].split(",").index("steps")] == "499999"
    print("C4 at full length: %.1f s wall, %.3g agent steps/s" % (s["wall_s"], s["agent_steps_per_s"]))
