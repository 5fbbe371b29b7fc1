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
    # the bench contract's objects, and the hardware-anchored fields of the episode-parallel kernel's roofline
    r = one["roofline"]
    assert r["kernel"] == "k_rollout_epi" and r["bound"] == "valu_issue" and one["config"]["settle_launches"] == 80
    for k in ("achieved", "peak", "frac", "traffic", "frac_at_measured_int_rate", "lds_atomic_pipe_frac", "step_loop_frac", "kernel_ms"):
        assert k in r, k
    assert 0.0 < r["lds_atomic_pipe_frac"] < 1.0 and 0.0 < r["step_loop_frac"] < 1.0
    assert set(r["kernel_ms"]) == {"k_rollout_epi", "k_reward_scan"} and max(r["kernel_ms"].values()) <= one["ms_per_step"] * 1.5
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


def test_c4_at_its_real_length_reference_streams(need_gpu, tmp_path):
    """Config C4 at full length in the mode whose rows ARE the reference's (the runner's default): MT19937 transition samplers
    and the reference's per-triple caches of 5000 Beta samples filled from every MDP's own numpy stream
    (colosseum/mdp/base.py:1187-1207; budget colosseum/benchmark/experiment_config.yml:1-4).  The summary is pinned, and five
    instances -- spread over the suites, Beta-reward settings among them, every one crossing cache refills and MDPLoop's
    training freeze where it happens -- are re-run alone through the per-instance path `GpuMDP` + `MDPLoop` + numpy agent
    (bit-equal to the reference loop on the goldens G7 / G10 / G17), set up for the full run and followed for its first
    40 000 - 120 000 steps (the Python loop takes 0.5-2.3 ms per step): those rows of the batch's CSV files must be theirs,
    column for column."""
    import csv

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    suites = ["benchmark_episodic_ergodic", "benchmark_episodic_communicating", "benchmark_continuous_ergodic",
              "benchmark_continuous_communicating"]
    args = ["tools/run_benchmark.py", "--configs-json", os.path.join(GOLDEN, "G11_benchmark_configs.json")]
    for b in suites:
        args += ["--benchmark", b]
    args += ["--out", str(tmp_path / "c4r"), "--beta-rewards", "reference"]
    s = json.loads(_run([sys.executable] + args, timeout=600).strip().splitlines()[-1])
    assert s["instances"] == s["run"] == 1000 and s["steps_each"] == 500000 and s["skipped_existing"] == 0
    assert s["mean_normalized_cumulative_regret"] == pytest.approx(378726.1747486397, rel=1e-12)
    print("C4 at full length, reference streams: %.1f s wall" % s["wall_s"])
    # five instances alone (side by side, a process each) against the files the batch run wrote: the per-instance loop is set
    # up for all 500 000 steps and cut after 120 000 (deterministic rewards: past MDPLoop's freeze point 0.2 T) / 40 000 steps
    # (Beta rewards: reward caches of 5000 samples refilled several times) -- the loop is causal, its rows are the full run's
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor

    from helpers_c4 import run_instance_alone

    picks = [("benchmark_episodic_ergodic", 3, 120000), ("benchmark_episodic_communicating", 41, 40000),
             ("benchmark_continuous_ergodic", 7, 40000), ("benchmark_continuous_communicating", 123, 120000),
             ("benchmark_episodic_ergodic", 200, 40000)]
    with ProcessPoolExecutor(max_workers=5, mp_context=mp.get_context("spawn")) as pool:
        alone = list(pool.map(run_instance_alone, [(su, i, 500000, 100, GOLDEN, stop) for su, i, stop in picks]))
    for (suite, idx, stop), res in zip(picks, alone):
        # (several suites in one run: the runner prefixes the k-th suite's settings with b<k>_)
        path = tmp_path / "c4r" / "logs" / ("b%d_%s" % (suites.index(suite), res["label"])) / ("seed%d_logs.csv" % res["seed"])
        assert path.exists(), (suite, idx, res["label"])
        rows = list(csv.DictReader(open(path, newline="")))
        assert len(rows) == 5000 and len(res["rows"]) == (stop - 1) // 100, (len(rows), len(res["rows"]))
        for got, ref in zip(rows, res["rows"]):
            for k in ref:
                assert float(got[k]) == pytest.approx(ref[k], rel=1e-6, abs=1e-5), (suite, idx, k, got["steps"])
    assert sum(r["beta"] for r in alone) >= 2


def _n_devices():
    """HIP devices visible, asked of a child process: this process must not touch the GPU before it spawns ranks."""
    code = "import sys; sys.path.insert(0, %r); from colosseum_amd import _lib as L; print(L.load().cmdp_device_count())" % ROOT
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=ENV, capture_output=True, text=True, timeout=300)
    return int(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 and p.stdout.strip() else 0


def test_two_ranks_over_rccl_when_the_box_has_two_gpus(need_gpu, tmp_path):
    """The "nccl" (= RCCL over xGMI) branches of bench.py, tools/run_benchmark.py and tools/run_c5.py -- one rank per GPU, the
    driver's launcher -- on the first box that HAS two GPUs; on the one-GPU test boxes this skips itself (the same entry points
    run there with gloo and a shared device, above).  Ranks are spawned before anything in this process touches HIP."""
    if _n_devices() < 2:
        pytest.skip("one HIP device visible: the RCCL branches need two (gloo + shared device covers the rest above)")
    small = [a for a in SMALL if a not in ("--share-gpu", "--dist-backend", "gloo")]
    line = json.loads(_run(_torchrun(2, 29741) + ["bench.py", "--gpus", "2"] + small).strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "gather_ms" in line and line["roofline"]["frac"] <= 1.0
    args = ["tools/run_benchmark.py", "--configs-json", os.path.join(GOLDEN, "G11_benchmark_configs.json"), "--benchmark",
            "benchmark_episodic_quick_test", "--steps", "1500", "--seeds", "3", "--log-every", "500"]
    s1 = json.loads(_run([sys.executable] + args + ["--out", str(tmp_path / "one")]).strip().splitlines()[-1])
    s2 = json.loads(_run(_torchrun(2, 29742) + args + ["--out", str(tmp_path / "two")]).strip().splitlines()[-1])
    assert s1["mean_normalized_cumulative_regret"] == s2["mean_normalized_cumulative_regret"]
    c5 = ["tools/run_c5.py", "--room-size", "6", "--n-rooms", "4"]
    one = json.loads(_run([sys.executable] + c5).strip().splitlines()[-1])
    two = json.loads(_run(_torchrun(2, 29743) + c5).strip().splitlines()[-1])
    assert two["world"] == 2 and one["diameter"] == two["diameter"]


def test_c5_at_its_real_size_diameter_and_mixing_time(need_gpu):
    """Config C5 as BASELINE.json states it: MiniGridRoomsContinuous(room_size=28, n_rooms=16), S = 50 272 -- the diameter
    over ALL targets (K5C; three targets re-solved by the CPU oracle, bit-equal) and the build-defined mixing time of the
    uniform policy's chain (dense float64 matrix powers in HBM, 20 GB each), both pinned."""
    out = json.loads(_run([sys.executable, "tools/run_c5.py", "--mixing", "--check", "3"], timeout=900).strip().splitlines()[-1])
    assert out["n_states"] == 50272 and out["targets_this_rank"] == 50272
    assert out["diameter"] == 259.99786376953125 and out["oracle_check"] is True
    mx = out["mixing_time"]
    assert mx["t_mix"] == 49391 and 0.0 < mx["tv_at_t_mix"] <= 0.25
    print("C5 at full size: diameter %.2f s, mixing time %.1f s" % (out["solve_s"], mx["seconds"]))
