#!/usr/bin/env python3
"""Runs a Colosseum-style benchmark folder (mdp_configs/*.gin, experiment_config.yml) for the four supported MDP
families with tabular Q-learning agents on the GPU(s) and writes the reference's CSV log files.

    python tools/run_benchmark.py --folder <benchmark folder> --out results [--steps N --seeds K --log-every L]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/run_benchmark.py ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import yaml

# Device batches are driven concurrently, one stream each; the HIP runtime multiplexes streams onto 4 hardware queues by
# default, which serialises the (small, latency-bound) kernels of different batches.  16 queues: 89 s -> 55 s for the
# four default suites on one MI355X.  Read by the runtime at its first call, so it is set before anything touches HIP.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")   # two streams per continuous batch, 16 batches in flight

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# many host threads, one device batch each: they sleep while they wait for the device instead of spinning on a core apiece
# (the GPU boxes give a job a CPU quota; spinning threads starve the reward draws and the log writers of it)
os.environ.setdefault("CMDP_SYNC_MODE", "block")
from colosseum_amd import benchmark as bm  # noqa: E402
from colosseum_amd.sharding import gather_instances, shard_range  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--folder", help="benchmark folder laid out like the reference's (mdp_configs/*.gin, experiment_config.yml)")
    ap.add_argument("--configs-json", help="alternative: JSON of benchmark definitions (tests/golden/G11_benchmark_configs.json)")
    ap.add_argument("--benchmark", action="append", help="with --configs-json: benchmark name(s), e.g. benchmark_episodic_ergodic")
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int)
    ap.add_argument("--seeds", type=int)
    ap.add_argument("--log-every", type=int)
    ap.add_argument("--concurrent-groups", type=int, default=16, help="device batches driven concurrently (host threads, one stream each)")
    ap.add_argument("--max-batch", type=int, default=128, help="instances per device batch (larger groups are split)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (rehearsals)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a one-GPU box: every rank uses device 0")
    ap.add_argument("--max-time", type=float, help="seconds of training per instance (default: the experiment's max_interaction_time_s)")
    ap.add_argument("--beta-rewards", default="reference", choices=["reference", "philox", "philox-gammas"],
                    help="stochastic (Beta) rewards: 'reference' = the reference's per-triple caches of 5000 samples from the MDP's own "
                         "numpy stream (rows equal the reference's); 'philox' = sampled on the device (distribution-exact, no host work); "
                         "'philox-gammas' = the same with the two-gamma sampler of rounds 1-2 for every shape (reproduces their numbers)")
    ap.add_argument("--overwrite", action="store_true",
                    help="run every instance; default: skip those whose log file exists, as the reference's resume does")
    args = ap.parse_args()
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    dist = None
    if args.share_gpu:
        local = 0
    if args.folder:
        cfg = yaml.safe_load(open(os.path.join(args.folder, "experiment_config.yml")))
        benchmarks = [bm.load_mdp_configs(args.folder)]
    else:
        allcfg = json.load(open(args.configs_json))
        cfg = allcfg[args.benchmark[0]]["experiment_config"]
        benchmarks = [allcfg[b]["mdp_configs"] for b in args.benchmark]
    n_steps = args.steps or cfg["n_steps"]
    n_seeds = args.seeds or cfg["n_seeds"]
    log_every = args.log_every or cfg["log_performance_indicators_every"]
    max_time = args.max_time if args.max_time is not None else float(cfg.get("max_interaction_time_s", float("inf")))
    instances = []
    for k, mdp_cfg in enumerate(benchmarks):
        for ins in bm.enumerate_instances(mdp_cfg, n_seeds):
            ins.mdp_scope = f"b{k}_{ins.mdp_scope}" if len(benchmarks) > 1 else ins.mdp_scope
            instances.append(ins)
    todo = set(range(len(instances)) if args.overwrite else bm.unfinished_instances(args.out, instances))
    skip = [i for i in range(len(instances)) if i not in todo]
    t0 = time.time()
    # order matters: (1) host model construction in a fork()ed pool, (2) the process group -- right away and with a
    # generous timeout, so that no rank waits in rendezvous while another is still running its shard, and a rank that
    # dies mid-run is noticed at the gather -- (3) only then the first HIP call of this process
    models = bm.build_shard_models(instances, rank, world, workers=max(1, min(16, (os.cpu_count() or 1) // world)), skip=skip)
    print(f"[rank {rank}] {len(models)} models built in {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
    if world > 1:
        import datetime

        import torch
        import torch.distributed as dist

        timeout = datetime.timedelta(hours=6)
        if args.dist_backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=timeout)
        else:
            dist.init_process_group(args.dist_backend, timeout=timeout)
    # log files are written as each device batch finishes (spawned writer processes: the text conversion must not hold
    # the GIL of the threads driving the other batches), each file atomically: an interrupted run resumes from them
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor

    writers = ProcessPoolExecutor(max_workers=max(1, min(16, (os.cpu_count() or 1) // world // 2)), mp_context=mp.get_context("spawn"))
    pending = []
    results = bm.run_instances(instances, n_steps, log_every, rank, world, device=local, max_concurrent_groups=args.concurrent_groups, max_batch=args.max_batch,
                               models=models, max_time=max_time, skip=skip, beta_rewards=args.beta_rewards,
                               on_group_done=lambda idx, rows: pending.extend(bm.submit_group_logs(writers, args.out, instances, idx, rows)),
                               progress=lambda msg: print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True))
    t_run = time.time() - t0
    n_written = sum(f.result() for f in pending)
    writers.shutdown()
    assert n_written == len(results), (n_written, len(results))
    print(f"[rank {rank}] instances done in {t_run:.1f} s, last log files written {time.time() - t0 - t_run:.1f} s later", file=sys.stderr, flush=True)
    lo, hi = shard_range(len(instances), rank, world)
    local_vec = (np.stack([bm.summary_vector(results[i]) if i in results else bm.read_summary(args.out, instances[i])
                           for i in range(lo, hi)]) if hi > lo else np.zeros((0, 3)))
    allv = gather_instances(local_vec, len(instances), dist,
                            device="cuda" if dist is not None and args.dist_backend == "nccl" else None)
    if rank == 0:
        n_run = len(instances) - len(skip)  # instances found on disk were not run: they do not count as throughput
        print(json.dumps(dict(instances=len(instances), skipped_existing=len(skip), run=n_run, steps_each=n_steps, wall_s=time.time() - t0,
                              beta_rewards=args.beta_rewards, agent_steps_per_s=n_run * n_steps / (time.time() - t0),
                              mean_normalized_cumulative_regret=float(allv[:, 1].mean()))))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
