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

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colosseum_amd import benchmark as bm  # noqa: E402
from colosseum_amd.sharding import gather_instances, shard_range  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--folder", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int)
    ap.add_argument("--seeds", type=int)
    ap.add_argument("--log-every", type=int)
    args = ap.parse_args()
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    cfg = yaml.safe_load(open(os.path.join(args.folder, "experiment_config.yml")))
    n_steps = args.steps or cfg["n_steps"]
    n_seeds = args.seeds or cfg["n_seeds"]
    log_every = args.log_every or cfg["log_performance_indicators_every"]
    instances = bm.enumerate_instances(bm.load_mdp_configs(args.folder), n_seeds)
    t0 = time.time()
    results = bm.run_instances(instances, n_steps, log_every, rank, world, device=local)
    bm.write_csv_logs(args.out, instances, results)
    lo, hi = shard_range(len(instances), rank, world)
    local_vec = np.stack([bm.summary_vector(results[i]) for i in range(lo, hi)]) if hi > lo else np.zeros((0, 3))
    allv = gather_instances(local_vec, len(instances), dist, device="cuda" if dist is not None else None)
    if rank == 0:
        print(json.dumps(dict(instances=len(instances), steps_each=n_steps, wall_s=time.time() - t0,
                              agent_steps_per_s=len(instances) * n_steps / (time.time() - t0),
                              mean_normalized_cumulative_regret=float(allv[:, 1].mean()))))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
