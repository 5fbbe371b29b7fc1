"""N > 1 path on CPU: two gloo ranks each take their contiguous shard of the instances, roll it out (the CPU
oracle stands in for the device here -- this tests the sharding / gather logic, not the kernels) and gather;
the result must equal the single-process run instance for instance."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_instances, n_steps, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from colosseum_amd.mdp.fast_batch import deepsea_episodic_tables
    from colosseum_amd.sharding import gather_instances, shard_range
    from oracle import oracle as O

    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_instances, rank, world)
    seeds = np.arange(lo, hi)
    tables = deepsea_episodic_tables(seeds, 6)
    last, rsum = O.batch_rollout(tables, 0, hi - lo, n_steps, rng_mode=1, philox_keys=seeds.astype(np.uint64))
    local = np.stack([last.astype(np.float64), rsum], 1)
    full = gather_instances(local, n_instances, dist)
    assert full.shape == (n_instances, 2)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), full)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from colosseum_amd.sharding import shard_range

    for n in (1, 7, 8, 65536, 65537):
        for w in (1, 2, 3, 8):
            blocks = [shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(600)
def test_two_rank_gloo_matches_single_process(tmp_path):
    import torch.multiprocessing as mp

    from colosseum_amd.mdp.fast_batch import deepsea_episodic_tables
    from oracle import oracle as O

    n_instances, n_steps, world = 11, 500, 2  # odd count: ragged shards
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_instances, n_steps, str(tmp_path)), nprocs=world, join=True)
    seeds = np.arange(n_instances)
    tables = deepsea_episodic_tables(seeds, 6)
    last, rsum = O.batch_rollout(tables, 0, n_instances, n_steps, rng_mode=1, philox_keys=seeds.astype(np.uint64))
    ref = np.stack([last.astype(np.float64), rsum], 1)
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"rank{r}.npy"), ref)
