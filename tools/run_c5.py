"""Config C5: diameter of one large MiniGridRoomsContinuous MDP (S ~ 50 000) with the 64-targets-per-workgroup kernel.

    python tools/run_c5.py [--targets N] [--room-size 28 --n-rooms 16] [--check K]
    python -m torch.distributed.run --nproc-per-node N tools/run_c5.py      # targets sharded over GPUs, max-reduced

--targets N solves only the first N targets of this rank's shard (timing runs); --check K re-solves K targets with the
CPU oracle (single-target Jacobi value iteration on T_es) and requires bit-equal hitting times."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--room-size", type=int, default=28)
    ap.add_argument("--n-rooms", type=int, default=16)
    ap.add_argument("--targets", type=int, default=0)
    ap.add_argument("--check", type=int, default=0)
    ap.add_argument("--eps", type=float, default=1e-3)
    ap.add_argument("--workspace-mb", type=int, default=0)
    ap.add_argument("--dist-backend", default="nccl")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses device 0 (gloo backend)")
    ap.add_argument("--mixing", action="store_true",
                    help="also the build-defined mixing time of the uniform policy's chain (rank 0; matrix powers in HBM)")
    ap.add_argument("--no-diameter", action="store_true")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    from colosseum_amd import _lib as L
    from colosseum_amd.batched import BatchedMDP
    from colosseum_amd.mdp import make_model
    from colosseum_amd.sharding import shard_range

    t0 = time.time()
    m = make_model("MiniGridRoomsContinuous", seed=0, room_size=args.room_size, n_rooms=args.n_rooms,
                   n_starting_states=2, p_lazy=0.1)
    S, A = m.n_states, m.n_actions
    t_build = time.time() - t0
    if args.share_gpu:
        local = 0
    if world > 1:
        import torch
        import torch.distributed as dist

        if args.dist_backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.dist_backend)
    L.check(L.load().cmdp_set_device(local))
    dp = BatchedMDP([m], with_env=False)
    if args.workspace_mb:
        dp.set_option(L.OPT_DIAMETER_WORKSPACE_MB, args.workspace_mb)
    lo, hi = shard_range(S, rank, world)
    if args.targets:
        hi = min(hi, lo + args.targets)
    t0 = time.time()
    per = dp.diameter_range(lo, hi, args.eps) if not args.no_diameter else np.zeros(0, np.float32)
    t_solve = time.time() - t0
    mixing = None
    if args.mixing and rank == 0:
        from colosseum_amd.hardness import mixing_time

        dp.close()  # the S x S float64 matrices want the HBM
        tm = time.time()
        tmix, tv = mixing_time([m], threshold=0.25, max_steps=1 << 22)
        mixing = dict(t_mix=int(tmix[0]), tv_at_t_mix=float(tv[0]), threshold=0.25, policy="uniform", seconds=round(time.time() - tm, 2))
        dp = BatchedMDP([m], with_env=False)
    local_max = float(per.max()) if len(per) else 0.0
    diameter = local_max
    if world > 1:
        x = torch.tensor([local_max], device=f"cuda:{local}" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
        diameter = float(x.item())
    ok = None
    if args.check:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O

        ptr, col, val = m.csr()
        ok = True
        sweeps = []
        t_or = time.time()
        for es in np.linspace(lo, hi - 1, args.check).astype(int):
            p2, c2, v2 = [0], [], []
            for s in range(S):
                for a in range(A):
                    r = s * A + a
                    if s == es:
                        c2.append(np.array([es], np.int32)); v2.append(np.array([1.0], np.float32))
                    else:
                        c2.append(col[ptr[r]:ptr[r + 1]]); v2.append(val[ptr[r]:ptr[r + 1]])
                    p2.append(p2[-1] + len(c2[-1]))
            R2 = -np.ones((S, A), np.float32)
            R2[es] = 0
            _, V, it, _ = O.vi_discounted(S, A, (np.array(p2, np.int64), np.concatenate(c2), np.concatenate(v2)), R2,
                                      gamma=1.0, eps=args.eps, scheme=1)
            ok &= bool(-V.min() == per[es - lo])
            sweeps.append(it)
    if rank == 0:
        print(json.dumps(dict(config="C5", mdp="MiniGridRoomsContinuous", room_size=args.room_size, n_rooms=args.n_rooms,
                              n_states=S, n_actions=A, nnz=int(len(m.csr()[1])), world=world, targets_this_rank=int(hi - lo),
                              build_s=round(t_build, 2), solve_s=round(t_solve, 3),
                              targets_per_s=round((hi - lo) / max(t_solve, 1e-9), 1), diameter=diameter, mixing_time=mixing, oracle_check=ok,
                              oracle_sweeps=(sweeps if args.check else None),
                              oracle_s_per_target=(round((time.time() - t_or) / args.check, 2) if args.check else None))))
    dp.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
