#!/usr/bin/env python3
"""bench.py -- the headline measurement (BASELINE.json: env steps/s on 65 536 parallel DeepSea(size=30)
instances, random policy + value-iteration sweeps/s), one rank per GPU.

    python bench.py --gpus 1 --steps 30 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one fused rollout launch: `--launch-steps` (default 5000) transitions of every instance of the
rank's shard (default 65 536 instances per GPU -> weak scaling), inputs resident in HBM.  K timed steps are
bracketed by barrier + device synchronise on both sides, the max over ranks is taken, rank 0 prints ONE JSON
line.  `value` = transitions of all ranks / that time.

Extra objects on the line (prompt section 4):
  roofline      dominant kernel (k_rollout): algorithmic bytes per launch / HIP-event launch time vs 8 TB/s
  cpu_baseline  the CPU oracle (oracle/cmdp_oracle.c, "port") timed on this host, one core, bounded sample
  vi            config C3: FrozenLake 20x20 discounted value iteration to 1e-6, sweeps/s (secondary metric)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


class HipEvents:
    """Minimal ctypes view of the HIP event API (libamdhip64 is already mapped by libcmdp.so)."""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [C.c_void_p]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.hip.hipEventDestroy.argtypes = [C.c_void_p]

    def create(self):
        ev = C.c_void_p()
        assert self.hip.hipEventCreate(C.byref(ev)) == 0
        return ev

    def record(self, ev, stream):
        assert self.hip.hipEventRecord(ev, C.c_void_p(stream)) == 0

    def elapsed_ms(self, a, b):
        assert self.hip.hipEventSynchronize(b) == 0
        ms = C.c_float()
        assert self.hip.hipEventElapsedTime(C.byref(ms), a, b) == 0
        return float(ms.value)


def frozenlake_dp_tables(seeds, size, workers):
    from colosseum_amd.mdp.fast_batch import frozenlake_dp_tables as build

    return build(seeds, size, workers, context="fork")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--instances", type=int, default=65536, help="instances per GPU (weak scaling)")
    ap.add_argument("--size", type=int, default=30)
    ap.add_argument("--launch-steps", type=int, default=5000, help="transitions per instance per launch (one bench step)")
    ap.add_argument("--vi-instances", type=int, default=4096, help="FrozenLake instances per GPU for the VI leg (0: skip)")
    ap.add_argument("--cpu-instances", type=int, default=4096, help="instances of the CPU-oracle sample (0: skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the default) or gloo (CPU rehearsal of the N>1 path)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: ranks share the visible GPUs (device = local_rank %% n_devices)")
    ap.add_argument("--lds-groups", type=int, default=0, help="LDS rollout workgroups per CU (0: library default)")
    ap.add_argument("--rollout-kernel", type=int, default=0, help="0 auto, 1 HBM tables, 2 LDS-resident")
    ap.add_argument("--layout", default="csr", choices=["csr", "dense"],
                    help="csr (default, fastest) or dense = per-instance float32 P[s,a,:] rows in HBM (north-star layout)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    # Host-side model construction of the VI leg first: it uses a fork()ed process pool, which must not happen
    # after the HIP runtime / RCCL have been initialised in this process.
    fl, fl_build_s = None, 0.0
    if args.vi_instances > 0:
        workers = max(1, min(16, (os.cpu_count() or 1) // max(1, world)))
        tb = time.time()
        fl = frozenlake_dp_tables(np.arange(rank * args.vi_instances, (rank + 1) * args.vi_instances), 20, workers)
        fl_build_s = time.time() - tb

    dist = None
    coll_device = "cpu"
    device_index = local_rank
    if world > 1:
        import torch
        import torch.distributed as dist

        if args.share_gpu:
            device_index = local_rank % max(1, torch.cuda.device_count())
        if args.dist_backend == "nccl":
            torch.cuda.set_device(device_index)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))  # RCCL over xGMI
            coll_device = "cuda"
        else:
            dist.init_process_group(args.dist_backend)

    from colosseum_amd import _lib as L
    from colosseum_amd.batched import BatchedMDP
    from colosseum_amd.mdp.fast_batch import deepsea_episodic_tables

    lib = L.load()
    assert lib.cmdp_device_count() > 0, "no HIP device visible: the product path has no CPU fallback"
    L.check(lib.cmdp_set_device(device_index))

    def barrier():
        if dist is not None:
            dist.barrier()

    # ---- C2 workload: this rank's shard of DeepSeaEpisodic(seed=i, size=30), i in [rank*B, (rank+1)*B) ----
    B = args.instances
    seeds = np.arange(rank * B, (rank + 1) * B, dtype=np.int64)
    t_build = time.time()
    dense = args.layout == "dense"
    tables = deepsea_episodic_tables(seeds, args.size, with_dp=dense)
    keys = seeds.astype(np.uint64)  # Philox key = global instance id
    env = BatchedMDP(tables=tables, rng_mode=L.RNG_PHILOX, philox_keys=keys,
                     layout=L.LAYOUT_DENSE if dense else L.LAYOUT_CSR)
    env.reset()
    if args.rollout_kernel:
        env.set_rollout_kernel(args.rollout_kernel)
    if args.lds_groups:
        env.set_option(L.OPT_LDS_GROUPS_PER_CU, args.lds_groups)
    t_build = time.time() - t_build
    S = int(env.n_states[0])
    plan = env.lds_plan() if not dense else dict(kernel="", eligible=False)
    lds_kernel = plan["kernel"]

    ev = HipEvents()
    stream = env.stream
    for _ in range(args.warmup):
        env.rollout_async(args.launch_steps)
    env.synchronize()
    marks = [ev.create() for _ in range(args.steps + 1)]
    barrier()
    env.synchronize()
    t0 = time.perf_counter()
    ev.record(marks[0], stream)
    for k in range(args.steps):
        env.rollout_async(args.launch_steps)
        ev.record(marks[k + 1], stream)
    env.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    launch_ms = [ev.elapsed_ms(marks[k], marks[k + 1]) for k in range(args.steps)]

    if dist is not None:
        import torch

        tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- final gather over RCCL (the only collective): per-instance episode counts = visits of the start state ----
    vs, _ = env.visits(sa=False)
    start = int(tables["start_state"][0])
    episodes = vs.reshape(B, S)[:, start].copy()
    gather_ms = None
    if dist is not None:
        import torch

        local = torch.from_numpy(episodes).to(coll_device)
        allv = torch.empty(world * B, dtype=torch.int64, device=coll_device)
        if coll_device == "cuda":
            torch.cuda.synchronize()
        g0 = time.perf_counter()
        dist.all_gather_into_tensor(allv, local)
        if coll_device == "cuda":
            torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        assert bool((allv[rank * B:(rank + 1) * B] == local).all())

    total_steps = world * B * args.launch_steps * args.steps
    value = total_steps / elapsed
    avg_launch_s = float(np.mean(launch_ms)) * 1e-3
    # SURVEY 8(d), CSR companion figure: 8 (row pointer pair) + 8*nnz(s,a) + 28 B per transition = 44 B at C2
    bytes_per_step = (4 * S + 28) if dense else (8 + 8 * 1 + 28)  # dense-row figure / CSR figure
    algo_bytes = bytes_per_step * B * args.launch_steps
    achieved = algo_bytes / avg_launch_s / 1e9
    # HBM bytes per launch from the PMC counters: they need their own rocprofv3 passes (--pmc FETCH_SIZE,
    # --pmc WRITE_SIZE), so the figure is read from the committed summary of those passes when the workload
    # is the one they were collected on; null otherwise.
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_rollout_pmc.json")
    if os.path.exists(pmc):
        try:
            j = json.load(open(pmc))
            if j.get("transitions_per_launch") == B * args.launch_steps and args.size == 30 and not dense:
                traffic = j.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    line = {
        "metric": "env steps/sec (whole node) + value-iteration sweeps/sec, DeepSea size=%d x%d" % (args.size, B),
        "value": value,
        "unit": "env steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "i32 state/visit indices, f64 rewards+CDF",
        "data": "synthetic",
        "config": {
            "workload": "C2: DeepSeaEpisodic(seed=i,size=%d), %d instances per GPU, on-device uniform random policy "
                        "(Philox-4x32-10), %d transitions per instance per step, auto-reset at h>=H" % (args.size, B, args.launch_steps),
            "instances_per_gpu": B, "states": S, "actions": 2, "horizon": int(env.H),
            "transitions_per_instance_per_step": args.launch_steps, "layout": args.layout, "rng": "philox4x32-10",
            "build_s": round(t_build, 2),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "k_rollout_dense<0,NV>" if dense else (lds_kernel if args.rollout_kernel != 1 else "k_rollout<0,false>"), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "lds_plan": plan,
            "algorithmic_bytes_per_transition": bytes_per_step, "accounting": "SURVEY 8(d) dense-row figure (4*S+28 B/transition)" if dense else "SURVEY 8(d) CSR figure (44 B/transition)",
            "launch_ms_avg": avg_launch_s * 1e3, "launch_ms_min": float(np.min(launch_ms)),
            # what actually crosses the HBM interface (PMC), as a fraction of peak: the LDS-resident kernels keep the
            # tables on chip, so `frac` (algorithmic bytes) can exceed 1 while this stays small
            "traffic_frac": (traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
        },
    }
    if gather_ms is not None:
        line["gather_ms"] = gather_ms

    # ---- VI leg (config C3): FrozenLake 20x20, gamma .99, eps 1e-6, the reference's own scheme rule --------
    if args.vi_instances > 0:
        tb = time.time()
        dp = BatchedMDP(tables=fl, with_env=False)
        tb = time.time() - tb + fl_build_s
        bufs = dp.dp_buffers()          # page-locked result buffers, reused by both calls
        dp.value_iteration(0.99, 1e-6, out=bufs)  # warm-up (also sizes the device buffers)
        barrier()
        t1 = time.perf_counter()
        Q, V, sw = dp.value_iteration(0.99, 1e-6, out=bufs)
        barrier()
        vi_s = time.perf_counter() - t1
        sweeps = float(sw.sum())
        if dist is not None:
            import torch

            tt = torch.tensor([vi_s], dtype=torch.float64, device=coll_device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            vi_s = float(tt[0].item())
            ts = torch.tensor([sweeps], dtype=torch.float64, device=coll_device)
            dist.all_reduce(ts)
            sweeps = float(ts.item())
        line["vi"] = {
            "workload": "C3: FrozenLakeContinuous(seed=i,size=20,p_frozen=0.9,is_slippery=True,p_rand=0.1), %d instances "
                        "per GPU, discounted VI gamma=0.99 eps=1e-6, scheme by the reference rule (Jacobi)" % args.vi_instances,
            "sweeps_per_s": sweeps / vi_s, "instances_per_s": world * args.vi_instances / vi_s, "total_sweeps": sweeps,
            "wall_ms": vi_s * 1e3, "includes": "H2D of nothing, D2H of Q,V,sweeps into reused page-locked host buffers", "build_s": round(tb, 2),
        }
        dp.close()

    # ---- CPU baseline: the oracle, one core, bounded sample of the same workload (rank 0, N = 1 only) -----
    if rank == 0 and world == 1 and not args.no_cpu and args.cpu_instances > 0:
        from oracle import oracle as O

        n = min(args.cpu_instances, B)
        n_steps = args.launch_steps * (args.steps + args.warmup)  # everything the GPU did since reset()
        c0 = time.perf_counter()
        last, rsum, cvs, _ = O.batch_rollout(tables, 0, n, n_steps, rng_mode=1, philox_keys=keys, want_visits=True)
        cpu_s = time.perf_counter() - c0
        # the timed GPU work is checked, not just timed: state-visit counts of the sampled instances are bit-equal
        verified = bool(np.array_equal(cvs, vs[: n * S]))
        assert verified, "GPU visit counts differ from the CPU oracle"
        line["verified_against_oracle"] = "state-visit counts of instances 0..%d after %d transitions: bit-equal" % (n - 1, n_steps)
        line["cpu_baseline"] = {
            "value": n * n_steps / cpu_s, "unit": "env steps/s", "cores": 1, "kind": "port",
            "sample": "instances 0..%d of the same batch, %d transitions each (same Philox streams), reset included; "
                      "%.1f s on 1 of %d host cores" % (n - 1, n_steps, cpu_s, os.cpu_count() or 0),
        }
        # the reference's multiprocessing model (config.py:22-26: os.cpu_count() - 2 workers): the same oracle on all but
        # two host cores, every thread its own contiguous range of the batch (the C call releases the GIL)
        cores = max(1, (os.cpu_count() or 1) - 2)
        try:  # a container's CPU quota (cgroup v2 cpu.max = "<quota> <period>") is what is really available
            q, per_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                cores = max(1, min(cores, int(int(q) / int(per_))))
        except Exception:
            pass
        try:
            cores = max(1, min(cores, len(os.sched_getaffinity(0))))
        except Exception:
            pass
        if cores > 1:
            from concurrent.futures import ThreadPoolExecutor

            n_mt = min(B, 1024 * cores)
            per = -(-n_mt // cores)
            ranges = [(lo, min(n_mt, lo + per)) for lo in range(0, n_mt, per)]
            c0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=cores) as pool:
                list(pool.map(lambda r: O.batch_rollout(tables, r[0], r[1], n_steps, rng_mode=1, philox_keys=keys), ranges))
            mt_s = time.perf_counter() - c0
            model = ""
            try:
                model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
            except Exception:
                pass
            line["cpu_baseline_all_cores"] = {
                "value": n_mt * n_steps / mt_s, "unit": "env steps/s", "cores": len(ranges), "kind": "port",
                "sample": "instances 0..%d, %d transitions each, %.1f s on %d threads (CPU quota of this box; %d host cores, %s)"
                          % (n_mt - 1, n_steps, mt_s, len(ranges), os.cpu_count() or 0, model),
            }
        if "vi" in line and args.vi_instances > 0:
            nv = min(256, args.vi_instances)
            c0 = time.perf_counter()
            _, _, swc = O.batch_vi(fl, 0, nv, 0.99, 1e-6, 0)
            cv = time.perf_counter() - c0
            line["vi"]["cpu_baseline"] = {"value": float(swc.sum()) / cv, "unit": "sweeps/s", "cores": 1, "kind": "port",
                                          "sample": "instances 0..%d, %.1f s" % (nv - 1, cv)}
    env.close()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
