#!/usr/bin/env python3
"""bench.py -- the headline measurement (BASELINE.json: env steps/s on 65 536 parallel DeepSea(size=30)
instances, random policy + value-iteration sweeps/s), one rank per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...          (no WORLD_SIZE in the environment: spawns N child ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the batch = SURVEY 8(d)'s C2 job: ONE fused rollout launch of
`--launch-steps` (default 30 000 = 1 000 episodes) transitions of every instance of the rank's shard (65 536
instances per GPU -> weak scaling), inputs resident in HBM.  K timed steps are bracketed by barrier + device
synchronise on both sides, the max over ranks is taken, rank 0 prints ONE JSON line.  `value` = transitions of all
ranks / that time.

Objects on the line (prompt section 4; DESIGN.md section 4 explains every figure):
  roofline      the dominant kernel (K1T `k_rollout_tmpl` at the default configuration: one successor table per
                workgroup, per-instance swap bits and count deltas).  The tables are in LDS, so it is bound by the
                latency of the dependent LDS reads of a transition, not by HBM: bound = "lds_latency", achieved = transitions/s,
                peak = chains the LDS can hold / the dependent-read chain latency CALIBRATED IN THIS RUN
                (cmdp_calibrate), frac <= 1.  The HBM view (algorithmic bytes and PMC traffic against 8 TB/s) is the
                sub-object roofline.hbm.
  dense         the north star's layout (per-instance dense float32 P[s,a,:] rows streamed from HBM, K1D): its own
                roofline, bound = "hbm".
  vi            config C3 (FrozenLake 20x20 discounted value iteration to 1e-6): sweeps/s + its own roofline
                (bound = "valu_issue": the CSR stays in registers, V in LDS).
  cpu_baseline  the CPU oracle (oracle/cmdp_oracle.c, "port") timed on this host, one core, bounded sample;
                cpu_baseline_numpy: the numpy restatement (oracle/numpy_port.py); reference_in_container: the reference's
                own per-step rate measured at survey time -- quoted, not re-measured
  sustained     >= 10 s of back-to-back launches of the headline kernel after the timed region: achieved rate, the
                engine clock and GPU-busy figure the SMI shows meanwhile (a witness for the burst number above)
  strong        SURVEY 8(e)'s partition of 65 536 instances TOTAL over 8 GPUs, rehearsed on this GPU with one rank's
                share (8 192 instances): what strong scaling of a latency-bound chain can give
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
N_CUS = 256
CLOCK_HZ = 2.4e9        # MI355X peak engine clock
VALU_CYCLES = 2         # MI355X_MICROARCH.md, per-instruction constants: wave64 v_fma_f32 = 2 cycles per SIMD
LDS_BYTES = 160 * 1024  # per CU
PMC_FILE = os.path.join(ROOT, "profiles", "r04_pmc.json")


class HipEvents:
    """Minimal ctypes view of the HIP event API (libamdhip64 is already mapped by libcmdp.so)."""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        self.hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [C.c_void_p]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.hip.hipEventDestroy.argtypes = [C.c_void_p]

    def create(self):
        # hipEventDisableSystemFence: a timestamp on the stream, without the system-scope cache flush a default event performs
        # when it completes (nothing is handed to the host between the steps of a timed region)
        ev = C.c_void_p()
        assert self.hip.hipEventCreateWithFlags(C.byref(ev), 0x20000000) == 0
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


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: N child ranks of this very script, started BEFORE this process
    makes any HIP / RCCL call; rank 0's stdout (the JSON line) is relayed, the exit code is the worst child's."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def smi_sample():
    """(engine clock in MHz, busy percent) of the busiest card the SMI's sysfs files show, or (None, None)."""
    import glob

    best = (None, None)
    for dev in glob.glob("/sys/class/drm/card*/device"):
        try:
            busy = int(open(dev + "/gpu_busy_percent").read().strip())
            mhz = None
            for ln in open(dev + "/pp_dpm_sclk").read().splitlines():
                if ln.strip().endswith("*"):
                    mhz = int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))
            if best[1] is None or busy > best[1]:
                best = (mhz, busy)
        except Exception:
            continue
    return best


def sustained_leg(env, ev, launch_steps, seconds, B):
    """Back-to-back launches of the headline kernel for >= `seconds`, in queue-sized bursts; the SMI's engine clock and
    busy figure are sampled from sysfs by a thread meanwhile."""
    import threading

    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(smi_sample())
            stop.wait(0.25)

    th = threading.Thread(target=sampler, daemon=True)
    stream = env.stream
    a, b = ev.create(), ev.create()
    env.synchronize()
    th.start()
    t0 = time.perf_counter()
    ev.record(a, stream)
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(64):
            env.rollout_async(launch_steps)
        n += 64
        env.synchronize()
        if n % 1024 == 0:
            env.reset_visits()   # (the device visit counters are int32: the library refuses launches that could wrap them)
    ev.record(b, stream)
    ms = ev.elapsed_ms(a, b)
    wall = time.perf_counter() - t0
    stop.set()
    th.join()
    clocks = [c for c, _ in samples if c]
    busy = [x for _, x in samples if x is not None]
    return {
        "seconds": wall, "launches": n, "value": n * B * launch_steps / (ms * 1e-3), "unit": "env steps/s",
        "ms_per_launch": ms / n,
        "sclk_mhz": {"min": min(clocks), "median": float(np.median(clocks)), "max": max(clocks), "samples": len(clocks)} if clocks else None,
        "gpu_busy_percent": {"min": min(busy), "median": float(np.median(busy)), "max": max(busy)} if busy else None,
        "smi_source": "/sys/class/drm/card*/device/{pp_dpm_sclk,gpu_busy_percent}, 4 Hz, busiest card",
    }


LIVE_PMC = None   # {kernel base name: {"read": bytes, "write": bytes, "launches": n, SQ counters...}} measured by child runs under rocprofv3
# one SQ pass (8 counters): instructions issued, LDS pipe cycles and bank-conflict cycles, kernel cycles (GRBM, summed over 8 XCDs)
SQ_COUNTERS = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT",
               "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE"]
# cycles a wave64 instruction of K1E's step (v_bfe_u32, v_lshl_or_b32, v_and_or_b32, v_alignbit_b32) occupies a SIMD at 4-8
# wavefronts per SIMD: measured, tools/calib/valu_int_rate.hip -> profiles/r04_valu_int_rate.txt (float32 v_fma: 2.7-3.0)
VALU_INT_VOP3_CYCLES = 4.4
# cycles a wave64 ds_add_rtn_u32 (bank-conflict-free) occupies a CU's LDS pipe, and cycles per wave-transition per SIMD of the
# walk's step alone -- v_add_co_u32 . v_cndmask_b32 . v_and_or_b32 . ds_add_rtn_u32 . v_alignbit_b32, four chains per lane, 16
# wavefronts per CU, nothing else in the loop: measured, tools/calib/lds_atomic_rate.hip -> profiles/r04_lds_atomic_rate.txt
# (a ds_read_b32 takes 1.1 cycles; SQ_LDS_IDX_ACTIVE counts ~2 per LDS instruction whatever it is, so it UNDERSTATES atomics)
LDS_ATOMIC_RTN_CYCLES = 4.31
VALU_FP32_CYCLES_2_WAVES = 4.22   # v_fma_f32 at two wavefronts per SIMD (profiles/r04_valu_int_rate.txt)
WALK_STEP_CYCLES_PER_SIMD = 20.9


def sq_view(kernel, launch_s, units=None, build_id=None):
    """Hardware-anchored fractions of one kernel: VALU issue (wave-instructions issued x cycles per instruction / SIMD-cycles
    of the launch) and LDS pipe (SQ_LDS_IDX_ACTIVE per CU-cycle).  Counters from the SQ child pass of THIS run; without it
    (N > 1, no rocprofv3, run under a profiler) from the committed summary of the same passes (profiles/r04_pmc.json,
    tools/collect_profiles.sh), scaled to this launch's units -- instructions per transition belong to the build, which the
    summary names.  The cycles are the launch's OWN duration in the timed region x the nominal clock."""
    v, src, current = (LIVE_PMC or {}).get(kernel), None, True
    if v and "SQ_INSTS_VALU" in v:
        src = "this run: rocprofv3 --pmc %s child pass of bench.py, per-launch means" % " ".join(SQ_COUNTERS)
    else:
        v = None
        try:
            j = json.load(open(PMC_FILE))
            for k in j.get("kernels", []):
                if k["kernel"].split("<")[0] == kernel and "SQ_INSTS_VALU" in k and k.get("units_per_launch") and units:
                    f = units / k["units_per_launch"]
                    v = {c: k[c] * f for c in SQ_COUNTERS if c in k}
                    current = j.get("build_id") == build_id
                    src = "profiles/%s (collected on build %s = %s build), scaled by units per launch x %.4g" % (
                        os.path.basename(PMC_FILE), j.get("build_id", "")[:16], "this" if current else "ANOTHER", f)
        except Exception:
            pass
    if not v or not launch_s:
        return None
    simd_cycles = N_CUS * 4 * launch_s * CLOCK_HZ
    out = {
        "source": src, "counters_measured_in_this_run": src.startswith("this run"), "counters_current": current,
        "valu_wave_insts_per_launch": v["SQ_INSTS_VALU"], "lds_wave_insts_per_launch": v.get("SQ_INSTS_LDS"),
        "salu_wave_insts_per_launch": v.get("SQ_INSTS_SALU"),
        "valu_G_wave_insts_per_s": v["SQ_INSTS_VALU"] / launch_s / 1e9,
        "valu_issue_frac_at_2_cycles": v["SQ_INSTS_VALU"] * VALU_CYCLES / simd_cycles,
        "valu_issue_frac_at_measured_int_rate": v["SQ_INSTS_VALU"] * VALU_INT_VOP3_CYCLES / simd_cycles,
        "lds_pipe_frac": v.get("SQ_LDS_IDX_ACTIVE", 0.0) / (N_CUS * launch_s * CLOCK_HZ),
        "lds_bank_conflict_frac_of_lds_cycles": v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0),
    }
    if "GRBM_GUI_ACTIVE" in v and src.startswith("this run"):
        out["kernel_ms_under_profiler"] = v["GRBM_GUI_ACTIVE"] / 8.0 / CLOCK_HZ * 1e3   # rocprofv3 sums the 8 XCDs
    return out


def live_pmc(child_args, timeout=240):
    """HBM traffic per launch MEASURED IN THIS RUN: two child runs of this script's headline and dense legs under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes: the two counters do not fit the TCC slots
    together; --kernel-trace is the only other option on the command line, the program comes directly after `--`), per-kernel
    means of the collected values.  FETCH_SIZE x 2 per the gfx950 correction for wide coalesced reads, both counters in
    KiB.  Returns None when rocprofv3 is missing or a pass fails (the committed summary is used then)."""
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    out = {}
    passes = [(["FETCH_SIZE"], child_args), (["WRITE_SIZE"], child_args), (SQ_COUNTERS, child_args + ["--pmc-child-vi"])]
    for counters, cargs in passes:
        d = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
        if counters is SQ_COUNTERS:
            cargs = cargs + ["--pmc-child-out", os.path.join(d, "child.json")]
        cmd = [exe, "--pmc"] + counters + ["--kernel-trace", "-d", d, "--output-format", "csv", "--", sys.executable,
               os.path.abspath(__file__), "--pmc-child"] + cargs
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=timeout,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
            if r.returncode != 0 or not files:
                err = r.stderr.decode(errors="replace")
                sys.stderr.write("bench.py: rocprofv3 --pmc %s child pass failed (rc %s): %s\n" % (
                    " ".join(counters), r.returncode, "\n".join(l for l in err.splitlines() if "rocprofv3" not in l and "output_stream" not in l)[-1500:]))
                if counters is SQ_COUNTERS:
                    continue      # the traffic passes stand without the SQ pass
                return None
            if counters is SQ_COUNTERS and os.path.exists(os.path.join(d, "child.json")):
                out["_child"] = json.load(open(os.path.join(d, "child.json")))
            acc = {}
            for row in csv.DictReader(open(files[0])):
                name = row["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]
                acc.setdefault(name, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for k, cs in acc.items():
                for c, v in cs.items():
                    key, scale = {"FETCH_SIZE": ("read", 2048.0), "WRITE_SIZE": ("write", 1024.0)}.get(c, (c, 1.0))
                    out.setdefault(k, {})[key] = scale * sum(v) / len(v)
                    if key in ("read", "write") or c == "SQ_INSTS_VALU":
                        out[k]["launches"] = len(v)
        except Exception as e:
            sys.stderr.write("bench.py: rocprofv3 --pmc %s child pass: %r\n" % (" ".join(counters), e))
            if counters is SQ_COUNTERS:
                continue
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {k: v for k, v in out.items() if ("read" in v and "write" in v) or "SQ_INSTS_VALU" in v or k == "_child"}


def pmc_entry(kernel_prefix, units_per_launch, build_id):
    """HBM bytes per launch of a kernel from the committed summary of this round's `--pmc FETCH_SIZE` / `--pmc
    WRITE_SIZE` passes (their own rocprofv3 runs, tools/collect_profiles.sh).  Returned only when the summary was
    collected on the same workload; `current` says whether it was collected on THIS build of the library."""
    base = kernel_prefix.split("<")[0]
    if LIVE_PMC and base in LIVE_PMC and "read" in LIVE_PMC[base] and "write" in LIVE_PMC[base]:   # measured in this run (child passes under rocprofv3)
        v = LIVE_PMC[base]
        return dict(bytes=v["read"] + v["write"], read=v["read"], write=v["write"], current=True,
                    source="this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of bench.py, %d launches" % v["launches"],
                    collected_on_build=build_id[:16])
    try:
        j = json.load(open(PMC_FILE))
    except Exception:
        return None
    for k in j.get("kernels", []):
        if k["kernel"].split("<")[0] == kernel_prefix.split("<")[0] and k.get("units_per_launch") == units_per_launch:
            return dict(bytes=k["hbm_bytes_per_launch"], read=k["hbm_read_bytes_per_launch"], write=k["hbm_write_bytes_per_launch"],
                        current=j.get("build_id") == build_id, source="profiles/" + os.path.basename(PMC_FILE),
                        collected_on_build=j.get("build_id", "")[:16])
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle-launches", type=int, default=80, help="untimed launches before the W warm-up steps (clock / first-touch settling; 0: none)")
    ap.add_argument("--instances", type=int, default=65536, help="instances per GPU (weak scaling)")
    ap.add_argument("--size", type=int, default=30)
    ap.add_argument("--launch-steps", type=int, default=30000,
                    help="transitions per instance per launch = one bench step (SURVEY 8d C2: 30 000 = 1 000 episodes)")
    ap.add_argument("--vi-instances", type=int, default=4096, help="FrozenLake instances per GPU for the VI leg (0: skip)")
    ap.add_argument("--dense-instances", type=int, default=65536, help="instances of the dense-row leg (0: skip)")
    ap.add_argument("--dense-launch-steps", type=int, default=200)
    ap.add_argument("--dense-steps", type=int, default=10)
    ap.add_argument("--cpu-instances", type=int, default=512, help="instances of the CPU-oracle sample (0: skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the default) or gloo (CPU rehearsal of the N>1 path)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: ranks share the visible GPUs (device = local_rank %% n_devices)")
    ap.add_argument("--lds-groups", type=int, default=0, help="K1L workgroups per CU (0: library default; needs CMDP_K1L_PIPE=0)")
    ap.add_argument("--rollout-kernel", type=int, default=0, help="0 auto (K1E), 1 HBM tables, 2 LDS-resident K1L/K1P, 4 shared-table K1T, 5 K1U, 6 K1E")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --instances per GPU; strong: --instances in TOTAL, rank r takes the contiguous block "
                         "[r*B/N, (r+1)*B/N) of SURVEY 8(e)")
    ap.add_argument("--sustained-seconds", type=float, default=10.0, help="length of the sustained leg (0: skip)")
    ap.add_argument("--strong-share", type=int, default=8, help="rehearse one rank's share of a strong split over this many GPUs (0: skip)")
    ap.add_argument("--no-live-pmc", action="store_true", help="take roofline.traffic from the committed PMC summary instead of "
                    "measuring it in child runs under rocprofv3 (about 40 s)")
    ap.add_argument("--save-pmc", default="", help="write the counters measured by this run's child passes as the committed summary "
                    "(profiles/r04_pmc.json): what runs without rocprofv3 -- N > 1, a profiled run -- scale their fractions from")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the child run live_pmc() profiles
    ap.add_argument("--pmc-child-vi", action="store_true", help=argparse.SUPPRESS)   # ... with the VI leg (the SQ pass)
    ap.add_argument("--pmc-child-out", default="", help=argparse.SUPPRESS)           # ... which leaves its sweep count here
    ap.add_argument("--layout", default="csr", choices=["csr", "dense"],
                    help="layout of the HEADLINE leg: csr (default, fastest) or dense (then --launch-steps applies to K1D)")
    args = ap.parse_args()

    if args.pmc_child:   # a few launches of the headline and dense kernels, nothing else, nothing printed
        args.steps, args.warmup, args.dense_steps = 3, 1, 2
        args.no_cpu, args.no_live_pmc, args.sustained_seconds, args.strong_share = True, True, 0.0, 0
        # the VI leg of the SQ pass: a few instances built WITHOUT a process pool (the profiler's preload has initialised the GPU
        # before main(): no fork() from here) -- instructions per sweep do not depend on how many instances are solved
        args.vi_instances = min(256, args.vi_instances) if args.pmc_child_vi else 0
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    # Host-side model construction of the VI leg first: it uses a fork()ed process pool, which must not happen
    # after the HIP runtime / RCCL have been initialised in this process.
    fl, fl_build_s = None, 0.0
    if args.vi_instances > 0:
        workers = 1 if args.pmc_child else max(1, min(16, (os.cpu_count() or 1) // max(1, world)))
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

    lib = L.load()  # raises if the binary's build id is not the hash of the sources in the tree
    build_id = lib.cmdp_build_id().decode()
    assert lib.cmdp_device_count() > 0, "no HIP device visible: the product path has no CPU fallback"
    L.check(lib.cmdp_set_device(device_index))

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        import torch

        t = torch.tensor([x], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(x):
        if dist is None:
            return x
        import torch

        t = torch.tensor([x], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t)
        return float(t.item())

    ev = HipEvents()

    def timed_launches(env, n_launch, launch_steps, warmup):
        """W untimed + K timed fused-rollout launches; (wall seconds incl. barriers, per-launch HIP-event ms)."""
        stream = env.stream
        for _ in range(warmup):
            env.rollout_async(launch_steps)
        env.synchronize()
        marks = [ev.create() for _ in range(n_launch + 1)]
        barrier()
        env.synchronize()
        t0 = time.perf_counter()
        ev.record(marks[0], stream)
        for k in range(n_launch):
            env.rollout_async(launch_steps)
            ev.record(marks[k + 1], stream)
        env.synchronize()
        barrier()
        wall = time.perf_counter() - t0
        return wall, [ev.elapsed_ms(marks[k], marks[k + 1]) for k in range(n_launch)]

    # ---- C2 workload: this rank's shard of DeepSeaEpisodic(seed=i, size=30), i in [rank*B, (rank+1)*B) ----
    if args.scaling == "strong":
        assert args.instances % world == 0, "--scaling strong: --instances must be divisible by the number of ranks"
        B = args.instances // world
    else:
        B = args.instances
    seeds = np.arange(rank * B, (rank + 1) * B, dtype=np.int64)
    t_build = time.time()
    dense_headline = args.layout == "dense"
    tables = deepsea_episodic_tables(seeds, args.size, with_dp=dense_headline)
    keys = seeds.astype(np.uint64)  # Philox key = global instance id
    env = BatchedMDP(tables=tables, rng_mode=L.RNG_PHILOX, philox_keys=keys,
                     layout=L.LAYOUT_DENSE if dense_headline else L.LAYOUT_CSR)
    env.reset()
    if args.rollout_kernel:
        env.set_rollout_kernel(args.rollout_kernel)
    if args.lds_groups:
        env.set_option(L.OPT_LDS_GROUPS_PER_CU, args.lds_groups)
    t_build = time.time() - t_build
    S, A = int(env.n_states[0]), 2
    plan = env.lds_plan() if not dense_headline else dict(kernel="", eligible=False)
    lds_kernel = plan["kernel"]

    # Settling, before the W warm-up steps and outside every timed region: the first ~40 launches on a handle that has just been
    # built run 2-9 % slower than the ones after them (0.60 -> 0.55 ms per step, decaying launch by launch; the `sustained` leg
    # below shows the level they settle at) -- first-touch of the 1.6 GB of code-word buffers, clock / power management
    # settling after the idle build phase.  A fixed 0.05 s of untimed launches puts the W + K steps at that level.
    settle = 0 if dense_headline else max(0, args.settle_launches)
    for _ in range(settle):
        env.rollout_async(args.launch_steps)
    env.synchronize()
    elapsed, launch_ms = timed_launches(env, args.steps, args.launch_steps, args.warmup)
    # the kernels of the LAST timed step (HIP events inside the library, on the stream the kernels run on, recorded while the
    # steps ran back to back): parts of the step, so they add up to at most ms_per_step
    timed_kernel_ms = {}
    second_kernel = {"k_rollout_epi": "k_reward_scan", "k_rollout_tmpl_stream": "k_trace_hist"}.get(lds_kernel)
    if second_kernel:
        for name, which in ((lds_kernel, L.STAT_ROLLOUT_KERNEL_MS), (second_kernel, L.STAT_HIST_KERNEL_MS)):
            v = C.c_double()
            L.check(lib.cmdp_stat(env.handle, which, C.byref(v)))
            timed_kernel_ms[name] = v.value
    elapsed = max_over_ranks(elapsed)

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

    # ---- sustained leg: back-to-back launches for >= --sustained-seconds, SMI sampled meanwhile (N = 1 only) -------------
    sustained = None
    if args.sustained_seconds > 0 and world == 1 and not dense_headline:
        sustained = sustained_leg(env, ev, args.launch_steps, args.sustained_seconds, B)

    # ---- HBM traffic of the headline and dense kernels, measured now (child runs under rocprofv3 --pmc; N = 1 only) ------
    global LIVE_PMC
    # (not when this run is itself profiled: the profiler's preload has initialised the GPU before main(), and the GPU boxes
    # refuse a fork + exec from such a process -- the committed summary is used then)
    under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROFILER_", "ROCPROF_")) for k in os.environ)
    if world == 1 and not args.no_live_pmc and not dense_headline and not under_profiler:
        t_pmc = time.time()
        LIVE_PMC = live_pmc(["--instances", str(B), "--size", str(args.size), "--launch-steps", str(args.launch_steps),
                             "--dense-instances", str(args.dense_instances), "--dense-launch-steps", str(args.dense_launch_steps),
                             "--rollout-kernel", str(args.rollout_kernel), "--vi-instances", str(args.vi_instances), "--settle-launches", "0"], timeout=240)
        t_pmc = time.time() - t_pmc
        if LIVE_PMC and args.save_pmc:
            ks = []
            child_sweeps = (LIVE_PMC.get("_child") or {}).get("vi_sweeps_per_launch")
            for k, v in LIVE_PMC.items():
                if k == "_child":
                    continue
                e = {"kernel": k}
                if "read" in v and "write" in v:
                    e.update(hbm_read_bytes_per_launch=v["read"], hbm_write_bytes_per_launch=v["write"], hbm_bytes_per_launch=v["read"] + v["write"])
                e.update({c: v[c] for c in SQ_COUNTERS if c in v})
                if k in ("k_rollout_epi", "k_reward_scan", "k_epi_fold", "k_rollout_tmpl_stream", "k_trace_hist", "k_rollout_tmpl", "k_rollout_pipe", "k_rollout_lds"):
                    e["units_per_launch"] = B * args.launch_steps
                if k == "k_rollout_dense":
                    e["units_per_launch"] = args.dense_instances * args.dense_launch_steps
                if k.startswith("k_dp_reg") and child_sweeps and "SQ_INSTS_VALU" in v:
                    e.update(sweeps_per_launch=child_sweeps, valu_insts_per_sweep=v["SQ_INSTS_VALU"] / child_sweeps,
                             lds_insts_per_sweep=v.get("SQ_INSTS_LDS", 0.0) / child_sweeps,
                             lds_bank_conflict_frac=v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0))
                ks.append(e)
            json.dump({"build_id": build_id, "kernels": ks,
                       "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced streaming reads -> x2 "
                                     "(MI355X_MICROARCH.md, HBM section; calibrated with tools/calib/pmc_calib.hip); WRITE_SIZE exact",
                       "source": "bench.py --save-pmc: per-launch means of its rocprofv3 child passes (--pmc FETCH_SIZE; --pmc WRITE_SIZE; "
                                 "--pmc " + " ".join(SQ_COUNTERS) + "), --kernel-trace the only other option"},
                      open(args.save_pmc, "w"), indent=1)

    units_per_launch = B * args.launch_steps
    total_steps = world * units_per_launch * args.steps
    value = total_steps / elapsed
    avg_launch_s = float(np.mean(launch_ms)) * 1e-3
    kernel_rate = units_per_launch / avg_launch_s  # transitions/s of this rank's kernel

    def hbm_view(bytes_per_unit, accounting, kernel_prefix, units, launch_s):
        """Algorithmic bytes (SURVEY 8d per-unit figure x units per launch) and PMC traffic against the HBM peak."""
        pm = pmc_entry(kernel_prefix, units, build_id)
        algo = bytes_per_unit * units / launch_s / 1e9
        return {
            "algorithmic_bytes_per_unit": bytes_per_unit, "accounting": accounting,
            "algorithmic_GBps": algo, "algorithmic_over_peak": algo / HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "traffic": pm["bytes"] if pm else None,
            "traffic_GBps": pm["bytes"] / launch_s / 1e9 if pm else None,
            "traffic_frac": pm["bytes"] / launch_s / 1e9 / HBM_PEAK_GBS if pm else None,
            "traffic_source": ("%s (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, separate --pmc passes; collected on "
                               "build %s = %s build)" % (pm["source"], pm["collected_on_build"], "this" if pm["current"] else "ANOTHER")) if pm else None,
            "traffic_measured_in_this_run": bool(LIVE_PMC) and pm is not None and pm["source"].startswith("this run"),
            "traffic_current": pm["current"] if pm else None,
        }

    if dense_headline:
        bytes_per_step = 4 * S + 28
        hv = hbm_view(bytes_per_step, "SURVEY 8(d) dense-row figure (4*S+28 B/transition)", "k_rollout_dense", units_per_launch, avg_launch_s)
        roofline = {"bound": "hbm", "kernel": "k_rollout_dense<0,NV>", "achieved": hv["algorithmic_GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": hv["algorithmic_over_peak"], "traffic": hv["traffic"], "hbm": hv}
    elif lds_kernel == "k_rollout_epi":
        # K1E, the episode-parallel rollout: lane = (instance, episode), 4 096 independent 30-step chains per CU -- bound by
        # instruction THROUGHPUT, not by a chain's latency.  Its roofline is hardware-anchored: wave-instructions issued
        # (SQ_INSTS_VALU) and LDS-pipe cycles (SQ_LDS_IDX_ACTIVE) per launch from an SQ counter pass of THIS run, against the
        # SIMDs' issue capacity and the CUs' LDS cycles over the kernel's duration.
        hv = hbm_view(8 + 8 * 1 + 28, "SURVEY 8(d) CSR figure (8 + 8*nnz + 28 = 44 B/transition); the tables are LDS-resident, so "
                      "these bytes never cross HBM -- reported as an equivalent rate, NOT a roofline fraction",
                      "k_rollout_epi", units_per_launch, avg_launch_s)
        k_ms = timed_kernel_ms.get("k_rollout_epi")
        dom_s = (k_ms or avg_launch_s * 1e3) * 1e-3
        sq = sq_view("k_rollout_epi", dom_s, units_per_launch, build_id)
        sq2 = sq_view("k_reward_scan", (timed_kernel_ms.get("k_reward_scan") or 0.0) * 1e-3, units_per_launch, build_id)
        episodes_pl = -(-args.launch_steps // int(env.H)) + 1
        own_bytes = B * S * 4 + 12 * B * episodes_pl + 16 * B * S        # table image in, code + count words out, departure image RMW
        scan_bytes = 12 * B * episodes_pl + 16 * B
        pm_e, pm_r = pmc_entry("k_rollout_epi", units_per_launch, build_id), pmc_entry("k_reward_scan", units_per_launch, build_id)
        peak2 = N_CUS * 4 * CLOCK_HZ / VALU_CYCLES / 1e9
        roofline = {
            "bound": "valu_issue", "kernel": "k_rollout_epi",
            "unit": "G wave-instructions/s",
            "achieved": sq["valu_G_wave_insts_per_s"] if sq else None, "peak": peak2,
            "frac": sq["valu_G_wave_insts_per_s"] / peak2 if sq else None,
            "model": "achieved = SQ_INSTS_VALU per launch (SQ pass of this run) / the kernel's duration in the timed region (HIP events "
                     "inside the library); peak = %d CUs x 4 SIMDs x %.1f GHz / %d cycles per wave64 VALU instruction (MI355X_MICROARCH.md: "
                     "the float32 figure)" % (N_CUS, CLOCK_HZ / 1e9, VALU_CYCLES),
            "frac_at_measured_int_rate": sq["valu_G_wave_insts_per_s"] / (N_CUS * 4 * CLOCK_HZ / VALU_INT_VOP3_CYCLES / 1e9) if sq else None,
            "int_rate_note": "the step's VOP3 instructions (v_and_or_b32, v_alignbit_b32; v_bfe_u32, v_lshl_or_b32) occupy a SIMD for %.1f cycles "
                             "per wavefront at 4-8 wavefronts per SIMD, not 2 (tools/calib/valu_int_rate.hip, profiles/r04_valu_int_rate.txt): "
                             "against THAT issue rate the kernel is at frac_at_measured_int_rate" % VALU_INT_VOP3_CYCLES,
            "lds_atomic_pipe_frac": (units_per_launch / 64.0) * LDS_ATOMIC_RTN_CYCLES / (N_CUS * dom_s * CLOCK_HZ),
            "lds_atomic_note": "one returning LDS atomic per transition (it is the table read and the visit count): units / 64 wave-atomics x %.2f "
                               "cycles of a CU's LDS pipe each (measured: tools/calib/lds_atomic_rate.hip, profiles/r04_lds_atomic_rate.txt; a "
                               "ds_read_b32 takes 1.1) over the kernel's CU-cycles.  SQ_LDS_IDX_ACTIVE (`lds_pipe_frac`) counts ~2 cycles per LDS "
                               "instruction of any kind and understates it" % LDS_ATOMIC_RTN_CYCLES,
            "step_loop_frac": (units_per_launch / 64.0) * WALK_STEP_CYCLES_PER_SIMD / (N_CUS * 4 * dom_s * CLOCK_HZ),
            "step_loop_note": "the walk's step ALONE (4 VALU + 1 returning LDS atomic, four chains per lane, 16 wavefronts per CU, nothing else "
                              "in the loop) runs at %.1f cycles per wave-transition per SIMD on this chip -- both pipes nearly full (VALU ~17, LDS "
                              "4 x 4.31 = 17.2): step_loop_frac = that floor / the kernel's duration, i.e. how much of the kernel is "
                              "the irreducible step; the rest is Philox, code / count words, staging, flush" % WALK_STEP_CYCLES_PER_SIMD,
            "transitions_per_s_in_kernel": units_per_launch / dom_s,
            "valu_wave_insts_per_transition": sq["valu_wave_insts_per_launch"] * 64 / units_per_launch if sq else None,
            "sq": sq, "lds_pipe_frac": sq["lds_pipe_frac"] if sq else None,
            "kernel_ms": timed_kernel_ms,
            "kernel_ms_note": "of the last timed step, back to back with its neighbours (HIP events inside the library).  k_reward_scan runs on the "
                              "handle's SECOND stream under the k_rollout_epi of the next step (CMDP_K1E_OVERLAP=0: one stream): the two durations "
                              "overlap in time, ms_per_step ~ k_rollout_epi's; each of them is <= ms_per_step",
            "lds_plan": plan, "traffic": hv["traffic"],
            "hbm": dict(hv, own_algorithmic_bytes_per_launch=own_bytes,
                        own_accounting="k_rollout_epi itself: table image read once per group of 32 instances (4 B per state), 12 B of reward-code + count "
                                       "words written per episode, 16 B per state of the departure-count image (a returnless 64-bit atomic add: "
                                       "line in, line out)",
                        own_algorithmic_GBps=own_bytes / dom_s / 1e9,
                        own_traffic_over_algorithmic=(pm_e["bytes"] / own_bytes) if pm_e else None),
            "step": {
                "what": "a step = k_rollout_epi (walk + departure counts) followed by k_reward_scan (float64 reward sums in transition order, "
                        "added exactly in integers inside a binade); `achieved` / `frac` above are k_rollout_epi's over ITS duration",
                "transitions_per_s": kernel_rate,
                "reward_scan": {"kernel": "k_reward_scan", "bound": "latency (one wavefront per SIMD: 65 536 sequential sums)",
                                "ms": timed_kernel_ms.get("k_reward_scan"), "sq": sq2,
                                "algorithmic_bytes_per_launch": scan_bytes, "traffic": pm_r["bytes"] if pm_r else None},
            },
        }
    else:
        # The LDS-resident kernels: what bounds a launch is (instances / chains resident on the chip) x transitions x
        # the latency of the dependent LDS read chain of one transition.  The chain latency is measured in this run.
        ns_read, ns_chain = C.c_double(), C.c_double()
        L.check(lib.cmdp_calibrate(L.CALIB_LDS_READ, 200000, C.byref(ns_read)))
        streamed = plan.get("kernel") == "k_rollout_tmpl_stream"   # K1U: K1T's chain, visit counts histogrammed from an HBM trace
        shared_table = plan.get("kernel") == "k_rollout_tmpl" or streamed   # the chain reads the word pair and the swap bit
        L.check(lib.cmdp_calibrate(L.CALIB_LDS_CHAIN_SHARED if shared_table else L.CALIB_LDS_CHAIN, 200000, C.byref(ns_chain)))
        kernel_ms = {}
        if streamed:
            # K1U: ONE successor table per CU and per instance only the swap bits in LDS, so LDS capacity (and the 32 wavefronts
            # of a CU) would hold ~2 000 chains: what caps the resident chains is the BATCH -- B / 256 CUs
            min_footprint = (S + 7) // 8
            chains_per_cu = min((LDS_BYTES - 2 * S * A) // min_footprint, 32 * 64, -(-B // N_CUS))
            footprint_txt = "swap bits per instance beside one shared %d-B table; capped by the 32 wavefronts of a CU and by the batch (B / %d CUs)" % (2 * S * A, N_CUS)
            # per-kernel times of a step (HIP events inside the library, on the stream the kernels run on): three more
            # launches outside the timed region, each read back
            kernel_ms = dict(timed_kernel_ms)   # of the last timed step (back to back): parts of ms_per_step
        elif shared_table:
            # K1T: ONE uint16 successor table per CU; a chain needs its 8-bit visit-count deltas and one swap bit per state
            min_footprint = S * A + (S + 7) // 8
            chains_per_cu = (LDS_BYTES - 2 * S * A) // min_footprint
            footprint_txt = "count deltas + swap bits per instance beside one shared %d-B table" % (2 * S * A)
        else:
            min_footprint = 2 * S * A + S * A  # uint16 successor words + 8-bit visit-count deltas: what a chain needs in LDS
            chains_per_cu = LDS_BYTES // min_footprint
            footprint_txt = "tables+count deltas per instance"
        peak = N_CUS * chains_per_cu / (ns_chain.value * 1e-9)  # transitions/s with every CU's LDS full of chains
        hv = hbm_view(8 + 8 * 1 + 28, "SURVEY 8(d) CSR figure (8 + 8*nnz + 28 = 44 B/transition); the tables are LDS-resident, so "
                      "these bytes never cross HBM -- reported as an equivalent rate, NOT a roofline fraction",
                      lds_kernel if args.rollout_kernel != 1 else "k_rollout<", units_per_launch, avg_launch_s)
        dom_rate = units_per_launch / (kernel_ms["k_rollout_tmpl_stream"] * 1e-3) if streamed else kernel_rate
        roofline = {
            "bound": "lds_latency",
            "kernel": lds_kernel if args.rollout_kernel != 1 else "k_rollout<0,false>",
            "achieved": dom_rate / 1e9, "peak": peak / 1e9, "unit": "G transitions/s (one dependent LDS read each)",
            "frac": dom_rate / peak,
            "frac_against_bare_lds_read": dom_rate / (N_CUS * chains_per_cu / (ns_read.value * 1e-9)),
            "model": "peak = %d CUs x floor(160 KiB / %d B of %s) = %d resident chains per CU / "
                     "calibrated dependent-read chain latency" % (N_CUS, min_footprint, footprint_txt, chains_per_cu),
            "calibrated_chain_ns": ns_chain.value, "calibrated_bare_lds_read_ns": ns_read.value,
            "resident_chains_per_cu": plan.get("instances_per_workgroup"), "lds_plan": plan,
            "traffic": hv["traffic"], "hbm": hv,
        }
        if streamed:
            epp = 12 if S * A <= 1024 else 8   # trace entries per 16-byte piece (ten-bit rows when they fit)
            hist_bytes = units_per_launch * 16 / epp + 8 * B * S * A + 8 * B * S
            roofline["kernel_ms"] = kernel_ms
            pm_h = pmc_entry("k_trace_hist", units_per_launch, build_id)
            roofline["step"] = {
                "what": "a step = k_rollout_tmpl_stream (the chains; 16-bit trace -> HBM) followed by k_trace_hist (visit counts from "
                        "the trace): `achieved` / `frac` above are the dominant kernel's over ITS duration, these are the whole step's",
                "achieved": kernel_rate / 1e9, "frac": kernel_rate / peak,
                "hbm_traffic_bytes": (hv["traffic"] + pm_h["bytes"]) if (hv["traffic"] and pm_h) else None,
                "frac_against_bare_lds_read": kernel_rate / (N_CUS * chains_per_cu / (ns_read.value * 1e-9)),
                "hist_roofline": {"bound": "hbm", "kernel": "k_trace_hist",
                                  "algorithmic_bytes_per_launch": hist_bytes,
                                  "accounting": "trace read (16 B per %d transitions) + read-modify-write of visits_sa and visits_s" % epp,
                                  "achieved": (hist_bytes) / (kernel_ms["k_trace_hist"] * 1e-3) / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": pm_h["bytes"] if pm_h else None,
                                  "traffic_frac": pm_h["bytes"] / (kernel_ms["k_trace_hist"] * 1e-3) / 1e9 / HBM_PEAK_GBS if pm_h else None,
                                  "frac": (hist_bytes) / (kernel_ms["k_trace_hist"] * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    roofline["launch_ms_avg"] = avg_launch_s * 1e3
    roofline["launch_ms_min"] = float(np.min(launch_ms))

    line = {
        "metric": "env steps/sec (whole node) + value-iteration sweeps/sec, DeepSea size=%d x%d" % (args.size, B),
        "value": value,
        "unit": "env steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "i32 state/visit indices, f64 rewards+CDF",
        "data": "synthetic",
        "config": {
            "workload": "C2: DeepSeaEpisodic(seed=i,size=%d), %d instances per GPU, on-device uniform random policy "
                        "(Philox-4x32-10), %d transitions per instance per step, auto-reset at h>=H" % (args.size, B, args.launch_steps),
            "instances_per_gpu": B, "states": S, "actions": A, "horizon": int(env.H),
            "transitions_per_instance_per_step": args.launch_steps, "layout": args.layout, "rng": "philox4x32-10",
            "build_s": round(t_build, 2),
            "settle_launches": settle,
            "settle_note": "untimed launches of the same step before the W warm-up steps: the first ~40 launches on a freshly built handle "
                           "run 2-9 % slower than the level `sustained` shows (first-touch of the code-word buffers, clock / power "
                           "management after the idle build phase); the in-run oracle check covers them too",
        },
        "build_id": build_id[:16],
        "roofline": roofline,
        "parity_notes": "state indices / visit counts bit-exact; Jacobi VI/PE bit-exact incl. sweep counts; Gauss-Seidel VI/PE and "
                        "episodic values within 2e-6 absolute of the reference's run in the development container (NOT the 1e-6 of "
                        "the north star: the reference sums rows with BLAS sgemv, whose accumulation order belongs to the BLAS kernel "
                        "of the CPU -- the reference's own runs under OpenBLAS's Sandybridge and Haswell kernels differ by 2.86e-6 "
                        "on the same instances, profiles/r03_gs_blas_kernels.json; this build is 1.07e-6 from the Haswell-family runs); "
                        "Beta rewards bit-exact through the batched path (reward caches filled from each MDP's numpy stream)",
    }
    if gather_ms is not None:
        line["gather_ms"] = gather_ms
    if sustained is not None:
        line["sustained"] = sustained

    # ---- strong-scaling rehearsal: one rank's share of 65 536 instances TOTAL over `--strong-share` GPUs ---------------
    if args.strong_share > 1 and world == 1 and args.scaling == "weak" and not dense_headline and B % args.strong_share == 0:
        Bs = B // args.strong_share
        senv = BatchedMDP(tables=deepsea_episodic_tables(seeds[:Bs], args.size, with_dp=False), rng_mode=L.RNG_PHILOX,
                          philox_keys=keys[:Bs])
        senv.reset()
        if args.rollout_kernel:
            senv.set_rollout_kernel(args.rollout_kernel)
        for _ in range(settle):   # the same settling launches as the headline handle got
            senv.rollout_async(args.launch_steps)
        senv.synchronize()
        s_wall, s_ms = timed_launches(senv, args.steps, args.launch_steps, args.warmup)
        s_launch = float(np.mean(s_ms)) * 1e-3
        line["strong"] = {
            "what": "SURVEY 8(e): instance_id -> gpu = id*N/B.  %d instances in TOTAL over %d GPUs = %d per GPU, rehearsed on this "
                    "GPU with rank 0's share; no data-path collective, so the N-GPU step time is one rank's step time" % (B, args.strong_share, Bs),
            "instances_per_gpu": Bs, "n_gpus_modelled": args.strong_share, "lds_plan": senv.lds_plan(),
            "ms_per_step": s_launch * 1e3, "env_steps_per_s_per_gpu": Bs * args.launch_steps / s_launch,
            "implied_value_at_n_gpus": args.strong_share * Bs * args.launch_steps / s_launch,
            "implied_speedup_vs_1_gpu": avg_launch_s / s_launch,
            "note": "K1E walks episodes, not instances: an eighth of the instances is an eighth of the workgroups and of the walk; "
                    "what does not shrink is the reward scan (one sequential float64 sum per instance: latency-bound whatever the batch)",
        }
        senv.close()

    # ---- dense-row leg (north-star layout, K1D): same instances, rows streamed from HBM -------------------------
    dense_env = None
    if args.dense_instances > 0 and not dense_headline:
        Bd = args.dense_instances
        td = time.time()
        dseeds = seeds[:Bd]
        dtables = deepsea_episodic_tables(dseeds, args.size, with_dp=True)
        dense_env = BatchedMDP(tables=dtables, rng_mode=L.RNG_PHILOX, philox_keys=dseeds.astype(np.uint64), layout=L.LAYOUT_DENSE)
        dense_env.reset()
        td = time.time() - td
        d_wall, d_ms = timed_launches(dense_env, args.dense_steps, args.dense_launch_steps, 2)
        d_wall = max_over_ranks(d_wall)
        d_units = Bd * args.dense_launch_steps
        d_launch_s = float(np.mean(d_ms)) * 1e-3
        hv = hbm_view(4 * S + 28, "SURVEY 8(d) dense-row figure (4*S+28 = %d B/transition)" % (4 * S + 28), "k_rollout_dense", d_units, d_launch_s)
        line["dense"] = {
            "workload": "C2 in the north star's layout: per-instance dense float32 P[s,a,:] rows in HBM (%.0f GB), wavefront "
                        "prefix-sum CDF lookup, %d instances per GPU, %d transitions per instance per launch, %d launches"
                        % (Bd * S * A * 512 * 4 / 1e9, Bd, args.dense_launch_steps, args.dense_steps),
            "value": world * d_units * args.dense_steps / d_wall, "unit": "env steps/s", "build_s": round(td, 2),
            "roofline": {"bound": "hbm", "kernel": "k_rollout_dense<0,NV>", "achieved": hv["algorithmic_GBps"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": hv["algorithmic_over_peak"], "traffic": hv["traffic"], "traffic_frac": hv["traffic_frac"],
                         "traffic_source": hv["traffic_source"], "traffic_current": hv["traffic_current"],
                         "algorithmic_bytes_per_transition": 4 * S + 28, "launch_ms_avg": d_launch_s * 1e3, "launch_ms_min": float(np.min(d_ms))},
        }

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
        vi_V, vi_sweeps = np.array(V, copy=True), np.array(sw, copy=True)
        vi_s = max_over_ranks(time.perf_counter() - t1)
        my_sweeps = float(sw.sum())
        if args.pmc_child and args.pmc_child_out:
            json.dump({"vi_sweeps_per_launch": my_sweeps}, open(args.pmc_child_out, "w"))
        sweeps = sum_over_ranks(my_sweeps)
        kms, kid, kms_dev = C.c_double(), C.c_double(), C.c_double()
        L.check(lib.cmdp_stat(dp.handle, L.STAT_DP_KERNEL_MS, C.byref(kms)))
        L.check(lib.cmdp_stat(dp.handle, L.STAT_DP_KERNEL, C.byref(kid)))
        # the same solve with the results left in device memory (pageable outputs: the kernel does not store across PCIe):
        # the sweep kernel's own duration, which the VALU-issue roofline below is taken against
        dp.value_iteration(0.99, 1e-6)
        L.check(lib.cmdp_stat(dp.handle, L.STAT_DP_KERNEL_MS, C.byref(kms_dev)))
        kname = {1: "k_dp_block", 2: "k_dp_reg", 5: "k_dp_regu", 6: "k_dp_wave_gs", 7: "k_dp_regw"}.get(int(kid.value), "?")
        nS = int(np.max(np.diff(fl["state_off"])))
        nnz = int(len(fl["csr_col"])) / args.vi_instances
        line["vi"] = {
            "workload": "C3: FrozenLakeContinuous(seed=i,size=20,p_frozen=0.9,is_slippery=True,p_rand=0.1), %d instances "
                        "per GPU, discounted VI gamma=0.99 eps=1e-6, scheme by the reference rule (Jacobi)" % args.vi_instances,
            "sweeps_per_s": sweeps / vi_s, "instances_per_s": world * args.vi_instances / vi_s, "total_sweeps": sweeps,
            "wall_ms": vi_s * 1e3, "includes": "H2D of nothing; Q, V, sweep counts and status stored by the kernel straight into reused page-locked host "
                                             "buffers as instances converge (no copy after the kernel)", "build_s": round(tb, 2),
            "kernel_ms": kms.value, "kernel_ms_note": "HIP-event time of the sweep kernel INCLUDING its result stores across PCIe",
            "kernel_ms_results_on_device": kms_dev.value, "kernel_sweeps_per_s": my_sweeps / (kms_dev.value * 1e-3),
        }
        # VI roofline.  The CSR is read from HBM once per solve and kept in registers, V lives in LDS: neither HBM nor MFMA
        # is in play.  The sweep is bound by VALU issue; the instruction count per workgroup-sweep comes from the
        # committed SQ_INSTS_VALU pass of this kernel (profiles/r02_pmc.json), sweeps and kernel time are live.
        vr = {"bound": "valu_issue", "kernel": kname, "peak": N_CUS * 4 * CLOCK_HZ / VALU_CYCLES / 1e9,
              "unit": "G wave-instructions/s", "achieved": None, "frac": None,
              "model": "peak = %d CUs x 4 SIMDs x %.1f GHz / %d cycles per wave64 VALU instruction" % (N_CUS, CLOCK_HZ / 1e9, VALU_CYCLES),
              "hbm": {"bytes_per_sweep_csr_figure": 8 * nnz + 4 * (nS * 4 + 1) + 4 * nS * 4 + 8 * nS,
                      "note": "SURVEY 8(d) CSR figure; read from HBM once per SOLVE, not per sweep -- not a roofline"}}
        live = (LIVE_PMC or {}).get(kname)
        child_sweeps = ((LIVE_PMC or {}).get("_child") or {}).get("vi_sweeps_per_launch")
        if live and "SQ_INSTS_VALU" in live and child_sweeps:   # the SQ child pass of this run solved a slice of this batch
            per_sweep = live["SQ_INSTS_VALU"] / child_sweeps
            vr["achieved"] = per_sweep * my_sweeps / (kms_dev.value * 1e-3) / 1e9
            vr["frac"] = vr["achieved"] / vr["peak"]
            vr["valu_wave_insts_per_workgroup_sweep"] = per_sweep
            vr["lds_insts_per_sweep"] = live.get("SQ_INSTS_LDS", 0.0) / child_sweeps
            vr["lds_bank_conflict_frac"] = live.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(live.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)
            vr["counter_source"] = ("this run: rocprofv3 --pmc SQ child pass of bench.py (SQ_INSTS_VALU per launch / sweeps of the launch, "
                                    "the first 256 instances of this batch) x the sweeps of the timed solve")
            vr["counters_measured_in_this_run"] = True
        try:
            j = json.load(open(PMC_FILE)) if vr["achieved"] is None else {}
            for k in j.get("kernels", []):
                if k["kernel"].startswith(kname) and "valu_insts_per_sweep" in k:
                    inst = k["valu_insts_per_sweep"] * my_sweeps
                    vr["achieved"] = inst / (kms_dev.value * 1e-3) / 1e9
                    vr["frac"] = vr["achieved"] / vr["peak"]
                    vr["valu_wave_insts_per_workgroup_sweep"] = k["valu_insts_per_sweep"]
                    vr["counter_source"] = "profiles/%s: SQ_INSTS_VALU / total sweeps, collected on build %s = %s build" % (
                        os.path.basename(PMC_FILE), j.get("build_id", "")[:16], "this" if j.get("build_id") == build_id else "ANOTHER")
                    for extra in ("lds_insts_per_sweep", "lds_idx_active_frac", "lds_bank_conflict_frac", "valu_active_frac"):
                        if extra in k:
                            vr[extra] = k[extra]
        except Exception:
            pass
        if vr["frac"] is not None and kname == "k_dp_regw":
            # k_dp_regw keeps an instance's whole CSR in registers (174 of 253 VGPRs at C3): TWO wavefronts per SIMD, and at two
            # wavefronts per SIMD a float32 VALU instruction issues every 4.22 cycles on this chip, not every 2 (measured:
            # tools/calib/valu_int_rate.hip, profiles/r04_valu_int_rate.txt, v_fma_f32 at waves/SIMD 2)
            vr["frac_at_measured_rate_two_waves_per_simd"] = vr["frac"] * VALU_FP32_CYCLES_2_WAVES / VALU_CYCLES
            vr["measured_rate_note"] = ("the kernel runs at two wavefronts per SIMD (its tables fill the register file); there a wave64 float32 VALU "
                                        "instruction issues every %.2f cycles, not every %d (tools/calib/valu_int_rate.hip, profiles/r04_valu_int_rate.txt): "
                                        "against THAT rate the sweep is at frac_at_measured_rate_two_waves_per_simd -- the SIMDs issue back to back; "
                                        "packed float32 (bit-equal, 144 fewer instructions per sweep) did not shorten the sweep -- a v_pk_mul_f32 / v_pk_add_f32 issues "
                                        "every 6.1 cycles there, two plain ones every 8.4, and the pairs cost register moves; more wavefronts "
                                        "per SIMD would (K2U, the instance over four wavefronts, pays a barrier per sweep for them: 2.4 ms)"
                                        % (VALU_FP32_CYCLES_2_WAVES, VALU_CYCLES))
        line["vi"]["roofline"] = vr
        dp.close()

    # ---- CPU baseline: the oracle, one core, bounded sample of the same workload (rank 0, N = 1 only) -----
    if rank == 0 and world == 1 and not args.no_cpu and args.cpu_instances > 0:
        from oracle import oracle as O

        n = min(args.cpu_instances, B)
        n_steps = args.launch_steps * (args.steps + args.warmup + settle)  # everything the GPU did since reset()
        c0 = time.perf_counter()
        last, rsum, cvs, _ = O.batch_rollout(tables, 0, n, n_steps, rng_mode=1, philox_keys=keys, want_visits=True)
        cpu_s = time.perf_counter() - c0
        # the timed GPU work is checked, not just timed: state-visit counts of the sampled instances are bit-equal
        verified = bool(np.array_equal(cvs, vs[: n * S]))
        assert verified, "GPU visit counts differ from the CPU oracle"
        line["verified_against_oracle"] = "state-visit counts of instances 0..%d after %d transitions: bit-equal" % (n - 1, n_steps)
        if dense_env is not None:
            nd = min(64, args.dense_instances)
            d_steps = args.dense_launch_steps * (args.dense_steps + 2)
            _, _, dcvs, _ = O.batch_rollout(dtables, 0, nd, d_steps, rng_mode=1, philox_keys=keys, want_visits=True)
            dvs, _ = dense_env.visits(sa=False)
            assert np.array_equal(dcvs, dvs[: nd * S]), "dense-layout visit counts differ from the CPU oracle"
            line["dense"]["verified_against_oracle"] = "state-visit counts of instances 0..%d after %d transitions: bit-equal" % (nd - 1, d_steps)
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

            n_mt = min(B, 128 * cores)
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
        # ---- numpy restatement of the step loop (oracle/numpy_port.py), same Philox streams, checked against the
        # C port: (i) one numpy operation per step over a slice of instances, (ii) one Python iteration per step of one
        # instance -- the reference's execution model.  The reference's own rate is quoted beside them.
        from oracle import numpy_port as NP

        nn, nsteps_np = min(512, B), 30000
        c0 = time.perf_counter()
        _, np_rsum, np_vs, _ = NP.rollout_vectorised(tables, 0, nn, nsteps_np, keys)
        np_s = time.perf_counter() - c0
        _, o_rsum, o_vs, _ = O.batch_rollout(tables, 0, nn, nsteps_np, rng_mode=1, philox_keys=keys, want_visits=True)
        assert np.array_equal(np_vs, o_vs) and np.array_equal(np_rsum, o_rsum), "numpy restatement differs from the C oracle"
        c0 = time.perf_counter()
        NP.step_loop_python(tables, 0, 300_000, int(keys[0]))
        py_s = time.perf_counter() - c0
        line["cpu_baseline_numpy"] = {
            "value": nn * nsteps_np / np_s, "unit": "env steps/s", "cores": 1, "kind": "port (numpy restatement, vectorised over instances)",
            "sample": "instances 0..%d, %d transitions each, %.1f s; visit counts and reward sums bit-equal to the C port" % (nn - 1, nsteps_np, np_s),
            "python_step_loop": {"value": 300_000 / py_s, "unit": "env steps/s", "cores": 1,
                                 "what": "one instance, one Python iteration per step over plain tables (the reference's execution model "
                                         "without its dm_env / sampler-object overhead)"},
        }
        line["reference_in_container"] = {
            "value": 5.8e4, "unit": "env steps/s per core", "measured": False,
            "source": "BASELINE.md section 2: bare BaseMDP.step loop of the reference itself, DeepSeaEpisodic(size=30), 200 000 random "
                      "actions, one core of the development container (survey-time measurement; the reference does not travel to "
                      "the GPU box).  With its Q-learning agent in the loop: 1.9e4.  This, not the C port above, is what "
                      "'Colosseum on a CPU core' means (reference default: 1 core, colosseum/config.py:19)"}
        if "vi" in line and args.vi_instances > 0:
            nv = min(256, args.vi_instances)
            c0 = time.perf_counter()
            _, oracle_V, swc = O.batch_vi(fl, 0, nv, 0.99, 1e-6, 0)
            cv = time.perf_counter() - c0
            line["vi"]["cpu_baseline"] = {"value": float(swc.sum()) / cv, "unit": "sweeps/s", "cores": 1, "kind": "port",
                                          "sample": "instances 0..%d, %.1f s" % (nv - 1, cv)}
            # the timed VI leg is checked, not just timed: V and the sweep counts of the sampled instances are bit-equal
            n_states_v = int(fl["state_off"][nv])
            assert np.array_equal(swc, vi_sweeps[:nv]), "GPU sweep counts differ from the CPU oracle"
            assert np.array_equal(oracle_V[:n_states_v], vi_V[:n_states_v]), "GPU value functions differ from the CPU oracle"
            line["vi"]["verified_against_oracle"] = "V and sweep counts of instances 0..%d: bit-equal" % (nv - 1)
            # numpy restatement of the sweep (oracle/numpy_port.py), a few instances
            from oracle import numpy_port as NP

            so, cp = fl["state_off"], fl["csr_ptr"]
            c0 = time.perf_counter()
            nsw = 0
            for b in range(4):
                r0, r1 = int(so[b]) * 4, int(so[b + 1]) * 4
                e0, e1 = int(cp[r0]), int(cp[r1])
                _, Vn, k = NP.jacobi_vi(np.asarray(cp[r0:r1 + 1]) - e0, fl["csr_col"][e0:e1], fl["csr_val"][e0:e1], fl["R"][r0:r1],
                                        int(so[b + 1] - so[b]), 4, 0.99, 1e-6)
                assert k == int(swc[b]) and np.array_equal(Vn, oracle_V[int(so[b]):int(so[b + 1])])
                nsw += k
            line["vi"]["cpu_baseline_numpy"] = {"value": nsw / (time.perf_counter() - c0), "unit": "sweeps/s", "cores": 1,
                                                "kind": "port (numpy restatement, bit-equal to the C port)", "sample": "instances 0..3"}
    if dense_env is not None:
        dense_env.close()
    env.close()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
