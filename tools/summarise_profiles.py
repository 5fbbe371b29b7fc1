#!/usr/bin/env python3
"""Copies what tools/collect_profiles.sh left under gpurun_out/TAG_* into profiles/ and writes profiles/r04_pmc.json,
the summary bench.py reads for `roofline.traffic` (HBM bytes per launch: FETCH_SIZE x2 per the gfx950 correction for
wide coalesced reads + WRITE_SIZE, separate --pmc passes) and for the VI leg's instruction counts (SQ_INSTS_VALU etc. per
sweep).  The summary carries the build id of the library the passes ran on; bench.py reports whether that is the build
it runs.      python tools/summarise_profiles.py TAG"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "rNN"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
prof = os.path.join(root, "profiles")


def newest(pat):
    return max(glob.glob(pat), key=os.path.getmtime)


def counters(leg):
    """{kernel name: {counter: (mean per launch, launches)}} of one pass."""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(newest(f"{out}/{tag}_{leg}/*/*_counter_collection.csv"))):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in cs.items()} for k, cs in acc.items()}


line = json.loads(open(f"{out}/{tag}_bench_line.json").read().strip().splitlines()[-1])
under = json.loads(open(f"{out}/{tag}_sq1.out").read().strip().splitlines()[-1])
print("bench:", line["value"], line["ms_per_step"], line["roofline"]["frac"], "vi", line["vi"]["sweeps_per_s"], line["vi"]["kernel_ms"])
ks = newest(f"{out}/{tag}_stats/*/*_kernel_stats.csv")
for r in list(csv.DictReader(open(ks)))[:4]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
shutil.copy(ks, f"{prof}/{tag}_kernel_stats.csv")
shutil.copy(f"{out}/{tag}_bench_line.json", f"{prof}/{tag}_bench_line.json")
shutil.copy(f"{out}/{tag}_bench_line_under_rocprof.json", f"{prof}/{tag}_bench_line_under_rocprof.json")

legs = {leg: counters(leg) for leg in ("fetch", "write", "sq1", "sq2")}
w = csv.writer(open(f"{prof}/{tag}_pmc_counters.csv", "w"))
w.writerow(["pass", "kernel", "counter", "avg_value_per_launch", "launches"])
for leg, ks_ in legs.items():
    for k, cs in ks_.items():
        for c, (v, n) in cs.items():
            w.writerow([f"{tag}_{leg}", k, c, v, n])

cfg = line["config"]
units = {"k_rollout_epi": cfg["instances_per_gpu"] * cfg["transitions_per_instance_per_step"],
         "k_reward_scan": cfg["instances_per_gpu"] * cfg["transitions_per_instance_per_step"],
         "k_rollout_pipe": cfg["instances_per_gpu"] * cfg["transitions_per_instance_per_step"],
         "k_rollout_tmpl_stream": cfg["instances_per_gpu"] * cfg["transitions_per_instance_per_step"],
         "k_trace_hist": cfg["instances_per_gpu"] * cfg["transitions_per_instance_per_step"],
         "k_rollout_tmpl": cfg["instances_per_gpu"] * cfg["transitions_per_instance_per_step"],
         "k_rollout_lds": cfg["instances_per_gpu"] * cfg["transitions_per_instance_per_step"],
         "k_rollout_dense": 65536 * 200}
kernels = []
names = set(legs["fetch"]) | set(legs["sq1"])
for k in sorted(names):
    short = k.split("(")[0].replace("void ", "")
    e = {"kernel": short}
    f = legs["fetch"].get(k, {}).get("FETCH_SIZE")
    wr = legs["write"].get(k, {}).get("WRITE_SIZE")
    if f and wr:
        e.update(FETCH_SIZE_KB_reported=f[0], WRITE_SIZE_KB_reported=wr[0], hbm_read_bytes_per_launch=f[0] * 2048,
                 hbm_write_bytes_per_launch=wr[0] * 1024, hbm_bytes_per_launch=f[0] * 2048 + wr[0] * 1024)
    for pre, u in units.items():
        if short.split("<")[0] == pre:
            e["units_per_launch"] = u
    s1, s2 = legs["sq1"].get(k, {}), legs["sq2"].get(k, {})
    for c, (v, n) in list(s1.items()) + list(s2.items()):
        e[c] = v
    if short.startswith("k_dp_reg") and "SQ_INSTS_VALU" in e:
        sweeps = under["vi"]["total_sweeps"]
        e.update(sweeps_per_launch=sweeps, valu_insts_per_sweep=e["SQ_INSTS_VALU"] / sweeps,
                 lds_insts_per_sweep=e["SQ_INSTS_LDS"] / sweeps, salu_insts_per_sweep=e.get("SQ_INSTS_SALU", 0) / sweeps,
                 lds_bank_conflict_frac=e["SQ_LDS_BANK_CONFLICT"] / max(e["SQ_LDS_IDX_ACTIVE"], 1))
        if "SQ_WAVE_CYCLES" in e:
            e["wave_cycles_waiting_frac"] = e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"]          # parked: s_waitcnt / barrier
            e["wave_cycles_issue_stalled_frac"] = e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"]
    if "GRBM_GUI_ACTIVE" in e and "SQ_LDS_IDX_ACTIVE" in e:
        cyc = e["GRBM_GUI_ACTIVE"] / 8  # rocprofv3 sums the 8 XCDs
        e["kernel_cycles"] = cyc
        e["lds_idx_active_frac"] = e["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)                     # LDS pipe busy, per CU
        e["valu_issue_frac_at_2_cycles"] = e["SQ_INSTS_VALU"] * 2 / (1024 * cyc)            # 1024 SIMDs, wave64 fp32 op = 2 cycles
    if len(e) > 1 and ("rollout" in short or "dp_reg" in short or "trace_hist" in short or "reward_scan" in short or "epi_fold" in short):
        kernels.append(e)
j = dict(build_id=None, tag=tag, kernels=kernels,
         correction="gfx950: FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced streaming reads -> x2 "
                    "(MI355X_MICROARCH.md, HBM section; calibrated with tools/calib/pmc_calib.hip); WRITE_SIZE exact",
         source=f"rocprofv3 --pmc passes of tools/collect_profiles.sh, per-kernel means in profiles/{tag}_pmc_counters.csv")
sys.path.insert(0, root)
try:
    j["build_id"] = under.get("build_id") and None
    from colosseum_amd import _lib
    j["build_id"] = _lib.source_hash()  # the passes ran on the tree this script runs in (gpurun ships the tree)
    assert j["build_id"][:16] == under["build_id"], (j["build_id"][:16], under["build_id"])
except ImportError:
    pass
json.dump(j, open(f"{prof}/r04_pmc.json", "w"), indent=1)
for e in kernels:
    print(e["kernel"], {k: (round(v, 3) if isinstance(v, float) else v) for k, v in e.items() if k != "kernel"})
