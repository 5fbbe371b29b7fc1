#!/usr/bin/env python3
"""Summary of tools/prof_stoch.sh's passes: per (kernel, batch size) the rocprofv3 average kernel time, HBM traffic per
launch (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE, both reported in KiB), SQ counters, and the roofline object
against SURVEY 8(d)'s CSR accounting.   python tools/summarise_stoch.py TAG  ->  profiles/TAG_stoch_rollout_profile.json"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")
HBM_PEAK = 8.0e12


def counters(path, kernel_prefix):
    acc = collections.defaultdict(list)
    f = glob.glob(f"{path}/*/*_counter_collection.csv")
    if not f:
        return {}
    for r in csv.DictReader(open(f[0])):
        if r["Kernel_Name"].replace("void ", "").startswith(kernel_prefix):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


res = []
for line_file in sorted(glob.glob(f"{out}/{tag}_stoch_*_line.json")):
    line = json.loads(open(line_file).read().strip().splitlines()[-1])
    base = line_file[:-len("_line.json")]
    kname = "k_rollout_stoch" if line["kernel"] == "k1s" else "k_rollout<"
    stats = glob.glob(f"{base}_stats/*/*_kernel_stats.csv")
    avg_ns = None
    for r in csv.DictReader(open(stats[0])):
        if r["Name"].replace("void ", "").startswith(kname):
            avg_ns, calls, full = float(r["AverageNs"]), int(r["Calls"]), r["Name"]
    c = {}
    for leg in ("fetch", "write", "sq1", "sq2"):
        c.update(counters(f"{base}_{leg}", kname))
    units = line["instances"] * line["steps_per_launch"]
    alg = line["algorithmic_bytes_per_launch"]
    e = dict(kernel=full.split("(")[0].replace("void ", ""), family=line["family"], instances=line["instances"],
             transitions_per_launch=units, rocprof_avg_ms=avg_ns / 1e6, rocprof_calls=calls,
             wall_ms_per_launch=line["launch_ms"], transitions_per_s=units / (avg_ns / 1e9),
             lds_plan=line["lds_plan"], build_id=line["build_id"])
    traffic = None
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # FETCH_SIZE under-counts wide reads by 2 on gfx950 (MI355X_MICROARCH.md, HBM section; calibrated with
        # tools/calib/pmc_calib.hip: 8 GiB read reports 0.5000, 8 GiB written 1.0000)
        traffic = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        e["pmc"] = dict(FETCH_SIZE_KiB=c["FETCH_SIZE"], WRITE_SIZE_KiB=c["WRITE_SIZE"], traffic_bytes_per_launch=traffic,
                        traffic_bytes_per_transition=traffic / units)
    e["roofline"] = dict(bound="hbm", achieved=alg / (avg_ns / 1e9) / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                         frac=alg / (avg_ns / 1e9) / HBM_PEAK, traffic=traffic,
                         traffic_frac=None if traffic is None else traffic / (avg_ns / 1e9) / HBM_PEAK,
                         algorithmic_bytes_per_transition=line["algorithmic_bytes_per_transition"],
                         accounting="SURVEY 8(d) CSR figure 8 + 8*nnz(s,a) + 28 with the batch's mean row length")
    sq = {k: v for k, v in c.items() if k.startswith(("SQ_", "GRBM"))}
    if sq:
        e["sq"] = sq
        if "SQ_INSTS_VALU" in sq:
            e["sq_per_transition"] = {k: sq[k] / units for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM") if k in sq}
        if sq.get("SQ_WAVE_CYCLES"):
            e["wave_cycle_shares"] = {k: sq[k] / sq["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS") if k in sq}
        if sq.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = sq.get("SQ_LDS_BANK_CONFLICT", 0) / sq["SQ_LDS_IDX_ACTIVE"]
    res.append(e)
    print(e["kernel"], e["instances"], f"{e['rocprof_avg_ms']:.3f} ms", f"{e['transitions_per_s']:.3g}/s", "frac", f"{e['roofline']['frac']:.3f}",
          "traffic B/transition", None if traffic is None else round(traffic / units, 1))
json.dump(res, open(os.path.join(root, "profiles", f"{tag}_stoch_rollout_profile.json"), "w"), indent=1)
