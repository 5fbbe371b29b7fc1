#!/usr/bin/env python3
"""Per-kernel summary of tools/prof_sq.sh's passes: average duration (rocprofv3 --stats), SQ counters per launch and the
hardware-anchored fractions derived from them (VALU issue at 2 cycles per wave64 instruction on 1 024 SIMDs; LDS pipe =
SQ_LDS_IDX_ACTIVE per CU-cycle; share of wave-cycles parked at s_waitcnt / barriers; HBM bytes when PROF_HBM was set:
FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE).  Writes gpurun_out/TAG_summary.json as well."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out")


def newest(pat):
    g = glob.glob(pat)
    return max(g, key=os.path.getmtime) if g else None


def counters(leg):
    f = newest(f"{out}/{tag}_{leg}/*/*_counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


stats = {}
f = newest(f"{out}/{tag}_stats/*/*_kernel_stats.csv")
if f:
    for r in csv.DictReader(open(f)):
        stats[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]), float(r["Percentage"]))
legs = {leg: counters(leg) for leg in ("sq1", "sq2", "fetch", "write")}
res = []
for k in sorted(set(legs["sq1"]) | set(legs["sq2"]), key=lambda k: -stats.get(k, (0, 0, 0))[2]):
    e = {"kernel": k.split("(")[0].replace("void ", "")}
    if k in stats:
        e["calls"], e["avg_us"], e["pct"] = stats[k][0], round(stats[k][1] / 1e3, 2), stats[k][2]
    for leg in legs.values():
        e.update(leg.get(k, {}))
    if "GRBM_GUI_ACTIVE" in e:
        cyc = e["GRBM_GUI_ACTIVE"] / 8  # rocprofv3 sums the 8 XCDs
        e["kernel_cycles"] = cyc
        if "SQ_LDS_IDX_ACTIVE" in e:
            e["lds_pipe_frac"] = e["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)
            e["lds_bank_conflict_frac"] = e["SQ_LDS_BANK_CONFLICT"] / max(e["SQ_LDS_IDX_ACTIVE"], 1)
        if "SQ_INSTS_VALU" in e:
            e["valu_issue_frac_at_2_cycles"] = e["SQ_INSTS_VALU"] * 2 / (1024 * cyc)
    if "SQ_WAVE_CYCLES" in e:
        e["wave_cycles_waiting_frac"] = e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"]
        e["wave_cycles_issue_stalled_frac"] = e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_read_bytes"] = e["FETCH_SIZE"] * 2048
        e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
    res.append(e)
json.dump(res, open(f"{out}/{tag}_summary.json", "w"), indent=1)
for e in res[:8]:
    print(e["kernel"][:60])
    print("   ", {k: (round(v, 4) if isinstance(v, float) and v < 100 else (int(v) if isinstance(v, float) else v)) for k, v in e.items() if k != "kernel"})
