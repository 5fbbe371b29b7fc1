#!/usr/bin/env python3
"""Copies what tools/collect_profiles.sh left under gpurun_out/TAG_* into profiles/: the two bench lines, the kernel
stats CSV, a per-kernel summary of the FETCH_SIZE / WRITE_SIZE passes and profiles/r01_rollout_pmc.json (what bench.py
reads for roofline.traffic; FETCH_SIZE x2 per the gfx950 correction).   python tools/summarise_profiles.py TAG"""
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


line = json.loads(open(f"{out}/{tag}_bench_line.json").read().strip().splitlines()[-1])
print("bench:", line["value"], line["ms_per_step"], line["roofline"]["frac"], line["roofline"]["launch_ms_avg"],
      "vi", line["vi"]["sweeps_per_s"], line["vi"]["wall_ms"], "cpu", line["cpu_baseline"]["value"],
      line.get("cpu_baseline_all_cores", {}).get("value"))
ks = newest(f"{out}/{tag}_stats/*/*_kernel_stats.csv")
for r in list(csv.DictReader(open(ks)))[:3]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
shutil.copy(ks, f"{prof}/{tag}_kernel_stats.csv")
shutil.copy(f"{out}/{tag}_bench_line.json", f"{prof}/{tag}_bench_line.json")
shutil.copy(f"{out}/{tag}_bench_line_under_rocprof.json", f"{prof}/{tag}_bench_line_under_rocprof.json")
rows, vals, kernel = [], {}, None
for leg, cnt in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(f"{out}/{tag}_{leg}/*/*_counter_collection.csv"))):
        if r["Counter_Name"] == cnt:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        rows.append((f"{tag}_{leg}_size", k, cnt, sum(v) / len(v), len(v)))
        if "rollout" in k:
            vals[cnt] = sum(v) / len(v)
            kernel = k.split("(")[0]
w = csv.writer(open(f"{prof}/{tag}_pmc_fetch_write.csv", "w"))
w.writerow(["run", "kernel", "counter", "avg_value_KB_per_launch", "launches"])
w.writerows(rows)
j = json.load(open(f"{prof}/r01_rollout_pmc.json"))
f, wv = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
j.update(kernel=kernel, FETCH_SIZE_KB_reported=f, WRITE_SIZE_KB_reported=wv, hbm_read_bytes_per_launch=f * 2048,
         hbm_write_bytes_per_launch=wv * 1024, hbm_bytes_per_launch=f * 2048 + wv * 1024,
         source=f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/collect_profiles.sh), profiles/{tag}_pmc_fetch_write.csv")
json.dump(j, open(f"{prof}/r01_rollout_pmc.json", "w"), indent=1)
print("hbm bytes per launch:", j["hbm_bytes_per_launch"])
