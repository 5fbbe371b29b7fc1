"""python tools/summarise_c5.py TAG: profiles/r03_c5_diameter_pmc.json + kernel stats from gpurun_out/TAG_c5_* (tools/prof_c5.sh)."""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, prof = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def counter(kind, name):
    tot, kern = 0.0, None
    for f in glob.glob(f"{out}/{tag}_c5_{kind}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(("void k_diam_lanes", "void k_diam_cluster")) and r["Counter_Name"] == name:
                tot += float(r["Counter_Value"])
                kern = r["Kernel_Name"]
    return tot, kern


fetch_kb, kern = counter("fetch", "FETCH_SIZE")
write_kb, _ = counter("write", "WRITE_SIZE")
stats = glob.glob(f"{out}/{tag}_c5_stats/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(stats, f"{prof}/r03_c5_diameter_kernel_stats.csv")
kernel_ns = sum(float(r["TotalDurationNs"]) for r in csv.DictReader(open(stats)) if r["Name"].startswith(("void k_diam_lanes", "void k_diam_cluster")))
run = json.loads(open(f"{out}/{tag}_c5_line.json").read().strip().splitlines()[-1])
rd, wr = fetch_kb * 1024 * 2, write_kb * 1024
alg = 2 * wr
j = dict(kernel=kern, workload="C5: MiniGridRoomsContinuous(seed=0,room_size=28,n_rooms=16,n_starting_states=2,p_lazy=0.1), all %d targets, eps 1e-3" % run["n_states"],
         FETCH_SIZE_KB_reported=fetch_kb, WRITE_SIZE_KB_reported=write_kb,
         correction="gfx950: FETCH_SIZE reports 1/2 of the bytes of coalesced dword-per-lane reads (tools/calib/pmc_calib.hip) -> x2; WRITE_SIZE exact",
         hbm_read_bytes=rd, hbm_write_bytes=wr, kernel_s=kernel_ns * 1e-9,
         algorithmic_bytes="8 B per state per (target, sweep) = 2 x the bytes written", algorithmic_bytes_total=alg,
         achieved_algorithmic_GBps=alg / (kernel_ns * 1e-9) / 1e9, frac_of_8TBps=alg / (kernel_ns * 1e-9) / 8e12,
         traffic_GBps=(rd + wr) / (kernel_ns * 1e-9) / 1e9, read_amplification=rd / wr, run=run)
sq = {}
for name in ("SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"):
    v, _ = counter("sq", name)
    if v:
        sq[name] = v
if sq:
    cyc = sq["GRBM_GUI_ACTIVE"] / 8   # rocprofv3 sums the 8 XCDs
    j["sq"] = dict(sq, kernel_cycles=cyc, valu_issue_frac_at_2_cycles=sq["SQ_INSTS_VALU"] * 2 / (1024 * cyc),
                   wave_cycles_waiting_frac=sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
                   wave_cycles_issue_stalled_frac=sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
                   valu_insts_per_state_and_group_sweep=sq["SQ_INSTS_VALU"] / (wr / 256.0),
                   salu_insts_per_state_and_group_sweep=sq["SQ_INSTS_SALU"] / (wr / 256.0))
json.dump(j, open(f"{prof}/r03_c5_diameter_pmc.json", "w"), indent=1)
print({k: v for k, v in j.items() if k != "run"})
