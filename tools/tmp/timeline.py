import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "k_rollout_epi" in r["Kernel_Name"] or "k_reward_scan" in r["Kernel_Name"]]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = {}
for r in rows[-16:]:
    n = "epi " if "epi" in r["Kernel_Name"] else "scan"
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%s start %10.1f us  end %10.1f us  dur %7.1f us  gap since same-kind end %6.1f us" % (n, s / 1e3, e / 1e3, (e - s) / 1e3, (s - prev_end.get(n, s)) / 1e3))
    prev_end[n] = e
