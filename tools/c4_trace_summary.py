#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace of a C4 run (the trace itself is too large to keep): per kernel name calls / total /
mean duration, the wall span, the union of busy intervals and the time-weighted mean number of kernels in flight.
    python tools/c4_trace_summary.py <kernel_trace.csv> <out.json>"""
import csv
import json
import sys
from collections import defaultdict

rows = []
per = defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.append((s, e))
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    per[k][0] += 1
    per[k][1] += e - s
ev = sorted([(s, 1) for s, _ in rows] + [(e, -1) for _, e in rows])
t0, t1 = ev[0][0], ev[-1][0]
busy = 0
weighted = 0
depth = 0
prev = t0
hist = defaultdict(int)
for t, d in ev:
    if depth > 0:
        busy += t - prev
    weighted += depth * (t - prev)
    hist[min(depth, 32)] += t - prev
    depth += d
    prev = t
out = dict(kernels=len(rows), span_s=(t1 - t0) / 1e9, busy_union_s=busy / 1e9, mean_kernels_in_flight=weighted / max(1, t1 - t0),
           time_share_by_kernels_in_flight={str(k): v / max(1, t1 - t0) for k, v in sorted(hist.items())},
           per_kernel={k: dict(calls=c, total_s=t / 1e9, mean_us=t / c / 1e3) for k, (c, t) in sorted(per.items(), key=lambda x: -x[1][1])[:14]})
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
