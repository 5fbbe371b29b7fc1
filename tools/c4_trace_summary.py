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
streams = defaultdict(list)   # (thread, stream) -> [(start, end, kernel)]: one device batch each (one host thread, one stream)
for r in csv.DictReader(open(sys.argv[1])):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.append((s, e))
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    per[k][0] += 1
    per[k][1] += e - s
    streams[(r.get("Thread_Id"), r.get("Stream_Id"), r.get("Queue_Id"))].append((s, e, k))
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
# the streams that ran longest: how much of their span a kernel of theirs was executing, and the mean duration of each
# kernel ON THAT STREAM (to set beside the same batch run alone)
longest = []
for key, ks in streams.items():
    ks.sort()
    span = ks[-1][1] - ks[0][0]
    busy_s = sum(e - s for s, e, _ in ks)
    pk = defaultdict(lambda: [0, 0])
    for s, e, k in ks:
        pk[k][0] += 1
        pk[k][1] += e - s
    longest.append(dict(thread=key[0], stream=key[1], queue=key[2], kernels=len(ks), span_s=span / 1e9, kernel_s=busy_s / 1e9,
                        idle_share=1 - busy_s / max(1, span),
                        per_kernel={k: dict(calls=c, mean_us=t / c / 1e3, total_s=t / 1e9) for k, (c, t) in sorted(pk.items(), key=lambda x: -x[1][1])[:8]}))
longest.sort(key=lambda d: -d["span_s"])
out["queues_in_use"] = len({k[2] for k in streams})
out["longest_streams"] = longest[:6]
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
