#!/bin/bash
# HBM reads (FETCH_SIZE) and writes of the C5 diameter under K5C (CMDP_K5C=cl) and K5S (CMDP_K5C=0)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
for cl in "$@"; do
  export CMDP_K5C=$cl
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/r03_k5c_fetch_$cl --output-format csv -- python3 $R/tools/run_c5.py > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/r03_k5c_write_$cl --output-format csv -- python3 $R/tools/run_c5.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob
for what in ("fetch", "write"):
    tot = {}
    for f in glob.glob("$OUT/r03_k5c_%s_$cl/*/*counter_collection.csv" % what):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "diam" in k:
                tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
    for k, v in tot.items():
        # FETCH_SIZE / WRITE_SIZE in KiB; reported x2 correction for FETCH on gfx950 (tools/calib)
        print("CMDP_K5C=$cl", what, k, "raw %.4g" % v, "=> %.3f TB" % (v * 1024 * (2 if what == "fetch" else 1) / 1e12))
PY
done
