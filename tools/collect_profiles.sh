#!/bin/bash
# Collects the evidence bench.py's roofline object cites, on the GPU box:  bash tools/collect_profiles.sh TAG
# (kernel stats and the two PMC passes each need their own rocprofv3 run; outputs under gpurun_out/TAG_*)
set -e
TAG=${1:-rNN}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
python3 $R/bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
echo "bench done" > $OUT/${TAG}_progress.txt
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats --output-format csv -- python3 $R/bench.py --no-cpu > $OUT/${TAG}_bench_line_under_rocprof.json 2> $OUT/${TAG}_rocprof.err
echo "stats done" >> $OUT/${TAG}_progress.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_fetch --output-format csv -- python3 $R/bench.py --no-cpu --vi-instances 0 --steps 10 > /dev/null 2> $OUT/${TAG}_fetch.err
echo "fetch done" >> $OUT/${TAG}_progress.txt
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_write --output-format csv -- python3 $R/bench.py --no-cpu --vi-instances 0 --steps 10 > /dev/null 2> $OUT/${TAG}_write.err
echo "write done" >> $OUT/${TAG}_progress.txt
