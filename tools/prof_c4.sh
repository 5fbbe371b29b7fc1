#!/bin/bash
# Where does a C4 device batch spend its time?  kernel stats of four representative batches at the reference's logging
# cadence, 100 000 steps each:  bash tools/prof_c4.sh TAG
set -e
TAG=${1:-rNN}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
python3 $R/tools/prof_agents.py 100000 100 > $OUT/${TAG}_c4prof_plain.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_c4prof --output-format csv -- python3 $R/tools/prof_agents.py 100000 100 > $OUT/${TAG}_c4prof_rocprof.txt 2>&1
