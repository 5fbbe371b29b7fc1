#!/bin/bash
# kernel-trace stats of the headline leg only (no dense / VI / CPU legs):  bash tools/prof_k1u.sh TAG [env assignments...]
TAG=${1:-rNN}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu --dense-instances 0 --vi-instances 0 --sustained-seconds 0 --strong-share 0 > $OUT/${TAG}_line.json 2> $OUT/${TAG}.err
head -4 $OUT/${TAG}_stats/*/*kernel_stats.csv | cut -c1-200
python3 -c "import json;l=json.loads(open('$OUT/${TAG}_line.json').read().strip().splitlines()[-1]);print(l['value'], l['ms_per_step'], l['roofline']['lds_plan'])"
