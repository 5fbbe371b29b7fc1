#!/bin/bash
# Config C4 on one GPU: the reference's four default benchmark suites (1 000 instances x 500 000 steps, a log row every 100
# steps), device agents, CSV files written.   bash tools/run_c4.sh TAG [concurrent groups ...]
TAG=${1:-rNN}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
for CG in "${@:-6}"; do
  rm -rf /tmp/c4_$CG
  python3 $R/tools/run_benchmark.py --configs-json $R/tests/golden/G11_benchmark_configs.json \
    --benchmark benchmark_episodic_ergodic --benchmark benchmark_episodic_communicating \
    --benchmark benchmark_continuous_ergodic --benchmark benchmark_continuous_communicating \
    --out /tmp/c4_$CG --concurrent-groups $CG > $OUT/${TAG}_c4_cg$CG.log 2>&1
  tail -1 $OUT/${TAG}_c4_cg$CG.log
done
