#!/bin/bash
# Config C4 on one GPU: the reference's four default benchmark suites (1 000 instances x 500 000 steps, a log row every 100
# steps), device agents, CSV files written.   bash tools/run_c4.sh TAG [reference|philox] [concurrent groups ...]
# Beta rewards: "reference" = the reference's per-triple caches filled from each MDP's numpy stream (rows equal the
# reference's), "philox" = sampled on the device (distribution-exact).
TAG=${1:-rNN}; shift
MODE=${1:-reference}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
for CG in "${@:-16}"; do
  rm -rf /tmp/c4_$CG
  python3 $R/tools/run_benchmark.py --configs-json $R/tests/golden/G11_benchmark_configs.json \
    --benchmark benchmark_episodic_ergodic --benchmark benchmark_episodic_communicating \
    --benchmark benchmark_continuous_ergodic --benchmark benchmark_continuous_communicating \
    --out /tmp/c4_$CG --concurrent-groups $CG --beta-rewards $MODE > $OUT/${TAG}_c4_${MODE}_cg$CG.log 2>&1
  tail -1 $OUT/${TAG}_c4_${MODE}_cg$CG.log
done
