#!/bin/bash
# Config C4 on one GPU: the reference's four default benchmark suites (1 000 instances x 500 000 steps, a log row every 100
# steps), device agents, CSV files written.   bash tools/run_c4.sh TAG [reference|philox] [concurrent groups ...]
# Beta rewards: "reference" = the reference's per-triple caches filled from each MDP's numpy stream (rows equal the
# reference's), "philox" = sampled on the device (distribution-exact).
# The last line printed is the host-CPU accounting of the run's cgroup (the GPU boxes give a job 16 cores' worth of quota:
# `nr_throttled` counts the 100 ms periods in which the job ran out of it and every thread was stopped).
TAG=${1:-rNN}; shift
MODE=${1:-reference}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cpustat() { python3 - <<'PY'
try:
    d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
    print(d.get("usage_usec", 0), d.get("nr_periods", 0), d.get("nr_throttled", 0), d.get("throttled_usec", 0))
except OSError:
    print("0 0 0 0")
PY
}
for CG in "${@:-16}"; do
  rm -rf /tmp/c4_$CG
  S0=($(cpustat))
  python3 $R/tools/run_benchmark.py --configs-json $R/tests/golden/G11_benchmark_configs.json \
    --benchmark benchmark_episodic_ergodic --benchmark benchmark_episodic_communicating \
    --benchmark benchmark_continuous_ergodic --benchmark benchmark_continuous_communicating \
    --out /tmp/c4_$CG --concurrent-groups $CG --beta-rewards $MODE > $OUT/${TAG}_c4_${MODE}_cg$CG.log 2>&1
  S1=($(cpustat))
  echo "host cpu: $(( (S1[0]-S0[0])/1000 )) ms used, $(( S1[2]-S0[2] )) of $(( S1[1]-S0[1] )) periods throttled, $(( (S1[3]-S0[3])/1000 )) ms throttled (threads x time)" >> $OUT/${TAG}_c4_${MODE}_cg$CG.log
  tail -2 $OUT/${TAG}_c4_${MODE}_cg$CG.log
done
