#!/bin/bash
# kernel trace of a FULL C4 run, summarised on the box (the trace is ~1e6 kernels):  bash tools/prof_c4_full.sh TAG
TAG=${1:-rNN}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/c4_prof /tmp/c4_trace
rocprofv3 --kernel-trace -d /tmp/c4_trace --output-format csv -- python3 $R/tools/run_benchmark.py --configs-json $R/tests/golden/G11_benchmark_configs.json \
  --benchmark benchmark_episodic_ergodic --benchmark benchmark_episodic_communicating --benchmark benchmark_continuous_ergodic --benchmark benchmark_continuous_communicating \
  --out /tmp/c4_prof --beta-rewards philox > $R/gpurun_out/${TAG}_c4_under_rocprof.log 2>&1
tail -1 $R/gpurun_out/${TAG}_c4_under_rocprof.log
python3 $R/tools/c4_trace_summary.py /tmp/c4_trace/*/*_kernel_trace.csv $R/gpurun_out/${TAG}_c4_trace_summary.json
