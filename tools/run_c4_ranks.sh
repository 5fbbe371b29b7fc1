#!/bin/bash
# Config C4 with several PROCESSES sharing one GPU (rehearsal of the multi-rank path; also measures how much of the
# one-process time is contention inside one HIP runtime):  bash tools/run_c4_ranks.sh TAG RANKS CONCURRENT_GROUPS
TAG=${1:-rNN}; NP=${2:-4}; CG=${3:-4}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
rm -rf /tmp/c4_np$NP
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $NP --master-addr 127.0.0.1 --master-port 29533 \
  $R/tools/run_benchmark.py --configs-json $R/tests/golden/G11_benchmark_configs.json \
  --benchmark benchmark_episodic_ergodic --benchmark benchmark_episodic_communicating \
  --benchmark benchmark_continuous_ergodic --benchmark benchmark_continuous_communicating \
  --out /tmp/c4_np$NP --concurrent-groups $CG --share-gpu --dist-backend gloo > $OUT/${TAG}_c4_np${NP}_cg$CG.log 2>&1
tail -2 $OUT/${TAG}_c4_np${NP}_cg$CG.log
