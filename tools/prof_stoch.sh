#!/bin/bash
# rocprofv3 evidence for the stochastic-dynamics rollout kernels (K1S k_rollout_stoch, K1 k_rollout<0,false,..>):
#   bash tools/prof_stoch.sh TAG        -> gpurun_out/TAG_stoch_*  (tools/summarise_stoch.py TAG copies the summaries)
# One run per purpose; --pmc passes carry --kernel-trace only; the program directly after `--`.
set -e
TAG=${1:-rNN}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
for CASE in "4096 k1s" "4096 k1" "131072 k1" "131072 k1s"; do
  set -- $CASE
  B=$1; K=$2
  P=$OUT/${TAG}_stoch_${K}_B${B}
  ARGS="$R/tools/prof_stoch.py --instances $B --kernel $K --steps 2000 --launches 5"
  python3 $ARGS > ${P}_line.json
  rocprofv3 --kernel-trace --stats -d ${P}_stats --output-format csv -- python3 $ARGS > /dev/null 2> ${P}_stats.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d ${P}_fetch --output-format csv -- python3 $ARGS > /dev/null 2> ${P}_fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d ${P}_write --output-format csv -- python3 $ARGS > /dev/null 2> ${P}_write.err
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
    --kernel-trace -d ${P}_sq1 --output-format csv -- python3 $ARGS > /dev/null 2> ${P}_sq1.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM \
    --kernel-trace -d ${P}_sq2 --output-format csv -- python3 $ARGS > /dev/null 2> ${P}_sq2.err
  echo "$CASE done $(date +%T)" | tee -a $OUT/${TAG}_stoch_progress.txt
done
