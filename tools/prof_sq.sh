#!/bin/bash
# SQ counter passes (two sets of 8) + kernel stats of ONE python tool invocation, summarised per kernel:
#   bash tools/prof_sq.sh TAG tools/time_rollout.py [args...]     (GPU box; outputs under gpurun_out/TAG_*)
# --kernel-trace is the only option next to --pmc and the program comes directly after `--` (gpurun's rules).
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats --output-format csv -- python3 $R/$@ > $OUT/${TAG}_stats.out 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --kernel-trace -d $OUT/${TAG}_sq1 --output-format csv -- python3 $R/$@ > $OUT/${TAG}_sq1.out 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM \
  --kernel-trace -d $OUT/${TAG}_sq2 --output-format csv -- python3 $R/$@ > $OUT/${TAG}_sq2.out 2>&1 || exit 1
if [ -n "$PROF_HBM" ]; then
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_fetch --output-format csv -- python3 $R/$@ > $OUT/${TAG}_fetch.out 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_write --output-format csv -- python3 $R/$@ > $OUT/${TAG}_write.out 2>&1 || exit 1
fi
python3 $R/tools/summarise_sq.py $TAG > $OUT/${TAG}_summary.txt 2>&1
cat $OUT/${TAG}_summary.txt
