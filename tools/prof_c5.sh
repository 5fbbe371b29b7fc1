#!/bin/bash
# Config C5 evidence on the GPU box:  bash tools/prof_c5.sh TAG
# kernel stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes (they do not fit one pass on gfx950), the program
# directly after `--`.  tools/summarise_c5.py TAG turns gpurun_out/TAG_c5_* into profiles/r02_c5_diameter_pmc.json.
set -e
TAG=${1:-rNN}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
python3 $R/tools/run_c5.py --check 3 > $OUT/${TAG}_c5_line.json 2> $OUT/${TAG}_c5.err
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_c5_stats --output-format csv -- python3 $R/tools/run_c5.py > $OUT/${TAG}_c5_line_under_rocprof.json 2>> $OUT/${TAG}_c5.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_c5_fetch --output-format csv -- python3 $R/tools/run_c5.py > /dev/null 2>> $OUT/${TAG}_c5.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_c5_write --output-format csv -- python3 $R/tools/run_c5.py > /dev/null 2>> $OUT/${TAG}_c5.err
echo done > $OUT/${TAG}_c5_progress.txt
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-trace -d $OUT/${TAG}_c5_sq --output-format csv -- python3 $R/tools/run_c5.py > /dev/null 2>> $OUT/${TAG}_c5.err
echo "sq done" >> $OUT/${TAG}_c5_progress.txt
