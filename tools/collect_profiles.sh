#!/bin/bash
# Collects the evidence bench.py's roofline objects cite, on the GPU box:  bash tools/collect_profiles.sh TAG
# One rocprofv3 run per purpose (kernel stats; FETCH_SIZE; WRITE_SIZE; two SQ counter sets -- 8 SQ slots and 4 TCC slots
# per pass on gfx950, FETCH_SIZE and WRITE_SIZE do not fit together), --kernel-trace only next to --pmc, the program
# directly after `--`.  Outputs under gpurun_out/TAG_*; tools/summarise_profiles.py TAG copies the summaries to profiles/.
set -e
TAG=${1:-rNN}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
echo "start $(date +%T)"
python3 $R/bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err
echo "bench done" | tee -a $OUT/${TAG}_progress.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats --output-format csv -- python3 $R/bench.py --no-cpu --no-live-pmc --sustained-seconds 0 > $OUT/${TAG}_bench_line_under_rocprof.json 2> $OUT/${TAG}_rocprof.err
echo "stats done" | tee -a $OUT/${TAG}_progress.txt
PROF_ARGS="--no-cpu --no-live-pmc --steps 6 --warmup 2 --dense-steps 4 --sustained-seconds 0 --strong-share 0"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_fetch --output-format csv -- python3 $R/bench.py $PROF_ARGS > /dev/null 2> $OUT/${TAG}_fetch.err
echo "fetch done" | tee -a $OUT/${TAG}_progress.txt
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_write --output-format csv -- python3 $R/bench.py $PROF_ARGS > /dev/null 2> $OUT/${TAG}_write.err
echo "write done" | tee -a $OUT/${TAG}_progress.txt
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --kernel-trace -d $OUT/${TAG}_sq1 --output-format csv -- python3 $R/bench.py $PROF_ARGS > $OUT/${TAG}_sq1.out 2> $OUT/${TAG}_sq1.err
echo "sq1 done" | tee -a $OUT/${TAG}_progress.txt
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM \
  --kernel-trace -d $OUT/${TAG}_sq2 --output-format csv -- python3 $R/bench.py $PROF_ARGS > $OUT/${TAG}_sq2.out 2> $OUT/${TAG}_sq2.err
echo "sq2 done" | tee -a $OUT/${TAG}_progress.txt
