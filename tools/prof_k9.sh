cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/dbg_k9.py benchmark_continuous_ergodic MiniGridEmptyContinuous prms_3 90 5000 30000 | tail -2
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03_k9f_stats --output-format csv -- python3 $R/tools/dbg_k9.py benchmark_continuous_ergodic MiniGridEmptyContinuous prms_3 90 20000 > /dev/null 2>&1
head -12 $R/gpurun_out/r03_k9f_stats/*/*kernel_stats.csv | cut -c1-160
