# Instruction / wait counters of the agent kernels on one continuous batch (90 MiniGridEmpty instances, one lane each):
# how much of a step is ALU (the Beta sampler) and how much is waiting on memory.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $R/gpurun_out/r03_agent_pmc1 --output-format csv -- python3 $R/tools/dbg_k9.py benchmark_continuous_ergodic MiniGridEmptyContinuous prms_3 90 5000 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -d $R/gpurun_out/r03_agent_pmc2 --output-format csv -- python3 $R/tools/dbg_k9.py benchmark_continuous_ergodic MiniGridEmptyContinuous prms_3 90 5000 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("r03_agent_pmc1", "r03_agent_pmc2"):
    for f in glob.glob("$R/gpurun_out/%s/*/*counter_collection.csv" % d):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        for k, v in acc.items():
            if "qlearn" in k or "episodic" in k:
                print(d, k, dict(v))
PY
