# PMC counters of a many-energy run (on the GPU box): bash scripts/pmc_ne_r04.sh <tag> <bench_ne.py arguments>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d gpurun_out/pmc_${TAG}_1 -- python3 scripts/bench_ne.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d gpurun_out/pmc_${TAG}_2 -- python3 scripts/bench_ne.py "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_${TAG}_*")):
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "pc_trace" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print(k, "%.4g" % v[-1], "(dispatches %d)" % len(v))
PY
