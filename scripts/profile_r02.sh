# Collects the round's rocprofv3 evidence for bench.py's dominant kernel (run on the GPU box):
#   bash scripts/profile_r02.sh <tag>     -> gpurun_out/prof_<tag>_*/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-v}
[ -n "$QUICK" ] || python bench.py $BENCH_OPTS > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_kt -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras $BENCH_OPTS > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d gpurun_out/prof_${TAG}_pmc1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $BENCH_OPTS > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prof_${TAG}_pmc2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $BENCH_OPTS > /dev/null 2>&1
[ -n "$QUICK" ] || rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${TAG}_pmc3 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $BENCH_OPTS > /dev/null 2>&1
[ -n "$QUICK" ] || rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${TAG}_pmc4 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $BENCH_OPTS > /dev/null 2>&1
[ -n "$QUICK" ] || tail -1 gpurun_out/bench_${TAG}.json
cat gpurun_out/prof_${TAG}_kt/*/*_kernel_stats.csv
