# rocprofv3 counters of one leak run (run on the GPU box):  bash scripts/profile_leak.sh <tag> [slots]
#   -> gpurun_out/prof_<tag>_{kt,pmc1..3}/ ; condense with  python scripts/summarize_profile.py <tag> r04 --last
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; N=${2:-262144}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_kt -- python3 scripts/leak_one.py $N 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d gpurun_out/prof_${TAG}_pmc1 -- python3 scripts/leak_one.py $N 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prof_${TAG}_pmc2 -- python3 scripts/leak_one.py $N 1 > /dev/null 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_TC_INST_REQ --output-format csv -d gpurun_out/prof_${TAG}_pmc3 -- python3 scripts/leak_one.py $N 1 > /dev/null 2>&1
cat gpurun_out/prof_${TAG}_kt/*/*_kernel_stats.csv
