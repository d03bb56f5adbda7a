# rocprofv3 evidence for one command (run on the GPU box):  bash scripts/profile_r04.sh <tag> <program and arguments>
#   -> gpurun_out/prof_<tag>_{kt,pmc1..4}/ ; condense with  python scripts/summarize_profile.py <tag> r04 [--last]
# The program comes directly after `--` (python3 script: no env / bash -c hop under the profiler).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_kt -- "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d gpurun_out/prof_${TAG}_pmc1 -- "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prof_${TAG}_pmc2 -- "$@" > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${TAG}_pmc3 -- "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${TAG}_pmc4 -- "$@" > /dev/null 2>&1
cat gpurun_out/prof_${TAG}_kt/*/*_kernel_stats.csv
