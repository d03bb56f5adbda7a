# A/B harness for kernel launch options on the GPU box: bash scripts/ab_bench.sh > gpurun_out/ab.log
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { echo -n "$*: "; $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); s=d['scheduler']; print('%.4g ph/s  kernel %.2f ms | march %.1f lanes x %.3g steps, event %.1f x %.3g, new %.1f x %.3g'%(d['value'], d['roofline']['kernel_ms'], s['march']['avg_lanes'], s['march']['phases'], s['event']['avg_lanes'], s['event']['phases'], s['new']['avg_lanes'], s['new']['phases']))"; }
run
run --opt event_threshold=24 --opt new_threshold=4
run --opt event_threshold=16 --opt new_threshold=4
run --opt event_threshold=8 --opt new_threshold=2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d gpurun_out/prof_pmc1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prof_pmc4 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_pmc4.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_pmc3 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_pmc3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_pmc2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_pmc2.log 2>&1
