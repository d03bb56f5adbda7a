# A/B harness for kernel launch options on the GPU box: bash scripts/ab_bench.sh > gpurun_out/ab.log
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1 || true
tail -3 gpurun_out/pytest_gpu.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { echo -n "$*: "; $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); s=d['scheduler']; print('%.4g ph/s  kernel %.2f ms | march %.1f lanes x %.3g steps, event %.1f x %.3g, new %.1f x %.3g'%(d['value'], d['roofline']['kernel_ms'], s['march']['avg_lanes'], s['march']['phases'], s['event']['avg_lanes'], s['event']['phases'], s['new']['avg_lanes'], s['new']['phases']))"; }
run
for et in 8 12 16 20 28; do run --opt event_threshold=$et --opt new_threshold=4; done
run --opt event_threshold=16 --opt new_threshold=8
run --opt event_threshold=12 --opt new_threshold=8
