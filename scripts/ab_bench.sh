cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "two_photon" > gpurun_out/pytest_gpu.log 2>&1 || true
tail -15 gpurun_out/pytest_gpu.log
B="timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { echo -n "$*: "; $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); s=d['scheduler']; print('%.4g ph/s  kernel %.2f ms | march %.1f lanes x %.3g steps, event %.1f x %.3g, new %.1f x %.3g'%(d['value'], d['roofline']['kernel_ms'], s['march']['avg_lanes'], s['march']['phases'], s['event']['avg_lanes'], s['event']['phases'], s['new']['avg_lanes'], s['new']['phases']))"; }
run --opt two_photons=0
run
for et in 24 32 40 48; do run --opt event_threshold=$et; done
run --opt event_threshold=32 --opt new_threshold=16
