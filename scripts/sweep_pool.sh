# options of the pool kernel on the headline workload (run on the GPU box)
cd $GRAFT_REPO_ROOT
run() { echo -n "$*: "; timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --opt pool=1 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); s=d['scheduler']; print('%.4g ph/s kernel %.2f ms | event %.1f lanes x %.3g, new %.1f x %.3g' % (d['value'], d['roofline']['kernel_ms'], s['event']['avg_lanes'], s['event']['phases'], s['new']['avg_lanes'], s['new']['phases']))"; }
run
for r in 8 16 28; do run --opt pool_refill=$r; done
for m in 8 24 32; do run --opt pool_march_min=$m; done
for e in 32 40 56 64; do run --opt pool_event_min=$e; done
for n in 16 32 64; do run --opt pool_new_min=$n; done
for b in 4 8 32; do run --opt march_burst=$b; done
