cd $GRAFT_REPO_ROOT
B="timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { echo -n "$POLYCAP_AMD_LIB $*: "; $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); s=d['scheduler']; print('%.4g ph/s  kernel %.2f ms | march %.1f lanes x %.3g steps, event %.1f x %.3g, new %.1f x %.3g'%(d['value'], d['roofline']['kernel_ms'], s['march']['avg_lanes'], s['march']['phases'], s['event']['avg_lanes'], s['event']['phases'], s['new']['avg_lanes'], s['new']['phases']))"; }
run
for u in 1 2 3; do export POLYCAP_AMD_LIB=$GRAFT_REPO_ROOT/polycap_amd/lib/libpolycap_u$u.so; run; run --opt event_threshold=24; done
