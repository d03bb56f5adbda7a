# A/B of two builds of libpolycap.so on the headline workload (run on the GPU box): the in-tree library against
# polycap_amd/lib/libpolycap_prev.so, plus option sweeps passed as arguments ("--opt event_threshold=24" ...)
cd $GRAFT_REPO_ROOT
B="timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
run() { echo -n "${POLYCAP_AMD_LIB:-in-tree} $*: "; $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); s=d['scheduler']; print('%.4g ph/s  kernel %.2f ms | march %.1f lanes x %.3g steps, event %.1f x %.3g, new %.1f x %.3g'%(d['value'], d['roofline']['kernel_ms'], s['march']['avg_lanes'], s['march']['phases'], s['event']['avg_lanes'], s['event']['phases'], s['new']['avg_lanes'], s['new']['phases']))"; }
run
run --opt event_threshold=16
run --opt event_threshold=24
run --opt event_threshold=28 --opt march_burst=8
run --opt pool=1
POLYCAP_AMD_LIB=$GRAFT_REPO_ROOT/polycap_amd/lib/libpolycap_prev.so run
