# run-time options of the pool kernel around their defaults (run on the GPU box)
cd $GRAFT_REPO_ROOT
for o in "pool_refill=20" "pool_refill=12" "pool_refill=28" "pool_march_min=8" "pool_march_min=24" "pool_event_min=40" "pool_event_min=56" "pool_new_min=32" "pool_new_min=60"; do
  echo -n "$o: "; timeout 100 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --opt $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['scheduler']
print('%.2f ms march %.3g x %.1f event %.3g x %.1f new %.3g x %.1f' % (d['roofline']['kernel_ms'], s['march']['phases'], s['march']['avg_lanes'], s['event']['phases'], s['event']['avg_lanes'], s['new']['phases'], s['new']['avg_lanes']))"
done
