cd $GRAFT_REPO_ROOT
for et in 12 16 20 24 28; do for mb in 4 8 16; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --opt event_threshold=$et --opt march_burst=$mb 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('et=$et mb=$mb kernel_ms %.2f' % d['roofline']['kernel_ms'])"
done; done
for ne in 4 8 16 291; do timeout -k 10 200 python scripts/bench_ne.py xos1 $ne 2000000 2>&1 | grep -o "n_E=.*started photons/s"; done
