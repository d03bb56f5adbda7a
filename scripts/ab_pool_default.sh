# would the pool kernel pay as the default single-energy kernel?  (run on the GPU box)
cd $GRAFT_REPO_ROOT
for o in "" "--opt pool=1"; do
  echo -n "bench $o: "; timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.4g ph/s kernel %.2f ms' % (d['value'], d['roofline']['kernel_ms']))"
done
for d in xos1 ellip_l9 cone; do for o in "pool=0" "pool=1"; do
  timeout -k 10 200 python scripts/bench_ne.py $d 1 4000000 5.0 $o 2>&1 | grep -o "^[a-z_0-9]* n_E=.*started photons/s" | sed 's/sig=5.0//; s/: .*slots,//'
done; done
