# A/B of the scan-march build parameters of the pool kernel (run on the GPU box): variants are compiled there, four at
# a time, and each runs the headline bench without its side legs.   bash scripts/ab_scan.sh "NAME:-Dflags" ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab_scan /tmp/abs
build() {
  name=$1; shift
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden "$@" -Iinclude -Ipolycap_amd/csrc/hip -c polycap_amd/csrc/hip/pc_kernels.hip -o /tmp/abs/k_$name.o 2>/tmp/abs/k_$name.log &&
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/abs/lib_$name.so polycap_amd/lib/obj/pc_*.c.o /tmp/abs/k_$name.o -ldl -lm -lpthread
}
names=""
n=0
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  build $name $flags &
  names="$names $name"
  n=$((n+1))
  if [ $((n % 4)) -eq 0 ]; then wait; fi
done
wait
for name in $names; do
  if [ -f /tmp/abs/lib_$name.so ]; then
    POLYCAP_AMD_LIB=/tmp/abs/lib_$name.so timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-extras --no-cpu-baseline $BENCH_OPTS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['scheduler']
print('$name: %.2f ms  %.4g started/s  march %.3g steps x %.1f lanes, event %.3g x %.1f, new %.3g x %.1f' % (d['roofline']['kernel_ms'], d['value'], s['march']['phases'], s['march']['avg_lanes'], s['event']['phases'], s['event']['avg_lanes'], s['new']['phases'], s['new']['avg_lanes']))" | tee -a gpurun_out/ab_scan/results.txt
  else
    echo "$name: build failed: $(tail -2 /tmp/abs/k_$name.log)" | tee -a gpurun_out/ab_scan/results.txt
  fi
done
