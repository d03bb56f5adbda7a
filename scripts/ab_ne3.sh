# A/B of build parameters of the many-energy kernel (run on the GPU box):  bash scripts/ab_ne3.sh "NAME:-Dflags" ...
# every variant: bench_ne.py on xos1 291 energies (1e6 slots), ellip_l9 291 energies sig 5 A (5e5) and xos1 100 energies
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/abs
build() {
  name=$1; shift
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden "$@" -Iinclude -Ipolycap_amd/csrc/hip -c polycap_amd/csrc/hip/pc_kernels.hip -o /tmp/abs/k_$name.o 2>/tmp/abs/k_$name.log &&
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/abs/lib_$name.so polycap_amd/lib/obj/pc_*.c.o /tmp/abs/k_$name.o -ldl -lm -lpthread
}
names=""
n=0
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  [ "$flags" = "$spec" ] && flags=""
  build $name $flags &
  names="$names $name"
  n=$((n+1))
  if [ $((n % 6)) -eq 0 ]; then wait; fi
done
wait
for name in $names; do
  if [ -f /tmp/abs/lib_$name.so ]; then
    for cfg in "xos1 291 1000000 -" "ellip_l9 291 500000 5.0" "xos1 100 1000000 -"; do
      echo -n "$name: " | tee -a gpurun_out/ab_ne3.txt
      POLYCAP_AMD_LIB=/tmp/abs/lib_$name.so timeout -k 10 120 python scripts/bench_ne.py $cfg $OPTS 2>&1 | head -1 | tee -a gpurun_out/ab_ne3.txt
    done
  else
    echo "$name: build failed: $(tail -2 /tmp/abs/k_$name.log)" | tee -a gpurun_out/ab_ne3.txt
  fi
done
