# A/B of build parameters of the headline kernel (run on the GPU box):  bash scripts/ab_build.sh "NAME:-Dflags" ...
# every variant: kernel ms of the 1e7-slot launch (scripts/analysis/opt_scan.py, min of 4) and a check of the totals
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
  build $name $flags &
  names="$names $name"
  n=$((n+1))
  if [ $((n % 6)) -eq 0 ]; then wait; fi
done
wait
for name in $names; do
  if [ -f /tmp/abs/lib_$name.so ]; then
    echo -n "$name: " | tee -a gpurun_out/ab_build.txt
    POLYCAP_AMD_LIB=/tmp/abs/lib_$name.so timeout -k 10 120 python scripts/analysis/opt_scan.py ${OPT:-event_march} ${OPTVALS:-0} 2>&1 | tail -${TAILN:-1} | tee -a gpurun_out/ab_build.txt
  else
    echo "$name: build failed: $(tail -2 /tmp/abs/k_$name.log)" | tee -a gpurun_out/ab_build.txt
  fi
done
