# A/B of build parameters of the leak kernel (run on the GPU box):  bash scripts/ab_leak2.sh "NAME:-Dflags" ...
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/abs
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden $flags -Iinclude -Ipolycap_amd/csrc/hip -c polycap_amd/csrc/hip/pc_kernels.hip -o /tmp/abs/kl_$name.o 2>/tmp/abs/kl_$name.log &&
    hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/abs/libl_$name.so polycap_amd/lib/obj/pc_*.c.o /tmp/abs/kl_$name.o -ldl -lm -lpthread ) &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  echo "== $name"
  POLYCAP_AMD_LIB=/tmp/abs/libl_$name.so timeout -k 10 200 python scripts/bench_leak.py ${SIZES:-262144} 2>&1 | grep -v "^      "
done
