# A/B of the leak kernel: variants of libpolycap built with extra -D flags.
#   here (no GPU):  bash scripts/ab_leak.sh build <name> "<flags>"     -> build_ab/libpolycap_<name>.so (travels with gpurun)
#   GPU box:        bash scripts/ab_leak.sh run "<name> ..." [slots] [leak_one.py options, e.g. leak_stack_mb=16384]
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
if [ "$1" = build ]; then
  mkdir -p build_ab
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden $3 -Iinclude -Ipolycap_amd/csrc/hip -c polycap_amd/csrc/hip/pc_kernels.hip -o build_ab/k_$2.o 2>/dev/null \
   && hipcc -shared -fPIC --offload-arch=gfx950 -o build_ab/libpolycap_$2.so polycap_amd/lib/obj/pc_*.c.o build_ab/k_$2.o -ldl -lm -lpthread && rm build_ab/k_$2.o && echo built build_ab/libpolycap_$2.so
else
  N=${3:-262144}; shift; NAMES=$1; shift; shift
  for h in $NAMES; do
    echo "== $h $@"
    POLYCAP_AMD_LIB=$PWD/build_ab/libpolycap_$h.so POLYCAP_LEAK_TIMING=1 timeout -k 10 200 python scripts/leak_one.py $N 1 "$@" 2>&1
  done
fi
