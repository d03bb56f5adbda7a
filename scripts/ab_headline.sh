# A/B of the headline kernel: variants of libpolycap built with extra -D flags.
#   here (no GPU):  bash scripts/ab_headline.sh build <name> "<flags>"   -> build_ab/libpolycap_<name>.so (travels with gpurun)
#   GPU box:        bash scripts/ab_headline.sh run "<name> ..."         (each twice, interleaved; scripts/analysis/headline_ab.py)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
if [ "$1" = build ]; then
  mkdir -p build_ab
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden $3 -Iinclude -Ipolycap_amd/csrc/hip -c polycap_amd/csrc/hip/pc_kernels.hip -o build_ab/k_$2.o 2>/dev/null \
   && hipcc -shared -fPIC --offload-arch=gfx950 -o build_ab/libpolycap_$2.so polycap_amd/lib/obj/pc_*.c.o build_ab/k_$2.o -ldl -lm -lpthread && rm build_ab/k_$2.o && echo built build_ab/libpolycap_$2.so
else
  for rep in 1 2; do
    for h in $2; do
      echo "== $h"
      POLYCAP_AMD_LIB=$PWD/build_ab/libpolycap_$h.so timeout -k 10 200 python scripts/analysis/headline_ab.py march_stats=0 2>&1 | tail -1
    done
  done
fi
