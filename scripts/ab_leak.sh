# A/B of the leak kernel for values of PC_WALL_HOPS (run on the GPU box): bash scripts/ab_leak.sh "1 2 4"
cd $GRAFT_REPO_ROOT
for h in $1; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DPC_WALL_HOPS=$h -Iinclude -Ipolycap_amd/csrc/hip -c polycap_amd/csrc/hip/pc_kernels.hip -o /tmp/k_$h.o 2>/dev/null
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libpolycap_$h.so polycap_amd/lib/obj/pc_*.c.o /tmp/k_$h.o -ldl -lm
  echo "== PC_WALL_HOPS=$h"
  POLYCAP_AMD_LIB=/tmp/libpolycap_$h.so timeout -k 10 200 python scripts/bench_leak.py 262144 2>&1 | grep -v lanes
done
