# march unroll 4 vs 8 on the other energy counts (run on the GPU box)
cd $GRAFT_REPO_ROOT
for u in 4 8; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DPC_MARCH_UNROLL=$u -Iinclude -Ipolycap_amd/csrc/hip -c polycap_amd/csrc/hip/pc_kernels.hip -o /tmp/ku_$u.o 2>/dev/null
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libpolycap_u$u.so polycap_amd/lib/obj/pc_*.c.o /tmp/ku_$u.o -ldl -lm -lpthread
  echo "== PC_MARCH_UNROLL=$u"
  for ne in 4 8 16 30 291; do POLYCAP_AMD_LIB=/tmp/libpolycap_u$u.so timeout -k 10 200 python scripts/bench_ne.py xos1 $ne 4000000 2>&1 | grep -o "n_E=.*started photons/s"; done
  POLYCAP_AMD_LIB=/tmp/libpolycap_u$u.so timeout -k 10 200 python scripts/bench_ne.py ellip_l9 1 4000000 5.0 2>&1 | grep -o "n_E=.*started photons/s"
  POLYCAP_AMD_LIB=/tmp/libpolycap_u$u.so timeout -k 10 200 python scripts/bench_leak.py 2>&1 | tail -3
done
