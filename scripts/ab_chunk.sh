# A/B of the slot chunk a wave takes from the global counter (tail of a launch); run on the GPU box
cd $GRAFT_REPO_ROOT
for c in 128 64 32 16 8; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DPC_CHUNK=$c -Iinclude -Ipolycap_amd/csrc/hip -c polycap_amd/csrc/hip/pc_kernels.hip -o /tmp/kc_$c.o 2>/dev/null
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libpolycap_c$c.so polycap_amd/lib/obj/pc_*.c.o /tmp/kc_$c.o -ldl -lm -lpthread
  echo "== PC_CHUNK=$c"
  POLYCAP_AMD_LIB=/tmp/libpolycap_c$c.so timeout -k 10 200 python scripts/bench_fetch.py 10000000 2>&1 | grep "run_parts=1:\|run_parts=4\|run_parts=16"
  POLYCAP_AMD_LIB=/tmp/libpolycap_c$c.so timeout -k 10 100 python scripts/bench_ne.py xos1 1 300000 2>&1 | grep -o "kernel [0-9.]* ms"
done
