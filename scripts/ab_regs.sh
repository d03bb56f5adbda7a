# A/B of the register-weight kernels: waves per SIMD the register allocator targets (run on the GPU box)
cd $GRAFT_REPO_ROOT
for cfg in "4 512" "3 384" "2 512"; do
  set -- $cfg
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DPC_MIN_WAVES=$1 -Iinclude -Ipolycap_amd/csrc/hip -c polycap_amd/csrc/hip/pc_kernels.hip -o /tmp/kr_$1.o 2>/dev/null
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libpolycap_r$1.so polycap_amd/lib/obj/pc_*.c.o /tmp/kr_$1.o -ldl -lm
  echo "== waves/SIMD=$1 block_size=$2"
  bpc=2; if [ "$1" = "2" ]; then bpc=1; fi
  for ne in 1 4 8; do
    POLYCAP_AMD_LIB=/tmp/libpolycap_r$1.so timeout -k 10 200 python scripts/bench_ne.py xos1 $ne 4000000 - block_size=$2 blocks_per_cu=$bpc 2>&1 | grep -v "avg lanes" | sed 's/sig=None.*kernel/kernel/'
  done
done
