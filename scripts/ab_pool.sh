# A/B builds of the pool kernel: waves per CU x parked photons per wave (run on the GPU box)
cd $GRAFT_REPO_ROOT
for cfg in "768 3 64" "512 2 64" "1024 4 48"; do
  set -- $cfg
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DPQ_BLOCK=$1 -DPQ_MIN_WAVES=$2 -DPQ_P=$3 -Iinclude -Ipolycap_amd/csrc/hip -c polycap_amd/csrc/hip/pc_kernels.hip -o /tmp/kp_$1.o 2>/dev/null
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libpolycap_p$1.so polycap_amd/lib/obj/pc_*.c.o /tmp/kp_$1.o -ldl -lm
  echo "== block=$1 waves/SIMD=$2 parked=$3"
  for o in "pool_refill=20" "pool_refill=12 pool_march_min=24"; do
    POLYCAP_AMD_LIB=/tmp/libpolycap_p$1.so timeout -k 10 120 python scripts/ab_pool.py xos1 4000000 $o 2>&1 | grep -v "images=True\|identity"
  done
done
