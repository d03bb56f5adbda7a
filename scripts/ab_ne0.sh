# A/B of the any-n_energies kernel: register budget / waves per CU (run on the GPU box): bash scripts/ab_ne0.sh
cd $GRAFT_REPO_ROOT
for cfg in "512 2 512" "768 3 768" "1024 4 1024"; do
  set -- $cfg
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DPC_BLOCK=$1 -DPC_MIN_WAVES_NE0=$2 -Iinclude -Ipolycap_amd/csrc/hip -c polycap_amd/csrc/hip/pc_kernels.hip -o /tmp/k_$1.o 2>/dev/null
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libpolycap_$1.so polycap_amd/lib/obj/pc_*.c.o /tmp/k_$1.o -ldl -lm
  echo "== PC_BLOCK=$1 waves/SIMD=$2 block_size=$3"
  for ne in 16 30 100 291; do
    POLYCAP_AMD_LIB=/tmp/libpolycap_$1.so timeout -k 10 200 python scripts/bench_ne.py xos1 $ne 1000000 - block_size=$3 2>&1 | grep -v "avg lanes" | sed 's/sig=None.*kernel/kernel/'
  done
done
