# block-certificate strides on decks / energy counts other than the headline (run on the GPU box)
cd $GRAFT_REPO_ROOT
for cfg in "8 32 1" "5 25 2" "4 16 2"; do
  set -- $cfg
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DPC_L1=$1 -DPC_L2=$2 -DPC_LV_LATER=$3 -Iinclude -Ipolycap_amd/csrc/hip -c polycap_amd/csrc/hip/pc_kernels.hip -o /tmp/ks_$1.o 2>/dev/null
  hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libpolycap_s$1.so polycap_amd/lib/obj/pc_*.c.o /tmp/ks_$1.o -ldl -lm -lpthread
  echo "== strides $1/$2, later flights start at level $3"
  for ne in 1 4 8 16 291; do POLYCAP_AMD_LIB=/tmp/libpolycap_s$1.so timeout -k 10 200 python scripts/bench_ne.py xos1 $ne 4000000 2>&1 | grep -o "n_E=.*started photons/s" | sed 's/sig=None.*slots, //'; done
  for d in ellip_l9 cone; do POLYCAP_AMD_LIB=/tmp/libpolycap_s$1.so timeout -k 10 200 python scripts/bench_ne.py $d 1 4000000 5.0 2>&1 | grep -o "^[a-z_0-9]* n_E=.*started photons/s" | sed 's/sig=5.0.*slots, //'; done
done
