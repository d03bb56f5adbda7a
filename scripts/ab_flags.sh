# A/B of compiler flags for the trace kernels on the headline workload (run on the GPU box)
cd $GRAFT_REPO_ROOT
i=0
while IFS= read -r flags; do
  i=$((i+1))
  if hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden $flags -Iinclude -Ipolycap_amd/csrc/hip -c polycap_amd/csrc/hip/pc_kernels.hip -o /tmp/kf_$i.o 2>/tmp/kf_$i.err; then
    hipcc -shared -fPIC --offload-arch=gfx950 -o /tmp/libpolycap_f$i.so polycap_amd/lib/obj/pc_*.c.o /tmp/kf_$i.o -ldl -lm -lpthread
    echo -n "[$flags] "
    POLYCAP_AMD_LIB=/tmp/libpolycap_f$i.so timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.4g ph/s kernel %.2f ms' % (d['value'], d['roofline']['kernel_ms']))"
  else
    echo "[$flags] does not compile: $(tail -1 /tmp/kf_$i.err)"
  fi
done <<'FLAGS'

-DPC_L1=4 -DPC_L2=16
-DPC_L1=6 -DPC_L2=24
-DPC_L1=8 -DPC_L2=32
-DPC_L1=4 -DPC_L2=24
-DPC_L1=3 -DPC_L2=15
-DPC_L1=5 -DPC_L2=35
FLAGS
