#!/bin/bash
# Round-4 evidence in one GPU call: the whole GPU suite, the default bench line, the rocprofv3 passes of the four kernels the
# bench line quotes, the BASELINE configurations at full per-GPU size, the other energy counts and the leak bench.  Results under
# gpurun_out/ (condense with scripts/summarize_profile.py <tag> r04 --kernel <name> [--last]).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/final_tests.log 2>&1
rc=$?
tail -4 gpurun_out/final_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > gpurun_out/bench_headline.json 2> gpurun_out/bench_headline.err || exit 1
timeout -k 10 500 bash scripts/r04_profiles.sh > gpurun_out/final_profiles.log 2>&1
timeout -k 10 300 python scripts/full_size_runs.py > gpurun_out/full_size_runs.txt 2>&1
{
for a in "xos1 12 1000000" "xos1 40 1000000" "xos1 100 1000000" "xos1 40 1000000 - range=10:30" "xos1 291 1000000" "xos1 291 4000000" "ellip_l9 291 500000 5.0"; do
  timeout -k 10 120 python scripts/bench_ne.py $a 2>&1 | head -1
done
} > gpurun_out/bench_ne_r04.txt
timeout -k 10 120 python scripts/analysis/log_kernel_sizes.py > gpurun_out/log_kernel_sizes.txt 2>&1
timeout -k 10 300 python scripts/bench_leak.py 16384,262144,1048576 > gpurun_out/leak_bench_r04.txt 2>&1
POLYCAP_LEAK_TIMING=1 timeout -k 10 120 python scripts/leak_one.py 262144 1 > gpurun_out/leak_timing_r04.txt 2>&1
tail -c 400 gpurun_out/bench_headline.json
