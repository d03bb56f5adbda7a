#!/bin/bash
# Round-3 evidence in one GPU call: the whole GPU suite, the default bench line, the rocprofv3 passes of the three kernels the
# bench line quotes, the optical-constant self-check and the leak bench.  Results under gpurun_out/ (condense with
# scripts/summarize_profile.py <tag> r03 [--last]).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/final_tests.log 2>&1
rc=$?
tail -4 gpurun_out/final_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > gpurun_out/bench_headline.json 2> gpurun_out/bench_headline.err || exit 1
bash scripts/profile_r03.sh headline python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/final_prof_headline.log 2>&1
bash scripts/profile_r03.sh ne291 python3 scripts/side_workload.py sweep_291 > gpurun_out/final_prof_ne291.log 2>&1
bash scripts/profile_r03.sh ellip291 python3 scripts/side_workload.py ellip_291 > gpurun_out/final_prof_ellip291.log 2>&1
timeout -k 10 300 python scripts/optconst_selfcheck.py --out gpurun_out/optconst_selfcheck.json > /dev/null 2> gpurun_out/optconst_selfcheck.err
timeout -k 10 300 python scripts/bench_leak.py 16384,262144 > gpurun_out/leak_bench_r03.txt 2>&1
POLYCAP_LEAK_TIMING=1 timeout -k 10 120 python scripts/leak_one.py 262144 1 > gpurun_out/leak_timing_r03.txt 2>&1
bash scripts/profile_leak.sh leak > gpurun_out/final_prof_leak.log 2>&1
tail -c 600 gpurun_out/bench_headline.json
