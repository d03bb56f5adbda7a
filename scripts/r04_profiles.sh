#!/bin/bash
# rocprofv3 evidence of round 4 in one GPU call (results under gpurun_out/; condense with
#   python scripts/summarize_profile.py <tag> r04 --kernel <name> [--last]):
#   headline  bench.py's timed steps                      --kernel pc_trace_producer_kernel
#   ne291     bench.py's sweep_291 leg                    --kernel pc_trace_log_kernel --last
#   ellip291  bench.py's ellip_l9_rough.n_energies_291    --kernel pc_trace_log_kernel --last
#   leak      scripts/leak_one.py 262144                  --kernel pc_leak_kernel --last
mkdir -p gpurun_out
bash scripts/profile_r04.sh headline python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/prof_headline.log 2>&1
bash scripts/profile_r04.sh ne291 python3 scripts/side_workload.py sweep_291 > gpurun_out/prof_ne291.log 2>&1
bash scripts/profile_r04.sh ellip291 python3 scripts/side_workload.py ellip_291 > gpurun_out/prof_ellip291.log 2>&1
bash scripts/profile_leak_r04.sh leak > gpurun_out/prof_leak.log 2>&1
tail -3 gpurun_out/prof_headline.log gpurun_out/prof_ne291.log gpurun_out/prof_ellip291.log gpurun_out/prof_leak.log
