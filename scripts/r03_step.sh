#!/bin/bash
# one GPU iteration of round 3: targeted parity tests, then the many-energy timings (results under gpurun_out/)
set -o pipefail
mkdir -p gpurun_out
TAG=${1:-step}
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "many_energies or batched or long_profile or c5_deck or energy_sweep or multi_energy" > gpurun_out/${TAG}_tests.log 2>&1
rc=$?
tail -5 gpurun_out/${TAG}_tests.log
[ $rc -ne 0 ] && exit $rc
{
timeout -k 10 120 python scripts/bench_ne.py xos1 291 1000000 &&
timeout -k 10 120 python scripts/bench_ne.py ellip_l9 291 500000 5.0 &&
timeout -k 10 120 python scripts/bench_ne.py xos1 100 1000000 &&
timeout -k 10 120 python scripts/bench_ne.py xos1 40 1000000 &&
timeout -k 10 120 python scripts/bench_ne.py xos1 12 1000000
} > gpurun_out/${TAG}_ne.log 2>&1
rc=$?
cat gpurun_out/${TAG}_ne.log
exit $rc
