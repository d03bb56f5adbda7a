# Sweep of the scheduler options of pc_trace_kernel on the headline workload (run on the GPU box):
#   bash scripts/sweep_opts.sh  -> one line per setting: kernel ms
cd $GRAFT_REPO_ROOT
for et in 8 12 16 20 24 32; do for nt in 2 4 8 16; do for mb in 8 16 32; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-images --opt event_threshold=$et --opt new_threshold=$nt --opt march_burst=$mb 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('et=$et nt=$nt mb=$mb kernel_ms %.2f' % d['roofline']['kernel_ms'])"
done; done; done
