"""Timeline of the last C-API call in a rocprofv3 --kernel-trace --memory-copy-trace run:
   python scripts/analysis/timeline.py gpurun_out/prof_api/<host>/<pid>  [n_parts]"""
import csv, sys
base = sys.argv[1]
nparts = int(sys.argv[2]) if len(sys.argv) > 2 else 4
k = list(csv.DictReader(open(base + '_kernel_trace.csv')))
m = list(csv.DictReader(open(base + '_memory_copy_trace.csv')))
ker = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id']) for r in k if 'pc_trace_pool' in r['Kernel_Name'])[-nparts:]
t0 = ker[0][0]
for a, b, q in ker:
    print("kernel %7.2f -> %7.2f ms  queue %s" % ((a - t0)/1e6, (b - t0)/1e6, q))
cp = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in m if int(r['Start_Timestamp']) >= t0)
busy = sum(b - a for a, b in cp)/1e6
print("%d copies, busy %.1f ms, first starts %.2f, last ends %.2f" % (len(cp), busy, (cp[0][0] - t0)/1e6, (cp[-1][1] - t0)/1e6))
gaps = [(cp[i][1], cp[i+1][0]) for i in range(len(cp) - 1) if cp[i+1][0] - cp[i][1] > 200000]
for a, b in gaps:
    print("  copy engine idle %7.2f -> %7.2f ms" % ((a - t0)/1e6, (b - t0)/1e6))
