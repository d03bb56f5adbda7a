"""Condenses gpurun_out/prof_<tag>_* (rocprofv3 csv) into profiles/<round>/<tag>_*.{csv,json}:
    python scripts/summarize_profile.py <tag> [round, default r02] [--last]
--last: the counters of the LAST dispatch of the trace kernel (commands whose earlier dispatches are small warm-up runs) instead
of the average over its dispatches."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

last_only = "--last" in sys.argv
if last_only:
    sys.argv.remove("--last")
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "profiles", sys.argv[2] if len(sys.argv) > 2 else "r02")
os.makedirs(out_dir, exist_ok=True)
ks = sorted(glob.glob(os.path.join(root, "gpurun_out", "prof_%s_kt" % tag, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if ks:
    shutil.copy(ks[-1], os.path.join(out_dir, "%s_kernel_stats.csv" % tag))
summary = {}
meta = {}
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", "prof_%s_pmc*" % tag))):
    for f in sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:     # the newest pass only
        rows = [r for r in csv.DictReader(open(f)) if "pc_trace" in r["Kernel_Name"] or "pc_leak" in r["Kernel_Name"]]
        # the trace kernel of the timed steps: the one with the most dispatches (a context's first big run is preceded by a
        # small probe launch of the default kernel, which is not what is profiled)
        names = collections.Counter(r["Kernel_Name"] for r in rows)
        if not names:
            continue
        main = names.most_common(1)[0][0]
        agg = collections.defaultdict(list)
        for r in rows:
            if r["Kernel_Name"] == main:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = dict(kernel=r["Kernel_Name"], VGPR=r["VGPR_Count"], SGPR=r["SGPR_Count"], LDS=r["LDS_Block_Size"],
                            grid=r["Grid_Size"], workgroup=r["Workgroup_Size"])
        for k, v in agg.items():
            summary[k] = v[-1] if last_only else sum(v) / len(v)
summary["meta"] = meta
summary["note"] = (("the last dispatch of the trace kernel in the profiled command (earlier ones are warm-up runs); " if last_only else
                    "averages per dispatch of the trace kernel over bench.py's launches (1e7 exit-photon slots each); ") +
                   "FETCH_SIZE / WRITE_SIZE in KB as reported by rocprofv3, collected in separate --pmc passes")
b = os.path.join(root, "gpurun_out", "bench_%s.json" % tag)
if os.path.exists(b):
    shutil.copy(b, os.path.join(out_dir, "%s_bench.json" % tag))
json.dump(summary, open(os.path.join(out_dir, "%s_pmc_summary.json" % tag), "w"), indent=1)
print(json.dumps(summary, indent=1))
