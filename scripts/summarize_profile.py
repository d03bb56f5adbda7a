"""Condenses gpurun_out/prof_<tag>_* (rocprofv3 csv) into profiles/<round>/<tag>_*.{csv,json}:
    python scripts/summarize_profile.py <tag> <round> --kernel <name prefix> [--last]
--kernel: the kernel the summary is about, by the start of its name (e.g. pc_leak_kernel, pc_trace_log_kernel,
          pc_trace_producer_kernel); required.  The script FAILS when a counter pass holds no dispatch of it -- round 3's
          leak summary silently described the 4.8 ms pre-pass kernel because the kernel was chosen by dispatch count.
--last:   the counters of the LAST dispatch of that kernel (commands whose earlier dispatches are small warm-up runs) instead
          of the average over its dispatches."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def main(argv):
    argv = list(argv)
    last_only = "--last" in argv
    if last_only:
        argv.remove("--last")
    if "--kernel" not in argv:
        sys.exit("summarize_profile.py: --kernel <name prefix> is required")
    k = argv.index("--kernel")
    prefix = argv[k + 1]
    del argv[k:k + 2]
    tag, rnd = argv[0], argv[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "profiles", rnd)
    os.makedirs(out_dir, exist_ok=True)
    ks = sorted(glob.glob(os.path.join(root, "gpurun_out", "prof_%s_kt" % tag, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if ks:
        shutil.copy(ks[-1], os.path.join(out_dir, "%s_kernel_stats.csv" % tag))
    summary, meta, passes = {}, {}, 0
    for d in sorted(glob.glob(os.path.join(root, "gpurun_out", "prof_%s_pmc*" % tag))):
        for f in sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:     # the newest pass only
            passes += 1
            rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].replace("void ", "").startswith(prefix)]
            if not rows:
                sys.exit("summarize_profile.py: %s holds no dispatch of a kernel named %s*" % (f, prefix))
            names = sorted(set(r["Kernel_Name"] for r in rows))
            if len(names) > 1:
                sys.exit("summarize_profile.py: %s matches several kernels in %s: %s" % (prefix, f, names))
            agg = collections.defaultdict(list)
            for r in rows:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = dict(kernel=r["Kernel_Name"], VGPR=r["VGPR_Count"], SGPR=r["SGPR_Count"], LDS=r["LDS_Block_Size"],
                            grid=r["Grid_Size"], workgroup=r["Workgroup_Size"])
            for c, v in agg.items():
                summary[c] = v[-1] if last_only else sum(v) / len(v)
                meta["dispatches"] = len(v)
    if not passes:
        sys.exit("summarize_profile.py: no counter passes under gpurun_out/prof_%s_pmc*" % tag)
    summary["meta"] = meta
    summary["note"] = (("the last dispatch of the kernel in the profiled command (earlier ones are warm-up runs); " if last_only else
                        "averages per dispatch of the kernel over the command's launches; ") +
                       "FETCH_SIZE / WRITE_SIZE in KB as reported by rocprofv3, collected in separate --pmc passes")
    b = os.path.join(root, "gpurun_out", "bench_%s.json" % tag)
    if os.path.exists(b):
        shutil.copy(b, os.path.join(out_dir, "%s_bench.json" % tag))
    json.dump(summary, open(os.path.join(out_dir, "%s_pmc_summary.json" % tag), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
