"""One 1e7-slot launch with the given context options; prints the scheduler's phase counts (run under rocprofv3 --pmc
SQ_INSTS_VALU to relate instruction counts to phase counts):  python3 scripts/analysis/phase_cost.py opt=value ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import polycap_amd
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
prob = polycap_amd.problem_from_inp(os.path.join(root, "tests", "golden", "example", "xos1.inp"), energies=[10.0])
with polycap_amd.TraceContext(prob) as ctx:
    ctx.set_option("plane_images", 1)
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    ctx.run(20000, 0, 10_000_000, keep_images=True)
    ms = ctx.wait()
    st = ctx.phase_stats()
    print("PHASES %s kernel_ms %.2f march %d %d event %d %d new %d %d" % (",".join(sys.argv[1:]), ms, st["march"]["phases"], st["march"]["lanes"],
          st["event"]["phases"], st["event"]["lanes"], st["new"]["phases"], st["new"]["lanes"]))
