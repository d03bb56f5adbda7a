"""Where the launching-wave kernel starts to pay: xos1 at rising energies (photons die earlier), both kernels, with the
reflections of transmitted photons per launch that the context chooses by.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import polycap_amd
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for deck, E in (("xos1", 10.0), ("xos1", 15.0), ("xos1", 20.0), ("xos1", 25.0), ("xos1", 30.0), ("ellip_l9", 20.0), ("ellip_l9", 30.0), ("cone", 10.0)):
    prob = polycap_amd.problem_from_inp(os.path.join(root, "tests", "golden", "example", deck + ".inp"), energies=[E])
    with polycap_amd.TraceContext(prob) as ctx:
        out = []
        for prod in (0, 1):
            ctx.set_option("producer", prod)
            best = 1e9
            for _ in range(3):
                ctx.run(3, 0, 2_000_000)
                best = min(best, ctx.wait())
            t = ctx.totals(check=False)
            out.append(best)
        c = t["counters"]
        ev = ctx.phase_stats()["event"]["lanes"]
        ctx.set_option("producer", -1)
        ctx.run(3, 0, 2_000_000)
        ctx.wait()
        print("%-9s %4.0f keV: lane %.2f ms, launching wave %.2f ms (%+.0f %%); segment visits per launch %.2f, exit/launch %.2f -> %s" %
              (deck, E, out[0], out[1], 100.0*(out[1]/out[0] - 1.0), ev/max(1, c[5]), c[0]/max(1, c[5]), ctx.last_kernel()), flush=True)
