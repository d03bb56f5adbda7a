"""Random optics against the invariants of the trace kernels (GPU; the oracle and the host compile are the checkers):
smooth random profiles (ext = ext0 (1 + a s + b s^2), capillary radius in proportion, 100 ... 999 segments, 2 ... 200000
capillaries), random sources (uniform / divergent, near / far, shifted), 1 / 3 / 12 / 40 energies.  Per case:
  (a) explicit photons: certified march == literal march, bit for bit;
  (b) explicit photons: kernel == host compile of the device header, bit for bit;
  (c) source runs: lane, pool and producer kernels (one energy), immediate and logged sweeps (40 energies): same counters, exact sums, planes;
  (d) source run against the oracle: started photons, reflections, summed weight within the chaos noise;
  (e) every second optic with leak_calc=true (40 keV, near divergent source): certified wall search == literal stepping, kernel == host compile, event for event.
TEST INFRASTRUCTURE (imports the oracle): tests/test_gpu_fuzz.py runs it; by hand  python tests/fuzz_optics.py [n_cases] [seed] [--debug]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def run(n_cases=24, seed=20260405, debug=False, out=lambda line: print(line, flush=True), leak_every=2):
    """Returns the number of cases with a remark; one line per case goes to `out`."""
    import polycap_amd as pa
    from oracle import pyoracle as oracle
    from tests.emul import pyemul
    from tests.common import GLASS, PIN_E, PIN_AMU, PIN_SCATF, synthetic_constants
    rng = np.random.default_rng(seed)
    oracle.build()
    bad = 0
    for case in range(n_cases):
        bad += _one_case(case, rng, pa, oracle, pyemul, GLASS, PIN_E, PIN_AMU, PIN_SCATF, synthetic_constants, debug, out, leak_every)
    out("cases with remarks: %d of %d" % (bad, n_cases))
    return bad


def _one_case(case, rng, pa, oracle, pyemul, GLASS, PIN_E, PIN_AMU, PIN_SCATF, synthetic_constants, debug, out, leak_every):
    nmax = int(rng.choice([100, 400, 999]))
    n_cap = int(rng.choice([2, 7, 61, 1027, 20419, 200000]))
    L = rng.uniform(2., 12.)
    ext0 = rng.uniform(0.05, 0.4)
    while True:
        a, b = rng.uniform(-0.8, 0.8), rng.uniform(-0.6, 0.6)
        s = np.linspace(0., 1., nmax + 1)
        shape = 1. + a*s + b*s*s
        if shape.min() > 0.25 and shape.max() < 1.6:
            break
    z = L*s
    ext = ext0*shape
    open_area = rng.uniform(0.3, 0.75)
    n_shells = round(np.sqrt(12.*n_cap - 3.)/6. - 0.5) if n_cap > 2 else 0
    if n_shells == 0:
        cap = ext*rng.uniform(0.3, 0.9)        # mono-capillary
    else:
        cap = ext*np.sqrt(open_area*2.598076/(np.pi*n_cap))
    d = float(rng.choice([5., 50., 2000.]))
    uniform = rng.random() < 0.5
    sx, sy = ext0*rng.uniform(0.2, 1.2), ext0*rng.uniform(0.2, 1.2)
    sig = (-1., 0.) if uniform else (rng.uniform(1e-3, 2e-2), rng.uniform(1e-3, 2e-2))
    shift = (0., 0.) if rng.random() < 0.6 else (ext0*rng.uniform(-0.3, 0.3), ext0*rng.uniform(-0.3, 0.3))
    source = (d, sx, sy, sig[0], sig[1], shift[0], shift[1], rng.uniform(0., 1.))
    ne = int(rng.choice([1, 1, 3, 12, 40]))
    E = np.array([PIN_E]) if ne == 1 else (np.array([6., 10., 17.]) if ne == 3 else np.linspace(4., 25., ne))
    amu, scatf = (np.array([PIN_AMU]), np.array([PIN_SCATF])) if ne == 1 else synthetic_constants(E)
    sig_rough = 0.0 if rng.random() < 0.7 else rng.uniform(1., 8.)
    tag = "case %d: nmax %d n_cap %d L %.2f ext0 %.3f a %.2f b %.2f open %.2f d %.0f %s nE %d sig %.1f" % (
        case, nmax, n_cap, L, ext0, a, b, open_area, d, "uniform" if uniform else "divergent", ne, sig_rough)
    optic = oracle.Optic(z, cap, ext, sig_rough, n_cap, GLASS["density"])
    src = oracle.make_source(*source)
    prob = pa.Problem(z, cap, ext, sig_rough, n_cap, GLASS["density"], E, amu, scatf, *source)
    notes = []
    with pa.TraceContext(prob) as ctx:
        n = 20000
        ph = ctx.sample_photons(9, np.arange(n))
        ok = np.isfinite(ph[:, :9]).all(axis=1)
        ph = ph[ok]
        fast = ctx.launch_photons(ph[:, 0:3], ph[:, 3:6], ph[:, 6:9])
        ctx.set_option("literal_march", 1)
        lit = ctx.launch_photons(ph[:, 0:3], ph[:, 3:6], ph[:, 6:9])
        ctx.set_option("literal_march", 0)
        for k in fast:
            if not np.array_equal(fast[k], lit[k], equal_nan=True):
                notes.append("(a) certified != literal in %s (%d photons)" % (k, int((fast[k] != lit[k]).reshape(len(ph), -1).any(axis=1).sum())))
        em = pyemul.launch_batch(prob, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9])
        # the C ABI hands a photon that never reached a reflection its caller's electric vector back (the kernels work with the
        # normalised one); with roughness (libm exp) or FORM 3 (hardware rsq / rcp, 2e-14 per reflection) the weights equal the host compile's to 1e-10
        touched = ~((fast["rc"] == -2) | (fast["rc"] == 2) | ((fast["rc"] == 1) & (fast["i_refl"] == 0)))
        exact_w = sig_rough == 0. and ne <= 8
        for k in fast:
            if k == "exit_elecv":
                if not np.array_equal(fast[k][touched], em[k][touched], equal_nan=True):
                    notes.append("(b) kernel != host compile in exit_elecv of photons that reflected")
                continue
            if k == "weights" and not exact_w:
                m = np.isin(fast["rc"], (0, 1))
                dw = np.abs(fast[k][m] - em[k][m])/np.maximum(np.abs(em[k][m]), 1e-300)
                if dw.size and dw.max() > 1e-10:
                    notes.append("(b) weights differ from the host compile by %.1e" % dw.max())
                continue
            if not np.array_equal(fast[k], em[k], equal_nan=True):
                rows = np.where((~((fast[k] == em[k]) | (np.isnan(fast[k]) & np.isnan(em[k])))).reshape(len(ph), -1).any(axis=1))[0]
                notes.append("(b) kernel != host compile in %s (%d photons, rc %s)" % (k, len(rows), sorted(set(fast["rc"][rows].tolist()))))
                if debug:
                    for j in rows[:4]:
                        print("   photon", j, "rc", fast["rc"][j], em["rc"][j], "i_refl", fast["i_refl"][j], k, [float(x).hex() for x in np.atleast_1d(fast[k][j])],
                              [float(x).hex() for x in np.atleast_1d(em[k][j])], "start elecv", [float(x).hex() for x in ph[j, 6:9]], "|E|^2 - 1 = %.3e" % (float(np.dot(ph[j, 6:9], ph[j, 6:9])) - 1.0))
        ns = 3000
        runs = {}
        variants = (("lane", dict(pool=0, producer=0)), ("pool", dict(pool=1, producer=0)), ("producer", dict(pool=0, producer=1))) if ne == 1 else \
                   ((("immediate", dict(batch_reflections=0)), ("logged", dict(batch_reflections=1))) if ne > 8 else (("default", {}),))
        for name, opts in variants:
            for o, v in opts.items():
                ctx.set_option(o, v)
            ctx.run(11, 500, ns, max_attempts=4000, keep_images=True)
            ctx.wait()
            r = ctx.totals(check=False)
            r.update(ctx.images(0, ns))
            r["kernel"] = ctx.last_kernel()
            r["i_start"] = int(r["counters"][0] + r["counters"][1] + r["counters"][2])
            r["i_exit"] = int(r["counters"][0])
            runs[name] = r
        names = list(runs)
        r0 = runs[names[0]]
        done = r0["exit_weights"][:, 0] > 0
        for name in names[1:]:
            r = runs[name]
            same_sums = np.array_equal(r0["sumw_fixed"], r["sumw_fixed"]) if (sig_rough == 0. or ne <= 8) else \
                np.abs(r["sum_weights"]/np.maximum(r0["sum_weights"], 1e-300) - 1.0).max() < 1e-13
            if not np.array_equal(r0["counters"][:6], r["counters"][:6]):
                notes.append("(c) %s != %s: counters %s %s" % (name, names[0], r["counters"][:6], r0["counters"][:6]))
            elif not same_sums:
                notes.append("(c) %s != %s: sums" % (name, names[0]))
            elif sig_rough == 0. or ne <= 8:
                if not np.array_equal(r0["exit_weights"], r["exit_weights"]) or not np.array_equal(r0["images"][done], r["images"][done], equal_nan=True):
                    notes.append("(c) %s != %s: weights / planes" % (name, names[0]))
        ot = oracle.transmission(optic, src, E, amu, scatf, 11, 500, ns, images=True, max_attempts=4000) if "max_attempts" in oracle.transmission.__code__.co_varnames \
            else oracle.transmission(optic, src, E, amu, scatf, 11, 500, ns, images=True)
        # the trace is chaotic in the last bits (a 1-ulp change flips 4-15 % of the photons): started photons, reflections and the
        # summed weight agree with the oracle like two runs of different seeds do, c / sqrt(N) with c ~ 0.5-2
        oc = ot["counters"]
        o_start, g_start = int(oc[0] + oc[1] + oc[2]), r0["i_start"]
        tol = 4.0/np.sqrt(max(1, min(o_start, g_start)))
        # (slots that run out of attempts on an optic that transmits next to nothing: the exit counts differ like the started ones)
        if abs(int(r0["counters"][0]) - int(oc[0])) > 4.0*np.sqrt(max(0.0, float(ns - min(r0["counters"][0], oc[0])))) or abs(g_start - o_start) > tol*o_start:
            notes.append("(d) started photons device %d oracle %d (exit %d %d)" % (g_start, o_start, r0["counters"][0], oc[0]))
        # (reflections are counted for the transmitted photons only: their noise goes with the exit count)
        if oc[3] > 1000 and abs(int(r0["counters"][3]) - int(oc[3])) > 6.0/np.sqrt(max(1.0, float(oc[0])))*oc[3]:
            notes.append("(d) reflections device %d oracle %d" % (r0["counters"][3], oc[3]))
        so, sg = float(np.sum(ot["sum_weights"])), float(np.sum(r0["sum_weights"]))
        if so > 0 and abs(sg - so) > 2*tol*so:
            notes.append("(d) summed weight device %.6g oracle %.6g" % (sg, so))
    if leak_every and case % leak_every == 0:
        # (e) leak_calc=true on the same optic: explicit photons, kernel == host compile event for event (geometry bit for bit,
        # weights with exp(-mu d) to 1e-12), certified wall search == literal stepping bit for bit
        # hard photons (40 keV: mu ~ 0.9 / cm, walls are transparent) from a near, divergent source: wall crossings by the hundred
        src_l = (5., sx, sy, 0.02, 0.02, 0., 0., 0.5)
        a40, s40 = synthetic_constants(np.array([40.0]))
        prob = pa.Problem(z, cap, ext, sig_rough, n_cap, GLASS["density"], np.array([40.0]), a40, s40, *src_l)
        with pa.TraceContext(prob) as ctx:
            ph = ctx.sample_photons(13, np.arange(600))
            ph = ph[np.isfinite(ph[:, :9]).all(axis=1)]
            m = len(ph)
            g = ctx.launch_photons(ph[:m, 0:3], ph[:m, 3:6], ph[:m, 6:9], leak_calc=True)
            gext, gint = ctx.leaks()
            ctx.set_option("literal_march", 1)
            gl = ctx.launch_photons(ph[:m, 0:3], ph[:m, 3:6], ph[:m, 6:9], leak_calc=True)
            lext, lint = ctx.leaks()
        for k in g:
            if not np.array_equal(g[k], gl[k], equal_nan=True):
                notes.append("(e) certified != literal with leaks in %s" % k)
        if gext.shape != lext.shape or gint.shape != lint.shape or not (np.array_equal(gext, lext, equal_nan=True) and np.array_equal(gint, lint, equal_nan=True)):
            notes.append("(e) certified != literal: leak events %s %s / %s %s" % (gext.shape, gint.shape, lext.shape, lint.shape))
        e = pyemul.launch_leak(prob, ph[:m, 0:3], ph[:m, 3:6], ph[:m, 6:9])
        eext, eint = pyemul.sort_leak_records(e["records"])
        for k in ("rc", "exit_coords", "exit_dir", "i_refl", "d_travel"):
            if not np.array_equal(g[k], e[k], equal_nan=True):
                notes.append("(e) kernel != host compile with leaks in %s" % k)
        for got, exp, nm in ((gext, eext, "ext"), (gint, eint, "int")):
            if got.shape[0] != exp.shape[0] or not np.array_equal(got[:, 0], exp[:, 0]) or not np.array_equal(got[:, 2:12], exp[:, 4:14], equal_nan=True):
                notes.append("(e) %sleak events: kernel %d, host compile %d" % (nm, got.shape[0], exp.shape[0]))
            elif got.shape[0] and not np.allclose(got[:, 12:], exp[:, 14:], rtol=1e-12, atol=0.):
                notes.append("(e) %sleak weights differ" % nm)
        tag += " | leaks %d + %d" % (gext.shape[0], gint.shape[0])
    out(tag + " | photons %d, rc %s | started %d exit %d kernels %s | %s" % (
        len(ph), dict(zip(*np.unique(fast["rc"], return_counts=True))), r0["i_start"], r0["i_exit"], [runs[k]["kernel"] for k in names],
        "OK" if not notes else "; ".join(notes)))
    return int(len(notes) > 0)


if __name__ == "__main__":
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    sys.exit(1 if run(int(a[0]) if a else 24, int(a[1]) if len(a) > 1 else 20260405, "--debug" in sys.argv) else 0)
