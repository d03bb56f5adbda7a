#!/usr/bin/env python3
"""What the built-in optical-constant tables are worth.

    python scripts/optconst_selfcheck.py [--out profiles/r03/optconst_selfcheck.json] [--photons 300000]

Part 1 (CPU): the O 53 % / Si 47 % glass of the reference's tests and decks, amu and scatf at the energies where the tables
were FITTED to the reference's known answers (10, 40, 80 keV) -- next to what the plain table entries (NIST grid value for
mu/rho of Si, interpolated f') give there (POLYCAP_OPTCONST_UNFITTED=1), and the relative difference.  That difference is
what the unfitted tables are off by at those three energies: the only points where this image offers a comparison.
Part 2 (needs the GPU): the reference's seven-energy transmission curve (tests/source.c:216-222, 30000 photons in the
reference; more here so that the statistical error is below the table error) with the fitted and with the unfitted tables,
residuals against the published efficiencies and their tolerances.  Every number in this file comes from the built-in
tables: nothing here is xraylib's.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import polycap_amd
E = %(energies)r
amu, scatf, syn = polycap_amd.optical_constants([8, 14], [53.0, 47.0], 2.23, E)
out = {"energies": E, "amu": [float(a) for a in amu], "scatf": [float(s) for s in scatf], "provider": polycap_amd.optical_constants_provider()}
n = %(photons)d
if n > 0 and polycap_amd.device_count() > 0:
    from polycap_amd import capi
    with open(os.path.join(%(root)r, "tests", "golden", "reference_known_answers.json")) as f:
        known = json.load(f)
    t, o = known["transmission_curve"], known["test_optic"]
    prof = capi.Profile(o["type"], o["length"], o["rad_ext_upstream"], o["rad_ext_downstream"], o["rad_int_upstream"],
                        o["rad_int_downstream"], o["focal_dist_upstream"], o["focal_dist_downstream"])
    desc = capi.Description(prof, o["sig_rough"], o["n_cap"], {"O": 53.0, "Si": 47.0}, known["glass"]["density"])
    src = capi.Source(desc, t["d_source"], t["src_x"], t["src_y"], t["src_sigx"], t["src_sigy"], t["src_shiftx"],
                      t["src_shifty"], t["hor_pol"], np.array(t["energies"], dtype=np.float64))
    sys.stdout.flush(); saved = os.dup(1); os.dup2(2, 1)
    eff = src.get_transmission_efficiencies(-1, n)
    import ctypes; ctypes.CDLL(None).fflush(None); os.dup2(saved, 1)
    out["curve"] = [float(x) for x in eff.data[1]]
print(json.dumps(out))
"""


def child(energies, photons, unfitted):
    env = dict(os.environ, POLYCAP_OPTCONST="builtin", POLYCAP_SEED="20000", POLYCAP_IMAGES="0")
    if unfitted:
        env["POLYCAP_OPTCONST_UNFITTED"] = "1"
    else:
        env.pop("POLYCAP_OPTCONST_UNFITTED", None)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "energies": energies, "photons": photons}],
                       capture_output=True, text=True, env=env, timeout=1200)
    if r.returncode != 0:
        raise SystemExit(r.stderr)
    return json.loads(r.stdout.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03", "optconst_selfcheck.json"))
    ap.add_argument("--photons", type=int, default=300000)
    args = ap.parse_args()
    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as f:
        known = json.load(f)
    with open(os.path.join(ROOT, "tests", "golden", "reference_leak_known_answers.json")) as f:
        leak = json.load(f)
    E = [10.0, 40.0, 80.0]
    fit, raw = child(E, 0, False), child(E, 0, True)
    pins = {"10": {"amu": known["glass"]["amu"], "scatf": known["glass"]["scatf"], "pinned_by": "tests/photon.c:75-76 (tolerances 1e-3 / 1e-5)"},
            "40": {"amu": leak["constants"]["40"]["amu"], "scatf": leak["constants"]["40"]["scatf"],
                   "pinned_by": "2-parameter fit to the nine leak weights of tests/leaks.c (DESIGN section 10)"},
            "80": {"amu": leak["constants"]["80"]["amu"], "scatf": None, "pinned_by": "fit to the one leak weight of tests/leaks.c:947"}}
    points = []
    for k, e in enumerate(E):
        p = pins[str(int(e))]
        row = {"energy_keV": e, "pinned_by": p["pinned_by"],
               "amu": {"fitted_table": fit["amu"][k], "plain_table": raw["amu"][k], "answer": p["amu"],
                       "plain_vs_answer_rel": raw["amu"][k] / p["amu"] - 1.0, "fitted_vs_answer_rel": fit["amu"][k] / p["amu"] - 1.0},
               "scatf": {"fitted_table": fit["scatf"][k], "plain_table": raw["scatf"][k], "answer": p["scatf"]}}
        if p["scatf"] is not None:
            row["scatf"]["plain_vs_answer_rel"] = raw["scatf"][k] / p["scatf"] - 1.0
            row["scatf"]["fitted_vs_answer_rel"] = fit["scatf"][k] / p["scatf"] - 1.0
        points.append(row)
    doc = {"what": "built-in optical-constant tables (polycap_amd/csrc/host/pc_optconst.c), O 53 % / Si 47 % glass, 2.23 g/cm3: the entries "
                   "fitted to the reference's known answers against the plain table entries at the same energies",
           "provider": fit["provider"], "points": points,
           "reading": "plain_vs_answer_rel is how far the unfitted tables are from values the reference's own tests pin -- the only "
                      "check of the tables this image allows; everywhere else the tables are unverified and flagged synthetic"}
    t = known["transmission_curve"]
    cf = child(t["energies"], args.photons, False)
    if "curve" in cf:
        cr = child(t["energies"], args.photons, True)
        doc["seven_energy_curve"] = {
            "source": "reference tests/source.c:216-222 (xraylib constants, 30000 photons, seed from /dev/urandom)",
            "photons_here": args.photons, "energies_keV": t["energies"], "published": t["efficiencies"], "tolerances": t["tolerances"],
            "fitted_tables": cf["curve"], "plain_tables": cr["curve"],
            "residual_fitted": [a - b for a, b in zip(cf["curve"], t["efficiencies"])],
            "residual_plain": [a - b for a, b in zip(cr["curve"], t["efficiencies"])],
            "within_tolerance_fitted": [abs(a - b) <= tol for a, b, tol in zip(cf["curve"], t["efficiencies"], t["tolerances"])],
            "within_tolerance_plain": [abs(a - b) <= tol for a, b, tol in zip(cr["curve"], t["efficiencies"], t["tolerances"])],
            "amu_fitted": cf["amu"], "scatf_fitted": cf["scatf"], "amu_plain": cr["amu"], "scatf_plain": cr["scatf"],
            "note": "published values carry 3 digits and the reference's own statistical error at 30000 photons (~0.002 at 10 keV)"}
    else:
        doc["seven_energy_curve"] = None
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
