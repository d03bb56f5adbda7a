#!/usr/bin/env python3
"""Fixture: totals of the CPU oracle's leak driver (oracle/polycap_oracle_leak.c, the reference's literal algorithm:
src/polycap-source.c:744-1087 with leak_calc, polycap_capil_trace_wall) on the reference's test optic under uniform illumination,
for many seeds -- what tests/test_gpu_leak.py::test_leak_driver_against_the_oracle_seed_by_seed needs for per-seed mean +- s.e.
    python scripts/make_oracle_leak_seeds.py [--jobs 6] [--tiny out.json] [--extend-seven N]   ->  tests/golden/oracle_leak_seeds.json
(--extend-seven N: N more seeds for the seven-energy group of the existing fixture)
(--tiny: 3 seeds x 100 slots and 2 seeds x 50 slots into out.json, to try the test's plumbing: POLYCAP_LEAK_SEEDS_FIXTURE=out.json)
Runs: 16 seeds x 8000 exit-photon slots at 10 keV (the leak bench's workload; 6 CPU-minutes per seed) and 4 (+ 4 by --extend-seven: 8) seeds x 4000 slots on
the seven energies of the reference's source test (tests/source.c:216-222: 1, 5, 10, 15, 20, 25, 30 keV).  The optical constants
each run used are stored with it: the device test feeds the same numbers."""
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
from oracle import pyoracle
from tests.conftest import GOLDEN
known = json.load(open(os.path.join(GOLDEN, "reference_known_answers.json")))
t = known["test_optic"]
optic = pyoracle.Optic.from_shape(t["type"], t["length"], t["rad_ext_upstream"], t["rad_ext_downstream"], t["rad_int_upstream"],
                                  t["rad_int_downstream"], t["focal_dist_upstream"], t["focal_dist_downstream"], t["sig_rough"],
                                  t["n_cap"], known["glass"]["density"])
src = (2000., 0.2065, 0.2065, -1., 0., 0., 0., 0.5)
E, A, S = %(E)r, %(A)r, %(S)r
o = pyoracle.transmission(optic, pyoracle.make_source(*src), E, A, S, %(seed)d, %(slot0)d, %(n)d, leak_calc=True)
ne = len(E)
c = [int(x) for x in o["counters"][:4]]
print(json.dumps({"seed": %(seed)d, "slot0": %(slot0)d, "n": %(n)d, "counters": c, "sum_weights": [float(x) for x in o["sum_weights"]],
                  "n_ext": int(len(o["ext"])), "n_int": int(len(o["int"])),
                  "ext_weights": [float(x) for x in o["ext"][:, 12:12 + ne].sum(axis=0)] if len(o["ext"]) else [0.0]*ne,
                  "int_weights": [float(x) for x in o["int"][:, 12:12 + ne].sum(axis=0)] if len(o["int"]) else [0.0]*ne}))
"""


def main():
    jobs = int(sys.argv[sys.argv.index("--jobs") + 1]) if "--jobs" in sys.argv else 6
    tiny = sys.argv[sys.argv.index("--tiny") + 1] if "--tiny" in sys.argv else None
    from oracle import pyoracle
    pyoracle.build()
    from tests.conftest import GOLDEN
    from tests.test_oracle_leak_known_answers import constants
    from polycap_amd.decks import optical_constants
    leaks = json.load(open(os.path.join(GOLDEN, "reference_leak_known_answers.json")))
    a10, s10 = constants(leaks, 10)
    E7 = [1.0, 5.0, 10.0, 15.0, 20.0, 25.0, 30.0]
    a7, s7, _ = optical_constants([8, 14], [0.53, 0.47], 2.23, E7)
    groups = [dict(name="10keV", energies=[10.0], amu=[float(a10)], scatf=[float(s10)], seeds=list(range(20000, 20016)), n=8000),
              dict(name="seven_energies", energies=E7, amu=[float(x) for x in a7], scatf=[float(x) for x in s7],
                   seeds=list(range(30000, 30004)), n=4000)]
    block = 1000
    extend = int(sys.argv[sys.argv.index("--extend-seven") + 1]) if "--extend-seven" in sys.argv else 0
    fixture = os.path.join(ROOT, "tests", "golden", "oracle_leak_seeds.json")
    if extend:
        # more seeds for the seven-energy group of the existing fixture (its other runs are kept as they are)
        with open(fixture) as f:
            old_doc = json.load(f)
        have = [r["seed"] for r in old_doc["groups"][1]["runs"]]
        groups[0]["seeds"] = []
        groups[1]["seeds"] = list(range(max(have) + 1, max(have) + 1 + extend))
        for gi in (0, 1):      # the constants the fixture's runs were made with
            assert groups[gi]["energies"] == old_doc["groups"][gi]["energies"]
            groups[gi]["amu"], groups[gi]["scatf"] = old_doc["groups"][gi]["amu"], old_doc["groups"][gi]["scatf"]
    if tiny:
        groups[0]["seeds"], groups[0]["n"] = [20000, 20001, 20002], 100
        groups[1]["seeds"], groups[1]["n"] = [30000, 30001], 50
        block = 50
    tasks = []
    for gi, g in enumerate(groups):
        for seed in g["seeds"]:
            for s0 in range(0, g["n"], block):
                tasks.append((gi, seed, s0, min(block, g["n"] - s0)))
    results = {}
    running = []
    it = iter(tasks)
    done = 0
    while True:
        while len(running) < jobs:
            try:
                t = next(it)
            except StopIteration:
                break
            g = groups[t[0]]
            p = subprocess.Popen([sys.executable, "-c", CHILD % {"root": ROOT, "seed": t[1], "slot0": t[2], "n": t[3],
                                                                  "E": g["energies"], "A": g["amu"], "S": g["scatf"]}],
                                 stdout=subprocess.PIPE, text=True, env=dict(os.environ, OMP_NUM_THREADS="1"))
            running.append((t, p))
        if not running:
            break
        t, p = running.pop(0)
        out, _ = p.communicate()
        if p.returncode != 0:
            raise SystemExit("oracle block failed: %r" % (t,))
        results.setdefault((t[0], t[1]), []).append(json.loads(out.strip().splitlines()[-1]))
        done += 1
        print("block %d / %d" % (done, len(tasks)), flush=True)
    doc = {"what": "oracle/polycap_oracle_leak.c, leak_calc=true driver: the reference's ellipsoidal test optic, uniform illumination "
                   "(2000 cm, 0.2065 x 0.2065, sigma -1); per group the energies and the optical constants of its runs, per run "
                   "(seed) the totals over its exit-photon slots [0, n): counters = i_exit, not_entered, not_transmitted, sum_irefl",
           "source": [2000., 0.2065, 0.2065, -1., 0., 0., 0., 0.5], "groups": []}
    for gi, g in enumerate(groups):
        runs = list(old_doc["groups"][gi]["runs"]) if extend else []
        for seed in g["seeds"]:
            bl = sorted(results[(gi, seed)], key=lambda b: b["slot0"])
            ne = len(g["energies"])
            runs.append({"seed": seed, "n": sum(b["n"] for b in bl),
                         "counters": [sum(b["counters"][k] for b in bl) for k in range(4)],
                         "sum_weights": [sum(b["sum_weights"][e] for b in bl) for e in range(ne)],
                         "n_ext": sum(b["n_ext"] for b in bl), "n_int": sum(b["n_int"] for b in bl),
                         "ext_weights": [sum(b["ext_weights"][e] for b in bl) for e in range(ne)],
                         "int_weights": [sum(b["int_weights"][e] for b in bl) for e in range(ne)]})
        doc["groups"].append({"name": g["name"], "energies": g["energies"], "amu": g["amu"], "scatf": g["scatf"], "runs": runs})
    with open(tiny or os.path.join(ROOT, "tests", "golden", "oracle_leak_seeds.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print("written", len(tasks), "blocks")


if __name__ == "__main__":
    main()
