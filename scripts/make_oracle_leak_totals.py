#!/usr/bin/env python3
"""Fixture: totals of the CPU oracle's leak driver (oracle/polycap_oracle_leak.c, the reference's literal algorithm:
src/polycap-source.c:744-1087 with leak_calc, polycap_capil_trace_wall) on the reference's test optic, uniform illumination,
10 keV, seed 20000 -- slots [0, n) in blocks, each block a process of its own.
    python scripts/make_oracle_leak_totals.py [n, default 8000] [block, default 1000]   ->  tests/golden/oracle_leak_totals.json
(47 ms per slot: 8000 slots are 6 CPU-minutes.)"""
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
from tests.test_oracle_leak_known_answers import constants
known = json.load(open(os.path.join(GOLDEN, "reference_known_answers.json")))
leaks = json.load(open(os.path.join(GOLDEN, "reference_leak_known_answers.json")))
t = known["test_optic"]
optic = pyoracle.Optic.from_shape(t["type"], t["length"], t["rad_ext_upstream"], t["rad_ext_downstream"], t["rad_int_upstream"],
                                  t["rad_int_downstream"], t["focal_dist_upstream"], t["focal_dist_downstream"], t["sig_rough"],
                                  t["n_cap"], known["glass"]["density"])
amu, scatf = constants(leaks, 10)
src = (2000., 0.2065, 0.2065, -1., 0., 0., 0., 0.5)
o = pyoracle.transmission(optic, pyoracle.make_source(*src), [10.0], [amu], [scatf], 20000, %(slot0)d, %(n)d, leak_calc=True)
c = [int(x) for x in o["counters"][:4]]
print(json.dumps({"slot0": %(slot0)d, "n": %(n)d, "counters": c, "sum_weight": float(o["sum_weights"][0]),
                  "n_ext": int(len(o["ext"])), "n_int": int(len(o["int"])),
                  "ext_weight": float(o["ext"][:, 12].sum()) if len(o["ext"]) else 0.0,
                  "int_weight": float(o["int"][:, 12].sum()) if len(o["int"]) else 0.0}))
"""


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
    block = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    from oracle import pyoracle
    pyoracle.build()
    procs = []
    blocks = []
    for s0 in range(0, n, block):
        procs.append(subprocess.Popen([sys.executable, "-c", CHILD % {"root": ROOT, "slot0": s0, "n": min(block, n - s0)}],
                                      stdout=subprocess.PIPE, text=True))
        if len(procs) == 8 or s0 + block >= n:
            for p in procs:
                out, _ = p.communicate()
                if p.returncode != 0:
                    raise SystemExit("oracle block failed")
                blocks.append(json.loads(out.strip().splitlines()[-1]))
            procs = []
    doc = {"what": "oracle/polycap_oracle_leak.c, leak_calc=true driver: the reference's ellipsoidal test optic, uniform illumination "
                   "(2000 cm, 0.2065 x 0.2065, sigma -1), 10 keV, seed 20000; one entry per block of exit-photon slots",
           "seed": 20000, "source": [2000., 0.2065, 0.2065, -1., 0., 0., 0., 0.5], "energy_keV": 10.0, "blocks": blocks}
    with open(os.path.join(ROOT, "tests", "golden", "oracle_leak_totals.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({k: sum(b[k] for b in blocks) for k in ("n", "n_ext", "n_int", "sum_weight")}))


if __name__ == "__main__":
    main()
