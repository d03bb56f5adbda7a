import os, sys, json
sys.path.insert(0, '.')
import numpy as np
os.environ["POLYCAP_SEED"] = "31"; os.environ["POLYCAP_RCCL"] = "0"; os.environ["POLYCAP_OPTCONST"] = "builtin"
from polycap_amd import capi
prof = capi.Profile(capi.Profile.ELLIPSOIDAL, 9., 0.2065, 0.0585, 0.00035, 9.9153e-5, 1000., 0.5)
desc = capi.Description(prof, 0.0, 200000, {"O": 53.0, "Si": 47.0}, 2.23)
src_args = [2000.0, 0.2065, 0.2065, -1.0, 0.0, 0.0, 0.0, 0.5]
def run(dev):
    if dev is None: os.environ.pop("POLYCAP_HIP_DEVICES", None)
    else: os.environ["POLYCAP_HIP_DEVICES"] = dev
    src = capi.Source(desc, *src_args, np.array([10.0, 17.0]))
    eff = src.get_transmission_efficiencies(-1, 3001, leak_calc=True)
    out = []
    for lst in (eff.extleak_data, eff.intleak_data):
        out.append(np.array([list(l.coords) + list(l.direction) + list(l.elecv) + [l.n_refl] + list(l.weight) for l in lst]))
    return out
a = run(None); b = run("0,0")
for k in (0, 1):
    print("kind", k, a[k].shape, b[k].shape)
    n = min(len(a[k]), len(b[k]))
    d = np.where(~np.all((a[k][:n] == b[k][:n]) | (np.isnan(a[k][:n]) & np.isnan(b[k][:n])), axis=1))[0]
    print(" differing rows", len(d), d[:10])
    for i in d[:3]:
        print(a[k][i]); print(b[k][i])
    # same multiset?
    sa = a[k][np.lexsort(a[k].T[::-1])]; sb = b[k][np.lexsort(b[k].T[::-1])]
    print(" same as multisets:", sa.shape == sb.shape and np.array_equal(sa, sb, equal_nan=True))
