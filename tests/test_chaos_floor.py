"""Why per-photon bit parity is not the bar: the reference's trace amplifies rounding differences.

Measured on the oracle alone (the reference's own algorithm, no GPU code involved): moving one start coordinate
by a single ulp (~2e-17 cm) changes the return code or the reflection count of more than 5 % of the photons and
shifts the transmitted weight by ~0.25/sqrt(N) relative.  Any two fp64 implementations that differ by rounding
(another compiler, another libm, FMA contraction, a GPU) therefore agree per photon only over the first few
reflections and agree statistically afterwards.  The GPU parity tests use c/sqrt(N) tolerances with c = 1.0,
i.e. a few times this self-noise."""
import numpy as np

from tests.common import make_pair


def test_one_ulp_perturbation_of_the_oracle(oracle):
    optic, src, prob, (E, A, S) = make_pair(oracle, "xos1", source=(2000., 0.2065, 0.2065, 0., 0., 0., 0., 0.0))
    n = 60000
    ph = oracle.sample_photons(optic, src, 20000, np.arange(n))
    a = oracle.launch_batch(optic, E, A, S, ph[:, 0:3], ph[:, 3:6], ph[:, 6:9])
    st = ph[:, 0:3].copy()
    st[:, 0] = np.nextafter(st[:, 0], 1.0)
    b = oracle.launch_batch(optic, E, A, S, st, ph[:, 3:6], ph[:, 6:9])
    flips = ((a["rc"] != b["rc"]) | (a["i_refl"] != b["i_refl"])).mean()
    assert 0.05 < flips < 0.30, flips
    # entrance decisions are stable, and photons with few reflections are still close
    ent = np.isin(a["rc"], (2, -2))
    assert np.array_equal(a["rc"][ent], b["rc"][ent])
    short = (a["rc"] == b["rc"]) & (a["i_refl"] == b["i_refl"]) & (a["i_refl"] <= 3) & np.isin(a["rc"], (0, 1))
    assert np.abs(a["exit_coords"][short] - b["exit_coords"][short]).max() < 1e-6
    sa, sb = a["weights"][a["rc"] == 1, 0].sum(), b["weights"][b["rc"] == 1, 0].sum()
    assert abs(sa - sb) / sa < 1.0 / np.sqrt(n)
