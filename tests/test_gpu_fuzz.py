"""Random optics against the invariants of the trace kernels (tests/fuzz_optics.py): certified == literal march, kernel == host
compile, lane == pool == producer kernels, logged == immediate sweeps, device ~ oracle, and with leak_calc=true certified wall
search == literal stepping and kernel == host compile event for event -- on profiles, capillary counts, sources and
energy grids that no deck of the reference has (mono-capillaries, 7 ... 200000 capillaries, 100 ... 999 segments, bulging and
waisted optics, near divergent sources, roughness)."""
import pytest


@pytest.mark.gpu
def test_random_optics_keep_the_kernels_invariants():
    from tests import fuzz_optics
    lines = []
    bad = fuzz_optics.run(16, 20260405, out=lines.append)
    assert bad == 0, "\n".join(l for l in lines if not l.endswith("| OK"))
    assert sum(l.endswith("| OK") for l in lines) == 16
