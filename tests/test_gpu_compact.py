"""The compact image store (option "compact_images"): exit photons are written in the order of completion -- one coalesced
run per plane for the photons a wave finalises together, the north-star's compaction -- and the planes are published block
by block while the kernel runs.  The SET of photons must be the slot-ordered run's, bit for bit: with the slot-index plane
("slot_ids") the compact planes are a permutation of the slot-ordered ones.  Reference stores: src/polycap-source.c:779-798,
893-923."""
import os

import numpy as np
import pytest

from tests.conftest import EXAMPLE

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import polycap_amd
    assert polycap_amd.device_count() >= 1, "no HIP device visible: the GPU tests need an MI355X"
    return polycap_amd


def _run(ctx, seed, n, **opts):
    for k, v in opts.items():
        ctx.set_option(k, v)
    ctx.run(seed, 0, n, keep_images=True)
    planes = ctx.image_planes(0, n)            # before wait(): a compact run is fetched block by block behind the kernel
    ms = ctx.wait()
    t = ctx.totals()
    ids = ctx.slot_ids(0, n)
    return planes, ids, t, ms


def _check_permutation(ref, cmp, ids, n):
    assert np.array_equal(np.sort(ids), np.arange(n)), "every slot exactly once"
    assert np.array_equal(cmp["planes"], ref["planes"][:, ids], equal_nan=True)
    assert np.array_equal(cmp["exit_weights"], ref["exit_weights"][ids])
    assert np.array_equal(cmp["nrefl"], ref["nrefl"][ids])


@pytest.mark.parametrize("kernel", ["producer", "lane"])
def test_compact_planes_are_a_permutation_of_the_slot_ordered_planes(pa, kernel):
    prob = pa.problem_from_inp(os.path.join(EXAMPLE, "xos1.inp"), energies=[10.0])
    n = 700_001
    with pa.TraceContext(prob) as ctx:
        ctx.set_option("producer", 1 if kernel == "producer" else 0)
        ref, ids0, t0, _ = _run(ctx, 31, n, plane_images=1, compact_images=0)
        assert np.array_equal(ids0, np.arange(n))
        assert ctx.last_kernel() == ("pc_trace_producer_kernel" if kernel == "producer" else "pc_trace_kernel")
        for shift in (18, 10):
            cmp, ids, t1, _ = _run(ctx, 31, n, plane_images=1, compact_images=1, slot_ids=1, block_shift=shift)
            assert np.array_equal(t0["counters"][:4], t1["counters"][:4]) and np.array_equal(t0["sumw_fixed"], t1["sumw_fixed"])
            _check_permutation(ref, cmp, ids, n)
            # completion order is not slot order (else nothing was compacted), but early slots do tend to finish early
            assert not np.array_equal(ids, np.arange(n)) and np.median(ids[: n // 10]) < n // 4
        # a sub-range of positions, fetched after the run
        ctx.wait()
        part = ctx.image_planes(1000, 5000)
        assert np.array_equal(part["planes"], cmp["planes"][:, 1000:6000], equal_nan=True)
        # without the slot-index plane the run works the same (ids then read as the identity: nothing recorded)
        cmp2, _, t2, _ = _run(ctx, 31, n, plane_images=1, compact_images=1, slot_ids=0)
        assert np.array_equal(t0["sumw_fixed"], t2["sumw_fixed"])
        o1, o2 = np.argsort(ref["planes"][16], kind="stable"), np.argsort(cmp2["planes"][16], kind="stable")
        assert np.array_equal(ref["planes"][:, o1], cmp2["planes"][:, o2], equal_nan=True)


@pytest.mark.parametrize("n_energies", [3, 12, 40])
def test_compact_store_with_several_energies(pa, n_energies):
    """register-weight kernels (3 energies) and the any-n_energies kernel (12: immediate sweeps, 40: batched flat sweep)"""
    E = np.linspace(5.0, 25.0, n_energies)
    prob = pa.problem_from_inp(os.path.join(EXAMPLE, "xos1.inp"), energies=E)
    n = 150_000
    with pa.TraceContext(prob) as ctx:
        ref, _, t0, _ = _run(ctx, 5, n, plane_images=1, compact_images=0)
        cmp, ids, t1, _ = _run(ctx, 5, n, plane_images=1, compact_images=1, slot_ids=1, block_shift=12)
    assert np.array_equal(t0["counters"][:4], t1["counters"][:4]) and np.array_equal(t0["sumw_fixed"], t1["sumw_fixed"])
    _check_permutation(ref, cmp, ids, n)
    assert cmp["exit_weights"].shape == (n, n_energies)


def test_compact_store_on_short_lived_photons_and_small_runs(pa):
    """cone.inp (photons hardly reflect: the lane kernel, NEW phases dominate) and runs smaller than a block / a wave"""
    prob = pa.problem_from_inp(os.path.join(EXAMPLE, "cone.inp"), energies=[10.0])
    with pa.TraceContext(prob) as ctx:
        for n in (1, 37, 5000, 300_000):
            ref, _, t0, _ = _run(ctx, 9, n, plane_images=1, compact_images=0)
            cmp, ids, t1, _ = _run(ctx, 9, n, plane_images=1, compact_images=1, slot_ids=1, block_shift=8)
            assert np.array_equal(t0["sumw_fixed"], t1["sumw_fixed"])
            _check_permutation(ref, cmp, ids, n)
