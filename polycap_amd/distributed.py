"""One process per GPU: shard the exit-photon slot range, trace locally, reduce the per-energy histogram.

The path shards into independent units (slot j depends on nothing but (seed, j)), so there is no data-path
collective; the only exchange is ONE all-reduce (RCCL over xGMI, backend "nccl") of n_energies + 6 numbers at the
end -- the MI355X counterpart of the reference's `omp critical` sum (src/polycap-source.c:973-980).  Weight sums
travel as exact 128-bit fixed point split into 32-bit limbs (each limb sum stays far below 2^63), so the reduced
result is bit-identical for every world size.  Image planes stay sharded on the GPU that produced them.
"""
import numpy as np

_LIMB = (1 << 32) - 1


def shard_slots(n_total, world_size, rank):
    """Contiguous slot range [slot0, slot0+n) of `rank`; ranges differ by at most one slot."""
    base, extra = divmod(int(n_total), int(world_size))
    n = base + (1 if rank < extra else 0)
    slot0 = rank * base + min(rank, extra)
    return slot0, n


def pack_totals(counters, sumw_fixed):
    """int64 vector: 6 counters, then 4 limbs (32 bit each, little endian) per energy."""
    counters = np.asarray(counters, dtype=np.int64)
    fx = np.asarray(sumw_fixed, dtype=np.uint64).reshape(-1, 2)
    limbs = np.zeros((fx.shape[0], 4), dtype=np.int64)
    limbs[:, 0] = (fx[:, 0] & np.uint64(_LIMB)).astype(np.int64)
    limbs[:, 1] = (fx[:, 0] >> np.uint64(32)).astype(np.int64)
    limbs[:, 2] = (fx[:, 1] & np.uint64(_LIMB)).astype(np.int64)
    limbs[:, 3] = (fx[:, 1] >> np.uint64(32)).astype(np.int64)
    return np.concatenate([counters[:6], limbs.ravel()])


def unpack_totals(vec):
    """-> (counters[6], sum_weights[n_energies] as float, exact integer sums as python ints in units of 2^-62)"""
    vec = np.asarray(vec, dtype=np.int64)
    counters = vec[:6].copy()
    limbs = vec[6:].reshape(-1, 4)
    exact = [int(l[0]) + (int(l[1]) << 32) + (int(l[2]) << 64) + (int(l[3]) << 96) for l in limbs]
    from fractions import Fraction
    sums = np.array([float(Fraction(v, 1 << 62)) for v in exact], dtype=np.float64)
    return counters, sums, exact


def allreduce_totals(vec, device=None):
    """Sum the packed totals over all ranks (no-op without an initialised process group)."""
    import os
    import sys
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 and "torch.distributed" not in sys.modules:
        return np.asarray(vec, dtype=np.int64)      # single process: do not even import torch
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(vec, dtype=np.int64)
    t = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.int64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def ranks_report(mean_kernel_ms, device=None):
    """What a multi-rank bench line says about the job itself: how many ranks the collective saw, over which backend, and the
    spread of the ranks' mean kernel times (one all-gather outside the timed region).  None without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    world = dist.get_world_size()
    mine = torch.tensor([float(mean_kernel_ms)], dtype=torch.float64, device=device if device is not None else "cpu")
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    per_rank = [float(x.item()) for x in every]
    return {"world_size": world, "backend": dist.get_backend(), "ranks_seen": len(per_rank),
            "per_rank_kernel_ms": [min(per_rank), max(per_rank)]}


def efficiencies_from_totals(counters, sums):
    """Efficiency formula of the reference (src/polycap-source.c:1066-1076)."""
    iexit, not_entered, not_trans = int(counters[0]), int(counters[1]), int(counters[2])
    open_area = float(iexit + not_trans) / float(iexit + not_entered + not_trans)
    return (np.asarray(sums, dtype=np.float64) / (float(iexit) + float(not_trans))) * open_area


def run_sharded(problem, seed, n_slots_total, rank=0, world_size=1, device_index=0, keep_images=False,
                max_attempts=1 << 20, trace_fn=None, reduce_device=None, leak_calc=False):
    """Trace this rank's share of [0, n_slots_total) and all-reduce the totals.  With leak_calc the leak events of the
    rank's slots stay with the rank (local["ext"], local["int"], tagged with global slot numbers: concatenating the ranks'
    arrays in rank order gives the single-device lists).

    trace_fn(problem, seed, slot0, n, keep_images, max_attempts) -> dict(counters, sumw_fixed, ...) replaces the
    HIP context in CPU-only tests of the sharding/reduction logic; the default is the GPU path."""
    slot0, n = shard_slots(n_slots_total, world_size, rank)
    local = None
    if n > 0:
        if trace_fn is None:
            from .hip import TraceContext
            with TraceContext(problem, device_index) as ctx:
                local = ctx.transmission(seed, slot0, n, max_attempts=max_attempts, keep_images=keep_images, leak_calc=leak_calc)
        else:
            local = trace_fn(problem, seed, slot0, n, keep_images, max_attempts)
        vec = pack_totals(local["counters"], local["sumw_fixed"])
    else:
        vec = np.zeros(6 + 4 * problem.n_energies, dtype=np.int64)
    vec = allreduce_totals(vec, reduce_device)
    counters, sums, exact = unpack_totals(vec)
    return dict(counters=counters, sum_weights=sums, sumw_exact=exact, efficiencies=efficiencies_from_totals(counters, sums),
                slot0=slot0, n_local=n, local=local)
