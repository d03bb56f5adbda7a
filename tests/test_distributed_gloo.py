"""N>1 path on CPU: two gloo ranks shard the slot range, each traces its share (here with the oracle standing in
for the GPU kernel -- only the sharding / packing / all-reduce logic of polycap_amd.distributed is under test),
and the reduced totals must equal the single-rank result bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

from tests.conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_trace_fn():
    """trace_fn for run_sharded: oracle driver + exact fixed-point sums built from the per-slot weights."""
    from oracle import pyoracle as O

    def fn(problem, seed, slot0, n, keep_images, max_attempts):
        optic = O.Optic(problem.z, problem.cap, problem.ext, problem.sig_rough, problem.n_cap, problem.density)
        src = O.make_source(*problem.source)
        r = O.transmission(optic, src, problem.energies, problem.amu, problem.scatf, seed, slot0, n, n_threads=2, images=True)
        ne = problem.n_energies
        fx = np.zeros((ne, 2), dtype=np.uint64)
        for e in range(ne):
            tot = sum(int(w * 4611686018427387904.0) for w in r["exit_weights"][:, e])   # same 2^62 truncation as the kernel
            fx[e, 0] = np.uint64(tot & (2**64 - 1))
            fx[e, 1] = np.uint64(tot >> 64)
        cnt = np.zeros(6, dtype=np.int64)
        cnt[:4] = r["counters"]
        return dict(counters=cnt, sumw_fixed=fx)
    return fn


def _worker(rank, world, port, n_total, seed, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from polycap_amd import distributed as pcd
    from tests.common import make_pair
    from oracle import pyoracle as O
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        _, _, prob, _ = make_pair(O, "ellip", energies=(8.0, 10.0))
        r = pcd.run_sharded(prob, seed, n_total, rank=rank, world_size=world, trace_fn=_oracle_trace_fn())
        rep = pcd.ranks_report(10.0 + rank)             # what bench.py prints as "rccl" under N > 1
        out_q.put((rank, r["counters"].tolist(), [str(v) for v in r["sumw_exact"]], r["efficiencies"].tolist(), r["slot0"], r["n_local"], rep))
    finally:
        dist.destroy_process_group()


def test_shard_slots_partition():
    from polycap_amd.distributed import shard_slots
    for n, w in ((10, 3), (7, 8), (1000003, 8), (5, 1)):
        ranges = [shard_slots(n, w, r) for r in range(w)]
        assert ranges[0][0] == 0 and sum(c for _, c in ranges) == n
        for (a, ca), (b, _) in zip(ranges, ranges[1:]):
            assert a + ca == b
        assert max(c for _, c in ranges) - min(c for _, c in ranges) <= 1


def test_pack_unpack_roundtrip_exact():
    from polycap_amd.distributed import pack_totals, unpack_totals
    rng = np.random.default_rng(1)
    fx = rng.integers(0, 2**63, size=(5, 2), dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    cnt = np.array([3, 1, 4, 1, 5, 9], dtype=np.int64)
    vec = pack_totals(cnt, fx)
    vec2 = vec + pack_totals(cnt, fx)              # two ranks
    c, sums, exact = unpack_totals(vec2)
    assert c.tolist() == (2 * cnt).tolist()
    for e in range(5):
        assert exact[e] == 2 * (int(fx[e, 0]) + (int(fx[e, 1]) << 64))
    assert np.all(sums > 0)


def test_two_rank_gloo_equals_single_rank():
    import torch.multiprocessing as mp
    from polycap_amd import distributed as pcd
    from tests.common import make_pair
    from oracle import pyoracle as O
    n_total, seed = 601, 77
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, seed, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _, _, prob, _ = make_pair(O, "ellip", energies=(8.0, 10.0))
    single = pcd.run_sharded(prob, seed, n_total, trace_fn=_oracle_trace_fn())
    for rank, counters, exact, eff, slot0, n_local, rep in res:
        # the "rccl" block of a multi-rank bench line (bench.py under N > 1): every rank saw both ranks and the spread of their
        # kernel times
        assert rep == {"world_size": 2, "backend": "gloo", "ranks_seen": 2, "per_rank_kernel_ms": [10.0, 11.0]}
        assert counters == single["counters"].tolist()
        assert exact == [str(v) for v in single["sumw_exact"]]
        assert eff == single["efficiencies"].tolist()
    assert [r[4] for r in res] == [0, 301] and [r[5] for r in res] == [301, 300]
    assert single["counters"][0] == n_total


def _gpu_worker(rank, world, port, n_total, seed, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from polycap_amd import distributed as pcd
    from tests.common import make_pair
    from oracle import pyoracle as O
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        _, _, prob, _ = make_pair(O, "xos1", source=(2000., 0.2065, 0.2065, 0., 0., 0., 0., 0.0))
        r = pcd.run_sharded(prob, seed, n_total, rank=rank, world_size=world, device_index=0)     # the HIP kernel
        out_q.put((rank, r["counters"].tolist(), [str(v) for v in r["sumw_exact"]], r["efficiencies"].tolist(), r["slot0"], r["n_local"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_with_the_hip_kernel_equal_one_rank():
    """The N > 1 path with the real kernel: two processes (gloo; they share the box's one GPU) trace their slot shares
    with the HIP kernel and all-reduce the packed totals; the result equals the single-rank run bit for bit."""
    import torch.multiprocessing as mp
    from polycap_amd import distributed as pcd
    from tests.common import make_pair
    from oracle import pyoracle as O
    n_total, seed = 400_001, 31
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, n_total, seed, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    _, _, prob, _ = make_pair(O, "xos1", source=(2000., 0.2065, 0.2065, 0., 0., 0., 0., 0.0))
    single = pcd.run_sharded(prob, seed, n_total, device_index=0)
    for rank, counters, exact, eff, slot0, n_local in res:
        assert counters[:4] == single["counters"][:4].tolist()
        assert exact == [str(v) for v in single["sumw_exact"]]
        assert eff == single["efficiencies"].tolist()
    assert [r[4] for r in res] == [0, 200_001] and [r[5] for r in res] == [200_001, 200_000]


@pytest.mark.gpu
def test_bench_under_the_launcher_with_rccl_world_of_one():
    """bench.py exactly as the driver starts it for N > 1 (python -m torch.distributed.run ... bench.py --gpus N), with the
    one rank a one-GPU box allows: process group on RCCL, the all-reduce of the packed totals on the device, the barrier."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--photons", "300000", "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 1e7 and d["roofline"]["kernel_ms"] > 0
