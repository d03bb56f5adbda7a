"""bench.py on a box without a GPU: it must refuse loudly (the trace has no CPU fallback), and the helpers that read the
committed rocprofv3 summary must find it.  The timed path itself needs an MI355X (driver runs it at round end)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_refuses_without_a_device():
    import polycap_amd
    if polycap_amd.device_count() > 0:
        pytest.skip("a HIP device is visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "no HIP device" in (r.stderr + r.stdout)
    assert r.stdout.strip() == ""            # no JSON line that could be mistaken for a measurement


def test_multi_gpu_flag_needs_the_launcher():
    env = dict(os.environ, WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "torch.distributed.run" in (r.stderr + r.stdout)


def test_committed_counters_are_readable():
    sys.path.insert(0, ROOT)
    import bench
    assert os.path.exists(bench.PMC_SUMMARY)
    with open(bench.PMC_SUMMARY) as f:
        s = json.load(f)
    for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "FETCH_SIZE", "WRITE_SIZE"):
        assert s[k] > 0
    t = bench.measured_traffic(10_000_000, True)
    assert 1.4e9 < t < 4e9                   # algorithmic 1.44 GB per launch; measured 2.6 GB
    assert bench.measured_traffic(1000, True) is None and bench.measured_traffic(10_000_000, False) is None
    v = bench.valu_issue(10_000_000, True, 33.0)
    assert 0.5 < v["frac"] < 1.0 and 0.3 < v["lane_utilisation"] < 0.7
