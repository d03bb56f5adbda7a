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


def test_multi_gpu_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` typed as is starts the two ranks itself (a fresh torch.distributed.run child, before
    anything touches the GPU) and returns their status: on a box without a GPU both ranks refuse loudly."""
    import polycap_amd
    if polycap_amd.device_count() > 0:
        pytest.skip("a HIP device is visible")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=900, env=env)
    out = r.stderr + r.stdout
    assert r.returncode != 0 and "no HIP device" in out
    assert not any(line.startswith("{") for line in r.stdout.splitlines())   # no JSON line that could pass for a measurement


def test_committed_counters_are_readable():
    sys.path.insert(0, ROOT)
    import bench
    assert os.path.exists(bench.PMC_SUMMARY)
    with open(bench.PMC_SUMMARY) as f:
        kernel = json.load(f)["meta"]["kernel"].split("void ")[-1].split("<")[0]
    s = bench.pmc_summary(10_000_000, True, kernel)
    for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "FETCH_SIZE", "WRITE_SIZE"):
        assert s[k] > 0
    t = (s["FETCH_SIZE"] + s["WRITE_SIZE"]) * 1024.0
    assert 1.4e9 < t < 2.9e9                 # algorithmic 1.44 GB per launch; measured 1.68 GB with the compact store (round 2: 7.95 GB)
    # the summary belongs to the default command only: other sizes, histogram-only runs and the lane kernel get none
    assert bench.pmc_summary(1000, True, kernel) is None and bench.pmc_summary(10_000_000, False, kernel) is None
    assert bench.pmc_summary(10_000_000, True, "pc_trace_pool_kernel") is None and bench.valu_issue(None, 27.0) is None
    v = bench.valu_issue(s, 27.0)
    assert 0.5 < v["frac"] < 1.0 and 0.3 < v["lane_utilisation"] < 0.8
    # the vector pipes' busy share from the counters alone (no clock frequency): the headline kernel saturates them
    assert 0.85 < v["simd_valu_busy"] < 1.0
    for name, lo, hi in (("ne291", 0.7, 0.9), ("ellip291", 0.65, 0.9), ("leak", 0.2, 0.5)):
        with open(os.path.join(ROOT, "profiles", "r04", name + "_pmc_summary.json")) as f:
            assert lo < bench.simd_valu_busy(json.load(f)) < hi, name


def test_host_cpu_description():
    sys.path.insert(0, ROOT)
    import bench
    n, quota, model = bench.host_cpus()
    assert n >= 1 and (quota is None or quota > 0) and isinstance(model, str) and model


def test_every_committed_summary_names_the_kernel_its_leg_claims():
    """Round 3's leak summary described the 4.8 ms pre-pass kernel (the summariser picked the kernel by dispatch count).  The
    summaries bench.py reads are produced with `scripts/summarize_profile.py --kernel <name>` now, and each must be of the kernel
    that runs the leg; bench.pmc_block refuses a summary of another kernel instead of quoting it."""
    sys.path.insert(0, ROOT)
    import bench
    want = {"headline": "pc_trace_producer_kernel<0, false>", "ne291": "pc_trace_log_kernel<", "ellip291": "pc_trace_log_kernel<",
            "leak": "pc_leak_kernel<"}
    for tag, name in want.items():
        path = os.path.join(ROOT, "profiles", "r04", tag + "_pmc_summary.json")
        with open(path) as f:
            d = json.load(f)
        assert name in d["meta"]["kernel"], (tag, d["meta"]["kernel"])
        for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"):
            assert d[k] > 0, (tag, k)
    b = bench.pmc_block("profiles/r04/leak_pmc_summary.json", "pc_leak_kernel", 180.0)
    assert b["kernel"].startswith("void pc_leak_kernel<") and 0.05 < b["frac"] < 1.0 and 0.1 < b["lane_utilisation"] <= 1.0
    assert "mismatch" in bench.pmc_block("profiles/r04/leak_pmc_summary.json", "pc_trace_kernel", 180.0)
    assert "missing" in bench.pmc_block("profiles/r04/no_such_summary.json", "pc_leak_kernel", 180.0)
    # the many-energy kernel: rows no longer stream through HBM (round 3: 47 GB per 1e6-slot launch)
    b = bench.pmc_block("profiles/r04/ne291_pmc_summary.json", "pc_trace_log_kernel", 28.0)
    assert b["hbm_traffic_bytes_per_launch"] < 8e9 and b["wait_share"] < 0.35


def test_summariser_fails_on_a_missing_kernel(tmp_path):
    """scripts/summarize_profile.py wants the kernel by name and exits non-zero when no dispatch of it was profiled."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "summarize_profile.py"), "no_such_tag", "r04"], capture_output=True, text=True)
    assert r.returncode != 0 and "--kernel" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "summarize_profile.py"), "no_such_tag", "r04", "--kernel", "pc_leak_kernel"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "no counter passes" in (r.stderr + r.stdout)


def test_committed_bench_line_has_the_contract_keys():
    """profiles/r04/headline_bench.json is the line `python bench.py` printed on the MI355X box of the round's final evidence run:
    the keys the driver reads, the two objects the tier asks for, and the extra legs DESIGN section 7 quotes."""
    with open(os.path.join(ROOT, "profiles", "r04", "headline_bench.json")) as f:
        d = json.load(f)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["traffic"] > 0
    assert r["kernel_ms"] <= d["ms_per_step"]          # a step is the kernel + totals + reduce
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    for leg in ("wall_incl_copyback", "sweep_291", "sweep_300", "ellip_l9_rough", "leak_262144", "parity_fixture"):
        assert leg in d, leg
    assert d["parity_fixture"]["within_tolerance"] and d["parity_fixture"]["eff_rel_delta_pooled"] <= 1e-4
    assert d["leak_262144"]["n_exit"] == 262144 and d["leak_262144"]["started_photons_per_s"] > 2e6
    assert d["leak_262144"]["kernel"] == "pc_leak_kernel" and "pc_leak_kernel<" in d["leak_262144"]["valu_issue"]["kernel"]
    assert d["leak_262144"]["cpu_baseline"]["value"] > 0
    for leg in (d["sweep_291"], d["sweep_300"], d["ellip_l9_rough"]["n_energies_291"]):
        assert leg["kernel"] == "pc_trace_log_kernel"
    assert d["sweep_291"]["valu_issue"]["kernel"].startswith("void pc_trace_log_kernel<") and d["sweep_300"]["n_energies"] == 300
