#!/usr/bin/env python3
"""One of bench.py's side workloads by itself (the command the rocprofv3 passes of scripts/profile_r03.sh run):
    python3 scripts/side_workload.py sweep_291 | ellip_291 | ellip_1"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "sweep_291"
args = {"sweep_291": ("xos1", None, None, 1_000_000), "ellip_291": ("ellip_l9", None, 5.0, 500_000),
        "ellip_1": ("ellip_l9", [10.0], 5.0, 4_000_000)}[which]
print(json.dumps(bench.side_workload(*args, 0)))
