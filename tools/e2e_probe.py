#!/usr/bin/env python3
"""End-to-end streaming records for a list of (batching, frames_per_batch, compute_lanes) settings: python3 tools/e2e_probe.py adaptive:0:2 fixed:64:1 ..."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

modes = [(m.split(":")[0], int(m.split(":")[1]), int(m.split(":")[2])) for m in sys.argv[1:]] or [("adaptive", 0, 2)]
bench.E2E_MODES = modes
wl = bench.WORKLOADS["kitti"]
bench.set_workload(wl)
from ros_gpu_stereo_processor_amd import capi  # noqa: E402

lraw, rraw, calib = bench.make_inputs(8, seed0=100, nseeds=8)
out = bench.e2e_streaming(lraw, rraw, calib, capi.BMParams(**bench.bm_kwargs()), 0, budget_s=float(os.environ.get("E2E_BUDGET", "4.0")))
print(json.dumps(out["measured_link_gbs"]))
for r in out["records"]:
    print(json.dumps({k: r[k] for k in ("what", "batching", "max_pairs_per_launch", "compute_lanes", "avg_pairs_per_launch", "pairs_per_s", "frac_of_pcie_bound", "frac_of_measured_link")}))
