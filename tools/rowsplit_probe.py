#!/usr/bin/env python3
"""Per-kernel view of the native row split (gip_rowsplit_*): N tiles of the uhd4k frame, every tile's phases run alone (run_serial),
for `rocprofv3 --kernel-trace --stats -- python3 tools/rowsplit_probe.py [tiles] [iters] [serial|concurrent]`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mode = sys.argv[3] if len(sys.argv) > 3 else "serial"
wl = bench.WORKLOADS["uhd4k"]
bench.W, bench.H, bench.NDISP, bench.BLOCK, bench.ENCODING = wl["W"], wl["H"], wl["NDISP"], wl["BLOCK"], wl["ENCODING"]
import torch  # noqa: E402
from ros_gpu_stereo_processor_amd import capi, rowsplit as RS  # noqa: E402

lraw, rraw, calib = bench.make_inputs(2, seed0=100, nseeds=2)
rs = RS.RowSplit(bench.W, bench.H, calib, capi.BMParams(**bench.bm_kwargs()), tiles, "mono8", None, bench.SPECKLE_SIZE, bench.SPECKLE_DIFF)
L, R = torch.from_numpy(lraw[0]).cuda(), torch.from_numpy(rraw[0]).cuda()
for t in range(tiles):
    rs.set_inputs(t, L, R)
runs = []
for i in range(iters):
    if mode == "serial":
        runs.append(rs.run_serial())
    else:
        rs.run()
        rs.wait()
if runs:
    for t in range(tiles):
        print("tile", t, {k: round(sorted(r[t][k] for r in runs)[len(runs) // 2], 4) for k in runs[0][t]})
rs.close()
