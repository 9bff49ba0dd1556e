#!/usr/bin/env python3
"""Where does a pair's submit time go?  Times the ctypes call gip_streamer_submit per pair (fixed batches of 64, one compute lane) and a plain
numpy copy of the same bytes into pinned memory, on the same box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

bench.set_workload(bench.WORKLOADS["kitti"])
import torch  # noqa: E402
from ros_gpu_stereo_processor_amd import capi  # noqa: E402
from ros_gpu_stereo_processor_amd import processor as P  # noqa: E402
from ros_gpu_stereo_processor_amd.streamer import Streamer  # noqa: E402

lraw, rraw, calib = bench.make_inputs(8, seed0=100, nseeds=8)
W, H = bench.W, bench.H
proc = P.GpuStereoProcessor(0)
proc.initStereoModel(P.camera_info(W, H, **calib["left"]), P.camera_info(W, H, **calib["right"]))
proc.setParams(capi.BMParams(**bench.bm_kwargs()))
proc.setMaxSpeckleSize(800)
proc.setMaxSpeckleDiff(5.0)
pin = torch.empty((64, H, W), dtype=torch.uint8).pin_memory().numpy()
t0 = time.perf_counter()
for i in range(2048):
    np.copyto(pin[i % 64], lraw[i % 8])
dt = time.perf_counter() - t0
print("numpy copy into pinned: %.1f us per image, %.1f GB/s" % (1e6 * dt / 2048, 2048 * W * H / dt / 1e9))
for lanes in (1, 2):
    st = Streamer(proc, W, H, "mono8", capi.TOPIC_DISPARITY, frames_per_batch=64, adaptive=False, compute_lanes=lanes)
    for i in range(256):
        st.submit(lraw[i % 8], rraw[i % 8])
    st.flush()
    L = capi.lib()
    ts = []
    t0 = time.perf_counter()
    for i in range(64 * 40):
        a, b = lraw[i % 8], rraw[i % 8]
        t1 = time.perf_counter()
        L.gip_streamer_submit(st._s, a.ctypes.data, b.ctypes.data, 0, None)
        ts.append(time.perf_counter() - t1)
    st.flush()
    dt = time.perf_counter() - t0
    ts = np.array(ts) * 1e6
    full = ts.reshape(40, 64)
    print("lanes %d: %.0f pairs/s; submit call us: median %.1f mean %.1f p99 %.1f; the launching call (every 64th): median %.1f; the other 63: mean %.1f" %
          (lanes, 64 * 40 / dt, np.median(ts), ts.mean(), np.percentile(ts, 99), np.median(full[:, 63]), full[:, :63].mean()))
    st.close()
proc.close()
