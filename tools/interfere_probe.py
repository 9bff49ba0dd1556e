#!/usr/bin/env python3
"""What slows the matcher down when something else runs next to it?  The SAD matcher (k_bm_c4, 1242x375, ndisp 128, block 9, B pairs) is launched
back to back on one stream while ONE kind of co-runner loops on a second stream (both from gip_concurrent_streams_create); printed per co-runner:
the matcher's time per launch against its time alone, and what the co-runner achieved meanwhile against ITS time alone.

    python3 tools/interfere_probe.py [B]          (one MI355X; about 20 s)
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ros_gpu_stereo_processor_amd import capi, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H, ND, BLK = 1242, 375, 128, 9
dev = torch.device("cuda:0")
L = capi.lib()
handles, nind = capi.concurrent_streams(2, 0)
sm, sc = [torch.cuda.ExternalStream(h, device=dev) for h in handles]
params = capi.BMParams(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=BLK, min_disparity=0, num_disparities=ND, texture_threshold=10,
                       uniqueness_ratio=15, disp12_max_diff=1, speckle_window_size=0, speckle_range=0)
calib = synth.scaled_calibration(W, H)
rng = np.random.default_rng(1)
pairs = [synth.raw_pair(W, H, ND, seed=10 + i, calib=calib) for i in range(4)]
lraw = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(B)])).to(dev)
rraw = torch.from_numpy(np.stack([pairs[i % 4][1] for i in range(B)])).to(dev)
from ros_gpu_stereo_processor_amd import kernels as K  # noqa: E402
from ros_gpu_stereo_processor_amd import processor as P  # noqa: E402
proc = P.GpuStereoProcessor(0)
proc.initStereoModel(P.camera_info(W, H, **calib["left"]), P.camera_info(W, H, **calib["right"]))
maps = []
for side in (capi.SIDE_L, capi.SIDE_R):
    m1, m2 = proc.rectifyMaps(side)
    maps += [torch.from_numpy(m1).to(dev), torch.from_numpy(m2.view(np.int16)).to(dev)]
model = proc.model()
rl, rr = K.remap(lraw, maps[0], maps[1]), K.remap(rraw, maps[2], maps[3])
pl0, pr0 = K.prefilter(rl, 1, 9, 31), K.prefilter(rr, 1, 9, 31)
WP = (W + 63) // 64 * 64                                  # the matcher's fast path wants rows that start on a 64-byte boundary (as in gip_batch_run)
pl = torch.zeros((B, H, WP), dtype=torch.uint8, device=dev)
pr = torch.zeros((B, H, WP), dtype=torch.uint8, device=dev)
pl[:, :, :W] = pl0
pr[:, :, :W] = pr0
disp = torch.empty((B, H, W), dtype=torch.int16, device=dev)
cost = torch.zeros((B, H, W), dtype=torch.int32, device=dev)
torch.cuda.synchronize()


def matcher():
    rc = L.gipk_bm_match(pl.data_ptr(), pr.data_ptr(), WP, W, H, C.byref(params), disp.data_ptr(), W * 2, cost.data_ptr(), W * 4, B, H * WP, H * W * 2, H * W * 4,
                         C.c_void_p(sm.cuda_stream))
    assert rc == 0, rc


# ---- co-runners: every one works on buffers of its own (the matcher's inputs are only read)
d2 = torch.empty_like(disp)
c2 = torch.zeros_like(cost)
rect2 = torch.empty_like(rl)
pf2 = torch.empty_like(rl)
f32 = torch.empty((B, H, W), dtype=torch.float32, device=dev)
pc = torch.empty((B, H, W, 32), dtype=torch.uint8, device=dev)
spk = torch.empty(int(L.gipk_speckle_scratch_bytes(W, H, B)), dtype=torch.uint8, device=dev)
big_a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
big_b = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
valu_x = torch.rand(1 << 24, device=dev) + 1.0
valu_y = torch.empty_like(valu_x)
Q = K._q(np.asarray(model["Q"], np.float64).reshape(4, 4))
st = C.c_void_p(sc.cuda_stream)
matcher()
torch.cuda.synchronize()
d_final = disp.clone()
L.gipk_bm_finalize(d_final.data_ptr(), W * 2, cost.data_ptr(), W * 4, W, H, C.byref(params), B, H * W * 2, H * W * 4, None)
torch.cuda.synchronize()


def co_remap():
    L.gipk_remap_u8_interp(lraw.data_ptr(), W, W, H, rect2.data_ptr(), W, W, H, 1, maps[0].data_ptr(), maps[1].data_ptr(), capi.INTER_LINEAR, B, H * W, H * W, st)


def co_prefilter():
    L.gipk_prefilter(rl.data_ptr(), W, pf2.data_ptr(), W, W, H, 1, 9, 31, B, H * W, H * W, st)


def co_finalize():
    d2.copy_(disp)                                       # (the copy is part of this co-runner: 2 B/px read + write)
    L.gipk_bm_finalize(d2.data_ptr(), W * 2, cost.data_ptr(), W * 4, W, H, C.byref(params), B, H * W * 2, H * W * 4, st)


def co_speckle():
    d2.copy_(d_final)
    L.gipk_filter_speckles(d2.data_ptr(), W * 2, W, H, -16, 800, 80, spk.data_ptr(), spk.numel(), B, H * W * 2, st)


def co_outputs():
    L.gipk_disparity_outputs_rows(d_final.data_ptr(), W * 2, rl.data_ptr(), W, 1, f32.data_ptr(), pc.data_ptr(), W, H, 0, Q, float(model["cx_l"]), float(model["cx_r"]), 0, B,
                                  H * W * 2, H * W, st)


def co_copy():
    big_b.copy_(big_a)


def co_fill():
    big_b.zero_()


def co_read():
    big_a.view(torch.int32).sum()


def co_valu():
    torch.lgamma(valu_x, out=valu_y)


CO = [("rectify (k_remap_r4)", co_remap), ("XSOBEL prefilter", co_prefilter), ("copy + L/R check (k_bm_finalize_r, 64-bit keys)", co_finalize),
      ("copy + speckle filter (5 kernels)", co_speckle), ("f32 + cloud (k_disp_outputs_sq, arithmetic form)", co_outputs),
      ("1 GiB device copy (torch)", co_copy), ("1 GiB fill (torch)", co_fill), ("1 GiB read-reduce (torch)", co_read), ("lgamma on 16 M floats (torch, VALU-bound)", co_valu)]


def timed_matcher(n, co=None):
    """n matcher launches back to back; the co-runner is kept queued a few launches ahead on its stream until the matchers are done"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    done = torch.cuda.Event()
    iters = 0
    if co is not None:
        with torch.cuda.stream(sc):
            for _ in range(3):
                co(); iters += 1
    e0.record(sm)
    for _ in range(n):
        matcher()
    e1.record(sm)
    done.record(sm)
    if co is not None:
        with torch.cuda.stream(sc):
            t_end = time.time() + 20
            while not done.query() and time.time() < t_end:
                co(); iters += 1
                if iters % 4 == 0:
                    # do not run ahead of the GPU by more than a few launches
                    ev = torch.cuda.Event(); ev.record(sc)
                    while not ev.query() and not done.query():
                        time.sleep(0.0001)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, iters


def co_alone(co, n=8):
    with torch.cuda.stream(sc):
        co()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            co()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3


N = max(6, 1536 // B)
timed_matcher(3)
alone, _ = timed_matcher(N)
print("matcher alone: %.3f ms per launch of %d pairs (independent streams: %d)" % (alone, B, nind))
print("%-52s %10s %9s %12s %14s" % ("co-runner", "matcher ms", "stretch", "co alone ms", "co next to it"))
for name, co in CO:
    ca = co_alone(co)
    t0 = time.perf_counter()
    ms, iters = timed_matcher(N, co)
    wall = (time.perf_counter() - t0) * 1e3
    print("%-52s %10.3f %8.1f%% %12.3f %11.3f ms" % (name, ms, 100 * (ms / alone - 1), ca, wall / max(iters, 1)))
