#!/usr/bin/env python3
"""A/B timing of gipk_filter_speckles from several library builds in ONE process on the same disparity maps."""
import ctypes as C, sys, os, statistics
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ros_gpu_stereo_processor_amd import capi, synth, kernels as K

W, H, B = 1242, 375, 64
paths = [a for a in sys.argv[1:] if a.endswith(".so")]
libs = [C.CDLL(os.path.abspath(p)) for p in paths]
for L in libs:
    L.gipk_speckle_scratch_bytes.restype = C.c_size_t
    L.gipk_speckle_scratch_bytes.argtypes = [C.c_int] * 3
    L.gipk_filter_speckles.argtypes = capi.PROTOTYPES["gipk_filter_speckles"][1]
dev = torch.device("cuda:0")
pairs = [synth.rectified_pair(W, H, 128, seed=300 + i)[:2] for i in range(4)]
l = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(B)])).to(dev)
r = torch.from_numpy(np.stack([pairs[i % 4][1] for i in range(B)])).to(dev)
d0 = K.stereobm(l, r, K.bm_params(block_size=9, num_disparities=128, disp12_max_diff=1))
scr = [torch.empty(L.gipk_speckle_scratch_bytes(W, H, B), dtype=torch.uint8, device=dev) for L in libs]
outs = []
def run(i, d):
    rc = libs[i].gipk_filter_speckles(d.data_ptr(), W * 2, W, H, -16, 800, 80, scr[i].data_ptr(), scr[i].numel(), B, H * W * 2, None)
    assert rc == 0, rc
for i in range(len(libs)):
    d = d0.clone(); run(i, d); torch.cuda.synchronize(); outs.append(d)
print("outputs identical:", [bool(torch.equal(outs[0], o)) for o in outs[1:]], " changed px:", int((outs[0] != d0).sum()))
times = [[] for _ in libs]
for rnd in range(10):
    for i in range(len(libs)):
        ds = [d0.clone() for _ in range(4)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for d in ds: run(i, d)
        e1.record(); torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) / 4)
for i in range(len(libs)):
    print("%s: median %.3f ms  min %.3f ms" % (os.path.basename(paths[i]), statistics.median(times[i]), min(times[i])))
