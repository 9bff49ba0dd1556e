#!/usr/bin/env python3
"""A/B timing of gipk_stereobm (prefilter + matcher + finalize) from two library builds in ONE process, interleaved rounds,
same inputs (rule: never compare timings taken on different devices).  usage: bm_ab.py libA.so libB.so [batch]"""
import ctypes as C, sys, os, statistics
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ros_gpu_stereo_processor_amd import capi, synth

W, H, B = int(os.environ.get('AB_W', 1242)), int(os.environ.get('AB_H', 375)), int(os.environ.get('AB_B', 64))
ND, BLK = int(os.environ.get('AB_ND', 128)), int(os.environ.get('AB_BLOCK', 9))
paths = [a for a in sys.argv[1:] if a.endswith(".so")]
libs = [C.CDLL(os.path.abspath(p)) for p in paths]
for L in libs:
    L.gipk_stereobm_scratch_bytes.restype = C.c_size_t
    L.gipk_stereobm_scratch_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(capi.BMParams)]
    L.gipk_stereobm.argtypes = capi.PROTOTYPES["gipk_stereobm"][1]
dev = torch.device("cuda:0")
pairs = [synth.rectified_pair(W, H, ND, seed=300 + i)[:2] for i in range(4)]
l = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(B)])).to(dev)
r = torch.from_numpy(np.stack([pairs[i % 4][1] for i in range(B)])).to(dev)
p = capi.BMParams(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=BLK, min_disparity=0, num_disparities=ND, texture_threshold=10,
                  uniqueness_ratio=int(os.environ.get("AB_UNIQ", 15)), disp12_max_diff=1, speckle_window_size=0, speckle_range=0)
outs, scr = [], []
for L in libs:
    n = L.gipk_stereobm_scratch_bytes(W, H, B, C.byref(p))
    scr.append(torch.empty(n, dtype=torch.uint8, device=dev))
    outs.append(torch.empty((B, H, W), dtype=torch.int16, device=dev))

def run(i):
    L = libs[i]
    rc = L.gipk_stereobm(l.data_ptr(), r.data_ptr(), W, W, H, C.byref(p), outs[i].data_ptr(), W * 2, scr[i].data_ptr(), scr[i].numel(), B, H * W, H * W * 2, None)
    assert rc == 0, rc

N = len(libs)
for i in range(N):
    run(i)
torch.cuda.synchronize()
print("outputs identical:", all(bool(torch.equal(outs[0], outs[i])) for i in range(1, N)))
for i in range(1, N):
    if not torch.equal(outs[0], outs[i]):
        a, b = outs[0][0].cpu().numpy().astype(int), outs[i][0].cpu().numpy().astype(int)
        bad = a != b
        ys, xs = np.nonzero(bad)
        print("  lib %d vs 0, image 0: %d of %d pixels differ; rows %d..%d cols %d..%d" % (i, bad.sum(), bad.size, ys.min(), ys.max(), xs.min(), xs.max()))
        print("  columns with most mismatches:", sorted(((int(c), int(n)) for c, n in zip(*np.unique(xs, return_counts=True))), key=lambda t: -t[1])[:12])
        print("  (x - lofs) mod 4 histogram:", np.bincount((xs - (ND - 1)) % 4, minlength=4).tolist())
        print("  rows with most mismatches:", sorted(((int(c), int(n)) for c, n in zip(*np.unique(ys, return_counts=True))), key=lambda t: -t[1])[:8])
        for k in range(min(6, len(ys))):
            print("   (y=%d,x=%d): %d vs %d" % (ys[k], xs[k], a[ys[k], xs[k]], b[ys[k], xs[k]]))
times = [[] for _ in range(N)]
for rnd in range(12):
    for i in range(N):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) / 5)
for i in range(N):
    print("%s: median %.3f ms  min %.3f ms  (per %d pairs)" % (os.path.basename(paths[i]), statistics.median(times[i]), min(times[i]), B))
