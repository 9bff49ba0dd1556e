#!/usr/bin/env python3
"""A/B timing of gipk_remap_u8 (mono) and gipk_prefilter (XSOBEL) from several library builds in ONE process, same inputs."""
import ctypes as C, sys, os, statistics
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ros_gpu_stereo_processor_amd import capi, synth

W, H, B = 1242, 375, 64
paths = [a for a in sys.argv[1:] if a.endswith(".so")]
libs = [C.CDLL(os.path.abspath(p)) for p in paths]
for L in libs:
    L.gipk_remap_u8.argtypes = capi.PROTOTYPES["gipk_remap_u8"][1]
    L.gipk_prefilter.argtypes = capi.PROTOTYPES["gipk_prefilter"][1]
    L.gip_build_rectify_maps.argtypes = capi.PROTOTYPES["gip_build_rectify_maps"][1]
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
src = torch.from_numpy(rng.integers(0, 256, (B, H, W), dtype=np.uint8)).to(dev)
info = synth.camera_info_pair(W, H)[0] if hasattr(synth, "camera_info_pair") else None
# mild radial distortion map built on the host the same way the tests do
yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
cx, cy, f = W / 2.0, H / 2.0, 700.0
xn, yn = (xx - cx) / f, (yy - cy) / f
r2 = xn * xn + yn * yn
k = 1 + (-0.25) * r2 + 0.08 * r2 * r2
u, v = xn * k * f + cx + 3.3, yn * k * f + cy - 1.7
iu, iv = np.floor(u * 32 + 0.5).astype(np.int64), np.floor(v * 32 + 0.5).astype(np.int64)
m1 = np.stack([np.clip(iu >> 5, -32768, 32767), np.clip(iv >> 5, -32768, 32767)], -1).astype(np.int16)
m2 = (((iv & 31) << 5) | (iu & 31)).astype(np.uint16)
map1, map2 = torch.from_numpy(m1).to(dev), torch.from_numpy(m2).to(dev)
RP = (W + 63) & ~63
outs_r = [torch.zeros((B, H, RP), dtype=torch.uint8, device=dev) for _ in libs]
outs_p = [torch.zeros((B, H, RP), dtype=torch.uint8, device=dev) for _ in libs]

def remap(i):
    rc = libs[i].gipk_remap_u8(src.data_ptr(), W, W, H, outs_r[i].data_ptr(), RP, W, H, 1, map1.data_ptr(), map2.data_ptr(), B, W * H, RP * H, None)
    assert rc == 0, rc
def pref(i):
    rc = libs[i].gipk_prefilter(outs_r[0].data_ptr(), RP, outs_p[i].data_ptr(), RP, W, H, 1, 9, 31, B, RP * H, RP * H, None)
    assert rc == 0, rc

for name, fn, outs in (("remap", remap, outs_r), ("prefilter", pref, outs_p)):
    for i in range(len(libs)):
        fn(i)
    torch.cuda.synchronize()
    print(name, "outputs identical:", [bool(torch.equal(outs[0][:, :, :W], o[:, :, :W])) for o in outs[1:]])
    times = [[] for _ in libs]
    for rnd in range(10):
        for i in range(len(libs)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(5): fn(i)
            e1.record(); torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / 5)
    for i in range(len(libs)):
        print("  %s: median %.4f ms  min %.4f ms" % (os.path.basename(paths[i]), statistics.median(times[i]), min(times[i])))
