import sys, os, ctypes as C, torch, numpy as np
sys.path.insert(0, "/root/repo")
from ros_gpu_stereo_processor_amd import capi
L = capi.lib()
W, H, B = 1242, 375, 64
dev = torch.device("cuda:0")
src = torch.randint(0, 256, (B, H, 1280), dtype=torch.uint8, device=dev)
dst = torch.empty_like(src)
for ws in (5, 9, 21):
    for _ in range(2):
        L.gipk_prefilter(src.data_ptr(), 1280, dst.data_ptr(), 1280, W, H, 0, ws, 31, B, 1280 * H, 1280 * H, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        L.gipk_prefilter(src.data_ptr(), 1280, dst.data_ptr(), 1280, W, H, 0, ws, 31, B, 1280 * H, 1280 * H, None)
    e1.record(); torch.cuda.synchronize()
    print("norm prefilter winsize %d: %.3f ms per 64 images" % (ws, e0.elapsed_time(e1) / 5))
