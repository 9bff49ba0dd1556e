import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from ros_gpu_stereo_processor_amd import capi, synth
from ros_gpu_stereo_processor_amd import processor as P
w, h, nd = 640, 240, 64
cal = synth.scaled_calibration(w, h)
l, r, _, _ = synth.raw_pair(w, h, nd, seed=3, calib=cal)
def cycle(n):
    for it in range(n):
        p = P.GpuStereoProcessor(0)
        p.initStereoModel(P.camera_info(w, h, **cal["left"]), P.camera_info(w, h, **cal["right"]))
        p.setNumDisparities(nd); p.setBlockSize(9); p.setDisp12MaxDiff(1); p.setMaxSpeckleSize(100); p.setMaxSpeckleDiff(2)
        p.processPair(l, r, "mono8", capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD)
        p.close()
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]
f0 = cycle(1)
f1 = cycle(20)
f2 = cycle(20)
f3 = cycle(40)
print("free MB after 1 / +20 / +20 / +40 cycles: %.1f %.1f %.1f %.1f" % (f0 / 1e6, f1 / 1e6, f2 / 1e6, f3 / 1e6))
