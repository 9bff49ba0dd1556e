import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
from ros_gpu_stereo_processor_amd import capi, synth
from ros_gpu_stereo_processor_amd import processor as P
from ros_gpu_stereo_processor_amd.pipeline import BatchPipeline
dev = torch.device("cuda:0")
w, h, nd = 640, 240, 64
cal = synth.scaled_calibration(w, h)
l, r, _, _ = synth.raw_pair(w, h, nd, seed=3, calib=cal)
free0 = torch.cuda.mem_get_info()[0]
ref = None
for it in range(40):
    p = P.GpuStereoProcessor(0)
    p.initStereoModel(P.camera_info(w, h, **cal["left"]), P.camera_info(w, h, **cal["right"]))
    p.setNumDisparities(nd); p.setBlockSize(9); p.setUniquenessRatio(15); p.setDisp12MaxDiff(1); p.setTextureThreshold(10)
    p.setMaxSpeckleSize(100); p.setMaxSpeckleDiff(2)
    for k in range(5):
        p.processPair(l, r, "mono8", capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD)
    d = p.downloadMat(P.L_DISPARITY)
    if ref is None: ref = d.copy()
    assert np.array_equal(d, ref), it
    p.close()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("create/destroy x40 ok; device memory delta %.1f MB" % ((free0 - free1) / 1e6))
kw = dict(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=9, min_disparity=0, num_disparities=nd, texture_threshold=10, uniqueness_ratio=15, disp12_max_diff=1, speckle_window_size=0, speckle_range=0)
B = 16
pipes = [BatchPipeline(w, h, cal, capi.BMParams(**kw), B, "mono8", capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD, 0, 100, 2.0) for _ in range(3)]
for p in pipes: p.install_maps(p.host_maps_to_device())
L = torch.from_numpy(np.stack([l] * B)).to(dev); R = torch.from_numpy(np.stack([r] * B)).to(dev)
streams = [torch.cuda.Stream(device=dev) for _ in pipes]; shared = torch.cuda.Stream(device=dev)
t0 = time.time()
for i in range(3000):
    k = i % 3
    with torch.cuda.stream(streams[k]):
        pipes[k].run(L, R, shared)
torch.cuda.synchronize()
print("3000 pipelined steps ok in %.1f s" % (time.time() - t0))
for p in pipes:
    assert np.array_equal(p.disparity[0].cpu().numpy(), p.disparity[B - 1].cpu().numpy())
    p.close()
print("soak ok")
