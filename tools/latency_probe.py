import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ros_gpu_stereo_processor_amd import capi, synth
from ros_gpu_stereo_processor_amd import processor as P
W, H = 1242, 375
calib = synth.scaled_calibration(W, H)
l, r, _, _ = synth.raw_pair(W, H, 128, seed=1, calib=calib)
p = P.GpuStereoProcessor(0)
p.initStereoModel(P.camera_info(W, H, **calib["left"]), P.camera_info(W, H, **calib["right"]))
p.setNumDisparities(128); p.setBlockSize(9); p.setUniquenessRatio(15); p.setDisp12MaxDiff(1); p.setTextureThreshold(10)
p.setMaxSpeckleSize(800); p.setMaxSpeckleDiff(5)
def T(name, f, n=1):
    t0 = time.perf_counter(); f(); p.waitForAllStreams(); dt = (time.perf_counter() - t0) * 1e3
    print("%-28s %8.3f ms" % (name, dt))
for it in range(3):
    print("--- iteration", it)
    T("cleanSenders", lambda: p.cleanSenders())
    T("uploadMat L", lambda: p.uploadMat(P.L_RAW, l, "mono8"))
    T("uploadMat R", lambda: p.uploadMat(P.R_RAW, r, "mono8"))
    T("convertRawToMono L", lambda: p.convertRawToMono(P.SIDE_L))
    T("convertRawToMono R", lambda: p.convertRawToMono(P.SIDE_R))
    T("convertRawToColor L", lambda: p.convertRawToColor(P.SIDE_L))
    T("rectify L", lambda: p.rectifyImage(P.L_MONO, P.L_RECT_MONO))
    T("rectify R", lambda: p.rectifyImage(P.R_MONO, P.R_RECT_MONO))
    T("rectify colour", lambda: p.rectifyImage(P.L_COLOR, P.L_RECT_COLOR))
    T("computeDisparity", lambda: p.computeDisparity(P.L_RECT_MONO, P.R_RECT_MONO, P.L_DISPARITY))
    T("filterSpeckles", lambda: p.filterSpeckles(P.L_DISPARITY))
    T("project3D", lambda: p.projectDisparityTo3DPoints(P.L_DISPARITY, P.L_POINTS2))
    t0 = time.perf_counter(); p.processPair(l, r, "mono8", capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD); print("processPair total %.3f ms" % ((time.perf_counter() - t0) * 1e3))
