"""Device-resident multi-frame pipeline (gip_batch_* of the C-ABI) for throughput runs.

One step = the per-frame schedule of the reference's StereoProcessor::imageCb (src/StereoProcessor.cpp:157-298) applied to a
batch of stereo pairs that already sit in HBM: [debayer] -> rectify L/R (+ colour L) -> StereoBM -> speckle -> float
disparity + PointCloud2.  torch only provides the HBM tensors and the stream.
"""
import ctypes as C

import numpy as np
import torch

from . import capi
from .processor import GpuStereoProcessor, camera_info

ENCODINGS = {"mono8": 0, "bgr8": 1, "rgb8": 2, "bayer_rggb8": 3, "bayer_bggr8": 4, "bayer_gbrg8": 5, "bayer_grbg8": 6}


class BatchPipeline:
    def __init__(self, width, height, calib, params, batch, encoding="mono8", outputs=capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD,
                 device=0, max_speckle_size=0, max_speckle_diff=0.0, want_f32=True, fused_front_end=False, batches_in_flight=1):
        self.w, self.h, self.batch, self.encoding = width, height, batch, encoding
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.proc = GpuStereoProcessor(device)
        self.proc.initStereoModel(camera_info(width, height, **calib["left"]), camera_info(width, height, **calib["right"]))
        self.proc.setParams(params)
        self.proc.setMaxSpeckleSize(max_speckle_size)
        self.proc.setMaxSpeckleDiff(max_speckle_diff)
        self.outputs = outputs
        self.fused_front_end = fused_front_end
        self.batches_in_flight = batches_in_flight     # how many pipelines like this one the caller keeps busy on this device
        self.disparity = torch.empty((batch, height, width), dtype=torch.int16, device=self.dev)
        self.disparity_f32 = torch.empty((batch, height, width), dtype=torch.float32, device=self.dev) if want_f32 else None
        self.pointcloud = torch.empty((batch, height, width, 32), dtype=torch.uint8, device=self.dev) if outputs & capi.TOPIC_POINTCLOUD else None
        self._maps = None
        self._lib = capi.lib()

    # ---- rectification tables: built on the host by rank 0's model, broadcast, installed from device memory
    def host_maps_to_device(self):
        out = []
        for side in (capi.SIDE_L, capi.SIDE_R):
            m1, m2 = self.proc.rectifyMaps(side)
            out += [torch.from_numpy(m1).to(self.dev), torch.from_numpy(m2.view(np.int16)).to(self.dev)]
        return out

    def empty_maps(self):
        return [torch.empty((self.h, self.w, 2), dtype=torch.int16, device=self.dev), torch.empty((self.h, self.w), dtype=torch.int16, device=self.dev),
                torch.empty((self.h, self.w, 2), dtype=torch.int16, device=self.dev), torch.empty((self.h, self.w), dtype=torch.int16, device=self.dev)]

    def install_maps(self, maps):
        self._maps = maps  # keep alive
        capi.check(self._lib.gip_batch_set_maps_device(self.proc.handle, self.w, self.h, *[m.data_ptr() for m in maps]), self.proc.handle)

    def run(self, left_raw, right_raw, matcher_stream=None, phase=capi.BATCH_WHOLE):
        """left_raw/right_raw: (B,H,W) or (B,H,W,3) uint8 CUDA tensors.  Asynchronous on torch's current stream.
        matcher_stream: optional torch.cuda.Stream shared by all pipelines of the process for the SAD matcher kernel.
        phase: capi.BATCH_FRONT / BATCH_BACK split the chain at the matcher (gip_batch_desc::phase) for two-stream software pipelining."""
        assert left_raw.is_cuda and left_raw.is_contiguous() and right_raw.is_contiguous() and left_raw.shape[0] == self.batch
        d = capi.BatchDesc()
        d.width, d.height, d.batch = self.w, self.h, self.batch
        d.encoding = ENCODINGS[self.encoding]
        d.outputs = self.outputs
        d.left_raw, d.right_raw = left_raw.data_ptr(), right_raw.data_ptr()
        d.disparity = self.disparity.data_ptr()
        d.disparity_f32 = self.disparity_f32.data_ptr() if self.disparity_f32 is not None else None
        d.pointcloud = self.pointcloud.data_ptr() if self.pointcloud is not None else None
        d.matcher_stream = matcher_stream.cuda_stream if matcher_stream is not None else None
        d.fused_front_end = 1 if self.fused_front_end else 0
        d.batches_in_flight = int(self.batches_in_flight)
        d.phase = int(phase)
        capi.check(self._lib.gip_batch_run(self.proc.handle, C.byref(d), C.c_void_p(torch.cuda.current_stream().cuda_stream)), self.proc.handle)

    def timing_reset(self):
        capi.check(self._lib.gip_batch_timing_reset(self.proc.handle), self.proc.handle)

    def timing_read(self):
        a, b, n = C.c_double(), C.c_double(), C.c_int()
        capi.check(self._lib.gip_batch_timing_read(self.proc.handle, C.byref(a), C.byref(b), C.byref(n)), self.proc.handle)
        return a.value, b.value, n.value

    def timing_read_spans(self):
        """(matcher ms, whole-chain ms, remap+SAD span ms, runs) summed over the runs since the last reset"""
        a, b, c, n = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        capi.check(self._lib.gip_batch_timing_read_spans(self.proc.handle, C.byref(a), C.byref(b), C.byref(c), C.byref(n)), self.proc.handle)
        return a.value, b.value, c.value, n.value

    def close(self):
        self.proc.close()
