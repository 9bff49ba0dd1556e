"""ctypes mirror of the streaming C-ABI (gip_streamer_*): the pipelined host link of the stereo hot path.

Frames are numpy arrays in host memory; results arrive through a callback on a HIP runtime thread (or are only counted when no
callback is given).  See include/gpuimageproc_hip.h for the contract."""
import ctypes as C

import numpy as np

from . import capi
from .pipeline import ENCODINGS


class Streamer:
    def __init__(self, proc, width, height, encoding="mono8", outputs=capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD, want_disparity16=False, depth=0,
                 frames_per_batch=0, on_result=None, inputs_pinned=False, adaptive=True, compute_lanes=0):
        """proc: a GpuStereoProcessor with model and parameters set.  on_result(seq, header, d16, f32, cloud): called per pair with
        numpy COPIES (None for outputs not requested).  inputs_pinned: the arrays given to submit() are views of pinned memory (e.g.
        torch.from_numpy(x).pin_memory().numpy()) that stay alive until delivered; they are then read in place by the H2D copy.
        adaptive (default): pairs are submitted one by one and the streamer forms the batches (frames_per_batch = the most pairs per launch,
        0 = automatic); adaptive=False: fixed batches of frames_per_batch pairs."""
        self._lib = capi.lib()
        self.proc = proc
        self.w, self.h = width, height
        self.channels = 3 if encoding in ("bgr8", "rgb8") else 1
        cfg = capi.StreamerConfig(width, height, ENCODINGS[encoding], outputs, 1 if want_disparity16 else 0, depth, frames_per_batch, 1 if inputs_pinned else 0,
                                  0 if adaptive else 1, compute_lanes)
        self._on_result = on_result
        self._errors = []

        def _cb(res, _user):
            try:
                r = res.contents

                def arr(ptr, ctype, shape):
                    if not ptr:
                        return None
                    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=shape).copy()
                self._on_result(int(r.seq), r.header, arr(r.disparity16, C.c_int16, (self.h, self.w)), arr(r.disparity_f32, C.c_float, (self.h, self.w)),
                                arr(r.pointcloud, C.c_uint8, (self.h, self.w, 32)))
            except Exception as e:  # never let an exception cross the C boundary
                self._errors.append(e)
        self._cb = capi.STREAMER_CB(_cb) if on_result is not None else C.cast(None, capi.STREAMER_CB)
        self._s = C.c_void_p()
        rc = self._lib.gip_streamer_open(proc.handle, C.byref(cfg), self._cb, None, C.byref(self._s))
        if rc != 0:
            raise capi.GipError(rc, "gip_streamer_open")

    def _check(self, rc):
        if rc != 0:
            raise capi.GipError(rc, self._lib.gip_streamer_last_error(self._s).decode())

    def submit(self, left, right, header=None):
        shape = (self.h, self.w) if self.channels == 1 else (self.h, self.w, self.channels)
        if left.shape != shape or right.shape != shape or left.dtype != np.uint8 or right.dtype != np.uint8:
            raise ValueError("Streamer.submit: images must be uint8 of shape %r (got %r / %r)" % (shape, left.shape, right.shape))
        assert left.flags["C_CONTIGUOUS"] and right.flags["C_CONTIGUOUS"]
        self._check(self._lib.gip_streamer_submit(self._s, left.ctypes.data, right.ctypes.data, 0, C.byref(header) if header is not None else None))

    def flush(self):
        self._check(self._lib.gip_streamer_flush(self._s))
        if self._errors:
            raise self._errors[0]

    def delivered(self):
        return int(self._lib.gip_streamer_delivered(self._s))

    def launches(self):
        return int(self._lib.gip_streamer_launches(self._s))

    def close(self):
        if self._s:
            self._lib.gip_streamer_close(self._s)
            self._s = C.c_void_p()
