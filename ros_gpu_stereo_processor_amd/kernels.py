"""Device-pointer front-end of the gipk_* kernel launchers for torch CUDA tensors.

torch is plumbing only (HBM allocations, streams): every function launches hand-written HIP kernels through the C-ABI
on the tensors' storage.  Inputs are (B, H, W[, C]) or (H, W[, C]) contiguous CUDA tensors.
"""
import ctypes as C

import numpy as np
import torch

from . import capi


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ck(rc):
    if rc != capi.OK:
        raise capi.GipError(rc, "kernel launcher")


def _b(t, nd):
    """-> (tensor with batch dim, batch)"""
    if t.dim() == nd:
        t = t.unsqueeze(0)
    assert t.is_cuda and t.is_contiguous(), "contiguous CUDA tensor required"
    return t, t.shape[0]


def bm_params(**kw):
    d = dict(prefilter_type=capi.PREFILTER_XSOBEL, prefilter_size=9, prefilter_cap=31, block_size=21, min_disparity=0, num_disparities=64,
             texture_threshold=10, uniqueness_ratio=15, disp12_max_diff=-1, speckle_window_size=0, speckle_range=0)
    d.update(kw)
    return capi.BMParams(**d)


def _cvt(name, src, out_c, out_dtype, *extra, src_nd):
    L = capi.lib()
    s, B = _b(src, src_nd)
    H, W = s.shape[1], s.shape[2]
    shape = (B, H, W) if out_c == 1 else (B, H, W, out_c)
    dst = torch.empty(shape, dtype=out_dtype, device=s.device)
    sp = s.stride(1) * s.element_size()
    dp = dst.stride(1) * dst.element_size()
    _ck(getattr(L, name)(s.data_ptr(), sp, dst.data_ptr(), dp, W, H, *extra, B, s.stride(0) * s.element_size(), dst.stride(0) * dst.element_size(),
                         _stream()))
    return dst if src.dim() == src_nd + 1 else dst[0]


def bgr_to_gray(src, src_is_rgb=False):
    return _cvt("gipk_cvt_bgr_to_gray", src, 1, torch.uint8, int(src_is_rgb), src_nd=3)


def gray_to_bgr(src):
    return _cvt("gipk_cvt_gray_to_bgr", src, 3, torch.uint8, src_nd=2)


def swap_rb(src):
    return _cvt("gipk_cvt_swap_rb", src, 3, torch.uint8, src_nd=3)


def bayer_to_bgr(src, encoding):
    return _cvt("gipk_cvt_bayer_to_bgr", src, 3, torch.uint8, capi.BAYER[encoding], src_nd=2)


def bayer_to_gray(src, encoding):
    return _cvt("gipk_cvt_bayer_to_gray", src, 1, torch.uint8, capi.BAYER[encoding], src_nd=2)


def mono8_to_mono16(src):
    return _cvt("gipk_cvt_mono8_to_mono16", src, 1, torch.int16, src_nd=2)  # bit pattern of uint16


def mono16_to_mono8(src):
    return _cvt("gipk_cvt_mono16_to_mono8", src, 1, torch.uint8, src_nd=2)


def remap(src, map1, map2, interpolation=capi.INTER_LINEAR):
    """src (B,H,W) or (B,H,W,3) uint8; map1 (h,w,2) int16; map2 (h,w) int16-bit-pattern of uint16."""
    L = capi.lib()
    cn = 3 if (src.dim() == 4 or (src.dim() == 3 and src.shape[-1] == 3 and map1.shape[1] != src.shape[-1])) else 1
    s, B = _b(src, 2 if cn == 1 else 3)
    sh, sw = s.shape[1], s.shape[2]
    h, w = map2.shape
    dst = torch.empty((B, h, w) if cn == 1 else (B, h, w, 3), dtype=torch.uint8, device=s.device)
    _ck(L.gipk_remap_u8_interp(s.data_ptr(), sw * cn, sw, sh, dst.data_ptr(), w * cn, w, h, cn, map1.data_ptr(), map2.data_ptr(), interpolation, B,
                               sh * sw * cn, h * w * cn, _stream()))
    return dst if src.dim() == (3 if cn == 1 else 4) else dst[0]


def prefilter(src, prefilter_type, prefilter_size, prefilter_cap):
    L = capi.lib()
    s, B = _b(src, 2)
    H, W = s.shape[1], s.shape[2]
    dst = torch.empty_like(s)
    _ck(L.gipk_prefilter(s.data_ptr(), W, dst.data_ptr(), W, W, H, prefilter_type, prefilter_size, prefilter_cap, B, H * W, H * W, _stream()))
    return dst if src.dim() == 3 else dst[0]


def rectify_xsobel(src, map1, map2, prefilter_cap, want_rect=True):
    """Fused remap + XSOBEL: returns (prefiltered, rectified or None).  src: (H,W) or (B,H,W) uint8; maps as for remap()."""
    L = capi.lib()
    s, B = _b(src, 2)
    H, W = map1.shape[0], map1.shape[1]
    SH, SW = s.shape[1], s.shape[2]
    pf = torch.empty((B, H, W), dtype=torch.uint8, device=s.device)
    rect = torch.empty((B, H, W), dtype=torch.uint8, device=s.device) if want_rect else None
    _ck(L.gipk_rectify_xsobel(s.data_ptr(), SW, SW, SH, pf.data_ptr(), W, rect.data_ptr() if want_rect else None, W, W, H, map1.data_ptr(), map2.data_ptr(),
                              prefilter_cap, B, SH * SW, H * W, H * W, _stream()))
    if src.dim() == 2:
        return pf[0], (rect[0] if want_rect else None)
    return pf, rect


def bm_match(left_f, right_f, params, want_cost=True):
    L = capi.lib()
    l, B = _b(left_f, 2)
    r, _ = _b(right_f, 2)
    H, W = l.shape[1], l.shape[2]
    disp = torch.empty((B, H, W), dtype=torch.int16, device=l.device)
    cost = torch.zeros((B, H, W), dtype=torch.int32, device=l.device) if want_cost else None
    _ck(L.gipk_bm_match(l.data_ptr(), r.data_ptr(), W, W, H, C.byref(params), disp.data_ptr(), W * 2, cost.data_ptr() if want_cost else None, W * 4, B,
                        H * W, H * W * 2, H * W * 4, _stream()))
    if left_f.dim() == 2:
        return disp[0], (cost[0] if want_cost else None)
    return disp, cost


def bm_finalize(disp, cost, params):
    L = capi.lib()
    d, B = _b(disp, 2)
    H, W = d.shape[1], d.shape[2]
    _ck(L.gipk_bm_finalize(d.data_ptr(), W * 2, cost.data_ptr() if cost is not None else None, W * 4, W, H, C.byref(params), B, H * W * 2, H * W * 4,
                           _stream()))
    return disp


def stereobm(left, right, params):
    """cv::StereoBM::compute on (B,H,W) uint8 rectified pairs -> int16 x16."""
    L = capi.lib()
    l, B = _b(left, 2)
    r, _ = _b(right, 2)
    H, W = l.shape[1], l.shape[2]
    disp = torch.empty((B, H, W), dtype=torch.int16, device=l.device)
    nbytes = L.gipk_stereobm_scratch_bytes(W, H, B, C.byref(params))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=l.device)
    _ck(L.gipk_stereobm(l.data_ptr(), r.data_ptr(), W, W, H, C.byref(params), disp.data_ptr(), W * 2, scratch.data_ptr(), nbytes, B, H * W, H * W * 2,
                        _stream()))
    return disp if left.dim() == 3 else disp[0]


def disparity_to_f32(disp, cx_l, cx_r):
    L = capi.lib()
    d, B = _b(disp, 2)
    H, W = d.shape[1], d.shape[2]
    out = torch.empty((B, H, W), dtype=torch.float32, device=d.device)
    _ck(L.gipk_disparity_to_f32(d.data_ptr(), W * 2, out.data_ptr(), W * 4, W, H, float(cx_l), float(cx_r), B, H * W * 2, H * W * 4, _stream()))
    return out if disp.dim() == 3 else out[0]


def filter_speckles(disp, new_val, max_speckle_size, max_diff):
    """in place on an int16 (B,H,W) tensor"""
    L = capi.lib()
    d, B = _b(disp, 2)
    H, W = d.shape[1], d.shape[2]
    nbytes = L.gipk_speckle_scratch_bytes(W, H, B)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=d.device)
    _ck(L.gipk_filter_speckles(d.data_ptr(), W * 2, W, H, int(new_val), int(max_speckle_size), int(max_diff), scratch.data_ptr(), nbytes, B, H * W * 2,
                               _stream()))
    return disp


class SpeckleLabels:
    """Phase 1 of the speckle filter on ONE (H,W) int16 disparity tile: component labels and sizes in device scratch."""

    def __init__(self, disp, new_val, max_diff):
        assert disp.dim() == 2 and disp.dtype == torch.int16 and disp.is_contiguous()
        self.L = capi.lib()
        self.disp, self.H, self.W, self.new_val = disp, disp.shape[0], disp.shape[1], int(new_val)
        self.nbytes = self.L.gipk_speckle_scratch_bytes(self.W, self.H, 1)
        self.scratch = torch.empty(self.nbytes, dtype=torch.uint8, device=disp.device)
        _ck(self.L.gipk_speckle_label(disp.data_ptr(), self.W * 2, self.W, self.H, self.new_val, int(max_diff), self.scratch.data_ptr(), self.nbytes, 1, self.H * self.W * 2,
                                      _stream()))

    def row_labels(self, row):
        """(roots int32[W] (-1 = invalid pixel), sizes int32[W]) of one row"""
        roots = torch.empty(self.W, dtype=torch.int32, device=self.disp.device)
        sizes = torch.empty(self.W, dtype=torch.int32, device=self.disp.device)
        _ck(self.L.gipk_speckle_row_labels(self.scratch.data_ptr(), self.nbytes, self.W, self.H, 1, int(row), roots.data_ptr(), sizes.data_ptr(), _stream()))
        return roots, sizes

    def set_sizes(self, roots, sizes):
        if len(roots) == 0:
            return
        r = torch.as_tensor(roots, dtype=torch.int32, device=self.disp.device).contiguous()
        z = torch.as_tensor(sizes, dtype=torch.int32, device=self.disp.device).contiguous()
        _ck(self.L.gipk_speckle_set_sizes(self.scratch.data_ptr(), self.nbytes, self.W, self.H, 1, r.data_ptr(), z.data_ptr(), r.numel(), _stream()))

    def border_record(self, tile):
        """Tile `tile`'s border record (uint8 tensor, gipk_speckle_border_record_bytes(W) bytes): first and last row as value / union-find
        parent (canonical border pixel of the component) / tile-local component size -- what the ranks all-gather (rowsplit.distributed_speckle)."""
        n = self.L.gipk_speckle_border_record_bytes(self.W)
        rec = torch.zeros(n, dtype=torch.uint8, device=self.disp.device)
        self._roots = torch.empty(2 * self.W, dtype=torch.int32, device=self.disp.device)
        _ck(self.L.gipk_speckle_border_record(self.disp.data_ptr(), self.W * 2, self.W, self.H, int(tile), self.scratch.data_ptr(), self.nbytes, rec.data_ptr(),
                                              self._roots.data_ptr(), _stream()))
        return rec

    def border_merge(self, records, n_tiles, tile, max_diff):
        """records: the border records of ALL tiles, concatenated in tile order, on this device (a private copy: modified in place).  Unites
        the components that touch across the tile borders (on the device) and overwrites the sizes of this tile's border components with the
        size of the whole component."""
        assert records.is_cuda and records.is_contiguous() and records.numel() == n_tiles * self.L.gipk_speckle_border_record_bytes(self.W)
        _ck(self.L.gipk_speckle_border_merge(records.data_ptr(), int(n_tiles), int(tile), self.W, self.H, self.new_val, int(max_diff), self.scratch.data_ptr(), self.nbytes,
                                             self._roots.data_ptr(), _stream()))

    def apply(self, max_speckle_size):
        _ck(self.L.gipk_speckle_apply(self.disp.data_ptr(), self.W * 2, self.W, self.H, self.new_val, int(max_speckle_size), self.scratch.data_ptr(), self.nbytes, 1,
                                      self.H * self.W * 2, _stream()))
        return self.disp


def _q(Q):
    return (C.c_double * 16)(*[float(v) for v in Q.reshape(-1)])


def reproject_to_3d(disp_f32, Q, missing_value, handle_missing=True):
    L = capi.lib()
    d, B = _b(disp_f32, 2)
    H, W = d.shape[1], d.shape[2]
    out = torch.empty((B, H, W, 3), dtype=torch.float32, device=d.device)
    _ck(L.gipk_reproject_to_3d(d.data_ptr(), W * 4, out.data_ptr(), W * 12, W, H, _q(Q), C.c_float(missing_value), int(handle_missing), B, H * W * 4,
                               H * W * 12, _stream()))
    return out if disp_f32.dim() == 3 else out[0]


def pack_pointcloud2(xyz, bgr):
    L = capi.lib()
    p, B = _b(xyz, 3)
    c, _ = _b(bgr, 3)
    H, W = p.shape[1], p.shape[2]
    out = torch.empty((B, H, W, 32), dtype=torch.uint8, device=p.device)
    _ck(L.gipk_pack_pointcloud2(p.data_ptr(), W * 12, c.data_ptr(), W * 3, out.data_ptr(), W, H, B, H * W * 12, H * W * 3, H * W * 32, _stream()))
    return out if xyz.dim() == 4 else out[0]


def disparity_to_pointcloud2(disp, bgr, Q, cx_l, cx_r, min_disparity):
    L = capi.lib()
    d, B = _b(disp, 2)
    H, W = d.shape[1], d.shape[2]
    out = torch.empty((B, H, W, 32), dtype=torch.uint8, device=d.device)
    cptr = None
    if bgr is not None:
        c, _ = _b(bgr, 3)
        cptr = c.data_ptr()
    _ck(L.gipk_disparity_to_pointcloud2(d.data_ptr(), W * 2, cptr, W * 3, out.data_ptr(), W, H, _q(Q), float(cx_l), float(cx_r), int(min_disparity), B,
                                        H * W * 2, H * W * 3, H * W * 32, _stream()))
    return out if disp.dim() == 3 else out[0]


def disparity_outputs(disp, colour, Q, cx_l, cx_r, min_disparity, want_f32=True, want_cloud=True, y_origin=0):
    """Fused tail of the chain on ONE (H,W) int16 disparity: float disparity d/16 - (cx_l - cx_r) and the 32-byte PointCloud2 records.
    colour: (H,W) mono (replicated, = GRAY2BGR) or (H,W,3) BGR rectified image, or None.  y_origin: the tile holds rows
    [y_origin, y_origin + H) of a larger frame (row-tile split)."""
    L = capi.lib()
    assert disp.dim() == 2 and disp.is_contiguous()
    H, W = disp.shape
    f32 = torch.empty((H, W), dtype=torch.float32, device=disp.device) if want_f32 else None
    pc = torch.empty((H, W, 32), dtype=torch.uint8, device=disp.device) if want_cloud else None
    cn = 0 if colour is None else (3 if colour.dim() == 3 else 1)
    if colour is not None:
        assert colour.is_contiguous() and colour.shape[0] == H and colour.shape[1] == W
    _ck(L.gipk_disparity_outputs_rows(disp.data_ptr(), W * 2, colour.data_ptr() if colour is not None else None, W * cn, cn if cn else 3, f32.data_ptr() if want_f32 else None,
                                      pc.data_ptr() if want_cloud else None, W, H, int(y_origin), _q(np.asarray(Q, np.float64)), float(cx_l), float(cx_r), int(min_disparity), 1,
                                      H * W * 2, H * W * cn, _stream()))
    return f32, pc


def draw_color_disp(disp, ndisp):
    L = capi.lib()
    d, B = _b(disp, 2)
    H, W = d.shape[1], d.shape[2]
    out = torch.empty((B, H, W, 4), dtype=torch.uint8, device=d.device)
    _ck(L.gipk_draw_color_disp(d.data_ptr(), W * 2, out.data_ptr(), W * 4, W, H, int(ndisp), B, H * W * 2, H * W * 4, _stream()))
    return out if disp.dim() == 3 else out[0]
