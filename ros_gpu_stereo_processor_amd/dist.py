"""Multi-GPU plumbing for the frame-sharded stereo pipeline (one process per GPU, torch.distributed: "nccl" = RCCL on the
GPUs, "gloo" in the CPU tests).

The path shards by independent units (stereo pairs): frame i goes to rank i mod N and no collective sits on the data path.
The only exchange is one broadcast, at model-init time, of the calibration (CameraInfo) and of the rectification tables
rank 0 built from it (6 bytes per pixel and eye), so that every rank remaps with bit-identical tables.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import capi
from .processor import camera_info

_CI_LEN = 2 + 9 + 8 + 1 + 9 + 12  # width, height, K, D, nD, R, P


def pack_camera_info(calib_side, width, height):
    D = np.zeros(8)
    d = np.asarray(calib_side["D"], np.float64).reshape(-1)
    D[:d.size] = d
    return np.concatenate([[width, height], np.asarray(calib_side["K"], np.float64).reshape(9), D, [d.size],
                           np.asarray(calib_side["R"], np.float64).reshape(9), np.asarray(calib_side["P"], np.float64).reshape(12)])


def unpack_camera_info(v):
    v = np.asarray(v, np.float64)
    w, h = int(v[0]), int(v[1])
    nD = int(v[19])
    return dict(K=v[2:11].copy(), D=v[11:11 + nD].copy(), R=v[20:29].copy(), P=v[29:41].copy()), w, h


def build_maps_host(calib_side, width, height):
    """Rectification tables through the C-ABI's host-only entry point (no GPU needed)."""
    ci = camera_info(width, height, **calib_side)
    m1 = np.empty((height, width, 2), np.int16)
    m2 = np.empty((height, width), np.uint16)
    capi.check(capi.lib().gip_build_rectify_maps(C.byref(ci), m1.ctypes.data, m2.ctypes.data))
    return m1, m2


def frame_shard(rank, world_size, n_frames):
    """Indices of the frames rank `rank` processes: frame i -> rank i mod N."""
    return list(range(rank, n_frames, world_size))


def broadcast_model(calib, width, height, device, src=0):
    """rank `src` passes its calibration; every rank returns (calib, [map1_l, map2_l, map1_r, map2_r] tensors on `device`).
    The tables are the ones rank `src` built; map2 travels as the int16 bit pattern of the uint16 table."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    ci = torch.zeros(2 * _CI_LEN, dtype=torch.float64, device=device)
    if rank == src:
        ci = torch.from_numpy(np.concatenate([pack_camera_info(calib["left"], width, height), pack_camera_info(calib["right"], width, height)])).to(device)
    if world > 1:
        dist.broadcast(ci, src=src)
    v = ci.cpu().numpy()
    left, w, h = unpack_camera_info(v[:_CI_LEN])
    right, _, _ = unpack_camera_info(v[_CI_LEN:])
    out_calib = dict(left=left, right=right, width=w, height=h)
    maps = [torch.empty((h, w, 2), dtype=torch.int16, device=device), torch.empty((h, w), dtype=torch.int16, device=device),
            torch.empty((h, w, 2), dtype=torch.int16, device=device), torch.empty((h, w), dtype=torch.int16, device=device)]
    if rank == src:
        for i, side in enumerate(("left", "right")):
            m1, m2 = build_maps_host(out_calib[side], w, h)
            maps[2 * i].copy_(torch.from_numpy(m1))
            maps[2 * i + 1].copy_(torch.from_numpy(m2.view(np.int16)))
    if world > 1:
        for m in maps:
            dist.broadcast(m.view(torch.uint8), src=src)   # byte view: neither RCCL nor gloo has a 16-bit integer type
    return out_calib, maps


def max_over_ranks(seconds, device):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
