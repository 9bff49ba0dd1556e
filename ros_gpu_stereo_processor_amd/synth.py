"""Deterministic synthetic stereo inputs (numpy only) for the parity tests and the benchmark.

Scene: the rectified left image is multi-octave value noise (every block textured); the disparity
field is a background plane plus a few fronto-parallel rectangles with integer disparities; right(x) = left(x + d);
optionally both views are warped back through the calibration ("inverse rectification") so that the pipeline's own
remap restores them.  Calibration = the reference fixture's intrinsics (test/stereobm/test_data/left.yaml, right.yaml)
scaled to the requested frame size.
"""
import numpy as np

# the numbers of the reference's calibration fixtures (data, 752x480)
_FIX_W, _FIX_H = 752, 480
_K_L = [463.241134, 0.0, 385.760092, 0.0, 463.072782, 221.039882, 0.0, 0.0, 1.0]
_D_L = [-0.374450, 0.109252, 0.000282, -0.001548, 0.0]
_R_L = [0.997310, 0.005745, 0.073072, -0.006100, 0.999971, 0.004637, -0.073043, -0.005070, 0.997316]
_P_L = [441.238411, 0.0, 322.811100, 0.0, 0.0, 441.238411, 230.623768, 0.0, 0.0, 0.0, 1.0, 0.0]
_K_R = [462.751523, 0.0, 347.278408, 0.0, 462.070418, 236.614124, 0.0, 0.0, 1.0]
_D_R = [-0.379317, 0.114524, -0.000156, -0.001440, 0.0]
_R_R = [0.998747, 0.009973, 0.049045, -0.009735, 0.999940, -0.005097, -0.049093, 0.004613, 0.998784]
_P_R = [441.238411, 0.0, 322.811100, -44.133133, 0.0, 441.238411, 230.623768, 0.0, 0.0, 0.0, 1.0, 0.0]


def scaled_calibration(width, height):
    """Fixture intrinsics scaled by (width/752, height/480); D and R unchanged; Tx*fx scales with fx (baseline kept).
    Returns dict(left=dict(K,D,R,P), right=..., width, height)."""
    sx, sy = width / _FIX_W, height / _FIX_H

    def scale(K, D, R, P):
        K = np.array(K, np.float64).reshape(3, 3).copy()
        P = np.array(P, np.float64).reshape(3, 4).copy()
        K[0] *= sx
        K[1] *= sy
        P[0] *= sx
        P[1] *= sy
        return dict(K=K.reshape(9), D=np.array(D, np.float64), R=np.array(R, np.float64), P=P.reshape(12))

    return dict(left=scale(_K_L, _D_L, _R_L, _P_L), right=scale(_K_R, _D_R, _R_R, _P_R), width=width, height=height)


def value_noise(h, w, rng, octaves=(4, 16, 64)):
    """3-octave value noise in [0,255]: random grids at the given cell sizes, bilinearly upsampled, summed."""
    acc = np.zeros((h, w), np.float64)
    for cell in octaves:
        gh, gw = h // cell + 2, w // cell + 2
        g = rng.random((gh, gw))
        ys, xs = np.arange(h) / cell, np.arange(w) / cell
        y0, x0 = ys.astype(int), xs.astype(int)
        fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
        a = g[y0][:, x0]
        b = g[y0][:, x0 + 1]
        c = g[y0 + 1][:, x0]
        d = g[y0 + 1][:, x0 + 1]
        acc += (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy
    acc /= len(octaves)
    acc = (acc - acc.min()) / max(acc.max() - acc.min(), 1e-9)
    return acc * 255.0


def disparity_field(h, w, ndisp, rng, nrect=5, background=8):
    d = np.full((h, w), min(background, ndisp - 1), np.int32)
    for _ in range(nrect):
        rh, rw = int(rng.integers(h // 6, h // 2)), int(rng.integers(w // 8, w // 3))
        y0, x0 = int(rng.integers(0, h - rh)), int(rng.integers(0, w - rw))
        d[y0:y0 + rh, x0:x0 + rw] = int(rng.integers(4, max(5, ndisp - 8)))
    return d


def rectified_pair(width, height, ndisp, seed, noise_sigma=2.0, nrect=5):
    """-> (left_rect u8, right_rect u8, true disparity int32) with right(x) = left(x + d(x)) forward-warped,
    occlusions filled from the background texture."""
    rng = np.random.default_rng(seed)
    pad = ndisp + 8
    tex = value_noise(height, width + pad, rng)
    left = tex[:, :width]
    disp = disparity_field(height, width, ndisp, rng, nrect=nrect)
    # right view: background layer first (texture shifted by the background disparity), then nearer layers on top
    right = np.empty((height, width), np.float64)
    bg = int(disp.min())
    xs = np.arange(width)
    right[:] = tex[:, np.clip(xs + bg, 0, width + pad - 1)]
    for dv in sorted(set(np.unique(disp).tolist()) - {bg}):
        ys_, xs_ = np.nonzero(disp == dv)
        xr = xs_ - dv
        ok = xr >= 0
        right[ys_[ok], xr[ok]] = left[ys_[ok], xs_[ok]]
    if noise_sigma > 0:
        left = left + rng.normal(0, noise_sigma, left.shape)
        right = right + rng.normal(0, noise_sigma, right.shape)
    to_u8 = lambda a: np.clip(np.rint(a), 0, 255).astype(np.uint8)  # noqa: E731
    return to_u8(left), to_u8(right), disp


def _undistort_normalized(xd, yd, D, iters=12):
    k1, k2, p1, p2, k3 = (list(D) + [0, 0, 0, 0, 0])[:5]
    x, y = xd.copy(), yd.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        radial = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (xd - dx) / radial
        y = (yd - dy) / radial
    return x, y


def unrectify(rect_img, cam, width, height):
    """Warp a rectified image back into the raw (distorted, unrotated) camera: for every raw pixel find where it lands in
    the rectified frame (undistort -> R -> P) and sample bilinearly.  Out-of-frame samples replicate the border."""
    K = np.asarray(cam["K"]).reshape(3, 3)
    R = np.asarray(cam["R"]).reshape(3, 3)
    P = np.asarray(cam["P"]).reshape(3, 4)
    v, u = np.mgrid[0:height, 0:width].astype(np.float64)
    xd, yd = (u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1]
    x, y = _undistort_normalized(xd, yd, cam["D"])
    X = R[0, 0] * x + R[0, 1] * y + R[0, 2]
    Y = R[1, 0] * x + R[1, 1] * y + R[1, 2]
    W = R[2, 0] * x + R[2, 1] * y + R[2, 2]
    ur = P[0, 0] * X / W + P[0, 2]
    vr = P[1, 1] * Y / W + P[1, 2]
    src = rect_img.astype(np.float64)
    ur = np.clip(ur, 0, width - 1.001)
    vr = np.clip(vr, 0, height - 1.001)
    x0, y0 = np.floor(ur).astype(int), np.floor(vr).astype(int)
    fx, fy = ur - x0, vr - y0
    if src.ndim == 3:
        fx, fy = fx[..., None], fy[..., None]
    out = (src[y0, x0] * (1 - fx) + src[y0, x0 + 1] * fx) * (1 - fy) + (src[y0 + 1, x0] * (1 - fx) + src[y0 + 1, x0 + 1] * fx) * fy
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def raw_pair(width, height, ndisp, seed, calib=None, noise_sigma=2.0):
    """-> (left_raw, right_raw, calib, true_disparity): mono8 raw views that rectify (approximately) to rectified_pair()."""
    calib = calib or scaled_calibration(width, height)
    l, r, d = rectified_pair(width, height, ndisp, seed, noise_sigma=noise_sigma)
    return unrectify(l, calib["left"], width, height), unrectify(r, calib["right"], width, height), calib, d


def bayer_mosaic(bgr, encoding):
    """Sample a BGR image into a Bayer mosaic with the given ROS encoding."""
    tiles = {"bayer_rggb8": (2, 1, 1, 0), "bayer_bggr8": (0, 1, 1, 2), "bayer_gbrg8": (1, 0, 2, 1), "bayer_grbg8": (1, 2, 0, 1)}[encoding]
    h, w = bgr.shape[:2]
    out = np.empty((h, w), np.uint8)
    for i, ch in enumerate(tiles):
        yy, xx = i // 2, i % 2
        out[yy::2, xx::2] = bgr[yy::2, xx::2, ch]
    return out


def colorize(gray, seed):
    """A plausible BGR image whose luminance pattern follows `gray` (for Bayer / colour inputs)."""
    rng = np.random.default_rng(seed)
    h, w = gray.shape
    tint = value_noise(h, w, rng, octaves=(32, 128)) / 255.0
    g = gray.astype(np.float64)
    b = np.clip(g * (0.7 + 0.6 * tint), 0, 255)
    r = np.clip(g * (1.3 - 0.6 * tint), 0, 255)
    return np.stack([np.rint(b), g, np.rint(r)], -1).astype(np.uint8)
