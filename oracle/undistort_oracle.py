"""numpy restatement of the image pre-processing ahead of load_current_img (TEST INFRASTRUCTURE).

CamBase::initialize_undist_map / initialize_undist_map_fisheye (ov_core/src/cam/CamBase.h:165-180) build two float maps
with OpenCV towards the virtual camera f = (320-1)/2 / tan(45 deg) = 159.5, c = (159.5, 111.5); undistort_and_resize_img
(:182-186) is cv::remap(INTER_LINEAR).  OpenCV is a third-party dependency that is neither in this image nor vendored by the
reference (README.md lists it as a prerequisite; no pinned version) and the reference holds no vectors for this step:
PARITY UNPINNED.  The formulas below are the published ones of cv::fisheye::initUndistortRectifyMap (equidistant model:
theta_d = theta (1 + k1 theta^2 + k2 theta^4 + k3 theta^6 + k4 theta^8)) and cv::initUndistortRectifyMap (radial-tangential,
D = (k1, k2, p1, p2)); the remap quantises sample positions to 1/32 px like cv::remap (INTER_BITS = 5).
"""
import numpy as np

ROWS, COLS = 224, 320
F_VIRTUAL = (COLS - 1.0) / 2.0 / np.tan(np.deg2rad(45.0))
CX, CY = (COLS - 1.0) / 2.0, (ROWS - 1.0) / 2.0


def build_maps(k, d, fisheye=True):
    """k = (fx, fy, cx, cy), d = distortion -> (map_x, map_y) float32 [224, 320]: where each output pixel samples the raw image"""
    v, u = np.meshgrid(np.arange(ROWS, dtype=np.float64), np.arange(COLS, dtype=np.float64), indexing="ij")
    x, y = (u - CX) / F_VIRTUAL, (v - CY) / F_VIRTUAL
    if fisheye:
        r = np.sqrt(x * x + y * y)
        th = np.arctan(r)
        t2 = th * th
        thd = th * (1.0 + t2 * (d[0] + t2 * (d[1] + t2 * (d[2] + t2 * d[3]))))
        sc = np.where(r == 0.0, 1.0, thd / np.where(r == 0.0, 1.0, r))
        xd, yd = x * sc, y * sc
    else:
        r2 = x * x + y * y
        kr = 1.0 + r2 * (d[0] + r2 * d[1])
        xd = x * kr + 2.0 * d[2] * x * y + d[3] * (r2 + 2.0 * x * x)
        yd = y * kr + d[2] * (r2 + 2.0 * y * y) + 2.0 * d[3] * x * y
    return (k[0] * xd + k[2]).astype(np.float32), (k[1] * yd + k[3]).astype(np.float32)


def remap(raw, map_x, map_y):
    """bilinear, positions rounded (half to even) to 1/32 px, zero outside the image, integer blend (sum + 512) >> 10"""
    raw = np.asarray(raw, np.uint8)
    rows, cols = raw.shape
    fx, fy = map_x.astype(np.float32) * np.float32(32.0), map_y.astype(np.float32) * np.float32(32.0)
    sane = (np.abs(fx) < 1e9) & (np.abs(fy) < 1e9)
    sx = np.where(sane, np.rint(np.where(sane, fx, 0)), -(1 << 20)).astype(np.int64)
    sy = np.where(sane, np.rint(np.where(sane, fy, 0)), -(1 << 20)).astype(np.int64)
    x0, y0, ax, ay = sx >> 5, sy >> 5, sx & 31, sy & 31

    def tap(y, x):
        ok = (y >= 0) & (y < rows) & (x >= 0) & (x < cols)
        return np.where(ok, raw[np.clip(y, 0, rows - 1), np.clip(x, 0, cols - 1)].astype(np.int64), 0)

    val = tap(y0, x0) * (32 - ax) * (32 - ay) + tap(y0, x0 + 1) * ax * (32 - ay) + tap(y0 + 1, x0) * (32 - ax) * ay + tap(y0 + 1, x0 + 1) * ax * ay
    return ((val + 512) >> 10).astype(np.uint8)
