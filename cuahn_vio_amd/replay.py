"""UZH-FPV trajectory replay (BASELINE.json config 5): frame pairs rendered along a ground-truth
trajectory the reference ships, with the priors the filter would hand to the network.

The real image bags are not available (cuahn/launch/uzhfpv.launch:9 points to the author's disk), so the frames are
synthesised: a procedural texture painted on the ground plane is seen by the 320x224, f = 159.5 virtual camera
(ov_core/src/cam/CamBase.h:165-169 — the camera every raw frame is undistorted to before the network sees it) from the body
poses of the committed fixture tests/golden/replay_<seq>.npz (tools/make_replay_fixture.py: the reference's
ov_data/uzh_fpv/<seq>_snapdragon_with_gt.txt resampled to 30 Hz) through the launch file's camera extrinsics
(uzhfpv.launch:84-91).  Consecutive frames form the (prev, curr) pairs, exactly as load_current_img / network_inference see
them (VioManager.cpp:188,236).

Prior of pair (k, k+1) = the corner offsets the filter's mean propagation produces over that interval, in pixels:
Propagator::predict_mean_discrete (cuahn/src/state/Propagator.cpp:342-364) integrated from zero offsets
(State::reset_4pt_offset after every update, State.cpp:101-111) with the body rates of the trajectory, then x 159.5
(VioManager.cpp:230-234).  The corner dynamics are restated here in numpy (host-side data generation, no oracle import);
tests/test_replay.py checks them against include/hnet_ekf.h and against the plane-induced homography of the two poses.
"""
from __future__ import annotations

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IMG_H, IMG_W = 224, 320
F_PIX = 159.5                                   # (320 - 1) / 2 / tan(45 deg), CamBase.h:166-169
K_VIRT = np.array([[F_PIX, 0.0, (IMG_W - 1) / 2.0], [0.0, F_PIX, (IMG_H - 1) / 2.0], [0.0, 0.0, 1.0]])
# the four image corners in normalised camera coordinates, order ul, bl, br, ur (State.h:110-113)
CORNERS = np.array([[-1.0, -0.69906, 1.0], [-1.0, 0.69906, 1.0], [1.0, 0.69906, 1.0], [1.0, -0.69906, 1.0]])
P4 = np.array([[0.0, 0.0], [0.0, IMG_H - 1.0], [IMG_W - 1.0, IMG_H - 1.0], [IMG_W - 1.0, 0.0]])
TEXELS_PER_M = 100.0                            # ground texture resolution: 1 texel = 1 cm
SKY = 128


def load_fixture(seq: str) -> dict:
    path = seq if os.path.exists(seq) else os.path.join(ROOT, "tests", "golden", f"replay_{seq}.npz")
    z = np.load(path)
    fx = {k: z[k] for k in z.files}
    fx["name"] = str(fx["name"])
    T = fx["T_ItoC"]
    fx["c_R_i"] = T[:3, :3].copy()                                  # camera <- IMU rotation (State.h:108)
    fx["i_t_i2c"] = -T[:3, :3].T @ T[:3, 3]                          # camera origin in the IMU frame (State.h:107)
    return fx


def quat_to_rot(q_xyzw: np.ndarray) -> np.ndarray:
    """Hamilton quaternion (x, y, z, w) of the body in the world -> R_WB"""
    x, y, z, w = q_xyzw
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def camera_pose(fx: dict, k: int):
    """(R_WC, p_WC) of frame k"""
    R_wb = quat_to_rot(fx["q_xyzw"][k])
    return R_wb @ fx["c_R_i"].T, fx["p"][k] + R_wb @ fx["i_t_i2c"]


def ground_homography(fx: dict, k: int) -> np.ndarray:
    """G: pixel (u, v, 1) of frame k -> (X, Y, s) with ground point (X/s, Y/s) in metres (plane z = floor_z); s <= 0: no hit"""
    R_wc, p_wc = camera_pose(fx, k)
    h = p_wc[2] - float(fx["floor_z"])                              # camera height above the floor
    M = R_wc @ np.linalg.inv(K_VIRT)                                 # ray direction in the world
    # point = p + t * d with p_z + t d_z = floor_z  =>  t = -h / d_z ; (X, Y) = p_xy + t d_xy  =>  homogeneous with s = -d_z
    G = np.stack([p_wc[0] * (-M[2]) + h * M[0], p_wc[1] * (-M[2]) + h * M[1], -M[2]])
    return G


def pair_homography(fx: dict, k: int) -> np.ndarray:
    """H mapping pixels of frame k to pixels of frame k + 1 (the network's convention, model_to_trace.py:148), plane induced"""
    G0, G1 = ground_homography(fx, k), ground_homography(fx, k + 1)
    H = np.linalg.inv(G1) @ G0
    return H / H[2, 2]


def true_offsets(fx: dict, k: int) -> np.ndarray:
    """4-corner offsets (px, order ul.u ul.v bl.u ... ur.v) of the plane-induced homography of pair (k, k+1)"""
    H = pair_homography(fx, k)
    q = (H @ np.c_[P4, np.ones(4)].T).T
    return (q[:, :2] / q[:, 2:3] - P4).reshape(8)


def _skew(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def corner_step(offset: np.ndarray, dt: float, wc: np.ndarray, vc: np.ndarray, muc: np.ndarray, dc: float) -> np.ndarray:
    """one discrete step of the corner dynamics, Propagator.cpp:357-362: offset += dt * (-(I - pt ez^T) H pt),
    H = skew(wc) + vc muc^T / dc (:356), pt = corner + offset (:217-220).  offset [4,3] (x, y, z) normalised camera units"""
    Hm = _skew(wc) + np.outer(vc, muc) / dc
    out = offset.copy()
    for c in range(4):
        pt = CORNERS[c] + offset[c]
        hp = Hm @ pt
        out[c] = offset[c] + dt * (-(hp - pt * hp[2]))
    return out


def prior_offsets(fx: dict, k: int, substeps: int = 16) -> np.ndarray:
    """prior of pair (k, k+1) in pixels: the corner dynamics integrated over the frame interval from zero offsets with the
    body rates of the trajectory (constant angular rate = the relative rotation's rotation vector / dt; velocity = the
    position difference / dt), the other state (attitude, height) taken from the interpolated trajectory at every substep"""
    dt = float(fx["t"][k + 1] - fx["t"][k])
    R0, R1 = quat_to_rot(fx["q_xyzw"][k]), quat_to_rot(fx["q_xyzw"][k + 1])
    dR = R0.T @ R1                                                   # body k+1 expressed in body k
    ang = np.arccos(np.clip((np.trace(dR) - 1) / 2, -1.0, 1.0))
    axis = np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]])
    w_body = axis / (2 * np.sin(ang)) * ang / dt if ang > 1e-9 else axis / (2 * dt)
    v_world = (fx["p"][k + 1] - fx["p"][k]) / dt
    c_R_i, t_i2c = fx["c_R_i"], fx["i_t_i2c"]
    off = np.zeros((4, 3))
    h = dt / substeps
    for s in range(substeps):
        a = (s + 0.5) / substeps                                     # midpoint state of the substep
        th = ang * a
        if ang > 1e-9:
            ax = axis / (2 * np.sin(ang))
            Kx = _skew(ax)
            Ra = R0 @ (np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx)
        else:
            Ra = R0
        p_w = fx["p"][k] + a * (fx["p"][k + 1] - fx["p"][k])
        v_body = Ra.T @ v_world
        wc = c_R_i @ w_body                                          # :212
        vc = c_R_i @ (v_body + np.cross(w_body, t_i2c))              # :213
        muc = c_R_i @ Ra.T @ np.array([0.0, 0.0, -1.0])              # :214 ground normal (pointing down) in the camera frame
        dc = (p_w + Ra @ t_i2c)[2] - float(fx["floor_z"])            # :215 height of the camera above the ground
        off = corner_step(off, h, wc, vc, muc, dc)
    return (off[:, :2] * F_PIX).reshape(8)                            # VioManager.cpp:230-234


# ---------------------------------------------------------------------------------------------- texture + rendering
def _hash_lattice(ix, iy, seed):
    x = (ix.astype(np.int64).astype(np.uint64) * np.uint64(0x9E3779B1) + iy.astype(np.int64).astype(np.uint64) * np.uint64(0x85EBCA77)
         + np.uint64((seed & 0xFFFFFFFF) * 0xC2B2AE3D & 0xFFFFFFFFFFFFFFFF)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(12)
    x = (x * np.uint64(0x297A2D39)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    return (x & np.uint64(0xFF)).astype(np.float64)


def ground_texture(x_m: np.ndarray, y_m: np.ndarray, seed: int = 7) -> np.ndarray:
    """multi-octave value noise over the (unbounded) ground plane, in [0, 255]; coordinates in metres"""
    u, v = x_m * TEXELS_PER_M, y_m * TEXELS_PER_M
    tot = np.zeros_like(u)
    for wgt, cell, s in ((4.0, 64.0, 1), (3.0, 16.0, 2), (2.0, 4.0, 3), (1.0, 1.5, 4)):
        a, b = u / cell, v / cell
        a0, b0 = np.floor(a), np.floor(b)
        fa, fb = a - a0, b - b0
        v00 = _hash_lattice(a0, b0, seed * 4 + s)
        v10 = _hash_lattice(a0 + 1, b0, seed * 4 + s)
        v01 = _hash_lattice(a0, b0 + 1, seed * 4 + s)
        v11 = _hash_lattice(a0 + 1, b0 + 1, seed * 4 + s)
        tot += wgt * ((v00 * (1 - fa) + v10 * fa) * (1 - fb) + (v01 * (1 - fa) + v11 * fa) * fb)
    t = tot / 10.0
    return np.clip((t - 64.0) * 2.0, 0.0, 255.0)                     # stretch the contrast


def render_frame(fx: dict, k: int) -> np.ndarray:
    """u8 [224, 320]: the ground texture seen from pose k; pixels whose ray does not hit the ground are SKY"""
    G = ground_homography(fx, k)
    vs, us = np.meshgrid(np.arange(IMG_H, dtype=np.float64), np.arange(IMG_W, dtype=np.float64), indexing="ij")
    X = G[0, 0] * us + G[0, 1] * vs + G[0, 2]
    Y = G[1, 0] * us + G[1, 1] * vs + G[1, 2]
    S = G[2, 0] * us + G[2, 1] * vs + G[2, 2]
    hit = S > 1e-6
    Ss = np.where(hit, S, 1.0)
    img = np.where(hit, ground_texture(X / Ss, Y / Ss), float(SKY))
    return np.floor(img + 0.5).astype(np.uint8)


def render_pairs(fx: dict, first: int, count: int):
    """(prev u8 [count,224,320], curr u8 [count,224,320], prior f32 [count,8]) for pairs (first+i, first+i+1)"""
    n = fx["t"].shape[0]
    if first + count + 1 > n:
        first = first % max(n - count - 1, 1)
    frames = [render_frame(fx, first + i) for i in range(count + 1)]
    prev = np.stack(frames[:-1])
    curr = np.stack(frames[1:])
    prior = np.stack([prior_offsets(fx, first + i) for i in range(count)]).astype(np.float32)
    return prev, curr, prior
