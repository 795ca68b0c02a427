"""CPU: the UZH-FPV replay generator (cuahn_vio_amd/replay.py, BASELINE.json config 5) — fixture, geometry, priors.

The prior of a pair is the corner-offset dynamics of Propagator::predict_mean_discrete (Propagator.cpp:342-364) integrated
over the frame interval; two independent derivations must agree with it: (1) include/hnet_ekf.h / oracle/ekf_oracle.py step
for step, and (2) the plane-induced homography between the two camera poses (closed form, no integration)."""
import os
import zlib

import numpy as np

from cuahn_vio_amd import replay
from oracle import ekf_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fx():
    return replay.load_fixture("indoor_forward_7")


def test_fixture_is_the_reference_trajectory_resampled():
    fx = _fx()
    assert fx["name"] == "indoor_forward_7" and str(fx["source"]) == "indoor_forward_7_snapdragon_with_gt.txt"
    n = fx["t"].shape[0]
    assert n == 640 and fx["p"].shape == (n, 3) and fx["q_xyzw"].shape == (n, 4)
    assert np.allclose(np.diff(fx["t"]), 1.0 / 30.0) and np.allclose(np.linalg.norm(fx["q_xyzw"], axis=1), 1.0, atol=1e-9)
    # launch constants (uzhfpv.launch:75-91)
    assert tuple(fx["cam0_wh"]) == (640, 480) and abs(fx["cam0_k"][0] - 275.46015578667294) < 1e-12
    assert np.allclose(fx["c_R_i"] @ fx["c_R_i"].T, np.eye(3), atol=1e-6)
    assert os.path.getsize(os.path.join(ROOT, "tests", "golden", "replay_indoor_forward_7.npz")) < 64 * 1024     # KB-scale
    assert (fx["p"][:, 2] - fx["floor_z"]).min() > 0.5                  # the camera stays above the floor


def test_corner_step_is_the_filters_mean_propagation():
    """replay.corner_step == the offset part of ekf_oracle.propagate_mean (numpy restatement of Propagator.cpp:211-220,342-364),
    which tests/test_ekf_cpu.py pins against include/hnet_ekf.h"""
    rng = np.random.default_rng(5)
    fx = _fx()
    for _ in range(20):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        st = dict(p=rng.standard_normal(3) + np.array([0, 0, 3.0]), q=q, v=rng.standard_normal(3), ba=np.zeros(3), bg=np.zeros(3),
                  offset=rng.standard_normal((4, 3)) * 0.02, cov=np.zeros((27, 27)))
        st["offset"][:, 2] = 0.0
        w, a, dt = rng.standard_normal(3) * 0.5, rng.standard_normal(3), 0.002
        ref = ekf_oracle.propagate_mean(st, fx["c_R_i"], fx["i_t_i2c"], dt, w, a)
        R = ekf_oracle.ham_quat_2_rot(st["q"])
        wc = fx["c_R_i"] @ w
        vc = fx["c_R_i"] @ (st["v"] + np.cross(w, fx["i_t_i2c"]))
        muc = fx["c_R_i"] @ R.T @ np.array([0.0, 0.0, -1.0])
        dc = (R @ (st["p"] + fx["i_t_i2c"]))[2]
        got = replay.corner_step(st["offset"], dt, wc, vc, muc, dc)
        assert np.abs(got - ref["offset"]).max() < 1e-14


def test_prior_agrees_with_the_plane_induced_homography():
    """the integrated prior (pixels) vs the closed-form 4-corner offsets of H = G_{k+1}^-1 G_k: Euler integration at 16 substeps
    per frame (the filter integrates at the ~500 Hz IMU rate) stays within 3 % + 0.05 px of the exact offsets (worst case: the
    180 px/frame flip manoeuvre around frame 39; median motion is 25 px/frame)"""
    fx = _fx()
    worst = 0.0
    for k in range(0, 639, 13):
        to, pr = replay.true_offsets(fx, k), replay.prior_offsets(fx, k)
        err = np.abs(to - pr).max()
        assert err < 0.05 + 0.03 * np.abs(to).max(), (k, err, np.abs(to).max())
        worst = max(worst, err / max(np.abs(to).max(), 1e-9))
    assert worst < 0.03
    # more substeps -> closer (first order)
    k = 50
    e16 = np.abs(replay.true_offsets(fx, k) - replay.prior_offsets(fx, k, 16)).max()
    e128 = np.abs(replay.true_offsets(fx, k) - replay.prior_offsets(fx, k, 128)).max()
    assert e128 < 0.3 * e16


def test_rendered_pairs_are_geometrically_consistent_and_reproducible():
    fx = _fx()
    prev, curr, prior = replay.render_pairs(fx, 300, 3)
    assert prev.dtype == np.uint8 and prev.shape == (3, 224, 320) and prior.shape == (3, 8) and prior.dtype == np.float32
    assert np.array_equal(curr[0], prev[1]) and np.array_equal(curr[1], prev[2])       # consecutive frames of ONE stream
    assert zlib.crc32(prev.tobytes()) == zlib.crc32(replay.render_pairs(fx, 300, 3)[0].tobytes())
    assert prev.std() > 20                                                            # textured, not flat
    # photometric check: sampling curr at H * x reproduces prev (same ground texture), well inside the overlap
    H = replay.pair_homography(fx, 300)
    vs, us = np.meshgrid(np.arange(60, 164, dtype=np.float64), np.arange(80, 240, dtype=np.float64), indexing="ij")
    X, Y, Z = (H[0, 0] * us + H[0, 1] * vs + H[0, 2]), (H[1, 0] * us + H[1, 1] * vs + H[1, 2]), (H[2, 0] * us + H[2, 1] * vs + H[2, 2])
    x, y = X / Z, Y / Z
    ok = (x > 1) & (x < 318) & (y > 1) & (y < 222)
    x0, y0 = np.floor(np.where(ok, x, 1)).astype(int), np.floor(np.where(ok, y, 1)).astype(int)
    fx_, fy_ = np.where(ok, x, 1) - x0, np.where(ok, y, 1) - y0
    c = curr[0].astype(np.float64)
    samp = (c[y0, x0] * (1 - fx_) + c[y0, x0 + 1] * fx_) * (1 - fy_) + (c[y0 + 1, x0] * (1 - fx_) + c[y0 + 1, x0 + 1] * fx_) * fy_
    d = np.abs(samp - prev[0, 60:164, 80:240].astype(np.float64))[ok]
    assert ok.mean() > 0.5 and np.median(d) < 6.0        # grey levels: bilinear resampling of a 1-texel octave + u8 rounding
