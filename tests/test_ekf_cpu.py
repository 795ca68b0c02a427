"""CPU: include/hnet_ekf.h (the filter update that consumes the network output) against the numpy
restatement oracle/ekf_oracle.py of cuahn::UpdaterHNet::update."""
import os
import subprocess

import numpy as np

from oracle import ekf_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "ekf_check.bin")
NSTATE = 3 + 4 + 3 + 3 + 3 + 12 + 729


def _flat(s):
    return np.concatenate([s["p"], s["q"], s["v"], s["ba"], s["bg"], s["offset"].reshape(-1), s["cov"].reshape(-1)])


def _cases(n, rng):
    out = []
    for i in range(n):
        a = rng.standard_normal((27, 27))
        cov = a @ a.T * 1e-3 + np.eye(27) * 1e-4
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        st = dict(p=rng.standard_normal(3), q=q, v=rng.standard_normal(3), ba=rng.standard_normal(3) * 0.1, bg=rng.standard_normal(3) * 0.01,
                  offset=rng.standard_normal((4, 3)) * 0.02, cov=cov)
        b = rng.standard_normal((8, 2))
        ncov = np.zeros((8, 8))
        for c in range(4):                                   # block diagonal like the network's output
            m = rng.standard_normal((2, 2))
            ncov[2 * c:2 * c + 2, 2 * c:2 * c + 2] = m @ m.T + np.eye(2) * 0.5
        mean = rng.standard_normal(8) * 5.0
        prop = st["offset"][:, :2].reshape(8).copy()
        out.append((st, mean, ncov, prop, 10.0 if i % 3 else 1.0, i % 2 == 0))
    # a zero innovation / zero rotation update (the reference divides 0/0 there)
    st = dict(out[0][0])
    st["cov"] = np.eye(27) * 1e-3
    out.append((st, out[0][3] * ekf_oracle.F_PIX, out[0][2], out[0][3], 10.0, True))
    return out


def _build():
    subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "ekf_check.cpp"), "-o", BIN], check=True)


def test_ekf_update_matches_numpy_restatement(tmp_path):
    _build()
    cases = _cases(12, np.random.default_rng(7))
    blob = [np.array([float(len(cases))])]
    for st, mean, ncov, prop, k, upd in cases:
        blob += [_flat(st), mean, ncov.reshape(-1), prop, np.array([k, 1.0 if upd else 0.0])]
    fin, fout = tmp_path / "in.f64", tmp_path / "out.f64"
    np.concatenate(blob).astype("<f8").tofile(fin)
    subprocess.run([BIN, str(fin), str(fout)], check=True, timeout=60)
    got = np.fromfile(fout, "<f8").reshape(len(cases), 1 + 2 * NSTATE)
    for (st, mean, ncov, prop, k, upd), g in zip(cases, got):
        assert g[0] == 1.0
        ref = ekf_oracle.update(st, mean, ncov, prop, k, upd)
        r1 = _flat(ref)
        assert np.abs(g[1:1 + NSTATE] - r1).max() < 1e-11 * max(1.0, np.abs(r1).max())
        r2 = _flat(ekf_oracle.reset_4pt_offset(ref))
        assert np.abs(g[1 + NSTATE:] - r2).max() < 1e-11 * max(1.0, np.abs(r2).max())
        # sanity of the restated update itself: the posterior offsets move towards the measurement
        if upd:
            before = np.abs(mean / ekf_oracle.F_PIX - prop)
            after = np.abs(mean / ekf_oracle.F_PIX - ref["offset"][:, :2].reshape(8))
            assert after.sum() < before.sum() + 1e-12


def test_prior_propagation_matches_numpy_restatement(tmp_path):
    """Propagator::predict_mean_discrete (the corner-offset prior): C++ header vs numpy restatement, and the
    physics of the restated model: a camera looking straight down, moving forward over the ground plane, sees every corner
    drift backwards by v/h per second (normalised coordinates)"""
    _build()
    rng = np.random.default_rng(11)
    cases = []
    for _ in range(10):
        q = rng.standard_normal(4) * np.array([1.0, 0.2, 0.2, 0.2])
        q /= np.linalg.norm(q)
        st = dict(p=np.array([0.0, 0.0, 0.0]) + rng.standard_normal(3) * 0.3 + np.array([0, 0, 1.5]), q=q, v=rng.standard_normal(3), ba=np.zeros(3),
                  bg=np.zeros(3), offset=rng.standard_normal((4, 3)) * 0.01, cov=np.eye(27))
        a = rng.standard_normal((3, 3))
        c_R_i, _ = np.linalg.qr(a)
        cases.append((st, c_R_i, rng.standard_normal(3) * 0.05, 0.005 + 0.01 * rng.random(), rng.standard_normal(3) * 0.5, rng.standard_normal(3) * 2.0))
    blob = [np.array([float(len(cases))])]
    for st, c_R_i, t, dt, w, a in cases:
        blob += [st["p"], st["q"], st["v"], st["ba"], st["bg"], st["offset"].reshape(-1), c_R_i.reshape(-1), t, np.array([dt]), w, a]
    fin, fout = tmp_path / "in.f64", tmp_path / "out.f64"
    np.concatenate(blob).astype("<f8").tofile(fin)
    subprocess.run([BIN, str(fin), str(fout), "prop"], check=True, timeout=60)
    got = np.fromfile(fout, "<f8").reshape(len(cases), 22)
    for (st, c_R_i, t, dt, w, a), g in zip(cases, got):
        ref = ekf_oracle.propagate_mean(st, c_R_i, t, dt, w, a)
        want = np.concatenate([ref["p"], ref["q"], ref["v"], ref["offset"].reshape(-1)])
        assert np.abs(g - want).max() < 1e-13 * max(1.0, np.abs(want).max())
    # nadir camera, height 2 (dc = (R (p + t))_z with the plane normal (0, 0, -1)), pure forward speed 1 along the camera x axis
    st = dict(p=np.array([0.0, 0.0, 2.0]), q=np.array([1.0, 0, 0, 0]), v=np.array([1.0, 0.0, 0.0]), ba=np.zeros(3), bg=np.zeros(3),
              offset=np.zeros((4, 3)), cov=np.eye(27))
    c_R_i = np.array([[1.0, 0, 0], [0, -1.0, 0], [0, 0, -1.0]])      # camera z = body -z: looks down
    out = ekf_oracle.propagate_mean(st, c_R_i, np.zeros(3), 0.01, np.zeros(3), np.array([0.0, 0.0, 9.81]))
    assert np.allclose(out["offset"][:, 0], -0.01 * 1.0 / 2.0, atol=1e-12) and np.allclose(out["offset"][:, 1:], 0.0, atol=1e-12)
