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


# ---------------------------------------------------------------------------------------------- f-2: covariance propagation
C_R_I = np.array([[-0.027256691772188965, -0.9996260641688061, 0.0021919370477445077],
                  [-0.7139206120417471, 0.017931469899155242, -0.6999970157716363],
                  [0.6996959571525168, -0.020644471939022302, -0.714142404092339]])      # uzhfpv.launch:84-91
T_I2C = -C_R_I.T @ np.array([0.02422852666805565, 0.008974432843748055, -0.000638971731537894])


def _hprod(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def _boxminus_q(q1, q0):
    """rotation vector d with q1 = q0 (x) dq(d): the attitude error of this filter is applied on the right
    (Ham_quat_update(d) * q is the matrix form of q (x) dq, quat_ops.h:526-538)"""
    r = _hprod(np.array([q0[0], -q0[1], -q0[2], -q0[3]]), q1)
    return 2.0 * (r[1:] if r[0] >= 0 else -r[1:])


def _prop_state(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    a = rng.standard_normal((27, 27))
    st = dict(p=np.array([0.3, -0.2, 2.5]) + rng.standard_normal(3) * 0.2, q=q, v=rng.standard_normal(3), ba=np.zeros(3), bg=np.zeros(3),
              offset=rng.standard_normal((4, 3)) * 0.02, cov=a @ a.T * 1e-4)
    st["offset"][:, 2] = 0.0
    return st


def test_transition_jacobian_equals_finite_differences_of_the_mean_propagation():
    """INDEPENDENT pin of the restated F (Propagator.cpp:222-319): every 3x3 block against central differences of
    propagate_mean — the mean map was pinned in round 1 (closed form for a nadir camera) and is a different piece of code than
    the Jacobian formulas.  All 81 blocks, five random states, 45-degree camera of the launch file."""
    rng = np.random.default_rng(1)
    h = 1e-6
    for _ in range(5):
        st = _prop_state(rng)
        dt, w, a = 0.002, rng.standard_normal(3) * 0.5, rng.standard_normal(3)
        F, _Fw = ekf_oracle.jacobians(st, C_R_I, T_I2C, dt, w)
        base = ekf_oracle.propagate_mean(st, C_R_I, T_I2C, dt, w, a)

        def perturbed(i, e):
            s = {k: np.array(v, float).copy() for k, v in st.items()}
            ww, aa = w.copy(), a.copy()
            if i < 3:
                s["p"][i] += e
            elif i < 6:
                d = np.zeros(3)
                d[i - 3] = e
                s["q"] = ekf_oracle.quatnorm(ekf_oracle.ham_quat_update(d) @ st["q"])
            elif i < 9:
                s["v"][i - 6] += e
            elif i < 12:
                aa[i - 9] -= e                       # a_hat = a_m - b_a
            elif i < 15:
                ww[i - 12] -= e                      # w_hat = w_m - b_g
            else:
                s["offset"][(i - 15) // 3, (i - 15) % 3] += e
            return ekf_oracle.propagate_mean(s, C_R_I, T_I2C, dt, ww, aa)

        def err_state(s1):
            return np.concatenate([s1["p"] - base["p"], _boxminus_q(s1["q"], base["q"]), s1["v"] - base["v"], np.zeros(6),
                                   (s1["offset"] - base["offset"]).reshape(-1)])

        Fd = np.zeros((27, 27))
        for i in range(27):
            Fd[:, i] = (err_state(perturbed(i, h)) - err_state(perturbed(i, -h))) / (2 * h)
        Fd[9:15, 9:15] = np.eye(6)                   # the biases are constant states
        assert np.abs(F - Fd).max() < 5e-9, np.abs(F - Fd).max()
        assert np.abs(F[15:, :15]).max() > 1e-4      # the corner rows do couple to the IMU states


def test_cpp_jacobians_and_covariance_propagation_match_numpy(tmp_path):
    _build()
    rng = np.random.default_rng(3)
    cases = []
    for _ in range(8):
        st = _prop_state(rng)
        cases.append((st, 0.002 + 0.003 * rng.random(), rng.standard_normal(3) * 0.5, rng.standard_normal(3) + np.array([0, 0, 9.81])))
    qd = np.diag(ekf_oracle.noise_q(0.00559017, 0.01118034, 8.94427e-04, 0.04472136))
    blob = [np.array([float(len(cases))])]
    for st, dt, w, a in cases:
        blob += [_flat(st), C_R_I.reshape(-1), T_I2C, np.array([dt]), w, a, qd]
    fin, fout = tmp_path / "in.f64", tmp_path / "out.f64"
    np.concatenate(blob).astype("<f8").tofile(fin)
    subprocess.run([BIN, str(fin), str(fout), "jac"], check=True, timeout=60)
    got = np.fromfile(fout, "<f8").reshape(len(cases), 729 + 405 + NSTATE)
    for (st, dt, w, a), g in zip(cases, got):
        F, Fw = ekf_oracle.jacobians(st, C_R_I, T_I2C, dt, w)
        assert np.abs(g[:729].reshape(27, 27) - F).max() < 1e-12
        assert np.abs(g[729:1134].reshape(27, 15) - Fw).max() < 1e-12
        ref = ekf_oracle.propagate_mean(st, C_R_I, T_I2C, dt, w, a)
        ref["cov"] = ekf_oracle.propagate_cov(st["cov"], F, Fw, np.diag(qd))
        assert np.abs(g[1134:] - _flat(ref)).max() < 1e-12
        assert np.abs(ref["cov"] - ref["cov"].T).max() < 1e-15       # F P F^T + Fw Q Fw^T stays symmetric


# ---------------------------------------------------------------------------------------------- f-1: independent derivations of the update
def test_update_equals_the_information_form_posterior():
    """UpdaterHNet::update computes K = P H^T S^-1, P+ = (I - K H) P, dx = K r (UpdaterHNet.cpp:31-41).  An independent route
    to the same posterior is the information form  P+ = (P^-1 + H^T R^-1 H)^-1,  dx = P+ H^T R^-1 r  (no gain matrix, no
    innovation covariance).  Both must agree, the posterior must be symmetric and  P - P+ = K S K^T  (positive semi-definite)."""
    rng = np.random.default_rng(11)
    for _ in range(6):
        a = rng.standard_normal((27, 27))
        P = a @ a.T * 1e-3 + np.eye(27) * 1e-5
        ncov = np.zeros((8, 8))
        for c in range(4):
            m = rng.standard_normal((2, 2))
            ncov[2 * c:2 * c + 2, 2 * c:2 * c + 2] = m @ m.T + np.eye(2) * 0.5
        q = rng.standard_normal(4)
        st = dict(p=rng.standard_normal(3), q=q / np.linalg.norm(q), v=rng.standard_normal(3), ba=np.zeros(3), bg=np.zeros(3),
                  offset=rng.standard_normal((4, 3)) * 0.02, cov=P)
        mean = rng.standard_normal(8) * 4.0
        prop = st["offset"][:, :2].reshape(8).copy()
        out = ekf_oracle.update(st, mean, ncov, prop, 10.0, True)
        H = ekf_oracle.H
        R = 10.0 * ncov / ekf_oracle.F_PIX ** 2
        P_info = np.linalg.inv(np.linalg.inv(P) + H.T @ np.linalg.inv(R) @ H)
        assert np.abs(out["cov"] - P_info).max() < 1e-9 * np.abs(P).max()
        assert np.abs(out["cov"] - out["cov"].T).max() < 1e-12 * np.abs(P).max()
        S = H @ P @ H.T + R
        K = P @ H.T @ np.linalg.inv(S)
        assert np.abs((P - out["cov"]) - K @ S @ K.T).max() < 1e-10 * np.abs(P).max()
        assert np.linalg.eigvalsh(P - out["cov"]).min() > -1e-12      # a measurement never increases the covariance
        dx = P_info @ H.T @ np.linalg.inv(R) @ (mean / ekf_oracle.F_PIX - prop)
        assert np.abs((out["p"] - st["p"]) - dx[0:3]).max() < 1e-9
        assert np.abs((out["v"] - st["v"]) - dx[6:9]).max() < 1e-9
        assert np.abs((out["offset"] - st["offset"]).reshape(-1) - dx[15:27]).max() < 1e-9


def test_update_hand_computed_scalar_case():
    """one measured component with everything else decoupled: the textbook scalar Kalman update, by hand.
    P = diag, only ul.x has prior variance p0; measurement variance r = K_net_Cov * c / 159.5^2; innovation y = z/159.5 - x0:
    gain k = p0 / (p0 + r), x+ = x0 + k y, p+ = (1 - k) p0 = p0 r / (p0 + r)"""
    P = np.eye(27) * 1e-12
    p0 = 4e-4
    P[15, 15] = p0
    ncov = np.eye(8) * 1e12              # the other seven components carry no information
    c = 2.5
    ncov[0, 0] = c
    st = dict(p=np.zeros(3), q=np.array([1.0, 0, 0, 0]), v=np.zeros(3), ba=np.zeros(3), bg=np.zeros(3), offset=np.zeros((4, 3)), cov=P)
    st["offset"][0, 0] = 0.01
    prop = st["offset"][:, :2].reshape(8).copy()
    z = np.zeros(8)
    z[0] = 3.19                          # pixels -> 0.02 in camera units
    z[1:] = prop[1:] * ekf_oracle.F_PIX
    out = ekf_oracle.update(st, z, ncov, prop, 10.0, True)
    r = 10.0 * c / 159.5 ** 2
    k = p0 / (p0 + r)
    assert abs(out["offset"][0, 0] - (0.01 + k * (3.19 / 159.5 - 0.01))) < 1e-12
    assert abs(out["cov"][15, 15] - p0 * r / (p0 + r)) < 1e-15
