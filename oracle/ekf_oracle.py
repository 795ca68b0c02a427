"""numpy (float64) restatement of the EKF update that consumes the network output (TEST INFRASTRUCTURE).

Follows cuahn/src/update/UpdaterHNet.cpp:28-61 with H / Hn of UpdaterHNet.h:57-64, quaternion helpers of
ov_core/src/utils/quat_ops.h:479-484 (quatnorm) and :526-538 (Ham_quat_update), State::reset_4pt_offset
(cuahn/src/state/State.cpp:101-111).  The reference is C++ on Eigen (not in this image) and holds no test vectors for
this step: parity of include/hnet_ekf.h is pinned against this restatement only ("parity unpinned" by the reference).
"""
import numpy as np

F_PIX = 159.5
H = np.zeros((8, 27))
for c in range(4):
    H[2 * c, 15 + 3 * c] = 1.0
    H[2 * c + 1, 16 + 3 * c] = 1.0
HN = np.eye(8)


def skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], float)


def ham_quat_update(rv):
    ang = np.linalg.norm(rv)
    d = np.sin(0.5 * ang) * rv / ang if ang > 0 else 0.5 * rv
    m = np.eye(4) * np.cos(0.5 * ang)
    m[1:, 1:] += skew(-d)
    m[0, 1:] = -d
    m[1:, 0] = d
    return m


def quatnorm(q):
    q = q.copy()
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def update(state, net_mean_px, net_cov_px, propagated, k_net_cov=10.0, update_offset=True):
    """state: dict(p, q, v, ba, bg, offset[4,3], cov[27,27]); returns the updated copy"""
    P = state["cov"]
    S = H @ P @ H.T + HN @ (k_net_cov * net_cov_px / (F_PIX * F_PIX)) @ HN.T
    K = P @ H.T @ np.linalg.inv(S)
    inno = net_mean_px / F_PIX - propagated
    out = {k: np.array(v, float).copy() for k, v in state.items()}
    out["cov"] = (np.eye(27) - K @ H) @ P
    dx = np.zeros(27)
    if update_offset:
        dx = K @ inno
    else:
        dx[:15] = K[:15] @ inno
    out["p"] = state["p"] + dx[0:3]
    out["q"] = quatnorm(ham_quat_update(dx[3:6]) @ state["q"])
    out["v"] = state["v"] + dx[6:9]
    out["ba"] = state["ba"] + dx[9:12]
    out["bg"] = state["bg"] + dx[12:15]
    if update_offset:
        out["offset"] = state["offset"] + dx[15:27].reshape(4, 3)
    return out


def reset_4pt_offset(state):
    out = {k: np.array(v, float).copy() for k, v in state.items()}
    out["offset"][:] = 0.0
    c = np.zeros((27, 27))
    c[:15, :15] = state["cov"][:15, :15]
    out["cov"] = c
    return out


# ---- prior generation: Propagator.cpp:211-220 and predict_mean_discrete :342-364 (mean only)
CORNERS = np.array([[-1.0, -0.69906, 1.0], [-1.0, 0.69906, 1.0], [1.0, 0.69906, 1.0], [1.0, -0.69906, 1.0]])


def ham_quat_2_rot(q):
    v = q[1:]
    return np.eye(3) * (q[0] ** 2 - v @ v) + 2 * np.outer(v, v) + 2 * q[0] * skew(v)


def propagate_mean(state, c_R_i, i_t_i2c, dt, w_hat, a_hat, gravity_mag=9.81):
    R = ham_quat_2_rot(state["q"])
    wc = c_R_i @ w_hat
    vc = c_R_i @ (state["v"] + skew(w_hat) @ i_t_i2c)
    muc = c_R_i @ R.T @ np.array([0.0, 0.0, -1.0])
    dc = (R @ (state["p"] + i_t_i2c))[2]
    Hm = skew(wc) + np.outer(vc, muc) / dc
    ez = np.array([[0.0, 0.0, 1.0]])
    out = {k: np.array(v, float).copy() for k, v in state.items()}
    for c in range(4):
        pt = CORNERS[c] + state["offset"][c]
        out["offset"][c] = state["offset"][c] + dt * (-(np.eye(3) - np.outer(pt, ez)) @ Hm @ pt)
    out["q"] = quatnorm(ham_quat_update(w_hat * dt) @ state["q"])
    out["v"] = state["v"] + dt * (-skew(w_hat) @ state["v"] + a_hat + R.T @ np.array([0.0, 0.0, -gravity_mag]))
    out["p"] = state["p"] + dt * (-skew(w_hat) @ state["p"] + state["v"])
    return out


# ---- covariance propagation: the state-transition and noise Jacobians of Propagator::predict_and_compute
# (cuahn/src/state/Propagator.cpp:222-330), StateHelper::propagate_Cov (StateHelper.cpp:28-32), the noise matrix of the
# Propagator constructor (Propagator.h:86-96).  Error-state order p q v ba bg ul bl br ur (3 each), noise order
# (gyro, accel, accel walk, gyro walk, 4pt) x 3.
def jr_theta(th):
    """quat_ops.h:573-580 (no guard at |th| = 0 in the reference; the limit I is used here)"""
    n = np.linalg.norm(th)
    if n < 1e-12:
        return np.eye(3)
    S = skew(th)
    return np.eye(3) - (1 - np.cos(n)) / n ** 2 * S + (n - np.sin(n)) / n ** 3 * S @ S


def rotvec_2_ham_quat(rv):
    n = np.linalg.norm(rv)
    if n < 1e-300:
        return np.array([1.0, 0.0, 0.0, 0.0])
    return np.concatenate([[np.cos(0.5 * n)], np.sin(0.5 * n) * rv / n])


def jacobians(state, c_R_i, i_t_i2c, dt, w_hat, gravity_mag=9.81):
    """(F [27,27], Fw [27,15]) evaluated at the state BEFORE the mean is advanced (Propagator.cpp:211-220 then :222-330)"""
    I3 = np.eye(3)
    ez = np.array([[0.0, 0.0, 1.0]])
    R = ham_quat_2_rot(state["q"])
    p, v = state["p"], state["v"]
    grav = np.array([0.0, 0.0, -gravity_mag])
    muw = np.array([0.0, 0.0, -1.0])
    wc = c_R_i @ w_hat                                           # :212
    vc = c_R_i @ (v + skew(w_hat) @ i_t_i2c)                     # :213
    muc = c_R_i @ R.T @ muw                                      # :214
    dc = (R @ (p + i_t_i2c))[2]                                  # :215
    F = np.zeros((27, 27))
    Fw = np.zeros((27, 15))
    P_, Q_, V_, BA, BG = 0, 3, 6, 9, 12
    F[P_:P_ + 3, P_:P_ + 3] = I3 - dt * skew(w_hat)             # :224
    F[P_:P_ + 3, V_:V_ + 3] = dt * I3
    F[P_:P_ + 3, BG:BG + 3] = -dt * skew(p)
    F[Q_:Q_ + 3, Q_:Q_ + 3] = ham_quat_2_rot(rotvec_2_ham_quat(w_hat * dt)).T    # :228
    F[Q_:Q_ + 3, BG:BG + 3] = -dt * jr_theta(w_hat * dt)
    F[V_:V_ + 3, Q_:Q_ + 3] = dt * skew(R.T @ grav)             # :231
    F[V_:V_ + 3, V_:V_ + 3] = I3 - dt * skew(w_hat)
    F[V_:V_ + 3, BA:BA + 3] = -dt * I3
    F[V_:V_ + 3, BG:BG + 3] = -dt * skew(v)
    F[BA:BA + 3, BA:BA + 3] = I3                                # :236-237
    F[BG:BG + 3, BG:BG + 3] = I3
    scalar = (ez @ vc).item() / dc                                # :240-241
    J_f_df = -dt * I3                                           # :292
    J_dc_p = ez @ R                                             # :293
    J_dc_q = ez @ (-R @ skew(p + i_t_i2c))                      # :294
    J_muc_q = c_R_i @ skew(R.T @ muw)                           # :295
    J_vc_v, J_vc_bw, J_wc_bw = c_R_i, c_R_i @ skew(i_t_i2c), -c_R_i      # Propagator.h:192-194
    for c in range(4):
        pt = (CORNERS[c] + state["offset"][c]).reshape(3, 1)    # :217-220
        mu = muc.reshape(3, 1)
        vcv = vc.reshape(3, 1)
        J_df_pt = (skew(wc) + vcv @ mu.T / dc - (ez @ skew(wc) @ pt).item() * I3 - pt @ ez @ skew(wc)
                   - scalar * ((mu.T @ pt).item() * I3 + pt @ mu.T))          # :244-247
        common = I3 - pt @ ez                                   # :248
        J_df_dc = 1.0 / dc / dc * (mu.T @ pt).item() * (-common) @ vcv         # :249  [3,1]
        J_df_vc = 1.0 / dc * (mu.T @ pt).item() * common          # :250
        J_df_muc = 1.0 / dc * common @ vcv @ pt.T               # :251
        J_df_wc = -common @ skew(pt.reshape(3))                 # :252
        o = 15 + 3 * c
        F[o:o + 3, P_:P_ + 3] = J_f_df @ J_df_dc @ J_dc_p       # :298-302
        F[o:o + 3, Q_:Q_ + 3] = J_f_df @ (J_df_dc @ J_dc_q + J_df_muc @ J_muc_q)
        F[o:o + 3, V_:V_ + 3] = J_f_df @ J_df_vc @ J_vc_v
        F[o:o + 3, BG:BG + 3] = J_f_df @ (J_df_vc @ J_vc_bw + J_df_wc @ J_wc_bw)
        F[o:o + 3, o:o + 3] = I3 + J_f_df @ J_df_pt
    Fw[P_:P_ + 3, 0:3] = -F[P_:P_ + 3, BG:BG + 3]               # :322-328
    Fw[P_:P_ + 3, 12:15] = F[P_:P_ + 3, V_:V_ + 3]
    Fw[Q_:Q_ + 3, 0:3] = -F[Q_:Q_ + 3, BG:BG + 3]
    Fw[V_:V_ + 3, 0:3] = -F[V_:V_ + 3, BG:BG + 3]
    Fw[V_:V_ + 3, 3:6] = Fw[P_:P_ + 3, 12:15]
    Fw[BA:BA + 3, 6:9] = Fw[P_:P_ + 3, 12:15]
    Fw[BG:BG + 3, 9:12] = Fw[P_:P_ + 3, 12:15]
    for c in range(4):                                          # :330-333
        o = 15 + 3 * c
        Fw[o:o + 3, 0:3] = -F[o:o + 3, BG:BG + 3]
    return F, Fw


def noise_q(sigma_w, sigma_a, sigma_wb, sigma_ab):
    """Propagator.h:86-96 — order gyro, accel, accel random walk, gyro random walk, 4pt (1e-4)"""
    return np.diag(np.repeat([sigma_w ** 2, sigma_a ** 2, sigma_ab ** 2, sigma_wb ** 2, 1.0e-4], 3))


def propagate_cov(P, F, Fw, Q):
    """StateHelper.cpp:28-32"""
    return F @ P @ F.T + Fw @ Q @ Fw.T
