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
