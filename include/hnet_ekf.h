/* hnet_ekf.h — the filter step that CONSUMES the HomographyNet output, dependency free.
 *
 * Restates, in plain C++ (double, no Eigen), what cuahn::UpdaterHNet::update does with the network's 8 corner
 * offsets and their 8x8 covariance (reference cuahn/src/update/UpdaterHNet.cpp:28-61, constants UpdaterHNet.h:57-64),
 * the glue around it in VioManager::feed_measurement (cuahn/src/core/VioManager.cpp:227-275: prior = state offsets
 * x 159.5, iterated EKF loop, offsets reset) and State::reset_4pt_offset (cuahn/src/state/State.cpp:101-111).
 * It exists so that the drop-in claim of INTEGRATION.md can be exercised end to end without ROS / Eigen / OpenCV:
 * tests/cpp/iekf_demo.cpp runs adapter -> this update for a few frames on the GPU box, and tests/test_ekf_cpu.py
 * checks every function against the numpy restatement oracle/ekf_oracle.py.
 *
 * State layout (27 error states, State.h:115-123): p 0..2, q 3..5, v 6..8, ba 9..11, bg 12..14, corner offsets
 * ul 15..17, bl 18..20, br 21..23, ur 24..26 (each corner (x, y, z) in normalised camera coordinates; the network
 * measures (x, y)).  Quaternion: Hamilton, (w, x, y, z).
 * Units: the network speaks pixels of the f = 159.5 virtual camera; the filter divides means by 159.5 and
 * covariances by 159.5^2 = 25440.25 (UpdaterHNet.cpp:31-33).
 */
#ifndef HNET_EKF_H
#define HNET_EKF_H

#include <cmath>
#include <cstring>

namespace hnet_ekf {

constexpr int NS = 27;                 /* error-state dimension */
constexpr double F_PIX = 159.5;        /* (320-1)/2 / tan(45 deg), CamBase.h:166-169 */

struct State {
    double p[3];
    double q[4];                       /* Hamilton (w, x, y, z) */
    double v[3];
    double ba[3];
    double bg[3];
    double offset[4][3];               /* ul, bl, br, ur (State.h:110-113 order) */
    double cov[NS * NS];               /* row major */
};

/* VioManager.cpp:230-234 — the prior handed to network_inference: (x, y) of the four corner offsets, x 159.5 */
inline void prior_pixels(const State& s, double prior_px[8], double prior_cam[8]) {
    for (int c = 0; c < 4; c++)
        for (int k = 0; k < 2; k++) {
            prior_cam[2 * c + k] = s.offset[c][k];
            prior_px[2 * c + k] = s.offset[c][k] * F_PIX;
        }
}

/* in-place inverse of an n x n matrix (n <= 8), Gauss-Jordan with partial pivoting; returns false if singular */
inline bool invert(double* a, int n) {
    double inv[64];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) inv[i * n + j] = i == j ? 1.0 : 0.0;
    for (int col = 0; col < n; col++) {
        int piv = col;
        for (int r = col + 1; r < n; r++)
            if (std::fabs(a[r * n + col]) > std::fabs(a[piv * n + col])) piv = r;
        if (a[piv * n + col] == 0.0) return false;
        if (piv != col)
            for (int j = 0; j < n; j++) {
                double t = a[col * n + j]; a[col * n + j] = a[piv * n + j]; a[piv * n + j] = t;
                t = inv[col * n + j]; inv[col * n + j] = inv[piv * n + j]; inv[piv * n + j] = t;
            }
        const double d = 1.0 / a[col * n + col];
        for (int j = 0; j < n; j++) { a[col * n + j] *= d; inv[col * n + j] *= d; }
        for (int r = 0; r < n; r++) {
            if (r == col) continue;
            const double f = a[r * n + col];
            if (f == 0.0) continue;
            for (int j = 0; j < n; j++) { a[r * n + j] -= f * a[col * n + j]; inv[r * n + j] -= f * inv[col * n + j]; }
        }
    }
    std::memcpy(a, inv, sizeof(double) * n * n);
    return true;
}

/* quat_ops.h:526-538 Ham_quat_update(rot_vec) * q, then quatnorm (quat_ops.h:479-484: sign flip on the LAST component) */
inline void quat_apply_rotvec(const double rv[3], double q[4]) {
    const double ang = std::sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    const double c = std::cos(0.5 * ang);
    /* the reference divides by the angle without a guard (0/0 for a zero update); the limit is used here */
    const double sc = ang > 0.0 ? std::sin(0.5 * ang) / ang : 0.5;
    const double d[3] = {sc * rv[0], sc * rv[1], sc * rv[2]};
    /* matrix: [[c, -d^T], [d, c I + skew(-d)]] */
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    double r[4];
    r[0] = c * w - d[0] * x - d[1] * y - d[2] * z;
    r[1] = d[0] * w + c * x + d[2] * y - d[1] * z;
    r[2] = d[1] * w - d[2] * x + c * y + d[0] * z;
    r[3] = d[2] * w + d[1] * x - d[0] * y + c * z;
    if (r[3] < 0.0) for (int i = 0; i < 4; i++) r[i] = -r[i];
    const double n = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    for (int i = 0; i < 4; i++) q[i] = r[i] / n;
}

/* ---- prior generation: the discrete mean propagation that produces the corner offsets the network
 * receives as prior.  Propagator::predict_and_compute prerequisites (cuahn/src/state/Propagator.cpp:211-220) and
 * Propagator::predict_mean_discrete (:342-364).  Body frame forward-left-up; the ground plane normal in the world is
 * (0, 0, -1) (Propagator.h:101).  The covariance propagation (F, Fw Jacobians, :222-330) follows below (propagate_jacobians). */
struct Extrinsics {
    double c_R_i[9];                   /* camera <- IMU rotation, row major (State.h:108) */
    double i_t_i2c[3];                 /* IMU -> camera translation in the IMU frame (State.h:107) */
};

/* the four image corners in normalised camera coordinates (State.h:110-113), order ul, bl, br, ur */
inline const double* corner_xy1(int c) {
    static const double k[4][3] = {{-1.0, -0.69906, 1.0}, {-1.0, 0.69906, 1.0}, {1.0, 0.69906, 1.0}, {1.0, -0.69906, 1.0}};
    return k[c];
}

/* quat_ops.h:549-553 Ham_quat_2_Rot: local -> global rotation of a Hamilton quaternion (w, x, y, z) */
inline void quat_to_rot(const double q[4], double R[9]) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double s = w * w - (x * x + y * y + z * z);
    const double v[3] = {x, y, z};
    const double sk[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = (i == j ? s : 0.0) + 2.0 * v[i] * v[j] + 2.0 * w * sk[i * 3 + j];
}

inline void mat3_vec(const double* M, const double* v, double* o) {
    for (int i = 0; i < 3; i++) o[i] = M[i * 3] * v[0] + M[i * 3 + 1] * v[1] + M[i * 3 + 2] * v[2];
}
inline void cross(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

inline void propagate_mean(State& s, const Extrinsics& e, double dt, const double w_hat[3], const double a_hat[3], double gravity_mag = 9.81) {
    double R[9], Rt[9];
    quat_to_rot(s.q, R);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rt[i * 3 + j] = R[j * 3 + i];
    /* Propagator.cpp:212-215 */
    double wc[3], vc[3], muc[3], tmp[3], tmp2[3];
    mat3_vec(e.c_R_i, w_hat, wc);
    cross(w_hat, e.i_t_i2c, tmp);
    for (int i = 0; i < 3; i++) tmp[i] += s.v[i];
    mat3_vec(e.c_R_i, tmp, vc);
    const double muw[3] = {0.0, 0.0, -1.0};
    mat3_vec(Rt, muw, tmp);
    mat3_vec(e.c_R_i, tmp, muc);
    for (int i = 0; i < 3; i++) tmp[i] = s.p[i] + e.i_t_i2c[i];
    mat3_vec(R, tmp, tmp2);
    const double dc = tmp2[2];
    /* corner dynamics use the state BEFORE the IMU part is advanced (:217-220, :357-362) */
    double Hm[9];
    const double skw[9] = {0, -wc[2], wc[1], wc[2], 0, -wc[0], -wc[1], wc[0], 0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Hm[i * 3 + j] = skw[i * 3 + j] + vc[i] * muc[j] / dc;
    double new_off[4][3];
    for (int c = 0; c < 4; c++) {
        double pt[3], Hp[3];
        for (int i = 0; i < 3; i++) pt[i] = corner_xy1(c)[i] + s.offset[c][i];
        mat3_vec(Hm, pt, Hp);
        /* -(I - pt ez^T) H pt = -(Hp - pt * Hp_z) */
        for (int i = 0; i < 3; i++) new_off[c][i] = s.offset[c][i] + dt * (-(Hp[i] - pt[i] * Hp[2]));
    }
    /* :347-354 (the position / velocity are expressed in the body frame in this filter) */
    double wdt[3] = {w_hat[0] * dt, w_hat[1] * dt, w_hat[2] * dt};
    double nq[4] = {s.q[0], s.q[1], s.q[2], s.q[3]};
    quat_apply_rotvec(wdt, nq);
    double wxv[3], wxp[3], g[3];
    cross(w_hat, s.v, wxv);
    cross(w_hat, s.p, wxp);
    const double grav[3] = {0.0, 0.0, -gravity_mag};
    mat3_vec(Rt, grav, g);
    double nv[3], np[3];
    for (int i = 0; i < 3; i++) {
        nv[i] = s.v[i] + dt * (-wxv[i] + a_hat[i] + g[i]);
        np[i] = s.p[i] + dt * (-wxp[i] + s.v[i]);
    }
    for (int i = 0; i < 3; i++) { s.p[i] = np[i]; s.v[i] = nv[i]; }
    for (int i = 0; i < 4; i++) s.q[i] = nq[i];
    std::memcpy(s.offset, new_off, sizeof new_off);
}

/* ---- covariance propagation: Propagator::predict_and_compute's Jacobians (Propagator.cpp:222-333),
 * the noise matrix of the Propagator constructor (Propagator.h:86-96) and StateHelper::propagate_Cov (StateHelper.cpp:28-32).
 * Error-state order p q v ba bg ul bl br ur (3 each); the attitude error is a rotation vector applied on the right
 * (q <- q (x) dq, what Ham_quat_update(dtheta) * q computes); noise order (gyro, accel, accel walk, gyro walk, 4pt) x 3.
 * tests/test_ekf_cpu.py checks every block of F against central differences of propagate_mean() (independent of the formulas)
 * and the whole thing against the numpy restatement. */
constexpr int NW = 15;                 /* noise dimension */

namespace m3 {
inline void skew(const double* w, double* S) { S[0] = 0; S[1] = -w[2]; S[2] = w[1]; S[3] = w[2]; S[4] = 0; S[5] = -w[0]; S[6] = -w[1]; S[7] = w[0]; S[8] = 0; }
inline void mul(const double* A, const double* B, double* C) {      /* C = A B (3x3) */
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    std::memcpy(C, t, sizeof t);
}
inline void transpose(const double* A, double* T) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[i * 3 + j] = A[j * 3 + i];
    std::memcpy(T, t, sizeof t);
}
inline void outer(const double* a, const double* b, double* C) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[i * 3 + j] = a[i] * b[j];
}
inline double dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
}  // namespace m3

/* quat_ops.h:573-580 (the reference divides by |theta| without a guard; the limit I is used at 0) */
inline void jr_theta(const double th[3], double J[9]) {
    const double n = std::sqrt(m3::dot(th, th));
    for (int i = 0; i < 9; i++) J[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (n < 1e-12) return;
    double S[9], SS[9];
    m3::skew(th, S);
    m3::mul(S, S, SS);
    const double a = (1.0 - std::cos(n)) / (n * n), b = (n - std::sin(n)) / (n * n * n);
    for (int i = 0; i < 9; i++) J[i] += -a * S[i] + b * SS[i];
}

inline void set_block(double* M, int ld, int r0, int c0, const double* B, double scale = 1.0) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) M[(r0 + i) * ld + c0 + j] = scale * B[i * 3 + j];
}

/* F [27 x 27], Fw [27 x 15], row major, evaluated at `s` BEFORE the mean is advanced (Propagator.cpp:211-220, :222-333) */
inline void propagate_jacobians(const State& s, const Extrinsics& e, double dt, const double w_hat[3], double* F, double* Fw,
                                double gravity_mag = 9.81) {
    std::memset(F, 0, sizeof(double) * NS * NS);
    std::memset(Fw, 0, sizeof(double) * NS * NW);
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double R[9], Rt[9];
    quat_to_rot(s.q, R);
    m3::transpose(R, Rt);
    const double grav[3] = {0.0, 0.0, -gravity_mag}, muw[3] = {0.0, 0.0, -1.0};
    double wc[3], vc[3], muc[3], t3[3], t3b[3];
    mat3_vec(e.c_R_i, w_hat, wc);                                           /* :212 */
    cross(w_hat, e.i_t_i2c, t3);
    for (int i = 0; i < 3; i++) t3[i] += s.v[i];
    mat3_vec(e.c_R_i, t3, vc);                                              /* :213 */
    mat3_vec(Rt, muw, t3);
    mat3_vec(e.c_R_i, t3, muc);                                             /* :214 */
    double ppt[3];
    for (int i = 0; i < 3; i++) ppt[i] = s.p[i] + e.i_t_i2c[i];
    mat3_vec(R, ppt, t3b);
    const double dc = t3b[2];                                               /* :215 */
    const int P_ = 0, Q_ = 3, V_ = 6, BA = 9, BG = 12;
    double Sw[9], Sp[9], Sv[9], B[9];
    m3::skew(w_hat, Sw); m3::skew(s.p, Sp); m3::skew(s.v, Sv);
    for (int i = 0; i < 9; i++) B[i] = I3[i] - dt * Sw[i];
    set_block(F, NS, P_, P_, B);                                            /* :224 */
    set_block(F, NS, P_, V_, I3, dt);
    set_block(F, NS, P_, BG, Sp, -dt);
    {                                                                       /* :228 rotation of the quaternion of (w_hat dt), transposed */
        const double rv[3] = {w_hat[0] * dt, w_hat[1] * dt, w_hat[2] * dt};
        const double n = std::sqrt(m3::dot(rv, rv));
        double qd[4] = {1.0, 0.0, 0.0, 0.0};
        if (n > 1e-300) { qd[0] = std::cos(0.5 * n); for (int i = 0; i < 3; i++) qd[1 + i] = std::sin(0.5 * n) * rv[i] / n; }
        double Rd[9], Rdt[9], Jr[9];
        quat_to_rot(qd, Rd);
        m3::transpose(Rd, Rdt);
        set_block(F, NS, Q_, Q_, Rdt);
        jr_theta(rv, Jr);
        set_block(F, NS, Q_, BG, Jr, -dt);                                   /* :229 */
    }
    mat3_vec(Rt, grav, t3);
    m3::skew(t3, B);
    set_block(F, NS, V_, Q_, B, dt);                                        /* :231 */
    for (int i = 0; i < 9; i++) B[i] = I3[i] - dt * Sw[i];
    set_block(F, NS, V_, V_, B);
    set_block(F, NS, V_, BA, I3, -dt);
    set_block(F, NS, V_, BG, Sv, -dt);
    set_block(F, NS, BA, BA, I3);                                           /* :236-237 */
    set_block(F, NS, BG, BG, I3);
    /* 4-point offsets (:239-319) */
    const double scalar = vc[2] / dc;                                       /* :240-241 */
    double Swc[9];
    m3::skew(wc, Swc);
    double J_dc_p[3] = {R[6], R[7], R[8]};                                  /* ez^T R (:293) */
    double Sppt[9], RS[9], J_dc_q[3];
    m3::skew(ppt, Sppt);
    m3::mul(R, Sppt, RS);
    for (int j = 0; j < 3; j++) J_dc_q[j] = -RS[6 + j];                     /* ez^T (-R skew(p + t)) (:294) */
    double Smu[9], J_muc_q[9];
    mat3_vec(Rt, muw, t3);
    m3::skew(t3, Smu);
    m3::mul(e.c_R_i, Smu, J_muc_q);                                         /* :295 */
    double St[9], J_vc_bw[9];
    m3::skew(e.i_t_i2c, St);
    m3::mul(e.c_R_i, St, J_vc_bw);                                          /* Propagator.h:193 */
    for (int c = 0; c < 4; c++) {
        double pt[3];
        for (int i = 0; i < 3; i++) pt[i] = corner_xy1(c)[i] + s.offset[c][i];        /* :217-220 */
        const double mupt = m3::dot(muc, pt);
        double ezSw[3] = {Swc[6], Swc[7], Swc[8]};                          /* ez^T skew(wc) */
        const double ezSwpt = m3::dot(ezSw, pt);
        double J_df_pt[9], vm[9], pte[9], ptm[9];
        m3::outer(vc, muc, vm);
        m3::outer(pt, ezSw, pte);
        m3::outer(pt, muc, ptm);
        for (int i = 0; i < 9; i++)                                         /* :244-247 */
            J_df_pt[i] = Swc[i] + vm[i] / dc - ezSwpt * I3[i] - pte[i] - scalar * (mupt * I3[i] + ptm[i]);
        double common[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) common[i * 3 + j] = I3[i * 3 + j] - (j == 2 ? pt[i] : 0.0);    /* I - pt ez^T (:248) */
        double cv[3];
        mat3_vec(common, vc, cv);
        double J_df_dc[3];
        for (int i = 0; i < 3; i++) J_df_dc[i] = -mupt * cv[i] / (dc * dc);                           /* :249 */
        double J_df_vc[9], J_df_muc[9], J_df_wc[9], Spt[9];
        for (int i = 0; i < 9; i++) J_df_vc[i] = mupt * common[i] / dc;                                /* :250 */
        m3::outer(cv, pt, J_df_muc);
        for (int i = 0; i < 9; i++) J_df_muc[i] /= dc;                                                 /* :251 */
        m3::skew(pt, Spt);
        m3::mul(common, Spt, J_df_wc);
        for (int i = 0; i < 9; i++) J_df_wc[i] = -J_df_wc[i];                                          /* :252 */
        const int o = 15 + 3 * c;
        double blk[9], t9[9], t9b[9];
        m3::outer(J_df_dc, J_dc_p, blk);
        set_block(F, NS, o, P_, blk, -dt);                                                             /* :298 */
        m3::outer(J_df_dc, J_dc_q, blk);
        m3::mul(J_df_muc, J_muc_q, t9);
        for (int i = 0; i < 9; i++) blk[i] += t9[i];
        set_block(F, NS, o, Q_, blk, -dt);                                                             /* :299 */
        m3::mul(J_df_vc, e.c_R_i, blk);
        set_block(F, NS, o, V_, blk, -dt);                                                             /* :300 */
        m3::mul(J_df_vc, J_vc_bw, t9);
        m3::mul(J_df_wc, e.c_R_i, t9b);                                                                /* J_wc_bw = -c_R_i */
        for (int i = 0; i < 9; i++) blk[i] = t9[i] - t9b[i];
        set_block(F, NS, o, BG, blk, -dt);                                                             /* :301 */
        for (int i = 0; i < 9; i++) blk[i] = I3[i] - dt * J_df_pt[i];
        set_block(F, NS, o, o, blk);                                                                   /* :302 */
    }
    /* noise Jacobian (:322-333) */
    auto copy_block = [&](int r0, int cw, int fr, int fc, double sc) {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Fw[(r0 + i) * NW + cw + j] = sc * F[(fr + i) * NS + fc + j];
    };
    copy_block(P_, 0, P_, BG, -1.0);
    copy_block(P_, 12, P_, V_, 1.0);
    copy_block(Q_, 0, Q_, BG, -1.0);
    copy_block(V_, 0, V_, BG, -1.0);
    copy_block(V_, 3, P_, V_, 1.0);
    copy_block(BA, 6, P_, V_, 1.0);
    copy_block(BG, 9, P_, V_, 1.0);
    for (int c = 0; c < 4; c++) copy_block(15 + 3 * c, 0, 15 + 3 * c, BG, -1.0);
}

/* Propagator.h:86-96: diagonal of Q, order gyro, accel, accel random walk, gyro random walk, 4pt */
inline void noise_q_diag(double sigma_w, double sigma_a, double sigma_wb, double sigma_ab, double q[NW]) {
    const double v[5] = {sigma_w * sigma_w, sigma_a * sigma_a, sigma_ab * sigma_ab, sigma_wb * sigma_wb, 1.0e-4};
    for (int i = 0; i < NW; i++) q[i] = v[i / 3];
}

/* StateHelper::propagate_Cov (StateHelper.cpp:28-32): P <- F P F^T + Fw diag(q) Fw^T */
inline void propagate_cov(double* P, const double* F, const double* Fw, const double q[NW]) {
    static thread_local double T[NS * NS], O[NS * NS];
    for (int i = 0; i < NS; i++)
        for (int j = 0; j < NS; j++) {
            double a = 0.0;
            for (int k = 0; k < NS; k++) a += F[i * NS + k] * P[k * NS + j];
            T[i * NS + j] = a;
        }
    for (int i = 0; i < NS; i++)
        for (int j = 0; j < NS; j++) {
            double a = 0.0;
            for (int k = 0; k < NS; k++) a += T[i * NS + k] * F[j * NS + k];
            for (int k = 0; k < NW; k++) a += Fw[i * NW + k] * q[k] * Fw[j * NW + k];
            O[i * NS + j] = a;
        }
    std::memcpy(P, O, sizeof(double) * NS * NS);
}

/* one IMU interval of Propagator::propagate_with_imu's loop (:63-67): Jacobians at the old state, mean, covariance */
inline void propagate(State& s, const Extrinsics& e, double dt, const double w_hat[3], const double a_hat[3], const double q[NW],
                      double gravity_mag = 9.81) {
    static thread_local double F[NS * NS], Fw[NS * NW];
    propagate_jacobians(s, e, dt, w_hat, F, Fw, gravity_mag);
    propagate_mean(s, e, dt, w_hat, a_hat, gravity_mag);
    propagate_cov(s.cov, F, Fw, q);
}

/* UpdaterHNet::update (UpdaterHNet.cpp:28-61).  net_mean_px[8], net_cov_px[64]: what get_pred_mean()/get_pred_Cov()
 * return; propagated[8]: the prior in camera units (prior_pixels() / 159.5); k_net_cov: UpdaterOptions.h:33 (10.0).
 * Returns false if the innovation covariance is singular (the reference would produce inf/nan). */
inline bool update(State& s, const double net_mean_px[8], const double net_cov_px[64], const double propagated[8], double k_net_cov,
                   bool update_offset) {
    /* H selects (x, y) of each corner: rows 2c+k <- state 15 + 3c + k;  Hn = I8 */
    int sel[8];
    for (int c = 0; c < 4; c++) { sel[2 * c] = 15 + 3 * c; sel[2 * c + 1] = 16 + 3 * c; }
    double S[64], PHt[NS * 8];
    for (int i = 0; i < NS; i++)
        for (int j = 0; j < 8; j++) PHt[i * 8 + j] = s.cov[i * NS + sel[j]];
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) S[i * 8 + j] = s.cov[sel[i] * NS + sel[j]] + k_net_cov * net_cov_px[i * 8 + j] / (F_PIX * F_PIX);
    if (!invert(S, 8)) return false;
    double K[NS * 8];
    for (int i = 0; i < NS; i++)
        for (int j = 0; j < 8; j++) {
            double a = 0.0;
            for (int k = 0; k < 8; k++) a += PHt[i * 8 + k] * S[k * 8 + j];
            K[i * 8 + j] = a;
        }
    double inno[8];
    for (int i = 0; i < 8; i++) inno[i] = net_mean_px[i] / F_PIX - propagated[i];
    /* Cov <- (I - K H) Cov  (not the Joseph form, as the reference) */
    double KH_P[NS * NS];
    for (int i = 0; i < NS; i++)
        for (int j = 0; j < NS; j++) {
            double a = 0.0;
            for (int k = 0; k < 8; k++) a += K[i * 8 + k] * s.cov[sel[k] * NS + j];
            KH_P[i * NS + j] = a;
        }
    for (int i = 0; i < NS * NS; i++) s.cov[i] -= KH_P[i];
    double dx[NS];
    const int rows = update_offset ? NS : 15;          /* last IEKF iteration: the offsets are about to be reset anyway */
    for (int i = 0; i < NS; i++) dx[i] = 0.0;
    for (int i = 0; i < rows; i++)
        for (int k = 0; k < 8; k++) dx[i] += K[i * 8 + k] * inno[k];
    for (int i = 0; i < 3; i++) s.p[i] += dx[i];
    quat_apply_rotvec(dx + 3, s.q);
    for (int i = 0; i < 3; i++) { s.v[i] += dx[6 + i]; s.ba[i] += dx[9 + i]; s.bg[i] += dx[12 + i]; }
    if (update_offset)
        for (int c = 0; c < 4; c++)
            for (int k = 0; k < 3; k++) s.offset[c][k] += dx[15 + 3 * c + k];
    return true;
}

/* State::reset_4pt_offset (State.cpp:101-111): offsets to zero, covariance keeps only the IMU 15 x 15 block */
inline void reset_4pt_offset(State& s) {
    std::memset(s.offset, 0, sizeof s.offset);
    for (int i = 0; i < NS; i++)
        for (int j = 0; j < NS; j++)
            if (i >= 15 || j >= 15) s.cov[i * NS + j] = 0.0;
}

/* The iterated update of VioManager.cpp:227-275 around any object with the reference's HomographyNet surface
 * (network_inference / get_pred_mean / get_pred_Cov returning something indexable as (i) resp. (i, j),
 * get_latest_inference_time(), public img_counter).  As in the reference the network runs in EVERY iteration, but the filter is
 * only updated when the network's latest image is the frame being processed and more than 10 images have been seen
 * (VioManager.cpp:257: `HNet->get_latest_inference_time() == time_stamp && HNet->img_counter > 10`); the offsets are reset
 * afterwards either way (:275).  Returns the number of updates applied. */
template <class Net, class Vec8>
inline int iterated_update(State& s, Net& net, int max_iekf_iteration, double k_net_cov, Vec8& prior_px_vec, double time_stamp) {
    int done = 0;
    for (int it = 0; it < max_iekf_iteration; it++) {
        double prior_px[8], prior_cam[8];
        prior_pixels(s, prior_px, prior_cam);
        for (int i = 0; i < 8; i++) prior_px_vec[i] = prior_px[i];
        net.network_inference(prior_px_vec, it);
        if (net.get_latest_inference_time() == time_stamp && net.img_counter > 10) {
            const auto m = net.get_pred_mean();
            const auto C = net.get_pred_Cov();
            double mean[8], cov[64];
            for (int i = 0; i < 8; i++) {
                mean[i] = m(i, 0);
                for (int j = 0; j < 8; j++) cov[i * 8 + j] = C(i, j);
            }
            if (!update(s, mean, cov, prior_cam, k_net_cov, it != max_iekf_iteration - 1)) break;
            done++;
        }
    }
    reset_4pt_offset(s);
    return done;
}

}  // namespace hnet_ekf
#endif  /* HNET_EKF_H */
