// geom.h — 3x3 homography algebra shared by host and device code (double precision inside).
//
// The reference does this part in fp32 through torch.inverse / bmm (model_to_trace.py:42-61, :18-38);
// at 320x224 the 8x8 DLT system has entries up to 319*319 ~ 1e5, and the reference's own fp32 result
// sits up to ~1e-4 px from exact.  Geometry here is evaluated in double and rounded
// to fp32 only where the reference stores a tensor.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace hnet {

#define HNET_HD __host__ __device__ inline

constexpr int IMG_H = 224, IMG_W = 320, NPIX = IMG_H * IMG_W;

// image corners ul, bl, br, ur as (u, v) — model_to_trace.py:79-83
HNET_HD double p4(int i) {
    // {0,0, 0,223, 319,223, 319,0}
    const int c = i >> 1, isv = i & 1;
    if (isv) return (c == 1 || c == 2) ? (double)(IMG_H - 1) : 0.0;
    return (c >= 2) ? (double)(IMG_W - 1) : 0.0;
}

// DLT_solve (model_to_trace.py:42-61) for src = image corners: rows
//   [x y 1 0 0 0 -u'x -u'y | u'] and [0 0 0 x y 1 -v'x -v'y | v'];  solved by Gaussian elimination with
// partial pivoting instead of an explicit inverse; H = [h8, 1].
HNET_HD void dlt_solve(const double dst[8], double H[9]) {
    double A[8][9];
    for (int i = 0; i < 4; i++) {
        const double x = p4(2 * i), y = p4(2 * i + 1), u = dst[2 * i], v = dst[2 * i + 1];
        double* r0 = A[2 * i];
        double* r1 = A[2 * i + 1];
        r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -u * x; r0[7] = -u * y; r0[8] = u;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -v * x; r1[7] = -v * y; r1[8] = v;
    }
    for (int c = 0; c < 8; c++) {
        int piv = c;
        double best = fabs(A[c][c]);
        for (int r = c + 1; r < 8; r++) {
            double a = fabs(A[r][c]);
            if (a > best) { best = a; piv = r; }
        }
        if (piv != c)
            for (int j = c; j < 9; j++) { double t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
        const double inv = 1.0 / A[c][c];
        for (int r = c + 1; r < 8; r++) {
            const double f = A[r][c] * inv;
            if (f != 0.0)
                for (int j = c; j < 9; j++) A[r][j] -= f * A[c][j];
        }
    }
    for (int i = 7; i >= 0; i--) {
        double s = A[i][8];
        for (int j = i + 1; j < 8; j++) s -= A[i][j] * H[j];
        H[i] = s / A[i][i];
    }
    H[8] = 1.0;
}

// C = A * B  (torch.bmm, model_to_trace.py:168,188,323)
HNET_HD void mat3_mul(const double a[9], const double b[9], double c[9]) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
    for (int i = 0; i < 9; i++) c[i] = t[i];
}

// ensemble statistics + transfer + output assembly for one pair
//   (model_to_trace.py:274-281 ensemble, :18-38 transfer_mean_var_single, :311-317 assembly, :321-323 H_total)
// mean_s / logvar_s: [n][8] per-sample head outputs (logvar already x1e-3).
HNET_HD void finish_pair(const float* mean_s, const float* logvar_s, int n, const float* H1f,
                         float* mean8, float* cov64, float* Htot9) {
    double H1[9];
    for (int i = 0; i < 9; i++) H1[i] = (double)H1f[i];
    double pbar[8], ens[8];
    for (int i = 0; i < 8; i++) {
        double sm = 0, sv = 0;
        for (int s = 0; s < n; s++) {
            sm += (double)mean_s[s * 8 + i];
            sv += exp((double)logvar_s[s * 8 + i]);
        }
        const float mb = (float)(sm / n), vb = (float)(sv / n);     // tensors are fp32 in the reference
        double se = 0;
        for (int s = 0; s < n; s++) { const double d = (double)mb - (double)mean_s[s * 8 + i]; se += d * d; }
        ens[i] = (double)(float)((double)(float)(se / n) + (double)vb);
        pbar[i] = (double)(float)(p4(i) + (double)mb);
    }
    for (int i = 0; i < 64; i++) cov64[i] = 0.0f;
    for (int c = 0; c < 4; c++) {
        const double pu = pbar[2 * c], pv = pbar[2 * c + 1];
        const double X = H1[0] * pu + H1[1] * pv + H1[2];
        const double Y = H1[3] * pu + H1[4] * pv + H1[5];
        const double S = H1[6] * pu + H1[7] * pv + H1[8];
        mean8[2 * c] = (float)(X / S - p4(2 * c));
        mean8[2 * c + 1] = (float)(Y / S - p4(2 * c + 1));
        const double g00 = H1[0] / S, g01 = H1[1] / S, g10 = H1[3] / S, g11 = H1[4] / S;
        const double vu = ens[2 * c], vv = ens[2 * c + 1];
        cov64[(2 * c) * 8 + 2 * c] = (float)(g00 * vu * g00 + g01 * vv * g01);
        cov64[(2 * c) * 8 + 2 * c + 1] = (float)(g00 * vu * g10 + g01 * vv * g11);
        cov64[(2 * c + 1) * 8 + 2 * c] = (float)(g10 * vu * g00 + g11 * vv * g01);
        cov64[(2 * c + 1) * 8 + 2 * c + 1] = (float)(g10 * vu * g10 + g11 * vv * g11);
    }
    if (Htot9) {
        double Hb[9], Ht[9];
        dlt_solve(pbar, Hb);
        mat3_mul(H1, Hb, Ht);
        for (int i = 0; i < 9; i++) Htot9[i] = (float)Ht[i];
    }
}

}  // namespace hnet
