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

// DLT_solve (model_to_trace.py:42-61) for src = image corners.  The reference builds the 8x8 system
//   [x y 1 0 0 0 -u'x -u'y | u'], [0 0 0 x y 1 -v'x -v'y | v']  and multiplies inverse(A) by b.
// The 4-point homography with h33 = 1 is unique, so it is evaluated here in closed form (Heckbert's
// square-to-quadrilateral mapping composed with the scaling of the source rectangle), in double: ~40 flops, no
// pivoting, no local arrays (the elimination loop's dynamically indexed scratch cost ~10 us on one lane).
//   unit square (s,t): (0,0)->ul', (1,0)->ur', (1,1)->br', (0,1)->bl';  s = u/(W-1), t = v/(H-1)
HNET_HD void dlt_solve(const double dst[8], double H[9]) {
    const double x0 = dst[0], y0 = dst[1];       // ul'
    const double x3 = dst[2], y3 = dst[3];       // bl'  (s=0,t=1)
    const double x2 = dst[4], y2 = dst[5];       // br'  (s=1,t=1)
    const double x1 = dst[6], y1 = dst[7];       // ur'  (s=1,t=0)
    const double dx1 = x1 - x2, dx2 = x3 - x2, sx = x0 - x1 + x2 - x3;
    const double dy1 = y1 - y2, dy2 = y3 - y2, sy = y0 - y1 + y2 - y3;
    const double det = dx1 * dy2 - dx2 * dy1;
    const double g = (sx * dy2 - sy * dx2) / det;
    const double h = (dx1 * sy - dy1 * sx) / det;
    const double iw = 1.0 / (double)(IMG_W - 1), ih = 1.0 / (double)(IMG_H - 1);
    H[0] = (x1 - x0 + g * x1) * iw; H[1] = (x3 - x0 + h * x3) * ih; H[2] = x0;
    H[3] = (y1 - y0 + g * y1) * iw; H[4] = (y3 - y0 + h * y3) * ih; H[5] = y0;
    H[6] = g * iw;                  H[7] = h * ih;                  H[8] = 1.0;
}

// C = A * B  (torch.bmm, model_to_trace.py:168,188,323)
HNET_HD void mat3_mul(const double a[9], const double b[9], double c[9]) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
    for (int i = 0; i < 9; i++) c[i] = t[i];
}

// ensemble statistics of output component i (0..7) over the n MC samples (model_to_trace.py:274-281):
//   m_bar = mean_s m_i ; v_bar = mean_s exp(logvar_i) ; ens = mean_s (m_bar - m_i)^2 + v_bar ; p_bar = p4 + m_bar
// mean_s / logvar_s: [n][8] per-sample head outputs (logvar already x1e-3).  fp32 where the reference holds a tensor.
HNET_HD void ensemble_component(const float* mean_s, const float* logvar_s, int n, int i, double* pbar_i, double* ens_i) {
    double sm = 0, sv = 0;
    for (int s = 0; s < n; s++) {
        sm += (double)mean_s[s * 8 + i];
        sv += exp((double)logvar_s[s * 8 + i]);
    }
    const float mb = (float)(sm / n), vb = (float)(sv / n);
    double se = 0;
    for (int s = 0; s < n; s++) { const double d = (double)mb - (double)mean_s[s * 8 + i]; se += d * d; }
    *ens_i = (double)(float)((double)(float)(se / n) + (double)vb);
    *pbar_i = (double)(float)(p4(i) + (double)mb);
}

// transfer to the original frame + output assembly for one pair
//   (model_to_trace.py:18-38 transfer_mean_var_single, :311-317 assembly, :321-323 H_total)
HNET_HD void transfer_pair(const double pbar[8], const double ens[8], const float* H1f, float* mean8, float* cov64,
                           float* Htot9) {
    double H1[9];
    for (int i = 0; i < 9; i++) H1[i] = (double)H1f[i];
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
