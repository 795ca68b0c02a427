/* hnet_oracle.c — CPU restatement of the reference HomographyNet forward.  TEST INFRASTRUCTURE ONLY
 * (see hnet_oracle.h).  Plain C, no dependencies; NCHW like the reference.  Every function cites the
 * reference lines it follows (paths relative to /root/reference/trace_pytorch_model/).
 *
 * ORACLE_ACC (double | float) selects the accumulation / geometry type, see header.
 */
#include "hnet_oracle.h"
#include "../include/hnet_rng.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef ORACLE_ACC
#define ORACLE_ACC double
#endif
typedef ORACLE_ACC acc_t;

#define IH ORACLE_IMG_H
#define IW ORACLE_IMG_W
#define NPIX (IH * IW)

/* ------------------------------------------------------------------------------------------------
 * weights
 * ---------------------------------------------------------------------------------------------- */
typedef struct { const char* name; int cin, cout, k, stride; int part; } conv_spec;
/* execution order; model_to_trace.py:88-94, :99-104, :107-113, :210-216 */
static const conv_spec CONVS[20] = {
    {"block_1_1", 2, 128, 7, 2, 1}, {"block_1_2", 128, 128, 5, 2, 1}, {"block_1_3", 128, 256, 3, 2, 1},
    {"block_2_1", 2, 64, 7, 2, 1},  {"block_2_2", 64, 128, 5, 2, 1},  {"block_2_3", 128, 256, 3, 2, 1},
    {"block_2_4", 256, 256, 3, 2, 1},
    {"block_3_0", 2, 16, 7, 1, 1},  {"block_3_1", 16, 32, 5, 2, 1},   {"block_3_2", 32, 64, 3, 2, 1},
    {"block_3_3", 64, 128, 3, 2, 1}, {"block_3_4", 128, 256, 3, 2, 1}, {"block_3_5", 256, 256, 3, 2, 1},
    {"block_4_0", 2, 8, 7, 1, 4},   {"block_4_1", 8, 16, 5, 2, 4},    {"block_4_2", 16, 32, 3, 2, 4},
    {"block_4_3", 32, 64, 3, 2, 4}, {"block_4_4", 64, 128, 3, 2, 4},  {"block_4_5", 128, 256, 3, 2, 4},
    {"block_4_6", 256, 256, 3, 2, 4},
};

struct oracle_model {
    float* data;                       /* owned copy of all tensors */
    const float *cw[20], *cb[20];      /* conv weight [Cout,Cin,k,k], bias */
    const float *fcw[3], *fcb[3];      /* fc_block_1..3: [8,5120], [8] */
    const float *h1w[2], *h1b[2];      /* heads: Linear(5120,256)   0 = mean, 1 = uncertainty */
    const float *h2w[2], *h2b[2];      /* heads: Linear(256,8) */
};

static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

typedef struct { char name[128]; uint32_t dims[4]; uint32_t ndim; uint64_t off; size_t count; } blob_entry;

static const float* find_tensor(const blob_entry* e, int n, const float* data, const char* name, size_t expect) {
    for (int i = 0; i < n; i++)
        if (strcmp(e[i].name, name) == 0) return e[i].count == expect ? data + e[i].off / 4 : NULL;
    return NULL;
}

int oracle_load(const void* blob, size_t len, oracle_model** out) {
    const uint8_t* p = (const uint8_t*)blob;
    if (len < 12 || memcmp(p, "HNETW001", 8) != 0) return -1;
    uint32_t n = rd32(p + 8);
    if (n > 256) return -1;
    blob_entry* e = (blob_entry*)calloc(n, sizeof(blob_entry));
    size_t pos = 12, data_bytes = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (pos + 4 > len) { free(e); return -1; }
        uint32_t ln = rd32(p + pos); pos += 4;
        if (ln >= sizeof(e[i].name) || pos + ln + 4 > len) { free(e); return -1; }
        memcpy(e[i].name, p + pos, ln); e[i].name[ln] = 0; pos += ln;
        e[i].ndim = rd32(p + pos); pos += 4;
        if (e[i].ndim > 4 || pos + 4 * e[i].ndim + 8 > len) { free(e); return -1; }
        e[i].count = 1;
        for (uint32_t d = 0; d < e[i].ndim; d++) { e[i].dims[d] = rd32(p + pos); pos += 4; e[i].count *= e[i].dims[d]; }
        e[i].off = rd64(p + pos); pos += 8;
        size_t end = e[i].off + e[i].count * 4;
        if (end > data_bytes) data_bytes = end;
    }
    size_t data0 = (pos + 63) / 64 * 64;
    if (data0 + data_bytes > len) { free(e); return -1; }
    oracle_model* m = (oracle_model*)calloc(1, sizeof(oracle_model));
    m->data = (float*)malloc(data_bytes);
    memcpy(m->data, p + data0, data_bytes);
    char nm[160];
    int ok = 1;
    for (int i = 0; i < 20; i++) {
        const conv_spec* c = &CONVS[i];
        const char* prefix = c->part == 4 ? "model_last_block_list.0" : "model_part1";
        snprintf(nm, sizeof nm, "%s.%s.0.weight", prefix, c->name);
        m->cw[i] = find_tensor(e, n, m->data, nm, (size_t)c->cout * c->cin * c->k * c->k);
        snprintf(nm, sizeof nm, "%s.%s.0.bias", prefix, c->name);
        m->cb[i] = find_tensor(e, n, m->data, nm, c->cout);
        ok &= m->cw[i] && m->cb[i];
    }
    for (int i = 0; i < 3; i++) {
        snprintf(nm, sizeof nm, "model_part1.fc_block_%d.weight", i + 1);
        m->fcw[i] = find_tensor(e, n, m->data, nm, 8 * 5120);
        snprintf(nm, sizeof nm, "model_part1.fc_block_%d.bias", i + 1);
        m->fcb[i] = find_tensor(e, n, m->data, nm, 8);
        ok &= m->fcw[i] && m->fcb[i];
    }
    static const char* heads[2] = {"fc_block_4_mean", "fc_block_4_uncertainty"};
    for (int h = 0; h < 2; h++) {
        snprintf(nm, sizeof nm, "model_last_block_list.0.%s.1.weight", heads[h]);
        m->h1w[h] = find_tensor(e, n, m->data, nm, 256 * 5120);
        snprintf(nm, sizeof nm, "model_last_block_list.0.%s.1.bias", heads[h]);
        m->h1b[h] = find_tensor(e, n, m->data, nm, 256);
        snprintf(nm, sizeof nm, "model_last_block_list.0.%s.4.weight", heads[h]);
        m->h2w[h] = find_tensor(e, n, m->data, nm, 8 * 256);
        snprintf(nm, sizeof nm, "model_last_block_list.0.%s.4.bias", heads[h]);
        m->h2b[h] = find_tensor(e, n, m->data, nm, 8);
        ok &= m->h1w[h] && m->h1b[h] && m->h2w[h] && m->h2b[h];
    }
    free(e);
    if (!ok) { oracle_free(m); return -2; }
    *out = m;
    return 0;
}

void oracle_free(oracle_model* m) {
    if (!m) return;
    free(m->data);
    free(m);
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_acc_bytes(void) { return (int)sizeof(acc_t); }

/* ------------------------------------------------------------------------------------------------
 * building blocks
 * ---------------------------------------------------------------------------------------------- */

/* HomographyNet.cpp:139-141,144-146: `toType(kFloat) / 255.0` (fp32 division) */
void oracle_u8_to_f32(const uint8_t* in, size_t n, float* out) {
    for (size_t i = 0; i < n; i++) out[i] = (float)in[i] / 255.0f;
}

/* nn.AvgPool2d(k, stride=k, padding=0), model_to_trace.py:117-119 (used :139,157,175) */
void oracle_avgpool(const float* in, int C, int H, int W, int k, float* out) {
    int Ho = H / k, Wo = W / k;
    for (int c = 0; c < C; c++)
        for (int oy = 0; oy < Ho; oy++)
            for (int ox = 0; ox < Wo; ox++) {
                acc_t s = 0;
                for (int a = 0; a < k; a++)
                    for (int b = 0; b < k; b++) s += in[((size_t)c * H + oy * k + a) * W + ox * k + b];
                out[((size_t)c * Ho + oy) * Wo + ox] = (float)(s / (acc_t)(k * k));
            }
}

/* conv(): nn.Conv2d(k, padding=(k-1)//2, stride) + nn.LeakyReLU(0.1), model_to_trace.py:7-15.
 * Cross-correlation with zero padding; Ho = floor((H + 2p - k)/s) + 1. */
void oracle_conv_lrelu(const float* in, int Cin, int H, int W, const float* w, const float* b,
                       int Cout, int k, int stride, float* out) {
    const int p = (k - 1) / 2;
    const int Ho = (H + 2 * p - k) / stride + 1, Wo = (W + 2 * p - k) / stride + 1;
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static)
#endif
    for (int co = 0; co < Cout; co++)
        for (int oy = 0; oy < Ho; oy++) {
            acc_t row[ORACLE_IMG_W];
            for (int ox = 0; ox < Wo; ox++) row[ox] = (acc_t)b[co];
            for (int ci = 0; ci < Cin; ci++)
                for (int kh = 0; kh < k; kh++) {
                    int iy = oy * stride - p + kh;
                    if (iy < 0 || iy >= H) continue;
                    const float* irow = in + ((size_t)ci * H + iy) * W;
                    const float* wrow = w + (((size_t)co * Cin + ci) * k + kh) * k;
                    for (int kw = 0; kw < k; kw++) {
                        int lo = p - kw;                 /* need ox*stride >= p - kw */
                        int ox0 = lo <= 0 ? 0 : (lo + stride - 1) / stride;
                        int hi = W - 1 + p - kw;                     /* ox*stride <= W-1+p-kw */
                        if (hi < 0) continue;
                        int ox1 = hi / stride;
                        if (ox1 > Wo - 1) ox1 = Wo - 1;
                        const acc_t wv = (acc_t)wrow[kw];
                        const float* ip = irow + (kw - p);
                        for (int ox = ox0; ox <= ox1; ox++) row[ox] += wv * (acc_t)ip[ox * stride];
                    }
                }
            float* orow = out + ((size_t)co * Ho + oy) * Wo;
            for (int ox = 0; ox < Wo; ox++) {
                float v = (float)row[ox];
                orow[ox] = v > 0.0f ? v : v * 0.1f;
            }
        }
}

/* nn.Linear: y = x W^T + b, model_to_trace.py:97,105,115,224,227,231,234 */
void oracle_linear(const float* x, int n_in, const float* w, const float* b, int n_out, float* y) {
    for (int o = 0; o < n_out; o++) {
        acc_t s = (acc_t)b[o];
        const float* wr = w + (size_t)o * n_in;
        for (int i = 0; i < n_in; i++) s += (acc_t)wr[i] * (acc_t)x[i];
        y[o] = (float)s;
    }
}

/* image corners ul, bl, br, ur as (u, v): model_to_trace.py:79-83 */
static const float P4[8] = {0.f, 0.f, 0.f, IH - 1.f, IW - 1.f, IH - 1.f, IW - 1.f, 0.f};

/* DLT_solve, model_to_trace.py:42-61.  Row 2i of A = [x y 1 0 0 0 -u'x -u'y], row 2i+1 =
 * [0 0 0 x y 1 -v'x -v'y]; b = dst; h8 = inverse(A) b (explicit inverse then product, :57-58);
 * H = [h8, 1].  The inverse is Gauss-Jordan with partial pivoting in acc_t. */
static void dlt_acc(const acc_t dst[8], acc_t H[9]) {
    acc_t A[8][16];
    for (int i = 0; i < 4; i++) {
        acc_t x = P4[2 * i], y = P4[2 * i + 1], u = dst[2 * i], v = dst[2 * i + 1];
        acc_t r0[8] = {x, y, 1, 0, 0, 0, -u * x, -u * y};
        acc_t r1[8] = {0, 0, 0, x, y, 1, -v * x, -v * y};
        for (int j = 0; j < 8; j++) { A[2 * i][j] = r0[j]; A[2 * i + 1][j] = r1[j]; }
    }
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) A[i][8 + j] = (i == j) ? 1 : 0;
    for (int c = 0; c < 8; c++) {
        int piv = c;
        acc_t best = (acc_t)fabs((double)A[c][c]);
        for (int r = c + 1; r < 8; r++) {
            acc_t a = (acc_t)fabs((double)A[r][c]);
            if (a > best) { best = a; piv = r; }
        }
        if (piv != c)
            for (int j = 0; j < 16; j++) { acc_t t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
        acc_t inv = (acc_t)1 / A[c][c];
        for (int j = 0; j < 16; j++) A[c][j] *= inv;
        for (int r = 0; r < 8; r++) {
            if (r == c) continue;
            acc_t f = A[r][c];
            if (f == 0) continue;
            for (int j = 0; j < 16; j++) A[r][j] -= f * A[c][j];
        }
    }
    for (int i = 0; i < 8; i++) {
        acc_t s = 0;
        for (int j = 0; j < 8; j++) s += A[i][8 + j] * dst[j];
        H[i] = s;
    }
    H[8] = 1;
}

void oracle_dlt(const float dst[8], float H[9]) {
    acc_t d[8], h[9];
    for (int i = 0; i < 8; i++) d[i] = dst[i];
    dlt_acc(d, h);
    for (int i = 0; i < 9; i++) H[i] = (float)h[i];
}

/* torch.bmm(H, H_block), model_to_trace.py:168,188,323 — fp32 tensors in the reference */
static void mat3_mul(const float a[9], const float b[9], float c[9]) {
    float t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            acc_t s = 0;
            for (int k = 0; k < 3; k++) s += (acc_t)a[3 * i + k] * (acc_t)b[3 * k + j];
            t[3 * i + j] = (float)s;
        }
    memcpy(c, t, sizeof t);
}

/* WarpImg.warpSingleImage_H_Mtrx, warp.py:60-79: (X,Y,Z) = H (u,v,1); x = X/Z, y = Y/Z (:65-66);
 * normalise g = x * 2/(W-1) - 1 (:40,70); F.grid_sample(bilinear, zeros, align_corners=True) (:77)
 * un-normalises ix = ((g + 1) / 2) * (W - 1) and blends the 4 neighbours, out-of-range taps give 0. */
void oracle_warp(const float* img, const float H[9], float* out) {
    const acc_t fx = (acc_t)(2.0 / (IW - 1)), fy = (acc_t)(2.0 / (IH - 1));
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int v = 0; v < IH; v++)
        for (int u = 0; u < IW; u++) {
            acc_t X = (acc_t)H[0] * u + (acc_t)H[1] * v + (acc_t)H[2];
            acc_t Y = (acc_t)H[3] * u + (acc_t)H[4] * v + (acc_t)H[5];
            acc_t Z = (acc_t)H[6] * u + (acc_t)H[7] * v + (acc_t)H[8];
            acc_t gx = (X / Z) * fx - 1, gy = (Y / Z) * fy - 1;
            acc_t ix = ((gx + 1) / 2) * (IW - 1), iy = ((gy + 1) / 2) * (IH - 1);
            acc_t x0f = (acc_t)floor((double)ix), y0f = (acc_t)floor((double)iy);
            acc_t wx1 = ix - x0f, wx0 = 1 - wx1, wy1 = iy - y0f, wy0 = 1 - wy1;
            acc_t s = 0;
            /* NaN / huge coordinates (Z ~ 0): every tap is out of range -> 0, like grid_sample */
            if (x0f >= -1 && x0f <= IW && y0f >= -1 && y0f <= IH) {
                int x0 = (int)x0f, y0 = (int)y0f;
                if (y0 >= 0 && y0 < IH) {
                    if (x0 >= 0 && x0 < IW) s += (acc_t)img[y0 * IW + x0] * wx0 * wy0;
                    if (x0 + 1 >= 0 && x0 + 1 < IW) s += (acc_t)img[y0 * IW + x0 + 1] * wx1 * wy0;
                }
                if (y0 + 1 >= 0 && y0 + 1 < IH) {
                    if (x0 >= 0 && x0 < IW) s += (acc_t)img[(y0 + 1) * IW + x0] * wx0 * wy1;
                    if (x0 + 1 >= 0 && x0 + 1 < IW) s += (acc_t)img[(y0 + 1) * IW + x0 + 1] * wx1 * wy1;
                }
            }
            out[v * IW + u] = (float)s;
        }
}

/* statistics of one layer output, identical to tools/gen_golden.py layer_stats */
static void layer_stats(const float* a, size_t n, double st[ORACLE_N_STAT]) {
    double s = 0, q = 0;
    for (size_t i = 0; i < n; i++) { s += a[i]; q += (double)a[i] * a[i]; }
    st[0] = s; st[1] = sqrt(q); st[2] = (double)n;
    for (uint64_t i = 0; i < 16; i++) st[3 + i] = a[(i * 2654435761ULL + 12345ULL) % n];
}

/* ------------------------------------------------------------------------------------------------
 * heads, ensemble, transfer
 * ---------------------------------------------------------------------------------------------- */

/* fc_block_4_mean / fc_block_4_uncertainty = Dropout -> Linear(5120,256) -> LeakyReLU(0.1) -> Dropout ->
 * Linear(256,8) (model_to_trace.py:222-235); run_fc scales the uncertainty head by 1e-3 (:252-256).
 * Dropout in train mode (forced at :266-268): out = x * (keep / (1-p)). */
void oracle_heads(const oracle_model* m, const float* feat, int s0, int s1, float p,
                  uint64_t mc_seed, uint64_t pair_seq, float* mean_s, float* logvar_s) {
    const uint64_t key = hnet_pair_key(mc_seed, pair_seq);
    const uint32_t thr = hnet_drop_threshold(p);
    const float scale = 1.0f / (1.0f - p);
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static)
#endif
    for (int s = s0; s < s1; s++)
        for (int h = 0; h < 2; h++) {
            float d1[5120], hid[256], o[8];
            uint32_t pre_in = hnet_mask_prefix(key, (uint32_t)(2 * h), (uint32_t)s);
            uint32_t pre_hid = hnet_mask_prefix(key, (uint32_t)(2 * h + 1), (uint32_t)s);
            for (int k = 0; k < 5120; k++) d1[k] = hnet_mask_keep(pre_in, (uint32_t)k, thr) ? feat[k] * scale : 0.0f;
            oracle_linear(d1, 5120, m->h1w[h], m->h1b[h], 256, hid);
            for (int j = 0; j < 256; j++) {
                float v = hid[j] > 0.0f ? hid[j] : hid[j] * 0.1f;
                hid[j] = hnet_mask_keep(pre_hid, (uint32_t)j, thr) ? v * scale : 0.0f;
            }
            oracle_linear(hid, 256, m->h2w[h], m->h2b[h], 8, o);
            float* dst = (h == 0 ? mean_s : logvar_s) + (size_t)(s - s0) * 8;
            for (int i = 0; i < 8; i++) dst[i] = h == 0 ? o[i] : o[i] * 1e-3f;
        }
}

/* ensemble (model_to_trace.py:274-281), transfer_mean_var_single (:18-38), output assembly (:311-317),
 * and H_total = H_part1 * DLT(p4, p_bar) for the error map (:321-323) */
void oracle_finish(const float* mean_s, const float* logvar_s, int n_mc, const float H1[9],
                   float mean[8], float cov[64], float H_total[9]) {
    float ens[8], pbar[8];
    for (int i = 0; i < 8; i++) {
        acc_t sm = 0, sv = 0;
        for (int s = 0; s < n_mc; s++) {
            sm += (acc_t)mean_s[s * 8 + i];
            sv += (acc_t)exp((double)logvar_s[s * 8 + i]);          /* torch.exp, :274 */
        }
        float mb = (float)(sm / n_mc), vb = (float)(sv / n_mc);     /* .mean(0), :275-276 */
        acc_t se = 0;
        for (int s = 0; s < n_mc; s++) { acc_t d = (acc_t)mb - (acc_t)mean_s[s * 8 + i]; se += d * d; }   /* :278 */
        ens[i] = (float)((acc_t)(float)(se / n_mc) + (acc_t)vb);    /* :279-280 */
        pbar[i] = P4[i] + mb;                                       /* :281 */
    }
    memset(cov, 0, 64 * sizeof(float));
    for (int c = 0; c < 4; c++) {
        acc_t pu = pbar[2 * c], pv = pbar[2 * c + 1];
        acc_t X = (acc_t)H1[0] * pu + (acc_t)H1[1] * pv + (acc_t)H1[2];
        acc_t Y = (acc_t)H1[3] * pu + (acc_t)H1[4] * pv + (acc_t)H1[5];
        acc_t S = (acc_t)H1[6] * pu + (acc_t)H1[7] * pv + (acc_t)H1[8];
        mean[2 * c] = (float)(X / S - (acc_t)P4[2 * c]);            /* :23, :311 */
        mean[2 * c + 1] = (float)(Y / S - (acc_t)P4[2 * c + 1]);
        acc_t G[2][2] = {{(acc_t)H1[0] / S, (acc_t)H1[1] / S}, {(acc_t)H1[3] / S, (acc_t)H1[4] / S}};   /* :30 */
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++)                             /* :33-34, var = diag(var_u, var_v, 0) */
                cov[(2 * c + a) * 8 + 2 * c + b] =
                    (float)(G[a][0] * (acc_t)ens[2 * c] * G[b][0] + G[a][1] * (acc_t)ens[2 * c + 1] * G[b][1]);
    }
    if (H_total) {
        float Hb[9];
        oracle_dlt(pbar, Hb);
        mat3_mul(H1, Hb, H_total);
    }
}

/* ------------------------------------------------------------------------------------------------
 * full forward
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float *a, *b, *cat, *warped; } scratch;

static int conv_out(int n, int k, int s) { return (n + 2 * ((k - 1) / 2) - k) / s + 1; }

/* runs convs [first, last] of CONVS on `x` (C=2, h x w) and returns the pointer holding [256,4,5] */
static const float* run_convs(const oracle_model* m, int first, int last, const float* x, int h, int w,
                              scratch* sc, oracle_trace* tr) {
    const float* in = x;
    float* bufs[2] = {sc->a, sc->b};
    int cur = 0;
    for (int i = first; i <= last; i++) {
        const conv_spec* c = &CONVS[i];
        oracle_conv_lrelu(in, c->cin, h, w, m->cw[i], m->cb[i], c->cout, c->k, c->stride, bufs[cur]);
        h = conv_out(h, c->k, c->stride);
        w = conv_out(w, c->k, c->stride);
        if (tr) layer_stats(bufs[cur], (size_t)c->cout * h * w, tr->layer_stats[i]);
        in = bufs[cur];
        cur ^= 1;
    }
    return in;
}

/* one part-1 block: cat(img1, img2_or_warped) -> AvgPool(k) -> convs -> FC -> DLT (model_to_trace.py:136-188) */
static void run_block(const oracle_model* m, int blk, const float* img1, const float* img2w, scratch* sc,
                      oracle_trace* tr, float Hb[9]) {
    static const int first[3] = {0, 3, 7}, last[3] = {2, 6, 12}, pool[3] = {8, 4, 2};
    memcpy(sc->cat, img1, NPIX * sizeof(float));                    /* torch.cat(dim=1), :138,156,174 */
    memcpy(sc->cat + NPIX, img2w, NPIX * sizeof(float));
    int k = pool[blk], h = IH / k, w = IW / k;
    float* pooled = sc->warped + NPIX;                              /* second half of the warp buffer */
    oracle_avgpool(sc->cat, 2, IH, IW, k, pooled);
    const float* feat = run_convs(m, first[blk], last[blk], pooled, h, w, sc, tr);
    float fc[8], dst[8];
    oracle_linear(feat, 5120, m->fcw[blk], m->fcb[blk], 8, fc);     /* .view(bs,-1) = NCHW flatten, :143 */
    if (tr) layer_stats(fc, 8, tr->layer_stats[20 + blk]);
    for (int i = 0; i < 8; i++) dst[i] = P4[i] + fc[i];             /* :145 */
    if (tr && tr->n_dlt < 5) memcpy(tr->dlt_dst[tr->n_dlt++], dst, sizeof dst);
    oracle_dlt(dst, Hb);
}

int oracle_forward(const oracle_model* m, const float* img1, const float* img2, const float* prior,
                   int blocks_to_run, int n_mc, float p, uint64_t mc_seed, uint64_t pair_seq,
                   float mean[8], float cov[64], float* err_map, oracle_trace* tr) {
    if (n_mc < 1 || n_mc > 4096) return -1;
    if (prior && (blocks_to_run < 1 || blocks_to_run > 3)) return -1;
    scratch sc;
    sc.a = (float*)malloc(sizeof(float) * 8 * NPIX);                /* largest activation: 8x224x320 */
    sc.b = (float*)malloc(sizeof(float) * 8 * NPIX);
    sc.cat = (float*)malloc(sizeof(float) * 2 * NPIX);
    sc.warped = (float*)malloc(sizeof(float) * 2 * NPIX);
    if (tr) {
        for (int i = 0; i < ORACLE_N_LAYERS; i++)
            for (int j = 0; j < ORACLE_N_STAT; j++) tr->layer_stats[i][j] = NAN;
        for (int i = 0; i < 5; i++)
            for (int j = 0; j < 8; j++) tr->dlt_dst[i][j] = NAN;
        tr->n_dlt = 0;
    }
    float H[9], Hb[9];
    if (prior) {                                                    /* :129-130 */
        float dst[8];
        for (int i = 0; i < 8; i++) dst[i] = P4[i] + prior[i];
        if (tr) memcpy(tr->dlt_dst[tr->n_dlt++], dst, sizeof dst);
        oracle_dlt(dst, H);
    } else {                                                        /* block 1 on the raw pair, :136-148 */
        run_block(m, 0, img1, img2, &sc, tr, H);
    }
    if (!prior || blocks_to_run == 3) {                             /* block 2, :153-168 */
        oracle_warp(img2, H, sc.warped);
        run_block(m, 1, img1, sc.warped, &sc, tr, Hb);
        mat3_mul(H, Hb, H);
    }
    if (!prior || blocks_to_run >= 2) {                             /* block 3, :171-188 */
        oracle_warp(img2, H, sc.warped);
        run_block(m, 2, img1, sc.warped, &sc, tr, Hb);
        mat3_mul(H, Hb, H);
    }
    /* block 4: HomoNet_last_block.forward, :258-282 */
    oracle_warp(img2, H, sc.warped);
    memcpy(sc.cat, img1, NPIX * sizeof(float));
    memcpy(sc.cat + NPIX, sc.warped, NPIX * sizeof(float));
    const float* feat = run_convs(m, 13, 19, sc.cat, IH, IW, &sc, tr);
    float* mean_s = (float*)malloc(sizeof(float) * 8 * n_mc);
    float* lv_s = (float*)malloc(sizeof(float) * 8 * n_mc);
    oracle_heads(m, feat, 0, n_mc, p, mc_seed, pair_seq, mean_s, lv_s);
    if (tr) {
        memcpy(tr->feat, feat, sizeof(float) * 5120);
        memcpy(tr->H_part1, H, sizeof H);
        layer_stats(mean_s, (size_t)8 * n_mc, tr->layer_stats[23]);
        float* raw = (float*)malloc(sizeof(float) * 8 * n_mc);      /* hook sees the head before the 1e-3 */
        for (int i = 0; i < 8 * n_mc; i++) raw[i] = lv_s[i] / 1e-3f;
        layer_stats(raw, (size_t)8 * n_mc, tr->layer_stats[24]);
        free(raw);
    }
    float Htot[9];
    oracle_finish(mean_s, lv_s, n_mc, H, mean, cov, Htot);
    if (err_map) {                                                  /* :319-327 */
        oracle_warp(img2, Htot, sc.warped);
        for (int i = 0; i < NPIX; i++) err_map[i] = fabsf(sc.warped[i] - img1[i]) * 255.0f;
    }
    free(mean_s); free(lv_s);
    free(sc.a); free(sc.b); free(sc.cat); free(sc.warped);
    return 0;
}
