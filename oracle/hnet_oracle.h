/* hnet_oracle.h — CPU restatement of the HomographyNet forward (TEST INFRASTRUCTURE, NOT PRODUCT).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
 * The product path (cuahn_vio_amd/csrc, include/hnet.h) never links or calls this code.
 *
 * What it restates: reference trace_pytorch_model/model_to_trace.py (combined_stu_model.forward,
 * :299-330 and everything it calls) and trace_pytorch_model/warp.py:60-79.  Layout is NCHW like the
 * reference.  Parity status: PINNED against the reference itself — the tests/golden npz vectors were produced by
 * importing the reference Python model in the build container (tools/gen_golden.py) on seeded synthetic
 * weights; tests/test_oracle_golden.py checks this oracle against every one of them.
 *
 * Two builds of the same source (oracle/Makefile):
 *   liboracle.so      ORACLE_ACC=double : sums and geometry accumulate in double, activations are stored as
 *                     float between layers.  This is the checker (closer to exact math than the reference's
 *                     own fp32 run, which sits up to 1.4e-4 px from its fp64 evaluation).
 *   liboracle_f32.so  ORACLE_ACC=float  : plain fp32 arithmetic like the reference; the timed CPU "port"
 *                     baseline of bench.py.
 */
#ifndef HNET_ORACLE_H
#define HNET_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_IMG_H 224
#define ORACLE_IMG_W 320
#define ORACLE_N_LAYERS 25   /* 20 convs (execution order) + fc_block_1..3 + fc_block_4_mean + fc_block_4_uncertainty */
#define ORACLE_N_STAT 19     /* sum, l2, numel, 16 samples: same definition as tools/gen_golden.py layer_stats */

typedef struct oracle_model oracle_model;

typedef struct {
    double layer_stats[ORACLE_N_LAYERS][ORACLE_N_STAT]; /* NaN-filled for layers not run */
    float  H_part1[9];            /* homography after blocks <=3 / prior */
    float  dlt_dst[5][8];         /* destination corners of each DLT call, call order; unused = NaN */
    int    n_dlt;
    float  feat[5120];            /* block-4 trunk output, NCHW flatten */
} oracle_trace;

int  oracle_load(const void* blob, size_t len, oracle_model** out);   /* HNETW001 blob; 0 = ok */
void oracle_free(oracle_model* m);
void oracle_set_threads(int n);                                       /* OpenMP threads (no-op without OpenMP) */
int  oracle_acc_bytes(void);                                          /* 8 for the double build, 4 for float */

/* building blocks (each cites the reference line it restates in the .c file) */
void oracle_avgpool(const float* in, int C, int H, int W, int k, float* out);
void oracle_conv_lrelu(const float* in, int Cin, int H, int W, const float* w, const float* b,
                       int Cout, int k, int stride, float* out);
void oracle_linear(const float* x, int n_in, const float* w, const float* b, int n_out, float* y);
void oracle_dlt(const float dst[8], float H[9]);                      /* src is always the image corners */
void oracle_warp(const float* img, const float H[9], float* out);     /* 224x320 */
void oracle_u8_to_f32(const uint8_t* in, size_t n, float* out);

/* Full forward for one frame pair.
 *   prior        : NULL = full 4-block model; else 8 px offsets (ul.u ul.v bl.u bl.v br.u br.v ur.u ur.v)
 *   blocks_to_run: with prior, number of part-1 blocks still run (3,2,1 — the reference attribute,
 *                  model_to_trace.py:72); ignored without prior
 *   n_mc, p      : MC-dropout samples and drop probability; masks from include/hnet_rng.h keyed by
 *                  (mc_seed, pair_seq)
 *   err_map      : NULL or 224*320 floats  (|warp(img2,H_total) - img1| * 255, model_to_trace.py:319-327)
 *   trace        : NULL or per-layer statistics
 */
int oracle_forward(const oracle_model* m, const float* img1, const float* img2, const float* prior,
                   int blocks_to_run, int n_mc, float p, uint64_t mc_seed, uint64_t pair_seq,
                   float mean[8], float cov[64], float* err_map, oracle_trace* trace);

/* The MC-dropout heads for a range of GLOBAL sample indices [s0, s1): per-sample mean offsets and
 * log-variances (already x1e-3), [s1-s0][8] each.  feat = block-4 trunk output (5120, NCHW flatten). */
void oracle_heads(const oracle_model* m, const float* feat, int s0, int s1, float p,
                  uint64_t mc_seed, uint64_t pair_seq, float* mean_s, float* logvar_s);

/* Ensemble statistics + transfer to the original frame + output assembly from per-sample head outputs
 * (model_to_trace.py:274-281, :18-38, :311-317). */
void oracle_finish(const float* mean_s, const float* logvar_s, int n_mc, const float H_part1[9],
                   float mean[8], float cov[64], float H_total[9]);

#ifdef __cplusplus
}
#endif
#endif
