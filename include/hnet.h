/* hnet.h — C ABI of the MI355X-native HomographyNet inference path (libhnet_hip.so).
 *
 * Drop-in boundary for the reference class `pytorch::HomographyNet`
 * (reference cuahn_ros/homography_network/src/HomographyNet.h:23-67).  The header-only adapter
 * include/HomographyNet.h re-creates that class surface on top of these entry points; INTEGRATION.md shows
 * the binding.  No C++ / torch types cross this boundary: plain pointers, sizes and status codes; every
 * buffer is caller-owned.  One context per device; a context is NOT thread-safe (the reference object is
 * driven by a single ROS spin thread, ros_subscribe_cuahn.cpp:123-135).
 *
 * Conventions (reference model_to_trace.py:79-83, HomographyNet.cpp:160-165):
 *   images  : 224 rows x 320 cols, 8-bit gray (or float32 already scaled to [0,1])
 *   corners : ul, bl, br, ur ; each (u = column, v = row) ; 8 floats in that order
 *   mean    : total 4-corner offsets img1 -> img2 in pixels (includes the prior), float[8]
 *   cov     : 8x8 row-major, block diagonal of four 2x2 blocks, symmetric, float[64]
 *   err map : |warp(img2, H_total) - img1| * 255, clamped to [0,255] when emitted as u8
 */
#ifndef HNET_H
#define HNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HNET_IMG_ROWS 224
#define HNET_IMG_COLS 320

/* status codes */
enum {
    HNET_OK = 0,
    HNET_ERR_INVALID_ARG = 1,
    HNET_ERR_BAD_WEIGHTS = 2,     /* missing / malformed HNETW001 blob */
    HNET_ERR_DEVICE = 3,          /* HIP runtime error (hnet_last_error has the text) */
    HNET_ERR_NOT_READY = 4,       /* fewer than two images pushed (HomographyNet.cpp:155-158) */
    HNET_ERR_CAPACITY = 5,        /* batch larger than max_batch; hnet_create: max_batch beyond what the kernels address (1 779 frame pairs per context) */
    HNET_ERR_UNSUPPORTED = 6
};

/* arithmetic of the conv contractions:
 *   HNET_PREC_FP32   exact fp32 MFMA (v_mfma_f32_32x32x2_f32), the reference's arithmetic
 *   HNET_PREC_BF16X3 fp32-grade accuracy on the bf16 matrix cores: every value is carried as three bf16 planes
 *                    (an exact split of its 24-bit significand) and each product is six bf16 MFMAs (csrc/igemm_s3.h)
 *   HNET_PREC_BF16   plain bf16 operands, fp32 accumulation: the same kernels reading ONE bf16 plane, one MFMA per product.
 *                    A REPORTED mode (BASELINE config 2 names "bf16"): ~2x the throughput at ~4e-3 .. 1e-1 px from the reference,
 *                    i.e. outside the 1e-4 px parity gate; tests/test_gpu_bf16_mode.py pins what it computes
 *   HNET_PREC_F16X2  fp32-grade accuracy on the fp16 matrix cores with THREE MFMAs per product: an activation is two fp16 planes
 *                    (a = A0 + A1 / 4096, 22 + 2 significand bits), a weight three (4096 w = W0 + W1, W0 / 4096), the accumulator
 *                    carries 4096 x the sum (csrc/s3_format.h).  Same parity gates as HNET_PREC_BF16X3.  Range: |weight| < 16
 *                    and |activation| < 32768 (guaranteed; up to 65520 all but 0.04 % of the values still split finitely, s3_format.h).
 *                    Outside it an overflow shows as a NON-FINITE result, never as a silently wrong one: the host-buffer entry points
 *                    (hnet_infer, hnet_infer_batch) then repeat the call in HNET_PREC_BF16X3 (hnet_precision below); the device-resident
 *                    entry points raise hnet_overflow_flag, which the caller polls after its own synchronisation */
enum { HNET_PREC_FP32 = 0, HNET_PREC_BF16 = 1, HNET_PREC_BF16X3 = 2, HNET_PREC_F16X2 = 3 };
enum { HNET_PIX_U8 = 0, HNET_PIX_F32 = 1 };        /* pixel format of image buffers */

/* Replaces: the variant choice the reference bakes into the traced .pt file
 * (trace_pytorch_model/trace_model.py:36-46; blocks_to_run model_to_trace.py:72; MC_dropout_num :202;
 * dropout_rate trace_model.py:16; "_showError" HomographyNet.cpp:96-100). */
/* use_prior, blocks_to_run, mc_samples, emit_error_map = HNET_FROM_FILE (dropout_p: any negative value): hnet_create takes the field from the blob's
 * `hnet.variant` record (python -m cuahn_vio_amd.weights --variant ...: one blob per traced variant, as the reference has one .pt per variant); a blob
 * without the record gives the reference's launch values (use_prior 1, blocks_to_run 3, N 16, p 0.05, no error map).  hnet_get_config returns what is in effect. */
#define HNET_FROM_FILE (-1)
typedef struct hnet_config {
    uint32_t struct_size;      /* sizeof(hnet_config), for forward compatibility */
    int32_t  device_id;        /* HIP device ordinal */
    int32_t  use_prior;        /* 0: full 4-block model.  1: H0 = DLT(prior), then `blocks_to_run` part-1 blocks, then block 4 */
    int32_t  blocks_to_run;    /* 1..3, the reference attribute (3 = "traced_model_3_blocks_using_prior"); used only with prior */
    int32_t  mc_samples;       /* N of the MC-dropout ensemble (reference 16) */
    float    dropout_p;        /* drop probability (reference 0.05); 0 = deterministic */
    uint64_t mc_seed;          /* key of the mask function, include/hnet_rng.h */
    int32_t  emit_error_map;   /* 1: "_showError" variant, the photometric error map is computed */
    int32_t  precision;        /* HNET_PREC_*; hnet_default_config: HNET_PREC_F16X2 */
    int32_t  max_batch;        /* capacity (frame pairs) of the persistent activation buffers, >= 1 */
    int32_t  mc_sample_begin;  /* this context evaluates global samples [begin, end) in the *_partial entry points; */
    int32_t  mc_sample_end;    /* 0,0 = all */
    /* Kernel selection (round 4: the library reads no environment variable; 0 everywhere = the measured defaults = what hnet_default_config sets).
     * These fields replace the HNET_* switches that rounds 1 - 3 read with getenv at hnet_create: a stray variable in a deployment can no longer
     * change kernels or summation order.  The tests and tools/ab_bench.py set them explicitly (the Python mirror maps its own environment onto them). */
    int32_t  warp_exact;       /* 1: the warp keeps grid_sample's sampling positions bit for bit (warp.py:70); 0: fast sampler, positions within 6e-5 px */
    int32_t  graph;            /* HNET_GRAPH_*: hipGraph replay of the batch-1 forward of hnet_infer */
    uint32_t variant;          /* HNET_VARIANT_*: reference kernels for in-process A/B measurements and the bitwise cross-kernel tests */
} hnet_config;

enum { HNET_GRAPH_DEFAULT = 0 /* replay */, HNET_GRAPH_OFF = 1 /* eager launches */, HNET_GRAPH_TIMING = 2 /* replay, also inside hnet_time_batch_device */ };
enum {
    HNET_VARIANT_GEMM_MASK = 0xff,            /* low byte: implicit-GEMM kernel selection (csrc/s3_dispatch.h); an unknown code is HNET_ERR_INVALID_ARG:
                                                  0 = defaults; 13 = heads FC1 on the four-wave 128 x 64 kernel; 22 = on the eight-wave kernel of round 3;
                                                  20 = conv layers on the four-wave lean kernels (no pipelined LDS-DMA / region kernels); 21 = the pipelined and the
                                                  region kernel at any batch (tests); 25 = no region kernel; 30 = split-K layers with splitk_reduce launches and the
                                                  split-K heads at every batch (the round-4 latency path: A/B and bitwise tests of round 5) */
    HNET_VARIANT_NO_LATENCY_PATH = 1u << 8,   /* batch <= 8 on the multi-launch path (bit-identical; tests/test_gpu_latency_path.py) */
    HNET_VARIANT_UNFUSED_B3 = 1u << 9,        /* block_3_0 and block_3_1 as separate launches (fp16-plane mode) */
    HNET_VARIANT_UNFUSED_B42 = 1u << 10,      /* block_4_2 and block_4_3 as separate launches (fp16-plane mode) */
    HNET_VARIANT_NO_CHAIN = 1u << 11,         /* batch <= 8: the tail layers of every block as separate launches instead of the one-XCD chain launch of round 6
                                                  (csrc/chain_lat.h; fp16-plane mode; same arithmetic, another summation order: results agree to fp32 rounding) */
    HNET_VARIANT_CHAIN_GRID_8 = 1u << 12,     /* tests: the chain launches with 8 workgroups instead of 256 (fewer resident workgroups than items: every workgroup works
                                                  through several items of a layer) and */
    HNET_VARIANT_CHAIN_GRID_3 = 1u << 13,     /* with 3 (XCDs without a workgroup: pairs are picked up by whoever is done) - the same bits as the 256-workgroup launch */
    HNET_VARIANT_GRAPH_COPIES = 1u << 15,     /* hnet_infer's graph moves {sequence number, prior} and {mean, cov, error map, flag} with memcpy nodes, as until round 6, instead of
                                                  letting the kernels read / write the pinned host block directly (A/B and tests; same results) */
    HNET_VARIANT_CHAIN_NO_FC = 1u << 16,      /* batch <= 8: the block-tail Linear(5120, 8) recomputed by every workgroup of the next warp + pool launch (rounds 3 - 5) instead of
                                                  summed from the 32 partial sums the tail chain's last layer leaves (round 6; another summation order: fp32 rounding) */
    HNET_VARIANT_WARP_FUSE = 1u << 14         /* batch > 8: block 4's warp + concat sampled INSIDE the block_4_0 + block_4_1 kernel (csrc/conv_b4_fused.h WARPIN, round 6; fp16-plane
                                                  mode, 4-byte aligned u8 images) instead of a launch of its own that writes the padded fp16 planes.  Same sampler, same bits
                                                  (tests/test_gpu_warp_fuse.py) - and 0.10 ms per 256 pairs SLOWER (the sampling sits in every workgroup's own timeline:
                                                  profiles/r06_experiments_not_shipped.log item 15), so it is opt-in, not the default */
};

typedef struct hnet_ctx hnet_ctx;

typedef struct hnet_timing {
    double device_ms;          /* "pure network inference": device time of the last forward (HomographyNet.cpp:178-188) */
    double host_ms;            /* wall time of the last hnet_infer / hnet_infer_batch call, incl. H2D and D2H */
    int64_t n_inferences;      /* every forward of hnet_infer / hnet_infer_batch (also the mask sequence number of hnet_infer) */
    double sum_device_ms_after_100;   /* running sum over the iteration == 0 calls that skips the first 100 (HomographyNet.cpp:245-251) */
    int64_t n_main_inferences; /* `inference_counting` (HomographyNet.cpp:189): calls with iteration == 0 only */
} hnet_timing;

/* fills `cfg` with the reference's launch defaults — full model, N=16, p=0.05, max_batch 1 — and precision =
 * HNET_PREC_F16X2 (fp32-grade results on the fp16 matrix cores; passes the same parity gates as HNET_PREC_BF16X3 and as
 * HNET_PREC_FP32, which is the reference's own fp32 arithmetic; both stay selectable) */
void hnet_default_config(hnet_config* cfg);

/* Replaces HomographyNet::load_network_model (HomographyNet.cpp:81-103): `weights_path` names an HNETW001
 * blob (cuahn_vio_amd/weights.py) instead of a TorchScript .pt.  Also runs the warm-up forward the reference
 * constructor does (HomographyNet.cpp:28-45). */
int hnet_create(const hnet_config* cfg, const char* weights_path, hnet_ctx** out);
int hnet_create_from_memory(const hnet_config* cfg, const void* blob, size_t len, hnet_ctx** out);
void hnet_destroy(hnet_ctx* ctx);

const char* hnet_status_string(int status);
const char* hnet_last_error(const hnet_ctx* ctx);   /* text of the last HNET_ERR_DEVICE etc.; never NULL */
const char* hnet_version(void);
/* Device-resident entry points and the fp16-plane range.  Every forward ORs bit 0 of a device word when one of its outputs (mean / cov, or
 * the per-sample head outputs of the *_partial path) is not finite; this call synchronises `stream` (NULL = the context's), returns the word in
 * *flags and clears it.  In HNET_PREC_F16X2 a set bit with finite inputs means an activation left the fp16-plane range: run that batch
 * again on a HNET_PREC_BF16X3 context (same results as fp32 arithmetic, fp32 range).  Costs no extra launch. */
int hnet_overflow_flag(hnet_ctx* ctx, void* stream, int* flags);

/* the arithmetic mode in effect (HNET_PREC_*).  It differs from the requested one in two cases, both HNET_PREC_F16X2 -> HNET_PREC_BF16X3
 * (same results, twice the matrix-core work): a weight >= 16 at hnet_create, or an activation beyond the fp16 range seen by hnet_infer /
 * hnet_infer_batch (non-finite outputs): the context re-packs its weights, repeats the call and stays in HNET_PREC_BF16X3.  The
 * device-resident entry points cannot look at their results: there such an overflow shows as non-finite outputs and raises
 * hnet_overflow_flag.  A non-finite INPUT (NaN prior from a diverged filter, NaN float image) is not an overflow: the call returns
 * the non-finite outputs as the reference would and the context keeps its mode. */
int hnet_precision(const hnet_ctx* ctx);
/* the configuration in effect: HNET_FROM_FILE fields resolved from the blob, defaults filled in (what a deployment logs next to the file name) */
int hnet_get_config(const hnet_ctx* ctx, hnet_config* out);

/* Replaces HomographyNet::load_current_img (HomographyNet.cpp:127-151): copies the 224x320 8-bit image
 * (row_stride in bytes) to the device, prev <- curr, curr <- img; counts images; records `t` from the second
 * image on. */
int hnet_push_image(hnet_ctx* ctx, const uint8_t* data, int rows, int cols, int row_stride, double t);

/* The reference's IEKF keeps a SECOND traced model for iteration > 0 (`HomographyNet_model_iterative`, loaded from network_model_iterative_path when
 * num_of_iteration > 1: HomographyNet.cpp:20-24,104-124, run at :209-219) - a separate file that may be another variant (fewer blocks).  Both
 * modules see the same nn_inputs (:160-172).  hnet_attach_images makes `ctx` (the iterative model's context) read the frame pair, the image counter,
 * the time stamp and the MC-dropout sequence number of `source` (the main model's context, same device): images are pushed to `source` only, and
 * hnet_infer(ctx, ...) runs on source's current pair with the next sequence number of the shared count.  `source` must outlive `ctx`. */
int hnet_attach_images(hnet_ctx* ctx, hnet_ctx* source);
int hnet_image_count(const hnet_ctx* ctx);             /* the public `img_counter` (HomographyNet.h:33) */

/* ---- image pre-processing ahead of load_current_img --------------------------------------------
 * The reference undistorts and resizes the raw camera image on the CPU before handing it to the network:
 * CamBase::initialize_undist_map / initialize_undist_map_fisheye build two 224x320 float maps with
 * cv::initUndistortRectifyMap / cv::fisheye::initUndistortRectifyMap towards the virtual camera f = 159.5,
 * c = (159.5, 111.5) (ov_core/src/cam/CamBase.h:165-180) and undistort_and_resize_img is cv::remap(INTER_LINEAR)
 * (:182-186), called from VioManager.cpp:184.  Here the maps are built once on the host with the published
 * formulas of those two OpenCV functions and the remap is a HIP kernel that writes straight into the context's image
 * ring, so a raw frame goes host -> device once and never comes back.
 * PARITY UNPINNED: OpenCV is neither in this image nor vendored by the reference and the reference holds no vectors
 * for this step.  The kernel interpolates with sample positions quantised to 1/32 px like cv::remap (INTER_BITS = 5)
 * in exact integer arithmetic.  Deviation class against cv::remap, exactly: OpenCV blends with a 32 x 32 table of 15-bit coefficients
 * (INTER_REMAP_COEF_BITS = 15: each 1-D weight k/32 is rounded into a pair that sums to 32768) and rounds the 2-D sum once, this
 * kernel blends with the exact products (32 - fx)(32 - fy) ... fx fy / 1024 and rounds half up: same sample positions, same four
 * taps, a result that can differ by ONE grey level where the exact blend sits within 2^-10 of a rounding boundary; and OpenCV's maps
 * pass through its fixed-point convertMaps, whose rounding of positions exactly half way between two 1/32-px steps may pick the other
 * step (again <= one grey level at a unit gradient).  What IS pinned: bit-exactness against the numpy restatement under tests (see tests/test_undistort.py), four analytic
 * map properties, and an end-to-end property independent of the restated formulas - the remap of an independently simulated fisheye
 * photograph of an analytic scene recovers the scene to 0.34 grey levels RMS, 1.0 max (tests/test_undistort.py). */
typedef struct hnet_camera {
    int32_t fisheye;           /* 1: equidistant model (cam0_is_fisheye, uzhfpv.launch:77), 0: radial-tangential */
    int32_t raw_rows, raw_cols;/* size of the raw image (cam0_wh, uzhfpv.launch:75: 640 x 480) */
    double  k[4];              /* fx, fy, cx, cy (cam0_k) */
    double  d[4];              /* fisheye: k1..k4; radtan: k1, k2, p1, p2 (cam0_d) */
} hnet_camera;
/* builds and uploads the maps for `cam` (initialize_undist_map[_fisheye]) */
int hnet_set_camera(hnet_ctx* ctx, const hnet_camera* cam);
/* or supplies them directly (the reference's undist_map1 / undist_map2: 224x320 float each, x and y source coordinates) */
int hnet_set_undistort_maps(hnet_ctx* ctx, const float* map_x, const float* map_y, int raw_rows, int raw_cols);
/* copies the maps in use back (224x320 floats each) */
int hnet_get_undistort_maps(hnet_ctx* ctx, float* map_x, float* map_y);
/* undistort_and_resize_img + load_current_img: raw 8-bit image (row_stride in bytes) -> remap on the device -> image ring */
int hnet_push_raw_image(hnet_ctx* ctx, const uint8_t* raw, int rows, int cols, int row_stride, double t);
/* operator-level: the remapped 224x320 image back on the host (parity tests) */
int hnet_op_undistort(hnet_ctx* ctx, const uint8_t* raw, int rows, int cols, int row_stride, uint8_t* out);
double hnet_latest_time(const hnet_ctx* ctx);          /* get_latest_inference_time() (HomographyNet.h:31) */

/* Replaces HomographyNet::network_inference (HomographyNet.cpp:153-252) on (prev, curr).
 * prior_px: 8 doubles (pixels) — required when the context was created with use_prior, else ignored/NULL.
 * iteration: IEKF iteration index; >0 selects the reference's "iterative" model, which is the same network
 *            run again with the updated prior, so it is accepted and otherwise ignored.
 * err_map_out: NULL or 224*320 bytes (needs emit_error_map).  Returns HNET_ERR_NOT_READY before 2 images. */
int hnet_infer(hnet_ctx* ctx, const double* prior_px, int iteration,
               float mean_out[8], float cov_out[64], uint8_t* err_map_out);

/* Batched frame pairs, host buffers: prev/curr [B][224][320] (pix_fmt), prior [B][8] floats or NULL,
 * pair_seq0 = sequence number of pair 0 (pair b uses pair_seq0 + b in the mask key),
 * mean [B][8], cov [B][64], err_map [B][224][320] float or NULL.  Semantics = the batch-1 reference applied
 * independently to each pair (the reference itself is batch-1 only, warp.py:64). */
int hnet_infer_batch(hnet_ctx* ctx, const void* prev, const void* curr, int pix_fmt, const float* prior,
                     int batch, uint64_t pair_seq0, float* mean, float* cov, float* err_map);

/* Same with every buffer resident in device memory; enqueues on `stream` (a hipStream_t) and does not synchronise.
 * NULL = the context's own (non-blocking) stream; to enqueue on HIP's legacy default stream pass hipStreamLegacy
 * ((hipStream_t)1) — the handle 0 some frameworks report for it cannot be told apart from NULL. */
int hnet_infer_batch_device(hnet_ctx* ctx, const void* d_prev, const void* d_curr, int pix_fmt,
                            const float* d_prior, int batch, uint64_t pair_seq0,
                            float* d_mean, float* d_cov, float* d_err_map, void* stream);

/* The same forward with PACKED outputs: d_out72 [batch][72] fp32, record of pair b = mean (8 corner offsets) followed by the row-major 8 x 8 covariance
 * (64).  This is the message a multi-GPU caller exchanges (one ncclAllGather / all_gather_into_tensor of batch x 288 bytes per rank:
 * tests/cpp/rccl_gather_example.cpp, cuahn_vio_amd/dist.py); written directly by the ensemble kernel, no packing copies.  Bit-identical values. */
#define HNET_PACKED_FLOATS 72
int hnet_infer_batch_packed_device(hnet_ctx* ctx, const void* d_prev, const void* d_curr, int pix_fmt, const float* d_prior, int batch,
                                   uint64_t pair_seq0, float* d_out72, float* d_err_map, void* stream);

/* MC-dropout sharding: the trunk runs on every rank, the heads only for this context's
 * global samples [mc_sample_begin, mc_sample_end).  Outputs per pair: mean_s / logvar_s
 * [B][n_local][8] and H_part1 [B][9].  After gathering all N samples (rank order = sample order) the
 * ensemble is finished with hnet_mc_finish_device in the reference's two-pass order
 * (model_to_trace.py:274-280). */
int hnet_infer_mc_partial_device(hnet_ctx* ctx, const void* d_prev, const void* d_curr, int pix_fmt,
                                 const float* d_prior, int batch, uint64_t pair_seq0,
                                 float* d_mean_s, float* d_logvar_s, float* d_h_part1, void* stream);
int hnet_mc_finish_device(hnet_ctx* ctx, const float* d_mean_s, const float* d_logvar_s, int n_total,
                          const float* d_h_part1, int batch, float* d_mean, float* d_cov, void* stream);
/* ... with the packed [batch][72] record of hnet_infer_batch_packed_device as output */
int hnet_mc_finish_packed_device(hnet_ctx* ctx, const float* d_mean_s, const float* d_logvar_s, int n_total, const float* d_h_part1,
                                 int batch, float* d_out72, void* stream);
/* Round 5 (BASELINE config 4 without layout launches): the ensemble straight from the buffer an all-gather fills.  Every rank passes ONE [2][B][n_local][8]
 * array to hnet_infer_mc_partial_device (d_mean_s = its first half, d_logvar_s = its second) and all-gathers it into d_gathered [world][2][B][n_local][8]
 * (ncclAllGather / all_gather_into_tensor: rank-major = global sample order); this call reads sample s of pair b at rank s / n_local - no stack, permute or
 * copy between the collective and the finish.  Same two-pass arithmetic and the same bits as hnet_mc_finish_packed_device on the re-ordered samples. */
int hnet_mc_finish_gathered_device(hnet_ctx* ctx, const float* d_gathered, int world, int n_local, const float* d_h_part1, int batch, float* d_out72,
                                   void* stream);

/* ---- context groups (round 6): INDEPENDENT steps on several contexts ------------------------------------------------------------
 * A forward of <= 64 frame pairs leaves most of the chip waiting on its own chain of dependent launches.  Where the steps are independent - a server's batches,
 * a rank's share of a streamed sequence (BASELINE configs 3 and 5) - issuing them round-robin on a few contexts, each with its own HIP stream and buffers, runs
 * one step's chain under the others' kernels: + 30 ... 40 % pairs/s at 32 - 64 pairs per step, + 5 % at 256 (bench.py --contexts).
 * A group is n_ctx contexts of ONE configuration on one device (weights uploaded per member).  It creates its streams before anything else, in a fixed order,
 * each on its own stream-priority level as far as the device has levels (three on this part): streams of one priority share the runtime's hardware queues in an
 * order that depends on the process's stream history, and two members on one queue serialise; queues of different priority are never shared.
 * hnet_group_infer_batch_packed_device = hnet_infer_batch_packed_device on member (call count mod n_ctx), on THAT member's stream; it does not synchronise.
 * Outputs of consecutive calls must not alias (two steps are in flight at once).  hnet_group_join makes `stream` wait for everything enqueued on the members so far
 * (one event per member); hnet_group_synchronize waits on the host.  Results are those of a single context, bit for bit (tests/test_gpu_group.py).
 * Not thread-safe (one caller thread, like a context). */
#define HNET_GROUP_MAX 8
typedef struct hnet_group hnet_group;
int hnet_create_group(const hnet_config* cfg, const char* weights_path, int n_ctx, hnet_group** out);
int hnet_create_group_from_memory(const hnet_config* cfg, const void* blob, size_t len, int n_ctx, hnet_group** out);
void hnet_destroy_group(hnet_group* group);
int hnet_group_size(const hnet_group* group);
hnet_ctx* hnet_group_context(hnet_group* group, int i);      /* member i (hnet_get_config, hnet_precision, per-member calls); owned by the group */
void* hnet_group_stream(hnet_group* group, int i);           /* member i's hipStream_t (to order a caller's own work - a collective, a copy - behind its step) */
const char* hnet_group_last_error(const hnet_group* group);
/* member: NULL, or receives the index of the member the step was enqueued on */
int hnet_group_infer_batch_packed_device(hnet_group* group, const void* d_prev, const void* d_curr, int pix_fmt, const float* d_prior, int batch,
                                         uint64_t pair_seq0, float* d_out72, float* d_err_map, int* member);
int hnet_group_join(hnet_group* group, void* stream);
int hnet_group_synchronize(hnet_group* group);
int hnet_group_overflow_flag(hnet_group* group, int* flags);   /* hnet_overflow_flag of every member, ORed; synchronises the members */

int hnet_synchronize(hnet_ctx* ctx, void* stream);
int hnet_last_timing(const hnet_ctx* ctx, hnet_timing* out);

/* Device-time measurement of `iters` back-to-back forwards on resident buffers (HIP events on the
 * context's stream).  per_iter_ms may be NULL.  Used by bench.py for latency percentiles. */
int hnet_time_batch_device(hnet_ctx* ctx, const void* d_prev, const void* d_curr, int pix_fmt,
                           const float* d_prior, int batch, uint64_t pair_seq0, float* d_mean, float* d_cov,
                           int iters, float* per_iter_ms, float* total_ms);

/* Per-stage device timing: a "stage" is one kernel launch of the forward (prep / conv layer / fc+DLT / heads).
 * hnet_profile_batch_device runs `iters` forwards with a HIP event after every launch on the context's stream
 * and returns the average milliseconds per stage ([hnet_stage_count] floats).  flops_per_pair = 2 x MACs.  The stage list is that of a
 * forward of max_batch pairs until a profile call names another batch: batches <= 8 take the latency path, which has fewer launches
 * (never more than the max_batch list), and hnet_stage_count / _name then describe the batch last profiled. */
int hnet_stage_count(const hnet_ctx* ctx);
const char* hnet_stage_name(const hnet_ctx* ctx, int i);
double hnet_stage_flops_per_pair(const hnet_ctx* ctx, int i);
/* kernels the stage's launch consisted of in the last profiled forward (a split-K layer with a separate reduce launch: 2); the fp32-MFMA mode reports 1 */
int hnet_stage_kernels(const hnet_ctx* ctx, int i);
int hnet_profile_batch_device(hnet_ctx* ctx, const void* d_prev, const void* d_curr, int pix_fmt,
                              const float* d_prior, int batch, uint64_t pair_seq0, float* d_mean, float* d_cov,
                              int iters, float* stage_ms_avg);

/* ---- operator-level entry points (parity tests of single kernels; host buffers, NCHW like the reference) ---- */

/* warp.py:60-79 — img [224][320] float32, H[9] -> out [224][320] */
int hnet_op_warp(hnet_ctx* ctx, const float* img, const float* H, float* out);
/* model_to_trace.py:42-61 — dst corners [n][8] -> H [n][9] */
int hnet_op_dlt(hnet_ctx* ctx, const float* dst, int n, float* H);
/* conv layer `layer` (0..19, execution order of cuahn_vio_amd/weights.py CONV_LAYERS) with its own weights:
 * in [B][Cin][H][W] -> out [B][Cout][Ho][Wo], + bias + LeakyReLU(0.1)   (model_to_trace.py:7-15) */
int hnet_op_conv(hnet_ctx* ctx, int layer, const float* in, int batch, int h, int w, float* out);
/* the fused block_4_0 + block_4_1 kernel of the split-bf16 mode (csrc/conv_b4_fused.h) on its own:
 * in [B][2][224][320] -> out [B][16][112][160] = conv_lrelu(conv_lrelu(in, block_4_0), block_4_1)   (model_to_trace.py:210-211).
 * reverse != 0 walks the tiles from the end of the batch.  HNET_ERR_UNSUPPORTED in the other arithmetic modes. */
int hnet_op_block4_fused(hnet_ctx* ctx, const float* in, int batch, int reverse, float* out);
/* the fused block_3_0 + block_3_1 kernel alone (fp16-plane mode): in [B][2][112][160] (NCHW fp32) -> out [B][32][56][80] = conv(conv(in)) with
 * the reference's conv() (model_to_trace.py:7-15, layers :108-109) */
int hnet_op_block3_fused(hnet_ctx* ctx, const float* in, int batch, float* out);
/* the fused block_4_2 + block_4_3 kernel alone (fp16-plane mode): in [B][16][112][160] -> out [B][64][28][40] (model_to_trace.py:212-213) */
int hnet_op_block42_fused(hnet_ctx* ctx, const float* in, int batch, float* out);
/* cat(img1, warp(img2,H)) -> AvgPool(k): img1,img2 [224][320] f32, H[9] or NULL (no warp), k in {1,2,4,8}
 * -> out [2][224/k][320/k]   (model_to_trace.py:153-157) */
int hnet_op_prep(hnet_ctx* ctx, const float* img1, const float* img2, const float* H, int k, float* out);
/* the same on u8 images as load_current_img receives them (u8 -> f32 / 255.0, HomographyNet.cpp:139-146) */
int hnet_op_prep_u8(hnet_ctx* ctx, const uint8_t* img1, const uint8_t* img2, const float* H, int k, float* out);
/* after a forward: copies the output of layer `layer` (0..19 convs) of pair `pair` as [Cout][Ho][Wo] */
int hnet_debug_layer_output(hnet_ctx* ctx, int layer, int pair, float* out, size_t capacity_floats);
/* after a forward: part-1 homography of pair `pair`, 9 floats */
int hnet_debug_h_part1(hnet_ctx* ctx, int pair, float* out9);

#ifdef __cplusplus
}
#endif
#endif /* HNET_H */
