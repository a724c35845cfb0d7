/*
 * grl_hip.h -- C ABI of libgrl_hip.so: the MI355X (gfx950) kernels behind the
 * GRL per-clip forward/backward path and the evaluator distance matrix.
 *
 * The reference (flysnowtiger/GRL) has no FFI layer: every op on this path is a
 * stock PyTorch op called from Python.  Each entry point below names the
 * reference call site(s) it replaces (paths relative to /root/reference).
 * The Python host (grl_amd/engine.py) binds these with ctypes; see
 * INTEGRATION.md for the binding a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer is caller-owned DEVICE memory
 *     (fp32, contiguous unless a leading dimension is given); the library never
 *     allocates and keeps no global state;
 *   - activations are channels-last: an activation of N images, HxW pixels and C
 *     channels is the row-major matrix [N*H*W][C];
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, not waited;
 *   - return 0 on success, a negative GRL_E* code otherwise; grl_last_error()
 *     returns a thread-local message for the last failure.
 */
#ifndef GRL_HIP_H
#define GRL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRL_OK          0
#define GRL_EINVAL     -1   /* bad shape / pointer / alignment */
#define GRL_ELAUNCH    -2   /* hipLaunch failed; message has the HIP error string */

const char* grl_last_error(void);
/* Bumped on every incompatible change of a struct layout or an argument list below; grl_amd/_lib.py refuses a
 * library whose version differs from the one it was written against (round 1: 1, round 2: 2 -- GrlGemm / GrlWgrad
 * grew, grl_bn_bwd gained two pointers -- round 3: 3, then 4 with grl_stem_wgrad, relu_bits, 5: GrlGemm.bn_*;
 * round 4: 6 with grl_bottleneck_tail_bf16, 7 grl_gemm_force_tile; round 5: 8 with grl_conv_gemm_f32_group;
 * round 6: 9 with the grl_jpeg_* entry points). */
#define GRL_ABI_VERSION 9
int grl_abi_version(void);
/* `waiter` (a hipStream_t) waits for everything enqueued on `signaler` so far: hipEventRecord + hipStreamWaitEvent on a
 * pooled event, one call instead of the host framework's Event / stream-context objects (round 6: the train step is
 * host-bound in bf16 storage; this is the side-stream hand-off of the weight-gradient launches).  Not a reference op. */
int grl_stream_wait_stream(void* waiter, void* signaler);

/* epilogue selector of grl_conv_gemm_f32 */
#define GRL_EPI_AFFINE  0   /* y = relu?( rs[m]*(acc + gbias[m/rpg][n])*scale[n] + shift[n] + res[m][n] ) */
#define GRL_EPI_NEGDOT  1   /* y = -acc                                     (attevaluator.py:44-46)  */
#define GRL_EPI_EUCLID  2   /* y = sqrt(max(rnorm[m]+cnorm[n]-2acc,1e-12))  (attevaluator.py:33-41)  */
#define GRL_EPI_SQDIFF  3   /* TRL step (grl_model.py:146-149): v = relu(acc*scale + shift) is NOT stored;
                               y[m/32][n] = sum over the 32 rows of (v - res[row'][n])^2, row' = (m/res_rows)*
                               res_gstride + m%res_rows -- the conv_f1 output never reaches HBM.  fp32 storage,
                               128 x 128 tiles (M % 128 == 0), y is [M/32][ldy] fp32 partial sums            */

/* multiplier datapath of grl_conv_gemm_f32 (operands are fp32 in HBM in every mode) */
#define GRL_MATH_F32     0  /* exact fp32 MFMA: the documented fmaf chain (default)              */
#define GRL_MATH_BF16    1  /* operands rounded to bf16 while staging, bf16 MFMA (BASELINE cfg 2) */
#define GRL_MATH_BF16X3  3  /* split-bf16: hi*hi + hi*lo + lo*hi on the bf16 MFMA, ~2^-16 rel.   */
#define GRL_MATH_BF16S   2  /* bf16 STORAGE: a, w, res and y are bf16 arrays (lda/ldw/ldy/ldres in
                             * elements), fp32 accumulate + epilogue; `out_f32` keeps y fp32      */

/*
 * One fp32 MFMA GEMM  Y[M][N] = epilogue( A[M][K] . W[N][K]^T ), K-contiguous on
 * both operands.  With conv geometry set, A is gathered on the fly from a
 * channels-last image tensor (implicit GEMM): K = kh*kw*C, ordered tap-major then
 * channel, zero padding outside the image.
 *
 * Replaces: nn.Conv2d + eval-mode nn.BatchNorm2d + ReLU + residual add in
 *   reid/models/resnets1.py:76-91 (Bottleneck), reid/models/basebranch.py:42-50,61-62
 *   (GCE convs), reid/models/grl_model.py:71-83 (memo block), :146-147,160-161
 *   (f1/f2 biased convs); nn.Linear+BatchNorm1d in basebranch.py:38-40 and
 *   Siamese.py:84-94; torch.mm / addmm_ in reid/evaluator/attevaluator.py:33-46.
 *
 * Numerics: each output element is one fp32 accumulator updated by a k-ordered
 * chain of fused multiply-adds (v_mfma_f32_32x32x2_f32); the k order is documented
 * in DESIGN.md and reproduced bit-exactly by oracle/ref_c.
 */
typedef struct GrlGemm {
    const float* a;        /* dense: [M][lda]; conv: images [nimg][H][W][C]               */
    const float* w;        /* [N][ldw], K-contiguous (3x3 weights packed [N][tap][C])     */
    float*       y;        /* [M][ldy]                                                    */
    const float* scale;    /* [N] or NULL (=1)                                            */
    const float* shift;    /* [N] or NULL (=0)                                            */
    const float* res;      /* [M][ldres] residual or NULL                                 */
    const float* gbias;    /* [M/rows_per_group][N] added to acc before scale, or NULL    */
    const float* rowscale; /* [M] multiplies acc first, or NULL                           */
    const float* rnorm;    /* EUCLID: |a_m|^2 [M]                                         */
    const float* cnorm;    /* EUCLID: |w_n|^2 [N]                                         */
    float*       stats;    /* train mode: per-channel partial sums [gridM][2][N], or NULL */
    int32_t M, N, K;
    int32_t lda, ldw, ldy, ldres;
    int32_t rows_per_group;
    int32_t relu;
    int32_t epilogue;      /* GRL_EPI_*                                                   */
    /* conv geometry; conv == 0 means dense A */
    int32_t conv, H, W, C, Ho, Wo, kh, kw, stride, pad;
    int32_t math;          /* GRL_MATH_*: multiplier datapath (accumulation is always fp32)   */
    int32_t out_f32;       /* GRL_MATH_BF16S only: write y as fp32                            */
    int32_t res_rows, res_gstride;   /* GRL_EPI_SQDIFF: row mapping of `res` (rows per group, row stride between groups) */
    int32_t kblock;        /* GRL_MATH_F32 only: 1 = cut the accumulation chain every 512 k and sum the
                              segments (K-blocked accumulation: the accuracy class of a blocked CPU
                              sgemm; ~4 % slower on K >= 1024).  The train-mode forward sets it -- ReLU
                              masks, hence parameter gradients, agree with the reference's only as well
                              as the forward does; 0 = ONE k-ordered fmaf chain (eval path, evaluator). */
    /* Optional split-K scratch (round 3).  A skinny K-blocked GEMM (M <= 256, a handful of tiles: the per-clip
     * linears of GCE / TRL / the Siamese heads in train mode) walks K = 1024..2048 in ONE workgroup per tile --
     * latency-bound, ~60 us at 1 % MFMA busy.  With scratch the 512-k segments of the K-blocked chain run as
     * separate workgroups (grid.y = segments) writing raw partials [segment][M][N], and a second kernel adds them
     * in segment order and applies the epilogue: bit-identical to the one-workgroup K-blocked result.  NULL or
     * fewer floats than grl_conv_gemm_f32_workspace_floats() asks for: the one-workgroup form runs. */
    float*  splitk_ws;
    int64_t splitk_ws_floats;
    /* Optional BatchNorm-BACKWARD reduce in the epilogue (round 3; fp32 storage, AFFINE epilogue, 16-byte aligned
     * rows): when this GEMM produces the last contribution to the gradient of y = relu?(bn(z) (+res)) -- a data-
     * gradient GEMM whose output (+ `res`, the contributions so far) IS that gradient -- the epilogue masks it,
     * g = v * mask, writes g to `y` and leaves the two column sums of the BatchNorm backward in `stats`:
     * stats[tile][0][n] = sum_rows g, stats[tile][1][n] = sum_rows g * (z - mean) * invstd -- what
     * grl_bn_bwd's reduce pass computes, without re-reading the gradient (grl_bn_bwd_finish does the rest).
     * mask: bn_bits (the forward's recorded (y > 0) bytes, grl_bn_apply_centered) if given, else
     * ((z - mean) * bn_mscale + bn_mbeta > 0) if bn_mscale is given, else none.  bn_z NULL = off.
     * Round 5: GRL_MATH_BF16S too -- bn_z is then a bf16 [M][N] tensor, bn_bits one byte per EIGHT outputs, the sums are
     * taken from the fp32 value before it is rounded to bf16; needs M % 128 == 0 and N % 128 == 0 (or N == 64): the
     * reduce lives in the branch-free interior epilogue of the 128-row tile family (grl_bn_bwd_finish_bf16 does the rest). */
    const float*   bn_z;       /* [M][N] (row stride N) */
    const float*   bn_mean;    /* [N] */
    const float*   bn_invstd;  /* [N] */
    const float*   bn_mscale;  /* [N] or NULL */
    const float*   bn_mbeta;   /* [N] or NULL */
    const uint8_t* bn_bits;    /* [M * N / 4] (bf16 storage: [M * N / 8]) or NULL */
} GrlGemm;

int grl_conv_gemm_f32(const GrlGemm* desc, void* stream);
/* floats of split-K scratch the call above can use for this shape (0: it would not split) */
int64_t grl_conv_gemm_f32_workspace_floats(const GrlGemm* desc);
/* Kernel-tuning / test hook of the GRL_MATH_BF16S datapath: which launches take the 256 x 256
 * LDS-DMA tile (gemm_bf16.hip): -1 = automatic (enough tiles to fill the chip; the default),
 * 0 = never, 1 = whenever the shape is legal.  Returns the previous mode; results do not depend
 * on it (same MFMA, same k order).  Not thread-safe. */
int grl_gemm_bf16_tile_mode(int mode);
/* Kernel-tuning / test hook of the fp32-storage datapaths: force the workgroup tile of the following grl_conv_gemm_f32
 * calls to bm x bn (128x128, 128x64 or 64x64); (0, 0) returns to the per-shape rule.  Returns the previous setting as
 * (bm << 16) | bn (0 = automatic), GRL_EINVAL for any other pair.  Results never depend on the tile (one k-ordered
 * chain per output; the parity tests run every shape on every tile through this).  TEST HOOK: process-wide state,
 * not synchronised -- single-threaded callers only.  A forced tile is final: the statistics GEMMs' promotion to the
 * 128 x 128 tile (choose_tile) does not apply while one is set. */
int grl_gemm_force_tile(int bm, int bn);
/* rows of the stats slab the call above writes (= number of M tiles it will use) */
int grl_conv_gemm_f32_stat_rows(const GrlGemm* desc);
/* n (1..4) GEMMs in ONE launch where the kernels allow it (round 5): the same result, bit for bit, as n calls of
 * grl_conv_gemm_f32 in order -- which is also what runs when they do not (different shapes or flags, fp32 storage,
 * conv geometry, statistics ...).  Grouped today: GRL_MATH_BF16S dense GEMMs of one shape with the plain affine
 * epilogue (scale, shift, optional res, relu) that differ only in a, w, y, scale, shift, res.
 * Replaces: the conv1 / conv2 1x1 convolutions of the forward AND the backward memo block of one TRL step
 * (reid/models/grl_model.py:56-64 called at :155 and :167) -- two independent 8192-row GEMMs of the same shape that
 * each fill half a chip.  GRL_GEMM_GROUP=0: always n separate launches. */
int grl_conv_gemm_f32_group(const GrlGemm* descs, int n, void* stream);

/* [N][C][kh][kw] (torch layout) -> [N][kh*kw][C]; replaces nothing in the
 * reference (layout packing for the implicit GEMM). */
int grl_pack_conv_weight(const float* w, float* out, int N, int C, int kh, int kw, void* stream);

/* eval-mode BatchNorm folding: scale = g/sqrt(var+eps), shift = b - mean*scale
 * (+ scale*bias when the producing layer has a bias).  nn.BatchNorm{1,2}d in eval
 * mode everywhere on the path (e.g. resnets1.py:77,81,85). Any of gamma/beta NULL
 * means 1/0; mean/var NULL means 0/1 (plain bias -> shift). */
int grl_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                const float* bias, float eps, float* scale, float* shift, int C, void* stream);

/* Stem: 7x7 stride-2 pad-3 conv on NCHW input [n][3][H][W], y = relu?(conv*scale+shift) ->
 * channels-last [n][H/2][W/2][64]   (resnets1.py:101-103 / basebranch.py:28-30). */
int grl_stem_conv7x7(const float* x, const float* w /*[64][3][7][7]*/, const float* scale,
                     const float* shift, float* y, int n, int H, int W, int relu,
                     const float* wp /* optional: [64][164] image from grl_stem_pack_weight */,
                     void* stream);
/* the same stem reading RAW u8 pixels [n][3][H][W] and normalising them while the input patch
 * is staged in LDS: (u/255 - mean[c]) / std[c] with mean_std = {mean[3], std[3]} -- the fp32
 * operations of ToTensor + Normalize (reid/data/seqtransforms.py:190,195-216; constants
 * reid/data/dataloader.py:20), so the result is bit-identical to normalising on the host first.
 * a quarter of the input bytes over PCIe and HBM. */
int grl_stem_conv7x7_u8(const uint8_t* x, const float* mean_std, const float* w, const float* scale,
                        const float* shift, float* y, int n, int H, int W, int relu, const float* wp,
                        void* stream);
/* the normalisation alone (train mode keeps a float clip for the stem's weight gradient):
 * y[n][3][plane] = (x/255 - mean[c]) / std[c] */
int grl_normalize_u8(const uint8_t* x, const float* mean_std, float* y, int n, int64_t plane, void* stream);
/* Training input pipeline on the device (replaces the per-frame PIL transforms of
 * reid/data/seqtransforms.py:92-190 as composed in reid/data/dataloader.py:51-57): RandomHorizontalFlip
 * (per clip) + RandomSizedEarser (per frame: a constant-colour w x h patch pasted at (left, top) of the
 * flipped frame) + ToTensor + Normalize, raw uint8 clips [n_clips][T][3][H][W] -> float32.
 * params: int32 [n_clips][1 + 8*T] = {flip, T x {erase, left, top, w, h, R, G, B}}, the reference's
 * random draws made on the host (grl_amd/reid/data/augment.py).  W % 4 == 0. */
int grl_augment_normalize_u8(const uint8_t* x, const int32_t* params, const float* mean_std, float* y,
                             int n_clips, int T, int H, int W, void* stream);
/* RectScale (reid/data/seqtransforms.py:30-47: `frame.resize((W, H), Image.BILINEAR)`) on uint8 planes
 * [planes][Hin][Win] -> [planes][Hout][Wout], bit-identical to Pillow: horizontal pass, rounding to
 * uint8, vertical pass, 22-bit fixed-point taps.  bounds_*: int32 [out][2] = (first input index, taps);
 * coefs_*: int32 [out][k*]; both from grl_amd/reid/data/augment.py:pil_bilinear_coeffs. */
int grl_resize_bilinear_u8(const uint8_t* x, uint8_t* y, const int32_t* bounds_h, const int32_t* coefs_h, int kh,
                           const int32_t* bounds_v, const int32_t* coefs_v, int kv, int64_t planes, int Hin,
                           int Win, int Hout, int Wout, void* stream);
/* the stem's LDS weight image (K padded 147 -> 160, rows padded to 164 floats), made once per
 * weight version so that every workgroup copies it with 16-byte loads */
int grl_stem_pack_weight(const float* w, float* wp /* 64*164 floats */, void* stream);

/* 3x3 stride-2 pad-1 max pool, channels-last (resnets1.py:104). */
int grl_maxpool3x3s2(const float* x, float* y, int n, int H, int W, int C, void* stream);

/* Stem + max-pool in ONE launch, exact fp32 (round 5; eval: resnets1.py:101-104, basebranch.py:27-36): NCHW input
 * [n][3][H][128] (fp32, or raw u8 with mean_std as in grl_stem_conv7x7_u8) -> y = maxpool3x3s2(relu(conv7x7s2 * scale +
 * shift)) channels-last [n][H/4][32][64]; the stem map is neither written nor re-read.  W must be 128, H % 4 == 0.
 * wq: the register image of the weights from grl_stem_pack_weight_pool (64 * 168 floats).  Same products as
 * grl_stem_conv7x7 in another fp32 summation order (k-steps pair kx with kx + 4). */
int grl_stem_pack_weight_pool(const float* w /*[64][3][7][7]*/, float* wq /* 64*168 floats */, void* stream);
int grl_stem_pool_f32(const void* x, int x_is_u8, const float* mean_std, const float* scale, const float* shift, float* y,
                      int n, int H, int W, const float* wq, void* stream);

/* mean over `rows` consecutive rows: x [groups][rows][C] -> y [groups][ldy>=C]
 * (x.mean(-1).mean(-1)[.mean(1)] in basebranch.py:58, grl_model.py:151,165,178). */
int grl_group_mean(const float* x, float* y, int groups, int rows, int C, int ldy,
                   float out_scale, int accumulate, void* stream);

/* GCE tail (basebranch.py:49-50,62-66): map = sigmoid(bn(h[m].w3)); x_corr = x*map,
 * x_uncorr = x*(1-map).  h [M][256], x [M][C]. */
int grl_gce_gate(const float* h, const float* w3, const float* bn_scale, const float* bn_shift,
                 const float* x, float* corr_map, float* x_corr, float* x_uncorr,
                 int M, int Ch, int C, void* stream);

/* mean over T of x [b][T][rows*C] -> [b][rows*C]  (grl_model.py:137-138). */
int grl_temporal_mean(const float* x, float* y, int b, int T, int64_t inner, void* stream);

/* d[b][c] = mean_px (f1[b][px][c] - f2[b*f2_bstride + px][c])^2  (grl_model.py:149,163) */
int grl_sqdiff_mean(const float* f1, const float* f2, float* d, int b, int rows, int C,
                    int64_t f2_clip_stride, void* stream);

/* channel attention MLP: c = sigmoid(W2 relu(W1 d))  (grl_model.py:103-108,149,163);
 * then f_step[b][c] (+)= (1 + c) * gap[b][c]   (grl_model.py:150-151,164-165). */
int grl_channel_atte(const float* d, const float* w1 /*[Hd][C]*/, const float* w2t /*[Hd][C] = W2^T*/,
                     const float* gap, int64_t gap_stride, float* catte, float* fstep,
                     int64_t fstep_stride, int accumulate, int b, int C, int Hd,
                     float* hid_ws /* workspace [b][Hd] */, void* stream);

/* y = a + b elementwise (grl_model.py:68, memo + x_uncorr_t); b rows strided per clip */
int grl_add_strided(const float* a, const float* b, float* y, int nb, int64_t inner,
                    int64_t b_clip_stride, void* stream);

/* y[row] = l2normalize(x[row]*scale + shift)   (corr_bn/uncorr_bn + F.normalize,
 * grl_model.py:222-226); out rows strided so they can land inside a feature row. */
int grl_affine_l2norm(const float* x, const float* scale, const float* shift, float* y,
                      int rows, int C, int64_t ldy, void* stream);

/* Siamese.self_attention tail (Siamese.py:87-104): qk [b*T][2*D] (folded-BN Q|K),
 * x [b][T][C] -> pooled [b][ldy].  T <= 16. */
int grl_siamese_attn(const float* qk, const float* x, float* pooled, int b, int T, int D, int C,
                     int64_t ldy, void* stream);

/* y[b][c] = mean_T x[b][T][c] into a strided destination (attevaluator.py:112). */
int grl_mean_T(const float* x, float* y, int b, int T, int C, int64_t ldy, void* stream);

/* Pair verification head, eval mode (Siamese.py:127-140, Siamese_video.py:169-182):
 * out[i][j][c] = bias[c] + sum_k W[c][k]*(scale[k]*(p[i][k]-g[j][k])^2 + shift[k]). */
int grl_pair_verify(const float* p, const float* g, const float* scale, const float* shift,
                    const float* w, const float* bias, float* out, int np, int ng, int K,
                    int ncls, void* stream);

/* Row-wise ascending argsort of a distance matrix d [rows][ld] (first n columns), int32
 * indices out [rows][n]; ties go to the smaller index.  Replaces np.argsort(distmat, axis=1)
 * in reid/evaluator/eva_functions.py:139.  n <= 16384. */
int grl_row_argsort(const float* d, int64_t ld, int rows, int n, int32_t* idx, void* stream);
/* the same ranking for rows wider than one LDS network (16384 < n <= 2^24): the bitonic network cut at
 * the LDS size -- 16384-entry chunks sorted in LDS, the long-distance steps as global passes over
 * (key, index) pairs kept in `workspace` (grl_row_argsort_workspace_bytes(rows, n) bytes).  Same total
 * order (distance, then index) = np.argsort(kind='stable') of eva_functions.py:139. */
int64_t grl_row_argsort_workspace_bytes(int rows, int n);
int grl_row_argsort_wide(const float* d, int64_t ld, int rows, int n, int32_t* idx, void* workspace,
                         void* stream);

/* |x_row|^2 for the Euclidean epilogue (attevaluator.py:37-38). */
int grl_row_sqnorm(const float* x, float* out, int rows, int K, int ld, void* stream);

/* ------------------------------------------------------------------------------------
 * bf16-STORAGE pipeline (BASELINE configs[2]; engine math mode 'bf16s', GRL_MATH_BF16S):
 * bf16 twins of the bandwidth-bound kernels above -- activations are bf16 arrays (void*),
 * per-channel / per-clip vectors and all reductions stay fp32.  Same reference call sites.
 * ---------------------------------------------------------------------------------- */
int grl_cast_bf16(const float* x, void* y, int64_t n, void* stream);            /* n % 8 == 0 */
int grl_stem_conv7x7_bf16(const float* x, const float* w, const float* scale, const float* shift,
                          void* y, int n, int H, int W, int relu,
                          const void* wp /* optional: image from grl_stem_pack_weight_bf16 */, void* stream);
int grl_stem_conv7x7_u8_bf16(const uint8_t* x, const float* mean_std, const float* w, const float* scale,
                             const float* shift, void* y, int n, int H, int W, int relu, const void* wp,
                             void* stream);                                      /* u8 input, as grl_stem_conv7x7_u8 */
int grl_stem_pack_weight_bf16(const float* w, void* wp /* 64*184 bf16: rows of 368 bytes, k ordered (channel, ky, kx padded to 8) + zero pad */, void* stream);
int grl_maxpool3x3s2_bf16(const void* x, void* y, int n, int H, int W, int C, void* stream);
int grl_group_mean_bf16(const void* x, float* y, int groups, int rows, int C, int ldy,
                        float out_scale, int accumulate, void* stream);
int grl_sqdiff_mean_bf16(const void* f1, const void* f2, float* d, int b, int rows, int C,
                         int64_t f2_clip_stride, void* stream);
int grl_gce_gate_bf16(const void* h, const float* w3, const float* bn_scale, const float* bn_shift,
                      const void* x, float* corr_map, void* x_corr, void* x_uncorr, int M, int Ch,
                      int C, void* stream);
int grl_temporal_mean_bf16(const void* x, void* y, int b, int T, int64_t inner, void* stream);
int grl_add_strided_bf16(const void* a, const void* b, void* y, int nb, int64_t inner,
                         int64_t b_clip_stride, void* stream);

/* ------------------------------------------------------------------------------------
 * Train mode: batch-statistics BatchNorm and the backward pass (autograd of the modules
 * above, driven by reid/train/trainer.py:54 `loss.backward()`).
 * ---------------------------------------------------------------------------------- */

/* column partials of x [M][C] (row stride ld): slab [grl_col_stats_rows(M)][2][C] = sum, sum sq of
 * (x - pivot[c]); pivot NULL = 0.  For BatchNorm statistics pass a row of x (and the same vector to
 * grl_bn_stats_finalize): the shifted moments keep the variance when |mean| >> spread. */
int grl_col_stats_rows(int M);
int grl_col_stats(const float* x, float* slab, int M, int C, int ld, const float* pivot, void* stream);
/* out[c] (+)= sum_r slab[r*stride + c]  (bias gradients, partial reductions) */
int grl_slab_sum(const float* slab, int rows, int64_t stride, int C, float* out, int accumulate,
                 void* stream);

/* nn.BatchNorm forward in training mode, step 1: from a partial slab [rows][2][C]
 * (written by grl_conv_gemm_f32's `stats` or by grl_col_stats) to batch mean / invstd,
 * folded scale/shift and the running-stat update (momentum, unbiased running var;
 * num_batches_tracked, if given, is incremented -- torch's int64 buffer). */
int grl_bn_stats_finalize(const float* slab, int rows, int C, int64_t count, const float* gamma,
                          const float* beta, float* running_mean, float* running_var,
                          int64_t* num_batches_tracked, float momentum, float eps, float* mean,
                          float* invstd, float* scale, float* shift,
                          const float* pivot /* the grl_col_stats pivot, or NULL */, void* stream);
/* step 2: y = relu?(z*scale + shift + res) */
int grl_bn_apply(const float* z, const float* scale, const float* shift, const float* res,
                 float* y, int64_t M, int C, int relu, void* stream);
/* BatchNorm finalize INSIDE the apply pass (round 6; train_bnfuse.hip): grl_bn_stats_finalize + grl_bn_apply_centered as ONE
 * launch for layers whose statistics slab is small (rows <= 64, i.e. M <= 8192 pixel rows -- the TRL memo bottleneck of
 * grl_model.py:93-128,186-205 --, C % 64 == 0): every workgroup of the apply pass reduces the slab columns of its own 64
 * channels, in the order of grl_bn_stats_finalize, so mean / invstd / scale / shift / running statistics and y are
 * bit-identical to the two-launch form.  Arguments: those of the two calls.  The backward entry points (grl_bn_bwd*,
 * grl_bn_bwd_finish*) take the same form by themselves when the slab allows it.  OFF unless GRL_BN_FINAPPLY=1 (or
 * grl_bn_finalize_apply_mode(1)): measured 0.4-0.9 ms per step slower than the separate launches (EXPERIMENTS.md round 6). */
int grl_bn_finalize_apply_takes(int rows, int C);      /* 1 if the fused form covers the layer */
int grl_bn_finalize_apply_mode(int on);                 /* test hook: 0 / 1 = off / on for the process, -1 = query; returns the previous setting */
int grl_bn_finalize_apply(const float* slab, int rows, int C, int64_t count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                          float* mean, float* invstd, float* scale, float* shift, const float* pivot, const float* z,
                          const float* res, float* y, int M, int relu, uint8_t* relu_bits, void* stream);
int grl_bn_finalize_apply_bf16(const float* slab, int rows, int C, int64_t count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                               float* mean, float* invstd, float* scale, float* shift, const float* pivot, const void* z,
                               const void* res, void* y, int M, int relu, uint8_t* relu_bits, void* stream);
/* step 2, train-mode form: y = relu?((z - mean)*scale + beta + res), centred before the multiply as
 * torch's training kernel does (F.batch_norm(training=True); resnets1.py:76-91, grl_model.py:222-226):
 * exact where the folded form cancels (BatchNorm1d over a few similar rows).  beta may be NULL. */
int grl_bn_apply_centered(const float* z, const float* mean, const float* scale, const float* beta,
                          const float* res, float* y, int64_t M, int C, int relu,
                          uint8_t* relu_bits /* may be NULL: M*C/4 bytes, bit e of byte i = (y[4i + e] > 0) */,
                          void* stream);

/* BatchNorm (+ReLU) backward: g = dy*(act>0) (act NULL: no mask -- unless mask_scale is given: then the ReLU
 * mask of y = relu((z - mean)*mask_scale + mask_beta) is RECOMPUTED from z with the forward's own three fp32
 * operations (grl_bn_apply_centered without a residual), so the activation is not read at all);
 * dgamma += sum g*xhat; dbeta += sum g; dz = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)).
 * slab_ws: grl_col_stats_rows(M)*2*C floats, coef_ws: 2*C floats. gamma/dgamma/dbeta may be NULL.
 * gres (may be NULL): gradient of the residual input of y = relu(bn(z) + res), which is the same
 * masked g: gres (+)= g in the same pass (resnets1.py:88-91).  gres == dy with act given and gres_accumulate == 0:
 * IN-PLACE form -- the reduce pass overwrites dy with g (dy is NOT const then) and the apply pass reads it back, so
 * the caller's dy buffer becomes the residual's gradient: the activation is read once and no second tensor is written.
 * relu_bits (may be NULL): the mask bytes grl_bn_apply_centered recorded in the forward; they replace `act` (which
 * is then not read at all -- 1/16 of its bytes): same mask, same result. */
int grl_bn_bwd(const float* dy, const float* z, const float* act, const float* mean,
               const float* invstd, const float* gamma, float* dz, float* dgamma, float* dbeta,
               float* slab_ws, float* coef_ws, int M, int C, float* gres, int gres_accumulate,
               const float* mask_scale, const float* mask_beta, const uint8_t* relu_bits, void* stream);
/* The second half of grl_bn_bwd for a gradient that is ALREADY masked and reduced (GrlGemm.bn_z: the producing
 * data-gradient GEMM's epilogue left g in `g` and the partial sums in `slab`, `rows` of them): finalize (dgamma,
 * dbeta, the two means) and apply dz = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)); gres (+)= g as in grl_bn_bwd. */
int grl_bn_bwd_finish(const float* g, const float* z, const float* mean, const float* invstd, const float* gamma,
                      float* dz, float* dgamma, float* dbeta, const float* slab, int rows, float* coef_ws, int M, int C,
                      float* gres, int gres_accumulate, void* stream);
/* the bf16-storage twin (g, z, dz, gres bf16; round 5): the tail of grl_bn_bwd_bf16 behind a GrlGemm.bn_z GEMM */
int grl_bn_bwd_finish_bf16(const void* g, const void* z, const float* mean, const float* invstd, const float* gamma,
                           void* dz, float* dgamma, float* dbeta, const float* slab, int rows, float* coef_ws, int M, int C,
                           void* gres, int gres_accumulate, void* stream);

/* out (+)= dy * (act > 0)   (ReLU backward; act NULL = plain copy/accumulate) */
int grl_relu_bwd(const float* dy, const float* act, float* out, int64_t n, int accumulate, void* stream);
/* y = alpha*a + beta*b (b may be NULL) */
int grl_axpby(const float* a, const float* b, float* y, float alpha, float beta, int64_t n, void* stream);

/* dst[b*dst_stride + i] (+)= alpha*src[b*src_stride + i], i < inner (gradient accumulation
 * into one frame of a [b][T][...] tensor) */
int grl_axpy_strided(float* dst, int64_t dst_stride, const float* src, int64_t src_stride, int nb,
                     int64_t inner, float alpha, int accumulate, void* stream);
/* y[C][R] = x[R][C]^T (weights for the data-gradient GEMM of 1x1 convs / linears) */
int grl_transpose(const float* x, float* y, int R, int C, int ldx, void* stream);
/* [N][C][kh][kw] -> [C][flipped tap][N]: data gradient of a kxk conv as a conv over dz */
int grl_pack_dgrad_weight(const float* w, float* out, int N, int C, int kh, int kw, void* stream);
/* zero-stuffing of a stride-2 conv's output gradient: up[img][2oy+oy_off][2ox+ox_off] =
 * dz[img][oy][ox], zero elsewhere; accumulate == 1: += at those pixels, accumulate == 2: = at those
 * pixels, and nothing else is touched either way (the data gradient of a stride-2 conv is computed
 * at OUTPUT resolution, one GEMM per input-pixel parity class, and scattered this way) */
int grl_dilate2(const float* dz, float* up, int n, int Ho, int Wo, int H, int W, int C, int accumulate,
                int oy_off, int ox_off, void* stream);
/* nn.MaxPool2d(3,2,1) backward (first-maximum rule, deterministic gather form) */
int grl_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int n, int H, int W, int C,
                         void* stream);
/* Stem tail of the training forward in one pass: y = maxpool3x3/s2/p1(relu((z - mean) * scale + beta)) (resnets1.py:108-
 * 110 in train mode: bn1, relu, maxpool) plus idx[pixel][C] (uint8: ky*3 + kx of the window's FIRST maximum in scan
 * order, torch's argmax rule) -- the post-ReLU map is never written.  grl_maxpool3x3s2_bwd_idx routes dy through idx
 * (dx = gradient of the post-ReLU map, H x W = the map's size); the ReLU mask then comes from grl_bn_bwd's mask_scale
 * form.  idx: 4-byte aligned (bf16: 8-byte). */
int grl_bn_relu_maxpool3x3s2(const float* z, const float* mean, const float* scale, const float* beta, float* y,
                             uint8_t* idx, int n, int H, int W, int C, void* stream);
int grl_maxpool3x3s2_bwd_idx(const uint8_t* idx, const float* dy, float* dx, int n, int H, int W, int C, void* stream);
/* Weight gradient of the 7x7/s2/p3 stem conv (resnets1.py:106-107 / basebranch.py:27) straight from the NCHW clip:
 * dw[64][3][7][7] (+)= sum over output pixels of dz[pixel][64] (x) the pixel's 147 input taps -- no im2col matrix
 * (grl_stem_im2col + grl_conv_wgrad_f32 write and re-read 671 MB per 32 x 4 step).  x: fp32 [n][3][H][W]; dz: fp32, or
 * bf16 when dz_bf16 (converted exactly); exact fp32 products.  ws: grl_stem_wgrad_workspace_floats(n, H, W) floats
 * (one partial [64][160] per persistent workgroup, summed in workgroup order: deterministic). */
int64_t grl_stem_wgrad_workspace_floats(int n, int H, int W);
int grl_stem_wgrad(const float* x, const void* dz, int dz_bf16, float* dw, float* ws, int n, int H, int W,
                   int accumulate, void* stream);
/* im2col of the NCHW stem input for the 7x7 weight gradient: [n*H/2*W/2][Kp], 147 real columns */
int grl_stem_im2col(const float* x, float* col, int n, int H, int W, int Kp, void* stream);

/* Weight gradient  dW[n][k] (+)= sum_m dz[m][n] * X[m][k]  as an fp32 MFMA GEMM with the
 * pixel reduction split over workgroups (deterministic slab reduction).  conv != 0: X is
 * gathered from channels-last images and dW is written in torch layout [N][C][kh][kw]. */
typedef struct GrlWgrad {
    const float* dz;        /* [M][ldz]                                   */
    const float* x;         /* dense [M][ldx] or images [nimg][H][W][C]   */
    float*       dw;        /* [N][k_out] (dense) or [N][C][kh][kw]       */
    float*       workspace; /* grl_wgrad_workspace_floats(desc) floats    */
    int32_t M, N, K, ldz, ldx;
    int32_t k_out;          /* dense: columns of dw actually written (0 = K) */
    int32_t accumulate;
    int32_t conv, H, W, C, Ho, Wo, kh, kw, stride, pad;
    int32_t math;           /* GRL_MATH_F32 (exact), GRL_MATH_BF16X3 (split-bf16 products) or GRL_MATH_BF16: the
                               bf16 MFMA datapaths apply to 128 x 128 tiles (N >= 128, C or K % 128 == 0), other
                               shapes run exact fp32; accumulation and the slab reduction are always fp32 */
    int32_t in_bf16;        /* 1: dz and x are bf16 tensors (ldz / ldx in elements; train_engine math 'bf16s'): plain
                               bf16 products on 128 x 128 tiles for every shape (N, K, ld % 8 == 0), fp32 dW */
} GrlWgrad;
int64_t grl_wgrad_workspace_floats(const GrlWgrad* desc);
int grl_conv_wgrad_f32(const GrlWgrad* desc, void* stream);

/* GCE gate in train mode (logit already batch-normalised, first column of y[M][ldy]):
 * map = sigmoid(y), x_corr = x*map, x_uncorr = x*(1-map)          (basebranch.py:63-66) */
int grl_gate_apply(const float* y, int ldy, const float* x, float* cmap, float* xc, float* xu,
                   int M, int C, void* stream);
/* its backward: dx (+)= dxc*map + dxu*(1-map); dy[m*ldy] = map(1-map) sum_c (dxc-dxu)*x */
int grl_gate_bwd(const float* dxc, const float* dxu, const float* x, const float* cmap, float* dx,
                 int accumulate, float* dy, int ldy, int M, int C, void* stream);
/* dst[m][c] (+)= v[m / rows_per_group][c] * scale: backward of every mean over rows
 * (GAP, global descriptor, temporal mean with C = a whole frame) */
int grl_add_rowbcast(float* dst, const float* v, int64_t M, int64_t C, int64_t rows_per_group,
                     float scale, int accumulate, void* stream);
/* backward of grl_sqdiff_mean: df1 = 2(f1-f2)dd/rows, df2 (+)= -df1 */
int grl_sqdiff_bwd(const float* f1, const float* f2, const float* dd, float* df1, float* df2,
                   int b, int rows, int C, int64_t f2_clip_stride, int accumulate_df2, void* stream);
/* backward of f_step = gap*c + gap through the sigmoid: ds = dfs*gap*c(1-c), dgap (+)= dfs*(1+c) */
int grl_catte_bwd(const float* dfs, int64_t dfs_stride, const float* gap, int64_t gap_stride,
                  const float* catte, float* ds, float* dgap, int64_t dgap_stride, int accumulate,
                  int b, int C, void* stream);
/* backward of y = v/|v| (F.normalize, grl_model.py:223,226): dv = (dy - y(y.dy))/|v| */
int grl_l2norm_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* v,
                   float* dv, int rows, int C, void* stream);
/* backward of grl_siamese_attn: gradients of the (BN-ed) Q|K rows and of the frame features */
int grl_siamese_attn_bwd(const float* qk, const float* x, const float* out, int64_t ldo,
                         const float* dout, int64_t lddo, float* dqk, float* dx, int dx_accumulate,
                         int b, int T, int D, int C, void* stream);
/* verification head, train mode: diff[i*ng+j] = (p_i - g_j)^2 and its backward (Siamese.py:133-137) */
int grl_pair_sqdiff(const float* p, const float* g, float* diff, int np, int ng, int K, void* stream);
int grl_pair_sqdiff_bwd(const float* p, const float* g, const float* ddiff, float* dp, float* dg,
                        int np, int ng, int K, void* stream);

/* OIM look-up-table update performed inside OIM.backward (reid/loss/oim.py:24-26): per sample,
 * in batch order, lut[y] = m*lut[y] + (1-m)*x, then renormalise the row.  labels: int64; a
 * label outside [0, num_classes) updates nothing (lut is [num_classes][D]). */
int grl_oim_update(float* lut, const float* x, const int64_t* labels, int n, int D, int num_classes,
                   float momentum, void* stream);

/* ---- the trainer's loss block, forward and backward on the device ---- */
/* F.cross_entropy of OIMLoss.forward (reid/loss/oim.py:52; mean reduction, optional class
 * weight): loss[0] = sum_i w[y_i] (logsumexp(z_i) - z_i[y_i]) / sum_i w[y_i];
 * dlogits[i][j] = d loss[0] / d z_i[j] (may be NULL); correct[0] (may be NULL) = number of rows
 * whose arg-max (first index on ties, as topk in eva_functions.py:118-131) equals the label.
 * Labels outside [0,c) carry weight 0 (torch's ignore_index).  ws: 2*n floats of scratch. */
int grl_softmax_ce(const float* logits, int64_t ld, const int64_t* labels, const float* weight, int n,
                   int c, float* loss, float* correct, float* dlogits, int64_t ldd, float* ws,
                   void* stream);
/* OIM.backward's input gradient (oim.py:22) with the scalar of oim.py:50 and the upstream
 * gradient g[0] (device scalar, NULL = 1) folded in: dx = alpha*g * dlogits . lut */
int grl_oim_grad(const float* dlogits, int64_t ldd, const float* lut, const float* g, float alpha,
                 float* dx, int n, int c, int D, void* stream);
/* TripletLoss('soft', batch_hard=True).forward, mode 'id', dis_func 'eu' (reid/loss/triplet.py:16-76,
 * cdist :78-90): dist[i][j] = sqrt(sum_k (f_i-f_j)^2 + 1e-12); z_i = max_j dist*[same id, j != i]
 * - min_j (dist + 1e5*[same id]); loss[i] = soft ? log(1 + exp(z_i)) : max(z_i + margin, 0).
 * sel[i] = {arg-max or -1 when the maximum is a masked zero, arg-min} (first index on ties) for
 * the backward. */
int grl_triplet_fwd(const float* feat, const int64_t* ids, int n, int D, int soft, float margin,
                    float* loss, float* dist, int32_t* sel, float* z, void* stream);
int grl_triplet_bwd(const float* feat, const float* dist, const int32_t* sel, const float* z,
                    const float* dloss, int soft, float margin, float* dfeat, int n, int D,
                    void* stream);
/* prob[t] = softmax(scores[t][0..1])[1] (reid/train/trainer.py:146-148; prob0, optional, keeps the
 * class-0 probability for the backward) and its backward, term for term as torch's softmax backward */
int grl_softmax2(const float* scores, float* prob, float* prob0, int64_t m, void* stream);
int grl_softmax2_bwd(const float* prob, const float* prob0, const float* dprob, float* dscores, int64_t m,
                     void* stream);
/* PairLoss.forward (reid/loss/pairloss.py:18-45): label[a][b] = [tar_probe[b] == tar_gallery[a]];
 * loss[0] = BCE(prob, label) (mean, logs clamped at -100); prec[0] (may be NULL) = top-1 precision of
 * the (1-s, s) pseudo-logits; dprob (may be NULL) = d loss[0] / d prob. */
int grl_pair_bce(const float* prob, const int64_t* tar_probe, const int64_t* tar_gallery, int n,
                 float* loss, float* prec, float* dprob, void* stream);
/* y = alpha * g[0] * x with g a device scalar (chains an upstream loss gradient without a host sync) */
int grl_scale_dev(const float* x, const float* g, float alpha, float* y, int64_t n, void* stream);

/* per-query ranking metrics of eva_functions.evaluate (reid/evaluator/eva_functions.py:134-184)
 * over a row-wise argsort (grl_row_argsort): gallery entries with the query's pid AND camera are
 * dropped; first_hit[q] = 0-based rank of the first match among the kept entries (-1: the
 * identity never appears, the query is skipped upstream), n_hits[q] = matches kept,
 * ap[q] = (1/n_hits) sum over hits of (hits so far)/(rank+1) in fp64.  CMC[r] = mean over valid
 * queries of [first_hit <= r]; mAP = mean of ap over valid queries (host, nq numbers). */
int grl_rank_metrics(const int32_t* idx, int64_t ld, const int32_t* q_pids, const int32_t* q_cams,
                     const int32_t* g_pids, const int32_t* g_cams, int nq, int ng, int32_t* first_hit,
                     int32_t* n_hits, double* ap, void* stream);

/* ---- k-reciprocal re-ranking on the device (reid/evaluator/rerank.py:37-104) ----
 * N = nq + ng samples (<= 16384).  All matrices fp32 row-major, caller-owned:
 *   D [N][N], rank int32 [N][N] (grl_row_argsort of D), V / V2T [N][N] and V2q [nq][N]
 *   ZERO-FILLED by the caller, lcnt int32 [N], lidx int32 [N][256]. */
/* :41-47  D[i][j] = S[j][i] / max_r S[r][i],  S = [[q_q, q_g], [q_g^T, g_g]] squared */
int grl_rerank_build(const float* q_g, const float* q_q, const float* g_g, int nq, int ng, float* D,
                     float* colmax_ws /* N floats */, void* stream);
/* :55-75  k-reciprocal set of every sample (k1 <= 20), its 2/3-overlap expansion by the
 * int(around(k1/2))-reciprocal sets, V[i][e] = exp(-D[i][e]) / sum (np.sum's pairwise order);
 * lidx[i][0..lcnt[i]) = the sorted unique expansion indices */
int grl_rerank_krecip(const float* D, const int32_t* rank, int N, int k1, float* V, int32_t* lcnt,
                      int32_t* lidx, void* stream);
/* :77-83  local query expansion V2[i] = mean_{t<k2} V[rank[i][t]] (k2 <= 8; k2 == 1: V itself),
 * stored transposed V2T[e][i] and, for i < nq, as dense rows V2q[i][e] */
int grl_rerank_expand(const float* V, const int32_t* rank, const int32_t* lcnt, const int32_t* lidx,
                      int N, int nq, int k2, float* V2T, float* V2q, void* stream);
/* :86-104 out[i][j-nq] = (1-lambda) * (1 - t/(2-t)) + lambda * D[i][j],
 * t = sum over the non-zero k of V2[i] (ascending) of min(V2[i][k], V2[j][k]);  out [nq][ng] */
int grl_rerank_jaccard(const float* V2q, const float* V2T, const float* D, int N, int nq,
                       float lambda_value, float one_minus_lambda, float* out, void* stream);

/* All weight re-layouts (and bf16 casts) of one training step in one launch: a table of gathers
 * dst[j] = src[base + i0*strides[0] + i1*strides[1] + i2*strides[2] + i3*strides[3]], j = ((i0*dims[1] + i1)*dims[2] +
 * i2)*dims[3] + i3, fp32 in, fp32 or bf16 out; `tiled` = a 2-D transpose (dims[0] = dims[1] = 1, strides[2] = 1)
 * through LDS tiles.  The table is DEVICE memory (built once per model by the host); it replaces, per step, the
 * grl_pack_conv_weight / grl_transpose / grl_pack_dgrad_weight / grl_cast_bf16 launches of
 * resnets1.py:62-68's and grl_model.py:95-121's weights. */
typedef struct GrlPrepEntry {
    const float* src;
    void*        dst;
    int64_t      base;
    int64_t      strides[4];
    int32_t      dims[4];
    int32_t      tiled, out_bf16;
} GrlPrepEntry;
int grl_weight_prep(const GrlPrepEntry* table_dev, int count, void* stream);

/* ---- bf16-STORAGE training (train_engine.set_math('bf16s'); BASELINE configs[2] as a training batch) -------------
 * The twins of the train-mode kernels above for bf16 activations / saved tensors / activation gradients in HBM
 * (`void*` = bf16 tensor); statistics, per-channel vectors and parameter gradients stay fp32.  Same reference call
 * sites as the fp32 entry points they mirror (resnets1.py:76-91, basebranch.py:38-66, grl_model.py:71-83,131-180
 * and their autograd backward, trainer.py:54).  C % 8 == 0, 16-byte aligned tensors. */
int grl_bn_apply_centered_bf16(const void* z, const float* mean, const float* scale, const float* beta,
                               const void* res, void* y, int64_t M, int C, int relu,
                               uint8_t* relu_bits /* may be NULL: M*C/8 bytes, one bit per stored output */, void* stream);
/* pivot: an fp32 VECTOR [C] (or NULL), not a row of x as in grl_col_stats */
int grl_col_stats_bf16(const void* x, float* slab, int M, int C, int ld, const float* pivot, void* stream);
int grl_bn_bwd_bf16(const void* dy, const void* z, const void* act, const float* mean, const float* invstd,
                    const float* gamma, void* dz, float* dgamma, float* dbeta, float* slab_ws, float* coef_ws,
                    int M, int C, void* gres, int gres_accumulate, const float* mask_scale,
                    const float* mask_beta, const uint8_t* relu_bits, void* stream);
int grl_relu_bwd_bf16(const void* dy, const void* act, void* out, int64_t n, int accumulate, void* stream);
int grl_axpby_bf16(const void* a, const void* b, void* y, float alpha, float beta, int64_t n, void* stream);
int grl_axpy_strided_bf16(void* dst, int64_t dst_stride, const void* src, int64_t src_stride, int nb,
                          int64_t inner, float alpha, int accumulate, void* stream);
int grl_dilate2_bf16(const void* dz, void* up, int n, int Ho, int Wo, int H, int W, int C, int accumulate,
                     int oy_off, int ox_off, void* stream);
int grl_maxpool3x3s2_bwd_bf16(const void* x, const void* dy, void* dx, int n, int H, int W, int C, void* stream);
int grl_bn_relu_maxpool3x3s2_bf16(const void* z, const float* mean, const float* scale, const float* beta, void* y,
                                  uint8_t* idx, int n, int H, int W, int C, void* stream);
int grl_maxpool3x3s2_bwd_idx_bf16(const uint8_t* idx, const void* dy, void* dx, int n, int H, int W, int C, void* stream);
/* fp32 clip in, bf16 im2col columns out (the stem's weight gradient) */
int grl_stem_im2col_bf16(const float* x, void* col, int n, int H, int W, int Kp, void* stream);
int grl_gate_apply_bf16(const void* y, int ldy, const void* x, float* cmap, void* xc, void* xu, int M, int C,
                        void* stream);
int grl_gate_bwd_bf16(const void* dxc, const void* dxu, const void* x, const float* cmap, void* dx,
                      int accumulate, void* dy, int ldy, int M, int C, void* stream);
/* v: fp32 per-group vectors, or (v_is_bf16) a bf16 tensor -- the temporal-mean backward */
int grl_add_rowbcast_bf16(void* dst, const void* v, int64_t M, int64_t C, int64_t rows_per_group, float scale,
                          int accumulate, int v_is_bf16, void* stream);
int grl_sqdiff_bwd_bf16(const void* f1, const void* f2, const float* dd, void* df1, void* df2, int b, int rows,
                        int C, int64_t f2_clip_stride, int accumulate_df2, void* stream);
/* bf16 -> fp32 (n % 8 == 0) */
int grl_cast_f32(const void* x, float* y, int64_t n, void* stream);

/* ---- cross-layer fusion of the bf16-storage trunk (round 4, fuse_bf16.hip) -------------------------------------
 * END of one ResNet bottleneck + START of the next in one launch (layers 1-2, where both 1x1 convolutions are
 * HBM-bound):   y = relu(bn3(conv3(t2)) + res)         reid/models/resnets1.py:86-91
 *               u = relu(bn1'(conv1'(y)))               reid/models/resnets1.py:76-78 of the next block
 * y (the widest tensor of the block) is written once and never re-read; a pixel's 4P outputs are contracted against
 * conv1' in the registers of the wave that produced them.  All activations / weights bf16 (row-major, channels
 * last), per-channel vectors fp32 (eval-folded BatchNorm: grl_bn_fold), fp32 accumulate, epilogues term for term
 * those of grl_conv_gemm_f32's bf16-storage datapath.  w1n must be in the chained k order: grl_bneck_perm32.
 * Shapes: (P, C4) = (64, 256) with Pn in {0, 64, 128}; (128, 512) with Pn in {0, 128, 256}; Pn = 0: no chain. */
typedef struct GrlBneckTail {
    const void*  t2;        /* [M][P]  bf16: conv2's output (after bn2 + ReLU)              */
    const void*  w3;        /* [C4][P] bf16: conv3 weight                                   */
    const float* scale3;    /* [C4] or NULL (= 1)                                           */
    const float* shift3;    /* [C4] or NULL (= 0)                                           */
    const void*  res;       /* [M][C4] bf16: the block's input (or its downsample branch)   */
    void*        y;         /* [M][C4] bf16 out                                             */
    const void*  w1n;       /* [Pn][C4] bf16, k-permuted (grl_bneck_perm32), or NULL        */
    const float* scale1n;   /* [Pn] or NULL                                                 */
    const float* shift1n;   /* [Pn] or NULL                                                 */
    void*        u;         /* [M][Pn] bf16 out, or NULL                                    */
    int32_t M, P, C4, Pn;
    /* Kd > 0: the residual is the block's DOWNSAMPLE branch computed in the same launch (resnets1.py:83-84, the first
     * block of a layer): res = bf16(scaled * (wd . x0) + shiftd); `res` is ignored.  (P, C4, Pn, Kd) = (64, 256, 64, 64). */
    const void*  x0;        /* [M][Kd] bf16: the block's input                              */
    const void*  wd;        /* [C4][Kd] bf16: downsample conv weight                        */
    const float* scaled;    /* [C4] or NULL                                                 */
    const float* shiftd;    /* [C4] or NULL                                                 */
    int32_t Kd, reserved;
} GrlBneckTail;
int grl_bottleneck_tail_bf16(const GrlBneckTail* desc, void* stream);
int grl_bottleneck_tail_bf16_supported(int P, int C4, int Pn);      /* 1 if the shape has a kernel */
/* w [Pn][C4] (fp32 if !w_is_bf16) -> out [Pn][C4] bf16 in the k order the chained MFMA of grl_bottleneck_tail_bf16 consumes */
int grl_bneck_perm32(const void* w, int w_is_bf16, void* out, int Pn, int C4, void* stream);

/* Stem + max-pool in ONE launch (eval, bf16 storage; resnets1.py:101-104: conv 7x7/s2 + folded bn1 + ReLU + MaxPool2d(3, 2, 1)):
 * x [n][3][H][W] fp32 (or raw uint8 with mean_std, normalised while the patch is staged), W == 128, H % 4 == 0;
 * y [n*(H/4)*(W/4)][64] bf16 = the POOLED map; the stem map itself never reaches HBM.  wp: grl_stem_pack_weight_bf16. */
int grl_stem_pool_bf16(const void* x, int x_is_u8, const float* mean_std, const float* scale, const float* shift, void* y,
                       int n, int H, int W, const void* wp, void* stream);

/* Layer 1's 3x3 / stride 1 convolution (64 -> 64 channels, maps W == 32 wide, H % 8 == 0; resnets1.py:79-81) + folded
 * BatchNorm + optional ReLU, bf16 storage: weights LDS-resident, every input pixel staged once per tile (fuse_bf16.hip).
 * x [n_img][H][W][64] bf16, w [64][9*64] bf16 packed tap-major (grl_pack_conv_weight + grl_cast_bf16), y [n_img*H*W][64]. */
int grl_conv3x3_c64_bf16(const void* x, const void* w, const float* scale, const float* shift, void* y, int n_img, int H, int W,
                         int relu, void* stream);

/* The same fusion for the EXACT-fp32 trunk (BASELINE configs[1]; fuse_f32.hip): all operands fp32.  Results are BIT-IDENTICAL
 * to grl_conv_gemm_f32 (GRL_MATH_F32, one chain) run on conv3 (+res, ReLU) and then on conv1' -- the transposed MFMA
 * keeps that kernel's documented k-ordered fmaf chain.  w1n is the plain [Pn][C4] weight (no permutation).
 * Shapes: (P, C4) = (64, 256) with Pn in {0, 64, 128}; (128, 512) with Pn in {0, 128}. */
typedef struct GrlBneckTailF32 {
    const float* t2;        /* [M][P]                                                       */
    const float* w3;        /* [C4][P]                                                      */
    const float* scale3;    /* [C4] or NULL (= 1)                                           */
    const float* shift3;    /* [C4] or NULL (= 0)                                           */
    const float* res;       /* [M][C4]                                                      */
    float*       y;         /* [M][C4] out                                                  */
    const float* w1n;       /* [Pn][C4] or NULL                                             */
    const float* scale1n;   /* [Pn] or NULL                                                 */
    const float* shift1n;   /* [Pn] or NULL                                                 */
    float*       u;         /* [M][Pn] out, or NULL                                         */
    int32_t M, P, C4, Pn;
} GrlBneckTailF32;
int grl_bottleneck_tail_f32(const GrlBneckTailF32* desc, void* stream);
int grl_bottleneck_tail_f32_supported(int P, int C4, int Pn);

/* ------------------------------------------------------------------------------------------------------------------
 * Frame decode on the device (round 6).  Replaces the per-frame
 * `Image.open(img_path).convert('RGB')` of reid/data/video_loader.py:124-141 (and :91-96, :108-113): baseline JPEG
 * -> uint8 RGB, BIT-IDENTICAL to Pillow / libjpeg-turbo with libjpeg's defaults (ISLOW integer IDCT, fancy
 * upsampling): 0xFF00 unstuffing, Huffman decoding (one 256-lane workgroup per frame over self-synchronising
 * subsequences of the bit stream; one lane per frame for scans with restart intervals or frames too large for the
 * workgroup form), dequantisation + jidctint's integer IDCT (one lane per 8 x 8 block), triangle-filter chroma
 * upsampling + YCbCr -> RGB (one lane per pixel).
 * Scope: 8-bit baseline / extended-sequential Huffman, one interleaved scan, 1 or 3 components, luma sampling 1x1,
 * 2x1 or 2x2 (4:4:4, 4:2:2, 4:2:0 -- MARS' frames are 256 x 128 4:2:0), table ids 0..1, restart intervals.
 * Anything else is refused by the parser with GRL_EUNSUPPORTED: the caller (grl_amd/reid/data/jpeg.py) says so
 * loudly -- it does not decode on the host behind the caller's back.
 */
#define GRL_EUNSUPPORTED -3  /* a valid JPEG outside the scope above (progressive, arithmetic, CMYK, 12-bit, 4:4:0 ...) */

typedef struct GrlJpegFrame {      /* one parsed frame: filled on the host by grl_jpeg_parse, read by the kernels */
    uint32_t scan_off;             /* entropy-coded segment: offset into the batch's byte buffer, length */
    uint32_t scan_len;
    uint16_t width, height;
    uint16_t restart_interval;     /* MCUs between RSTn markers, 0 = none */
    uint8_t  ncomp, hmax, vmax, rgb;   /* rgb: components are R, G, B (Adobe transform 0): no colour conversion */
    uint8_t  hs[4], vs[4];         /* sampling factors per component */
    uint8_t  tq[4], td[4], ta[4];  /* quantisation / DC / AC table of each component */
    uint16_t tabset;               /* index of this frame's Huffman table set inside its batch (grl_jpeg_assign_tables) */
    uint8_t  pad_[8];              /* (q starts at byte 48; sizeof == 2160 == 16 * 135: 16-byte loads of q rows) */
    uint16_t q[4][64];             /* quantisation tables, natural (row-major) order */
    int32_t  maxcode[4][18];       /* Huffman tables [DC0, DC1, AC0, AC1]: largest code of each length (-1: none) */
    int32_t  valoff[4][18];        /*   symbol index = code + valoff[length] */
    uint8_t  vals[4][256];         /*   symbols in code order */
} GrlJpegFrame;

/* HOST function (no GPU call): parse the headers of ONE JPEG stream `data[0..len)` that will sit at byte `base_off` of
 * the batch buffer.  Returns GRL_OK, GRL_EINVAL (not a JPEG / truncated) or GRL_EUNSUPPORTED. */
int grl_jpeg_parse(const uint8_t* data, int64_t len, int64_t base_off, GrlJpegFrame* out);
/* test / A-B hook: 1 (default) = entropy decoding by one 256-lane workgroup per frame (self-synchronising subsequences),
 * 0 = one lane per frame; -1 = query.  Same coefficients either way.  Returns the previous setting. */
int grl_jpeg_parallel_mode(int on);
/* HOST function: number the Huffman table sets of a batch (frames[i].tabset).  Up to 4 distinct sets get shared
 * look-ahead tables in LDS (frames of one encoder share ONE set); beyond that every frame keeps its own.  Returns
 * the number of sets (> 0) or a negative GRL_E* code.  Call after grl_jpeg_parse, before the descriptors are copied. */
int grl_jpeg_assign_tables(GrlJpegFrame* frames, int n);
/* HOST function: grl_jpeg_parse for the n streams of one batch buffer (stream i = buf[offsets[i] .. offsets[i+1])), then
 * grl_jpeg_assign_tables.  On failure *bad_index is the offending frame and the return value its code. */
int grl_jpeg_parse_batch(const uint8_t* buf, const int64_t* offsets, int n, GrlJpegFrame* frames, int* bad_index);
/* bytes of device scratch grl_jpeg_decode_batch needs for the n parsed frames of a batch (coefficients, planes,
 * look-ahead tables, unstuffed streams) */
int64_t grl_jpeg_workspace_bytes(const GrlJpegFrame* frames_host, int n);
/* n frames of ONE geometry (width, height, components, sampling: as frames[0]; frames_host is checked) ->
 * out uint8 [n][3][height][width] (planar RGB: the layout the clip tensors [B][T][3][H][W] have).
 * bytes: the concatenated streams (device) -- the buffer grl_jpeg_parse's base_off values refer to: it must hold
 * every frames[i].scan_off + scan_len (like every pointer of this ABI its extent is the caller's contract; the kernels
 * read no byte outside the scans); frames_dev: the n parsed descriptors (device copy of frames_host). */
int grl_jpeg_decode_batch(const uint8_t* bytes, const GrlJpegFrame* frames_dev, const GrlJpegFrame* frames_host, int n,
                          uint8_t* out, void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GRL_HIP_H */
