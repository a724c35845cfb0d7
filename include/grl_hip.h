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
int grl_abi_version(void);

/* epilogue selector of grl_conv_gemm_f32 */
#define GRL_EPI_AFFINE  0   /* y = relu?( rs[m]*(acc + gbias[m/rpg][n])*scale[n] + shift[n] + res[m][n] ) */
#define GRL_EPI_NEGDOT  1   /* y = -acc                                     (attevaluator.py:44-46)  */
#define GRL_EPI_EUCLID  2   /* y = sqrt(max(rnorm[m]+cnorm[n]-2acc,1e-12))  (attevaluator.py:33-41)  */

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
} GrlGemm;

int grl_conv_gemm_f32(const GrlGemm* desc, void* stream);
/* rows of the stats slab the call above writes (= number of M tiles it will use) */
int grl_conv_gemm_f32_stat_rows(const GrlGemm* desc);

/* [N][C][kh][kw] (torch layout) -> [N][kh*kw][C]; replaces nothing in the
 * reference (layout packing for the implicit GEMM). */
int grl_pack_conv_weight(const float* w, float* out, int N, int C, int kh, int kw, void* stream);

/* eval-mode BatchNorm folding: scale = g/sqrt(var+eps), shift = b - mean*scale
 * (+ scale*bias when the producing layer has a bias).  nn.BatchNorm{1,2}d in eval
 * mode everywhere on the path (e.g. resnets1.py:77,81,85). Any of gamma/beta NULL
 * means 1/0; mean/var NULL means 0/1 (plain bias -> shift). */
int grl_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                const float* bias, float eps, float* scale, float* shift, int C, void* stream);

/* Stem: 7x7 stride-2 pad-3 conv on NCHW input [n][3][H][W] + folded BN + ReLU ->
 * channels-last [n][H/2][W/2][64]   (resnets1.py:101-103 / basebranch.py:28-30). */
int grl_stem_conv7x7(const float* x, const float* w /*[64][3][7][7]*/, const float* scale,
                     const float* shift, float* y, int n, int H, int W, void* stream);

/* 3x3 stride-2 pad-1 max pool, channels-last (resnets1.py:104). */
int grl_maxpool3x3s2(const float* x, float* y, int n, int H, int W, int C, void* stream);

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

/* |x_row|^2 for the Euclidean epilogue (attevaluator.py:37-38). */
int grl_row_sqnorm(const float* x, float* out, int rows, int K, int ld, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GRL_HIP_H */
