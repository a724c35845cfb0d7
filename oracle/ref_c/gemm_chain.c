/* ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Bit-exact CPU model of grl_conv_gemm_f32's accumulation: every output element is
 * ONE fp32 accumulator updated by a chain of fused multiply-adds in the kernel's k
 * order (grl_amd/csrc/gemm_f32.hip header): for each 32-wide K stage j, chunk
 * q = 0..3, step s = 0..3:  k0 = 32j + 8q + s, then k1 = k0 + 4.  With kblock (GrlGemm.kblock,
 * the train-mode forward): every 512 k (16 stages) that is not the end of K the accumulator is
 * added to a running total and restarts at zero; the result is last_segment + total.
 * v_mfma_f32_32x32x2_f32 is bitwise a k-ordered fmaf chain (CDNA4 guide, "FP32-input
 * MFMA"), so this reproduces the GPU distance matrix of
 * reid/evaluator/attevaluator.py:33-46 (-q.g^T and the Euclidean form) bit for bit,
 * which makes "ranking indices bit-exact" checkable at any size.
 *
 * Build: gcc -O2 -mfma -ffp-contract=off -fopenmp -shared -fPIC (see oracle/build.py)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SEG_K 512   /* gemm_f32.hip: SEG_STAGES * 32 */

static void k_order(int K, int* order) {
    int n = 0;
    for (int j = 0; j < K / 32; ++j)
        for (int q = 0; q < 4; ++q)
            for (int s = 0; s < 4; ++s) {
                order[n++] = 32 * j + 8 * q + s;
                order[n++] = 32 * j + 8 * q + s + 4;
            }
}

/* y[m][n] = chain_k a[m][k] * w[n][k];  mode 0: acc, 1: -acc,
 * 2: sqrtf(max(rn[m] + cn[n] - 2*acc, 1e-12)) with rn/cn given. */
int grl_oracle_chain_gemm(const float* a, const float* w, float* y, int M, int N, int K,
                          int lda, int ldw, int ldy, int mode, const float* rn, const float* cn, int kblock) {
    if (K % 32) return -1;
    int* order = (int*)malloc(sizeof(int) * K);
    float* wt = (float*)malloc(sizeof(float) * (size_t)K * N);      /* [k'][n], k' in chain order */
    if (!order || !wt) return -2;
    k_order(K, order);
    for (int kk = 0; kk < K; ++kk)
        for (int n = 0; n < N; ++n) wt[(size_t)kk * N + n] = w[(size_t)n * ldw + order[kk]];
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        float* acc = y + (size_t)m * ldy;
        for (int n = 0; n < N; ++n) acc[n] = 0.f;
        float* tot = (float*)calloc((size_t)N, sizeof(float));
        for (int kk = 0; kk < K; ++kk) {
            const float av = a[(size_t)m * lda + order[kk]];
            const float* wr = wt + (size_t)kk * N;
            for (int n = 0; n < N; ++n) acc[n] = fmaf(av, wr[n], acc[n]);
            if (kblock && (kk + 1) % SEG_K == 0 && kk + 1 < K)
                for (int n = 0; n < N; ++n) { tot[n] = tot[n] + acc[n]; acc[n] = 0.f; }
        }
        if (kblock && K > SEG_K)
            for (int n = 0; n < N; ++n) acc[n] = acc[n] + tot[n];
        free(tot);
        if (mode == 1) {
            for (int n = 0; n < N; ++n) acc[n] = -acc[n];
        } else if (mode == 2) {
            for (int n = 0; n < N; ++n) {
                float v = rn[m] + cn[n] - 2.f * acc[n];
                acc[n] = sqrtf(v > 1e-12f ? v : 1e-12f);
            }
        }
    }
    free(order);
    free(wt);
    return 0;
}

/* |x_row|^2 the way grl_row_sqnorm reduces it is NOT modelled here (tree order);
 * tests feed the GPU's own row norms to mode 2. */
