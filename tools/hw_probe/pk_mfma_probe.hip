// Probe: does a VALU kernel change its results when bf16-MFMA waves of ANOTHER kernel share its SIMDs?
// Two streams: an MFMA spinner (no memory traffic) and variants of a small dot-product kernel.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off pk_mfma_probe.hip -o pk_mfma_probe && ./pk_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/grl_hip.h"
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int KIND>   // 0: bf16 32x32x16, 1: f32 32x32x2, 2: VALU fma spin
__global__ __launch_bounds__(256) void spinner(float* out, int iters) {
    f32x16 acc = {0};
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(1.0f + e * 0.01f); }
    float fa = threadIdx.x * 0.001f, fb = 1.0001f;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        } else if (KIND == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 64; ++u) fa = fa * fb + 0.5f;
        }
    }
    float s = fa;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) out[0] = s;
}

// victim: y[b][c..c+3] = sum_j w[j][c..c+3] * hs[j], hs staged in LDS (the channel_atte_out loop)
__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }
template <int VAR>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ hid, const float* __restrict__ w2t,
                                              float* __restrict__ y, int C, int Hd) {
    extern __shared__ __attribute__((aligned(16))) float hs[];
    const int b = blockIdx.y;
    for (int j = threadIdx.x; j < Hd; j += 256) hs[j] = hid[(int64_t)b * Hd + j];
    __syncthreads();
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= C) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (VAR == 0) {               // as compiled in the product (packed mul/add expected)
        for (int j = 0; j < Hd; ++j) s += *reinterpret_cast<const f32x4*>(w2t + (int64_t)j * C + c) * hs[j];
    } else if (VAR == 1) {        // scalar ops forced (asm barriers break SLP vectorisation)
        for (int j = 0; j < Hd; ++j) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(w2t + (int64_t)j * C + c);
            const float h = hs[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) { float p = w[e] * h; asm volatile("" : "+v"(p)); s[e] += p; }
        }
    } else {                      // packed, but hs[j] from a register copy made through readfirstlane (SGPR operand)
        for (int j = 0; j < Hd; ++j) {
            const float h = __builtin_amdgcn_readfirstlane(hs[j]);
            s += *reinterpret_cast<const f32x4*>(w2t + (int64_t)j * C + c) * h;
        }
    }
    if (VAR == 3 || VAR == 4) {   // the product kernel's tail: sigmoid per component (v_exp / v_rcp: TRANS ops)
        for (int j = 0; j < Hd; ++j) s += *reinterpret_cast<const f32x4*>(w2t + (int64_t)j * C + c) * hs[j];
        f32x4 a;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = sigmoidf_(s[e]);
            if (VAR == 4) asm volatile("s_nop 7\n s_nop 7" : "+v"(a[e]));
        }
        s = a;
    }
    *reinterpret_cast<f32x4*>(y + (int64_t)b * C + c) = s;
}

int main() {
    const int B = 32, C = 2048, Hd = 128;
    std::vector<float> h_hid(B * Hd), h_w((size_t)Hd * C);
    srand(1);
    for (auto& v : h_hid) v = (rand() % 2000) / 1000.f;
    for (auto& v : h_w) v = ((rand() % 2000) - 1000) / 20000.f;
    float *hid, *w, *y, *yref, *sp;
    CK(hipMalloc(&hid, h_hid.size() * 4)); CK(hipMalloc(&w, h_w.size() * 4));
    CK(hipMalloc(&y, (size_t)B * C * 4)); CK(hipMalloc(&yref, (size_t)B * C * 4)); CK(hipMalloc(&sp, 64));
    CK(hipMemcpy(hid, h_hid.data(), h_hid.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, h_w.data(), h_w.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    std::vector<float> ref((size_t)B * C), got((size_t)B * C);
    const char* vname[5] = {"packed (as shipped)", "scalar mul/add", "packed, SGPR h", "packed + sigmoid tail", "sigmoid tail + nops"};
    const char* aname[5] = {"bf16 MFMA spinner", "f32 MFMA spinner", "VALU spinner", "nothing", "bf16s GEMM M=4096"};
    void *ga, *gw, *gy; const int GM = 4096, GN = 2048, GK = 2048;
    CK(hipMalloc(&ga, (size_t)GM * GK * 2)); CK(hipMalloc(&gw, (size_t)GN * GK * 2)); CK(hipMalloc(&gy, (size_t)GM * GN * 2));
    CK(hipMemset(ga, 0x3c, (size_t)GM * GK * 2)); CK(hipMemset(gw, 0x3c, (size_t)GN * GK * 2));
    GrlGemm gd; memset(&gd, 0, sizeof gd); gd.a = (const float*)ga; gd.w = (const float*)gw; gd.y = (float*)gy;
    gd.M = GM; gd.N = GN; gd.K = GK; gd.lda = GK; gd.ldw = GK; gd.ldy = GN; gd.ldres = GN; gd.relu = 1; gd.math = GRL_MATH_BF16S;
    for (int var = 0; var < 5; ++var) {
        auto run_v = [&](float* out, hipStream_t s) {
            dim3 g((C + 1023) / 1024, B);
            if (var == 0) hipLaunchKernelGGL(victim<0>, g, dim3(256), Hd * 4, s, hid, w, out, C, Hd);
            if (var == 1) hipLaunchKernelGGL(victim<1>, g, dim3(256), Hd * 4, s, hid, w, out, C, Hd);
            if (var == 2) hipLaunchKernelGGL(victim<2>, g, dim3(256), Hd * 4, s, hid, w, out, C, Hd);
            if (var == 3) hipLaunchKernelGGL(victim<3>, g, dim3(256), Hd * 4, s, hid, w, out, C, Hd);
            if (var == 4) hipLaunchKernelGGL(victim<4>, g, dim3(256), Hd * 4, s, hid, w, out, C, Hd);
        };
        run_v(yref, s2); CK(hipDeviceSynchronize());
        CK(hipMemcpy(ref.data(), yref, ref.size() * 4, hipMemcpyDeviceToHost));
        for (int ag = 0; ag < 5; ++ag) {
            int bad = 0, worst_cnt = 0;
            for (int it = 0; it < 40; ++it) {
                CK(hipMemsetAsync(y, 0xff, (size_t)B * C * 4, s2));
                CK(hipDeviceSynchronize());
                if (ag == 0) hipLaunchKernelGGL(spinner<0>, dim3(512), dim3(256), 0, s1, sp, 20000);
                if (ag == 1) hipLaunchKernelGGL(spinner<1>, dim3(512), dim3(256), 0, s1, sp, 5000);
                if (ag == 2) hipLaunchKernelGGL(spinner<2>, dim3(1024), dim3(256), 0, s1, sp, 20000);
                if (ag == 4) { if (grl_conv_gemm_f32(&gd, s1) != 0) { printf("gemm failed\n"); return 1; } }
                for (int rep = 0; rep < 60; ++rep) run_v(y, s2);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(got.data(), y, got.size() * 4, hipMemcpyDeviceToHost));
                int cnt = 0;
                for (size_t i = 0; i < got.size(); ++i) cnt += memcmp(&got[i], &ref[i], 4) != 0;
                bad += cnt != 0; if (cnt > worst_cnt) worst_cnt = cnt;
            }
            printf("victim %-22s under %-18s: bad runs %2d/40 (max differing elements %d)\n", vname[var], aname[ag], bad, worst_cnt);
        }
    }
    return 0;
}
