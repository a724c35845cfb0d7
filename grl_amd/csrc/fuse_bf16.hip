// Cross-layer fusion for the HBM-bound half of the bf16-storage trunk (BASELINE configs[2]) on gfx950 (MI355X).
//
//   grl_bottleneck_tail_bf16:  one launch for the END of a ResNet bottleneck and the START of the next one
//
//       y = relu( bn3(conv3(t2)) + res )          (reid/models/resnets1.py:86-91: 1x1 expansion P -> 4P, residual, ReLU)
//       u = relu( bn1'(conv1'(y)) )                (resnets1.py:76-78 of the NEXT block: 1x1 reduction 4P -> P')
//
// Unfused, conv3 writes y (the widest tensor of layers 1-2: 537 MB per launch at 64 x 8 frames) and the next block's
// conv1 reads it straight back -- both launches sit at the HBM roof at < 25 % MFMA busy.  Here a pixel's 4P outputs
// never leave the wave that computed them before they have also been contracted against conv1': y is written once
// (the next block still needs it as its residual), never re-read, and one launch boundary per block goes.
//
// Design (MI355X-first; nothing here is a tiled GEMM):
//   * the contraction is tiny (K = P <= 128 and K = 4P <= 512) and the launch is HBM-bound at ~20 % MFMA duty, so the
//     kernel is built for memory-level parallelism, not operand reuse: 16 waves per workgroup, ONE workgroup per CU,
//     every wave owns 16 pixels and walks ALL output channels of them -- waves never exchange data;
//   * transposed MFMA (v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A operand, pixels as columns): a lane then
//     holds 4 consecutive channels of one pixel, which is (a) the bf16 B-operand fragment of the chained conv1' for
//     free -- two 16-channel blocks of y are one k-step, with the k order of conv1' permuted to match
//     (weights pre-permuted on the host, `grl_bneck_perm32`) -- and (b) two v_permlane16_swap away from 16 contiguous
//     bytes per lane / 64 contiguous bytes per pixel for the residual load and the y store;
//   * weights live in LDS in FRAGMENT order (1 KiB per (16-channel block, k-step): lane l's 16 bytes at 16*l, filled
//     by LDS-DMA with the per-lane gather as the DMA source), so an A fragment is one conflict-free ds_read_b128.
//     Layer 1 (64 KB / 96 KB of weights) keeps them resident for the whole persistent workgroup: no barrier in steady
//     state.  Layer 2 (256-384 KB) streams them in channel chunks through two LDS buffers, the next chunk's DMA in
//     flight under the current chunk's MFMAs;
//   * per-channel scale / shift vectors sit in LDS (a lane needs a different float4 per block).
//
// Numerics: fp32 accumulate; epilogues term for term those of the unfused GEMM kernels (acc*scale + shift (+ res),
// ReLU, round-to-nearest-even bf16); conv1' consumes the bf16-rounded y, exactly what the unfused pipeline feeds it.
// A pixel's results do not depend on the tile, the batch or the grid (each wave owns whole pixels).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grl_hip.h"
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// one LDS-DMA wave-instruction: lane l's 16 bytes at `g` land at lds + 16*l
__device__ __forceinline__ void glds16(const char* g, char* lds) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds, 16, 0, 0);
}

// rows (16 lanes) 1 and 3 of `a` trade places with rows 0 and 2 of `b`
__device__ __forceinline__ void swap16(uint32_t& a, uint32_t& b) {
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

__device__ __forceinline__ float bf_lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {       // round-to-nearest-even, as the GEMM epilogues
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return *reinterpret_cast<uint32_t*>(&v);
}

constexpr int TILE_PX = 256;                  // 16 waves x 16 pixels

// P: conv3 input channels, C4: its output channels (= conv1' input channels), PN: conv1' output channels (0: no chain),
// CH: channels of y per LDS weight chunk (C4 / CH chunks; one chunk = resident weights, no barriers)
template <int P, int C4, int PN, int CH, int GBMAX = 4>
__global__ __launch_bounds__(1024) void bneck_tail_kernel(const GrlBneckTail p, const int num_tiles) {
    constexpr bool CHAIN = PN > 0;
    constexpr int NCH = C4 / CH;
    constexpr int KS3 = P / 32;               // k-steps of conv3
    constexpr int CB = CH / 16;               // 16-channel blocks of y per chunk
    constexpr int GB = CB < GBMAX ? CB : GBMAX; // blocks per register group (16 * GB channels of accumulators live)
    constexpr int OB = CHAIN ? PN / 16 : 0;   // 16-channel blocks of u
    constexpr int KS1 = CH / 32;              // conv1' k-steps per chunk
    constexpr int W3_FR = CB * KS3, W1_FR = OB * KS1;
    constexpr int BUF = (W3_FR + W1_FR) * 1024;
    constexpr int VEC = (2 * C4 + 2 * (CHAIN ? PN : 0)) * 4;
    static_assert(P % 32 == 0 && CH % 32 == 0 && C4 % CH == 0 && CB % GB == 0 && GB % 2 == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const sc3 = reinterpret_cast<float*>(smem);
    float* const sh3 = sc3 + C4;
    float* const sc1 = sh3 + C4;
    float* const sh1 = sc1 + (CHAIN ? PN : 0);
    char* const wbuf = smem + VEC;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
    const char* const w3 = reinterpret_cast<const char*>(p.w3);
    const char* const w1 = reinterpret_cast<const char*>(p.w1n);

    // chunk c of the weights -> LDS buffer `buf`, fragment order; wave w issues fragments w, w+16, ...
    auto stage = [&](int c, int buf) {
        char* const dst = wbuf + buf * BUF;
#pragma unroll
        for (int f = wave; f < W3_FR + W1_FR; f += 16) {
            const char* src;
            if (f < W3_FR) {
                const int cb = f / KS3, s = f - cb * KS3;
                src = w3 + ((int64_t)(c * CH + 16 * cb + j) * P + 32 * s + 8 * q) * 2;
            } else {
                const int g = f - W3_FR, ob = g / KS1, s = g - ob * KS1;
                src = w1 + ((int64_t)(16 * ob + j) * C4 + c * CH + 32 * s + 8 * q) * 2;
            }
            glds16(src, dst + f * 1024);
        }
    };

    for (int i = tid; i < C4; i += 1024) {
        sc3[i] = p.scale3 ? p.scale3[i] : 1.f;
        sh3[i] = p.shift3 ? p.shift3[i] : 0.f;
    }
    if (CHAIN)
        for (int i = tid; i < PN; i += 1024) {
            sc1[i] = p.scale1n ? p.scale1n[i] : 1.f;
            sh1[i] = p.shift1n ? p.shift1n[i] : 0.f;
        }
    stage(0, 0);
    __syncthreads();                            // (drains the DMA: vmcnt(0) in front of the barrier)

    const char* const t2 = reinterpret_cast<const char*>(p.t2);
    const char* const res = reinterpret_cast<const char*>(p.res);
    char* const y = reinterpret_cast<char*>(p.y);
    char* const u = reinterpret_cast<char*>(p.u);
    const int sw_ch = 16 * (q & 1) + 8 * (q >> 1);          // channel offset of a lane's 16 bytes in a block PAIR (swapped layout)
    int it = 0;                                 // chunk iterations so far (buffer = it & 1 when streaming)

    for (int tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const int row = tile * TILE_PX + wave * 16 + j;
        const bool live = row < p.M;
        const int64_t rowc = live ? row : p.M - 1;
        bf16x8 bfr[KS3];
#pragma unroll
        for (int s = 0; s < KS3; ++s)
            bfr[s] = *reinterpret_cast<const bf16x8*>(t2 + (rowc * P + 32 * s + 8 * q) * 2);
        f32x4 acc1[CHAIN ? OB : 1];
#pragma unroll
        for (int ob = 0; ob < (CHAIN ? OB : 1); ++ob) acc1[ob] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
        for (int c = 0; c < NCH; ++c, ++it) {
            const char* wb = wbuf;
            if (NCH > 1) {
                wb += (it & 1) * BUF;
                if (it > 0) __syncthreads();    // chunk `it` has landed (issued one iteration ago); buffer (it+1)&1 is free
                const bool more = c + 1 < NCH || tile + (int)gridDim.x < num_tiles;
                if (more) stage(c + 1 < NCH ? c + 1 : 0, (it + 1) & 1);
            }
            const char* const w3f = wb;
            const char* const w1f = wb + W3_FR * 1024;
#pragma unroll 1
            for (int g = 0; g < CB / GB; ++g) {
                const int ch0 = c * CH + 16 * g * GB;              // first channel of the group
                uint4 rr[GB / 2];
#pragma unroll
                for (int t = 0; t < GB / 2; ++t)
                    rr[t] = *reinterpret_cast<const uint4*>(res + (rowc * C4 + ch0 + 32 * t + sw_ch) * 2);
                f32x4 acc3[GB];
#pragma unroll
                for (int b = 0; b < GB; ++b) acc3[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS3; ++s)
#pragma unroll
                    for (int b = 0; b < GB; ++b) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(w3f + ((g * GB + b) * KS3 + s) * 1024 + lane * 16);
                        acc3[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[s], acc3[b], 0, 0, 0);
                    }
#pragma unroll
                for (int t = 0; t < GB / 2; ++t) {
                    uint32_t ax = rr[t].x, ay = rr[t].y, bx = rr[t].z, by = rr[t].w;
                    swap16(ax, bx);             // 16 contiguous bytes per lane -> this lane's 4 channels of block 2t | of block 2t+1
                    swap16(ay, by);
                    const int cA = ch0 + 32 * t + 4 * q, cB = cA + 16;
                    const f32x4 sA = *reinterpret_cast<const f32x4*>(sc3 + cA), hA = *reinterpret_cast<const f32x4*>(sh3 + cA);
                    const f32x4 sB = *reinterpret_cast<const f32x4*>(sc3 + cB), hB = *reinterpret_cast<const f32x4*>(sh3 + cB);
                    f32x4 vA = acc3[2 * t] * sA + hA, vB = acc3[2 * t + 1] * sB + hB;
                    vA += f32x4{bf_lo(ax), bf_hi(ax), bf_lo(ay), bf_hi(ay)};
                    vB += f32x4{bf_lo(bx), bf_hi(bx), bf_lo(by), bf_hi(by)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vA[e] = vA[e] > 0.f ? vA[e] : 0.f;
                        vB[e] = vB[e] > 0.f ? vB[e] : 0.f;
                    }
                    ax = pack2(vA[0], vA[1]); ay = pack2(vA[2], vA[3]);
                    bx = pack2(vB[0], vB[1]); by = pack2(vB[2], vB[3]);
                    if (CHAIN) {                // blocks (2t, 2t+1) of y ARE k-step ks of conv1' (k order: grl_bneck_perm32)
                        uint4 kv = {ax, ay, bx, by};
                        const bf16x8 bk = *reinterpret_cast<bf16x8*>(&kv);
                        const int ks = (g * GB) / 2 + t;
#pragma unroll
                        for (int ob = 0; ob < OB; ++ob) {
                            const bf16x8 a = *reinterpret_cast<const bf16x8*>(w1f + (ob * KS1 + ks) * 1024 + lane * 16);
                            acc1[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bk, acc1[ob], 0, 0, 0);
                        }
                    }
                    swap16(ax, bx);
                    swap16(ay, by);
                    if (live) *reinterpret_cast<uint4*>(y + ((int64_t)row * C4 + ch0 + 32 * t + sw_ch) * 2) = uint4{ax, ay, bx, by};
                    __builtin_amdgcn_sched_barrier(0);      // keep the block pairs in program order: hoisted A-fragment reads spill
                }
            }
        }
        if (CHAIN) {
#pragma unroll
            for (int t = 0; t < OB / 2; ++t) {
                const int cA = 32 * t + 4 * q, cB = cA + 16;
                const f32x4 sA = *reinterpret_cast<const f32x4*>(sc1 + cA), hA = *reinterpret_cast<const f32x4*>(sh1 + cA);
                const f32x4 sB = *reinterpret_cast<const f32x4*>(sc1 + cB), hB = *reinterpret_cast<const f32x4*>(sh1 + cB);
                f32x4 vA = acc1[2 * t] * sA + hA, vB = acc1[2 * t + 1] * sB + hB;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vA[e] = vA[e] > 0.f ? vA[e] : 0.f;
                    vB[e] = vB[e] > 0.f ? vB[e] : 0.f;
                }
                uint32_t ax = pack2(vA[0], vA[1]), ay = pack2(vA[2], vA[3]);
                uint32_t bx = pack2(vB[0], vB[1]), by = pack2(vB[2], vB[3]);
                swap16(ax, bx);
                swap16(ay, by);
                if (live) *reinterpret_cast<uint4*>(u + ((int64_t)row * PN + 32 * t + sw_ch) * 2) = uint4{ax, ay, bx, by};
            }
        }
    }
}

// conv1' weights [Pn][C4] (fp32 masters or bf16) -> bf16 with the k order the chained MFMA consumes: inside every
// 32-channel group, position 8q + e (q = 0..3) holds channel 4q + e for e < 4 and 16 + 4q + (e - 4) for e >= 4 --
// the 8 channels a lane of lane-group q holds after two 16-channel blocks of the transposed conv3.
template <typename T>
__global__ void bneck_perm_kernel(const T* __restrict__ w, __bf16* __restrict__ out, int64_t total) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int pos = (int)(i & 31), qq = pos >> 3, e = pos & 7;
    const int src = 4 * qq + (e < 4 ? e : 12 + e);
    out[i] = (__bf16)(float)w[(i & ~(int64_t)31) + src];
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <int P, int C4, int PN, int CH>
int launch(const GrlBneckTail& d, hipStream_t s) {
    constexpr int NCH = C4 / CH;
    constexpr int FR = (CH / 16) * (P / 32) + (PN / 16) * (CH / 32);
    constexpr int LDS = (2 * C4 + 2 * PN) * 4 + (NCH > 1 ? 2 : 1) * FR * 1024;
    static_assert(LDS <= 160 * 1024, "LDS");
    static const bool attr = [] {
        (void)hipFuncSetAttribute((const void*)bneck_tail_kernel<P, C4, PN, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        return true;
    }();
    (void)attr;
    static const int cus = [] {
        int dev = 0, n = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n = prop.multiProcessorCount;
        return n;
    }();
    const int num_tiles = (d.M + TILE_PX - 1) / TILE_PX;
    const unsigned grid = (unsigned)(num_tiles < cus ? num_tiles : cus);
    hipLaunchKernelGGL((bneck_tail_kernel<P, C4, PN, CH>), dim3(grid), dim3(1024), LDS, s, d, num_tiles);
    return grl_check_launch("grl_bottleneck_tail_bf16");
}

}  // namespace

extern "C" int grl_bneck_perm32(const void* w, int w_is_bf16, void* out, int Pn, int C4, void* stream) {
    if (!w || !out || Pn <= 0 || C4 <= 0 || C4 % 32) return grl_fail(GRL_EINVAL, "grl_bneck_perm32: bad arguments");
    const int64_t total = (int64_t)Pn * C4;
    hipStream_t s = (hipStream_t)stream;
    if (w_is_bf16)
        hipLaunchKernelGGL(bneck_perm_kernel<__bf16>, dim3(grl_ceil_div(total, 256)), dim3(256), 0, s, (const __bf16*)w, (__bf16*)out, total);
    else
        hipLaunchKernelGGL(bneck_perm_kernel<float>, dim3(grl_ceil_div(total, 256)), dim3(256), 0, s, (const float*)w, (__bf16*)out, total);
    return grl_check_launch("grl_bneck_perm32");
}

extern "C" int grl_bottleneck_tail_bf16_supported(int P, int C4, int Pn) {
    return (P == 64 && C4 == 256 && (Pn == 64 || Pn == 128 || Pn == 0)) || (P == 128 && C4 == 512 && (Pn == 128 || Pn == 256 || Pn == 0));
}

extern "C" int grl_bottleneck_tail_bf16(const GrlBneckTail* dp, void* stream) {
    if (!dp) return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_bf16: null descriptor");
    const GrlBneckTail& d = *dp;
    if (d.M <= 0 || !d.t2 || !d.w3 || !d.res || !d.y) return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_bf16: null operand or M <= 0");
    if (d.Pn > 0 && (!d.w1n || !d.u)) return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_bf16: Pn > 0 needs w1n and u");
    if (!al16(d.t2) || !al16(d.w3) || !al16(d.res) || !al16(d.y) || !al16(d.w1n) || !al16(d.u))
        return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_bf16: operands must be 16-byte aligned");
    if (!grl_bottleneck_tail_bf16_supported(d.P, d.C4, d.Pn))
        return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_bf16: unsupported shape P %d, C4 %d, Pn %d", d.P, d.C4, d.Pn);
    hipStream_t s = (hipStream_t)stream;
    if (d.P == 64) {
        if (d.Pn == 64) return launch<64, 256, 64, 256>(d, s);
        if (d.Pn == 128) return launch<64, 256, 128, 256>(d, s);
        return launch<64, 256, 0, 256>(d, s);
    }
    if (d.Pn == 128) return launch<128, 512, 128, 128>(d, s);
    if (d.Pn == 256) return launch<128, 512, 256, 64>(d, s);
    return launch<128, 512, 0, 128>(d, s);
}
