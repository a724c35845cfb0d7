// Cross-layer fusion for layers 1-2 of the EXACT-fp32 trunk (BASELINE configs[1], the headline) on gfx950 (MI355X).
//
//   grl_bottleneck_tail_f32:  one launch for the END of a ResNet bottleneck and the START of the next one
//
//       y = relu( bn3(conv3(t2)) + res )          (reid/models/resnets1.py:86-91: 1x1 expansion P -> 4P, residual, ReLU)
//       u = relu( bn1'(conv1'(y)) )                (resnets1.py:76-78 of the NEXT block: 1x1 reduction 4P -> P')
//
// Unfused these are the two shapes of the step that sit furthest below the fp32 MFMA roof (262144 x 256 x 64 at 38 % of
// peak, 65536 x 512 x 128 at 50 %): one K <= 128 stage of arithmetic per 128 x 128 tile against a full tile of
// residual reads and output writes, and the next conv1 reads the 4P-wide tensor straight back.  Here a wave owns 32
// pixels, walks all their output channels with the WEIGHTS as the MFMA's A operand (v_mfma_f32_32x32x2_f32, pixels as
// columns), and a pixel's 4P outputs are contracted against conv1' in the registers that produced them: y is written
// once (the next block's residual), never re-read, and a launch per block goes.
//
// Numerics -- BIT-IDENTICAL to the unfused gemm_f32_kernel launches (tested): the transposed accumulator layout (a lane
// holds channels 8g + 4h + r of its pixel, h = lane / 32) is exactly the pairing of that kernel's documented k-ordered
// fmaf chain -- for 32-wide stage j, chunk q, step s: k0 = 32j + 8q + s (lane half 0), k1 = k0 + 4 (lane half 1) -- so
// register (g, r) of a 32-channel block of y IS the B operand of the chained MFMA (j = block, q = g, s = r), and conv3
// reads t2 the same way; one accumulator per output, the same order, the same epilogue (acc*scale + shift, + res, ReLU).
//
// Weights sit in LDS in fragment order (a lane's four k of a (block, chunk) as one float4: one ds_read_b128 feeds four
// MFMAs), resident for layer 1, streamed through two static buffers in channel chunks elsewhere: the next chunk is
// fetched into registers at the top of a chunk and written to the other buffer at its end -- plain loads the compiler
// counts, no LDS-DMA inside the loop (hipcc drops every counted wait to zero while one is pending: fuse_bf16.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/grl_hip.h"
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

// P: conv3 input channels, C4: its output channels (= conv1' input channels), PN: conv1' output channels (0: no chain),
// CH: channels of y per LDS weight chunk, NW: waves per workgroup (32 pixels each), PD: 32-channel blocks of residual in flight
template <int P, int C4, int PN, int CH, int NW, int PD, int KO = 0>     // KO: timing knock-outs (1: no MFMA, 2: no residual / store traffic)
__global__ __launch_bounds__(NW * 64) void bneck_tail_f32_kernel(const GrlBneckTailF32 p, const int num_tiles) {
    constexpr int NT = NW * 64;
    constexpr int TILE_PX = NW * 32;
    constexpr bool CHAIN = PN > 0;
    constexpr int NCH = C4 / CH;
    constexpr int KQ = P / 8;                 // float4 k-chunks of conv3 (4 MFMAs each)
    constexpr int CB = CH / 32;               // 32-channel blocks of y per chunk
    constexpr int OB = CHAIN ? PN / 32 : 0;   // 32-channel blocks of u
    constexpr int W3_FR = CB * KQ, W1_FR = OB * CB * 4;          // 1 KiB fragments per chunk
    constexpr int FR = W3_FR + W1_FR;
    constexpr int BUF = FR * 1024;
    constexpr bool PFN = NCH == 1 && P <= 64;                   // next tile's conv3 operand prefetched into its own registers
    constexpr int ST = (FR * 64 + NT - 1) / NT;                  // float4 items a thread stages per chunk
    static_assert(P % 32 == 0 && CH % 64 == 0 && C4 % CH == 0 && (NCH == 1 || NCH % 2 == 0) && (PD == 1 || PD == 2), "shape");
    static_assert((FR * 64) % NT == 0, "staging");
    __shared__ __attribute__((aligned(16))) char bufA[BUF];
    __shared__ __attribute__((aligned(16))) char bufB[NCH > 1 ? BUF : 16];
    __shared__ __attribute__((aligned(16))) float sc3[C4], sh3[C4], sc1[CHAIN ? PN : 4], sh1[CHAIN ? PN : 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    // weights of chunk c, item `it` of this thread (fragment f = item / 64, fragment lane fl = item % 64):
    //   W3 fragment (cb, kq):    lane (i, hh) holds W3[c*CH + 32cb + i][8kq + 4hh .. +3]
    //   W1' fragment (ob, cb, g): lane (i, hh) holds W1'[32ob + i][c*CH + 32cb + 8g + 4hh .. +3]
    auto wsrc = [&](int c, int it) -> const f32x4* {
        const int item = tid + it * NT;
        const int f = item >> 6, fl = item & 63, i = fl & 31, hh = fl >> 5;
        if (f < W3_FR) {
            const int cb = f / KQ, kq = f - cb * KQ;
            return reinterpret_cast<const f32x4*>(p.w3 + (int64_t)(c * CH + 32 * cb + i) * P + 8 * kq + 4 * hh);
        }
        const int g2 = f - W3_FR, ob = g2 / (CB * 4), r2 = g2 - ob * (CB * 4), cb = r2 >> 2, g = r2 & 3;
        return reinterpret_cast<const f32x4*>(p.w1n + (int64_t)(32 * ob + i) * C4 + c * CH + 32 * cb + 8 * g + 4 * hh);
    };
    // the next chunk travels in two halves (half the staging registers): first half fetched at the chunk top and parked in
    // LDS after the first block pair, second half fetched then and parked at the chunk end
    constexpr int SH = NCH > 1 ? (ST + 1) / 2 : 1;
    f32x4 wst[SH];
    auto wload = [&](int c, int half) {
#pragma unroll
        for (int it = 0; it < SH; ++it)
            if (half * SH + it < ST) wst[it] = *wsrc(c, half * SH + it);
    };
    auto wstore = [&](char* dst, int half) {
#pragma unroll
        for (int it = 0; it < SH; ++it)
            if (half * SH + it < ST) *reinterpret_cast<f32x4*>(dst + (size_t)(tid + (half * SH + it) * NT) * 16) = wst[it];
    };

    const uint32_t h16 = (uint32_t)h * 16;          // a lane half's 4 floats inside an 8-channel group (bytes)
    const char* const t2 = reinterpret_cast<const char*>(p.t2);
    const char* const res = reinterpret_cast<const char*>(p.res);
    char* const y = reinterpret_cast<char*>(p.y);
    char* const u = reinterpret_cast<char*>(p.u);

    int tile = blockIdx.x;
    int row = tile * TILE_PX + wave * 32 + j;
    uint32_t rowc = (uint32_t)(row < p.M ? row : p.M - 1);
    f32x4 b3[KQ];                                   // conv3's B operand: this pixel's t2 row, k = 8kq + 4h + s
    f32x4 rr[PD][4];                                // residual ring: PD blocks x 4 groups of 8 channels
    if (tile < num_tiles) {
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) b3[kq] = *reinterpret_cast<const f32x4*>(t2 + (size_t)(rowc * (uint32_t)(P * 4) + 32 * kq + h16));
#pragma unroll
        for (int k = 0; k < PD; ++k)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                rr[k][g] = *reinterpret_cast<const f32x4*>(res + (size_t)(rowc * (uint32_t)(C4 * 4) + (32 * k + 8 * g) * 4 + h16));
    }
    for (int i = tid; i < C4; i += NT) {
        sc3[i] = p.scale3 ? p.scale3[i] : 1.f;
        sh3[i] = p.shift3 ? p.shift3[i] : 0.f;
    }
    if (CHAIN)
        for (int i = tid; i < PN; i += NT) {
            sc1[i] = p.scale1n ? p.scale1n[i] : 1.f;
            sh1[i] = p.shift1n ? p.shift1n[i] : 0.f;
        }
#pragma unroll 1
    for (int it0 = 0; it0 < ST; it0 += 8) {         // (one-off; eight loads in flight per thread, nothing else is live yet)
        f32x4 tmp[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (it0 + k < ST) tmp[k] = *wsrc(0, it0 + k);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (it0 + k < ST) *reinterpret_cast<f32x4*>(bufA + (size_t)(tid + (it0 + k) * NT) * 16) = tmp[k];
    }
    __syncthreads();

    bool first = true;

#pragma unroll 1
    for (; tile < num_tiles;) {
        const int ntile = tile + (int)gridDim.x;
        const bool has_next = ntile < num_tiles;
        const int rown = ntile * TILE_PX + wave * 32 + j;
        const uint32_t rowcn = has_next ? (uint32_t)(rown < p.M ? rown : p.M - 1) : rowc;
        const bool live = row < p.M;
        f32x4 b3n[PFN ? KQ : 1];
        if (PFN && has_next) {
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) b3n[kq] = *reinterpret_cast<const f32x4*>(t2 + (size_t)(rowcn * (uint32_t)(P * 4) + 32 * kq + h16));
        }
        f32x16 acc1[CHAIN ? OB : 1];
#pragma unroll
        for (int ob = 0; ob < (CHAIN ? OB : 1); ++ob)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[ob][e] = 0.f;
        const uint32_t yrow = (uint32_t)row * (uint32_t)(C4 * 4);

        // one chunk: `mine` holds its weights; the next chunk travels through registers into `other`
        auto chunk = [&](const int c, const char* const mine, char* const other) {
            const bool stream = NCH > 1 && (c + 1 < NCH || has_next);
            if (NCH > 1) {
                if (!(first && c == 0)) __syncthreads();        // every wave's part of chunk c is in `mine`; `other` is free
                if (stream) wload(c + 1 < NCH ? c + 1 : 0, 0);
            }
            const char* const w3f = mine;
            const char* const w1f = mine + W3_FR * 1024;
            // Software pipeline over the 32-channel blocks: conv3 of block cb + 1 is ISSUED before the epilogue of block cb, so
            // that block cb's epilogue VALU (scale / shift / residual / ReLU / stores) runs in the shadow of those MFMAs --
            // an in-order wave issues nothing behind a blocked MFMA, but it does issue VALU between two of them.
            // (Measured and dropped: two blocks' accumulators alternating inside conv3 -- a dependent-chain theory of the
            // ~70 % matrix-pipe ceiling of this kernel -- changed nothing; neither did 16 waves per CU.)
            auto conv3_block = [&](int cb, f32x16& acc) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                // fragments double-buffered in registers: the read of k-chunk kq + 1 is issued BEFORE the MFMAs of kq (pinned
                // with sched_barrier -- hipcc otherwise re-uses one register quad and the LDS latency lands between MFMA groups)
                f32x4 fa[2];
                fa[0] = *reinterpret_cast<const f32x4*>(w3f + (cb * KQ) * 1024 + lane * 16);
#pragma unroll
                for (int kq = 0; kq < KQ; ++kq) {
                    if (kq + 1 < KQ && KO != 4) fa[(kq + 1) & 1] = *reinterpret_cast<const f32x4*>(w3f + (cb * KQ + kq + 1) * 1024 + lane * 16);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < 4; ++s) { if (KO == 1) acc[s] += fa[kq & 1][s] * b3[kq][s]; else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kq & 1][s], b3[kq][s], acc, 0, 0, 0); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            f32x16 accs[2];
            conv3_block(0, accs[0]);
#pragma unroll 1
            for (int cb0 = 0; cb0 < CB; cb0 += 2)
#pragma unroll
            for (int par = 0; par < 2; ++par) {                     // (accumulator parity and ring slot are compile-time, the block is not)
                const int cb = cb0 + par;
                const int slot = par % PD;
                const int ch0 = c * CH + 32 * cb;                   // first channel of the block
                if (cb + 1 < CB) conv3_block(cb + 1, accs[par ^ 1]);
                if (!PFN && c + 1 == NCH && cb + 2 == CB && has_next) {
                    // the next tile's conv3 operand, requested as soon as this tile's last conv3 MFMA has read the current one
                    // (its MFMAs have been issued: an MFMA reads its sources at issue)
#pragma unroll
                    for (int kq = 0; kq < KQ; ++kq) b3[kq] = *reinterpret_cast<const f32x4*>(t2 + (size_t)(rowcn * (uint32_t)(P * 4) + 32 * kq + h16));
                }
                const f32x16& acc3 = accs[par];
                f32x4 v[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cg = ch0 + 8 * g + 4 * h;
                    const f32x4 s4 = *reinterpret_cast<const f32x4*>(sc3 + cg), h4 = *reinterpret_cast<const f32x4*>(sh3 + cg);
                    f32x4 t = {acc3[4 * g], acc3[4 * g + 1], acc3[4 * g + 2], acc3[4 * g + 3]};
                    if (KO < 3) {
                        t = t * s4 + h4;
                        t += rr[slot][g];
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = t[e] > 0.f ? t[e] : 0.f;
                    }
                    v[g] = t;
                    if (live && ((KO != 2 && KO < 3) || t[0] == 123.f)) *reinterpret_cast<f32x4*>(y + (size_t)(yrow + (uint32_t)((ch0 + 8 * g) * 4) + h16)) = t;
                }
                // refill this slot: PD blocks ahead -- same chunk, the next chunk, or chunk 0 of the next tile
                {
                    const int bn = cb + PD;
                    const bool wrap = bn >= CB && c + 1 == NCH;
                    if ((!wrap || has_next) && KO != 2 && KO < 3) {
                        const uint32_t rbase = (wrap ? rowcn : rowc) * (uint32_t)(C4 * 4);
                        const int chn = wrap ? (bn - CB) * 32 : (c + (bn >= CB ? 1 : 0)) * CH + (bn % CB) * 32;
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            rr[slot][g] = *reinterpret_cast<const f32x4*>(res + (size_t)(rbase + (uint32_t)((chn + 8 * g) * 4) + h16));
                    }
                }
                if (CHAIN) {                    // block cb of y IS stage cb of conv1' (chunk q = g, step s = r): same fmaf chain
                    f32x4 fc[2];
                    fc[0] = *reinterpret_cast<const f32x4*>(w1f + ((0 * CB + cb) * 4 + 0) * 1024 + lane * 16);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int ob = 0; ob < OB; ++ob) {
                            const int i = g * OB + ob;               // fragment (ob, cb, g); the next one is read ahead
                            if (i + 1 < 4 * OB && KO != 4) {
                                const int gn = (i + 1) / OB, obn = (i + 1) % OB;
                                fc[(i + 1) & 1] = *reinterpret_cast<const f32x4*>(w1f + ((obn * CB + cb) * 4 + gn) * 1024 + lane * 16);
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int r = 0; r < 4; ++r) { if (KO == 1) acc1[ob][r] += fc[i & 1][r] * v[g][r]; else acc1[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(fc[i & 1][r], v[g][r], acc1[ob], 0, 0, 0); }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);      // one block at a time: hoisted fragment reads of later blocks spill
                if (NCH > 1 && par == 0 && cb0 == 0 && stream) {
                    wstore(other, 0);
                    wload(c + 1 < NCH ? c + 1 : 0, 1);
                }
            }
            if (stream) wstore(other, 1);
        };
        if (NCH == 1) {
            chunk(0, bufA, bufA);
        } else {
#pragma unroll 1
            for (int c = 0; c < NCH; c += 2) {
                chunk(c, bufA, bufB);
                chunk(c + 1, bufB, bufA);
            }
        }
        first = false;
        if (CHAIN) {
            const uint32_t urow = (uint32_t)row * (uint32_t)(PN * 4);
#pragma unroll
            for (int ob = 0; ob < OB; ++ob)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cg = 32 * ob + 8 * g + 4 * h;
                    const f32x4 s4 = *reinterpret_cast<const f32x4*>(sc1 + cg), h4 = *reinterpret_cast<const f32x4*>(sh1 + cg);
                    f32x4 t = {acc1[ob][4 * g], acc1[ob][4 * g + 1], acc1[ob][4 * g + 2], acc1[ob][4 * g + 3]};
                    t = t * s4 + h4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = t[e] > 0.f ? t[e] : 0.f;
                    if (live) *reinterpret_cast<f32x4*>(u + (size_t)(urow + (uint32_t)((32 * ob + 8 * g) * 4) + h16)) = t;
                }
        }
        if (PFN && has_next) {
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) b3[kq] = b3n[kq];
        }
        tile = ntile;
        row = rown;
        rowc = rowcn;
    }
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <int P, int C4, int PN, int CH, int NW = 8, int PD = 2, int KO = 0>
int launch(const GrlBneckTailF32& d, hipStream_t s) {
    static const int cus = [] {
        int dev = 0, n = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n = prop.multiProcessorCount;
        return n;
    }();
    constexpr int TILE_PX = NW * 32;
    const int num_tiles = (d.M + TILE_PX - 1) / TILE_PX;
    const unsigned grid = (unsigned)(num_tiles < cus ? num_tiles : cus);
    hipLaunchKernelGGL((bneck_tail_f32_kernel<P, C4, PN, CH, NW, PD, KO>), dim3(grid), dim3(NW * 64), 0, s, d, num_tiles);
    return grl_check_launch("grl_bottleneck_tail_f32");
}

}  // namespace

extern "C" int grl_bottleneck_tail_f32_supported(int P, int C4, int Pn) {
    return (P == 64 && C4 == 256 && (Pn == 64 || Pn == 128 || Pn == 0)) || (P == 128 && C4 == 512 && (Pn == 128 || Pn == 0));
}

extern "C" int grl_bottleneck_tail_f32(const GrlBneckTailF32* dp, void* stream) {
    if (!dp) return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_f32: null descriptor");
    const GrlBneckTailF32& d = *dp;
    if (d.M <= 0 || !d.t2 || !d.w3 || !d.res || !d.y) return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_f32: null operand or M <= 0");
    if (d.Pn > 0 && (!d.w1n || !d.u)) return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_f32: Pn > 0 needs w1n and u");
    if (!al16(d.t2) || !al16(d.w3) || !al16(d.res) || !al16(d.y) || !al16(d.w1n) || !al16(d.u))
        return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_f32: operands must be 16-byte aligned");
    if ((int64_t)d.M * d.C4 * 4 >= (1ll << 32)) return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_f32: M * C4 too large for 32-bit offsets");
    if (!grl_bottleneck_tail_f32_supported(d.P, d.C4, d.Pn))
        return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_f32: unsupported shape P %d, C4 %d, Pn %d", d.P, d.C4, d.Pn);
    hipStream_t s = (hipStream_t)stream;
    static const int cfg = [] { const char* e = getenv("GRL_TAILF32_CFG"); return e ? atoi(e) : 0; }();     // A/B only
    if (d.P == 64) {
        if (d.Pn == 64 && cfg == 11) return launch<64, 256, 64, 256, 8, 2, 1>(d, s);
        if (d.Pn == 64 && cfg == 12) return launch<64, 256, 64, 256, 8, 2, 2>(d, s);
        if (d.Pn == 64 && cfg == 13) return launch<64, 256, 64, 256, 8, 2, 3>(d, s);
        if (d.Pn == 64 && cfg == 14) return launch<64, 256, 64, 256, 8, 2, 4>(d, s);
        if (d.Pn == 64) return launch<64, 256, 64, 256>(d, s);
        if (d.Pn == 128) return launch<64, 256, 128, 64>(d, s);
        return launch<64, 256, 0, 256>(d, s);
    }
    if (d.Pn == 128) return launch<128, 512, 128, 64, 8, 1>(d, s);
    return launch<128, 512, 0, 64>(d, s);
}
