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

// The same instruction, invisible to hipcc's waitcnt pass: `sbase` + lane offset `voff` -> LDS byte address `lds` (+ 16*l).
// While an LDS-DMA it knows about is pending hipcc treats the wave like one with a FLAT access in flight -- every vmcnt AND
// lgkmcnt wait it inserts becomes (0), so a DMA issued inside the steady-state loop serialises the loop's residual-load
// ring and its fragment reads.  Issued this way the compiler's counted waits only ever over-wait (the hidden pieces are
// extra outstanding operations), and the kernel waits for the pieces itself in front of the chunk barrier.
// ("m0" is on the clobber list: the statement overwrites it, and the compiler keeps its own LDS-DMA / indexing state there.
//  hipcc accepts the clobber with a -Winline-asm note about reserved registers, silenced for these statements only.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void glds16_hidden(const char* sbase, uint32_t voff, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop

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


// P: conv3 input channels, C4: its output channels (= conv1' input channels), PN: conv1' output channels (0: no chain),
// CH: channels of y per LDS weight chunk (C4 / CH chunks; one chunk = resident weights, no barriers),
// GB: 16-channel blocks per register group (16 * GB channels of accumulators live), PD: residual groups in flight,
// NW: waves per workgroup (16 pixels each; 16 waves = 128 registers per lane, 8 waves = 256 for the widest conv1').
//
// Memory pipeline of a wave (the launch is latency-bound long before it is bandwidth-bound: a group's MFMAs take ~50 ns,
// an HBM load ~2 us): the residual rows of the next PD groups -- across the chunk and the tile boundary -- and the NEXT
// tile's conv3 operand are always in flight; vmcnt retires in order, so waiting for the oldest group leaves the younger
// ones (and the stores issued since) outstanding.
//
// Streaming weights (NCH > 1): two STATIC LDS buffers, the chunk loop unrolled by two so that every LDS-DMA target and
// every fragment read names its buffer at compile time -- hipcc then knows the DMA into buffer B cannot alias the
// ds_reads of buffer A and does not put a vmcnt(0) between them (with one dynamic array it does, and the "prefetch" of
// chunk c + 1 is waited for before chunk c computes).  Chunk boundary: this wave's DMA pieces are older than the
// Chunk boundary: a counted vmcnt that leaves the youngest residual refills in flight + a raw s_barrier (no fence).
// KD > 0: the residual is the block's DOWNSAMPLE branch, computed here: res = bf16(bnd(convd(x0))), x0 [M][KD] (resident variants).
template <int P, int C4, int PN, int CH, int GB, int PD, int NW, int KD = 0>
__global__ __launch_bounds__(NW * 64) void bneck_tail_kernel(const GrlBneckTail p, const int num_tiles) {
    constexpr int TILE_PX = NW * 16;
    constexpr bool CHAIN = PN > 0;
    constexpr int NCH = C4 / CH;
    constexpr int KS3 = P / 32;               // k-steps of conv3
    constexpr int CB = CH / 16;               // 16-channel blocks of y per chunk
    constexpr int GPC = CB / GB;              // register groups per chunk
    constexpr int OB = CHAIN ? PN / 16 : 0;   // 16-channel blocks of u
    constexpr int KS1 = CH / 32;              // conv1' k-steps per chunk
    constexpr bool DS = KD > 0;
    constexpr int KSD = DS ? KD / 32 : 0;     // k-steps of the downsample conv
    constexpr int W3_FR = CB * KS3, W1_FR = OB * KS1, WD_FR = CB * KSD;
    constexpr int BUF = (W3_FR + W1_FR + WD_FR) * 1024;
    constexpr bool PFN = NCH == 1;            // next tile's conv3 operand prefetched into its own registers at the tile top
    static_assert(P % 32 == 0 && CH % 32 == 0 && C4 % CH == 0 && CB % GB == 0 && GB % 2 == 0, "shape");
    static_assert(GPC % PD == 0 && (NCH == 1 || NCH % 2 == 0) && (W3_FR + W1_FR + WD_FR) % NW == 0 && (!DS || NCH == 1), "pipeline");
    __shared__ __attribute__((aligned(16))) char bufA[BUF];
    __shared__ __attribute__((aligned(16))) char bufB[NCH > 1 ? BUF : 16];
    __shared__ __attribute__((aligned(16))) float sc3[C4], sh3[C4], sc1[CHAIN ? PN : 4], sh1[CHAIN ? PN : 4], scd[DS ? C4 : 4], shd[DS ? C4 : 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
    const char* const wd = reinterpret_cast<const char*>(p.wd);
    const uint32_t wdl = (uint32_t)(j * (DS ? KD : 1) + 8 * q) * 2;
    const char* const w3 = reinterpret_cast<const char*>(p.w3);
    const char* const w1 = reinterpret_cast<const char*>(p.w1n);
    // per-lane source offsets of a fragment's 16 bytes: W3 rows are P, W1' rows C4 elements long
    const uint32_t w3l = (uint32_t)(j * P + 8 * q) * 2, w1l = (uint32_t)(j * C4 + 8 * q) * 2;

    // chunk c of the weights -> LDS buffer `dst`, fragment order; wave w issues fragments w, w + NW, ...
    auto stage = [&](int c, char* dst) {
        const uint32_t lds0 = (uint32_t)(size_t)((lptr_t)dst);
#pragma unroll
        for (int k = 0; k < (W3_FR + W1_FR + WD_FR) / NW; ++k) {
            const int f = wave + k * NW;
            if (f < W3_FR) {
                const int cb = f / KS3, s = f - cb * KS3;
                glds16_hidden(w3, w3l + (uint32_t)(((c * CH + 16 * cb) * P + 32 * s) * 2), lds0 + f * 1024);
            } else if (DS && f >= W3_FR + W1_FR) {
                const int g = f - W3_FR - W1_FR, cb = g / (DS ? KSD : 1), s = g - cb * (DS ? KSD : 1);
                glds16_hidden(wd, wdl + (uint32_t)(((c * CH + 16 * cb) * KD + 32 * s) * 2), lds0 + f * 1024);
            } else {
                const int g = f - W3_FR, ob = g / KS1, s = g - ob * KS1;
                glds16_hidden(w1, w1l + (uint32_t)((16 * ob * C4 + c * CH + 32 * s) * 2), lds0 + f * 1024);
            }
        }
    };

    for (int i = tid; i < C4; i += NW * 64) {
        sc3[i] = p.scale3 ? p.scale3[i] : 1.f;
        sh3[i] = p.shift3 ? p.shift3[i] : 0.f;
    }
    if (CHAIN)
        for (int i = tid; i < PN; i += NW * 64) {
            sc1[i] = p.scale1n ? p.scale1n[i] : 1.f;
            sh1[i] = p.shift1n ? p.shift1n[i] : 0.f;
        }
    if (DS)
        for (int i = tid; i < C4; i += NW * 64) {
            scd[i] = p.scaled ? p.scaled[i] : 1.f;
            shd[i] = p.shiftd ? p.shiftd[i] : 0.f;
        }
    stage(0, bufA);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const char* const t2 = reinterpret_cast<const char*>(p.t2);
    const char* const res = reinterpret_cast<const char*>(p.res);
    char* const y = reinterpret_cast<char*>(p.y);
    char* const u = reinterpret_cast<char*>(p.u);
    // 32-bit byte offsets from the wave-uniform bases (the dispatcher guarantees every operand is < 4 GiB)
    const uint32_t sw_b = (uint32_t)(16 * (q & 1) + 8 * (q >> 1)) * 2;   // a lane's 16 bytes inside a block PAIR (swapped layout)
    const uint32_t q16 = (uint32_t)q * 16;

    int tile = blockIdx.x;
    int row = tile * TILE_PX + wave * 16 + j;
    uint32_t rowc = (uint32_t)(row < p.M ? row : p.M - 1);
    bf16x8 bfr[KS3];
    bf16x8 xfr[DS ? KSD : 1];                   // the downsample conv's operand: this pixel's x0 row
    const char* const x0 = reinterpret_cast<const char*>(p.x0);
    uint4 rr[DS ? 1 : PD][GB / 2];
    if (tile < num_tiles) {
#pragma unroll
        for (int s = 0; s < KS3; ++s) bfr[s] = *reinterpret_cast<const bf16x8*>(t2 + (size_t)(rowc * (uint32_t)(P * 2) + 64 * s + q16));
        if (DS) {
#pragma unroll
            for (int s = 0; s < KSD; ++s) xfr[s] = *reinterpret_cast<const bf16x8*>(x0 + (size_t)(rowc * (uint32_t)(KD * 2) + 64 * s + q16));
        }
        if (!DS)
#pragma unroll
        for (int k = 0; k < PD; ++k)
#pragma unroll
            for (int t = 0; t < GB / 2; ++t)
                rr[k][t] = *reinterpret_cast<const uint4*>(res + (size_t)(rowc * (uint32_t)(C4 * 2) + (k * GB * 16 + 32 * t) * 2 + sw_b));
    }
    bool first = true;

#pragma unroll 1
    for (; tile < num_tiles;) {
        const int ntile = tile + (int)gridDim.x;
        const bool has_next = ntile < num_tiles;
        const int rown = ntile * TILE_PX + wave * 16 + j;
        const uint32_t rowcn = has_next ? (uint32_t)(rown < p.M ? rown : p.M - 1) : rowc;
        const bool live = row < p.M;
        bf16x8 bfn[PFN ? KS3 : 1];
        bf16x8 xfn[DS ? KSD : 1];
        if (PFN && has_next) {
#pragma unroll
            for (int s = 0; s < KS3; ++s) bfn[s] = *reinterpret_cast<const bf16x8*>(t2 + (size_t)(rowcn * (uint32_t)(P * 2) + 64 * s + q16));
            if (DS) {
#pragma unroll
                for (int s = 0; s < KSD; ++s) xfn[s] = *reinterpret_cast<const bf16x8*>(x0 + (size_t)(rowcn * (uint32_t)(KD * 2) + 64 * s + q16));
            }
        }
        f32x4 acc1[CHAIN ? OB : 1];
#pragma unroll
        for (int ob = 0; ob < (CHAIN ? OB : 1); ++ob) acc1[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint32_t yrow = (uint32_t)row * (uint32_t)(C4 * 2);
        if (!PFN) {
#pragma unroll
            for (int s = 0; s < KS3; ++s) asm volatile("" ::"v"(bfr[s]));      // the operand's vmcnt wait happens HERE
        }

        // one chunk: `mine` holds its weights, the next chunk is DMA'd into `other` meanwhile
        auto chunk = [&](const int c, const char* const mine, char* const other) {
            if (NCH > 1) {
                if (!(first && c == 0)) {
                    // this wave's pieces of chunk c are the OLDEST operations it has in flight (issued at the top of the
                    // previous chunk); behind them sit that chunk's stores and residual refills.  vmcnt retires in order:
                    // leaving only the refills' count outstanding (GB / 2 per group, always issued when a next chunk exists)
                    // covers the pieces whether or not dead lanes skipped their stores.  Then the barrier publishes every
                    // wave's pieces and says `other` is no longer being read.
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GPC * GB / 2) : "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                if (c + 1 < NCH || has_next) stage(c + 1 < NCH ? c + 1 : 0, other);
            }
            const char* const w3f = mine;
            const char* const w1f = mine + W3_FR * 1024;
            const char* const wdf = mine + (W3_FR + W1_FR) * 1024;
#pragma unroll
            for (int gi = 0; gi < GPC; ++gi) {
                const int slot = DS ? 0 : gi % PD;                  // (compile-time: PD divides the groups of a chunk)
                const int ch0 = (c * CB + gi * GB) * 16;            // first channel of the group
                f32x4 acc3[GB];
#pragma unroll
                for (int b = 0; b < GB; ++b) acc3[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS3; ++s)
#pragma unroll
                    for (int b = 0; b < GB; ++b) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(w3f + ((gi * GB + b) * KS3 + s) * 1024 + lane * 16);
                        acc3[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[s], acc3[b], 0, 0, 0);
                    }
                f32x4 accd[DS ? GB : 1];
                if (DS) {                       // the downsample branch of the same pixels, same blocks
#pragma unroll
                    for (int b = 0; b < GB; ++b) accd[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KSD; ++s)
#pragma unroll
                        for (int b = 0; b < GB; ++b) {
                            const bf16x8 a = *reinterpret_cast<const bf16x8*>(wdf + ((gi * GB + b) * KSD + s) * 1024 + lane * 16);
                            accd[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xfr[s], accd[b], 0, 0, 0);
                        }
                }
#pragma unroll
                for (int t = 0; t < GB / 2; ++t) {
                    uint32_t ax, ay, bx, by;
                    const int cA = ch0 + 32 * t + 4 * q, cB = cA + 16;
                    if (DS) {                   // res = bf16(acc * scale_d + shift_d): the value the unfused launch stores and conv3 re-reads
                        const f32x4 dA = accd[2 * t] * *reinterpret_cast<const f32x4*>(scd + cA) + *reinterpret_cast<const f32x4*>(shd + cA);
                        const f32x4 dB = accd[2 * t + 1] * *reinterpret_cast<const f32x4*>(scd + cB) + *reinterpret_cast<const f32x4*>(shd + cB);
                        ax = pack2(dA[0], dA[1]); ay = pack2(dA[2], dA[3]);
                        bx = pack2(dB[0], dB[1]); by = pack2(dB[2], dB[3]);
                    } else {
                        ax = rr[slot][t].x; ay = rr[slot][t].y; bx = rr[slot][t].z; by = rr[slot][t].w;
                        swap16(ax, bx);         // 16 contiguous bytes per lane -> this lane's 4 channels of block 2t | of block 2t+1
                        swap16(ay, by);
                    }
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
                        const int ks = (gi * GB) / 2 + t;
#pragma unroll
                        for (int ob = 0; ob < OB; ++ob) {
                            const bf16x8 a = *reinterpret_cast<const bf16x8*>(w1f + (ob * KS1 + ks) * 1024 + lane * 16);
                            acc1[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bk, acc1[ob], 0, 0, 0);
                        }
                    }
                    swap16(ax, bx);
                    swap16(ay, by);
                    if (live) *reinterpret_cast<uint4*>(y + (size_t)(yrow + (ch0 + 32 * t) * 2 + sw_b)) = uint4{ax, ay, bx, by};
                    __builtin_amdgcn_sched_barrier(0);      // keep the block pairs in program order: hoisted A-fragment reads spill
                }
                // refill this slot: PD groups ahead -- same chunk, the next chunk, or chunk 0 of the next tile
                if (!DS) {
                    const int gn = gi + PD;                          // compile-time
                    const bool wrap = gn >= GPC && c + 1 == NCH;     // ... of the next tile
                    if (!wrap || has_next) {
                        const uint32_t rbase = (wrap ? rowcn : rowc) * (uint32_t)(C4 * 2);
                        const int chn = wrap ? (gn - GPC) * GB * 16 : ((c + (gn >= GPC ? 1 : 0)) * CB + (gn % GPC) * GB) * 16;
#pragma unroll
                        for (int t = 0; t < GB / 2; ++t)
                            rr[slot][t] = *reinterpret_cast<const uint4*>(res + (size_t)(rbase + (uint32_t)((chn + 32 * t) * 2) + sw_b));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
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
        if (!PFN && has_next) {
            // streaming variants have no registers to spare for a second operand: the next tile's is requested as soon as
            // this tile's last conv3 MFMA has read the current one (its wait is pinned at the tile top, outside the chunk loop)
#pragma unroll
            for (int s = 0; s < KS3; ++s) bfr[s] = *reinterpret_cast<const bf16x8*>(t2 + (size_t)(rowcn * (uint32_t)(P * 2) + 64 * s + q16));
        }
        if (CHAIN) {
            const uint32_t urow = (uint32_t)row * (uint32_t)(PN * 2);
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
                if (live) *reinterpret_cast<uint4*>(u + (size_t)(urow + 64 * t + sw_b)) = uint4{ax, ay, bx, by};
            }
        }
        if (PFN && has_next) {
#pragma unroll
            for (int s = 0; s < KS3; ++s) bfr[s] = bfn[s];
            if (DS) {
#pragma unroll
                for (int s = 0; s < KSD; ++s) xfr[s] = xfn[s];
            }
        }
        tile = ntile;
        row = rown;
        rowc = rowcn;
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

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 convolution of layer 1 (64 -> 64 channels on 64 x 32 maps; resnets1.py:79-81) + folded BatchNorm + ReLU.
//
// As an implicit GEMM on the generic kernel this shape moves 221 KB out of L2 per 128 x 64 tile for 9.4 MFLOP (every
// input pixel is gathered nine times, the 74 KB weight matrix once per tile): it runs at the L2 -> CU limit, 17 % MFMA
// busy, 3.5x its HBM time.  Here the persistent workgroup keeps ALL weights in LDS (72 fragments of 1 KiB, A operand of
// the transposed 16x16x32 MFMA) and stages each tile's input patch ONCE -- 10 rows x 32 pixels x 128 B for 8 output
// rows, double-buffered, filled by LDS-DMA with the 16-byte chunk index XOR-swizzled by the pixel on the SOURCE side
// (conflict-free tap reads); out-of-frame rows come from a zero page, the two out-of-row taps are masked per lane.
// A wave owns one output row (32 pixels = two MFMA column blocks sharing every weight fragment).
__device__ uint4 g_zero_px[8];                 // 128 zero bytes

template <int NW>
__global__ __launch_bounds__(NW * 64) void conv3x3_c64_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ w,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              __bf16* __restrict__ y, const int H, const int tiles_per_img,
                                                              const int num_tiles, const int relu) {
    constexpr int TH = NW;                    // output rows per tile (one per wave)
    constexpr int PR = TH + 2;                // patch rows
    constexpr int PATCH = PR * 32 * 128;      // bytes
    constexpr int NFR = 4 * 18;               // weight fragments: 4 output blocks x (9 taps x 2 channel halves)
    static_assert((PR * 4) % NW == 0 && NFR % NW == 0, "staging");
    __shared__ __attribute__((aligned(16))) char wf[NFR * 1024];
    __shared__ __attribute__((aligned(16))) char patA[PATCH];
    __shared__ __attribute__((aligned(16))) char patB[PATCH];
    __shared__ __attribute__((aligned(16))) float scs[64], shs[64];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
    const char* const x8 = reinterpret_cast<const char*>(x);

    if (tid < 64) {
        scs[tid] = scale ? scale[tid] : 1.f;
        shs[tid] = shift ? shift[tid] : 0.f;
    }
    {   // weights: fragment (ob, ks) lane (i, qq) = w[16ob + i][32ks + 8qq .. +7]; k = tap*64 + channel
        const uint32_t wl = (uint32_t)(j * 576 + 8 * q) * 2;
        const uint32_t lds0 = (uint32_t)(size_t)((lptr_t)wf);
#pragma unroll
        for (int k = 0; k < NFR / NW; ++k) {
            const int f = wave + k * NW, ob = f / 18, ks = f - ob * 18;
            glds16_hidden(reinterpret_cast<const char*>(w), wl + (uint32_t)((16 * ob * 576 + 32 * ks) * 2), lds0 + f * 1024);
        }
    }
    // patch of tile t -> buffer: DMA piece d covers 8 pixels (row d / 4, pixels 8 (d % 4) ..): lane l = (pixel l / 8, LDS chunk
    // l % 8); the chunk it fetches is (l % 8) ^ swz(pixel), so that LDS position c' of a pixel holds source chunk c' ^ swz
    const int dpx = lane >> 3, dch = lane & 7;
    auto swz = [](int px) { return (px ^ (px >> 3)) & 7; };
    auto stage = [&](int t, char* dst) {
        const int img = t / tiles_per_img, ty = t - img * tiles_per_img;
        const uint32_t lds0 = (uint32_t)(size_t)((lptr_t)dst);
        const char* const zero = reinterpret_cast<const char*>(g_zero_px);
#pragma unroll
        for (int k = 0; k < (PR * 4) / NW; ++k) {
            const int d = wave + k * NW, r = d >> 2, px = 8 * (d & 3) + dpx;
            const int iy = ty * TH - 1 + r;
            const bool ok = iy >= 0 && iy < H;                      // wave-uniform
            const uint32_t off = ok ? (uint32_t)((((img * H + iy) * 32 + px) * 64 + 8 * (dch ^ swz(px))) * 2) : (uint32_t)(dch * 16);
            glds16_hidden(ok ? x8 : zero, off, lds0 + d * 1024);
        }
    };
    int tile = blockIdx.x;
    if (tile < num_tiles) stage(tile, patA);
    const uint32_t sw_b = (uint32_t)(16 * (q & 1) + 8 * (q >> 1)) * 2;

    auto compute = [&](int t, const char* const pat) {
        const int img = t / tiles_per_img, ty = t - img * tiles_per_img;
        f32x4 acc[2][4];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) acc[pb][ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3 - 1;               // patch row = wave + dy
            bf16x8 b[2][2];
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const int px = 16 * pb + j + dx;
                const bool ok = px >= 0 && px < 32;
                const int pc = ok ? px : (px < 0 ? 0 : 31);
                const char* const row = pat + ((wave + dy) * 32 + pc) * 128;
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    uint4 v = *reinterpret_cast<const uint4*>(row + (((4 * ch + q) ^ swz(pc)) << 4));
                    if (!ok) v = uint4{0u, 0u, 0u, 0u};
                    b[pb][ch] = *reinterpret_cast<bf16x8*>(&v);
                }
            }
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(wf + (ob * 18 + 2 * tap + ch) * 1024 + lane * 16);
                    acc[0][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[0][ch], acc[0][ob], 0, 0, 0);
                    acc[1][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[1][ch], acc[1][ob], 0, 0, 0);
                }
        }
        const int oy = ty * TH + wave;
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const uint32_t orow = (uint32_t)((((img * H + oy) * 32) + 16 * pb + j) * 128);
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int cA = 32 * t2 + 4 * q, cB = cA + 16;
                f32x4 vA = acc[pb][2 * t2] * *reinterpret_cast<const f32x4*>(scs + cA) + *reinterpret_cast<const f32x4*>(shs + cA);
                f32x4 vB = acc[pb][2 * t2 + 1] * *reinterpret_cast<const f32x4*>(scs + cB) + *reinterpret_cast<const f32x4*>(shs + cB);
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        vA[e] = vA[e] > 0.f ? vA[e] : 0.f;
                        vB[e] = vB[e] > 0.f ? vB[e] : 0.f;
                    }
                }
                uint32_t ax = pack2(vA[0], vA[1]), ay = pack2(vA[2], vA[3]);
                uint32_t bx = pack2(vB[0], vB[1]), by = pack2(vB[2], vB[3]);
                swap16(ax, bx);
                swap16(ay, by);
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(y) + (size_t)(orow + 64 * t2 + sw_b)) = uint4{ax, ay, bx, by};
            }
        }
    };

    // tile loop, two tiles per trip so that every LDS-DMA target and every tap read names its buffer at compile time.
    // Per tile: this wave's patch pieces are the oldest operations it has in flight, the 4 output stores of the previous
    // tile the only younger ones (every lane stores: the geometry has no ragged edge) -> vmcnt(4), barrier, next DMA.
    bool first = true;
#pragma unroll 1
    for (; tile < num_tiles;) {
        if (first) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // weights, first patch, the scale / shift vectors
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        first = false;
        const int t1 = tile + (int)gridDim.x;
        if (t1 < num_tiles) stage(t1, patB);
        compute(tile, patA);
        if (t1 >= num_tiles) break;
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int t2 = t1 + (int)gridDim.x;
        if (t2 < num_tiles) stage(t2, patA);
        compute(t1, patB);
        tile = t2;
    }
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <int P, int C4, int PN, int CH, int GB = 4, int PD = 4, int NW = 16, int KD = 0>
int launch(const GrlBneckTail& d, hipStream_t s) {
    constexpr int NCH = C4 / CH;
    constexpr int FR = (CH / 16) * (P / 32) + (PN / 16) * (CH / 32) + (CH / 16) * (KD / 32);
    constexpr int LDS = (2 * C4 + 2 * PN + (KD ? 2 * C4 : 0)) * 4 + (NCH > 1 ? 2 : 1) * FR * 1024;      // static, in the kernel descriptor
    static_assert(LDS <= 160 * 1024 - 64, "LDS");
    static const int cus = [] {
        int dev = 0, n = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n = prop.multiProcessorCount;
        return n;
    }();
    constexpr int TILE_PX = NW * 16;
    const int num_tiles = (d.M + TILE_PX - 1) / TILE_PX;
    const unsigned grid = (unsigned)(num_tiles < cus ? num_tiles : cus);
    hipLaunchKernelGGL((bneck_tail_kernel<P, C4, PN, CH, GB, PD, NW, KD>), dim3(grid), dim3(NW * 64), 0, s, d, num_tiles);
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

extern "C" int grl_conv3x3_c64_bf16(const void* x, const void* w, const float* scale, const float* shift, void* y, int n_img,
                                    int H, int W, int relu, void* stream) {
    if (!x || !w || !y || n_img <= 0) return grl_fail(GRL_EINVAL, "grl_conv3x3_c64_bf16: null operand");
    if (W != 32 || H % 8 || H <= 0) return grl_fail(GRL_EINVAL, "grl_conv3x3_c64_bf16: needs W == 32 and H %% 8 == 0 (got %d x %d)", H, W);
    if (!al16(x) || !al16(w) || !al16(y)) return grl_fail(GRL_EINVAL, "grl_conv3x3_c64_bf16: operands must be 16-byte aligned");
    if ((int64_t)n_img * H * W * 128 >= (1ll << 32)) return grl_fail(GRL_EINVAL, "grl_conv3x3_c64_bf16: tensor too large for 32-bit offsets");
    static const int cus = [] {
        int dev = 0, n = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n = prop.multiProcessorCount;
        return n;
    }();
    const int tiles_per_img = H / 8, num_tiles = n_img * tiles_per_img;
    const unsigned grid = (unsigned)(num_tiles < cus ? num_tiles : cus);
    hipLaunchKernelGGL((conv3x3_c64_kernel<8>), dim3(grid), dim3(512), 0, (hipStream_t)stream, (const __bf16*)x, (const __bf16*)w,
                       scale, shift, (__bf16*)y, H, tiles_per_img, num_tiles, relu);
    return grl_check_launch("grl_conv3x3_c64_bf16");
}

extern "C" int grl_bottleneck_tail_bf16_supported(int P, int C4, int Pn) {
    return (P == 64 && C4 == 256 && (Pn == 64 || Pn == 128 || Pn == 0)) || (P == 128 && C4 == 512 && (Pn == 128 || Pn == 256 || Pn == 0));
}

extern "C" int grl_bottleneck_tail_bf16(const GrlBneckTail* dp, void* stream) {
    if (!dp) return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_bf16: null descriptor");
    const GrlBneckTail& d = *dp;
    if (d.M <= 0 || !d.t2 || !d.w3 || !d.y) return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_bf16: null operand or M <= 0");
    if (d.Kd > 0 ? (!d.x0 || !d.wd || !al16(d.x0) || !al16(d.wd)) : !d.res)
        return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_bf16: needs res, or (Kd > 0) x0 and wd, 16-byte aligned");
    if (d.Kd > 0 && !(d.P == 64 && d.C4 == 256 && d.Pn == 64 && d.Kd == 64))
        return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_bf16: the fused downsample branch exists for P 64, C4 256, Pn 64, Kd 64 only");
    if (d.Pn > 0 && (!d.w1n || !d.u)) return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_bf16: Pn > 0 needs w1n and u");
    if (!al16(d.t2) || !al16(d.w3) || !al16(d.res) || !al16(d.y) || !al16(d.w1n) || !al16(d.u))
        return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_bf16: operands must be 16-byte aligned");
    if ((int64_t)d.M * d.C4 * 2 >= (1ll << 32)) return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_bf16: M * C4 too large for 32-bit offsets");
    if (!grl_bottleneck_tail_bf16_supported(d.P, d.C4, d.Pn))
        return grl_fail(GRL_EINVAL, "grl_bottleneck_tail_bf16: unsupported shape P %d, C4 %d, Pn %d", d.P, d.C4, d.Pn);
    hipStream_t s = (hipStream_t)stream;
    if (d.P == 64) {
        if (d.Pn == 64 && d.Kd == 64) return launch<64, 256, 64, 256, 4, 4, 16, 64>(d, s);
        if (d.Pn == 64) return launch<64, 256, 64, 256>(d, s);
        if (d.Pn == 128) return launch<64, 256, 128, 256>(d, s);
        return launch<64, 256, 0, 256>(d, s);
    }
    if (d.Pn == 128) return launch<128, 512, 128, 128, 4, 2>(d, s);
    if (d.Pn == 256) return launch<128, 512, 256, 64, 2, 2, 16>(d, s);
    return launch<128, 512, 0, 128, 4, 2>(d, s);
}
