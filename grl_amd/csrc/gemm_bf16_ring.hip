// bf16-storage GEMM / implicit-GEMM convolution for gfx950 (MI355X): large tiles fed by an LDS ring (round 5).
//
//   Y[M][N] = epilogue( A[M][K] . W[N][K]^T ),  A, W, residual, Y bf16 in HBM, fp32 accumulate
//
// Successor of gemm_bf16_256_kernel (gemm_bf16.hip, rounds 2-4: two 64-k stages of 64 KiB, the DMA of stage kt+1
// issued inside stage kt and drained with vmcnt(0) at the one barrier per stage).  That kernel's matrix pipe was busy
// 33-45 % of the time: a stage's 64 KiB had to come out of L2 in less than the stage's own 0.85 us of matrix work,
// every barrier drained the vector-memory queue, and the epilogue kept four 1 KiB residual loads in flight per wave.
// Same tile, same MFMA, same k order (bit-identical results) -- what changes is the memory side:
//
//   * the K loop consumes HALF-stages of 32 k (64-byte LDS rows, 16-byte chunks XOR-swizzled by (row >> 2) & 3: conflict-
//     free ds_read_b128) out of a ring of five slots of (TM + TN) * 64 bytes, one barrier per half-stage; the DMA is
//     issued per 64-k STAGE -- the two k-halves of the same 16 rows by two wave-instructions back to back, into two
//     slots -- three to four half-stages ahead of the MFMAs and never drains (`s_waitcnt vmcnt(2 * NP)`): whole 128-byte
//     lines cross the L2 -> L1 fabric once (half-lines requested a half-stage apart cross it twice: 75 instead of
//     110-120 GB/s per CU, tools/hw_probe/dma_rate.hip), yet LDS is freed in 32-k units;
//   * the ring does not stop at tile boundaries: the loader is a separate little state machine (its own tile, its own
//     k position) that keeps issuing the NEXT tile's first half-stages under the current tile's last MFMAs and epilogue;
//   * DMA pieces are hidden from the compiler (inline asm, SGPR base + 32-bit lane offset): hipcc degrades every wait
//     to (0) while a global_load_lds it knows about is pending (EXPERIMENTS.md, round 4);
//   * the barrier sits in front of a half-stage's LAST eight MFMAs and the next half-stage's first fragment reads right
//     behind it, so the LDS latency after a barrier is covered by matrix work;
//   * epilogue: the whole wave tile's residual rows (16 x 1 KiB per wave) are requested before the first accumulator
//     block goes through the LDS slab; the slab is the wave's own DMA footprint of the slot it consumed last, so the
//     ring needs no LDS of its own for it (NS = 5 x 32 KiB = all 160 KiB of a CU).
//
// Tiles: 256 x 256 (8 waves as 2 x 4, wave tile 128 x 64), 256 x 128 (4 x 2, 64 x 64) and 128 x 256 (2 x 4, 64 x 64).
// `GrlGemmGroup`: up to four problems of one shape in one launch (the two directions of a TRL step: grl_model.py:131-167).
//
// Reference call sites of the convolutions it runs: reid/models/resnets1.py:62-68,73-93, basebranch.py:42-50,
// grl_model.py:56-64,95-121.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "../../include/grl_hip.h"
#include "common.h"

namespace ring {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef GRL_RING_SCHED
#define GRL_RING_SCHED 2
#endif
#ifndef GRL_RING_M16
#define GRL_RING_M16 0     // timing-only probe: v_mfma_f32_16x16x32_bf16 on the same fragment traffic (wrong results)
#endif
#ifndef GRL_RING_KO
#define GRL_RING_KO 0      // timing-only knock-outs (wrong results): 1 no DMA in the loop, 2 no epilogue, 4 no MFMA, 8 every half-stage re-reads k block 0, 16 no fragment reads
#endif

__device__ uint4 g_zero_row16[8];             // 128 zero bytes: source of out-of-image taps

typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA piece the compiler does not see: lane l's 16 bytes at sbase + voff land at lds + 16 * l
// ("m0" is on the clobber list: the statement overwrites it, and the compiler keeps its own LDS-DMA / indexing state there.
//  hipcc accepts the clobber with a -Winline-asm note about reserved registers, silenced for these statements only.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16(const char* sbase, uint32_t voff, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void dma16_v(const char* vaddr, uint32_t lds) {       // per-lane 64-bit address (conv gather)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(vaddr), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop

// wave-uniform pointer held in SGPRs (the "s" operand of dma16 must not end up in a VGPR pair)
template <class T> __device__ __forceinline__ T* uniform_ptr(T* q) {
    const uint64_t v = (uint64_t)(uintptr_t)q;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<T*>((uintptr_t)(((uint64_t)hi << 32) | lo));
}
// element g of a by-value kernel-argument array without dynamic indexing (which would move the struct to scratch)
template <class T> __device__ __forceinline__ T pick4(const T (&arr)[4], int g) {
    return g == 0 ? arr[0] : g == 1 ? arr[1] : g == 2 ? arr[2] : arr[3];
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Up to four problems of ONE shape per launch: everything that differs between them
struct Group {
    const void* a[4];
    const void* w[4];
    void* y[4];
    const float* scale[4];
    const float* shift[4];
    const void* res[4];
    int n;
};

// TM x TN tile, 8 waves; wave tile WM x 64.  NS ring slots.  CONV: implicit-GEMM gather of A.  STATS / SQD: see
// gemm_bf16.hip (same epilogues, same summation orders).  DEEPRES: residual rows of the whole wave tile requested up
// front (else per 32-row block).
template <int TM, int TN, int NS, bool CONV, bool STATS, bool SQD, bool GROUPED, bool RES, bool GBIAS>
__global__ __launch_bounds__(512, 2) void gemm_bf16_ring_kernel(const GrlGemm p, const Group grp, const int tiles_n,
                                                                const int tiles_per_problem, const int num_tiles,
                                                                const int phase_cycles) {
    constexpr int WGN = TN / 64, WGM = 8 / WGN;           // waves along N / M
    constexpr int WM = TM / WGM, MI = WM / 32;            // wave tile rows, MFMA row blocks
    constexpr int SLOT = (TM + TN) * 64;                  // bytes of one ring slot (A rows then W rows, 64 B each)
    constexpr int PA = TM / 128, PB = TN / 128;           // 1 KiB DMA pieces per wave per half-stage
    constexpr int NP = PA + PB;
    // DMA issue schedule.  A vector-memory instruction blocks its wave until the texture-address unit takes it (about
    // 18 cycles per 1 KiB instruction at the 110 GB/s a CU gets out of L2): eight waves that reach their DMA
    // instructions together (they leave every barrier together) all sit in that queue, the two waves of a SIMD
    // included, and nobody issues MFMAs -- the DMA's transfer time ADDS to the matrix time (knock-outs: MFMA only
    // 0.53 us per half-stage, MFMA + DMA 0.80, everything 1.08).  So every wave gets its own window: wave w issues its
    // NP instructions of the half-stage in one burst in front of MFMA pair w of the half-stage's eight (waves 0-3 in
    // front of the barrier, their SIMD partners 4-7 behind it).  A stage is 2 * NP instructions per wave: chunk A
    // (pieces 0 .. NP/2-1, both k-halves) in the half-stage that opens it, chunk B one half-stage later.  Waves 0-3 open
    // stage (2s, 2s+1) in half-stage 2s-3, waves 4-7 behind the barrier of half-stage 2s-4 (both slots are free there).
    constexpr bool SLAB_ALIAS = (NS * SLOT + 8 * 4096) > 160 * 1024;
    static_assert(!SLAB_ALIAS || (PA == 2 && PB == 2), "slab aliasing: the wave's DMA footprint must be 2 + 2 KiB");
    static_assert(MI == 2 || MI == 4, "wave tile");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WGN, wc = wave % WGN;
    const bool late = wave >= 4;                             // DMA schedule: waves 4-7 issue behind the barrier
    const int win = MI == 4 ? wave : wave >> 1;              // ... in window `win` of the half-stage's 2 * MI
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;

    // ------------------------------------------------------------------ loader (runs D half-stages ahead)
    const int prow = lane >> 2;                                              // row inside a 16-row piece
    const unsigned psw = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);   // swizzled SOURCE chunk (bytes)
    const int nkh = p.K / 32;                                                // half-stages per tile
    const int my_tiles = (num_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * nkh;                                        // half-stages this workgroup consumes
    unsigned l_aoff[PA], l_boff[PB], l_yx0[PA];
    const char* l_a8 = reinterpret_cast<const char*>(p.a);
    const char* l_w8 = reinterpret_cast<const char*>(p.w);
    const char* const zrow = reinterpret_cast<const char*>(g_zero_row16);
    int l_t = blockIdx.x, l_hk = 0, l_slot = 0;
    int l_tap_ky = 0, l_tap_kx = 0, l_c0 = 0;                                // conv: the tap of half-stage l_hk

    // tile t -> (problem, tile_m, tile_n).  XCD-aware order: blocks b, b+8, ... share an XCD; each XCD gets a
    // contiguous run of tiles, column tile fastest, so its 32 CUs share A row panels and sweep W together.
    auto tile_of = [&](int t, int& g, int& tile_m, int& tile_n) {
        int bid = t;
        {
            const int q = num_tiles >> 3, r = num_tiles & 7, xcd = bid & 7;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        }
        // (uniform integer division is expanded on the vector ALU: without the readfirstlane the whole loader state --
        //  tile, k position, operand bases -- ends up in VGPRs behind divergent branches)
        g = 0;
        if constexpr (GROUPED) {
            g = __builtin_amdgcn_readfirstlane(bid / tiles_per_problem);
            bid -= g * tiles_per_problem;
        }
        tile_m = __builtin_amdgcn_readfirstlane(bid / tiles_n);
        tile_n = bid - tile_m * tiles_n;
    };
    auto loader_setup = [&](int t) {
        int g, tile_m, tile_n;
        tile_of(t, g, tile_m, tile_n);
        if constexpr (GROUPED) {
            l_a8 = uniform_ptr(reinterpret_cast<const char*>(pick4(grp.a, g)));
            l_w8 = uniform_ptr(reinterpret_cast<const char*>(pick4(grp.w, g)));
        }
        const int m0 = tile_m * TM, n0 = tile_n * TN;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            int m = m0 + (wave * PA + i) * 16 + prow;
            m = m < p.M ? m : p.M - 1;                              // edge rows are loaded, never stored
            if constexpr (CONV) {
                const int hw = p.Ho * p.Wo;
                const int img = m / hw, rem = m - img * hw;
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                l_aoff[i] = (unsigned)(((int64_t)img * p.H * p.W * p.C) * 2 + psw);
                l_yx0[i] = ((unsigned)(oy * p.stride - p.pad + 1) << 16) | (unsigned)(ox * p.stride - p.pad + 1);
            } else {
                l_aoff[i] = (unsigned)((int64_t)m * p.lda * 2 + psw);
                l_yx0[i] = 0;
            }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            int n = n0 + (wave * PB + i) * 16 + prow;
            n = n < p.N ? n : p.N - 1;
            l_boff[i] = (unsigned)((int64_t)n * p.ldw * 2 + psw);
        }
    };
    // piece `i` (0 .. NP-1: the A pieces, then the W pieces) of the loader's current STAGE (64 k): two wave-instructions
    // back to back, the k-half 0 of 16 rows into slot l_slot and the k-half 1 of the same rows into the next slot.  The
    // second instruction hits the 128-byte lines the first one requested: measured 106-117 GB/s per CU against 74-81 when
    // the two halves of a line are requested a half-stage apart (every line then crosses the L2 -> L1 fabric twice) and
    // 116-126 for whole 128-byte rows (tools/hw_probe/dma_rate.hip).
    auto loader_instr = [&](int n) {                      // n = 2 * piece + k-half
        const int i = n >> 1, half = n & 1;
        const int sl = half ? (l_slot + 1 == NS ? 0 : l_slot + 1) : l_slot;
        const unsigned slot = lds0 + (unsigned)sl * SLOT;
        const unsigned kb = ((GRL_RING_KO & 8) ? 0u : (unsigned)l_hk * 64) + (unsigned)half * 64;
        if (i < PA) {
            const unsigned off = (unsigned)(wave * PA + i) * 1024;
            if constexpr (CONV) {
                const int iy = (int)(l_yx0[i] >> 16) - 1 + l_tap_ky, ix = (int)(l_yx0[i] & 0xffff) - 1 + l_tap_kx;
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const char* src = ok ? l_a8 + (l_aoff[i] + (unsigned)(((iy * p.W + ix) * p.C + l_c0) * 2 + half * 64)) : zrow + psw;
                dma16_v(src, slot + off);
            } else {
                dma16(l_a8 + kb, l_aoff[i], slot + off);
            }
        } else {
            const int j = i - PA;
            dma16(l_w8 + kb, l_boff[j], slot + TM * 64 + (unsigned)(wave * PB + j) * 1024);
        }
    };
    // after the last piece of a stage: next stage (two half-stages, two slots), next tile
    auto loader_advance = [&]() {
        l_slot = l_slot + 2 >= NS ? l_slot + 2 - NS : l_slot + 2;
        l_hk += 2;
        if constexpr (CONV) {
            l_c0 += 64;
            if (l_c0 == p.C) {
                l_c0 = 0;
                if (++l_tap_kx == p.kw) { l_tap_kx = 0; ++l_tap_ky; }
            }
        }
        if (l_hk == nkh) {                                   // next tile (past the last one nothing is issued: `issued`)
            l_hk = 0;
            if constexpr (CONV) l_tap_ky = l_tap_kx = l_c0 = 0;
            l_t += (int)gridDim.x;
            if (l_t < num_tiles) loader_setup(l_t);
        }
    };

    // ------------------------------------------------------------------ consumer geometry
    const int frow = lane & 31, fhalf = lane >> 5;
    const unsigned fch0 = (unsigned)((fhalf ^ ((frow >> 2) & 3)) << 4);           // k-step 0 chunk; k-step 1 = ^ 32
    const unsigned a_lane = (unsigned)((wr * WM + frow) * 64) + fch0;
    const unsigned b_lane = (unsigned)(TM * 64 + (wc * 64 + frow) * 64) + fch0;
    const int lrow = lane >> 3, lcol = (lane & 7) * 8;                        // slab read-back: 8 lanes per 64-wide row

    // ------------------------------------------------------------------ prologue: two stages (four half-stages) in flight
    // Slot of half-stage h = h % 5.  Stage (2s, 2s+1) needs the slots of half-stages 2s-5 and 2s-4: both free once
    // half-stage 2s-4 has been consumed, i.e. the stage is issued in the iteration that consumes half-stage 2s-3 (odd).
    static_assert(NS == 5, "the pair schedule below is written for five slots");
    int issued = 0;                                          // half-stages whose stage the loader has opened (even)
    loader_setup(l_t);
#pragma unroll 1
    for (int d = 0; d < 2; ++d) {
        if (issued < total) {
#pragma unroll
            for (int n = 0; n < 2 * NP; ++n) loader_instr(n);
            loader_advance();
            issued += 2;
        }
    }
    if (issued > 2) wait_vm<2 * NP>(); else wait_vm<0>();    // stage 0 has landed (stage 1 may be in flight)
    __builtin_amdgcn_s_barrier();

    // Phase offset between workgroups.  Every workgroup starts at the same time with the same amount of work per tile, so
    // all 256 CUs run their main loops together (HBM idle) and their epilogues together (HBM saturated: 256 KiB of residual
    // reads and output stores per tile and CU).  Workgroups sleep p/8 of `phase_cycles` (the host's estimate of one
    // tile's main loop) before their first tile, p = their index within the XCD mod 8: afterwards some CUs are always in
    // their epilogue while the others compute.
    if (phase_cycles > 0) {
        const int ph = ((int)blockIdx.x >> 3) & 7;
        for (int c = ph * (phase_cycles >> 3); c > 0; c -= 64 * 100) __builtin_amdgcn_s_sleep(100);
    }
    bool l_open = false;                                     // the stage opened last exists (its chunk B follows one half-stage later)
    int c_slot = 0;
    for (int t = blockIdx.x; t < num_tiles; t += (int)gridDim.x) {
        f32x16 acc[MI][2];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        bf16x8 af[2][MI], bf[2][2];
        if (GRL_RING_KO & 16) {
#pragma unroll
            for (int z = 0; z < 2; ++z) {
#pragma unroll
                for (int i = 0; i < MI; ++i) asm volatile("" : "=v"(af[z][i]));
                asm volatile("" : "=v"(bf[z][0]));
                asm volatile("" : "=v"(bf[z][1]));
            }
        }
#define RING_RD(set, aa_, bb_)                                                                              \
        do {                                                                                            \
            if (GRL_RING_KO & 16) break;                                                                \
            asm volatile("ds_read_b128 %0, %1" : "=v"(af[set][0]) : "v"(aa_));                          \
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(af[set][1]) : "v"(aa_));              \
            if constexpr (MI == 4) {                                                                    \
                asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(af[set][2]) : "v"(aa_));          \
                asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(af[set][3]) : "v"(aa_));          \
            }                                                                                           \
            asm volatile("ds_read_b128 %0, %1" : "=v"(bf[set][0]) : "v"(bb_));                          \
            asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(bf[set][1]) : "v"(bb_));              \
        } while (0)
#if GRL_RING_M16
#define RING_MFMA(set, i_, j_)                                                                              \
        do {                                                                                            \
            f32x4 q0_ = __builtin_shufflevector(acc[i_][j_], acc[i_][j_], 8 * (set) + 0, 8 * (set) + 1, 8 * (set) + 2, 8 * (set) + 3); \
            f32x4 q1_ = __builtin_shufflevector(acc[i_][j_], acc[i_][j_], 8 * (set) + 4, 8 * (set) + 5, 8 * (set) + 6, 8 * (set) + 7); \
            q0_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[set][i_], bf[set][j_], q0_, 0, 0, 0);     \
            q1_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[set][i_], bf[set][j_], q1_, 0, 0, 0);     \
            _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) { acc[i_][j_][8 * (set) + e_] = q0_[e_]; acc[i_][j_][8 * (set) + 4 + e_] = q1_[e_]; } \
        } while (0)
#else
#define RING_MFMA(set, i_, j_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[set][i_], bf[set][j_], acc[i_][j_], 0, 0, 0)
#endif
#define RING_MM(set, i0, i1)                                                                                \
        do {                                                                                            \
            if (!(GRL_RING_KO & 4)) {                                                                   \
                _Pragma("unroll") for (int i_ = (i0); i_ < (i1); ++i_)                                  \
                    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                    \
                        RING_MFMA(set, i_, j_);                                                     \
            }                                                                                           \
        } while (0)
#define RING_LGKM(n)                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory");                                         \
        __builtin_amdgcn_sched_barrier(0)
        constexpr int NRD = MI + 2;                          // ds_reads per k-step

        {   // first fragments of the tile (the slot landed at the barrier that ended the previous tile's loop / the prologue)
            const unsigned sb = lds0 + (unsigned)c_slot * SLOT;
            const unsigned aa = sb + a_lane, bb = sb + b_lane;
            RING_RD(0, aa, bb);
        }
        // One half-stage.  PAR: parity of g (odd half-stages open a stage); OPEN: a stage is being issued during this
        // pair of half-stages.  What may stay in flight at the barrier (the NEXT half-stage's stage must have landed) is a
        // compile-time count: this half-stage's instructions so far, plus the odd half-stage's NP when this one is even.
        auto half_stage = [&](auto par_, const bool last_of_tile) {
            constexpr bool PAR = decltype(par_)::value;
            const unsigned sb = lds0 + (unsigned)c_slot * SLOT;
            const unsigned aa1 = (sb + a_lane) ^ 32u, bb1 = (sb + b_lane) ^ 32u;
            // This wave's DMA window of the half-stage (see SCHED): chunk A = the stage's first NP instructions (and the
            // decision whether there is a stage to open), chunk B = the rest, one half-stage later.
            const bool phase_a = (PAR != late);
            auto window = [&](int idx) {
                if (idx == win) {
                    if (phase_a) {
                        l_open = !(GRL_RING_KO & 1) && issued < total;
                        if (l_open) {
                            issued += 2;
#pragma unroll
                            for (int n = 0; n < NP; ++n) loader_instr(n);
                        }
                    } else if (l_open) {
#pragma unroll
                        for (int n = NP; n < 2 * NP; ++n) loader_instr(n);
                        loader_advance();
                    }
                }
            };
            RING_RD(1, aa1, bb1);
            window(0);
            if constexpr (NRD == 6) { RING_LGKM(6); } else { RING_LGKM(4); }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (i) window(i);
                RING_MM(0, i, i + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            RING_LGKM(0);                                    // every read of this slot has returned
            c_slot = c_slot + 1 == NS ? 0 : c_slot + 1;
            if (l_open) wait_vm<(PAR ? NP : 2 * NP)>(); else wait_vm<0>();
            __builtin_amdgcn_s_barrier();                    // slot g+1 landed for everyone; slot g is free
            __builtin_amdgcn_sched_barrier(0);
            if (!last_of_tile) {
                const unsigned sn = lds0 + (unsigned)c_slot * SLOT;
                const unsigned aa = sn + a_lane, bb = sn + b_lane;
                RING_RD(0, aa, bb);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                window(MI + i);
                RING_MM(1, i, i + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#pragma unroll 1
        for (int hk = 0; hk < nkh; hk += 2) {
            half_stage(std::false_type{}, false);
            half_stage(std::true_type{}, hk + 2 >= nkh);
        }
#undef RING_RD
#undef RING_MM
#undef RING_MFMA
#undef RING_LGKM
        if (GRL_RING_KO & 2) {
            float z = 0.f;                                   // (every accumulator stays live: no MFMA may be dropped)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) z += acc[i][j][r];
            if (z == 123.456f) reinterpret_cast<float*>(p.y)[0] = z;
            continue;
        }

        // ---- epilogue.  acc[i][j][r] is Y[row][col], row = (r&3) + 8*(r>>2) + 4*fhalf, col = lane&31.
        // The slab (16 rows x 64 fp32 per wave) is this wave's DMA footprint in the slot consumed last: free since the
        // loop's last barrier, and refilled only by this wave's own pieces, in program order behind this epilogue.
        int gi, tile_m, tile_n;
        tile_of(t, gi, tile_m, tile_n);
        const int m0 = tile_m * TM, n0 = tile_n * TN;
        const int cm0 = m0 + wr * WM, cn = n0 + wc * 64 + lcol;
        const int stat_row = (TM / 128) * tile_m + (wr * WM) / 128;      // STATS: one slab row per 128 rows
        float* slab_lo;
        float* slab_hi;
        {
            const int prev = c_slot == 0 ? NS - 1 : c_slot - 1;      // the slot consumed last
            if constexpr (SLAB_ALIAS) {
                slab_lo = reinterpret_cast<float*>(smem + prev * SLOT + wave * 2048);
                slab_hi = reinterpret_cast<float*>(smem + prev * SLOT + TM * 64 + wave * 2048);
            } else {
                slab_lo = reinterpret_cast<float*>(smem + NS * SLOT + wave * 4096);
                slab_hi = slab_lo + 8 * 64;
            }
        }
        __bf16* y16 = reinterpret_cast<__bf16*>(p.y);
        const __bf16* r16 = reinterpret_cast<const __bf16*>(p.res);
        const float* scale = p.scale;
        const float* shift = p.shift;
        if constexpr (GROUPED) {
            y16 = reinterpret_cast<__bf16*>(pick4(grp.y, gi));
            r16 = reinterpret_cast<const __bf16*>(pick4(grp.res, gi));
            scale = pick4(grp.scale, gi);
            shift = pick4(grp.shift, gi);
        }
        float* const y32 = reinterpret_cast<float*>(y16);
        // Two copies of the epilogue.  INTERIOR (the wave's 128 x 64 block lies inside the matrix: every tile but the
        // edge ones) has no branch at all between the first residual request and the last store: hipcc can then count
        // its loads and stores.  With the per-row `m < M` / null-pointer tests in the way it emitted `s_waitcnt vmcnt(0)`
        // in front of EVERY load and store (each store waited for the previous one's acknowledge: ~15 us per tile, 40 %
        // of the short-K layers' time, in this kernel and in gemm_bf16_256_kernel alike).
        auto epilogue = [&](auto interior_) {
            constexpr bool INT = decltype(interior_)::value;
            const bool n_ok = INT || cn < p.N;
            f32x4 sc[2] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}}, sh[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (n_ok) {
                if (scale) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) sc[u] = *reinterpret_cast<const f32x4*>(scale + cn + 4 * u);
                }
                if (shift) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) sh[u] = *reinterpret_cast<const f32x4*>(shift + cn + 4 * u);
                }
            }
            float relu_floor;      // 0 (ReLU) or a quiet NaN (no ReLU: v_max returns the other operand, a NaN accumulator stays NaN);
            {                       // through an asm move: told the constant, hipcc folds max(t, NaN) into a select per element
                const uint32_t floor_bits = p.relu ? 0u : 0x7fc00000u;
                asm("v_mov_b32 %0, %1" : "=v"(relu_floor) : "s"(floor_bits));
            }
            // residual rows: two 32-row blocks (8 x 1 KiB per wave) in flight -- block i+1 is requested before block i
            // goes through the slab
            bf16x8 res8[2][4];
            auto res_request = [&](int i) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int m = cm0 + i * 32 + it * 8 + lrow;
                    int64_t rr = m;
                    if constexpr (SQD) rr = (int64_t)(m / p.res_rows) * p.res_gstride + (m % p.res_rows);
                    if (INT || (m < p.M && n_ok)) res8[i & 1][it] = *reinterpret_cast<const bf16x8*>(r16 + rr * p.ldres + cn);
                }
            };
            if constexpr (RES) res_request(0);
            f32x4 ssum[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, ssq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if constexpr (RES) { if (i + 1 < MI) res_request(i + 1); }
                f32x4 part[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};        // SQD: per 32-row block
#pragma unroll
                for (int q = 0; q < 2; ++q) {                        // 16-row half of the 32-row block
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r8 = 0; r8 < 8; ++r8) {
                            const int r = q * 8 + r8;
                            const int row = (r & 3) + 4 * fhalf;              // row inside the 8-row half-slab
                            float* const half = ((r >> 2) & 1) ? slab_hi : slab_lo;
                            half[row * 64 + ((j * 32 + frow) ^ (((row >> 1) & 1) << 2))] = acc[i][j][r];
                        }
                    // (the same wave wrote and reads the slab: a wave's LDS operations complete in order)
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps) {
                        const int it = q * 2 + ps;
                        const float* const half = ps ? slab_hi : slab_lo;
                        const int m = cm0 + i * 32 + it * 8 + lrow;
                        f32x4 v[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            v[u] = *reinterpret_cast<const f32x4*>(half + lrow * 64 + ((lcol + 4 * u) ^ (((lrow >> 1) & 1) << 2)));
                        if constexpr (SQD) {
                            if (n_ok) {
#pragma unroll
                                for (int u = 0; u < 2; ++u) {
                                    f32x4 w_ = v[u] * sc[u] + sh[u];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const float f1v = (float)(__bf16)(w_[e] > 0.f ? w_[e] : 0.f);
                                        const float dd = f1v - (float)res8[i & 1][it][4 * u + e];
                                        part[u][e] += dd * dd;
                                    }
                                }
                            }
                        } else {
                            if (INT || (m < p.M && n_ok)) {
                                f32x8 o32;
#pragma unroll
                                for (int u = 0; u < 2; ++u) {
                                    f32x4 w_ = v[u];
                                    if constexpr (GBIAS)
                                        w_ += *reinterpret_cast<const f32x4*>(p.gbias + (int64_t)(m / p.rows_per_group) * p.N + cn + 4 * u);
                                    if constexpr (STATS) { ssum[u] += w_; ssq[u] += w_ * w_; }
                                    w_ = w_ * sc[u] + sh[u];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        float tt = w_[e];
                                        if constexpr (RES) tt = tt + (float)res8[i & 1][it][4 * u + e];
                                        else tt = tt + 0.f;
                                        o32[4 * u + e] = __builtin_fmaxf(tt, relu_floor);   // (ReLU: one v_max against 0 / NaN = identity, NaN-preserving)
                                    }
                                }
                                *reinterpret_cast<bf16x8*>(y16 + (int64_t)m * p.ldy + cn) = __builtin_convertvector(o32, bf16x8);
                            }
                        }
                    }
                }
                if constexpr (SQD) {
#pragma unroll
                    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int e = 0; e < 4; ++e) part[u][e] += __shfl_xor(part[u][e], o);
                    if (lrow == 0 && n_ok) {
                        float* const yq = y32 + (int64_t)((cm0 + i * 32) >> 5) * p.ldy + cn;
                        *reinterpret_cast<f32x4*>(yq) = part[0];
                        *reinterpret_cast<f32x4*>(yq + 4) = part[1];
                    }
                }
            }
            if constexpr (STATS) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            ssum[u][e] += __shfl_xor(ssum[u][e], o);
                            ssq[u][e] += __shfl_xor(ssq[u][e], o);
                        }
                }
                if (lrow == 0 && n_ok) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        *reinterpret_cast<f32x4*>(p.stats + ((int64_t)stat_row * 2 + 0) * p.N + cn + 4 * u) = ssum[u];
                        *reinterpret_cast<f32x4*>(p.stats + ((int64_t)stat_row * 2 + 1) * p.N + cn + 4 * u) = ssq[u];
                    }
                }
            }
        };
        const bool interior = cm0 + WM <= p.M && n0 + wc * 64 + 64 <= p.N;       // (wave-uniform)
        if (interior) epilogue(std::true_type{});
        else epilogue(std::false_type{});
    }
    wait_vm<0>();                                            // nothing of the ring may land in LDS after the wave has left
}

template <int TM, int TN, int NS, bool CONV, bool STATS, bool SQD, bool GROUPED, bool RES = false, bool GBIAS = false>
int launch_one(const GrlGemm& d, const Group& grp, hipStream_t s) {
    constexpr int SLOT = (TM + TN) * 64;
    constexpr bool alias = (NS * SLOT + 8 * 4096) > 160 * 1024;
    constexpr int lds = NS * SLOT + (alias ? 0 : 8 * 4096);
    auto kern = gemm_bf16_ring_kernel<TM, TN, NS, CONV, STATS, SQD, GROUPED, RES, GBIAS>;
    static const bool attr = [&] {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        return true;
    }();
    (void)attr;
    static const int cus = [] {                  // persistent grid: one 8-wave workgroup per CU, a multiple of 8
        int dev = 0, n = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n = prop.multiProcessorCount;
        return n / 8 * 8 > 0 ? n / 8 * 8 : 8;
    }();
    const int tiles_m = (d.M + TM - 1) / TM, tiles_n = (d.N + TN - 1) / TN;
    const int per = tiles_m * tiles_n;
    const int num_tiles = per * (GROUPED ? grp.n : 1);
    const unsigned grid = (unsigned)(num_tiles < cus ? num_tiles : cus);
    // phase offset (see the kernel): only when a workgroup walks several tiles whose epilogue moves a tile of HBM traffic
    static const int phase_env = [] { const char* e = getenv("GRL_RING_PHASE"); return e ? atoi(e) : -1; }();
    int phase = 0;
    if (num_tiles >= 2 * cus) phase = phase_env >= 0 ? phase_env * (d.K / 32) : 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, d, grp, tiles_n, per, num_tiles, phase);
    return grl_check_launch("grl_conv_gemm_f32 (bf16 ring)");
}

// what the ring kernels need from a launch (the dispatcher in gemm_bf16.hip asks before it routes one here)
inline bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }
inline bool shape_ok(const GrlGemm& d) {
    if (d.math != GRL_MATH_BF16S || d.rowscale || d.out_f32) return false;
    if (d.epilogue != GRL_EPI_AFFINE && d.epilogue != GRL_EPI_SQDIFF) return false;
    if (d.K % 64 || d.N % 8 || d.ldy % (d.epilogue == GRL_EPI_SQDIFF ? 4 : 8) || (d.res && d.ldres % 8) || d.ldw % 8) return false;
    if (d.conv ? (d.C % 64 != 0 || d.H + 2 > 65535 || d.W + 2 > 65535) : d.lda % 8 != 0) return false;
    if (!al16(d.a) || !al16(d.w) || !al16(d.y) || !al16(d.res) || !al16(d.scale) || !al16(d.shift) || !al16(d.gbias) || !al16(d.stats))
        return false;
    const int64_t a_bytes = d.conv ? (int64_t)(d.M / (d.Ho * d.Wo)) * d.H * d.W * d.C * 2 : (int64_t)d.M * d.lda * 2;
    if (a_bytes >= (1ll << 32) || (int64_t)d.N * d.ldw * 2 >= (1ll << 32)) return false;       // 32-bit lane offsets
    if (d.epilogue == GRL_EPI_SQDIFF && (d.conv || d.stats || d.gbias || !d.res || d.res_rows <= 0 || d.res_rows % 32 || d.M % 32)) return false;
    return true;
}

template <int TM, int TN, int NS>
int launch_tile(const GrlGemm& d, const Group& grp, hipStream_t s) {
    const bool grouped = grp.n > 1;
    const bool res = grouped ? grp.res[0] != nullptr : d.res != nullptr;
    if (d.epilogue == GRL_EPI_SQDIFF) {
        if (d.M % TM || d.N % TN) return 1;
        return grouped ? launch_one<TM, TN, NS, false, false, true, true, true>(d, grp, s) : launch_one<TM, TN, NS, false, false, true, false, true>(d, grp, s);
    }
    if (d.stats) {                                           // train forward: BatchNorm statistics (no residual there)
        if constexpr (TM == 256 && TN == 256) {
            if (grouped || res) return 1;
            if (d.conv) return d.gbias ? 1 : launch_one<TM, TN, NS, true, true, false, false>(d, grp, s);
            return d.gbias ? launch_one<TM, TN, NS, false, true, false, false, false, true>(d, grp, s)
                           : launch_one<TM, TN, NS, false, true, false, false>(d, grp, s);
        }
        return 1;
    }
    if (d.conv) {
        if (grouped || d.gbias) return 1;
        return res ? launch_one<TM, TN, NS, true, false, false, false, true>(d, grp, s) : launch_one<TM, TN, NS, true, false, false, false>(d, grp, s);
    }
    if (grouped) {
        if (d.gbias) return 1;
        return res ? launch_one<TM, TN, NS, false, false, false, true, true>(d, grp, s) : launch_one<TM, TN, NS, false, false, false, true>(d, grp, s);
    }
    if (d.gbias) return res ? 1 : launch_one<TM, TN, NS, false, false, false, false, false, true>(d, grp, s);
    return res ? launch_one<TM, TN, NS, false, false, false, false, true>(d, grp, s) : launch_one<TM, TN, NS, false, false, false, false>(d, grp, s);
}

// variant: 1 = 256 x 256, 2 = 256 x 128 (five ring slots each).
// 0 = launched, 1 = not covered, < 0 = error
int launch_variant(const GrlGemm& d, const Group& grp, hipStream_t s, int variant) {
    if (!shape_ok(d)) return 1;
    switch (variant) {
        case 0: return 1;
        case 1: return launch_tile<256, 256, 5>(d, grp, s);
        case 2: return launch_tile<256, 128, 5>(d, grp, s);
    }
    return 1;
}

}  // namespace ring

#ifndef GRL_RING_NO_CABI
// ---------------------------------------------------------------------------------------------------------------
// C ABI: several GEMMs of ONE shape in one launch (include/grl_hip.h).  The two directions of a TRL step run the same
// 8192-row GEMMs on different operands (grl_model.py:131-167): M = 8192 gives 64-128 tiles of the large bf16 tiles --
// half a chip -- and two launches on two streams cannot share a CU (one workgroup owns its LDS).  Grouped, they are one
// launch of 256 x 128 tiles that fills the chip: 8192 x 512 x 2048 x 2: 2 x 35 us -> 44 us.
extern "C" int grl_conv_gemm_f32_group(const GrlGemm* descs, int n, void* stream) {
    if (!descs || n < 1 || n > 4) return grl_fail(GRL_EINVAL, "gemm_group: 1..4 descriptors");
    // every descriptor passes the single-launch checks BEFORE anything is grouped (the grouped path used to
    // reach the DMA loader with null operands or K == 0; the fallback path validated, so the two accepted different inputs)
    for (int g = 0; g < n; ++g)
        if (const int e = grl_gemm_validate(descs[g])) return e;
    const GrlGemm& d0 = descs[0];
    bool same = true;
    for (int g = 1; g < n; ++g) {
        const GrlGemm& d = descs[g];
        same = same && d.M == d0.M && d.N == d0.N && d.K == d0.K && d.lda == d0.lda && d.ldw == d0.ldw && d.ldy == d0.ldy &&
               d.ldres == d0.ldres && d.relu == d0.relu && d.epilogue == d0.epilogue && d.math == d0.math && d.conv == d0.conv &&
               d.out_f32 == d0.out_f32 && (d.res != nullptr) == (d0.res != nullptr) && (d.scale != nullptr) == (d0.scale != nullptr) &&
               (d.shift != nullptr) == (d0.shift != nullptr);
    }
    bool plain = true;                      // what the grouped kernel covers: dense, affine, no optional operand but res
    for (int g = 0; g < n; ++g) {
        const GrlGemm& d = descs[g];
        plain = plain && ring::shape_ok(d) && !d.conv && d.epilogue == GRL_EPI_AFFINE && !d.stats && !d.gbias && !d.bn_z && !d.kblock;
    }
    static const bool group_on = [] { const char* e = getenv("GRL_GEMM_GROUP"); return !e || atoi(e) != 0; }();
    if (n >= 2 && same && plain && group_on && d0.scale && d0.shift) {
        ring::Group grp;
        grp.n = n;
        for (int g = 0; g < 4; ++g) {
            const GrlGemm& d = descs[g < n ? g : 0];
            grp.a[g] = d.a; grp.w[g] = d.w; grp.y[g] = d.y; grp.scale[g] = d.scale; grp.shift[g] = d.shift; grp.res[g] = d.res;
        }
        // tiles: 256 x 128 while 256 x 256 tiles would leave CUs without one
        const int64_t t256 = (int64_t)((d0.M + 255) / 256) * ((d0.N + 255) / 256) * n;
        const int rc = t256 < 256 ? ring::launch_tile<256, 128, 5>(d0, grp, (hipStream_t)stream)
                                  : ring::launch_tile<256, 256, 5>(d0, grp, (hipStream_t)stream);
        if (rc <= 0) return rc;                              // launched (0) or failed (< 0); 1 = not covered
    }
    for (int g = 0; g < n; ++g)
        if (const int rc = grl_conv_gemm_f32(&descs[g], stream)) return rc;
    return GRL_OK;
}
#endif  // GRL_RING_NO_CABI
