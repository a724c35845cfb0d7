// fp32 MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
//   Y[M][N] = epilogue( A[M][K] . W[N][K]^T )
//
// A is either a dense K-contiguous matrix or gathered on the fly from a
// channels-last image tensor (1x1 / 3x3, stride 1 / 2, zero padding).  Both
// operands are K-contiguous, so a lane can fetch FOUR consecutive k of its row
// with one ds_read_b128 and feed four v_mfma_f32_32x32x2_f32 from it: the k a
// lane half supplies to one MFMA is arbitrary as long as A and B agree, so lane
// half h takes k = 8q + 4h + s at step s of 8-wide chunk q.
//
// Accumulation order (documented for the bit-exact oracle): for K-stage j
// (32 wide), chunk q = 0..3, step s = 0..3 the accumulator receives
//   acc = fma(a[k0], b[k0], acc);  acc = fma(a[k1], b[k1], acc)
// with k0 = 32j + 8q + s, k1 = k0 + 4.  With GrlGemm.kblock the chain is cut every SEG_STAGES
// stages (512 k): at a segment boundary that is not the end of K the accumulator is added to a
// running total and restarts from zero, and the result is last_segment + total.  The rounding
// error of a sequential fp32 chain grows like sqrt(K); with K up to 4608 on the training path the
// blocked form is ~3x closer to the exact sum -- the accuracy class of a blocked CPU sgemm -- which
// keeps train-mode ReLU masks (and with them the parameter gradients) on the reference's side of
// zero.  It costs ~4 % on K >= 1024 (64 more VGPRs, a flush per segment), so the eval path and the
// evaluator keep the single chain.
//
// Tile: BM x BN x 32 per workgroup of 4 waves (2 x 2), each wave
// (BM/2) x (BN/2) as MT x NT MFMA tiles of 32 x 32.  LDS holds two stages of
// A and B tiles as [row][32 floats] with the 16-byte chunk index XOR-swizzled by
// (row >> 1) & 7, which makes both the ds_write_b128 of the staging pass and
// the ds_read_b128 of the fragment reads bank-conflict free.
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/grl_hip.h"
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BK = 32;
#ifndef GRL_GEMM_KO
#define GRL_GEMM_KO 0     // tools/gemm_ko.sh: timing-only knock-outs (1 no in-loop staging, 2 no stage barrier, 4 no epilogue, 8 all stages from k = 0); wrong results
#endif
constexpr int SEG_STAGES = 16;     // fp32 path: accumulator segment = 16 stages = 512 k
#ifndef GRL_GEMM_RING3
#define GRL_GEMM_RING3 1  // bf16-storage 128 x 64 LDS-DMA kernel: three stage buffers, DMA two stages ahead (0: compiler-scheduled two-stage loop)
#endif
#ifndef GRL_GEMM_PIPE
#define GRL_GEMM_PIPE 1   // hand-scheduled stage loop of the dense fp32 LDS-DMA kernels (0: the compiler-scheduled loop)
#endif

typedef __attribute__((address_space(3))) char* lds_cptr_t;
// LDS-DMA piece the compiler does not see (wave-uniform 64-bit base in SGPRs + per-lane 32-bit offset).  While a
// compiler-VISIBLE global_load_lds is outstanding hipcc waits vmcnt(0) in front of every LDS read that follows (it cannot
// tell the DMA's destination from the buffer being read), which forces all of a stage's pieces to the top of the stage
// and every fragment read behind a drained queue.  Hidden, the pieces sit between the MFMAs and the loop waits for them
// itself (`s_waitcnt vmcnt(0)` in front of the stage barrier); counted waits the compiler emits for its own loads only
// ever over-wait.
// ("m0" is on the clobber list: the statement overwrites it, and the compiler keeps its own LDS-DMA / indexing state there.
//  hipcc accepts the clobber with a -Winline-asm note about reserved registers, silenced for these statements only.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16_hidden(const char* sbase, uint32_t voff, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void dma16_hidden_v(const char* vaddr, uint32_t lds) {       // per-lane 64-bit address (the conv gather)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(vaddr), "s"(lds) : "memory", "m0");
}
#pragma clang diagnostic pop

__device__ uint4 g_zero_chunks[8];        // 128 zero bytes: LDS-DMA source of out-of-image conv taps

struct RowInfo {          // per staged A row: where it comes from
    int64_t base;         // dense: m*lda ; conv: image base offset (img*H*W*C)
    int iy0, ix0;         // conv: top-left input coordinate of the window
};

// MATH selects the multiplier datapath (the accumulator is always fp32):
//   0  exact fp32: v_mfma_f32_32x32x2_f32 (the documented fmaf chain; default)
//   1  bf16 operands (rounded to nearest-even while staging), v_mfma_f32_32x32x16_bf16
//   3  split-bf16 ("bf16x3"): x = hi + lo with hi = bf16(x), lo = bf16(x - hi); the product
//      is hi*hi + hi*lo + lo*hi on the bf16 MFMA (3/16 of the fp32 MFMA cycles); the
//      dropped lo*lo term and the 16-bit operand significands give ~2^-16 relative error
//      per product -- fp32-class accuracy for the 1e-3 parity budget, NOT bit-exact.
//   2  bf16 STORAGE ("bf16s"): activations, weights, residual and output are bf16 in HBM
//      (half the bytes), products on v_mfma_f32_32x32x16_bf16, fp32 accumulate and fp32
//      epilogue math.  A stage is 64 k wide, i.e. the same 128-byte rows, staging and LDS
//      swizzle as the fp32 path; one ds_read_b128 (8 consecutive k) feeds one MFMA.
// In modes 1/3 operands stay fp32 in HBM and are converted in the staging pass; LDS
// rows hold 32 bf16 (64 B) with the 16-byte chunk XOR-swizzled by (row >> 2) & 3.
// SEG: cut the fp32 accumulation chain every SEG_STAGES stages (launched when K > 512; K <= 512
// layers run the instantiation without the second accumulator set).
// SPLITK (skinny K-blocked GEMMs, GrlGemm.splitk_ws): blockIdx.y is a 512-k segment of the K-blocked chain; the
// workgroup runs that segment as a plain chain from zero and stores the raw accumulator to ws[segment][M][N].
template <int BM, int BN, bool CONV, int MATH, bool SEG, bool DMA = false, bool SPLITK = false>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GrlGemm p_in, const int tiles_n,
                                                           const int num_tiles, const int vec_epi) {
    GrlGemm p_seg;
    if constexpr (SPLITK) {
        p_seg = p_in;
        const int seg = blockIdx.y, kseg = SEG_STAGES * BK;
        p_seg.a = p_in.a + (int64_t)seg * kseg;
        p_seg.w = p_in.w + (int64_t)seg * kseg;
        p_seg.K = min(kseg, p_in.K - seg * kseg);
        p_seg.y = p_in.splitk_ws + (int64_t)seg * p_in.M * p_in.N;
        p_seg.ldy = p_in.N;
        p_seg.scale = p_seg.shift = p_seg.res = p_seg.gbias = p_seg.rowscale = nullptr;
        p_seg.stats = nullptr;
        p_seg.relu = 0;
    }
    const GrlGemm& p = SPLITK ? p_seg : p_in;
    constexpr int WTM = BM / 2, WTN = BN / 2;     // wave tile
    constexpr int MT = WTM / 32, NT = WTN / 32;   // MFMA tiles per wave
    constexpr int ESZ = MATH == 2 ? 2 : 4;        // bytes per operand element in HBM
    constexpr int EPC = 16 / ESZ;                 // elements per 16-byte chunk
    constexpr int KST = 8 * EPC;                  // k per stage: 128-byte rows (32 fp32 / 64 bf16)
    auto gp = [](const float* base, int64_t elem_off) {       // 16-byte chunk at an ELEMENT offset
        return reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + elem_off * ESZ);
    };
    constexpr int A_ITEMS = BM / 32, B_ITEMS = BN / 32;   // 16-B items per thread per stage
    // RING (bf16 storage, 128 x 64): a stage of that kernel is 8 MFMAs of 32 cycles per wave -- 0.1-0.2 us -- behind a DMA
    // one stage deep: it waited for memory every stage (10-14 % MFMA busy; the hand-scheduled two-stage loop alone: +-0).
    // Three 24 KB stage buffers (72 KB: still two workgroups per CU), pieces requested TWO stages ahead, counted
    // `vmcnt(NP)` at the stage barrier (the newest stage's pieces may still be in flight).
    constexpr bool RING = GRL_GEMM_PIPE && GRL_GEMM_RING3 && DMA && MATH == 2 && BM == 128 && BN == 64;
    constexpr int NST = RING ? 3 : 2;
    constexpr bool PIPE = (GRL_GEMM_PIPE && DMA && MATH == 0 && BM == 128) || RING;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                     // [2][BM*32]           (MATH == 0)
    float* Bs = smem + NST * BM * BK;     // [NST][BN*32]
    constexpr int PL = MATH == 3 ? 2 : 1; // bf16 planes per operand (hi, lo)
    char* const Ah = reinterpret_cast<char*>(smem);                // [2][PL][BM][64 B]
    char* const Bh = Ah + 2 * PL * BM * 64;                        // [2][PL][BN][64 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = PIPE ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ld_row = tid >> 3, ld_chunk = tid & 7;
    const int d_row = lane >> 3, d_chunk = lane & 7;          // LDS-DMA staging: lane = (row of an 8-row block, chunk)

    // Persistent workgroups: the grid holds as many workgroups as the chip keeps resident and
    // each walks tiles t = blockIdx.x, + gridDim.x, ...  The next tile's first K stage is
    // requested from HBM BEFORE the current tile's epilogue runs, so a tile no longer starts
    // with an exposed global-load latency (a quarter of the time of a K = 256 tile).
    //
    // XCD-aware tile order: blocks b, b+8, ... share an XCD (round-robin dispatch; gridDim.x is
    // a multiple of 8 whenever a workgroup has more than one tile); give each XCD a contiguous
    // run of tiles so that the W panel / A panel re-reads of neighbouring tiles hit that XCD's L2.
    RowInfo ai[A_ITEMS];
    int64_t bofs[B_ITEMS];
    const char* dma_a[DMA ? BM / 32 : 1];
    const char* dma_b[DMA ? BN / 32 : 1];
    int dma_iy0[DMA && CONV ? BM / 32 : 1], dma_ix0[DMA && CONV ? BM / 32 : 1];      // conv: window origin of the row
    // PIPE conv: per row, the address of its window origin (tap (0,0), channel 0; may lie outside the image -- only
    // dereferenced where the mask allows) and a bit per tap that falls inside the image.  A stage then costs a row one
    // AND, one compare, one 64-bit add of the wave-uniform tap offset and the select against the zero page, instead of two
    // adds, two range tests and a 64-bit multiply-add chain (a plain VALU instruction takes ~4 cycles from the matrix pipe
    // whatever the occupancy: tools/hw_probe/mfma_valu_overlap.hip)
    const char* cbase[PIPE && CONV ? BM / 32 : 1];
    uint32_t cmask[PIPE && CONV ? BM / 32 : 1];
    uint32_t pv_a[PIPE ? BM / 32 : 1], pv_b[PIPE ? BN / 32 : 1];                      // PIPE: per-lane byte offsets from the tile's row 0
    const char *pbase_a = nullptr, *pbase_b = nullptr;                                // PIPE: the tile's A / W row 0 (wave-uniform)
    int m0, n0, tile_m;
    const int tiles_m = num_tiles / tiles_n;
    const int panel_w = (vec_epi & 8) && tiles_n > 8 && (tiles_n & 7) == 0 ? 8 : 0;         // (host: bit 3 of vec_epi)
    auto setup_tile = [&](int t) {
        int bid = t;
        {
            const int q = num_tiles >> 3, r = num_tiles & 7, xcd = bid & 7;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        }
        // Column panels of 8 tiles (GRL_GEMM_PANEL): with more than 8 column tiles the plain row-major walk hands the 64
        // tiles an XCD runs at a time as 4 rows x 16 columns (N = 2048) -- 20 operand panels per k-slice through its L2 --
        // where 8 x 8 needs 16.  Each panel is walked over ALL row tiles before the next panel starts.
        int tile_n;
        if (panel_w > 0) {
            const int per_panel = tiles_m * panel_w;
            const int pn = bid / per_panel, rem = bid - pn * per_panel;
            tile_m = rem / panel_w;
            tile_n = pn * panel_w + (rem - tile_m * panel_w);
        } else {
            tile_m = bid / tiles_n;
            tile_n = bid - tile_m * tiles_n;
        }
        m0 = tile_m * BM;
        n0 = tile_n * BN;
        // ---- per-thread staging rows ----------------------------------------
#pragma unroll
        for (int i = 0; i < A_ITEMS; ++i) {
            int m = m0 + ld_row + 32 * i;
            m = m < p.M ? m : p.M - 1;                    // clamp: edge rows are never stored
            if (CONV) {
                const int hw = p.Ho * p.Wo;
                const int img = m / hw, rem = m - img * hw;
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                ai[i].base = (int64_t)img * p.H * p.W * p.C;
                ai[i].iy0 = oy * p.stride - p.pad;
                ai[i].ix0 = ox * p.stride - p.pad;
            } else {
                ai[i].base = (int64_t)m * p.lda;
                ai[i].iy0 = ai[i].ix0 = 0;
            }
        }
#pragma unroll
        for (int i = 0; i < B_ITEMS; ++i) {
            int n = n0 + ld_row + 32 * i;
            n = n < p.N ? n : p.N - 1;
            bofs[i] = (int64_t)n * p.ldw;
        }
        if constexpr (PIPE) {
            pbase_b = reinterpret_cast<const char*>(p.w) + ((int64_t)n0 * p.ldw) * ESZ;
#pragma unroll
            for (int i = 0; i < BN / 32; ++i) {
                const int r = (wave + 4 * i) * 8 + d_row;
                const int n = n0 + r < p.N ? r : p.N - 1 - n0;             // clamp: edge rows are never stored
                pv_b[i] = (uint32_t)(n * p.ldw * ESZ + ((d_chunk ^ ((r >> 1) & 7)) << 4));
            }
        }
        if constexpr (PIPE && !CONV) {
            pbase_a = reinterpret_cast<const char*>(p.a) + ((int64_t)m0 * p.lda) * ESZ;
#pragma unroll
            for (int i = 0; i < BM / 32; ++i) {
                const int r = (wave + 4 * i) * 8 + d_row;
                const int m = m0 + r < p.M ? r : p.M - 1 - m0;
                pv_a[i] = (uint32_t)(m * p.lda * ESZ + ((d_chunk ^ ((r >> 1) & 7)) << 4));
            }
        } else if constexpr (DMA) {
#pragma unroll
            for (int i = 0; i < BM / 32; ++i) {
                const int r = (wave + 4 * i) * 8 + d_row;
                int m = m0 + r;
                m = m < p.M ? m : p.M - 1;
                if constexpr (CONV) {                          // image base; the stage adds the tap's pixel and channel
                    const int hw = p.Ho * p.Wo;
                    const int img = m / hw, rem = m - img * hw;
                    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                    dma_a[i] = reinterpret_cast<const char*>(p.a) + ((int64_t)img * p.H * p.W * p.C) * ESZ + ((d_chunk ^ ((r >> 1) & 7)) << 4);
                    dma_iy0[i] = oy * p.stride - p.pad;
                    dma_ix0[i] = ox * p.stride - p.pad;
                    if constexpr (PIPE) {
                        cbase[i] = dma_a[i] + ((int64_t)(dma_iy0[i] * p.W + dma_ix0[i]) * p.C) * ESZ;
                        uint32_t mk = 0;
                        for (int ky = 0, t = 0; ky < p.kh; ++ky)
                            for (int kx = 0; kx < p.kw; ++kx, ++t)
                                if ((unsigned)(dma_iy0[i] + ky) < (unsigned)p.H && (unsigned)(dma_ix0[i] + kx) < (unsigned)p.W) mk |= 1u << t;
                        cmask[i] = mk;
                    }
                } else {
                    dma_a[i] = reinterpret_cast<const char*>(p.a) + ((int64_t)m * p.lda) * ESZ + ((d_chunk ^ ((r >> 1) & 7)) << 4);
                }
            }
#pragma unroll
            for (int i = 0; i < BN / 32; ++i) {
                const int r = (wave + 4 * i) * 8 + d_row;
                int n = n0 + r;
                n = n < p.N ? n : p.N - 1;
                dma_b[i] = reinterpret_cast<const char*>(p.w) + ((int64_t)n * p.ldw) * ESZ + ((d_chunk ^ ((r >> 1) & 7)) << 4);
            }
        }
    };

    // DMA (dense fp32 / bf16-storage operands): the stage goes global -> LDS by LDS-DMA
    // (`global_load_lds_dwordx4`): no staging VGPRs, no ds_write, no vmcnt wait in front of them.  A
    // wave-instruction fills 8 rows x 128 B linearly; the chunk swizzle moves to the per-lane SOURCE.
    // Wave w owns the 8-row blocks w, w+4, ... of both tiles; lane l is (row l>>3, chunk l&7).
    // PIPE: piece i of a stage (0..3: A row blocks wave, wave+4, ..; 4..7: W row blocks)
    const uint32_t lds_base = (uint32_t)(size_t)((lds_cptr_t) reinterpret_cast<char*>(smem));
    // conv: the tap (ky, kx) and channel offset of the NEXT stage to request, advanced by one stage per call instead of
    // two integer divisions per stage; the per-lane gather addresses of a stage are computed one chunk ahead of their
    // pieces (pn_src), so the address arithmetic interleaves with MFMAs instead of sitting in front of the DMA issue
    int pky = 0, pkx = 0, pc0 = 0;
    const char* pn_src[PIPE && CONV ? BM / 32 : 1];
    const char* czero = reinterpret_cast<const char*>(g_zero_chunks) + (d_chunk << 4);
    if constexpr (PIPE && CONV) asm volatile("" : "+v"(czero));       // (kept in registers: rematerialised it is a GOT load per use)
    auto conv_tap_reset = [&]() { pky = pkx = pc0 = 0; };
    auto conv_tap_advance = [&]() {
        pc0 += KST;
        if (pc0 >= p.C) {
            pc0 = 0;
            if (++pkx == p.kw) { pkx = 0; ++pky; }
        }
    };
    auto conv_addrs = [&]() {
        if constexpr (PIPE && CONV) {
            const int64_t toff = ((int64_t)(pky * p.W + pkx) * p.C + pc0) * ESZ;       // wave-uniform
            const uint32_t bit = 1u << (pky * p.kw + pkx);
#pragma unroll
            for (int i = 0; i < BM / 32; ++i)
                pn_src[i] = (cmask[i] & bit) ? cbase[i] + toff : czero;
        }
    };
    auto dma_piece = [&](int buf, int ks, int i) {
        if constexpr (PIPE) {
            const int64_t kb = (int64_t)ks * KST * ESZ;
            if (i < BM / 32) {
                const uint32_t dst = lds_base + (uint32_t)((buf * BM * BK) * 4 + (wave + 4 * i) * 1024);
                if constexpr (CONV) dma16_hidden_v(pn_src[i], dst);
                else dma16_hidden(pbase_a + kb, pv_a[i], dst);
            } else
                dma16_hidden(pbase_b + kb, pv_b[i - BM / 32],
                             lds_base + (uint32_t)((NST * BM * BK + buf * BN * BK) * 4 + (wave + 4 * (i - BM / 32)) * 1024));
        }
    };
    auto dma_stage = [&](int buf, int ks) {
        if constexpr (PIPE) {          // (a tile's first stage: ks == 0)
            if constexpr (CONV) { conv_tap_reset(); conv_addrs(); }
#pragma unroll
            for (int i = 0; i < BM / 32 + BN / 32; ++i) dma_piece(buf, ks, i);
        } else if constexpr (DMA) {
            if (GRL_GEMM_KO & 8) ks = 0;            // every stage re-reads the tile's first (cache-resident) k block
            char* const Asb = reinterpret_cast<char*>(As + buf * BM * BK);
            char* const Bsb = reinterpret_cast<char*>(Bs + buf * BN * BK);
            const int64_t kb = (int64_t)ks * KST * ESZ;
            if constexpr (CONV) {
                // implicit GEMM: the per-lane DMA source IS the gather -- the tap (ky, kx) and channel offset of the
                // stage are wave-uniform, out-of-image taps read a zero page
                const int k0 = ks * KST;
                const int tap = k0 / p.C, c0 = k0 - tap * p.C;
                const int ky = tap / p.kw, kx = tap - ky * p.kw;
                const char* const zsrc = reinterpret_cast<const char*>(g_zero_chunks) + (d_chunk << 4);
#pragma unroll
                for (int i = 0; i < BM / 32; ++i) {
                    const int rb = wave + 4 * i;
                    const int iy = dma_iy0[i] + ky, ix = dma_ix0[i] + kx;
                    const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                    const char* src = ok ? dma_a[i] + ((int64_t)(iy * p.W + ix) * p.C + c0) * ESZ : zsrc;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(Asb + rb * 1024), 16, 0, 0);
                }
            } else {
#pragma unroll
            for (int i = 0; i < BM / 32; ++i) {
                const int rb = wave + 4 * i;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dma_a[i] + kb),
                                                 (__attribute__((address_space(3))) void*)(Asb + rb * 1024), 16, 0, 0);
            }
            }
#pragma unroll
            for (int i = 0; i < BN / 32; ++i) {
                const int rb = wave + 4 * i;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dma_b[i] + kb),
                                                 (__attribute__((address_space(3))) void*)(Bsb + rb * 1024), 16, 0, 0);
            }
        }
    };

    f32x4 areg[A_ITEMS], breg[B_ITEMS];
    auto load_stage = [&](int ks) {
        if constexpr (DMA) return;
        const int k0 = ks * KST;
        if (CONV) {
            const int tap = k0 / p.C, c0 = k0 - tap * p.C;          // wave-uniform
            const int ky = tap / p.kw, kx = tap - ky * p.kw;
#pragma unroll
            for (int i = 0; i < A_ITEMS; ++i) {
                const int iy = ai[i].iy0 + ky, ix = ai[i].ix0 + kx;
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                // branch-free: out-of-image taps load pixel (0,0) of the same image and are
                // zeroed by a select, so the stage's loads issue back to back
                const int64_t pix = ok ? (int64_t)iy * p.W + ix : 0;
                f32x4 v = *gp(p.a, ai[i].base + pix * p.C + c0 + ld_chunk * EPC);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
                areg[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_ITEMS; ++i)
                areg[i] = *gp(p.a, ai[i].base + k0 + ld_chunk * EPC);
        }
#pragma unroll
        for (int i = 0; i < B_ITEMS; ++i)
            breg[i] = *gp(p.w, bofs[i] + k0 + ld_chunk * EPC);
    };
    auto split_store = [&](char* plane0, int plane_bytes, int row, const f32x4 v) {
        const int off = row * 64 + (((ld_chunk >> 1) ^ ((row >> 2) & 3)) << 4) + (ld_chunk & 1) * 8;
        bf16x4 hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) hi[e] = (__bf16)v[e];
        *reinterpret_cast<bf16x4*>(plane0 + off) = hi;
        if (MATH == 3) {
            bf16x4 lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) lo[e] = (__bf16)(v[e] - (float)hi[e]);
            *reinterpret_cast<bf16x4*>(plane0 + plane_bytes + off) = lo;
        }
    };
    auto store_stage = [&](int buf) {
        if constexpr (DMA) return;
        if constexpr (MATH == 0 || MATH == 2) {
#pragma unroll
            for (int i = 0; i < A_ITEMS; ++i) {
                const int row = ld_row + 32 * i;
                *reinterpret_cast<f32x4*>(As + buf * BM * BK + row * BK +
                                          ((ld_chunk ^ ((row >> 1) & 7)) << 2)) = areg[i];
            }
#pragma unroll
            for (int i = 0; i < B_ITEMS; ++i) {
                const int row = ld_row + 32 * i;
                *reinterpret_cast<f32x4*>(Bs + buf * BN * BK + row * BK +
                                          ((ld_chunk ^ ((row >> 1) & 7)) << 2)) = breg[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_ITEMS; ++i)
                split_store(Ah + buf * PL * BM * 64, BM * 64, ld_row + 32 * i, areg[i]);
#pragma unroll
            for (int i = 0; i < B_ITEMS; ++i)
                split_store(Bh + buf * PL * BN * 64, BN * 64, ld_row + 32 * i, breg[i]);
        }
    };

    const int nk = p.K / KST;
    const int frow = lane & 31, fhalf = lane >> 5;
    const int col_l = lane & 31;

    int t = blockIdx.x;
    setup_tile(t);
    load_stage(0);
    dma_stage(0, 0);
    for (;;) {
    f32x16 acc[MT][NT], tot[SEG ? MT : 1][SEG ? NT : 1];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if constexpr (SEG) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
    }

    store_stage(0);
    constexpr int NP_ALL = BM / 32 + BN / 32;              // DMA pieces of one stage per wave
    if constexpr (RING) {
        // the tile's SECOND stage follows its first at once (buffer 1 is free: the previous tile ended on a barrier)
        if (nk > 1) {
            if constexpr (CONV) { conv_tap_advance(); conv_addrs(); }
#pragma unroll
            for (int i = 0; i < NP_ALL; ++i) dma_piece(1, 1, i);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP_ALL) : "memory");      // stage 0 has landed, stage 1 may be in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else if constexpr (PIPE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    // PIPE: fragments double-buffered in registers ACROSS the stage boundary.  Stage ks (buffer b): chunk q's 16 MFMAs run
    // while chunk q+1's four ds_read_b128 are in flight; the eight LDS-DMA pieces of stage ks+1 go out two at a time
    // between the MFMA groups of chunk 0 (buffer b^1 was released by the previous stage's barrier); the stage barrier
    // sits in front of the LAST chunk's MFMAs -- its fragments have landed, so nobody reads b again -- and the first
    // fragments of stage ks+1 are requested right behind it, under those 16 MFMAs.  Same MFMA order as the plain loop:
    // bit-identical results.
    f32x4 paf[PIPE ? 2 : 1][MT], pbf[PIPE ? 2 : 1][NT];
    auto prd = [&](int set, int buf, int q) {
        if constexpr (PIPE) {
            const float* Ab = As + buf * BM * BK + (wm * WTM) * BK;
            const float* Bb = Bs + buf * BN * BK + (wn * WTN) * BK;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = i * 32 + frow;
                paf[set][i] = *reinterpret_cast<const f32x4*>(Ab + row * BK + (((2 * q + fhalf) ^ ((row >> 1) & 7)) << 2));
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int row = j * 32 + frow;
                pbf[set][j] = *reinterpret_cast<const f32x4*>(Bb + row * BK + (((2 * q + fhalf) ^ ((row >> 1) & 7)) << 2));
            }
        }
    };
    if constexpr (PIPE) prd(0, 0, 0);

    int rbuf = 0;                                          // RING: the stage buffer of stage ks (ks % 3)
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = RING ? rbuf : (ks & 1);
        if constexpr (!PIPE)
            if (!(GRL_GEMM_KO & 1) && ks + 1 < nk) { load_stage(ks + 1); dma_stage(buf ^ 1, ks + 1); }
        if constexpr (RING) {
            const int b1 = buf == 2 ? 0 : buf + 1;            // stage ks+1's buffer
            const int b2 = b1 == 2 ? 0 : b1 + 1;              // stage ks+2's (= stage ks-1's: released by the previous barrier)
            const bool more = ks + 1 < nk, more2 = ks + 2 < nk;
            constexpr int NA = BM / 32, NB = BN / 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < 3) {
                    prd((q + 1) & 1, buf, q + 1);
                } else {
                    if (more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");     // stage ks+1 landed; ks+2 in flight
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (more) prd(0, b1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // a chunk is ONE v_mfma_f32_32x32x16_bf16 per tile pair (the 16-byte fragment holds 8 k)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, paf[q & 1][i]),
                                                                            __builtin_bit_cast(bf16x8, pbf[q & 1][j]), acc[i][j], 0, 0, 0);
                // stage ks+2's pieces behind chunks 0 and 1 (dense: half and half; conv: the W pieces, then -- after the
                // gather addresses -- the A pieces)
                if (more2 && q < 2) {
                    const int first = CONV ? (q == 0 ? NA : 0) : (q == 0 ? 0 : (NA + NB) / 2);
                    const int cnt = CONV ? (q == 0 ? NB : NA) : (NA + NB) / 2;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < (CONV ? (NA > NB ? NA : NB) : (NA + NB) / 2); ++k)
                        if (k < cnt) dma_piece(b2, ks + 2, first + k);
                    __builtin_amdgcn_sched_barrier(0);
                    if (CONV && q == 0) { conv_tap_advance(); conv_addrs(); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            rbuf = b1;
        } else if constexpr (PIPE) {
            const bool more = ks + 1 < nk;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < 3) {
                    prd((q + 1) & 1, buf, q + 1);
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (more) prd(0, buf ^ 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(paf[q & 1][i][s], pbf[q & 1][j][s], acc[i][j], 0, 0, 0);
                    // stage ks+1's pieces, two per MFMA group.  Dense: A then W, all inside chunk 0.  Conv: the W pieces in
                    // chunk 0 while the gather addresses are computed (plain VALU work the compiler interleaves with the
                    // rest of the chunk's MFMAs), the A pieces in chunk 1.
                    constexpr int NA = BM / 32, NB = BN / 32;
                    const int first = CONV ? (q == 0 ? NA + 2 * s : q == 1 ? 2 * s : -1) : (q == 0 ? 2 * s : -1);
                    const int last = CONV ? (q == 0 ? NA + NB : NA) : NA + NB;
                    if (more && first >= 0 && first < last) {
                        __builtin_amdgcn_sched_barrier(0);
                        dma_piece(buf ^ 1, ks + 1, first);
                        dma_piece(buf ^ 1, ks + 1, first + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (CONV && q == 0 && s == 1 && more) { conv_tap_advance(); conv_addrs(); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (MATH == 2) {
            const float* Ab = As + buf * BM * BK + (wm * WTM) * BK;
            const float* Bb = Bs + buf * BN * BK + (wn * WTN) * BK;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                  // four K = 16 steps per 64-wide stage
                bf16x8 af[MT], bf[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int row = i * 32 + frow;
                    af[i] = *reinterpret_cast<const bf16x8*>(
                        Ab + row * BK + (((2 * q + fhalf) ^ ((row >> 1) & 7)) << 2));
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int row = j * 32 + frow;
                    bf[j] = *reinterpret_cast<const bf16x8*>(
                        Bb + row * BK + (((2 * q + fhalf) ^ ((row >> 1) & 7)) << 2));
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        } else if constexpr (MATH == 0) {
            const float* Ab = As + buf * BM * BK + (wm * WTM) * BK;
            const float* Bb = Bs + buf * BN * BK + (wn * WTN) * BK;
            if constexpr (SEG && DMA) {
            // (K-blocked train kernels only -- the LDS-DMA forms, which have the registers for it: +1.5 % on the train step; the eval kernels LOSE 3 % to it -- their compiler
            // schedule already overlaps the next chunk's reads with the last two MFMAs and the extra registers cost more)
            // fragments double-buffered in registers: the reads of chunk q+1 are issued, and fenced, in front of
            // the 4 * MT * NT MFMAs of chunk q (see wgrad_kernel in train.hip)
            f32x4 af[2][MT], bf[2][NT];
            auto rd = [&](int set, int q) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int row = i * 32 + frow;
                    af[set][i] = *reinterpret_cast<const f32x4*>(Ab + row * BK + (((2 * q + fhalf) ^ ((row >> 1) & 7)) << 2));
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int row = j * 32 + frow;
                    bf[set][j] = *reinterpret_cast<const f32x4*>(Bb + row * BK + (((2 * q + fhalf) ^ ((row >> 1) & 7)) << 2));
                }
            };
            rd(0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q + 1 < 4) rd((q + 1) & 1, q + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 1][i][s], bf[q & 1][j][s],
                                                                             acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 af[MT], bf[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int row = i * 32 + frow;
                    // (wm*WTM) is a multiple of 32, so (row>>1)&7 equals the tile-local swizzle
                    af[i] = *reinterpret_cast<const f32x4*>(
                        Ab + row * BK + (((2 * q + fhalf) ^ ((row >> 1) & 7)) << 2));
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int row = j * 32 + frow;
                    bf[j] = *reinterpret_cast<const f32x4*>(
                        Bb + row * BK + (((2 * q + fhalf) ^ ((row >> 1) & 7)) << 2));
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s],
                                                                             acc[i][j], 0, 0, 0);
            }
            }
        } else {
            const char* Ab = Ah + buf * PL * BM * 64 + (wm * WTM) * 64;
            const char* Bb = Bh + buf * PL * BN * 64 + (wn * WTN) * 64;
#pragma unroll
            for (int s = 0; s < 2; ++s) {                 // two K = 16 steps per 32-wide stage
                bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int row = i * 32 + frow;
                    const int off = row * 64 + (((2 * s + fhalf) ^ ((row >> 2) & 3)) << 4);
                    ah[i] = *reinterpret_cast<const bf16x8*>(Ab + off);
                    if (MATH == 3) al[i] = *reinterpret_cast<const bf16x8*>(Ab + BM * 64 + off);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int row = j * 32 + frow;
                    const int off = row * 64 + (((2 * s + fhalf) ^ ((row >> 2) & 3)) << 4);
                    bh[j] = *reinterpret_cast<const bf16x8*>(Bb + off);
                    if (MATH == 3) bl[j] = *reinterpret_cast<const bf16x8*>(Bb + BN * 64 + off);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if (MATH == 3) {                  // small terms first
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
        if constexpr (SEG) {
            if (((ks + 1) & (SEG_STAGES - 1)) == 0 && ks + 1 < nk) {      // segment boundary inside K
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        tot[i][j] += acc[i][j];
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                    }
            }
        }
        if constexpr (!PIPE) {
            if (!(GRL_GEMM_KO & 1) && ks + 1 < nk) store_stage(buf ^ 1);
            if (!(GRL_GEMM_KO & 2)) __syncthreads();
        }
    }
    if constexpr (SEG) {
        if (nk > SEG_STAGES) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] += tot[i][j];
        }
    }

    // the K loop ended on a barrier: stage registers and LDS are free.  Request the NEXT tile's
    // first stage now (its row tables replace this tile's, which the epilogue does not use).
    const int cur_m0 = m0, cur_n0 = n0, cur_tile_m = tile_m;
    const int next_t = t + (int)gridDim.x;
    if (next_t < num_tiles) {
        setup_tile(next_t);
        load_stage(0);
    }
    // (DMA: the next tile's first stage is issued AFTER the epilogue has left the LDS it parks C in)

    // ---- epilogue --------------------------------------------------------------
    // acc[i][j][r] is Y[row][col] with row = (r&3) + 8*(r>>2) + 4*fhalf, col = lane&31
    auto epilogue = [&](const int m0, const int n0, const int tile_m) {
    if (vec_epi & 1) {
        // Wide path (N, ldy, ldres multiples of 4, 16-B aligned bases): each wave parks its
        // sub-tile in LDS (the A/B stages are dead: the K loop ended on a barrier) and
        // reads it back row-major, so every lane owns 4 consecutive channels of one
        // row: float4 scale/shift/residual loads and 256-B contiguous row segments per
        // 16 lanes on the store, instead of 64 dword-wide store instructions.
        float* Cs = smem;                                   // [BM][BN]
        // ---- fast path (round 5): interior tile, plain affine epilogue (+ residual, ReLU) ----------------------------
        // The general code below tests `m < M`, p.rowscale, p.gbias, p.res, p.stats, p.bn_z ... inside its row loop:
        // dozens of basic blocks, and hipcc -- which tracks vmcnt per block -- put `s_waitcnt vmcnt(0)` in front of nearly
        // every residual load and output store (458 of the 463 vmcnt waits of the 128 x 128 kernel were (0)): a store
        // waited for the previous store's acknowledge.  Here the tile lies inside the matrix and nothing optional is set,
        // so the row loop is ONE basic block: residual rows are requested a chunk ahead, stores never wait for stores.
        // Same operations in the same order as below: bit-identical.  (Not in the K-blocked SEG instantiations: the
        // train-mode forward they serve always carries statistics, and their second accumulator set leaves no registers.)
        if constexpr (!SEG) {
            // (... or carrying a BatchNorm-backward reduce, GrlGemm.bn_z: the data-gradient GEMMs of a training step --
            //  a row of z and a mask word per output row on top of the residual, the launches the general loop hurt most)
            const bool bnz = p.bn_z != nullptr && p.stats != nullptr;
            const bool fast = p.epilogue == GRL_EPI_AFFINE && !p.rowscale && !p.gbias && (bnz || (!p.stats && !p.bn_z)) &&
                              m0 + BM <= p.M && n0 + BN <= p.N && (MATH != 2 || (!p.out_f32 && p.N % 8 == 0));
            if (fast) {
                auto park = [&]() {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                Cs[(wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf) * BN + wn * WTN + j * 32 + col_l] = acc[i][j][r];
                };
                float relu_floor;      // 0 (ReLU) or a quiet NaN (no ReLU: v_max returns the other operand, a NaN accumulator stays NaN);
            {                       // through an asm move: told the constant, hipcc folds max(t, NaN) into a select per element
                const uint32_t floor_bits = p.relu ? 0u : 0x7fc00000u;
                asm("v_mov_b32 %0, %1" : "=v"(relu_floor) : "s"(floor_bits));
            }      // bf16 storage: ReLU as one v_max
                const bool relu = p.relu != 0;
                auto run = [&](auto has_res_, auto bnz_) {
                    constexpr bool HAS_RES = decltype(has_res_)::value;
                    constexpr bool BNZ = decltype(bnz_)::value;
                    if constexpr (MATH == 2) {
                        constexpr int LPR8 = WTN / 8, RPI8 = 64 / LPR8, NIT = WTM / RPI8, CH = (BM == 128 && BN == 128) ? 2 : (NIT < 4 ? NIT : 4);
                        const int lrow = lane / LPR8, lcol = (lane % LPR8) * 8;
                        const int n = n0 + wn * WTN + lcol;
                        f32x4 sc[2], sh[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            sc[u] = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + n + 4 * u) : f32x4{1.f, 1.f, 1.f, 1.f};
                            sh[u] = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + n + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                        const __bf16* const r16 = reinterpret_cast<const __bf16*>(p.res);
                        __bf16* const y16 = reinterpret_cast<__bf16*>(p.y);
                        // BatchNorm-backward reduce on bf16 storage (round 5): z is a bf16 [M][N] tensor, the recorded
                        // ReLU bits one byte per eight outputs; the sums are taken from the fp32 value before it is
                        // rounded to bf16
                        const __bf16* const z16 = reinterpret_cast<const __bf16*>(p.bn_z);
                        const bool use_bits = BNZ && p.bn_bits != nullptr;
                        const bool use_ms = BNZ && !use_bits && p.bn_mscale != nullptr;
                        const uint8_t* const bbyte = use_bits ? p.bn_bits : reinterpret_cast<const uint8_t*>(p.bn_z);
                        f32x4 bmu[2], bis[2], bms[2], bmb[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            bmu[u] = bis[u] = bms[u] = bmb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                            if constexpr (BNZ) {
                                bmu[u] = *reinterpret_cast<const f32x4*>(p.bn_mean + n + 4 * u);
                                bis[u] = *reinterpret_cast<const f32x4*>(p.bn_invstd + n + 4 * u);
                                if (use_ms) {
                                    bms[u] = *reinterpret_cast<const f32x4*>(p.bn_mscale + n + 4 * u);
                                    if (p.bn_mbeta) bmb[u] = *reinterpret_cast<const f32x4*>(p.bn_mbeta + n + 4 * u);
                                }
                            }
                        }
                        f32x4 ssum8[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, ssq8[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                        bf16x8 r8[2][CH], z8[2][BNZ ? CH : 1];
                        uint32_t bb[2][BNZ ? CH : 1];
                        auto request = [&](int c) {
#pragma unroll
                            for (int k = 0; k < CH; ++k) {
                                const int m = m0 + wm * WTM + (c * CH + k) * RPI8 + lrow;
                                if constexpr (HAS_RES) r8[c & 1][k] = *reinterpret_cast<const bf16x8*>(r16 + (int64_t)m * p.ldres + n);
                                if constexpr (BNZ) {
                                    z8[c & 1][k] = *reinterpret_cast<const bf16x8*>(z16 + (int64_t)m * p.N + n);
                                    bb[c & 1][k] = bbyte[((int64_t)m * p.N + n) >> 3];
                                }
                            }
                        };
                        if constexpr (HAS_RES || BNZ) request(0);
                        park();
#pragma unroll
                        for (int c = 0; c < NIT / CH; ++c) {
                            if constexpr (HAS_RES || BNZ) { if (c + 1 < NIT / CH) request(c + 1); }
#pragma unroll
                            for (int k = 0; k < CH; ++k) {
                                const int row = wm * WTM + (c * CH + k) * RPI8 + lrow;
                                const int m = m0 + row;
                                f32x8 o32;
#pragma unroll
                                for (int u = 0; u < 2; ++u) {
                                    f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * BN + wn * WTN + lcol + 4 * u);
                                    v = v * sc[u] + sh[u];
                                    f32x4 tv;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        float t = v[e];
                                        if constexpr (HAS_RES) t = t + (float)r8[c & 1][k][4 * u + e];
                                        else t = t + 0.f;
                                        // ReLU: one v_max against 0 / NaN (no ReLU: v_max(t, NaN) == t, NaN stays NaN; v_max_f32 orders -0 < +0: == (t > 0 ? t : 0) bit for bit)
                                        tv[e] = __builtin_fmaxf(t, relu_floor);
                                    }
                                    if constexpr (BNZ) {
                                        f32x4 zc;
#pragma unroll
                                        for (int e = 0; e < 4; ++e) zc[e] = (float)z8[c & 1][k][4 * u + e] - bmu[u][e];
                                        const uint32_t mk = use_bits ? (bb[c & 1][k] >> (4 * u)) : 0xfu;
                                        const f32x4 tm = zc * bms[u] + bmb[u];
#pragma unroll
                                        for (int e = 0; e < 4; ++e) tv[e] = ((((mk >> e) & 1u) != 0) & (!use_ms | (tm[e] > 0.f))) ? tv[e] : 0.f;
                                        ssum8[u] += tv; ssq8[u] += tv * (zc * bis[u]);
                                    }
#pragma unroll
                                    for (int e = 0; e < 4; ++e) o32[4 * u + e] = tv[e];
                                }
                                *reinterpret_cast<bf16x8*>(y16 + (int64_t)m * p.ldy + n) = __builtin_convertvector(o32, bf16x8);
                            }
                        }
                        if constexpr (BNZ) {
                            // (the statistics epilogue's reduction: the lanes of a column group, the two wave rows through LDS)
#pragma unroll
                            for (int o2 = LPR8; o2 < 64; o2 <<= 1) {
#pragma unroll
                                for (int u = 0; u < 2; ++u)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        ssum8[u][e] += __shfl_xor(ssum8[u][e], o2);
                                        ssq8[u][e] += __shfl_xor(ssq8[u][e], o2);
                                    }
                            }
                            __syncthreads();                               // every wave has read its C slab: LDS is free
                            float* red = smem;                             // [2 wm][2][BN]
                            if (lrow == 0) {
#pragma unroll
                                for (int u = 0; u < 2; ++u) {
                                    *reinterpret_cast<f32x4*>(red + (wm * 2 + 0) * BN + wn * WTN + lcol + 4 * u) = ssum8[u];
                                    *reinterpret_cast<f32x4*>(red + (wm * 2 + 1) * BN + wn * WTN + lcol + 4 * u) = ssq8[u];
                                }
                            }
                            __syncthreads();
                            for (int c = tid; c < BN; c += 256) {
                                const int nn = n0 + c;
                                p.stats[((int64_t)tile_m * 2 + 0) * p.N + nn] = red[0 * BN + c] + red[2 * BN + c];
                                p.stats[((int64_t)tile_m * 2 + 1) * p.N + nn] = red[1 * BN + c] + red[3 * BN + c];
                            }
                        }
                    } else {
                        constexpr int LPR = WTN / 4, RPI = 64 / LPR, NIT = WTM / RPI, CH = (BM == 128 && BN == 128) ? 2 : (NIT < 4 ? NIT : 4);
                        const int lrow = lane / LPR, lcol = (lane % LPR) * 4;
                        const int n = n0 + wn * WTN + lcol;
                        const f32x4 sc = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
                        const f32x4 sh = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                        // BatchNorm-backward reduce: the three mask forms of the general loop without a branch -- recorded
                        // bits (a mask byte is loaded either way; all ones where there are none), and / or the forward's
                        // (z - mean) * mscale + mbeta > 0 (`+ 0` where mbeta is absent leaves the comparison as it is)
                        f32x4 bmu = sh, bis = sh, bms = f32x4{0.f, 0.f, 0.f, 0.f}, bmb = bms;
                        const bool use_bits = BNZ && p.bn_bits != nullptr;
                        if constexpr (BNZ) {
                            bmu = *reinterpret_cast<const f32x4*>(p.bn_mean + n);
                            bis = *reinterpret_cast<const f32x4*>(p.bn_invstd + n);
                            if (!use_bits && p.bn_mscale) {
                                bms = *reinterpret_cast<const f32x4*>(p.bn_mscale + n);
                                bmb = p.bn_mbeta ? *reinterpret_cast<const f32x4*>(p.bn_mbeta + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                            }
                        }
                        // (a valid word to load where there are no recorded bits: the z row itself)
                        const uint8_t* const bbyte = use_bits ? p.bn_bits : reinterpret_cast<const uint8_t*>(p.bn_z);
                        const bool use_ms = BNZ && !use_bits && p.bn_mscale != nullptr;
                        f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
                        f32x4 rr[2][CH], zz[2][BNZ ? CH : 1];
                        uint32_t bb[2][BNZ ? CH : 1];
                        auto request = [&](int c) {
#pragma unroll
                            for (int k = 0; k < CH; ++k) {
                                const int m = m0 + wm * WTM + (c * CH + k) * RPI + lrow;
                                if constexpr (HAS_RES) rr[c & 1][k] = *reinterpret_cast<const f32x4*>(p.res + (int64_t)m * p.ldres + n);
                                if constexpr (BNZ) {
                                    zz[c & 1][k] = *reinterpret_cast<const f32x4*>(p.bn_z + (int64_t)m * p.N + n);
                                    bb[c & 1][k] = bbyte[((int64_t)m * p.N + n) >> 2];
                                }
                            }
                        };
                        if constexpr (HAS_RES || BNZ) request(0);
                        park();
#pragma unroll
                        for (int c = 0; c < NIT / CH; ++c) {
                            if constexpr (HAS_RES || BNZ) { if (c + 1 < NIT / CH) request(c + 1); }
#pragma unroll
                            for (int k = 0; k < CH; ++k) {
                                const int row = wm * WTM + (c * CH + k) * RPI + lrow;
                                const int m = m0 + row;
                                f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * BN + wn * WTN + lcol);
                                v = v * sc + sh;
                                if constexpr (HAS_RES) v += rr[c & 1][k];
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = relu ? (v[e] > 0.f ? v[e] : 0.f) : v[e];    // (v_max here measured SLOWER on the fp32 headline: 14.54 -> 14.68 ms)
                                if constexpr (BNZ) {
                                    const f32x4 zc = zz[c & 1][k] - bmu;
                                    const uint32_t mk = use_bits ? bb[c & 1][k] : 0xfu;
                                    const f32x4 t = zc * bms + bmb;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = ((((mk >> e) & 1u) != 0) & (!use_ms | (t[e] > 0.f))) ? v[e] : 0.f;
                                    ssum += v; ssq += v * (zc * bis);
                                }
                                *reinterpret_cast<f32x4*>(p.y + (int64_t)m * p.ldy + n) = v;
                            }
                        }
                        if constexpr (BNZ) {
                            // the general path's reduction, step for step (a lane's rows above; then the lanes of a column
                            // group, the two wave rows through LDS): stats[tile_m][0|1][n]
#pragma unroll
                            for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) { ssum[e] += __shfl_xor(ssum[e], o); ssq[e] += __shfl_xor(ssq[e], o); }
                            }
                            __syncthreads();                               // every wave has read its C slab: LDS is free
                            float* red = smem;                             // [2 wm][2][BN]
                            if (lrow == 0) {
                                *reinterpret_cast<f32x4*>(red + (wm * 2 + 0) * BN + wn * WTN + lcol) = ssum;
                                *reinterpret_cast<f32x4*>(red + (wm * 2 + 1) * BN + wn * WTN + lcol) = ssq;
                            }
                            __syncthreads();
                            for (int c = tid; c < BN; c += 256) {
                                const int nn = n0 + c;
                                p.stats[((int64_t)tile_m * 2 + 0) * p.N + nn] = red[0 * BN + c] + red[2 * BN + c];
                                p.stats[((int64_t)tile_m * 2 + 1) * p.N + nn] = red[1 * BN + c] + red[3 * BN + c];
                            }
                        }
                    }
                };
                if (bnz) {
                    if (p.res) run(std::true_type{}, std::true_type{});
                    else run(std::false_type{}, std::true_type{});
                } else if (p.res) run(std::true_type{}, std::false_type{});
                else run(std::false_type{}, std::false_type{});
                return;
            }
        }
        // fp32 residual rows of this lane, requested BEFORE the accumulators take their round trip through LDS:
        // the epilogue of a short-K layer is a read-modify-write of the output at HBM speed, and a residual load
        // issued per row inside the store loop serialises its latency with the stores
        constexpr int EPI_ROWS = WTM / (64 / (WTN / 4));
        constexpr bool RES_PREFETCH = MATH != 2 && !SEG && !(BM == 128 && BN == 128);     // (16 rows x 4 VGPRs would spill there)
        constexpr int EPI_UNROLL = RES_PREFETCH ? EPI_ROWS : 4;
        f32x4 rpre[RES_PREFETCH ? EPI_ROWS : 1];
        // ... and the rows of z (+ ReLU mask words) of a fused BatchNorm-backward reduce (GrlGemm.bn_z): the data-gradient
        // GEMMs into the 4P-wide tensors are short-K, output-heavy launches whose epilogue was a chain of dependent loads
        f32x4 zpre[RES_PREFETCH ? EPI_ROWS : 1];
        uint32_t bpre[RES_PREFETCH ? EPI_ROWS : 1];
        if constexpr (RES_PREFETCH) {
            constexpr int LPRp = WTN / 4, RPIp = 64 / LPRp;
            const int np = n0 + wn * WTN + (lane % LPRp) * 4;
            if (p.res && p.epilogue == GRL_EPI_AFFINE) {
#pragma unroll
                for (int it = 0; it < EPI_ROWS; ++it) {
                    const int m = m0 + wm * WTM + it * RPIp + lane / LPRp;
                    if (m < p.M && np < p.N) rpre[it] = *reinterpret_cast<const f32x4*>(p.res + (int64_t)m * p.ldres + np);
                }
            }
            if (p.bn_z && p.epilogue == GRL_EPI_AFFINE) {
#pragma unroll
                for (int it = 0; it < EPI_ROWS; ++it) {
                    const int m = m0 + wm * WTM + it * RPIp + lane / LPRp;
                    if (m < p.M && np < p.N) {
                        zpre[it] = *reinterpret_cast<const f32x4*>(p.bn_z + (int64_t)m * p.N + np);
                        if (p.bn_bits) bpre[it] = p.bn_bits[((int64_t)m * p.N + np) >> 2];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Cs[(wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf) * BN + wn * WTN +
                       j * 32 + col_l] = acc[i][j][r];
        if constexpr (MATH == 2) {
            // bf16 output: 8 channels (16 bytes) per lane; N % 8 == 0 is validated on the host
            if (p.N % 8 == 0 && !p.out_f32 && !p.rowscale) {
                constexpr int LPR8 = WTN / 8, RPI8 = 64 / LPR8;
                const int lrow = lane / LPR8, lcol = (lane % LPR8) * 8;
                const int n = n0 + wn * WTN + lcol;
                // train-mode BatchNorm (bf16-storage training): column sums of the RAW fp32 accumulator -- before it is
                // rounded to bf16 -- per lane over its rows, then lanes / wave rows below, as in the fp32 epilogue
                f32x4 ssum8[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, ssq8[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                if (n < p.N) {
                    f32x4 sc[2], sh[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        sc[u] = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + n + 4 * u) : f32x4{1.f, 1.f, 1.f, 1.f};
                        sh[u] = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + n + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll 4
                    for (int it = 0; it < WTM / RPI8; ++it) {
                        const int row = wm * WTM + it * RPI8 + lrow;
                        const int m = m0 + row;
                        if (m >= p.M) continue;
                        f32x4 v[2];
                        bf16x8 r8;
                        if (p.res) r8 = *reinterpret_cast<const bf16x8*>(
                                       reinterpret_cast<const __bf16*>(p.res) + (int64_t)m * p.ldres + n);
                        bf16x8 o;
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            v[u] = *reinterpret_cast<const f32x4*>(Cs + row * BN + wn * WTN + lcol + 4 * u);
                            if (p.gbias)
                                v[u] += *reinterpret_cast<const f32x4*>(
                                    p.gbias + (int64_t)(m / p.rows_per_group) * p.N + n + 4 * u);
                            if (p.stats) { ssum8[u] += v[u]; ssq8[u] += v[u] * v[u]; }
                            v[u] = v[u] * sc[u] + sh[u];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float t = v[u][e] + (p.res ? (float)r8[4 * u + e] : 0.f);
                                if (p.relu) t = t > 0.f ? t : 0.f;
                                o[4 * u + e] = (__bf16)t;
                            }
                        }
                        *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.y) + (int64_t)m * p.ldy + n) = o;
                    }
                }
                if (p.stats) {
#pragma unroll
                    for (int o2 = LPR8; o2 < 64; o2 <<= 1) {
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                ssum8[u][e] += __shfl_xor(ssum8[u][e], o2);
                                ssq8[u][e] += __shfl_xor(ssq8[u][e], o2);
                            }
                    }
                    __syncthreads();                               // every wave has read its C slab: LDS is free
                    float* red = smem;                             // [2 wm][2][BN]
                    if (lrow == 0) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            *reinterpret_cast<f32x4*>(red + (wm * 2 + 0) * BN + wn * WTN + lcol + 4 * u) = ssum8[u];
                            *reinterpret_cast<f32x4*>(red + (wm * 2 + 1) * BN + wn * WTN + lcol + 4 * u) = ssq8[u];
                        }
                    }
                    __syncthreads();
                    for (int c = tid; c < BN; c += 256) {
                        const int nn = n0 + c;
                        if (nn < p.N) {
                            p.stats[((int64_t)tile_m * 2 + 0) * p.N + nn] = red[0 * BN + c] + red[2 * BN + c];
                            p.stats[((int64_t)tile_m * 2 + 1) * p.N + nn] = red[1 * BN + c] + red[3 * BN + c];
                        }
                    }
                }
                return;
            }
        }
        constexpr int LPR = WTN / 4;                        // lanes per row
        constexpr int RPI = 64 / LPR;                       // rows per pass
        const int lrow = lane / LPR, lcol = (lane % LPR) * 4;
        const int n = n0 + wn * WTN + lcol;
        if constexpr (MATH != 2 && BM == 128 && BN == 128) {
            if (p.epilogue == GRL_EPI_SQDIFF) {
                // TRL step: v = relu(acc*scale + shift) is compared with the hoisted conv_f2 output and only
                // the squared difference, summed over each 32-row quarter of a clip, leaves the workgroup
                // (grl_model.py:146-149).  Fixed order: a lane adds its 8 rows of the quarter (rows
                // 4k + lrow), then the 4 lanes of a channel group combine by xor-shuffles 16, 32; the
                // consumer adds the four quarters of a clip.  The tile shape is pinned (128 x 128), so
                // the order -- and a clip's feature row -- does not depend on the batch size.
                f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
                if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
                f32x4 part[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 4
                for (int it = 0; it < WTM / RPI; ++it) {           // RPI = 4, WTM = 64: 16 passes, 8 per quarter
                    const int row = wm * WTM + it * RPI + lrow;
                    const int m = m0 + row;
                    f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * BN + wn * WTN + lcol);
                    v = v * sc + sh;
                    const int64_t rr = (int64_t)(m / p.res_rows) * p.res_gstride + (m % p.res_rows);
                    const f32x4 f2 = *reinterpret_cast<const f32x4*>(p.res + rr * p.ldres + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = (v[e] > 0.f ? v[e] : 0.f) - f2[e];
                        part[it >> 3][e] += t * t;
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        part[h][e] += __shfl_xor(part[h][e], 16);
                        part[h][e] += __shfl_xor(part[h][e], 32);
                    }
                    if (lrow == 0 && n < p.N)
                        *reinterpret_cast<f32x4*>(p.y + (int64_t)((m0 + wm * WTM) / 32 + h) * p.ldy + n) = part[h];
                }
                return;
            }
        }
        if constexpr (MATH != 2 && BM == 128 && BN == 128) {
            // Statistics partials in the order of a SMALLER tile (vec_epi bits 1..2, host: choose_tile): the train-forward
            // statistics GEMMs run on this tile for its speed, but any change of the partial sums' association re-rolls
            // which near-zero pre-activations flip in the parity fixtures, so the sums are formed exactly as
            // the 64 x 64 tile (smode 1: one slab row per 64 rows) or the 128 x 64 tile (smode 2) forms them: a row class
            // r mod 8 is added in row order within a 32-row (1) / 64-row (2) block, classes combine as the xor-shuffle
            // tree of an 8-lane column group, ((0+1)+(2+3)) + ((4+5)+(6+7)), then the blocks / wave rows.
            const int smode = (vec_epi >> 1) & 3;
            if (smode != 0) {
                auto run = [&](auto mode_a) {
                    constexpr bool A = decltype(mode_a)::value;
                    f32x4 as[2][2], aq[2][2];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int c = 0; c < 2; ++c) { as[h][c] = f32x4{0.f, 0.f, 0.f, 0.f}; aq[h][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                    if (n < p.N) {
                        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                        if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
                        if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                            for (int i8 = 0; i8 < 8; ++i8) {
                                const int row = wm * WTM + (hf * 8 + i8) * RPI + lrow;
                                const int m = m0 + row;
                                if (m < p.M) {
                                    f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * BN + wn * WTN + lcol);
                                    if (p.rowscale) v *= p.rowscale[m];
                                    if (p.gbias)
                                        v += *reinterpret_cast<const f32x4*>(p.gbias + (int64_t)(m / p.rows_per_group) * p.N + n);
                                    as[A ? hf : 0][i8 & 1] += v;
                                    aq[A ? hf : 0][i8 & 1] += v * v;
                                    v = v * sc + sh;
                                    if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + (int64_t)m * p.ldres + n);
                                    if (p.relu) {
#pragma unroll
                                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                                    }
                                    *reinterpret_cast<f32x4*>(p.y + (int64_t)m * p.ldy + n) = v;
                                }
                            }
                    }
                    f32x4 ts[2], tq[2];
#pragma unroll
                    for (int h = 0; h < (A ? 2 : 1); ++h) {
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    as[h][c][e] += __shfl_xor(as[h][c][e], o);
                                    aq[h][c][e] += __shfl_xor(aq[h][c][e], o);
                                }
                        ts[h] = as[h][0] + as[h][1];
                        tq[h] = aq[h][0] + aq[h][1];
                    }
                    if constexpr (A) {
                        const int srow = tile_m * 2 + wm;                      // one slab row per 64 rows
                        if (lrow == 0 && n < p.N && (int64_t)srow * 64 < p.M) {
                            *reinterpret_cast<f32x4*>(p.stats + ((int64_t)srow * 2 + 0) * p.N + n) = ts[0] + ts[1];
                            *reinterpret_cast<f32x4*>(p.stats + ((int64_t)srow * 2 + 1) * p.N + n) = tq[0] + tq[1];
                        }
                    } else {
                        __syncthreads();                               // every wave has read its C slab: LDS is free
                        float* red = smem;                             // [2 wm][2][BN]
                        if (lrow == 0) {
                            *reinterpret_cast<f32x4*>(red + (wm * 2 + 0) * BN + wn * WTN + lcol) = ts[0];
                            *reinterpret_cast<f32x4*>(red + (wm * 2 + 1) * BN + wn * WTN + lcol) = tq[0];
                        }
                        __syncthreads();
                        for (int c = tid; c < BN; c += 256) {
                            const int nn = n0 + c;
                            if (nn < p.N) {
                                p.stats[((int64_t)tile_m * 2 + 0) * p.N + nn] = red[0 * BN + c] + red[2 * BN + c];
                                p.stats[((int64_t)tile_m * 2 + 1) * p.N + nn] = red[1 * BN + c] + red[3 * BN + c];
                            }
                        }
                    }
                };
                if (smode == 1) run(std::true_type{});
                else run(std::false_type{});
                return;
            }
        }
        f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};       // train-mode BN: column sums of the raw output
        if (n < p.N) {
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, cn = sh;
            if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
            if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
            if (p.epilogue == GRL_EPI_EUCLID) cn = *reinterpret_cast<const f32x4*>(p.cnorm + n);
            // BatchNorm-backward reduce (GrlGemm.bn_z): this GEMM's output is the gradient of relu?(bn(z) (+res)); it
            // leaves here masked, with its two column sums -- the reduce pass of grl_bn_bwd without a second read
            f32x4 bmu = sh, bis = sh, bms = sh, bmb = sh;
            if constexpr (MATH != 2) {
                if (p.bn_z) {
                    bmu = *reinterpret_cast<const f32x4*>(p.bn_mean + n);
                    bis = *reinterpret_cast<const f32x4*>(p.bn_invstd + n);
                    if (p.bn_mscale) bms = *reinterpret_cast<const f32x4*>(p.bn_mscale + n);
                    if (p.bn_mbeta) bmb = *reinterpret_cast<const f32x4*>(p.bn_mbeta + n);
                }
            }
#pragma unroll EPI_UNROLL
            for (int it = 0; it < WTM / RPI; ++it) {
                const int row = wm * WTM + it * RPI + lrow;
                const int m = m0 + row;
                if (m < p.M) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * BN + wn * WTN + lcol);
                    if (p.epilogue == GRL_EPI_AFFINE) {
                        if (p.rowscale) v *= p.rowscale[m];
                        if (p.gbias)
                            v += *reinterpret_cast<const f32x4*>(
                                p.gbias + (int64_t)(m / p.rows_per_group) * p.N + n);
                        if (p.stats && !p.bn_z) { ssum += v; ssq += v * v; }
                        v = v * sc + sh;
                        if (p.res) {
                            if constexpr (MATH == 2) {
                                const bf16x4 r = *reinterpret_cast<const bf16x4*>(
                                    reinterpret_cast<const __bf16*>(p.res) + (int64_t)m * p.ldres + n);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
                            } else if constexpr (RES_PREFETCH) {
                                v += rpre[it];
                            } else {
                                v += *reinterpret_cast<const f32x4*>(p.res + (int64_t)m * p.ldres + n);
                            }
                        }
                        if (p.relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                        }
                        if constexpr (MATH != 2) {
                            if (p.bn_z) {
                                f32x4 zraw;
                                if constexpr (RES_PREFETCH) zraw = zpre[it];
                                else zraw = *reinterpret_cast<const f32x4*>(p.bn_z + (int64_t)m * p.N + n);
                                const f32x4 zc = zraw - bmu;
                                if (p.bn_bits) {
                                    uint32_t mk;
                                    if constexpr (RES_PREFETCH) mk = bpre[it];
                                    else mk = p.bn_bits[((int64_t)m * p.N + n) >> 2];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = (mk >> e) & 1u ? v[e] : 0.f;
                                } else if (p.bn_mscale) {          // the forward's (z - mean) * scale + beta, term for term
                                    f32x4 t = zc * bms;
                                    if (p.bn_mbeta) t += bmb;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = t[e] > 0.f ? v[e] : 0.f;
                                }
                                ssum += v; ssq += v * (zc * bis);
                            }
                        }
                    } else if (p.epilogue == GRL_EPI_NEGDOT) {
                        v = -v;
                    } else {
                        v = p.rnorm[m] + cn - 2.f * v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = sqrtf(v[e] > 1e-12f ? v[e] : 1e-12f);
                    }
                    if constexpr (MATH == 2) {
                        if (p.out_f32) {                   // fp32 hand-off to the fp32 head kernels
                            *reinterpret_cast<f32x4*>(p.y + (int64_t)m * p.ldy + n) = v;
                        } else {
                            bf16x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.y) + (int64_t)m * p.ldy + n) = o;
                        }
                    } else {
                        *reinterpret_cast<f32x4*>(p.y + (int64_t)m * p.ldy + n) = v;
                    }
                }
            }
        }
        if constexpr (MATH != 2) {
            if (p.stats) {
                // per-channel sum / sum of squares over this tile's rows, fixed order: a lane over its rows (above),
                // the lanes of a wave that own the same 4 channels (xor-shuffles over the row groups), the two wave
                // rows through LDS; one deterministic partial per (tile_m, n): stats[tile_m][0|1][n]
#pragma unroll
                for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ssum[e] += __shfl_xor(ssum[e], o); ssq[e] += __shfl_xor(ssq[e], o); }
                }
                __syncthreads();                               // every wave has read its C slab: LDS is free
                float* red = smem;                             // [2 wm][2][BN]
                if (lrow == 0) {
                    *reinterpret_cast<f32x4*>(red + (wm * 2 + 0) * BN + wn * WTN + lcol) = ssum;
                    *reinterpret_cast<f32x4*>(red + (wm * 2 + 1) * BN + wn * WTN + lcol) = ssq;
                }
                __syncthreads();
                for (int c = tid; c < BN; c += 256) {
                    const int nn = n0 + c;
                    if (nn < p.N) {
                        p.stats[((int64_t)tile_m * 2 + 0) * p.N + nn] = red[0 * BN + c] + red[2 * BN + c];
                        p.stats[((int64_t)tile_m * 2 + 1) * p.N + nn] = red[1 * BN + c] + red[3 * BN + c];
                    }
                }
            }
        }
        return;
    }
    float csum[NT], csq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) csum[j] = csq[j] = 0.f;

#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * WTN + j * 32 + col_l;
        const bool n_ok = n < p.N;
        const int nn = n_ok ? n : p.N - 1;
        const float sc = p.scale ? p.scale[nn] : 1.f;
        const float sh = p.shift ? p.shift[nn] : 0.f;
        const float cn = (p.epilogue == GRL_EPI_EUCLID) ? p.cnorm[nn] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                if (m >= p.M) continue;
                float v = acc[i][j][r];
                if (p.epilogue == GRL_EPI_AFFINE) {
                    if (p.rowscale) v *= p.rowscale[m];
                    if (p.gbias) v += p.gbias[(int64_t)(m / p.rows_per_group) * p.N + nn];
                    if (p.stats) { csum[j] += v; csq[j] += v * v; }
                    v = v * sc + sh;
                    if (p.res && n_ok) v += p.res[(int64_t)m * p.ldres + n];
                    if (p.relu) v = v > 0.f ? v : 0.f;
                } else if (p.epilogue == GRL_EPI_NEGDOT) {
                    v = -v;
                } else {
                    v = p.rnorm[m] + cn - 2.f * v;
                    v = sqrtf(v > 1e-12f ? v : 1e-12f);
                }
                if (n_ok) p.y[(int64_t)m * p.ldy + n] = v;
            }
        }
    }

    if (p.stats) {
        // per-channel sum / sum-of-squares of the raw conv output over this tile's
        // rows (rows >= M contribute nothing: they were skipped above).  Combine the
        // two lane halves, then the two wave rows through LDS, and write one
        // deterministic partial per (tile_m, n): stats[tile_m][0|1][n].
        __syncthreads();                       // tiles in LDS are dead now
        float* red = smem;                     // [2 wm][2][BN]
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float s = csum[j] + __shfl_xor(csum[j], 32);
            float q2 = csq[j] + __shfl_xor(csq[j], 32);
            if (fhalf == 0) {
                const int c = wn * WTN + j * 32 + col_l;
                red[(wm * 2 + 0) * BN + c] = s;
                red[(wm * 2 + 1) * BN + c] = q2;
            }
        }
        __syncthreads();
        for (int c = tid; c < BN; c += 256) {
            const int n = n0 + c;
            if (n < p.N) {
                p.stats[((int64_t)tile_m * 2 + 0) * p.N + n] = red[0 * BN + c] + red[2 * BN + c];
                p.stats[((int64_t)tile_m * 2 + 1) * p.N + n] = red[1 * BN + c] + red[3 * BN + c];
            }
        }
    }
    };
    if constexpr ((GRL_GEMM_KO & 4) != 0) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 1.2345e-30f) p.y[0] = s;
    } else {
        epilogue(cur_m0, cur_n0, cur_tile_m);
    }
    if (next_t >= num_tiles) break;
    t = next_t;
    __syncthreads();          // every wave is done with the C staging before the stages are rewritten
    dma_stage(0, 0);
    }
}

// split-K finish: y = epilogue( seg[last] + (((seg[0] + seg[1]) + ...) + seg[last - 1]) ) -- the K-blocked chain's own
// association (the kernel above adds the running total to the LAST segment's accumulator), so the result is the
// one-workgroup K-blocked result bit for bit.  Epilogue arithmetic = the affine epilogue's: v*scale + shift (+res), ReLU.
__global__ void splitk_finish_kernel(const GrlGemm p, const int nseg) {
    const int64_t total = (int64_t)p.M * p.N, plane = total;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / p.N), n = (int)(i - (int64_t)m * p.N);
        float tot = 0.f;
        for (int sgm = 0; sgm + 1 < nseg; ++sgm) tot += p.splitk_ws[(int64_t)sgm * plane + i];
        float v = p.splitk_ws[(int64_t)(nseg - 1) * plane + i];
        v += tot;
        const float sc = p.scale ? p.scale[n] : 1.f, sh = p.shift ? p.shift[n] : 0.f;
        v = v * sc + sh;
        if (p.res) v += p.res[(int64_t)m * p.ldres + n];
        if (p.relu) v = v > 0.f ? v : 0.f;
        p.y[(int64_t)m * p.ldy + n] = v;
    }
}

// Which launches split K (and how much scratch that takes): fp32 K-blocked, dense, plain affine epilogue, at most 256
// rows and so few 64 x 64 tiles that one workgroup per tile leaves the chip empty.
int64_t splitk_floats(const GrlGemm& d) {
    if (d.math != GRL_MATH_F32 || !d.kblock || d.conv || d.stats || d.gbias || d.rowscale || d.epilogue != GRL_EPI_AFFINE)
        return 0;
    if (d.M > 256 || d.K <= SEG_STAGES * BK) return 0;
    const int64_t tiles = (int64_t)((d.M + 63) / 64) * ((d.N + 63) / 64);
    if (tiles > 128) return 0;
    const int nseg = (d.K + SEG_STAGES * BK - 1) / (SEG_STAGES * BK);
    return (int64_t)nseg * d.M * d.N;
}

struct TileChoice { int bm, bn, smode = 0; };
int g_force_tile = 0;                 // grl_gemm_force_tile: (bm << 16) | bn, 0 = automatic

TileChoice legacy_tile(const GrlGemm& d) {
    if (d.epilogue == GRL_EPI_SQDIFF) return {128, 128};          // pinned: fixed reduction order (see the epilogue)
    // GRL_GEMM_TILE=128x128|128x64|64x64 forces a tile (kernel tuning only; read once per process)
    static const TileChoice forced = [] {
        TileChoice f{0, 0, 0};
        if (const char* e = getenv("GRL_GEMM_TILE")) {
            int bm = 0, bn = 0;
            if (sscanf(e, "%dx%d", &bm, &bn) == 2 && (bm == 128 || bm == 64) && (bn == 128 || bn == 64) &&
                !(bm == 64 && bn == 128))
                f = {bm, bn};
        }
        return f;
    }();
    if (g_force_tile) return {g_force_tile >> 16, g_force_tile & 0xffff};
    if (forced.bm) return forced;
    // Measured on MI355X (tools/gemm_bench.py, profiles/r01_gemm_tiles.txt): the 128x128
    // tile wins when the K loop is long (>= 1024: 131-135 TFLOP/s); short-K layers are
    // prologue/epilogue bound and want more, smaller workgroups per CU (K <= 128: 64x64,
    // K <= 512: 128x64).  Small grids fall back to smaller tiles so that every CU (256,
    // two resident workgroups each) has work.
    auto tiles = [&](int bm, int bn) {
        return (int64_t)((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn);
    };
    if (d.N <= 64) return tiles(128, 64) >= 512 ? TileChoice{128, 64} : TileChoice{64, 64};
    // Round 3 (tools/gemm_tile_ab.py, per shape in child processes): with LDS-DMA staging and the vectorized epilogue
    // the 128 x 128 tile now wins on the SHORT-K layers too wherever it still gives every CU a tile -- 16384x1024x256
    // (+res) 104 -> 100 us, 16384x2048x512 (+res) 357 -> 338, 65536x128x512 93 -> 88, 16384x256x2304 188 -> 179,
    // 262144x256x64 157 -> 140 -- except the K <= 128 layers WITH a residual (its rows are not prefetched on this tile:
    // 180 -> 218 us) and 128-tile grids (4096x512x2048: 87 -> 143 us).  Results do not depend on the tile (one k-ordered
    // chain per output).  Not for the statistics GEMMs of the training forward HERE: their partial sums would cover 128
    // rows instead of 64 and the 4 x 8 fixture's outputs leave the 1e-4 pin -- choose_tile() below moves
    // them to the wide tile with the small tile's summation order instead.
    static const bool wide_on = [] { const char* e = getenv("GRL_GEMM_WIDE"); return !e || atoi(e) != 0; }();
    static const bool wide3_on = [] { const char* e = getenv("GRL_GEMM_WIDE3"); return !e || atoi(e) != 0; }();   // split-bf16 / bf16 products too
    // (the data-gradient GEMMs that carry a BatchNorm-backward reduce -- stats + bn_z -- take it too: their column sums
    // feed gradients, not ReLU masks, so their association is free; split-bf16 products: mixed 43.4 -> 43.05 ms, fp32 +-0)
    // (round 4: a fused BatchNorm-backward reduce reads a row of z (+ mask word) per output row; the 64-wide tiles request
    // those rows before the accumulators' LDS round trip (zpre), the 128 x 128 tile has no registers for that -- the
    // short-K, output-heavy data-gradient GEMMs that carry one go back to the narrow tiles.  GRL_GEMM_BNZ_NARROW=0: off)
    static const int bnz_narrow_k = [] { const char* e = getenv("GRL_GEMM_BNZ_NARROW"); return e ? atoi(e) : 512; }();
    const bool bnz_narrow = d.bn_z && d.math == GRL_MATH_F32 && d.K <= bnz_narrow_k;
    if (wide_on && !bnz_narrow && (!d.stats || d.bn_z) && (d.math == GRL_MATH_F32 || (wide3_on && d.math != GRL_MATH_BF16S)) && d.N >= 128 &&
        tiles(128, 128) >= (d.conv ? 448 : 256) && !(d.res && d.K <= (d.stats ? 128 : 512)))     // (conv with 256..447 wide tiles: 512+ tiles of 128 x 64 -- 16384x256x2304: 159 -> 152 us)
        return {128, 128};
    // (round 5, per shape with grl_gemm_force_tile: the K <= 128 expansion layer that carries a residual and fills the chip
    //  with wide tiles -- 65536x512x128 +res: 64x64 140.7, 128x64 129.4, 128x128 125.7 us; 262144x256x64 +res 188.8 / 174.2 /
    //  167.9 -- eval form only: the statistics GEMMs' tile is part of their partial sums' order)
    if (d.K <= 128 && d.res && !d.stats && d.math == GRL_MATH_F32 && d.N >= 128 && tiles(128, 128) >= 448) return {128, 128};
    if (d.K <= 128) return {64, 64};
    // (round 4, re-measured per shape with grl_gemm_force_tile after the hand-scheduled loop: a residual-carrying K <= 512
    // layer is better off on the 128 x 64 tile, which requests its residual rows before the LDS round trip -- 16384x2048x512
    // 299 -> 284 us, 16384x1024x256 86 -> 83 --, and from one dense 128 x 64 tile per CU on that tile beats twice as many
    // 64 x 64 ones -- 4096x512x512 25 -> 22 us.  The statistics GEMMs keep their thresholds: their tile is part of the
    // partial sums' order)
    if (d.K <= 512) return tiles(128, 64) >= ((d.stats || d.conv) ? 448 : 256) ? TileChoice{128, 64} : TileChoice{64, 64};
    if (tiles(128, 128) >= 448) return {128, 128};
    // (128-row tiles of a dense operand stage by LDS-DMA: from one tile per CU on they beat four times as many 64 x 64
    // tiles -- 4096 x 512 x 2048: 84.8 vs 95.7 us)
    if (tiles(128, 64) >= (d.conv ? 448 : 256)) return {128, 64};
    return {64, 64};
}

bool vec_epilogue_ok(const GrlGemm& d) {
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    return d.N % 4 == 0 && d.ldy % 4 == 0 && al16(d.y) && (!d.res || (d.ldres % 4 == 0 && al16(d.res))) && al16(d.scale) &&
           al16(d.shift) && al16(d.gbias) && al16(d.cnorm);
}

// The tile a launch runs on.  The train-forward statistics GEMMs (fp32 storage: exact and split-bf16 products) that the
// rule above keeps on 64 x 64 / 128 x 64 tiles for their statistics' sake take the 128 x 128 tile too wherever it fills
// the chip (>= 448 tiles) -- with the partial sums formed
// in the smaller tile's order (TileChoice.smode -> the kernel's statistics epilogue), so outputs AND statistics slabs are
// bit for bit the smaller tile's (tested), and the slab keeps the smaller tile's row count.  GRL_GEMM_WIDE_STATS=0: off.
TileChoice choose_tile(const GrlGemm& d) {
    TileChoice t = legacy_tile(d);
    const char* const ws_env = getenv("GRL_GEMM_WIDE_STATS");             // (read per call: the parity test toggles it)
    const bool wide_stats = !ws_env || atoi(ws_env) != 0;
    static const bool forced_env = getenv("GRL_GEMM_TILE") != nullptr;
    const bool forced = forced_env || g_force_tile != 0;       // (a forced tile -- environment or grl_gemm_force_tile -- is final)
    if (!wide_stats || forced || !d.stats || d.bn_z || d.math == GRL_MATH_BF16S || d.epilogue != GRL_EPI_AFFINE || d.N < 128 ||
        !vec_epilogue_ok(d) || ((uintptr_t)d.stats & 15) != 0)
        return t;
    const int64_t tiles128 = (int64_t)((d.M + 127) / 128) * ((d.N + 127) / 128);
    if (tiles128 < 448) return t;                    // (>= 256: measured, no gain)
    if (t.bm == 64 && t.bn == 64) return {128, 128, 1};
    if (t.bm == 128 && t.bn == 64) return {128, 128, 2};
    return t;
}

// Workgroups the chip keeps resident for one kernel instantiation = the persistent grid
// (CUs x occupancy, a multiple of 8 so that a workgroup's tiles stay on its XCD's run).
// GRL_GEMM_GRID overrides it (kernel tuning only; 0 = one workgroup per tile).
int resident_workgroups(const void* kernel, size_t lds) {
    if (lds > 65536) (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    static const int forced_grid = [] {                       // GRL_GEMM_GRID: tuning only, read once
        const char* e = getenv("GRL_GEMM_GRID");
        return e ? atoi(e) : -1;
    }();
    if (forced_grid >= 0) return forced_grid > 0 ? (forced_grid + 7) / 8 * 8 : 0x7fffffff;
    int dev = 0, cus = 256, occ = 2;
    if (hipGetDevice(&dev) == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, lds) != hipSuccess || occ < 1) occ = 2;
    (void)hipGetLastError();
    return (cus * occ + 7) / 8 * 8;
}

template <auto KERNEL>
void launch_kernel(const GrlGemm& d, hipStream_t s, size_t lds, int tiles_n, int num_tiles, int vec_epi) {
    // persistent grid = the workgroups the CURRENT device keeps resident, cached per instantiation and
    // per device (one process normally drives one GPU, but nothing here assumes it)
    static int slots_of[16] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& cached = slots_of[dev & 15];
    if (cached == 0) cached = resident_workgroups((const void*)KERNEL, lds);
    const int slots = cached;
    hipLaunchKernelGGL(KERNEL, dim3(num_tiles < slots ? num_tiles : slots), dim3(256), lds, s, d, tiles_n, num_tiles,
                       vec_epi);
}

template <int BM, int BN, int MATH>
int launch_math(const GrlGemm& d, hipStream_t s, int smode) {
    const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
    const int num_tiles = tiles_m * tiles_n;
    constexpr int NST = (GRL_GEMM_PIPE && GRL_GEMM_RING3 && MATH == 2 && BM == 128 && BN == 64) ? 3 : 2;      // (the kernel's RING)
    constexpr size_t stage_bytes = (MATH == 0 || MATH == 2) ? (size_t)NST * (BM + BN) * BK * sizeof(float)
                                                            : (size_t)2 * (MATH == 3 ? 2 : 1) * (BM + BN) * 64;
    constexpr size_t c_bytes = (size_t)BM * BN * sizeof(float);       // epilogue staging
    constexpr size_t lds = stage_bytes > c_bytes ? stage_bytes : c_bytes;
    static const bool panel_on = [] { const char* e = getenv("GRL_GEMM_PANEL"); return !e || atoi(e) != 0; }();
    const int vec_epi = (vec_epilogue_ok(d) ? 1 | (smode << 1) : 0) | (panel_on ? 8 : 0);     // (bits 1..2: the statistics' order, choose_tile; bit 3: column-panel tile walk)
    if (MATH == 2 && !(vec_epi & 1))
        return grl_fail(GRL_EINVAL, "gemm bf16s: y/res/scale/shift/gbias must be 16-byte aligned");
    if (MATH == 2 && d.bn_z && (d.M % BM || d.N % BN))
        return grl_fail(GRL_EINVAL, "gemm bf16s: bn_z on a %d x %d tile needs M, N multiples of the tile", BM, BN);
    constexpr bool CAN_SEG = MATH == 0;
    const bool seg = CAN_SEG && d.kblock && d.K > SEG_STAGES * BK;
    static const bool dma_conv_on = [] { const char* e = getenv("GRL_GEMM_DMA_CONV"); return !e || atoi(e) != 0; }();
    if (d.conv) {
        constexpr bool CAN_DMA_CONV = (MATH == 0 || MATH == 2) && BM == 128;
        if (CAN_DMA_CONV && dma_conv_on && d.ldw < (1 << 22) && d.kh * d.kw <= 32) {     // (one validity bit per tap)
            if (seg) launch_kernel<gemm_f32_kernel<BM, BN, true, MATH, CAN_SEG, CAN_DMA_CONV>>(d, s, lds, tiles_n, num_tiles, vec_epi);
            else launch_kernel<gemm_f32_kernel<BM, BN, true, MATH, false, CAN_DMA_CONV>>(d, s, lds, tiles_n, num_tiles, vec_epi);
        } else if (seg) launch_kernel<gemm_f32_kernel<BM, BN, true, MATH, CAN_SEG>>(d, s, lds, tiles_n, num_tiles, vec_epi);
        else launch_kernel<gemm_f32_kernel<BM, BN, true, MATH, false>>(d, s, lds, tiles_n, num_tiles, vec_epi);
    } else {
        // dense fp32 with a long K loop: LDS-DMA staging (GRL_GEMM_DMA=0 switches it off: tuning only)
        static const bool dma_on = [] { const char* e = getenv("GRL_GEMM_DMA"); return !e || atoi(e) != 0; }();
        constexpr bool CAN_DMA = (MATH == 0 || MATH == 2) && BM == 128;     // 128-byte operand rows (fp32 x 32 / bf16 x 64)
        // (the hand-scheduled loop addresses a tile's rows with 32-bit byte offsets from its first row: 127 rows x ld x 4 B)
        static const int dma_mink = [] { const char* e = getenv("GRL_GEMM_DMA_MINK"); return e ? atoi(e) : 256; }();
        if (CAN_DMA && dma_on && d.K >= dma_mink && d.lda < (1 << 22) && d.ldw < (1 << 22)) {
            if (seg) launch_kernel<gemm_f32_kernel<BM, BN, false, MATH, CAN_SEG, CAN_DMA>>(d, s, lds, tiles_n, num_tiles, vec_epi);
            else launch_kernel<gemm_f32_kernel<BM, BN, false, MATH, false, CAN_DMA>>(d, s, lds, tiles_n, num_tiles, vec_epi);
        } else if (seg) launch_kernel<gemm_f32_kernel<BM, BN, false, MATH, CAN_SEG>>(d, s, lds, tiles_n, num_tiles, vec_epi);
        else launch_kernel<gemm_f32_kernel<BM, BN, false, MATH, false>>(d, s, lds, tiles_n, num_tiles, vec_epi);
    }
    return grl_check_launch("grl_conv_gemm_f32");
}

template <int BM, int BN>
int launch(const GrlGemm& d, hipStream_t s, int smode = 0) {
    if (d.math == GRL_MATH_BF16S) return launch_math<BM, BN, 2>(d, s, 0);
    if (d.math == GRL_MATH_BF16X3) return launch_math<BM, BN, 3>(d, s, smode);
    if (d.math == GRL_MATH_BF16) return launch_math<BM, BN, 1>(d, s, smode);
    return launch_math<BM, BN, 0>(d, s, smode);
}

int validate(const GrlGemm& d) {
    if (!d.a || !d.w || !d.y) return grl_fail(GRL_EINVAL, "gemm: null operand");
    if (d.M <= 0 || d.N <= 0 || d.K <= 0) return grl_fail(GRL_EINVAL, "gemm: empty shape");
    if (d.K % BK) return grl_fail(GRL_EINVAL, "gemm: K must be a multiple of 32");
    if (d.ldw % 4 || ((uintptr_t)d.w & 15) || ((uintptr_t)d.a & 15))
        return grl_fail(GRL_EINVAL, "gemm: operands must be 16-byte aligned with ld % 4 == 0");
    if (d.conv) {
        if (d.C % BK) return grl_fail(GRL_EINVAL, "conv: C must be a multiple of 32");
        if (d.K != d.kh * d.kw * d.C) return grl_fail(GRL_EINVAL, "conv: K != kh*kw*C");
        if (d.M % (d.Ho * d.Wo)) return grl_fail(GRL_EINVAL, "conv: M must be nimg*Ho*Wo");
    } else if (d.lda % 4) {
        return grl_fail(GRL_EINVAL, "gemm: lda % 4 != 0");
    }
    if (d.math != GRL_MATH_F32 && d.math != GRL_MATH_BF16 && d.math != GRL_MATH_BF16X3 &&
        d.math != GRL_MATH_BF16S)
        return grl_fail(GRL_EINVAL, "gemm: unknown math mode");
    if (d.math == GRL_MATH_BF16S) {
        if (d.K % 64 || (d.conv && d.C % 64)) return grl_fail(GRL_EINVAL, "gemm bf16s: K (and C) must be multiples of 64");
        if (d.epilogue != GRL_EPI_AFFINE && !(d.epilogue == GRL_EPI_SQDIFF && grl_gemm_bf16_256_takes(d)))
            return grl_fail(GRL_EINVAL, "gemm bf16s: affine epilogue (or SQDIFF with M, N % 256 == 0, K % 64 == 0, res_rows % 32 == 0)");
        if (d.stats && (d.N % 8 || d.out_f32 || d.rowscale))
            return grl_fail(GRL_EINVAL, "gemm bf16s: stats need N % 8 == 0, bf16 output, no rowscale");
        if (d.N % 4 || d.ldy % 4 || (d.res && d.ldres % 4) || d.lda % 8 || d.ldw % 8 ||
            ((uintptr_t)d.y & 15) || ((uintptr_t)d.res & 15))
            return grl_fail(GRL_EINVAL, "gemm bf16s: N, ldy, ldres % 4, lda, ldw % 8, aligned pointers");
    }
    if (d.epilogue == GRL_EPI_EUCLID && (!d.rnorm || !d.cnorm))
        return grl_fail(GRL_EINVAL, "gemm: EUCLID needs rnorm and cnorm");
    if (d.gbias && d.rows_per_group <= 0) return grl_fail(GRL_EINVAL, "gemm: rows_per_group");
    if (d.epilogue == GRL_EPI_SQDIFF) {
        if (d.conv || d.stats || d.rowscale || d.gbias || !d.res)
            return grl_fail(GRL_EINVAL, "gemm: SQDIFF epilogue is dense, with res = the f2 tensor");
        if (d.math == GRL_MATH_BF16S) return GRL_OK;          // (checked above: the 256 x 256 kernel takes it)
        if (d.M % 128 || d.N % 128 || d.res_rows <= 0 || d.res_rows % 32 || d.ldres % 4 || d.ldy % 4 ||
            ((uintptr_t)d.res & 15) || ((uintptr_t)d.y & 15) || ((uintptr_t)d.scale & 15) || ((uintptr_t)d.shift & 15))
            return grl_fail(GRL_EINVAL, "gemm: SQDIFF needs M, N % 128 == 0, res_rows % 32 == 0, aligned operands");
    }
    if (d.kblock && d.math != GRL_MATH_F32) return grl_fail(GRL_EINVAL, "gemm: kblock needs GRL_MATH_F32");
    if (d.bn_z) {
        if (d.epilogue != GRL_EPI_AFFINE || d.kblock || !d.stats || !d.bn_mean || !d.bn_invstd)
            return grl_fail(GRL_EINVAL, "gemm: bn_z needs the AFFINE epilogue, no kblock, stats, bn_mean, bn_invstd");
        // bf16 storage (round 5): only the branch-free interior epilogue carries the reduce -- every tile must be one
        if (d.math == GRL_MATH_BF16S && (d.M % 128 || (d.N % 128 && d.N != 64) || d.out_f32 || d.ldy % 8 || (d.res && d.ldres % 8)))
            return grl_fail(GRL_EINVAL, "gemm bf16s: bn_z needs M %% 128 == 0, N %% 128 == 0 (or N == 64), bf16 output");
        if (d.N % 4 || d.ldy % 4 || (d.res && d.ldres % 4) || ((uintptr_t)d.y & 15) || ((uintptr_t)d.res & 15) ||
            ((uintptr_t)d.bn_z & 15) || ((uintptr_t)d.bn_mean & 15) || ((uintptr_t)d.bn_invstd & 15) ||
            ((uintptr_t)d.bn_mscale & 15) || ((uintptr_t)d.bn_mbeta & 15) || ((uintptr_t)d.bn_bits & 3) ||
            ((uintptr_t)d.scale & 15) || ((uintptr_t)d.shift & 15) || d.gbias || d.rowscale)
            return grl_fail(GRL_EINVAL, "gemm: bn_z needs N, ldy, ldres % 4 == 0 and 16-byte aligned operands");
    }
    return GRL_OK;
}

}  // namespace

extern "C" int grl_gemm_force_tile(int bm, int bn) {
    const int prev = g_force_tile;
    if (bm == 0 && bn == 0) { g_force_tile = 0; return prev; }
    if (!((bm == 128 || bm == 64) && (bn == 128 || bn == 64)) || (bm == 64 && bn == 128))
        return grl_fail(GRL_EINVAL, "gemm_force_tile: tiles are 128x128, 128x64, 64x64");
    g_force_tile = (bm << 16) | bn;
    return prev;
}

extern "C" int grl_conv_gemm_f32_stat_rows(const GrlGemm* desc) {
    if (!desc) return grl_fail(GRL_EINVAL, "null desc");
    GrlGemm probe = *desc;
    if (!probe.stats) probe.stats = reinterpret_cast<float*>((uintptr_t)16);         // (asked before the slab exists; the
    if (desc->math == GRL_MATH_BF16S) {                                               //  tile choice depends on `stats`)
        if (grl_gemm_bf16_256_takes(probe)) return grl_gemm_bf16_256_stat_rows(probe);      // two slab rows per 256-row tile
    }
    const TileChoice t = choose_tile(probe);
    const int rows_per = t.smode == 1 ? 64 : t.bm;          // (smode 1: the 64 x 64 tile's slab, one row per 64 rows)
    return (desc->M + rows_per - 1) / rows_per;
}

int grl_gemm_validate(const GrlGemm& d) { return validate(d); }      // (common.h: the grouped entry point checks every descriptor)

extern "C" int64_t grl_conv_gemm_f32_workspace_floats(const GrlGemm* desc) {
    return desc ? splitk_floats(*desc) : 0;
}

extern "C" int grl_conv_gemm_f32(const GrlGemm* desc, void* stream) {
    if (!desc) return grl_fail(GRL_EINVAL, "null desc");
    const GrlGemm& d = *desc;
    if (int e = validate(d)) return e;
    hipStream_t s = (hipStream_t)stream;
    if (const int64_t need = splitk_floats(d); need > 0 && d.splitk_ws && d.splitk_ws_floats >= need &&
        ((uintptr_t)d.splitk_ws & 15) == 0) {
        const int tiles_m = (d.M + 63) / 64, tiles_n = (d.N + 63) / 64, nseg = (int)(need / ((int64_t)d.M * d.N));
        constexpr size_t lds = (size_t)2 * (64 + 64) * BK * sizeof(float);          // (>= the 64 x 64 C staging)
        hipLaunchKernelGGL((gemm_f32_kernel<64, 64, false, 0, false, false, true>), dim3(tiles_m * tiles_n, nseg), dim3(256),
                           lds, s, d, tiles_n, tiles_m * tiles_n, d.N % 4 == 0 ? 1 : 0);
        const int64_t total = (int64_t)d.M * d.N;
        hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)),
                           dim3(256), 0, s, d, nseg);
        return grl_check_launch("grl_conv_gemm_f32 (split-K)");
    }
    if (d.math == GRL_MATH_BF16S) {
        const int r = grl_gemm_bf16_256(d, s);       // large-tile LDS-DMA kernel where it can fill the chip
        if (r != 0) return r < 0 ? r : GRL_OK;
    }
    const TileChoice t = choose_tile(d);
    if (t.bm == 128 && t.bn == 128) return launch<128, 128>(d, s, t.smode);
    if (t.bm == 128 && t.bn == 64) return launch<128, 64>(d, s);
    return launch<64, 64>(d, s);
}
