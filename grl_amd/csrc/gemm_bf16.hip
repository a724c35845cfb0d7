// bf16-storage GEMM / implicit-GEMM convolution for gfx950 (MI355X), large-tile form.
//
//   Y[M][N] = epilogue( A[M][K] . W[N][K]^T ),  A, W, residual, Y bf16 in HBM, fp32 accumulate
//
// This is the GRL_MATH_BF16S datapath of grl_conv_gemm_f32 for the shapes that can fill the chip
// with 256 x 256 output tiles (BASELINE configs[2]: every N >= 256 layer from layer 3 on, the GCE
// convs and the TRL 2048-wide 1x1s).  Round 1's 128 x 128 x 64 tile moved 32 KB out of L2 per
// 2*128*128*64 FLOP = 64 FLOP/B and saturated the L2 -> CU path at ~0.7 PFLOP/s; this tile halves
// the bytes per FLOP, and its staging costs no VGPRs and no ds_write:
//
//   * workgroup = 8 waves (2 x 4), wave tile 128 x 64 = 4 x 2 MFMA tiles of 32 x 32
//     (v_mfma_f32_32x32x16_bf16, 128 accumulator registers), one workgroup per CU;
//   * K stage = 64 bf16 = 128-byte rows, two stages of (256 + 256) rows = 128 KiB of LDS;
//   * staging is LDS-DMA (`global_load_lds_dwordx4`): a wave-instruction writes 8 rows x 128 B
//     linearly, the XOR swizzle of the 16-byte chunks ((row >> 1) & 7, conflict-free
//     ds_read_b128) is applied to the per-lane SOURCE address and again on the fragment read;
//   * implicit GEMM: a lane's source is row (image, oy, ox) at the stage's tap -- out-of-image
//     taps read a zero page -- so 1x1 / 3x3, stride 1 / 2 convolutions need no im2col;
//   * the loads of stage t+1 are issued before the MFMAs of stage t and retire at the one
//     barrier per stage (the compiler's vmcnt(0) in front of it);
//   * epilogue: each wave parks 32 x 64 accumulator blocks in its own LDS slab and reads them
//     back row-major: 8 channels = 16 bytes per lane for scale / shift / per-clip bias /
//     residual / ReLU / bf16 store.
//
// Reference call sites of the convolutions it runs: reid/models/resnets1.py:62-68,73-93,
// basebranch.py:42-50, grl_model.py:56-64,95-121.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "../../include/grl_hip.h"
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int TB = 256;                       // tile rows = tile cols
constexpr int ROWB = 128;                     // bytes of one staged row (64 bf16)
constexpr int STAGE = 2 * TB * ROWB;          // A + B tile of one stage: 64 KiB

__device__ uint4 g_zero_row[8];               // 128 zero bytes: source of out-of-image taps

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// one LDS-DMA wave-instruction: lane l's 16 bytes at `g` land at lds + 16*l
__device__ __forceinline__ void glds16(const char* g, char* lds) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds, 16, 0, 0);
}

// STATS (bf16-storage TRAINING, round 3): per-channel sum / sum of squares of the raw fp32 accumulator (+ per-clip
// bias) for train-mode BatchNorm, one partial per 128-row wave row: stats[2 * tile_m + wr][0|1][n].  A lane sums
// its 16 rows, the 8 lanes that own the same 8 channels combine by xor-shuffles -- a fixed order, no cross-wave
// step (each wave row writes its own slab row), so the epilogue keeps its one-barrier-per-tile structure.
// SQD (GRL_EPI_SQDIFF, round 3): the TRL step's d = GAP((ReLU(conv_f1(memo)) - f2_t)^2) reduced in the epilogue --
// v = bf16(relu(acc*scale + shift)) (the value the unfused pipeline stores), minus the hoisted conv_f2 output read through
// `res` (row mapping res_rows / res_gstride), squared, summed per 32-row block in a fixed order (a lane over its 4 rows,
// then the 8 lanes that own the same 8 channels by xor-shuffles) and written as fp32 partials y[M/32][N]: conv_f1's
// output never reaches HBM (grl_model.py:146-149), as in the fp32 kernel.
// BNZ (round 5; with STATS): the BatchNorm-backward reduce of GrlGemm.bn_z in the interior epilogue -- 1: mask from the
// recorded ReLU bits (one byte per eight outputs), 2: from z itself ((z - mean) * mscale + mbeta > 0), 3: no mask.  The two
// statistics rows of a tile then hold sum g and sum g * (z - mean) * invstd of the masked gradient g the kernel stores.
template <bool CONV, bool STATS = false, bool SQD = false, bool RES = false, bool GBIAS = false, int BNZ = 0>
__global__ __launch_bounds__(512, 2) void gemm_bf16_256_kernel(const GrlGemm p, const int tiles_n,
                                                               const int num_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- staging rows: wave w fills the 8-row blocks w, w+8, w+16, w+24 of both operands ----
    const int srow = lane >> 3, schunk = lane & 7;
    const int sw = (schunk ^ (((srow >> 1) + 4 * (wave & 1)) & 7)) << 4;   // swizzled source chunk (bytes)
    // 32-bit byte offsets from the (wave-uniform) operand bases: half the address registers of 64-bit
    // pointers (the dispatcher guarantees both operands are < 4 GiB); conv rows also carry their
    // window origin (iy0 + 1, ix0 + 1), 16 bits each
    unsigned aoff[4], boff[4], yx0[4];
    const char* const a8 = reinterpret_cast<const char*>(p.a);
    const char* const w8 = reinterpret_cast<const char*>(p.w);
    const char* const zrow = reinterpret_cast<const char*>(g_zero_row) + sw;
    int m0, n0;
    int cur_tile_m = 0;

    // Persistent workgroups (one per CU) walk tiles t = blockIdx.x, + gridDim.x, ...  XCD-aware
    // order: blocks b, b+8, ... share an XCD (gridDim.x is a multiple of 8 whenever a workgroup
    // has more than one tile); each XCD gets a contiguous run of tiles, column tile fastest, so
    // its 32 CUs share A row panels and sweep W together.
    auto setup_tile = [&](int t) {
        int bid = t;
        {
            const int q = num_tiles >> 3, r = num_tiles & 7, xcd = bid & 7;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        }
        const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
        m0 = tile_m * TB;
        n0 = tile_n * TB;
        if constexpr (STATS) cur_tile_m = tile_m;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (wave + 8 * i) * 8 + srow;
            int m = m0 + r;
            m = m < p.M ? m : p.M - 1;                           // edge rows are loaded, never stored
            if (CONV) {
                const int hw = p.Ho * p.Wo;
                const int img = m / hw, rem = m - img * hw;
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                aoff[i] = (unsigned)(((int64_t)img * p.H * p.W * p.C) * 2 + sw);
                yx0[i] = ((unsigned)(oy * p.stride - p.pad + 1) << 16) | (unsigned)(ox * p.stride - p.pad + 1);
            } else {
                aoff[i] = (unsigned)((int64_t)m * p.lda * 2 + sw);
                yx0[i] = 0;
            }
            int n = n0 + r;
            n = n < p.N ? n : p.N - 1;
            boff[i] = (unsigned)((int64_t)n * p.ldw * 2 + sw);
        }
    };

    // One stage = 8 LDS-DMA wave-instructions per wave: `stage_piece(s, kt, i)` issues 8-row block i of A
    // and of W (two instructions); `stage()` issues all four pieces back to back (prologue, next tile).
    int tap_ky = 0, tap_kx = 0, tap_c0 = 0;
    auto stage_tap = [&](int kt) {
        tap_c0 = kt * 64;
        if (CONV) {
            const int tap = tap_c0 / p.C;                    // wave-uniform: a stage lies inside one tap
            tap_c0 -= tap * p.C;
            tap_ky = tap / p.kw;
            tap_kx = tap - tap_ky * p.kw;
        }
    };
    auto stage_piece = [&](int s, int kt, int i) {
        char* const As = smem + s * STAGE;
        char* const Bs = As + TB * ROWB;
        const char* src;
        if (CONV) {
            const int iy = (int)(yx0[i] >> 16) - 1 + tap_ky, ix = (int)(yx0[i] & 0xffff) - 1 + tap_kx;
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            src = ok ? a8 + (aoff[i] + (unsigned)(((iy * p.W + ix) * p.C + tap_c0) * 2)) : zrow;
        } else {
            src = a8 + (aoff[i] + (unsigned)kt * ROWB);
        }
        glds16(src, As + (wave + 8 * i) * 1024);
        glds16(w8 + (boff[i] + (unsigned)kt * ROWB), Bs + (wave + 8 * i) * 1024);
    };
    auto stage = [&](int s, int kt) {
        stage_tap(kt);
#pragma unroll
        for (int i = 0; i < 4; ++i) stage_piece(s, kt, i);
    };

    const int frow = lane & 31, fhalf = lane >> 5;
    const int fsw = (frow >> 1) & 7;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;       // LDS byte address of the stage buffers
    const int nk = p.K / 64;
    // epilogue slab of this wave: 32 x 64 fp32 in the SECOND stage buffer (the next tile's first
    // stage is already landing in the first one); 16-byte chunks XOR-swizzled by (row >> 1) & 1
    float* const Cs = reinterpret_cast<float*>(smem + STAGE) + wave * (32 * 64);
    const int lrow = lane >> 3, lcol = (lane & 7) * 8;                        // 8 lanes per 64-wide row
    __bf16* const y16 = reinterpret_cast<__bf16*>(p.y);
    const __bf16* const r16 = reinterpret_cast<const __bf16*>(p.res);

    int t = blockIdx.x;
    setup_tile(t);
    stage(0, 0);
    for (;;) {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        __syncthreads();          // stage 0 of this tile has landed; every wave is out of the previous epilogue
        // One stage.  Fragments are double-buffered in registers: k-step q+1's six ds_read_b128 are
        // issued BEFORE k-step q's eight MFMAs and waited for with a counted lgkmcnt(6), and the DMA
        // pieces of stage kt+1 (`more`, compile time) are issued in the shadow of the LDS latency /
        // behind queued MFMAs -- the two waves of a SIMD run in lockstep behind the stage barrier, so
        // anything both of them do back to back leaves the matrix pipe idle.  hipcc sinks plain loads
        // below the MFMAs again (shortest live ranges), hence inline asm reads, explicit waits and
        // sched_barriers fencing the MFMA groups.
        auto do_stage = [&](auto more, int kt) {
            const int cur = kt & 1;
            const unsigned a_lds = lds0 + cur * STAGE + (wr * 128 + frow) * ROWB;
            const unsigned b_lds = lds0 + cur * STAGE + TB * ROWB + (wc * 64 + frow) * ROWB;
            bf16x8 af[2][4], bf[2][2];
#define GRL_RD(set, q)                                                                               \
            do {                                                                                     \
                const unsigned ch_ = ((2 * (q) + fhalf) ^ fsw) << 4;                                 \
                const unsigned aa_ = a_lds + ch_, bb_ = b_lds + ch_;                                 \
                asm volatile("ds_read_b128 %0, %1" : "=v"(af[set][0]) : "v"(aa_));                   \
                asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(af[set][1]) : "v"(aa_));       \
                asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(af[set][2]) : "v"(aa_));       \
                asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(af[set][3]) : "v"(aa_));      \
                asm volatile("ds_read_b128 %0, %1" : "=v"(bf[set][0]) : "v"(bb_));                   \
                asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(bf[set][1]) : "v"(bb_));       \
            } while (0)
#define GRL_MM(set)                                                                                  \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                        \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[set][i], bf[set][j], acc[i][j], 0, 0, 0)
#define GRL_WAIT(n)                                                                                  \
            asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory");                                  \
            __builtin_amdgcn_sched_barrier(0)
            // dense A: the pieces ride between the groups (+7 % over all eight at the top); the conv
            // gather's per-piece address arithmetic costs more there than it hides (-5 %): at the top
            constexpr bool SPREAD = !CONV;
            if (more && SPREAD) stage_tap(kt + 1);
            GRL_RD(0, 0);
            GRL_RD(1, 1);
            if (more && SPREAD) { stage_piece(cur ^ 1, kt + 1, 0); stage_piece(cur ^ 1, kt + 1, 1); }
            GRL_WAIT(6);
            GRL_MM(0);
            __builtin_amdgcn_sched_barrier(0);
            if (more && SPREAD) stage_piece(cur ^ 1, kt + 1, 2);
            GRL_RD(0, 2);
            GRL_WAIT(6);
            GRL_MM(1);
            __builtin_amdgcn_sched_barrier(0);
            if (more && SPREAD) stage_piece(cur ^ 1, kt + 1, 3);
            GRL_RD(1, 3);
            GRL_WAIT(6);
            GRL_MM(0);
            __builtin_amdgcn_sched_barrier(0);
            GRL_WAIT(0);
            GRL_MM(1);
#undef GRL_RD
#undef GRL_MM
#undef GRL_WAIT
            __syncthreads();                                // (vmcnt(0): stage kt+1 has landed)
        };
        if constexpr (CONV) {
            for (int kt = 0; kt < nk; ++kt) {
                if (kt + 1 < nk) stage((kt & 1) ^ 1, kt + 1);      // all eight pieces, then the stage body
                do_stage(std::false_type{}, kt);
            }
        } else {
            for (int kt = 0; kt + 1 < nk; ++kt) do_stage(std::true_type{}, kt);
            do_stage(std::false_type{}, nk - 1);
        }

        // the K loop ended on a barrier: both stage buffers are free.  Request the NEXT tile's first
        // stage now, so that it lands under this tile's epilogue.
        const int cm0 = m0 + wr * 128, cn = n0 + wc * 64 + lcol, n0c = n0;
        const int stat_row = 2 * cur_tile_m + wr;
        const int next_t = t + (int)gridDim.x;
        if (next_t < num_tiles) {
            setup_tile(next_t);
            stage(0, 0);
        }

        // ---- epilogue: acc[i][j][r] is Y[row][col], row = (r&3) + 8*(r>>2) + 4*fhalf, col = lane&31 ----
        // Two copies (round 5).  INTERIOR (the wave's 128 x 64 block lies inside the matrix) has no branch between its
        // first residual request and its last store, so hipcc counts its loads and stores; with the per-row `m < M` and
        // null-pointer tests in the way it emitted `s_waitcnt vmcnt(0)` in front of every load and store of the epilogue.
        auto epilogue = [&](auto interior_) {
            constexpr bool INT = decltype(interior_)::value;
            const bool n_ok = INT || cn < p.N;
            f32x4 sc[2] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}}, sh[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (n_ok) {
                if (p.scale) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) sc[u] = *reinterpret_cast<const f32x4*>(p.scale + cn + 4 * u);
                }
                if (p.shift) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) sh[u] = *reinterpret_cast<const f32x4*>(p.shift + cn + 4 * u);
                }
            }
            float relu_floor;      // 0 (ReLU) or a quiet NaN (no ReLU: v_max returns the other operand, a NaN accumulator stays NaN);
            {                       // through an asm move: told the constant, hipcc folds max(t, NaN) into a select per element
                const uint32_t floor_bits = p.relu ? 0u : 0x7fc00000u;
                asm("v_mov_b32 %0, %1" : "=v"(relu_floor) : "s"(floor_bits));
            }
            __bf16* yrow = SQD ? nullptr : y16 + (int64_t)(cm0 + lrow) * p.ldy + cn;
            const int64_t ystep = (int64_t)8 * p.ldy;
            f32x4 ssum[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, ssq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            // (BNZ: one block of rows in flight instead of two -- the z rows and the per-channel vectors need the registers)
            constexpr int NB = BNZ ? 1 : 2;
            bf16x8 res8[NB][4], z8[BNZ ? 1 : 1][BNZ ? 4 : 1];
            uint32_t bb[BNZ == 1 ? 4 : 1];
            const __bf16* const z16 = reinterpret_cast<const __bf16*>(p.bn_z);
            f32x4 bmu[2], bis[2], bms[2], bmb[2];
            if constexpr (BNZ != 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    bmu[u] = *reinterpret_cast<const f32x4*>(p.bn_mean + cn + 4 * u);
                    bis[u] = *reinterpret_cast<const f32x4*>(p.bn_invstd + cn + 4 * u);
                    if constexpr (BNZ == 2) {
                        bms[u] = *reinterpret_cast<const f32x4*>(p.bn_mscale + cn + 4 * u);
                        bmb[u] = p.bn_mbeta ? *reinterpret_cast<const f32x4*>(p.bn_mbeta + cn + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            // (row pointers are CARRIED -- + 8 rows per step -- instead of recomputed: a 64-bit multiply-add and two shift-adds
            //  per load and store were a tenth of the epilogue's instructions)
            const __bf16* rrow = RES && !SQD ? r16 + (int64_t)(cm0 + lrow) * p.ldres + cn : nullptr;
            const int64_t rstep = (int64_t)8 * p.ldres;
            auto res_request = [&](int i) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int m = cm0 + i * 32 + it * 8 + lrow;
                    if constexpr (RES) {
                        if constexpr (SQD) {
                            const int64_t rr = (int64_t)(m / p.res_rows) * p.res_gstride + (m % p.res_rows);
                            if (INT || (m < p.M && n_ok)) res8[i % NB][it] = *reinterpret_cast<const bf16x8*>(r16 + rr * p.ldres + cn);
                        } else {
                            if (INT || (m < p.M && n_ok)) res8[i % NB][it] = *reinterpret_cast<const bf16x8*>(rrow);
                            rrow += rstep;
                        }
                    }
                    if constexpr (BNZ != 0) {
                        z8[0][it] = *reinterpret_cast<const bf16x8*>(z16 + (int64_t)m * p.N + cn);
                        if constexpr (BNZ == 1) bb[it] = p.bn_bits[((int64_t)m * p.N + cn) >> 3];
                    }
                }
            };
            if constexpr (RES && NB == 2) res_request(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // two 32-row blocks of residual rows in flight (one dependent load per row would serialise the
                // HBM-bound epilogue of a short-K layer)
                if constexpr (NB == 2) { if constexpr (RES) { if (i + 1 < 4) res_request(i + 1); } }
                else if constexpr (RES || BNZ != 0) res_request(i);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                        Cs[row * 64 + ((j * 32 + frow) ^ (((row >> 1) & 1) << 2))] = acc[i][j][r];
                    }
                // (the same wave wrote and reads the slab: a wave's LDS operations complete in order)
                if constexpr (SQD) {
                    f32x4 part[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = it * 8 + lrow;
                        if (n_ok) {
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * 64 + ((lcol + 4 * u) ^ (((row >> 1) & 1) << 2)));
                                v = v * sc[u] + sh[u];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float f1v = (float)(__bf16)(v[e] > 0.f ? v[e] : 0.f);
                                    const float dd = f1v - (float)res8[i % NB][it][4 * u + e];
                                    part[u][e] += dd * dd;
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int e = 0; e < 4; ++e) part[u][e] += __shfl_xor(part[u][e], o);
                    if (lrow == 0 && n_ok) {
                        float* const yq = p.y + (int64_t)((cm0 + i * 32) >> 5) * p.ldy + cn;
                        *reinterpret_cast<f32x4*>(yq) = part[0];
                        *reinterpret_cast<f32x4*>(yq + 4) = part[1];
                    }
                    continue;
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + lrow;
                    const int m = cm0 + i * 32 + row;
                    if (INT || (m < p.M && n_ok)) {
                        f32x4 ov[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * 64 + ((lcol + 4 * u) ^ (((row >> 1) & 1) << 2)));
                            if constexpr (GBIAS)
                                v += *reinterpret_cast<const f32x4*>(p.gbias + (int64_t)(m / p.rows_per_group) * p.N + cn + 4 * u);
                            if constexpr (STATS && BNZ == 0) { ssum[u] += v; ssq[u] += v * v; }
                            v = v * sc[u] + sh[u];
                            f32x4 tv;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float tt = v[e];
                                if constexpr (RES) tt = tt + (float)res8[i % NB][it][4 * u + e];
                                else tt = tt + 0.f;
                                // ReLU as ONE v_max against 0 / NaN (v_max returns the other operand for a quiet NaN: no ReLU = identity,
                                // and a NaN accumulator stays NaN as in the edge tiles' select; was v_cmp + v_cndmask + s_or per element; v_max_f32
                                // orders -0 < +0, so max(t, +0) == (t > 0 ? t : 0) bit for bit)
                                tv[e] = __builtin_fmaxf(tt, relu_floor);
                            }
                            if constexpr (BNZ != 0) {
                                f32x4 zc;
#pragma unroll
                                for (int e = 0; e < 4; ++e) zc[e] = (float)z8[0][it][4 * u + e] - bmu[u][e];
                                if constexpr (BNZ == 1) {
                                    const uint32_t mk = bb[it] >> (4 * u);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) tv[e] = ((mk >> e) & 1u) ? tv[e] : 0.f;
                                } else if constexpr (BNZ == 2) {
                                    const f32x4 tm = zc * bms[u] + bmb[u];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) tv[e] = tm[e] > 0.f ? tv[e] : 0.f;
                                }
                                ssum[u] += tv; ssq[u] += tv * (zc * bis[u]);
                            }
                            ov[u] = tv;
                        }
                        // (one v_cvt_pk_bf16_f32 per pair: element-wise casts gave a convert and a v_perm per value)
                        const f32x8 o32 = {ov[0][0], ov[0][1], ov[0][2], ov[0][3], ov[1][0], ov[1][1], ov[1][2], ov[1][3]};
                        *reinterpret_cast<bf16x8*>(yrow) = __builtin_convertvector(o32, bf16x8);
                    }
                    yrow += ystep;
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
        const bool interior = cm0 + 128 <= p.M && n0c + wc * 64 + 64 <= p.N;       // (wave-uniform)
        if (interior) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        if (next_t >= num_tiles) break;
        t = next_t;
    }
}

bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }

// -1 auto (default; GRL_GEMM_BF16_256 overrides it at load time), 0 never, 1 whenever legal
int g_mode = [] {
    const char* e = getenv("GRL_GEMM_BF16_256");
    return e ? atoi(e) : -1;
}();

}  // namespace

extern "C" int grl_gemm_bf16_tile_mode(int mode) {
    const int old = g_mode;
    if (mode >= -1 && mode <= 1) g_mode = mode;
    return old;
}

// Returns 1 when the 256 x 256 kernel took the launch, 0 when the caller should use the
// 128 x 128 family (shape / feature not covered), < 0 on error.
// Would grl_gemm_bf16_256 take this launch?  (grl_conv_gemm_f32_stat_rows asks too: the statistics slab then has
// two rows per 256-row tile.)
// GRL_EPI_SQDIFF on bf16 storage: only this kernel has it (whatever the tile count)
static bool sqdiff_ok(const GrlGemm& d) {
    return d.math == GRL_MATH_BF16S && d.epilogue == GRL_EPI_SQDIFF && !d.conv && !d.stats && !d.gbias && !d.rowscale && d.res &&
           d.res_rows > 0 && d.res_rows % 32 == 0 && d.M % TB == 0 && d.N % TB == 0 && d.K % 64 == 0 && d.lda % 8 == 0 &&
           d.ldw % 8 == 0 && d.ldres % 8 == 0 && d.ldy % 4 == 0 && al16(d.a) && al16(d.w) && al16(d.y) && al16(d.res) &&
           al16(d.scale) && al16(d.shift) && (int64_t)d.M * d.lda * 2 < (1ll << 32) && (int64_t)d.N * d.ldw * 2 < (1ll << 32);
}

bool grl_gemm_bf16_256_takes(const GrlGemm& d) {
    const int mode = g_mode;
    // (the BatchNorm-backward reduce lives in the INTERIOR epilogue: every tile must be one)
    if (d.bn_z && (!d.stats || d.gbias || d.M % TB || d.N % TB || !d.bn_mean || !d.bn_invstd)) return false;
    if (sqdiff_ok(d)) return true;
    if (mode == 0) return false;
    if (d.math != GRL_MATH_BF16S || d.epilogue != GRL_EPI_AFFINE || d.rowscale || d.out_f32) return false;
    if (d.K % 64 || d.N % 8 || d.ldy % 8 || (d.res && d.ldres % 8) || d.lda % 8 || d.ldw % 8) return false;
    if (d.conv && d.C % 64) return false;
    if (!al16(d.a) || !al16(d.w) || !al16(d.y) || !al16(d.res) || !al16(d.scale) || !al16(d.shift) || !al16(d.gbias))
        return false;
    const int64_t a_bytes = d.conv ? (int64_t)(d.M / (d.Ho * d.Wo)) * d.H * d.W * d.C * 2 : (int64_t)d.M * d.lda * 2;
    if (a_bytes >= (1ll << 32) || (int64_t)d.N * d.ldw * 2 >= (1ll << 32) || (d.conv && (d.H + 2 > 65535 || d.W + 2 > 65535))) return false;
    const int64_t num_tiles = (int64_t)((d.M + TB - 1) / TB) * ((d.N + TB - 1) / TB);
    if (mode < 0 && (num_tiles < 192 || d.N < 256)) return false;
    return true;
}

int grl_gemm_bf16_256_stat_rows(const GrlGemm& d) { return 2 * ((d.M + TB - 1) / TB); }

namespace {
template <bool CONV, bool STATS, bool SQD, bool RES, bool GBIAS, int BNZ = 0>
void launch_256(const GrlGemm& d, hipStream_t s, unsigned grid, int tiles_n, int num_tiles) {
    auto kern = gemm_bf16_256_kernel<CONV, STATS, SQD, RES, GBIAS, BNZ>;
    static const bool attr = [&] {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        return true;
    }();
    (void)attr;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 2 * STAGE, s, d, tiles_n, num_tiles);
}
template <bool CONV, int BNZ>
void launch_256_bnz(const GrlGemm& d, hipStream_t s, unsigned grid, int tiles_n, int num_tiles) {
    if (d.res) launch_256<CONV, true, false, true, false, BNZ>(d, s, grid, tiles_n, num_tiles);
    else launch_256<CONV, true, false, false, false, BNZ>(d, s, grid, tiles_n, num_tiles);
}
template <bool CONV, bool STATS>
void launch_256_epi(const GrlGemm& d, hipStream_t s, unsigned grid, int tiles_n, int num_tiles) {
    if constexpr (STATS) {
        if (d.bn_z) {                       // (takes() has checked: every tile interior, no per-clip bias)
            if (d.bn_bits) launch_256_bnz<CONV, 1>(d, s, grid, tiles_n, num_tiles);
            else if (d.bn_mscale) launch_256_bnz<CONV, 2>(d, s, grid, tiles_n, num_tiles);
            else launch_256_bnz<CONV, 3>(d, s, grid, tiles_n, num_tiles);
            return;
        }
    }
    if (d.res) {
        if (d.gbias) launch_256<CONV, STATS, false, true, true>(d, s, grid, tiles_n, num_tiles);
        else launch_256<CONV, STATS, false, true, false>(d, s, grid, tiles_n, num_tiles);
    } else {
        if (d.gbias) launch_256<CONV, STATS, false, false, true>(d, s, grid, tiles_n, num_tiles);
        else launch_256<CONV, STATS, false, false, false>(d, s, grid, tiles_n, num_tiles);
    }
}
int cus_of_device() {                  // persistent grid: one 8-wave workgroup per CU, a multiple of 8
    static const int cus = [] {
        int dev = 0, n = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n = prop.multiProcessorCount;
        return n / 8 * 8 > 0 ? n / 8 * 8 : 8;
    }();
    return cus;
}
}  // namespace

int grl_gemm_bf16_256(const GrlGemm& d, hipStream_t s) {
    const int mode = g_mode;
    const int cus = cus_of_device();
    if (d.bn_z && (!d.stats || d.gbias || d.M % TB || d.N % TB || !d.bn_mean || !d.bn_invstd)) return 0;
    if (sqdiff_ok(d)) {
        const int tiles_n = d.N / TB;
        const int64_t num_tiles = (int64_t)(d.M / TB) * tiles_n;
        launch_256<false, false, true, true, false>(d, s, (unsigned)(num_tiles < cus ? num_tiles : cus), tiles_n, (int)num_tiles);
        const int e = grl_check_launch("grl_conv_gemm_f32 (bf16 256x256, SQDIFF)");
        return e ? e : 1;
    }
    if (mode == 0) return 0;
    if (d.math != GRL_MATH_BF16S || d.epilogue != GRL_EPI_AFFINE || d.rowscale || d.out_f32) return 0;
    if (d.K % 64 || d.N % 8 || d.ldy % 8 || (d.res && d.ldres % 8) || d.lda % 8 || d.ldw % 8) return 0;
    if (d.conv && d.C % 64) return 0;
    if (!al16(d.a) || !al16(d.w) || !al16(d.y) || !al16(d.res) || !al16(d.scale) || !al16(d.shift) || !al16(d.gbias))
        return 0;
    // 32-bit operand offsets inside the kernel
    const int64_t a_bytes = d.conv ? (int64_t)(d.M / (d.Ho * d.Wo)) * d.H * d.W * d.C * 2 : (int64_t)d.M * d.lda * 2;
    if (a_bytes >= (1ll << 32) || (int64_t)d.N * d.ldw * 2 >= (1ll << 32) || (d.conv && (d.H + 2 > 65535 || d.W + 2 > 65535))) return 0;
    const int tiles_m = (d.M + TB - 1) / TB, tiles_n = (d.N + TB - 1) / TB;
    const int64_t num_tiles = (int64_t)tiles_m * tiles_n;
    // one workgroup per CU: fewer than ~3/4 of a wave of tiles leaves the chip idle and the
    // 128 x 128 family (two workgroups per CU, 4x the tiles) does better
    if (mode < 0 && (num_tiles < 192 || d.N < 256)) return 0;      // (K = 64 included: full 512-byte output rows, 319 -> 283 us on 1048576 x 256 x 64)
    const unsigned grid = (unsigned)(num_tiles < cus ? num_tiles : cus);
    if (d.stats) {
        if (!al16(d.stats)) return 0;
        if (d.conv) launch_256_epi<true, true>(d, s, grid, tiles_n, (int)num_tiles);
        else launch_256_epi<false, true>(d, s, grid, tiles_n, (int)num_tiles);
    } else if (d.conv)
        launch_256_epi<true, false>(d, s, grid, tiles_n, (int)num_tiles);
    else
        launch_256_epi<false, false>(d, s, grid, tiles_n, (int)num_tiles);
    const int e = grl_check_launch("grl_conv_gemm_f32 (bf16 256x256)");
    return e ? e : 1;
}
