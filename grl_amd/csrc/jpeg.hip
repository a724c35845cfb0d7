// Baseline-JPEG frame decode on gfx950 (include/grl_hip.h, "Frame decode on the device"): replaces the per-frame
// `Image.open(img_path).convert('RGB')` of /root/reference/reid/data/video_loader.py:124-141, bit-identical to Pillow /
// libjpeg-turbo with libjpeg's defaults (JDCT_ISLOW, fancy upsampling).  Integer / byte work throughout -- no MFMA here:
//   0. jpeg_unstuff_kernel   0xFF00 stuffing, fill bytes and the trailing marker removed (a workgroup per frame, 4 bytes per
//                            lane, workgroup scan); jpeg_lut_kernel: look-ahead tables per Huffman table set.
//   1. jpeg_entropy_par_kernel  Huffman decoding (ITU-T T.81 F.2.2) by ONE 256-LANE WORKGROUP PER FRAME: the clean stream is
//                            cut into 512-bit subsequences, every lane decodes one from a guessed state, lanes re-walk until
//                            each starts where its left neighbour stopped (Huffman streams re-synchronise: 7-12 rounds for a
//                            quality-90 MARS frame), a scan of the block counts places every lane, a last walk writes the
//                            coefficients, a prefix sum per component turns DC differences into DC values (jpeg_par.h).
//                            Stream, look-ahead tables and DC values sit in LDS.  128 frames: 0.79 ms (the one-lane-per-frame
//                            form below: 10.3 ms for ANY batch size -- it stays for frames that do not fit the workgroup form
//                            and for scans with restart intervals; jpeg_core.h).
//   1'. jpeg_entropy_kernel  the same with a lane per frame (64 frames per workgroup): clean reader with the next dword loaded
//                            ahead, or the general reader (stuffing / RSTn handled while decoding); a block is assembled in LDS
//                            (dword-interleaved over the lanes) and leaves as eight 16-byte stores.
//   2. jpeg_idct_kernel      dequantisation + jidctint.c's jpeg_idct_islow (CONST_BITS 13, PASS1_BITS 2), one lane per 8 x 8
//                            block, both passes in registers, eight 8-byte row stores into the component plane.
//   3. jpeg_color_kernel     jdsample.c's triangle-filter ("fancy") chroma upsampling + jdcolor.c's fixed-point YCbCr -> RGB,
//                            one lane per output pixel, planar uint8 out ([n][3][H][W]: the clip tensor's layout).
// All frames of a batch share one geometry (MARS: 256 x 128, 4:2:0).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#include "../../include/grl_hip.h"
#include "common.h"
#define GRL_HD __host__ __device__
#include "jpeg_core.h"
#include "jpeg_par.h"

namespace {

__device__ __constant__ uint8_t kNaturalDev[80] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};
const uint8_t kNaturalHost[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// geometry shared by every frame of a batch (from frame 0, checked on the host for the others)
struct Geo {
    int width, height, ncomp, hmax, vmax;
    int mcux, mcuy, bpm, blocks;          // MCUs per row / column, blocks per MCU, blocks per frame
    int hs[3], vs[3];
    int pw[3], ph[3];                     // plane sizes (whole blocks)
    int cw[3], ch[3];                     // real samples per component (downsampled_width / height)
    int poff[3];                          // byte offset of each component plane inside a frame's plane block
    int plane_bytes;                      // planes of one frame
};

Geo make_geo(const GrlJpegFrame& f) {
    Geo g;
    memset(&g, 0, sizeof(g));
    g.width = f.width; g.height = f.height; g.ncomp = f.ncomp; g.hmax = f.hmax; g.vmax = f.vmax;
    g.mcux = (f.width + 8 * f.hmax - 1) / (8 * f.hmax);
    g.mcuy = (f.height + 8 * f.vmax - 1) / (8 * f.vmax);
    int off = 0;
    for (int c = 0; c < f.ncomp; ++c) {
        g.hs[c] = f.hs[c]; g.vs[c] = f.vs[c];
        g.bpm += f.hs[c] * f.vs[c];
        g.pw[c] = g.mcux * f.hs[c] * 8;
        g.ph[c] = g.mcuy * f.vs[c] * 8;
        g.cw[c] = (f.width * f.hs[c] + f.hmax - 1) / f.hmax;
        g.ch[c] = (f.height * f.vs[c] + f.vmax - 1) / f.vmax;
        g.poff[c] = off;
        off += g.pw[c] * g.ph[c];
    }
    g.blocks = g.mcux * g.mcuy * g.bpm;
    g.plane_bytes = (off + 15) & ~15;
    return g;
}

// ---- 1. entropy decoding (the per-lane logic lives in jpeg_core.h: it also compiles as host C++ for the CPU tests) -------
constexpr int EW = 64;                                   // frames per workgroup (one wave)
constexpr int MAX_LDS_SETS = 4;                          // Huffman table sets whose look-ahead tables fit LDS (4 x 18 KiB; ONE set: 26 KiB with the stage)
constexpr int LUT_PER_SET = GJ_LUT_PER_SET;              // uint16 entries: [DC0 512][DC1 512][AC0 4096][AC1 4096] = 18 KiB

struct Reps { int frame[MAX_LDS_SETS]; };               // the frame whose tables define table set u

// look-ahead tables of the batch's table sets: grid (sets, LUT_PER_SET / 256), 256 threads
__global__ __launch_bounds__(256) void jpeg_lut_kernel(const GrlJpegFrame* __restrict__ frames, Reps reps, int identity,
                                                               uint16_t* __restrict__ lut) {
    const int u = blockIdx.x;
    const GrlJpegFrame* fr = frames + (identity ? u : reps.frame[u]);
    const int e = blockIdx.y * 256 + threadIdx.x;                       // entry inside the set
    const int t = e < (2 << GJ_DC_BITS) ? e >> GJ_DC_BITS : 2 + ((e - (2 << GJ_DC_BITS)) >> GJ_AC_BITS);
    lut[(int64_t)u * LUT_PER_SET + e] = gj_lut_entry(fr, t, e - gj_lut_offset(t));
}

// Unstuffing pre-pass: one workgroup per frame walks the scan in 1 KiB chunks; a lane owns four bytes, applies the
// per-byte rule of jpeg_core.h (drop the 0x00 behind an 0xFF, drop fill bytes, stop at the first marker), an exclusive
// scan of the kept-byte counts over the workgroup gives every lane its output offset.  The clean stream of a frame lives at
// the dword-rounded offset of its scan (scans never touch: there are headers in between) and is zero-padded by 16 bytes.
constexpr int UT = 256;                                  // lanes of the pre-pass workgroup (4 bytes each per chunk)
__global__ __launch_bounds__(UT) void jpeg_unstuff_kernel(const uint8_t* __restrict__ bytes, const GrlJpegFrame* __restrict__ frames,
                                                          uint8_t* __restrict__ clean, uint32_t* __restrict__ clean_len) {
    __shared__ uint32_t s_wave[UT / 64];
    __shared__ uint32_t s_marker;
    const int f = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint8_t* src = bytes + frames[f].scan_off;
    const uint32_t len = frames[f].scan_len;
    uint8_t* dst = clean + ((frames[f].scan_off + 3u) & ~3u);
    uint32_t base = 0;
    for (uint32_t chunk = 0; chunk < len; chunk += UT * 4) {
        if (t == 0) s_marker = 0xffffffffu;
        __syncthreads();
        const uint32_t i0 = chunk + 4u * t;
        int by[6];                                                         // bytes i0 - 1 .. i0 + 4 (-1 beyond the scan)
        for (int k = 0; k < 6; ++k) {
            const uint32_t i = i0 + k - 1;
            by[k] = (i0 + k >= 1 && i < len) ? (int)src[i] : (i0 + k == 0 ? 0 : -1);
        }
        uint32_t first_marker = 0xffffffffu;
        for (int k = 3; k >= 0; --k)
            if (i0 + k < len && gj_marker_starts(by[k + 1], by[k + 2])) first_marker = i0 + k;
        if (first_marker != 0xffffffffu) atomicMin(&s_marker, first_marker);
        __syncthreads();
        const uint32_t marker = s_marker;
        uint32_t keep = 0, cnt = 0;
        for (int k = 0; k < 4; ++k)
            if (i0 + k < len && i0 + k < marker && gj_is_data(by[k], by[k + 1], by[k + 2])) { keep |= 1u << k; ++cnt; }
        // exclusive scan of cnt over the workgroup: wave-level shuffles, then the wave totals
        uint32_t incl = cnt;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t off = base + incl - cnt, total = 0;
        for (int w = 0; w < UT / 64; ++w) {
            if (w < wave) off += s_wave[w];
            total += s_wave[w];
        }
        for (int k = 0; k < 4; ++k)
            if (keep & (1u << k)) dst[off++] = (uint8_t)by[k + 1];
        base += total;
        __syncthreads();
        if (marker != 0xffffffffu) break;                                  // (uniform: every lane read the same s_marker)
    }
    if (t < 16) dst[base + t] = 0;                                         // zero padding: the reader's last dword and one more
    if (t == 0) clean_len[f] = base;
}

// READER 0: the clean stream of the pre-pass (no restart intervals in the batch); READER 1: the general reader on the raw bytes
template <int READER>
__global__ __launch_bounds__(EW) void jpeg_entropy_kernel(const uint8_t* __restrict__ bytes, uint32_t nbytes,
                                                          const GrlJpegFrame* __restrict__ frames, int n,
                                                          int16_t* __restrict__ coef, GjScanGeo sg, int blocks,
                                                          const uint16_t* __restrict__ lut, int lds_sets,
                                                          const uint8_t* __restrict__ clean, const uint32_t* __restrict__ clean_len) {
    extern __shared__ __align__(16) uint8_t lds[];
    uint8_t* const s_nat = lds;                                           // 80 bytes (+ pad to 128)
    int16_t* const s_stage = reinterpret_cast<int16_t*>(lds + 128);       // 8 KiB: one 8 x 8 block per lane, [dword][lane]
    uint16_t* const s_lut = reinterpret_cast<uint16_t*>(lds + 128 + 8192);    // lds_sets x 18 KiB
    const int lane = threadIdx.x;
    for (int i = lane; i < 80; i += EW) s_nat[i] = kNaturalDev[i];
    {   // the batch's look-ahead tables -> LDS (when they fit: a batch of camera frames shares ONE table set)
        const uint4* src = reinterpret_cast<const uint4*>(lut);
        uint4* dst = reinterpret_cast<uint4*>(s_lut);
        for (int i = lane; i < lds_sets * LUT_PER_SET / 8; i += EW) dst[i] = src[i];
    }
    __syncthreads();
    const int f = blockIdx.x * EW + lane;
    if (f >= n) return;
    const GrlJpegFrame* fr = frames + f;
    int16_t* const out = coef + (int64_t)f * blocks * 64;
    // (two calls per reader, not one table pointer chosen at run time: a pointer that may be LDS or global compiles to FLAT
    //  loads, each followed by s_waitcnt vmcnt(0) & lgkmcnt(0); with the provenance known per call they are ds_read / global_load)
    if constexpr (READER == 0) {
        GjClean b;
        gj_clean_init(b, clean + ((fr->scan_off + 3u) & ~3u), clean_len[f]);
        if (lds_sets) gj_decode_scan(b, fr, s_lut + (int)fr->tabset * LUT_PER_SET, s_nat, out, sg, s_stage + 2 * lane, 2 * EW);
        else gj_decode_scan(b, fr, lut + (int64_t)fr->tabset * LUT_PER_SET, s_nat, out, sg, s_stage + 2 * lane, 2 * EW);
    } else {
        GjBits b;
        gj_bits_init(b, bytes, nbytes & ~3u, fr);
        if (lds_sets) gj_decode_scan(b, fr, s_lut + (int)fr->tabset * LUT_PER_SET, s_nat, out, sg, s_stage + 2 * lane, 2 * EW);
        else gj_decode_scan(b, fr, lut + (int64_t)fr->tabset * LUT_PER_SET, s_nat, out, sg, s_stage + 2 * lane, 2 * EW);
    }
}

// ---- 1b. the same, with intra-frame parallelism: one 256-lane workgroup per frame, self-synchronising subsequences
//          (jpeg_par.h has the algorithm and the per-lane logic; this is its workgroup form) ------------------------------------
constexpr int PT = 256;                                  // lanes per frame
constexpr int PAR_MAX_BLOCKS = 8192;                     // blocks per frame the DC staging array holds
constexpr int PAR_MAX_SCAN = 32768 - 64;                 // bytes of one scan that fit the LDS stream buffer

struct ParLayout { int lut, exits, ints, part, dc, be, total; };      // byte offsets into the dynamic LDS
inline ParLayout par_layout(int blocks, int max_dw) {
    ParLayout l;
    int o = 128;                                         // nat table
    l.lut = o; o += GJ_LUT_PER_SET * 2;
    l.exits = o; o += PT * 16;                           // GjState + block count per lane
    l.ints = o; o += PT * 4;                             // scan scratch
    l.part = o; o += PT * 3 * 4;                         // DC partial sums per lane and component
    l.dc = o; o += ((blocks * 2 + 15) & ~15);
    l.be = o; o += (max_dw + 2) * 4;
    l.total = (o + 15) & ~15;
    return l;
}

__global__ __launch_bounds__(PT) void jpeg_entropy_par_kernel(const GrlJpegFrame* __restrict__ frames, int16_t* __restrict__ coef,
                                                              int blocks, int mcus, const uint16_t* __restrict__ lut_g,
                                                              const uint8_t* __restrict__ clean, const uint32_t* __restrict__ clean_len,
                                                              ParLayout lay, uint32_t min_bits) {
    extern __shared__ __align__(16) uint8_t lds[];
    uint8_t* const s_nat = lds;
    uint16_t* const s_lut = reinterpret_cast<uint16_t*>(lds + lay.lut);
    int32_t* const s_exit = reinterpret_cast<int32_t*>(lds + lay.exits);     // [lane][4]: bit, z, k, blocks
    int32_t* const s_int = reinterpret_cast<int32_t*>(lds + lay.ints);
    int32_t* const s_part = reinterpret_cast<int32_t*>(lds + lay.part);
    int16_t* const s_dc = reinterpret_cast<int16_t*>(lds + lay.dc);
    uint32_t* const s_be = reinterpret_cast<uint32_t*>(lds + lay.be);
    const int f = blockIdx.x, i = threadIdx.x;
    const GrlJpegFrame* fr = frames + f;
    const uint32_t nbytes = clean_len[f];
    const uint32_t ndw = (nbytes + 3u) >> 2;
    {   // stage: zigzag table, this frame's look-ahead tables, the clean stream byte-swapped to bit order
        for (int j = i; j < 80; j += PT) s_nat[j] = kNaturalDev[j];
        const uint4* src = reinterpret_cast<const uint4*>(lut_g + (int64_t)fr->tabset * LUT_PER_SET);
        uint4* dst = reinterpret_cast<uint4*>(s_lut);
        for (int j = i; j < LUT_PER_SET / 8; j += PT) dst[j] = src[j];
        const uint32_t* cs = reinterpret_cast<const uint32_t*>(clean + ((fr->scan_off + 3u) & ~3u));
        for (uint32_t j = i; j < ndw + 2; j += PT) s_be[j] = j < ndw ? gj_bswap(cs[j]) : 0u;
    }
    __syncthreads();
    GjParTables T;
    gj_par_tables(T, fr, s_lut, s_nat);
    const uint32_t nbits = nbytes * 8u;
    const uint32_t L = gj_par_seq_bits(nbits, PT, min_bits);
    const int S = nbits ? (int)((nbits + L - 1) / L) : 1;
    const bool active = i < S;
    const uint32_t end = (uint32_t)(i + 1) * L;
    GjState entry = {(uint32_t)i * L, 0, 0};
    int nb = 0;
    if (active) {
        GjState st = entry;
        nb = gj_par_walk(s_be, ndw, T, st, end);
        s_exit[4 * i] = (int32_t)st.bit; s_exit[4 * i + 1] = st.z; s_exit[4 * i + 2] = st.k;
    }
    __syncthreads();
    // synchronisation rounds: a lane whose left neighbour left in another state than the one it assumed walks again
    for (int round = 0; round < PT; ++round) {
        GjState prev = entry;
        if (active && i > 0) { prev.bit = (uint32_t)s_exit[4 * (i - 1)]; prev.z = s_exit[4 * (i - 1) + 1]; prev.k = s_exit[4 * (i - 1) + 2]; }
        __syncthreads();                                  // (every lane has read its neighbour before anyone overwrites)
        int changed = 0;
        if (active && i > 0 && !gj_same(prev, entry)) {
            entry = prev;
            GjState st = entry;
            nb = gj_par_walk(s_be, ndw, T, st, end);
            s_exit[4 * i] = (int32_t)st.bit; s_exit[4 * i + 1] = st.z; s_exit[4 * i + 2] = st.k;
            changed = 1;
        }
        if (!__syncthreads_or(changed)) break;
    }
    // exclusive scan of the completed-block counts (Hillis-Steele over the workgroup)
    s_int[i] = active ? nb : 0;
    __syncthreads();
    for (int o = 1; o < PT; o <<= 1) {
        const int v = i >= o ? s_int[i - o] : 0;
        __syncthreads();
        s_int[i] += v;
        __syncthreads();
    }
    int b = s_int[i] - (active ? nb : 0);                 // blocks completed before this lane's entry
    int16_t* const out = coef + (int64_t)f * blocks * 64;
    if (active) {
        GjState st = entry;
        const bool last = i == S - 1;
        auto emit = [&](int idx, int v) {
            if (b < blocks) {
                if (idx == 0) s_dc[b] = (int16_t)v;       // the DC DIFFERENCE; summed below
                else out[(int64_t)b * 64 + idx] = (int16_t)v;
            }
        };
        GjBeReader rd;
        gj_be_init(rd, s_be, ndw, st.bit);
        while (st.bit < end || (last && b < blocks))
            if (gj_par_step(rd, T, st, emit)) ++b;
    }
    __syncthreads();
    // DC differences -> DC values: per component a prefix sum in scan order.  A lane owns a run of MCUs.
    const int mpl = (mcus + PT - 1) / PT, m0 = i * mpl, m1 = min(mcus, m0 + mpl);
    int part[3] = {0, 0, 0};
    for (int m = m0; m < m1; ++m)
        for (int z = 0; z < T.bpm; ++z) part[T.comp[z]] += s_dc[m * T.bpm + z];
    for (int c = 0; c < 3; ++c) s_part[c * PT + i] = part[c];
    __syncthreads();
    for (int o = 1; o < PT; o <<= 1) {
        int v[3];
        for (int c = 0; c < 3; ++c) v[c] = i >= o ? s_part[c * PT + i - o] : 0;
        __syncthreads();
        for (int c = 0; c < 3; ++c) s_part[c * PT + i] += v[c];
        __syncthreads();
    }
    int pred[3];
    for (int c = 0; c < 3; ++c) pred[c] = s_part[c * PT + i] - part[c];
    for (int m = m0; m < m1; ++m)
        for (int z = 0; z < T.bpm; ++z) {
            const int bb = m * T.bpm + z, c = T.comp[z];
            pred[c] += s_dc[bb];
            out[(int64_t)bb * 64] = (int16_t)pred[c];
        }
}

// ---- 2. dequantisation + jidctint.c jpeg_idct_islow ---------------------------------------------------------------------
#define GJ_CONST_BITS 13
#define GJ_PASS1_BITS 2
#define GJ_DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

__device__ __forceinline__ uint32_t range_limit(int v) {       // post-IDCT range table: index v & 1023, centred on 128
    const int i = v & 1023;
    return i < 128 ? (uint32_t)(i + 128) : (i < 512 ? 255u : (i < 896 ? 0u : (uint32_t)(i - 896)));
}

__device__ __forceinline__ void idct_1d(int32_t in0, int32_t in1, int32_t in2, int32_t in3, int32_t in4, int32_t in5, int32_t in6,
                                        int32_t in7, int32_t (&t)[8]) {
    // one column (pass 1) or row (pass 2) of jpeg_idct_islow up to the final butterflies: t[0..3] = tmp10..13 (even part),
    // t[4..7] = tmp0..3 (odd part)
    int32_t z2 = in2, z3 = in6;
    int32_t z1 = (z2 + z3) * 4433;
    const int32_t e2 = z1 + z3 * (-15137), e3 = z1 + z2 * 6270;
    const int32_t e0 = (int32_t)((uint32_t)(in0 + in4) << GJ_CONST_BITS), e1 = (int32_t)((uint32_t)(in0 - in4) << GJ_CONST_BITS);
    t[0] = e0 + e3; t[3] = e0 - e3; t[1] = e1 + e2; t[2] = e1 - e2;
    int32_t tmp0 = in7, tmp1 = in5, tmp2 = in3, tmp3 = in1;
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int32_t z4 = tmp1 + tmp3;
    const int32_t z5 = (z3 + z4) * 9633;
    tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    t[4] = tmp0 + z1 + z3; t[5] = tmp1 + z2 + z4; t[6] = tmp2 + z2 + z3; t[7] = tmp3 + z1 + z4;
}

__global__ __launch_bounds__(256) void jpeg_idct_kernel(const int16_t* __restrict__ coef, const GrlJpegFrame* __restrict__ frames,
                                                        int n, uint8_t* __restrict__ planes, Geo g) {
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (gid >= (int64_t)n * g.blocks) return;
    const int f = (int)(gid / g.blocks), blk = (int)(gid % g.blocks);
    const int mcu = blk / g.bpm;
    int within = blk % g.bpm, c = 0;
    while (within >= g.hs[c] * g.vs[c]) { within -= g.hs[c] * g.vs[c]; ++c; }
    const int by = within / g.hs[c], bx = within % g.hs[c];
    const int mx = mcu % g.mcux, my = mcu / g.mcux;
    const uint16_t* q = frames[f].q[frames[f].tq[c] & 3];
    const int16_t* in = coef + gid * 64;
    int32_t v[64];
#pragma unroll
    for (int i = 0; i < 8; ++i) {                 // 8 x 16-byte loads, dequantised on the way in
        const uint4 cw = *reinterpret_cast<const uint4*>(in + 8 * i);
        const uint4 qw = *reinterpret_cast<const uint4*>(q + 8 * i);
        const uint32_t cc[4] = {cw.x, cw.y, cw.z, cw.w}, qq[4] = {qw.x, qw.y, qw.z, qw.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[8 * i + 2 * e] = (int32_t)(int16_t)(cc[e] & 0xffffu) * (int32_t)(qq[e] & 0xffffu);
            v[8 * i + 2 * e + 1] = (int32_t)(int16_t)(cc[e] >> 16) * (int32_t)(qq[e] >> 16);
        }
    }
    int32_t t[8];
#pragma unroll
    for (int col = 0; col < 8; ++col) {           // pass 1: columns
        idct_1d(v[col], v[8 + col], v[16 + col], v[24 + col], v[32 + col], v[40 + col], v[48 + col], v[56 + col], t);
        const int sh = GJ_CONST_BITS - GJ_PASS1_BITS;
        v[col] = GJ_DESCALE(t[0] + t[7], sh);      v[56 + col] = GJ_DESCALE(t[0] - t[7], sh);
        v[8 + col] = GJ_DESCALE(t[1] + t[6], sh);  v[48 + col] = GJ_DESCALE(t[1] - t[6], sh);
        v[16 + col] = GJ_DESCALE(t[2] + t[5], sh); v[40 + col] = GJ_DESCALE(t[2] - t[5], sh);
        v[24 + col] = GJ_DESCALE(t[3] + t[4], sh); v[32 + col] = GJ_DESCALE(t[3] - t[4], sh);
    }
    uint8_t* dst = planes + (int64_t)f * g.plane_bytes + g.poff[c] + (int64_t)((my * g.vs[c] + by) * 8) * g.pw[c] + (mx * g.hs[c] + bx) * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {                 // pass 2: rows, through the range table, one 8-byte store per row
        idct_1d(v[8 * r], v[8 * r + 1], v[8 * r + 2], v[8 * r + 3], v[8 * r + 4], v[8 * r + 5], v[8 * r + 6], v[8 * r + 7], t);
        const int sh = GJ_CONST_BITS + GJ_PASS1_BITS + 3;
        const uint32_t o0 = range_limit(GJ_DESCALE(t[0] + t[7], sh)), o7 = range_limit(GJ_DESCALE(t[0] - t[7], sh));
        const uint32_t o1 = range_limit(GJ_DESCALE(t[1] + t[6], sh)), o6 = range_limit(GJ_DESCALE(t[1] - t[6], sh));
        const uint32_t o2 = range_limit(GJ_DESCALE(t[2] + t[5], sh)), o5 = range_limit(GJ_DESCALE(t[2] - t[5], sh));
        const uint32_t o3 = range_limit(GJ_DESCALE(t[3] + t[4], sh)), o4 = range_limit(GJ_DESCALE(t[3] - t[4], sh));
        uint2 w;
        w.x = o0 | (o1 << 8) | (o2 << 16) | (o3 << 24);
        w.y = o4 | (o5 << 8) | (o6 << 16) | (o7 << 24);
        *reinterpret_cast<uint2*>(dst + (int64_t)r * g.pw[c]) = w;
    }
}

// ---- 3. chroma upsampling (jdsample.c) + colour conversion (jdcolor.c) ------------------------------------------------------
__device__ __forceinline__ int up_sample(const uint8_t* __restrict__ p, int stride, int cw, int ch, int h2, int v2, int ox, int oy) {
    const bool fancy = cw > 2;                    // jdsample.c: `do_fancy && compptr->downsampled_width > 2`
    const int x = h2 ? ox >> 1 : ox;
    if (!v2) {
        const uint8_t* in = p + (int64_t)oy * stride;
        if (!h2 || !fancy) return in[x];
        if (ox & 1) return x == cw - 1 ? in[x] : (in[x] * 3 + in[x + 1] + 2) >> 2;
        return x == 0 ? in[0] : (in[x] * 3 + in[x - 1] + 1) >> 2;
    }
    const int iy = oy >> 1, v = oy & 1;
    const uint8_t* in0 = p + (int64_t)iy * stride;
    if (!fancy || !h2) return in0[x];             // (h1v2 is outside the parser's scope)
    int ny = v ? iy + 1 : iy - 1;                 // next-nearest row; beyond the image the edge row is replicated (jdmainct.c)
    ny = ny < 0 ? 0 : (ny > ch - 1 ? ch - 1 : ny);
    const uint8_t* in1 = p + (int64_t)ny * stride;
    const int thiscol = in0[x] * 3 + in1[x];
    if (ox & 1) return x == cw - 1 ? (thiscol * 4 + 7) >> 4 : (thiscol * 3 + (in0[x + 1] * 3 + in1[x + 1]) + 7) >> 4;
    return x == 0 ? (thiscol * 4 + 8) >> 4 : (thiscol * 3 + (in0[x - 1] * 3 + in1[x - 1]) + 8) >> 4;
}

__device__ __forceinline__ uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

__device__ __forceinline__ void color_px(const uint8_t* __restrict__ pl, const Geo& g, int rgb, int x, int y, uint32_t (&o)[3]) {
    const int Y = pl[g.poff[0] + (int64_t)y * g.pw[0] + x];
    if (g.ncomp == 1) { o[0] = o[1] = o[2] = (uint32_t)Y; return; }
    const int h2 = g.hmax == 2, v2 = g.vmax == 2;
    const int cbv = up_sample(pl + g.poff[1], g.pw[1], g.cw[1], g.ch[1], h2, v2, x, y);
    const int crv = up_sample(pl + g.poff[2], g.pw[2], g.cw[2], g.ch[2], h2, v2, x, y);
    if (rgb) { o[0] = (uint32_t)Y; o[1] = (uint32_t)cbv; o[2] = (uint32_t)crv; return; }
    const int cb = cbv - 128, cr = crv - 128;
    // jdcolor.c build_ycc_rgb_table: FIX(x) = (int)(x * 65536 + 0.5), ONE_HALF = 32768, arithmetic right shifts
    o[0] = clamp8(Y + ((91881 * cr + 32768) >> 16));
    o[1] = clamp8(Y + ((-22554 * cb - 46802 * cr + 32768) >> 16));
    o[2] = clamp8(Y + ((116130 * cb + 32768) >> 16));
}

// PX pixels of a row per lane: 4 (one 4-byte store per colour plane) when the width is a multiple of 4, else 1
template <int PX>
__global__ __launch_bounds__(256) void jpeg_color_kernel(const uint8_t* __restrict__ planes, const GrlJpegFrame* __restrict__ frames,
                                                         int n, uint8_t* __restrict__ out, Geo g) {
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t hw = (int64_t)g.width * g.height, groups = hw / PX;
    if (gid >= n * groups) return;
    const int f = (int)(gid / groups);
    const int64_t p0 = (gid % groups) * PX;
    const int y = (int)(p0 / g.width), x0 = (int)(p0 % g.width);
    const uint8_t* pl = planes + (int64_t)f * g.plane_bytes;
    const int rgb = frames[f].rgb;
    uint8_t* o = out + (int64_t)f * 3 * hw + p0;
    if constexpr (PX == 1) {
        uint32_t c[3];
        color_px(pl, g, rgb, x0, y, c);
        o[0] = (uint8_t)c[0]; o[hw] = (uint8_t)c[1]; o[2 * hw] = (uint8_t)c[2];
    } else {
        uint32_t w[3] = {0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < PX; ++e) {
            uint32_t c[3];
            color_px(pl, g, rgb, x0 + e, y, c);
            w[0] |= c[0] << (8 * e); w[1] |= c[1] << (8 * e); w[2] |= c[2] << (8 * e);
        }
        *reinterpret_cast<uint32_t*>(o) = w[0];
        *reinterpret_cast<uint32_t*>(o + hw) = w[1];
        *reinterpret_cast<uint32_t*>(o + 2 * hw) = w[2];
    }
}

// ---- host: header parser ---------------------------------------------------------------------------------------------------
void build_huff(const uint8_t* bits /*[17]*/, int32_t* maxcode /*[18]*/, int32_t* valoff /*[18]*/) {
    int code = 0, k = 0;
    maxcode[0] = -1; valoff[0] = 0;
    for (int l = 1; l <= 16; ++l) {
        valoff[l] = k - code;
        if (bits[l]) {
            k += bits[l];
            code += bits[l];
            maxcode[l] = code - 1;
        } else {
            maxcode[l] = -1;
        }
        code <<= 1;
    }
    maxcode[17] = 0x7fffffff; valoff[17] = 0;
}

}  // namespace

static bool g_jpeg_parallel = [] { const char* e = getenv("GRL_JPEG_PARALLEL"); return !e || atoi(e) != 0; }();

// test / A-B hook: 0 / 1 = the one-lane-per-frame entropy decoder / the workgroup-per-frame one (default), -1 = query;
// returns the previous setting.  Both produce the same coefficients.
extern "C" int grl_jpeg_parallel_mode(int on) {
    const int was = g_jpeg_parallel ? 1 : 0;
    if (on >= 0) g_jpeg_parallel = on != 0;
    return was;
}

#define GRL_REQUIRE(cond, msg) do { if (!(cond)) return grl_fail(GRL_EINVAL, msg); } while (0)

extern "C" int grl_jpeg_parse(const uint8_t* p, int64_t len, int64_t base_off, GrlJpegFrame* out) {
    GRL_REQUIRE(p && out && len >= 4 && base_off >= 0, "jpeg_parse: null / empty");
    GRL_REQUIRE(base_off + len < (1ll << 32), "jpeg_parse: the batch buffer must stay below 4 GiB (32-bit stream offsets)");
    if (p[0] != 0xFF || p[1] != 0xD8) return grl_fail(GRL_EINVAL, "jpeg_parse: no SOI marker");
    GrlJpegFrame& f = *out;
    memset(&f, 0, sizeof(f));
    bool have_sof = false, have_scan = false, qpresent[4] = {false, false, false, false}, hpresent[4] = {false, false, false, false};
    int adobe = -1;
    const size_t n = (size_t)len;
    size_t i = 2;
    while (i + 4 <= n) {
        if (p[i] != 0xFF) return grl_fail(GRL_EINVAL, "jpeg_parse: marker expected at byte %zu", i);
        while (i < n && p[i] == 0xFF) ++i;
        if (i >= n) break;
        const int m = p[i++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (i + 2 > n) break;
        const size_t seg = ((size_t)p[i] << 8) | p[i + 1];
        if (seg < 2 || i + seg > n) return grl_fail(GRL_EINVAL, "jpeg_parse: truncated segment 0x%02X", m);
        const uint8_t* s = p + i + 2;
        const size_t sl = seg - 2;
        if (m == 0xC0 || m == 0xC1) {
            if (sl < 6 || s[0] != 8) return grl_fail(GRL_EUNSUPPORTED, "jpeg_parse: %d-bit samples", sl ? (int)s[0] : 0);
            f.height = (uint16_t)((s[1] << 8) | s[2]);
            f.width = (uint16_t)((s[3] << 8) | s[4]);
            f.ncomp = s[5];
            if (f.ncomp != 1 && f.ncomp != 3) return grl_fail(GRL_EUNSUPPORTED, "jpeg_parse: %d components", (int)f.ncomp);
            if (sl < 6 + 3 * (size_t)f.ncomp || !f.width || !f.height) return grl_fail(GRL_EINVAL, "jpeg_parse: bad SOF");
            for (int c = 0; c < f.ncomp; ++c) {
                f.hs[c] = s[7 + 3 * c] >> 4;
                f.vs[c] = s[7 + 3 * c] & 15;
                f.tq[c] = s[8 + 3 * c];
                if (f.tq[c] > 3 || !f.hs[c] || !f.vs[c]) return grl_fail(GRL_EINVAL, "jpeg_parse: bad SOF component");
            }
            have_sof = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return grl_fail(GRL_EUNSUPPORTED, "jpeg_parse: SOF%d (progressive / lossless / arithmetic) is outside the device decoder's scope", m - 0xC0);
        } else if (m == 0xC4) {
            size_t o = 0;
            while (o + 17 <= sl) {
                const int tc = s[o] >> 4, th = s[o] & 15;
                if (tc > 1 || th > 3) return grl_fail(GRL_EINVAL, "jpeg_parse: bad DHT");
                if (th > 1) return grl_fail(GRL_EUNSUPPORTED, "jpeg_parse: Huffman table id %d (baseline allows 0..1)", th);
                uint8_t bits[17];
                int cnt = 0;
                bits[0] = 0;
                for (int l = 1; l <= 16; ++l) { bits[l] = s[o + l]; cnt += s[o + l]; }
                if (cnt > 256 || o + 17 + cnt > sl) return grl_fail(GRL_EINVAL, "jpeg_parse: bad DHT counts");
                const int t = tc * 2 + th;
                if (tc == 0 && cnt > 16) return grl_fail(GRL_EUNSUPPORTED, "jpeg_parse: DC table with %d symbols", cnt);
                memset(f.vals[t], 0, 256);
                memcpy(f.vals[t], s + o + 17, cnt);
                build_huff(bits, f.maxcode[t], f.valoff[t]);
                hpresent[t] = true;
                o += 17 + cnt;
            }
        } else if (m == 0xDB) {
            size_t o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                if (tq > 3 || pq > 1) return grl_fail(GRL_EINVAL, "jpeg_parse: bad DQT");
                const size_t need = pq ? 129 : 65;
                if (o + need > sl) return grl_fail(GRL_EINVAL, "jpeg_parse: truncated DQT");
                for (int k = 0; k < 64; ++k)
                    f.q[tq][kNaturalHost[k]] = pq ? (uint16_t)((s[o + 1 + 2 * k] << 8) | s[o + 2 + 2 * k]) : s[o + 1 + k];
                qpresent[tq] = true;
                o += need;
            }
        } else if (m == 0xDD) {
            if (sl < 2) return grl_fail(GRL_EINVAL, "jpeg_parse: bad DRI");
            f.restart_interval = (uint16_t)((s[0] << 8) | s[1]);
        } else if (m == 0xEE) {
            if (sl >= 12 && !memcmp(s, "Adobe", 5)) adobe = s[11];
        } else if (m == 0xDA) {
            if (!have_sof) return grl_fail(GRL_EINVAL, "jpeg_parse: SOS before SOF");
            if (sl < 1 || s[0] != f.ncomp || sl < 1 + 2 * (size_t)f.ncomp + 3)
                return grl_fail(GRL_EUNSUPPORTED, "jpeg_parse: non-interleaved scans are outside the device decoder's scope");
            for (int c = 0; c < f.ncomp; ++c) {
                f.td[c] = s[2 + 2 * c] >> 4;
                f.ta[c] = s[2 + 2 * c] & 15;
                if (f.td[c] > 1 || f.ta[c] > 1) return grl_fail(GRL_EUNSUPPORTED, "jpeg_parse: Huffman table id > 1");
            }
            f.scan_off = (uint32_t)(base_off + (int64_t)(i + seg));
            f.scan_len = (uint32_t)(n - (i + seg));
            have_scan = true;
            break;
        }
        i += seg;
    }
    if (!have_scan) return grl_fail(GRL_EINVAL, "jpeg_parse: no scan");
    f.hmax = f.vmax = 1;
    for (int c = 0; c < f.ncomp; ++c) {
        if (f.hs[c] > f.hmax) f.hmax = f.hs[c];
        if (f.vs[c] > f.vmax) f.vmax = f.vs[c];
        if (!qpresent[f.tq[c]] || !hpresent[f.td[c]] || !hpresent[2 + f.ta[c]]) return grl_fail(GRL_EINVAL, "jpeg_parse: a table the scan uses is missing");
    }
    if (f.ncomp == 3) {
        const bool chroma11 = f.hs[1] == 1 && f.vs[1] == 1 && f.hs[2] == 1 && f.vs[2] == 1;
        const bool luma_ok = (f.hs[0] == 1 && f.vs[0] == 1) || (f.hs[0] == 2 && f.vs[0] == 1) || (f.hs[0] == 2 && f.vs[0] == 2);
        if (!chroma11 || !luma_ok)
            return grl_fail(GRL_EUNSUPPORTED, "jpeg_parse: sampling %dx%d / %dx%d / %dx%d (4:4:4, 4:2:2, 4:2:0 are in scope)", f.hs[0], f.vs[0],
                            f.hs[1], f.vs[1], f.hs[2], f.vs[2]);
        if (adobe == 0) f.rgb = 1;
        else if (adobe == 2) return grl_fail(GRL_EUNSUPPORTED, "jpeg_parse: Adobe YCCK");
    } else {
        f.hs[0] = f.vs[0] = f.hmax = f.vmax = 1;      // a single-component scan is never interleaved: the factors are moot
    }
    return GRL_OK;
}

// HOST: grl_jpeg_parse for the n streams of a batch buffer (stream i = buf[offsets[i] .. offsets[i + 1])) followed by
// grl_jpeg_assign_tables -- one call per batch instead of one per frame (a Python loop over 512 frames costs ~10 ms).
// On failure *bad_index names the frame and the return value is that frame's code.
extern "C" int grl_jpeg_parse_batch(const uint8_t* buf, const int64_t* offsets, int n, GrlJpegFrame* frames, int* bad_index) {
    if (!buf || !offsets || !frames || n <= 0) return grl_fail(GRL_EINVAL, "jpeg_parse_batch: null / empty");
    for (int i = 0; i < n; ++i) {
        const int rc = grl_jpeg_parse(buf + offsets[i], offsets[i + 1] - offsets[i], offsets[i], &frames[i]);
        if (rc) {
            if (bad_index) *bad_index = i;
            return rc;
        }
    }
    const int sets = grl_jpeg_assign_tables(frames, n);
    return sets > 0 ? GRL_OK : sets;
}

static uint32_t scan_extent(const GrlJpegFrame* frames, int n) {
    uint32_t nbytes = 0;
    for (int i = 0; i < n; ++i) {
        const uint32_t e = frames[i].scan_off + frames[i].scan_len;
        if (e > nbytes) nbytes = e;
    }
    return nbytes;
}

// workspace layout: coefficients | planes | look-ahead tables | clean (unstuffed) streams | their lengths
struct Layout { int64_t coef, planes, lut, clean, clean_len, total; };
static Layout make_layout(const Geo& g, int n, uint32_t nbytes) {
    auto up = [](int64_t v) { return (v + 255) & ~(int64_t)255; };
    Layout l;
    l.coef = 0;
    l.planes = up((int64_t)n * g.blocks * 64 * (int64_t)sizeof(int16_t));
    l.lut = l.planes + up((int64_t)n * g.plane_bytes);
    l.clean = l.lut + up((int64_t)n * LUT_PER_SET * 2);
    l.clean_len = l.clean + up((int64_t)nbytes + 32);
    l.total = l.clean_len + up((int64_t)n * 4);
    return l;
}

extern "C" int64_t grl_jpeg_workspace_bytes(const GrlJpegFrame* frames_host, int n) {
    if (!frames_host || n <= 0 || !frames_host[0].width || !frames_host[0].hmax) return 0;
    return make_layout(make_geo(frames_host[0]), n, scan_extent(frames_host, n)).total;
}

// HOST: give every frame the index of its Huffman table set (frames[i].tabset).  Up to 4 distinct sets share look-ahead
// tables that the entropy kernel keeps in LDS (a batch of frames from one camera / encoder has ONE); with more, every
// frame gets its own (tabset = i) and the tables are read through the L2.  Returns the number of sets.
extern "C" int grl_jpeg_assign_tables(GrlJpegFrame* frames, int n) {
    if (!frames || n <= 0) return grl_fail(GRL_EINVAL, "jpeg_assign_tables: null / empty");
    constexpr size_t TB = sizeof(frames[0].maxcode) + sizeof(frames[0].valoff) + sizeof(frames[0].vals);
    static_assert(offsetof(GrlJpegFrame, vals) + sizeof(frames[0].vals) - offsetof(GrlJpegFrame, maxcode) == TB, "tables are contiguous");
    int rep[MAX_LDS_SETS], sets = 0;
    bool many = false;
    for (int i = 0; i < n && !many; ++i) {
        int u = 0;
        for (; u < sets; ++u)
            if (!memcmp(frames[i].maxcode, frames[rep[u]].maxcode, TB)) break;
        if (u == sets) {
            if (sets == MAX_LDS_SETS) { many = true; break; }
            rep[sets++] = i;
        }
        frames[i].tabset = (uint16_t)u;
    }
    if (many) {
        if (n > 65535) return grl_fail(GRL_EINVAL, "jpeg_assign_tables: more than 65535 frames with distinct tables in one batch");
        for (int i = 0; i < n; ++i) frames[i].tabset = (uint16_t)i;
        return n;
    }
    return sets;
}

extern "C" int grl_jpeg_decode_batch(const uint8_t* bytes, const GrlJpegFrame* frames_dev, const GrlJpegFrame* frames_host, int n,
                                     uint8_t* out, void* workspace, int64_t workspace_bytes, void* stream) {
    GRL_REQUIRE(bytes && frames_dev && frames_host && out && workspace && n > 0, "jpeg_decode_batch: null / empty");
    GRL_REQUIRE(((uintptr_t)bytes & 3) == 0 && ((uintptr_t)workspace & 15) == 0 && ((uintptr_t)frames_dev & 15) == 0,
                "jpeg_decode_batch: bytes must be 4-byte aligned, the workspace and the descriptors 16-byte aligned");
    static_assert(sizeof(GrlJpegFrame) == 2160 && offsetof(GrlJpegFrame, q) == 48, "GrlJpegFrame layout (include/grl_hip.h)");
    const GrlJpegFrame& f0 = frames_host[0];
    GRL_REQUIRE(f0.width && f0.height && (f0.ncomp == 1 || f0.ncomp == 3) && f0.hmax >= 1 && f0.hmax <= 2 && f0.vmax >= 1 && f0.vmax <= 2,
                "jpeg_decode_batch: frame 0 was not parsed by grl_jpeg_parse");
    uint32_t nbytes = 0;
    for (int i = 0; i < n; ++i) {
        const GrlJpegFrame& f = frames_host[i];
        bool same = f.width == f0.width && f.height == f0.height && f.ncomp == f0.ncomp && f.hmax == f0.hmax && f.vmax == f0.vmax;
        for (int c = 0; c < f0.ncomp; ++c) same = same && f.hs[c] == f0.hs[c] && f.vs[c] == f0.vs[c];
        if (!same) return grl_fail(GRL_EINVAL, "jpeg_decode_batch: frame %d has another geometry than frame 0 (decode per geometry group)", i);
        const uint32_t e = f.scan_off + f.scan_len;
        if (e < f.scan_off) return grl_fail(GRL_EINVAL, "jpeg_decode_batch: frame %d: stream range wraps", i);
        if (e > nbytes) nbytes = e;
    }
    GRL_REQUIRE(workspace_bytes >= grl_jpeg_workspace_bytes(frames_host, n), "jpeg_decode_batch: workspace too small (grl_jpeg_workspace_bytes)");
    // table sets (grl_jpeg_assign_tables): a frame's tables must be the ones of its set's first frame
    constexpr size_t TB = sizeof(f0.maxcode) + sizeof(f0.valoff) + sizeof(f0.vals);
    int sets = 0;
    for (int i = 0; i < n; ++i) sets = frames_host[i].tabset + 1 > sets ? frames_host[i].tabset + 1 : sets;
    Reps reps;
    const bool identity = sets > MAX_LDS_SETS;
    if (identity) {
        for (int i = 0; i < n; ++i)
            if (frames_host[i].tabset != i) return grl_fail(GRL_EINVAL, "jpeg_decode_batch: frame %d: table sets not assigned (grl_jpeg_assign_tables)", i);
        GRL_REQUIRE(sets == n, "jpeg_decode_batch: table sets not assigned (grl_jpeg_assign_tables)");
    } else {
        for (int u = 0; u < MAX_LDS_SETS; ++u) reps.frame[u] = -1;
        for (int i = 0; i < n; ++i) {
            const int u = frames_host[i].tabset;
            if (reps.frame[u] < 0) reps.frame[u] = i;
            else if (memcmp(frames_host[i].maxcode, frames_host[reps.frame[u]].maxcode, TB))
                return grl_fail(GRL_EINVAL, "jpeg_decode_batch: frame %d does not carry the tables of its table set (grl_jpeg_assign_tables)", i);
        }
        for (int u = 0; u < sets; ++u) GRL_REQUIRE(reps.frame[u] >= 0, "jpeg_decode_batch: a table set without a frame");
    }
    const Geo g = make_geo(f0);
    GjScanGeo sg;
    sg.mcus = g.mcux * g.mcuy; sg.ncomp = g.ncomp;
    for (int c = 0; c < 3; ++c) sg.nb[c] = g.hs[c] * g.vs[c];
    hipStream_t s = (hipStream_t)stream;
    const Layout lay = make_layout(g, n, nbytes);
    uint8_t* const wsb = reinterpret_cast<uint8_t*>(workspace);
    int16_t* coef = reinterpret_cast<int16_t*>(wsb + lay.coef);
    uint8_t* planes = wsb + lay.planes;
    uint16_t* lut = reinterpret_cast<uint16_t*>(wsb + lay.lut);
    uint8_t* clean = wsb + lay.clean;
    uint32_t* clean_len = reinterpret_cast<uint32_t*>(wsb + lay.clean_len);
    hipLaunchKernelGGL(jpeg_lut_kernel, dim3(sets, LUT_PER_SET / 256), dim3(256), 0, s, frames_dev, reps, identity ? 1 : 0, lut);
    const int lds_sets = identity ? 0 : sets;
    const size_t e_lds = 128 + 8192 + (size_t)lds_sets * LUT_PER_SET * sizeof(uint16_t);
    static const bool attr = [] {
        const int most = 128 + 8192 + MAX_LDS_SETS * LUT_PER_SET * (int)sizeof(uint16_t);
        (void)hipFuncSetAttribute((const void*)jpeg_entropy_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, most);
        (void)hipFuncSetAttribute((const void*)jpeg_entropy_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, most);
        return true;
    }();
    (void)attr;
    bool any_restart = false;
    for (int i = 0; i < n; ++i) any_restart = any_restart || frames_host[i].restart_interval != 0;
    uint32_t max_scan = 0;
    for (int i = 0; i < n; ++i) max_scan = frames_host[i].scan_len > max_scan ? frames_host[i].scan_len : max_scan;
    if (!any_restart && g_jpeg_parallel && max_scan <= (uint32_t)PAR_MAX_SCAN && g.blocks <= PAR_MAX_BLOCKS) {
        // the common case: stuffing removed by a pre-pass, then ONE WORKGROUP PER FRAME decodes self-synchronising subsequences
        const ParLayout lay = par_layout(g.blocks, (int)((max_scan + 3) / 4));
        static int attr_bytes = 0;
        if (lay.total > attr_bytes) {
            (void)hipFuncSetAttribute((const void*)jpeg_entropy_par_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lay.total);
            attr_bytes = lay.total;
        }
        if (hipMemsetAsync(coef, 0, (size_t)n * g.blocks * 64 * sizeof(int16_t), s) != hipSuccess) return grl_check_launch("jpeg_decode_batch (memset)");
        hipLaunchKernelGGL(jpeg_unstuff_kernel, dim3(n), dim3(UT), 0, s, bytes, frames_dev, clean, clean_len);
        static const uint32_t seq_bits = [] { const char* e = getenv("GRL_JPEG_SEQ_BITS"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 32 ? (v + 31) & ~31 : 512); }();
        hipLaunchKernelGGL(jpeg_entropy_par_kernel, dim3(n), dim3(PT), (size_t)lay.total, s, frames_dev, coef, g.blocks, g.mcux * g.mcuy, lut,
                           clean, clean_len, lay, seq_bits);
    } else if (!any_restart) {
        // frames too large for the workgroup form: one lane per frame on the clean stream
        hipLaunchKernelGGL(jpeg_unstuff_kernel, dim3(n), dim3(UT), 0, s, bytes, frames_dev, clean, clean_len);
        hipLaunchKernelGGL(jpeg_entropy_kernel<0>, dim3(grl_ceil_div(n, EW)), dim3(EW), e_lds, s, bytes, nbytes, frames_dev, n, coef, sg,
                           g.blocks, lut, lds_sets, clean, clean_len);
    } else {
        hipLaunchKernelGGL(jpeg_entropy_kernel<1>, dim3(grl_ceil_div(n, EW)), dim3(EW), e_lds, s, bytes, nbytes, frames_dev, n, coef, sg,
                           g.blocks, lut, lds_sets, clean, clean_len);
    }
    const int64_t nblk = (int64_t)n * g.blocks;
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, s, coef, frames_dev, n, planes, g);
    const int64_t npix = (int64_t)n * g.width * g.height;
    if (g.width % 4 == 0 && ((uintptr_t)out & 3) == 0)
        hipLaunchKernelGGL(jpeg_color_kernel<4>, dim3((unsigned)((npix / 4 + 255) / 256)), dim3(256), 0, s, planes, frames_dev, n, out, g);
    else
        hipLaunchKernelGGL(jpeg_color_kernel<1>, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, planes, frames_dev, n, out, g);
    return grl_check_launch("grl_jpeg_decode_batch");
}
