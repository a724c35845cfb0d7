// CPU build of the device decoder's entropy core (grl_amd/csrc/jpeg_core.h compiles as plain C++): one frame at a time,
// exactly the per-lane logic of jpeg_entropy_kernel.  Built by tests/test_jpeg_cpu.py with g++ (no GPU, no HIP).
#include <stdint.h>
#include <string.h>
#include <vector>
#define GRL_HD
#include "../grl_amd/csrc/jpeg_core.h"
#include "../grl_amd/csrc/jpeg_par.h"

static const uint8_t kNat[80] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

static void setup(const GrlJpegFrame* fr, uint16_t* lut, GjScanGeo& g) {
    for (int t = 0; t < 4; ++t)
        for (int p = 0; p < (1 << gj_lut_bits(t)); ++p) lut[gj_lut_offset(t) + p] = gj_lut_entry(fr, t, p);
    const int mcux = (fr->width + 8 * fr->hmax - 1) / (8 * fr->hmax), mcuy = (fr->height + 8 * fr->vmax - 1) / (8 * fr->vmax);
    g.mcus = mcux * mcuy;
    g.ncomp = fr->ncomp;
    for (int c = 0; c < 3; ++c) g.nb[c] = fr->hs[c] * fr->vs[c];
}

// the GENERAL reader (stuffing / markers / restart intervals handled while decoding).
// buf: the batch byte buffer (nbytes long), fr: the parsed frame, out: int16 [blocks][64]
extern "C" int gj_host_decode(const uint8_t* buf, uint32_t nbytes, const GrlJpegFrame* fr, int16_t* out) {
    static uint16_t lut[GJ_LUT_PER_SET];
    GjScanGeo g;
    setup(fr, lut, g);
    alignas(16) int16_t stage[64];
    GjBits b;
    gj_bits_init(b, buf, nbytes & ~3u, fr);
    gj_decode_scan(b, fr, lut, kNat, out, g, stage, 2);
    return 0;
}

// the CLEAN reader behind the unstuffing pre-pass (here a serial loop over the same per-byte rule the kernel applies).
// Returns the number of data bytes the pre-pass kept.
extern "C" int gj_host_decode_clean(const uint8_t* buf, uint32_t nbytes, const GrlJpegFrame* fr, int16_t* out) {
    static uint16_t lut[GJ_LUT_PER_SET];
    GjScanGeo g;
    setup(fr, lut, g);
    const uint8_t* src = buf + fr->scan_off;
    const int len = (int)fr->scan_len;
    std::vector<uint32_t> clean((size_t)len / 4 + 4, 0u);
    uint8_t* dst = reinterpret_cast<uint8_t*>(clean.data());
    int kept = 0;
    for (int i = 0; i < len; ++i) {
        const int prev = i > 0 ? src[i - 1] : 0, cur = src[i], next = i + 1 < len ? src[i + 1] : -1;
        if (gj_marker_starts(cur, next)) break;
        if (gj_is_data(prev, cur, next)) dst[kept++] = (uint8_t)cur;
    }
    alignas(16) int16_t stage[64];
    GjClean b;
    gj_clean_init(b, dst, (uint32_t)kept);
    gj_decode_scan(b, fr, lut, kNat, out, g, stage, 2);
    return kept;
}

// the PARALLEL form (jpeg_par.h), its lanes emulated one after the other: unstuff, subsequences of L bits, walk from a
// guessed state, re-walk until no lane's entry state changes (Jacobi rounds, as the workgroup does), scan of the block
// counts, writing walk, DC prefix sums.  `lanes` / `min_bits` as in gj_par_seq_bits (small values stress the
// synchronisation).  Returns the number of re-walk rounds.
extern "C" int gj_host_decode_par(const uint8_t* buf, uint32_t nbytes, const GrlJpegFrame* fr, int16_t* out, int lanes, int min_bits) {
    (void)nbytes;
    static uint16_t lut[GJ_LUT_PER_SET];
    GjScanGeo g;
    setup(fr, lut, g);
    const uint8_t* src = buf + fr->scan_off;
    const int len = (int)fr->scan_len;
    std::vector<uint8_t> clean((size_t)len + 16, 0);
    int kept = 0;
    for (int i = 0; i < len; ++i) {
        const int prev = i > 0 ? src[i - 1] : 0, cur = src[i], next = i + 1 < len ? src[i + 1] : -1;
        if (gj_marker_starts(cur, next)) break;
        if (gj_is_data(prev, cur, next)) clean[kept++] = (uint8_t)cur;
    }
    const uint32_t ndw = (uint32_t)(kept + 3) / 4;
    std::vector<uint32_t> be(ndw + 2, 0u);
    for (uint32_t i = 0; i < ndw; ++i)
        be[i] = ((uint32_t)clean[4 * i] << 24) | ((uint32_t)clean[4 * i + 1] << 16) | ((uint32_t)clean[4 * i + 2] << 8) | clean[4 * i + 3];
    GjParTables T;
    gj_par_tables(T, fr, lut, kNat);
    int total_blocks = 0;
    for (int c = 0; c < g.ncomp; ++c) total_blocks += g.nb[c];
    total_blocks *= g.mcus;
    const uint32_t nbits = (uint32_t)kept * 8u;
    const uint32_t L = gj_par_seq_bits(nbits, (uint32_t)lanes, (uint32_t)min_bits);
    const int S = nbits ? (int)((nbits + L - 1) / L) : 1;
    std::vector<GjState> entry(S), exit_(S);
    std::vector<int> nblk(S);
    for (int i = 0; i < S; ++i) {
        entry[i] = GjState{(uint32_t)i * L, 0, 0};
        GjState s = entry[i];
        nblk[i] = gj_par_walk(be.data(), ndw, T, s, (uint32_t)(i + 1) * L);
        exit_[i] = s;
    }
    int rounds = 0;
    for (bool changed = true; changed; ++rounds) {
        changed = false;
        std::vector<GjState> prev = exit_;                    // Jacobi: every lane reads the previous round's exits
        for (int i = 1; i < S; ++i)
            if (!gj_same(prev[i - 1], entry[i])) {
                entry[i] = prev[i - 1];
                GjState s = entry[i];
                nblk[i] = gj_par_walk(be.data(), ndw, T, s, (uint32_t)(i + 1) * L);
                exit_[i] = s;
                changed = true;
            }
    }
    // writing walk
    memset(out, 0, (size_t)total_blocks * 64 * sizeof(int16_t));
    int first = 0;
    for (int i = 0; i < S; ++i) {
        GjState s = entry[i];
        int b = first;
        const uint32_t end = (uint32_t)(i + 1) * L;
        auto emit = [&](int idx, int v) { if (b < total_blocks) out[(size_t)b * 64 + idx] = (int16_t)v; };
        GjBeReader r;
        gj_be_init(r, be.data(), ndw, s.bit);
        while (s.bit < end || (i == S - 1 && b < total_blocks))
            if (gj_par_step(r, T, s, emit)) ++b;
        first += nblk[i];
    }
    // DC differences -> DC values, per component in scan order
    int pred[3] = {0, 0, 0};
    for (int b = 0; b < total_blocks; ++b) {
        const int c = T.comp[b % T.bpm];
        pred[c] += out[(size_t)b * 64];
        out[(size_t)b * 64] = (int16_t)pred[c];
    }
    return rounds;
}
