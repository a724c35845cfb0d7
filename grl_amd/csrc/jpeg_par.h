// Intra-frame parallel entropy decoding (round 6, second form): self-synchronising decode of the Huffman stream.
//
// The first device decoder gave a frame ONE lane (a scan is a serial bit stream): 10.3 ms per batch whatever its size, and
// a 128-frame batch occupied two waves of the chip for that long.  Huffman streams re-synchronise: a decoder started at an
// arbitrary bit with a wrong state falls into step with the true symbol sequence after a few symbols (Weissenberger & Schmidt,
// "Massively Parallel Huffman Decoding on GPUs", 2018; for JPEG the state is the bit position plus the place inside the
// MCU -- block slot z and coefficient index k).  So a frame's clean stream (jpeg_unstuff_kernel) is cut into subsequences of
// L bits (512 for a MARS frame: ~210 of them), one per lane of a 256-lane workgroup:
//   1. every lane walks its subsequence from a GUESSED state (z = 0, k = 0; lane 0's is the true one) and publishes where
//      and in which state it left it;
//   2. a lane whose left neighbour's exit state differs from the entry state it used re-walks from that exit state; repeat
//      until no lane changed (lane i is certainly right after i rounds; in practice 2-3 rounds);
//   3. an exclusive scan of the completed-block counts tells every lane which block it starts in;
//   4. a last walk writes the coefficients (DC as the DIFFERENCE it is coded as), the last lane carrying on past the data
//      until the frame's blocks are complete (zero bits, as libjpeg feeds them);
//   5. a prefix sum per component turns the DC differences into DC values.
// The result is the serial decoder's, coefficient for coefficient, damaged streams included: the fixed point of step 2 is
// unique (entry[0] is true, entry[i] = exit[i - 1]).  This header holds the per-lane logic as plain functions that also
// compile as host C++; tests/jpeg_core_host.cpp emulates the lanes one after the other and compares with the oracle.
#pragma once
#include <stdint.h>

#include "jpeg_core.h"

struct GjState {
    uint32_t bit;          // position in the clean stream
    int32_t z;             // block slot inside the MCU (0 .. blocks per MCU - 1)
    int32_t k;             // next coefficient index: 0 = the DC symbol is due, 1..63 inside the AC run
};
GRL_HD static inline bool gj_same(const GjState& a, const GjState& b) { return a.bit == b.bit && a.z == b.z && a.k == b.k; }

struct GjParTables {
    const uint16_t* lut;   // GJ_LUT_PER_SET look-ahead entries of the frame's table set
    const GrlJpegFrame* fr;
    const uint8_t* nat;    // zigzag -> natural order, 80 entries
    int bpm;               // blocks per MCU
    int8_t td[8], ta[8];   // per block slot: DC / AC table index (0..3 as in GrlJpegFrame.maxcode)
    int8_t comp[8];        // per block slot: its component
};

// one Huffman symbol from the top 16 bits of a 31-bit window; `len` = code length consumed
GRL_HD static inline int gj_par_symbol(uint32_t look31, const GjParTables& T, int t, int& len) {
    const uint32_t look = look31 >> 15;
    const int bits = gj_lut_bits(t);
    const uint32_t e = T.lut[gj_lut_offset(t) + (look >> (16 - bits))];
    if (e) { len = (int)(e >> 8); return (int)(e & 255u); }
    for (int l = bits + 1; l <= 16; ++l) {
        const int code = (int)(look >> (16 - l));
        if (code <= T.fr->maxcode[t][l]) {
            len = l;
            return T.fr->vals[t][(code + T.fr->valoff[t][l]) & 255];
        }
    }
    len = 16;              // corrupt stream: libjpeg warns and returns 0
    return 0;
}

// sequential reader over the byte-swapped stream: 64-bit window, refilled one dword at a time (the refill is not on the
// symbol's dependent chain -- a walk is a run of ~75 symbols from one bit position on)
struct GjBeReader {
    const uint32_t* be;
    uint32_t ndw, next;      // dwords in the stream, next dword to append
    uint64_t acc;            // the low `cnt` bits are the unread ones
    int cnt;
};
GRL_HD static inline void gj_be_init(GjBeReader& r, const uint32_t* be, uint32_t ndw, uint32_t bit) {
    const uint32_t i = bit >> 5, o = bit & 31u;
    r.be = be; r.ndw = ndw;
    r.acc = ((uint64_t)(i < ndw ? be[i] : 0u) << 32) | (uint64_t)(i + 1 < ndw ? be[i + 1] : 0u);
    r.cnt = 64 - (int)o;
    r.next = i + 2;
}
GRL_HD static inline uint32_t gj_be_look31(GjBeReader& r) {            // the next 31 bits (cnt >= 31 after the refill)
    if (r.cnt <= 32) {
        r.acc = (r.acc << 32) | (uint64_t)(r.next < r.ndw ? r.be[r.next] : 0u);
        r.cnt += 32;
        ++r.next;
    }
    return (uint32_t)(r.acc >> (r.cnt - 31)) & 0x7fffffffu;
}

// One symbol of the serial decoder's state machine (jpeg_core.h gj_decode_scan, same rules: run / size, EOB, ZRL, the guard
// entries of `nat` for runs that leave the block) applied to the 31-bit window `look31` at s.bit.  emit(coefficient index
// in natural order, value) is called for a coefficient; `used` = bits consumed; returns true when the symbol completed a
// block.
template <class Emit>
GRL_HD static inline bool gj_par_apply(uint32_t look31, const GjParTables& T, GjState& s, int& used, Emit&& emit) {
    int len;
    if (s.k == 0) {
        const int sz = gj_par_symbol(look31, T, T.td[s.z], len) & 15;
        int diff = 0;
        if (sz) diff = gj_extend((int)((look31 >> (31 - len - sz)) & ((1u << sz) - 1u)), sz);
        emit(0, diff);
        used = len + sz;
        s.k = 1;
        return false;
    }
    const int rs = gj_par_symbol(look31, T, T.ta[s.z], len);
    const int r = rs >> 4, sz = rs & 15;
    bool done;
    if (sz) {
        const int kk = s.k + r;
        emit((int)T.nat[kk], gj_extend((int)((look31 >> (31 - len - sz)) & ((1u << sz) - 1u)), sz));
        used = len + sz;
        s.k = kk + 1;
        done = s.k >= 64;
    } else {
        used = len;
        if (r != 15) {
            done = true;
        } else {
            s.k += 16;
            done = s.k >= 64;
        }
    }
    if (done) {
        s.k = 0;
        s.z = s.z + 1 == T.bpm ? 0 : s.z + 1;
    }
    return done;
}

// one symbol through the reader
template <class Emit>
GRL_HD static inline bool gj_par_step(GjBeReader& r, const GjParTables& T, GjState& s, Emit&& emit) {
    int used;
    const bool done = gj_par_apply(gj_be_look31(r), T, s, used, emit);
    r.cnt -= used;
    s.bit += (uint32_t)used;
    return done;
}

// walk [s.bit, end_bit): every symbol that STARTS before end_bit; returns the number of blocks completed
GRL_HD static inline int gj_par_walk(const uint32_t* be, uint32_t ndw, const GjParTables& T, GjState& s, uint32_t end_bit) {
    GjBeReader r;
    gj_be_init(r, be, ndw, s.bit);
    int blocks = 0;
    while (s.bit < end_bit)
        if (gj_par_step(r, T, s, [](int, int) {})) ++blocks;
    return blocks;
}

// subsequence length in bits for a stream of nbits and at most `lanes` lanes: >= 512 (measured: 1024 / 768 / 512 / 384 / 256
// bits -> 1.04 / 0.90 / 0.86 / 0.87 / 0.88 ms per 128-frame batch), a multiple of 32
GRL_HD static inline uint32_t gj_par_seq_bits(uint32_t nbits, uint32_t lanes, uint32_t min_bits = 512u) {
    uint32_t L = (nbits + lanes - 1) / lanes;
    L = (L + 31u) & ~31u;
    return L < min_bits ? min_bits : L;       // (any multiple of 32 >= 32 is correct: a symbol is at most 31 bits long)
}

GRL_HD static inline void gj_par_tables(GjParTables& T, const GrlJpegFrame* fr, const uint16_t* lut, const uint8_t* nat) {
    T.lut = lut; T.fr = fr; T.nat = nat;
    int z = 0;
    for (int c = 0; c < fr->ncomp; ++c)
        for (int b = 0; b < fr->hs[c] * fr->vs[c] && z < 8; ++b, ++z) {
            T.td[z] = (int8_t)(fr->td[c] & 1);
            T.ta[z] = (int8_t)(2 + (fr->ta[c] & 1));
            T.comp[z] = (int8_t)c;
        }
    T.bpm = z;
}
