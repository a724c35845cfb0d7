// Entropy-decoding core of the device JPEG decoder (jpeg.hip), written so that the SAME source also compiles as plain
// host C++ (GRL_HD expands to nothing): a frame is decoded by one lane with no cross-lane operation, so a CPU build of
// this header run on one frame at a time exercises exactly the logic a lane runs (tests/test_jpeg_cpu.py builds it with
// g++ and compares the coefficients with the oracle's -- sanitizers and debuggers work there; the GPU pool has neither).
//
// ITU-T T.81 F.2.2 (jdhuff.c decode_mcu): per block a DC difference and run/size coded AC coefficients.
//   * bit reader: 64-bit accumulator, refilled 32 bits at a time from an ALIGNED dword when that dword holds no 0xFF byte
//     (no stuffing, no marker: ~98 % of the dwords of a photographic scan), byte by byte otherwise (0xFF00 stuffing, fill
//     bytes, RSTn / EOI, the unaligned head and the tail of a scan);
//   * Huffman symbols through a look-ahead table (12 bits for AC, 9 for DC; one read: length << 8 | symbol; 0 = longer), the
//     rare long codes by the canonical maxcode / valoff search;
//   * a block's coefficients are assembled in a small staging area (LDS on the device) and leave as whole 128-byte blocks.
#pragma once
#include <stdint.h>

#include "../../include/grl_hip.h"

#ifndef GRL_HD
#define GRL_HD
#endif

// look-ahead bits: AC tables 12 (codes up to 16 bits, the long ones rare), DC tables 9 (baseline DC codes are <= 9 / 11 bits).
// One table SET = [DC0 512][DC1 512][AC0 4096][AC1 4096] uint16 = 18 KiB: with the 8 KiB block stage the entropy workgroup
// needs 26.3 KiB of LDS -- what is LEFT on a CU next to two 64 KiB GEMM workgroups (or one 128 KiB bf16 tile).  With 12-bit
// DC tables (41 KiB) the decoder could not co-reside: a persistent GEMM launch then ran with workgroups missing and the
// eval step fed from JPEG bytes took 20 ms instead of 14.5.
#define GJ_AC_BITS 12
#define GJ_DC_BITS 9
#define GJ_LUT_PER_SET (2 * (1 << GJ_DC_BITS) + 2 * (1 << GJ_AC_BITS))
GRL_HD static inline int gj_lut_bits(int t) { return t < 2 ? GJ_DC_BITS : GJ_AC_BITS; }
GRL_HD static inline int gj_lut_offset(int t) { return t < 2 ? t << GJ_DC_BITS : (2 << GJ_DC_BITS) + ((t - 2) << GJ_AC_BITS); }

struct GjBits {
    const uint8_t* base;      // the batch's byte buffer
    uint32_t pos, end;        // next stream byte, one past the scan
    uint32_t limit;           // bytes of the BUFFER that may be read as whole dwords (its length rounded down to 4)
    uint64_t acc;             // bits, MSB first, in the low `cnt` bits
    int cnt;
    int marker;               // a marker was met: zero bits from here on (libjpeg's "insufficient data" behaviour)
    int ffp;                  // the last byte seen was 0xFF (stuffing / marker decision pending)
    uint32_t caddr, cword;    // one-dword cache of the byte path (restart search)
};

GRL_HD static inline uint32_t gj_load_dword(const GjBits& b, uint32_t a) {
    if (a + 4 <= b.limit) return *reinterpret_cast<const uint32_t*>(b.base + a);
    uint32_t v = 0;                                                       // the buffer's last, partial dword
    for (int i = 0; i < 4; ++i)
        if (a + i < b.end) v |= (uint32_t)b.base[a + i] << (8 * i);
    return v;
}

GRL_HD static inline int gj_byte_at(GjBits& b, uint32_t p) {
    const uint32_t a = p & ~3u;
    if (a != b.caddr) { b.cword = gj_load_dword(b, a); b.caddr = a; }
    return (int)((b.cword >> (8 * (p & 3u))) & 255u);
}

GRL_HD static inline uint32_t gj_bswap(uint32_t w) { return (w >> 24) | ((w >> 8) & 0xff00u) | ((w << 8) & 0xff0000u) | (w << 24); }

// after the call cnt >= 33 (the most a symbol consumes is 16 code bits + 15 extra bits).  One aligned dword load per
// iteration; its bytes go through the stuffing state machine as STRAIGHT-LINE code (no inner loop, no second load): a wave
// decodes 64 frames in lockstep, and whatever one lane has to do every lane waits for.
GRL_HD static inline void gj_fill(GjBits& b) {
    while (b.cnt <= 32) {
        if (b.marker || b.pos >= b.end) {                                 // past the data: zero bits (libjpeg does the same)
            b.marker = 1;
            b.acc <<= 32;
            b.cnt += 32;
            continue;
        }
        const uint32_t a = b.pos & ~3u, i0 = b.pos & 3u;
        const uint32_t w = gj_load_dword(b, a);
        const uint32_t nvalid = b.end - a < 4u ? b.end - a : 4u;          // bytes of this dword that belong to the scan
        const uint32_t inv = ~w;
        if (i0 == 0 && nvalid == 4 && !b.ffp && ((inv - 0x01010101u) & ~inv & 0x80808080u) == 0) {
            b.acc = (b.acc << 32) | (uint64_t)gj_bswap(w);                // no 0xFF byte: 32 bits at once
            b.cnt += 32;
            b.pos += 4;
            continue;
        }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t i = 0; i < 4; ++i) {
            const uint32_t byte = (w >> (8 * i)) & 255u;
            if (i >= i0 && i < nvalid && !b.marker) {
                if (b.ffp) {
                    if (byte == 0) { b.acc = (b.acc << 8) | 0xFFu; b.cnt += 8; b.ffp = 0; }   // stuffed zero: a data byte 0xFF
                    else if (byte != 0xFF) { b.marker = 1; b.pos = a + i - 1; }               // RSTn / EOI / ...: stays unread
                } else if (byte == 0xFF) {
                    b.ffp = 1;
                } else {
                    b.acc = (b.acc << 8) | (uint64_t)byte;
                    b.cnt += 8;
                }
            }
        }
        if (!b.marker) b.pos = a + nvalid;
    }
}

GRL_HD static inline void gj_bits_init(GjBits& b, const uint8_t* bytes, uint32_t limit, const GrlJpegFrame* fr) {
    b.base = bytes;
    b.pos = fr->scan_off;
    b.end = fr->scan_off + fr->scan_len;
    b.limit = limit;
    b.acc = 0; b.cnt = 0; b.marker = 0; b.ffp = 0;
    b.caddr = 0xffffffffu; b.cword = 0;
}

// discard the partial byte, find the RSTn marker, step over it (restart intervals: the general reader only)
GRL_HD static inline void gj_restart(GjBits& b) {
    b.cnt = 0; b.acc = 0; b.ffp = 0;
    uint32_t q = b.pos;
    while (q + 1 < b.end) {
        if (gj_byte_at(b, q) == 0xFF) {
            const int m2 = gj_byte_at(b, q + 1);
            if (m2 >= 0xD0 && m2 <= 0xD7) break;
        }
        ++q;
    }
    if (q + 1 < b.end) b.pos = q + 2;
    b.marker = 0;
}

// ---- the CLEAN reader: a scan whose stuffing and trailing marker were removed by a pre-pass (jpeg_unstuff_kernel /
// gj_unstuff_host) into dword-aligned, zero-padded storage.  Refill is one already-loaded dword, byte-swapped, and the
// request for the next one -- no branch, no 0xFF handling, no exposed load latency.  With 64 frames decoded in lockstep the
// general reader's "rare" paths (an 0xFF byte in 1.6 % of the dwords, a refill every ~4.5 symbols at a different moment in
// every lane) run at almost every step and every lane waits for them; this reader has none.
struct GjClean {
    const uint32_t* p;        // next dword to LOAD
    int left;                 // dwords not loaded yet
    uint32_t nw;              // the dword loaded ahead
    uint64_t acc;
    int cnt;
};
GRL_HD static inline void gj_clean_init(GjClean& b, const uint8_t* clean, uint32_t nbytes) {
    b.p = reinterpret_cast<const uint32_t*>(clean);
    b.left = (int)((nbytes + 3u) >> 2);
    b.acc = 0; b.cnt = 0;
    b.nw = b.left > 0 ? *b.p : 0u;
    ++b.p; --b.left;
}
GRL_HD static inline void gj_fill(GjClean& b) {
    while (b.cnt <= 32) {
        const uint32_t w = b.nw;
        b.nw = b.left > 0 ? *b.p : 0u;      // past the data: zero bits (libjpeg does the same)
        ++b.p; --b.left;
        b.acc = (b.acc << 32) | (uint64_t)gj_bswap(w);
        b.cnt += 32;
    }
}
GRL_HD static inline void gj_restart(GjClean&) {}     // (scans with restart intervals take the general reader)

// the unstuffing rule, per byte of a scan (prev / next: the neighbours, -1 beyond the scan):
//   a marker starts at an 0xFF that is followed by neither 0x00 (stuffing) nor 0xFF (fill) -- or by nothing;
//   a byte is data unless it is the 0x00 behind an 0xFF, or an 0xFF that is not followed by 0x00.
GRL_HD static inline bool gj_marker_starts(int cur, int next) { return cur == 0xFF && next != 0x00 && next != 0xFF; }
GRL_HD static inline bool gj_is_data(int prev, int cur, int next) { return !(cur == 0x00 && prev == 0xFF) && !(cur == 0xFF && next != 0x00); }

template <class R>
GRL_HD static inline int gj_get_bits(R& b, int s) {                  // s in 1..16, cnt >= s
    b.cnt -= s;
    return (int)((b.acc >> b.cnt) & ((1u << s) - 1u));
}

GRL_HD static inline int gj_extend(int x, int s) { return x < (1 << (s - 1)) ? x + (int)((~0u) << s) + 1 : x; }

// one Huffman symbol of table t (0, 1: DC; 2, 3: AC): look-ahead table first, canonical search for the codes it does not cover
template <class R>
GRL_HD static inline int gj_symbol(R& b, const uint16_t* lut /* this table's look-ahead entries */, const GrlJpegFrame* fr, int t, const int bits) {
    const uint32_t look = (uint32_t)(b.acc >> (b.cnt - 16)) & 0xffffu;
    const uint32_t e = lut[look >> (16 - bits)];
    if (e) { b.cnt -= (int)(e >> 8); return (int)(e & 255u); }
    for (int l = bits + 1; l <= 16; ++l) {
        const int code = (int)(look >> (16 - l));
        if (code <= fr->maxcode[t][l]) {
            b.cnt -= l;
            return fr->vals[t][(code + fr->valoff[t][l]) & 255];
        }
    }
    b.cnt -= 16;       // corrupt stream: libjpeg warns and returns 0
    return 0;
}

// the look-ahead entry for the `bits` bits `p` of table t (jpeg_lut_kernel / the host test build the tables with this)
GRL_HD static inline uint16_t gj_lut_entry(const GrlJpegFrame* fr, int t, int p) {
    const int bits = gj_lut_bits(t);
    for (int l = 1; l <= bits; ++l) {
        const int code = p >> (bits - l);
        if (code <= fr->maxcode[t][l]) return (uint16_t)((l << 8) | fr->vals[t][(code + fr->valoff[t][l]) & 255]);
    }
    return 0;
}

// (the staging area is written as int16 and read back as dwords, the output as 16-byte rows: may_alias types, or the
//  compiler is free to forward the stage's zero-fill to the read-back -- it did)
typedef uint32_t __attribute__((may_alias)) gj_u32a;
typedef int16_t __attribute__((may_alias)) gj_i16a;
struct __attribute__((aligned(16), may_alias)) GjU4 { uint32_t x, y, z, w; };

struct GjScanGeo {
    int mcus;                 // MCUs per frame
    int ncomp;
    int nb[3];                // blocks of each component per MCU (hs * vs)
};

// Decode one frame's scan: `lut` = the GJ_LUT_PER_SET entries [DC0, DC1, AC0, AC1] of this frame's table set,
// `nat` = the 64 (+16 guard) entry zigzag -> natural order table, `out` = this frame's coefficients [blocks][64] (every
// block is written whole).  A block is assembled in `stage` -- coefficient i lives at stage[(i >> 1) * sstride + (i & 1)]:
// on the device that is LDS, dword-interleaved over the wave's lanes (sstride = 128 int16: lane l's dword w sits in bank l
// whatever w is), on the host a plain 64-entry array (sstride = 2) -- and leaves as eight 16-byte stores.  (The first
// device version scattered 2-byte stores straight to HBM: gfx9's vmcnt counts stores too, so every later load of the byte
// stream waited for 64 partial-line writes to be acknowledged -- 1700 cycles per symbol.)
template <class R>
GRL_HD static inline void gj_decode_scan(R& b, const GrlJpegFrame* fr, const uint16_t* lut,
                                         const uint8_t* nat, int16_t* out, const GjScanGeo& g, int16_t* stage, int sstride) {
    int pred[3] = {0, 0, 0};
    const int ri = fr->restart_interval;
    int tdc[3], tac[3];
    for (int c = 0; c < 3; ++c) { tdc[c] = fr->td[c] & 1; tac[c] = 2 + (fr->ta[c] & 1); }
    int left = ri;
    int blk = 0;
    for (int w = 0; w < 32; ++w) *reinterpret_cast<gj_u32a*>(stage + w * sstride) = 0u;
    for (int m = 0; m < g.mcus; ++m) {
        if (ri) {
            if (left == 0) {
                gj_restart(b);                       // DC predictions restart
                pred[0] = pred[1] = pred[2] = 0;
                left = ri;
            }
            --left;
        }
        for (int c = 0; c < g.ncomp; ++c) {
            const int td = tdc[c], ta = tac[c];
            const uint16_t* const lut_dc = lut + gj_lut_offset(td);
            const uint16_t* const lut_ac = lut + gj_lut_offset(ta);
            for (int bi = 0; bi < g.nb[c]; ++bi, ++blk) {
                gj_fill(b);
                int s = gj_symbol(b, lut_dc, fr, td, GJ_DC_BITS) & 15;
                if (s) s = gj_extend(gj_get_bits(b, s), s);
                pred[c] += s;
                *reinterpret_cast<gj_i16a*>(stage) = (int16_t)pred[c];
                for (int k = 1; k < 64; ++k) {
                    gj_fill(b);
                    const int rs = gj_symbol(b, lut_ac, fr, ta, GJ_AC_BITS);
                    const int r = rs >> 4, sz = rs & 15;
                    if (sz) {
                        k += r;
                        const int i = nat[k];
                        *reinterpret_cast<gj_i16a*>(stage + (i >> 1) * sstride + (i & 1)) = (int16_t)gj_extend(gj_get_bits(b, sz), sz);
                    } else {
                        if (r != 15) break;
                        k += 15;
                    }
                }
                // the block leaves as eight 16-byte rows; the stage is cleared for the next one on the way
                gj_u32a* const o = reinterpret_cast<gj_u32a*>(out + (int64_t)blk * 64);
                for (int w4 = 0; w4 < 8; ++w4) {
                    uint32_t v[4];
                    for (int e = 0; e < 4; ++e) {
                        gj_u32a* const sp = reinterpret_cast<gj_u32a*>(stage + (w4 * 4 + e) * sstride);
                        v[e] = *sp;
                        *sp = 0u;
                    }
                    GjU4 q4 = {v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<GjU4*>(o + w4 * 4) = q4;
                }
            }
        }
    }
}
