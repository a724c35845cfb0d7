/* ORACLE -- test infrastructure only (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 *
 * CPU restatement of what `Image.open(path).convert('RGB')` computes for a baseline JPEG -- the per-frame decode of
 * /root/reference/reid/data/video_loader.py:124-141 (PIL -> libjpeg-turbo; not vendored in the reference: Pillow's
 * decoder with libjpeg's defaults dct_method = JDCT_ISLOW, do_fancy_upsampling = TRUE).  The arithmetic follows the
 * published libjpeg algorithms, which libjpeg-turbo's SIMD paths reproduce bit for bit:
 *   entropy decoding     ITU-T T.81 Annex F.2.2 (jdhuff.c decode_mcu), byte stuffing, restart intervals
 *   dequantisation+IDCT  jidctint.c jpeg_idct_islow (CONST_BITS 13, PASS1_BITS 2), output through the post-IDCT range table
 *   chroma upsampling    jdsample.c h2v2_fancy_upsample / h2v1_fancy_upsample / h1v2_fancy_upsample (triangle filter),
 *                        edge context rows replicated as jdmainct.c does; plain replication when downsampled_width <= 2
 *   colour conversion    jdcolor.c ycc_rgb_convert (16-bit fixed-point tables)
 * Pinned by tests/test_jpeg_cpu.py against Pillow itself on generated frames (sizes off the MCU grid, 4:4:4 / 4:2:2 /
 * 4:2:0 / 4:4:0 / grey, qualities 30..100, restart intervals) and by tests/golden/jpeg_frames.npz.
 *
 * Scope: 8-bit baseline / extended-sequential Huffman (SOF0 / SOF1), one scan, 1 or 3 components (YCbCr by JFIF
 * convention, RGB when an Adobe marker says transform 0), luma sampling 1x1, 2x1 or 2x2 with 1x1 chroma.
 * Anything else (progressive, arithmetic, CMYK, 12-bit) returns an error code -- the caller decides.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GJ_OK 0
#define GJ_EFORMAT (-1)      /* not a JPEG / truncated header */
#define GJ_EUNSUPPORTED (-2) /* valid JPEG outside the scope above */
#define GJ_ENOMEM (-3)

static const uint8_t kNatural[64 + 16] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

typedef struct {
    int present;
    uint8_t bits[17];
    uint8_t vals[256];
    int32_t maxcode[18]; /* largest code of length l (-1 if none) */
    int32_t valoff[17];  /* huffval[] offset for codes of length l */
} HuffTab;

typedef struct {
    int width, height, ncomp;
    int hs[3], vs[3], tq[3], td[3], ta[3];
    int hmax, vmax;
    int restart_interval;
    int rgb;                /* 1: components are R,G,B (Adobe transform 0) */
    uint16_t q[4][64];      /* natural order */
    int qpresent[4];
    HuffTab dc[4], ac[4];
    const uint8_t* scan;
    size_t scan_len;
} Jpeg;

static void build_huff(HuffTab* t) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        t->valoff[l] = k - code;
        if (t->bits[l]) {
            k += t->bits[l];
            code += t->bits[l];
            t->maxcode[l] = code - 1;
        } else {
            t->maxcode[l] = -1;
        }
        code <<= 1;
    }
    t->maxcode[17] = 0x7fffffff;
}

static int parse(const uint8_t* p, size_t n, Jpeg* j) {
    memset(j, 0, sizeof(*j));
    if (n < 4 || p[0] != 0xFF || p[1] != 0xD8) return GJ_EFORMAT;
    size_t i = 2;
    int have_sof = 0, adobe = -1;
    while (i + 4 <= n) {
        if (p[i] != 0xFF) return GJ_EFORMAT;
        while (i < n && p[i] == 0xFF) ++i;           /* fill bytes */
        if (i >= n) return GJ_EFORMAT;
        const int m = p[i++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return GJ_EFORMAT;            /* EOI before SOS */
        if (i + 2 > n) return GJ_EFORMAT;
        const size_t len = ((size_t)p[i] << 8) | p[i + 1];
        if (len < 2 || i + len > n) return GJ_EFORMAT;
        const uint8_t* s = p + i + 2;
        const size_t sl = len - 2;
        if (m == 0xC0 || m == 0xC1) {
            if (sl < 6 || s[0] != 8) return GJ_EUNSUPPORTED;
            j->height = (s[1] << 8) | s[2];
            j->width = (s[3] << 8) | s[4];
            j->ncomp = s[5];
            if ((j->ncomp != 1 && j->ncomp != 3) || sl < 6 + 3 * (size_t)j->ncomp) return GJ_EUNSUPPORTED;
            if (j->width <= 0 || j->height <= 0) return GJ_EUNSUPPORTED;
            for (int c = 0; c < j->ncomp; ++c) {
                j->hs[c] = s[7 + 3 * c] >> 4;
                j->vs[c] = s[7 + 3 * c] & 15;
                j->tq[c] = s[8 + 3 * c];
                if (j->tq[c] > 3) return GJ_EFORMAT;
            }
            have_sof = 1;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return GJ_EUNSUPPORTED;                 /* progressive, lossless, arithmetic, hierarchical */
        } else if (m == 0xC4) {
            size_t o = 0;
            while (o + 17 <= sl) {
                const int tc = s[o] >> 4, th = s[o] & 15;
                if (tc > 1 || th > 3) return GJ_EFORMAT;
                HuffTab* t = tc ? &j->ac[th] : &j->dc[th];
                int cnt = 0;
                t->bits[0] = 0;
                for (int l = 1; l <= 16; ++l) { t->bits[l] = s[o + l]; cnt += s[o + l]; }
                if (cnt > 256 || o + 17 + cnt > sl) return GJ_EFORMAT;
                memcpy(t->vals, s + o + 17, cnt);
                t->present = 1;
                build_huff(t);
                o += 17 + cnt;
            }
        } else if (m == 0xDB) {
            size_t o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                if (tq > 3) return GJ_EFORMAT;
                if (pq == 0) {
                    if (o + 65 > sl) return GJ_EFORMAT;
                    for (int k = 0; k < 64; ++k) j->q[tq][kNatural[k]] = s[o + 1 + k];
                    o += 65;
                } else {
                    if (o + 129 > sl) return GJ_EFORMAT;
                    for (int k = 0; k < 64; ++k) j->q[tq][kNatural[k]] = (uint16_t)((s[o + 1 + 2 * k] << 8) | s[o + 2 + 2 * k]);
                    o += 129;
                }
                j->qpresent[tq] = 1;
            }
        } else if (m == 0xDD) {
            if (sl < 2) return GJ_EFORMAT;
            j->restart_interval = (s[0] << 8) | s[1];
        } else if (m == 0xEE) {
            if (sl >= 12 && !memcmp(s, "Adobe", 5)) adobe = s[11];
        } else if (m == 0xDA) {
            if (!have_sof) return GJ_EFORMAT;
            if (sl < 1 || s[0] != j->ncomp || sl < 1 + 2 * (size_t)j->ncomp + 3) return GJ_EUNSUPPORTED;   /* one interleaved scan */
            for (int c = 0; c < j->ncomp; ++c) {
                j->td[c] = s[2 + 2 * c] >> 4;
                j->ta[c] = s[2 + 2 * c] & 15;
                if (j->td[c] > 3 || j->ta[c] > 3) return GJ_EFORMAT;
            }
            j->scan = p + i + len;
            j->scan_len = n - (i + len);
            break;
        }
        i += len;
    }
    if (!j->scan) return GJ_EFORMAT;
    j->hmax = j->vmax = 1;
    for (int c = 0; c < j->ncomp; ++c) {
        if (j->hs[c] < 1 || j->vs[c] < 1) return GJ_EFORMAT;
        if (j->hs[c] > j->hmax) j->hmax = j->hs[c];
        if (j->vs[c] > j->vmax) j->vmax = j->vs[c];
        if (!j->qpresent[j->tq[c]] || !j->dc[j->td[c]].present || !j->ac[j->ta[c]].present) return GJ_EFORMAT;
    }
    if (j->ncomp == 3) {
        if (j->hs[1] != 1 || j->vs[1] != 1 || j->hs[2] != 1 || j->vs[2] != 1 || j->hs[0] > 2 || j->vs[0] > 2) return GJ_EUNSUPPORTED;
        if (j->hs[0] == 1 && j->vs[0] == 2) return GJ_EUNSUPPORTED;   /* 4:4:0: Pillow cannot write it, so it cannot be pinned here */
        if (adobe == 0) j->rgb = 1;
        else if (adobe == 2) return GJ_EUNSUPPORTED;
    } else if (j->hs[0] != 1 || j->vs[0] != 1) {
        j->hs[0] = j->vs[0] = j->hmax = j->vmax = 1;    /* a single-component scan is never interleaved: sampling factors are moot */
    }
    return GJ_OK;
}

/* ---- entropy decoder ------------------------------------------------------------------------------------------ */
typedef struct {
    const uint8_t* p;
    size_t n, pos;
    uint64_t acc;   /* bits, MSB first, in the low `cnt` bits */
    int cnt;
    int hit_marker; /* a marker was met: zero bits are fed from here on (libjpeg's "insufficient data" behaviour) */
} Bits;

static void fill(Bits* b) {
    while (b->cnt <= 48) {
        int byte = 0;
        if (!b->hit_marker && b->pos < b->n) {
            byte = b->p[b->pos];
            if (byte == 0xFF) {
                size_t q = b->pos + 1;
                while (q < b->n && b->p[q] == 0xFF) ++q;      /* fill bytes */
                if (q < b->n && b->p[q] == 0x00) {
                    b->pos = q + 1;                            /* stuffed zero: a data byte 0xFF */
                } else {
                    b->hit_marker = 1;                         /* RSTn / EOI / anything: stays unread */
                    byte = 0;
                }
            } else {
                b->pos++;
            }
        } else if (b->pos >= b->n) {
            b->hit_marker = 1;
        }
        b->acc = (b->acc << 8) | (uint64_t)byte;
        b->cnt += 8;
    }
}
static inline int getbits(Bits* b, int s) {
    if (b->cnt < s) fill(b);
    b->cnt -= s;
    return (int)((b->acc >> b->cnt) & ((1u << s) - 1));
}
static inline int huff_decode(Bits* b, const HuffTab* t) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | getbits(b, 1);
        if (code <= t->maxcode[l]) return t->vals[(code + t->valoff[l]) & 255];
    }
    return 0; /* corrupt: libjpeg warns and returns 0 */
}
static inline int extend(int x, int s) { return x < (1 << (s - 1)) ? x + (int)((~0u) << s) + 1 : x; }

static void restart(Bits* b) {
    /* discard the partial byte, find the RSTn marker, step over it */
    b->cnt = 0;
    b->acc = 0;
    size_t q = b->pos;
    while (q + 1 < b->n && !(b->p[q] == 0xFF && b->p[q + 1] >= 0xD0 && b->p[q + 1] <= 0xD7)) ++q;
    if (q + 1 < b->n) b->pos = q + 2;
    b->hit_marker = 0;
}

/* ---- jidctint.c jpeg_idct_islow ----------------------------------------------------------------------------------- */
#define CONST_BITS 13
#define PASS1_BITS 2
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))
static inline uint8_t range_limit(int v) {   /* post-IDCT table: index (v & 1023), centred on 128 */
    const int i = v & 1023;
    if (i < 128) return (uint8_t)(i + 128);
    if (i < 512) return 255;
    if (i < 896) return 0;
    return (uint8_t)(i - 896);
}
static void idct_islow(const int16_t* coef, const uint16_t* q, uint8_t* out, int stride) {
    int32_t ws[64];
    for (int c = 0; c < 8; ++c) {
        int32_t z2 = coef[16 + c] * q[16 + c], z3 = coef[48 + c] * q[48 + c];
        int32_t z1 = (z2 + z3) * 4433;
        int32_t tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
        z2 = coef[c] * q[c];
        z3 = coef[32 + c] * q[32 + c];
        int32_t tmp0 = (int32_t)((uint32_t)(z2 + z3) << CONST_BITS), tmp1 = (int32_t)((uint32_t)(z2 - z3) << CONST_BITS);
        const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = coef[56 + c] * q[56 + c];
        tmp1 = coef[40 + c] * q[40 + c];
        tmp2 = coef[24 + c] * q[24 + c];
        tmp3 = coef[8 + c] * q[8 + c];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int32_t z4 = tmp1 + tmp3;
        const int32_t z5 = (z3 + z4) * 9633;
        tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
        z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        ws[c] = DESCALE(tmp10 + tmp3, CONST_BITS - PASS1_BITS);
        ws[56 + c] = DESCALE(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
        ws[8 + c] = DESCALE(tmp11 + tmp2, CONST_BITS - PASS1_BITS);
        ws[48 + c] = DESCALE(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
        ws[16 + c] = DESCALE(tmp12 + tmp1, CONST_BITS - PASS1_BITS);
        ws[40 + c] = DESCALE(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
        ws[24 + c] = DESCALE(tmp13 + tmp0, CONST_BITS - PASS1_BITS);
        ws[32 + c] = DESCALE(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
    }
    for (int r = 0; r < 8; ++r) {
        const int32_t* w = ws + 8 * r;
        int32_t z2 = w[2], z3 = w[6];
        int32_t z1 = (z2 + z3) * 4433;
        int32_t tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
        int32_t tmp0 = (int32_t)((uint32_t)(w[0] + w[4]) << CONST_BITS), tmp1 = (int32_t)((uint32_t)(w[0] - w[4]) << CONST_BITS);
        const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int32_t z4 = tmp1 + tmp3;
        const int32_t z5 = (z3 + z4) * 9633;
        tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
        z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        uint8_t* o = out + (size_t)r * stride;
        const int sh = CONST_BITS + PASS1_BITS + 3;
        o[0] = range_limit(DESCALE(tmp10 + tmp3, sh));
        o[7] = range_limit(DESCALE(tmp10 - tmp3, sh));
        o[1] = range_limit(DESCALE(tmp11 + tmp2, sh));
        o[6] = range_limit(DESCALE(tmp11 - tmp2, sh));
        o[2] = range_limit(DESCALE(tmp12 + tmp1, sh));
        o[5] = range_limit(DESCALE(tmp12 - tmp1, sh));
        o[3] = range_limit(DESCALE(tmp13 + tmp0, sh));
        o[4] = range_limit(DESCALE(tmp13 - tmp0, sh));
    }
}

/* ---- upsampling (jdsample.c) --------------------------------------------------------------------------------------- */
/* one output row `oy` (0 .. 2*ch-1 for v2, 0 .. ch-1 for v1) of a chroma plane [ch][stride] with cw real columns */
static void up_row(const uint8_t* plane, int stride, int cw, int ch, int h2, int v2, int oy, uint8_t* out /* [h2 ? 2*cw : cw] */) {
    const int fancy = cw > 2;         /* jdsample.c: `do_fancy && compptr->downsampled_width > 2` */
    if (!v2) {
        const uint8_t* in = plane + (size_t)oy * stride;
        if (!h2) { memcpy(out, in, cw); return; }
        if (!fancy) { for (int x = 0; x < cw; ++x) out[2 * x] = out[2 * x + 1] = in[x]; return; }
        out[0] = in[0];
        out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        for (int x = 1; x < cw - 1; ++x) {
            const int v = in[x] * 3;
            out[2 * x] = (uint8_t)((v + in[x - 1] + 1) >> 2);
            out[2 * x + 1] = (uint8_t)((v + in[x + 1] + 2) >> 2);
        }
        out[2 * cw - 2] = (uint8_t)((in[cw - 1] * 3 + in[cw - 2] + 1) >> 2);
        out[2 * cw - 1] = in[cw - 1];
        return;
    }
    const int iy = oy >> 1, v = oy & 1;
    int ny = v ? iy + 1 : iy - 1;                      /* next-nearest row: above for the upper output row, below for the lower */
    if (ny < 0) ny = 0;                                /* jdmainct.c: context rows beyond the image replicate the edge row */
    if (ny > ch - 1) ny = ch - 1;
    const uint8_t* in0 = plane + (size_t)iy * stride;
    const uint8_t* in1 = plane + (size_t)ny * stride;
    if (!fancy) {                                      /* h2v2_upsample / h1v2 replication */
        if (h2) for (int x = 0; x < cw; ++x) out[2 * x] = out[2 * x + 1] = in0[x];
        else memcpy(out, in0, cw);
        return;
    }
    if (!h2) {                                         /* h1v2_fancy_upsample */
        const int bias = v ? 2 : 1;
        for (int x = 0; x < cw; ++x) out[x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
        return;
    }
    int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
    out[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
    out[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
    lastcol = thiscol; thiscol = nextcol;
    for (int x = 1; x < cw - 1; ++x) {
        nextcol = in0[x + 1] * 3 + in1[x + 1];
        out[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
        out[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
        lastcol = thiscol; thiscol = nextcol;
    }
    out[2 * cw - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
    out[2 * cw - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
}

static inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* ---- public ------------------------------------------------------------------------------------------------------- */
int grl_oracle_jpeg_info(const uint8_t* data, size_t len, int* width, int* height, int* ncomp, int* hsamp, int* vsamp) {
    Jpeg j;
    const int rc = parse(data, len, &j);
    if (rc) return rc;
    *width = j.width; *height = j.height; *ncomp = j.ncomp; *hsamp = j.hmax; *vsamp = j.vmax;
    return GJ_OK;
}

static int decode_impl(const uint8_t* data, size_t len, uint8_t* out, int16_t* coef_out);

/* out: RGB, interleaved [height][width][3] (what np.asarray(Image.open(..).convert('RGB')) holds) */
int grl_oracle_jpeg_decode(const uint8_t* data, size_t len, uint8_t* out) { return decode_impl(data, len, out, 0); }

/* the quantised coefficients as the entropy decoder leaves them: int16 [blocks in scan order][64 natural order]
 * (blocks = MCUs x blocks per MCU).  Pins the entropy stage of the device decoder on its own. */
int grl_oracle_jpeg_coefficients(const uint8_t* data, size_t len, int16_t* coef_out) { return decode_impl(data, len, 0, coef_out); }

int grl_oracle_jpeg_blocks(const uint8_t* data, size_t len) {
    Jpeg j;
    if (parse(data, len, &j)) return -1;
    int bpm = 0;
    for (int c = 0; c < j.ncomp; ++c) bpm += j.hs[c] * j.vs[c];
    return ((j.width + 8 * j.hmax - 1) / (8 * j.hmax)) * ((j.height + 8 * j.vmax - 1) / (8 * j.vmax)) * bpm;
}

static int decode_impl(const uint8_t* data, size_t len, uint8_t* out, int16_t* coef_out) {
    Jpeg j;
    int rc = parse(data, len, &j);
    if (rc) return rc;
    const int mcux = (j.width + 8 * j.hmax - 1) / (8 * j.hmax), mcuy = (j.height + 8 * j.vmax - 1) / (8 * j.vmax);
    uint8_t* plane[3] = {0, 0, 0};
    int pw[3], ph[3], cw[3], chh[3];
    for (int c = 0; c < j.ncomp; ++c) {
        pw[c] = mcux * j.hs[c] * 8;
        ph[c] = mcuy * j.vs[c] * 8;
        cw[c] = (j.width * j.hs[c] + j.hmax - 1) / j.hmax;       /* downsampled_width / height: the real samples */
        chh[c] = (j.height * j.vs[c] + j.vmax - 1) / j.vmax;
        plane[c] = (uint8_t*)malloc((size_t)pw[c] * ph[c]);
        if (!plane[c]) { rc = GJ_ENOMEM; goto done; }
    }
    {
        Bits b = {j.scan, j.scan_len, 0, 0, 0, 0};
        int pred[3] = {0, 0, 0};
        int left = j.restart_interval;
        int16_t coef[64];
        for (int my = 0; my < mcuy; ++my)
            for (int mx = 0; mx < mcux; ++mx) {
                if (j.restart_interval) {
                    if (left == 0) {
                        restart(&b);
                        pred[0] = pred[1] = pred[2] = 0;
                        left = j.restart_interval;
                    }
                    --left;
                }
                for (int c = 0; c < j.ncomp; ++c)
                    for (int by = 0; by < j.vs[c]; ++by)
                        for (int bx = 0; bx < j.hs[c]; ++bx) {
                            memset(coef, 0, sizeof(coef));
                            int s = huff_decode(&b, &j.dc[j.td[c]]);
                            if (s) { const int r = getbits(&b, s); s = extend(r, s); }
                            pred[c] += s;
                            coef[0] = (int16_t)pred[c];
                            for (int k = 1; k < 64; ++k) {
                                const int rs = huff_decode(&b, &j.ac[j.ta[c]]);
                                const int r = rs >> 4, sz = rs & 15;
                                if (sz) {
                                    k += r;
                                    const int v = extend(getbits(&b, sz), sz);
                                    coef[kNatural[k]] = (int16_t)v;
                                } else {
                                    if (r != 15) break;
                                    k += 15;
                                }
                            }
                            if (coef_out) { memcpy(coef_out, coef, sizeof(coef)); coef_out += 64; }
                            idct_islow(coef, j.q[j.tq[c]],
                                       plane[c] + (size_t)((my * j.vs[c] + by) * 8) * pw[c] + (mx * j.hs[c] + bx) * 8, pw[c]);
                        }
            }
    }
    if (!out) goto done;
    if (j.ncomp == 1) {
        for (int y = 0; y < j.height; ++y)
            for (int x = 0; x < j.width; ++x) {
                const uint8_t v = plane[0][(size_t)y * pw[0] + x];
                uint8_t* o = out + ((size_t)y * j.width + x) * 3;
                o[0] = o[1] = o[2] = v;
            }
    } else {
        const int h2 = j.hmax == 2, v2 = j.vmax == 2;
        const int upw = h2 ? 2 * cw[1] : cw[1];
        uint8_t* cbrow = (uint8_t*)malloc((size_t)upw + 8);
        uint8_t* crrow = (uint8_t*)malloc((size_t)upw + 8);
        if (!cbrow || !crrow) { free(cbrow); free(crrow); rc = GJ_ENOMEM; goto done; }
        for (int y = 0; y < j.height; ++y) {
            up_row(plane[1], pw[1], cw[1], chh[1], h2, v2, y, cbrow);
            up_row(plane[2], pw[2], cw[2], chh[2], h2, v2, y, crrow);
            for (int x = 0; x < j.width; ++x) {
                const int Y = plane[0][(size_t)y * pw[0] + x], cb = cbrow[x] - 128, cr = crrow[x] - 128;
                uint8_t* o = out + ((size_t)y * j.width + x) * 3;
                if (j.rgb) { o[0] = (uint8_t)Y; o[1] = cbrow[x]; o[2] = crrow[x]; continue; }
                /* jdcolor.c build_ycc_rgb_table: FIX(x) = (int)(x * 65536 + 0.5), ONE_HALF = 32768, arithmetic shifts */
                const int r = Y + ((91881 * cr + 32768) >> 16);
                const int g = Y + ((-22554 * cb - 46802 * cr + 32768) >> 16);
                const int bl = Y + ((116130 * cb + 32768) >> 16);
                o[0] = clamp8(r); o[1] = clamp8(g); o[2] = clamp8(bl);
            }
        }
        free(cbrow); free(crrow);
    }
done:
    for (int c = 0; c < 3; ++c) free(plane[c]);
    return rc;
}
