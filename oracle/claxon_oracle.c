/*
 * claxon_oracle.c -- CPU restatement of ruuda/claxon v0.4.3's frame decode path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check
 * in __graft_entry__.py and bench.py's `cpu_baseline` leg may load it.  The
 * shipped decoder (claxon_amd/, include/claxon_hip.h) never links, imports or
 * calls anything in this directory and has no CPU fallback.
 *
 * What it is: a plain-C restatement of the reference's Rust, function for
 * function, keeping the reference's structure (byte-at-a-time reader with a
 * 1-byte bit cache, CRC folded in per byte, i64 LPC accumulation, the 12-tap
 * zero-padded low-order LPC loop) so that it is an honest stand-in for
 * "Claxon's CPU path" when timed.  The real reference cannot be built here
 * (no rustc/cargo in the image, no network), so `oracle/_ref` does not exist;
 * parity is pinned instead by every known-answer vector in the reference's own
 * tests and by the STREAMINFO MD5s of its in-tree fixtures (tests/test_oracle*.py).
 *
 * Each function cites the reference file:line it follows (paths relative to
 * the claxon source tree).
 */
#define _GNU_SOURCE                     /* pthread_setaffinity_np, pthread barriers (clxo_bench_batch only) */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>
#include <sched.h>
#include <time.h>

#include "../include/claxon_hip.h"   /* status / message ids only (the shared contract) */

#define ST_OK CLX_OK

typedef struct { int status; uint32_t msg; } oerr;
static inline oerr ok(void) { oerr e = { CLX_OK, CLX_MSG_NONE }; return e; }
static inline oerr fmt_err(uint32_t m) { oerr e = { CLX_FORMAT_ERROR, m }; return e; }   /* error.rs:100-102 */
static inline oerr unsup(uint32_t m) { oerr e = { CLX_UNSUPPORTED, m }; return e; }
static inline oerr io_eof(void) { oerr e = { CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF }; return e; }
#define TRY(x) do { oerr _e = (x); if (_e.status != CLX_OK) return _e; } while (0)

/* ---------------------------------------------------------------- crc.rs */

static uint8_t  g_crc8_table[256];
static uint16_t g_crc16_table[256];
static pthread_once_t g_crc_once = PTHREAD_ONCE_INIT;

/* crc.rs:13-57 holds literal tables (taken from libFLAC).  They are the
 * standard MSB-first tables of x^8+x^2+x+1 and x^16+x^15+x^2+1 (crc.rs:61,69),
 * regenerated here from the polynomials; tests pin them with crc.rs:197-209. */
static void crc_tables_init(void) {
    for (int i = 0; i < 256; i++) {
        uint8_t c8 = (uint8_t)i;
        uint16_t c16 = (uint16_t)(i << 8);
        for (int b = 0; b < 8; b++) {
            c8 = (uint8_t)((c8 & 0x80) ? ((c8 << 1) ^ 0x07) : (c8 << 1));
            c16 = (uint16_t)((c16 & 0x8000) ? ((c16 << 1) ^ 0x8005) : (c16 << 1));
        }
        g_crc8_table[i] = c8;
        g_crc16_table[i] = c16;
    }
}

/* The reader stack of frame.rs:673,702,135:
 *   Bitstream<&mut Crc16Reader<&mut Cursor>> with a Crc8Reader stacked on top
 * during the frame header.  One struct with both CRC states; `crc8_on`
 * says whether the Crc8Reader layer is currently present. */
typedef struct {
    const uint8_t* buf;
    size_t len;
    size_t pos;
    uint16_t crc16;   /* Crc16Reader.state, crc.rs:72 */
    uint8_t crc8;     /* Crc8Reader.state, crc.rs:64 */
    int crc8_on;
} reader;

/* io::Cursor read_u8 (input.rs:236-244) + Crc16Reader::read_u8 (crc.rs:149-157,
 * update crc.rs:109-112) + Crc8Reader::read_u8 (crc.rs:117-125, update 90-92). */
static inline int rd_u8(reader* r, uint8_t* out) {
    if (r->pos >= r->len) return 0;
    uint8_t b = r->buf[r->pos++];
    r->crc16 = (uint16_t)((r->crc16 << 8) ^ g_crc16_table[(uint8_t)(r->crc16 >> 8) ^ b]);
    if (r->crc8_on) r->crc8 = g_crc8_table[r->crc8 ^ b];
    *out = b;
    return 1;
}

/* input.rs:87-91 */
static inline int rd_be_u16(reader* r, uint16_t* out) {
    uint8_t b0, b1;
    if (!rd_u8(r, &b0)) return 0;
    if (!rd_u8(r, &b1)) return 0;
    *out = (uint16_t)((b0 << 8) | b1);
    return 1;
}

/* ------------------------------------------------------------- input.rs Bitstream */

typedef struct {          /* input.rs:415-422 */
    reader* rd;
    uint8_t data;
    uint32_t bits_left;
} bitstream;

static inline void bs_new(bitstream* bs, reader* r) { bs->rd = r; bs->data = 0; bs->bits_left = 0; }   /* input.rs:426-432 */

static inline uint8_t shift_left_u8(uint8_t x, uint32_t s) { return (uint8_t)(((uint32_t)x) << s); }   /* input.rs:395-402 */
static inline uint8_t shift_right_u8(uint8_t x, uint32_t s) { return (uint8_t)(((uint32_t)x) >> s); }  /* input.rs:405-412 */
static inline uint8_t mask_u8(uint32_t bits) { return shift_left_u8(0xff, 8 - bits); }                 /* input.rs:435-440 */
static inline uint32_t lz8(uint8_t x) { return x ? (uint32_t)__builtin_clz((uint32_t)x) - 24u : 8u; }  /* u8::leading_zeros */

/* input.rs:447-468 */
static inline oerr bs_read_bit(bitstream* bs, int* out) {
    uint8_t result;
    if (bs->bits_left == 0) {
        uint8_t fresh;
        if (!rd_u8(bs->rd, &fresh)) return io_eof();
        bs->data = (uint8_t)(fresh << 1);
        bs->bits_left = 7;
        result = fresh & 0x80;
    } else {
        result = bs->data & 0x80;
        bs->data = (uint8_t)(bs->data << 1);
        bs->bits_left -= 1;
    }
    *out = result != 0;
    return ok();
}

/* input.rs:475-511 */
static inline oerr bs_read_unary(bitstream* bs, uint32_t* out) {
    uint32_t n = lz8(bs->data);
    if (n < bs->bits_left) {
        bs->data = (uint8_t)(bs->data << (n + 1));
        bs->bits_left -= n + 1;
    } else {
        n = bs->bits_left;
        for (;;) {
            uint8_t fresh;
            if (!rd_u8(bs->rd, &fresh)) return io_eof();
            uint32_t zeros = lz8(fresh);
            n += zeros;
            if (zeros < 8) {
                bs->bits_left = 8 - (zeros + 1);
                bs->data = shift_left_u8(fresh, zeros + 1);
                break;
            }
        }
    }
    *out = n;
    return ok();
}

/* input.rs:515-558 */
static inline oerr bs_read_leq_u8(bitstream* bs, uint32_t bits, uint8_t* out) {
    uint8_t result;
    if (bs->bits_left < bits) {
        uint8_t msb = bs->data;
        if (!rd_u8(bs->rd, &bs->data)) return io_eof();
        uint8_t lsb = (uint8_t)((bs->data & mask_u8(bits - bs->bits_left)) >> bs->bits_left);
        bs->data = shift_left_u8(bs->data, bits - bs->bits_left);
        bs->bits_left = 8 - (bits - bs->bits_left);
        result = msb | lsb;
    } else {
        result = bs->data & mask_u8(bits);
        bs->data = shift_left_u8(bs->data, bits);   /* `self.data << bits` with bits <= 8 on the u8 */
        bs->bits_left -= bits;
    }
    *out = shift_right_u8(result, 8 - bits);
    return ok();
}

/* input.rs:562-602 (8 < bits <= 16) */
static inline oerr bs_read_gt_u8_leq_u16(bitstream* bs, uint32_t bits, uint32_t* out) {
    uint32_t mask_msb = 0xffffffffu << (bits - bs->bits_left);
    uint32_t msb = (((uint32_t)bs->data) << (bits - 8)) & mask_msb;
    uint32_t bits_to_read = bits - bs->bits_left;
    uint8_t fb;
    if (!rd_u8(bs->rd, &fb)) return io_eof();
    uint32_t fresh = fb;
    uint32_t lsb = (bits_to_read >= 8) ? (fresh << (bits_to_read - 8)) : (fresh >> (8 - bits_to_read));
    uint32_t combined = msb | lsb;
    uint32_t result;
    if (bits_to_read <= 8) {
        bs->bits_left = 8 - bits_to_read;
        bs->data = (uint8_t)(fresh << ((8 - bs->bits_left) & 31));    /* wrapping_shl */
        result = combined;
    } else {
        uint8_t fb2;
        if (!rd_u8(bs->rd, &fb2)) return io_eof();
        uint32_t fresher = fb2;
        uint32_t lsb2 = fresher >> (16 - bits_to_read);
        bs->bits_left = 16 - bits_to_read;
        bs->data = (uint8_t)(fresher << ((8 - bs->bits_left) & 31));
        result = combined | lsb2;
    }
    *out = result;
    return ok();
}

/* input.rs:606-622 */
static inline oerr bs_read_leq_u16(bitstream* bs, uint32_t bits, uint16_t* out) {
    if (bits <= 8) {
        uint8_t v;
        TRY(bs_read_leq_u8(bs, bits, &v));
        *out = v;
    } else {
        uint8_t m, l;
        TRY(bs_read_leq_u8(bs, 8, &m));
        TRY(bs_read_leq_u8(bs, bits - 8, &l));
        *out = (uint16_t)((((uint16_t)m) << (bits - 8)) | l);
    }
    return ok();
}

/* input.rs:626-642 */
static inline oerr bs_read_leq_u32(bitstream* bs, uint32_t bits, uint32_t* out) {
    if (bits <= 16) {
        uint16_t v;
        TRY(bs_read_leq_u16(bs, bits, &v));
        *out = v;
    } else {
        uint16_t m, l;
        TRY(bs_read_leq_u16(bs, 16, &m));
        TRY(bs_read_leq_u16(bs, bits - 16, &l));
        *out = (((uint32_t)m) << (bits - 16)) | l;
    }
    return ok();
}

/* ------------------------------------------------------------- subframe.rs */

/* subframe.rs:96-101 */
static inline int16_t extend_sign_u16(uint16_t val, uint32_t bits) {
    return (int16_t)(((int16_t)(uint16_t)(val << (16 - bits))) >> (16 - bits));
}
/* subframe.rs:117-122 */
static inline int32_t extend_sign_u32(uint32_t val, uint32_t bits) {
    return ((int32_t)(val << (32 - bits))) >> (32 - bits);
}
/* subframe.rs:157-170 */
static inline int32_t rice_to_signed(uint32_t val) {
    int32_t half = (int32_t)(val >> 1);
    int32_t ext = ((int32_t)(val << 31)) >> 31;
    return half ^ ext;
}

enum { SF_CONSTANT, SF_VERBATIM, SF_FIXED, SF_LPC };
typedef struct { int type; uint32_t order; uint32_t wasted; } sf_header;

/* subframe.rs:29-91 */
static oerr read_subframe_header(bitstream* in, sf_header* h) {
    int bit;
    TRY(bs_read_bit(in, &bit));
    if (bit) return fmt_err(CLX_MSG_SUBFRAME_HEADER_INVALID);
    uint8_t n;
    TRY(bs_read_leq_u8(in, 6, &n));
    if (n == 0) { h->type = SF_CONSTANT; h->order = 0; }
    else if (n == 1) { h->type = SF_VERBATIM; h->order = 0; }
    else if (((n & 0x3e) == 0x02) || ((n & 0x3c) == 0x04) || ((n & 0x30) == 0x10)) {
        return fmt_err(CLX_MSG_SUBFRAME_HEADER_RESERVED);
    } else if ((n & 0x38) == 0x08) {
        uint32_t order = n & 0x07;
        if (order > 4) return fmt_err(CLX_MSG_SUBFRAME_HEADER_RESERVED);
        h->type = SF_FIXED; h->order = order;
    } else {
        h->type = SF_LPC; h->order = (uint32_t)(n & 0x1f) + 1;
    }
    int wastes;
    TRY(bs_read_bit(in, &wastes));
    uint32_t wasted = 0;
    if (wastes) {
        uint32_t u;
        TRY(bs_read_unary(in, &u));
        wasted = 1 + u;
    }
    if (wasted > 31) return fmt_err(CLX_MSG_WASTED_BITS_EXCEED_31);
    h->wasted = wasted;
    return ok();
}

/* subframe.rs:310-351 */
static inline oerr decode_rice_partition(bitstream* in, int32_t* buf, size_t n) {
    uint8_t p;
    TRY(bs_read_leq_u8(in, 4, &p));
    uint32_t rice_param = p;
    if (rice_param == 15) return unsup(CLX_MSG_UNENCODED_BINARY);
    if (rice_param <= 8) {
        for (size_t i = 0; i < n; i++) {
            uint32_t q; uint8_t r;
            TRY(bs_read_unary(in, &q));
            TRY(bs_read_leq_u8(in, rice_param, &r));
            buf[i] = rice_to_signed((q << rice_param) | (uint32_t)r);
        }
    } else {
        for (size_t i = 0; i < n; i++) {
            uint32_t q, r;
            TRY(bs_read_unary(in, &q));
            TRY(bs_read_gt_u8_leq_u16(in, rice_param, &r));
            buf[i] = rice_to_signed((q << rice_param) | r);
        }
    }
    return ok();
}

/* subframe.rs:358-380 */
static oerr decode_rice2_partition(bitstream* in, int32_t* buf, size_t n) {
    uint8_t p;
    TRY(bs_read_leq_u8(in, 5, &p));
    uint32_t rice_param = p;
    if (rice_param == 31) return unsup(CLX_MSG_UNENCODED_BINARY);
    for (size_t i = 0; i < n; i++) {
        uint32_t q, r;
        TRY(bs_read_unary(in, &q));
        TRY(bs_read_leq_u32(in, rice_param, &r));
        /* `q << rice_param` on u32: the reference is built in release mode
         * (wrapping shift semantics for in-range shift amounts, rice_param <= 30). */
        buf[i] = rice_to_signed((q << rice_param) | r);
    }
    return ok();
}

/* subframe.rs:236-304; `buf` is buffer[order..], `block_size` the full block (as u16) */
static oerr decode_residual(bitstream* in, uint16_t block_size, int32_t* buf, size_t buf_len) {
    uint8_t method;
    TRY(bs_read_leq_u8(in, 2, &method));
    if (method > 1) return fmt_err(CLX_MSG_RESIDUAL_RESERVED);
    uint8_t order;
    TRY(bs_read_leq_u8(in, 4, &order));
    uint32_t n_partitions = 1u << order;
    uint16_t n_per = (uint16_t)(block_size >> order);
    if ((block_size & (uint16_t)(n_partitions - 1)) != 0) return fmt_err(CLX_MSG_INVALID_PARTITION_ORDER);
    uint16_t n_warm_up = (uint16_t)(block_size - (uint16_t)buf_len);
    if (n_warm_up > n_per) return fmt_err(CLX_MSG_INVALID_RESIDUAL);
    size_t start = 0;
    uint16_t len = (uint16_t)(n_per - n_warm_up);
    for (uint32_t p = 0; p < n_partitions; p++) {
        if (method == 0) TRY(decode_rice_partition(in, buf + start, len));
        else             TRY(decode_rice2_partition(in, buf + start, len));
        start += len;
        len = n_per;
    }
    return ok();
}

/* subframe.rs:382-394 */
static oerr decode_constant(bitstream* in, uint32_t bps, int32_t* buf, size_t n) {
    uint32_t v;
    TRY(bs_read_leq_u32(in, bps, &v));
    int32_t s = extend_sign_u32(v, bps);
    for (size_t i = 0; i < n; i++) buf[i] = s;
    return ok();
}

/* subframe.rs:397-415 */
static oerr decode_verbatim(bitstream* in, uint32_t bps, int32_t* buf, size_t n) {
    for (size_t i = 0; i < n; i++) {
        uint32_t v;
        TRY(bs_read_leq_u32(in, bps, &v));
        buf[i] = extend_sign_u32(v, bps);
    }
    return ok();
}

/* subframe.rs:417-474.  Wrapping i32 arithmetic throughout (461-470). */
static void predict_fixed(uint32_t order, int32_t* buf, size_t n) {
    static const int32_t o1[] = { 1 };
    static const int32_t o2[] = { -1, 2 };
    static const int32_t o3[] = { 1, -3, 3 };
    static const int32_t o4[] = { -1, 4, -6, 4 };
    const int32_t* coef = NULL;
    switch (order) { case 1: coef = o1; break; case 2: coef = o2; break; case 3: coef = o3; break; case 4: coef = o4; break; default: break; }
    for (size_t i = 0; i + order < n; i++) {
        uint32_t pred = 0;
        for (uint32_t j = 0; j < order; j++) pred += (uint32_t)coef[j] * (uint32_t)buf[i + j];
        buf[i + order] = (int32_t)(pred + (uint32_t)buf[i + order]);
    }
}

/* subframe.rs:492-516 */
static oerr decode_fixed(bitstream* in, uint32_t bps, uint32_t order, int32_t* buf, size_t n) {
    if (n < order) return fmt_err(CLX_MSG_FIXED_ORDER_GT_BLOCK);
    TRY(decode_verbatim(in, bps, buf, order));
    TRY(decode_residual(in, (uint16_t)n, buf + order, n - order));
    predict_fixed(order, buf, n);
    return ok();
}

/* subframe.rs:524-583.  raw[] is in application order: raw[j] multiplies buf[i-order+j]. */
static void predict_lpc_low_order(const int16_t* raw, size_t order, int16_t qlp_shift, int32_t* buf, size_t n) {
    int64_t coef[12];
    for (int i = 0; i < 12; i++) coef[i] = 0;
    for (size_t i = 0; i < order; i++) coef[12 - order + i] = raw[i];

    size_t left = (n < 12 ? n : 12) - order;
    for (size_t i = 0; i < left; i++) {
        int64_t sum = 0;
        for (size_t j = 0; j < order; j++) sum += (int64_t)raw[j] * (int64_t)buf[i + j];
        int64_t pred = sum >> qlp_shift;
        buf[order + i] = (int32_t)(pred + (int64_t)buf[order + i]);
    }
    if (n <= 12) return;
    for (size_t i = 12; i < n; i++) {
        int64_t sum = 0;
        for (int j = 0; j < 12; j++) sum += coef[j] * (int64_t)buf[i - 12 + j];
        int64_t pred = sum >> qlp_shift;
        buf[i] = (int32_t)(pred + (int64_t)buf[i]);
    }
}

/* subframe.rs:586-614 */
static void predict_lpc_high_order(const int16_t* coef, size_t order, int16_t qlp_shift, int32_t* buf, size_t n) {
    for (size_t i = order; i < n; i++) {
        int64_t sum = 0;
        for (size_t j = 0; j < order; j++) sum += (int64_t)coef[j] * (int64_t)buf[i - order + j];
        int64_t pred = sum >> qlp_shift;
        buf[i] = (int32_t)(pred + (int64_t)buf[i]);
    }
}

/* subframe.rs:651-721 */
static oerr decode_lpc(bitstream* in, uint32_t bps, uint32_t order, int32_t* buf, size_t n) {
    if (n < order) return fmt_err(CLX_MSG_LPC_ORDER_GT_BLOCK);
    TRY(decode_verbatim(in, bps, buf, order));
    uint8_t pm1;
    TRY(bs_read_leq_u8(in, 4, &pm1));
    uint32_t qlp_precision = (uint32_t)pm1 + 1;
    if (qlp_precision - 1 == 15) return fmt_err(CLX_MSG_QLP_PRECISION_INVALID);
    uint16_t shift_u;
    TRY(bs_read_leq_u16(in, 5, &shift_u));
    int16_t qlp_shift = extend_sign_u16(shift_u, 5);
    if (qlp_shift < 0) return unsup(CLX_MSG_NEGATIVE_QLP_SHIFT);
    int16_t coef[32];
    memset(coef, 0, sizeof coef);
    for (uint32_t k = order; k-- > 0;) {           /* stored reversed, subframe.rs:696-701 */
        uint16_t cu;
        TRY(bs_read_leq_u16(in, qlp_precision, &cu));
        coef[k] = extend_sign_u16(cu, qlp_precision);
    }
    TRY(decode_residual(in, (uint16_t)n, buf + order, n - order));
    if (order <= 12) predict_lpc_low_order(coef, order, qlp_shift, buf, n);
    else             predict_lpc_high_order(coef, order, qlp_shift, buf, n);
    return ok();
}

/* subframe.rs:184-228 */
static oerr subframe_decode(bitstream* in, uint32_t bps, int32_t* buf, size_t n) {
    sf_header h;
    TRY(read_subframe_header(in, &h));
    if (h.wasted >= bps) return fmt_err(CLX_MSG_NO_NON_WASTED_BITS);
    uint32_t sf_bps = bps - h.wasted;
    switch (h.type) {
        case SF_CONSTANT: TRY(decode_constant(in, sf_bps, buf, n)); break;
        case SF_VERBATIM: TRY(decode_verbatim(in, sf_bps, buf, n)); break;
        case SF_FIXED:    TRY(decode_fixed(in, sf_bps, h.order, buf, n)); break;
        default:          TRY(decode_lpc(in, sf_bps, h.order, buf, n)); break;
    }
    if (h.wasted > 0) {
        for (size_t i = 0; i < n; i++) buf[i] = (int32_t)(((uint32_t)buf[i]) << (h.wasted & 31));   /* wrapping_shl */
    }
    return ok();
}

/* ------------------------------------------------------------- frame.rs */

/* frame.rs:64-105 */
static oerr read_var_length_int(reader* in, uint64_t* out) {
    uint8_t first;
    if (!rd_u8(in, &first)) return io_eof();
    uint8_t read_additional = 0, mask_data = 0x7f, mask_mark = 0x80;
    while (first & mask_mark) {
        read_additional++;
        mask_data >>= 1;
        mask_mark >>= 1;
    }
    if (read_additional > 0) {
        if (read_additional == 1) return fmt_err(CLX_MSG_INVALID_VARINT);
        read_additional--;
    }
    uint64_t result = ((uint64_t)(first & mask_data)) << (6 * read_additional);
    for (int i = (int)read_additional - 1; i >= 0; i--) {
        uint8_t b;
        if (!rd_u8(in, &b)) return io_eof();
        if ((b & 0xc0) != 0x80) return fmt_err(CLX_MSG_INVALID_VARINT);
        result |= ((uint64_t)(b & 0x3f)) << (6 * i);
    }
    *out = result;
    return ok();
}

typedef struct {
    int variable;            /* BlockingStrategy */
    uint64_t number;         /* frame number or sample number */
    uint16_t block_size;
    uint32_t sample_rate;    /* 0 = None */
    int channel_assignment;  /* CLX_CH_* */
    uint8_t n_channels;
    uint32_t bps;            /* 0 = None */
} frame_header;

/* frame.rs:131-316.  *eof=1 <=> Ok(None). */
static oerr read_frame_header_or_eof(reader* in, frame_header* h, int* eof, int check_crc) {
    *eof = 0;
    in->crc8 = 0; in->crc8_on = 1;                      /* Crc8Reader::new, frame.rs:135 */
    /* read_be_u16_or_eof, input.rs:94-101: EOF on either of the two bytes is Ok(None) */
    uint8_t b0, b1;
    if (!rd_u8(in, &b0) || !rd_u8(in, &b1)) { in->crc8_on = 0; *eof = 1; return ok(); }
    uint16_t sync_res_block = (uint16_t)((b0 << 8) | b1);
    oerr e = ok();
    do {
        if ((sync_res_block & 0xfffc) != 0xfff8) { e = fmt_err(CLX_MSG_FRAME_SYNC_MISSING); break; }
        if (sync_res_block & 0x0002) { e = fmt_err(CLX_MSG_FRAME_HEADER_RESERVED); break; }
        h->variable = (sync_res_block & 1) != 0;

        uint8_t bs_sr;
        if (!rd_u8(in, &bs_sr)) { e = io_eof(); break; }
        uint16_t block_size = 0;
        int read_8bit_bs = 0, read_16bit_bs = 0;
        uint8_t n = bs_sr >> 4;
        if (n == 0) { e = fmt_err(CLX_MSG_FRAME_HEADER_RESERVED); break; }
        else if (n == 1) block_size = 192;
        else if (n >= 2 && n <= 5) block_size = (uint16_t)(576u * (1u << (n - 2)));
        else if (n == 6) read_8bit_bs = 1;
        else if (n == 7) read_16bit_bs = 1;
        else block_size = (uint16_t)(256u * (1u << (n - 8)));

        uint32_t sample_rate = 0;
        int read_8bit_sr = 0, read_16bit_sr = 0, read_16bit_sr_ten = 0;
        switch (bs_sr & 0x0f) {
            case 0: sample_rate = 0; break;
            case 1: sample_rate = 88200; break;
            case 2: sample_rate = 176400; break;
            case 3: sample_rate = 192000; break;
            case 4: sample_rate = 8000; break;
            case 5: sample_rate = 16000; break;
            case 6: sample_rate = 22050; break;
            case 7: sample_rate = 24000; break;
            case 8: sample_rate = 32000; break;
            case 9: sample_rate = 44100; break;
            case 10: sample_rate = 48000; break;
            case 11: sample_rate = 96000; break;
            case 12: read_8bit_sr = 1; break;
            case 13: read_16bit_sr = 1; break;
            case 14: read_16bit_sr_ten = 1; break;
            default: e = fmt_err(CLX_MSG_FRAME_HEADER_INVALID); break;
        }
        if (e.status != CLX_OK) break;

        uint8_t chan_bps_res;
        if (!rd_u8(in, &chan_bps_res)) { e = io_eof(); break; }
        uint8_t ca = chan_bps_res >> 4;
        if (ca < 8) { h->channel_assignment = CLX_CH_INDEPENDENT; h->n_channels = (uint8_t)(ca + 1); }
        else if (ca == 8) { h->channel_assignment = CLX_CH_LEFT_SIDE; h->n_channels = 2; }
        else if (ca == 9) { h->channel_assignment = CLX_CH_RIGHT_SIDE; h->n_channels = 2; }
        else if (ca == 10) { h->channel_assignment = CLX_CH_MID_SIDE; h->n_channels = 2; }
        else { e = fmt_err(CLX_MSG_FRAME_HEADER_RESERVED); break; }

        switch ((chan_bps_res & 0x0e) >> 1) {
            case 0: h->bps = 0; break;
            case 1: h->bps = 8; break;
            case 2: h->bps = 12; break;
            case 4: h->bps = 16; break;
            case 5: h->bps = 20; break;
            case 6: h->bps = 24; break;
            default: e = fmt_err(CLX_MSG_FRAME_HEADER_RESERVED); break;
        }
        if (e.status != CLX_OK) break;
        if (chan_bps_res & 1) { e = fmt_err(CLX_MSG_FRAME_HEADER_RESERVED); break; }

        uint64_t num;
        e = read_var_length_int(in, &num);
        if (e.status != CLX_OK) break;
        if (!h->variable && num > 0x7fffffffull) { e = fmt_err(CLX_MSG_FRAME_NUMBER_TOO_LARGE); break; }
        h->number = num;

        if (read_8bit_bs) {
            uint8_t bs;
            if (!rd_u8(in, &bs)) { e = io_eof(); break; }
            block_size = (uint16_t)(bs + 1);
        }
        if (read_16bit_bs) {
            uint16_t bs;
            if (!rd_be_u16(in, &bs)) { e = io_eof(); break; }
            if (bs == 0xffff) { e = fmt_err(CLX_MSG_BLOCK_SIZE_EXCEEDS_65535); break; }
            block_size = (uint16_t)(bs + 1);
        }
        if (read_8bit_sr) {
            uint8_t sr;
            if (!rd_u8(in, &sr)) { e = io_eof(); break; }
            sample_rate = sr;
        }
        if (read_16bit_sr) {
            uint16_t sr;
            if (!rd_be_u16(in, &sr)) { e = io_eof(); break; }
            sample_rate = sr;
        }
        if (read_16bit_sr_ten) {
            uint16_t sr;
            if (!rd_be_u16(in, &sr)) { e = io_eof(); break; }
            sample_rate = (uint32_t)sr * 10;
        }
        uint8_t computed = in->crc8;
        uint8_t presumed;
        if (!rd_u8(in, &presumed)) { e = io_eof(); break; }
        if (check_crc && computed != presumed) { e = fmt_err(CLX_MSG_FRAME_HEADER_CRC_MISMATCH); break; }
        h->block_size = block_size;
        h->sample_rate = sample_rate;
    } while (0);
    in->crc8_on = 0;
    return e;
}

/* frame.rs:319-334 */
static void decode_left_side(int32_t* buf, size_t bs) {
    for (size_t i = 0; i < bs; i++) {
        int32_t left = buf[i], side = buf[bs + i];
        buf[bs + i] = (int32_t)((uint32_t)left - (uint32_t)side);
    }
}
/* frame.rs:345-360 */
static void decode_right_side(int32_t* buf, size_t bs) {
    for (size_t i = 0; i < bs; i++) {
        int32_t side = buf[i], right = buf[bs + i];
        buf[i] = (int32_t)((uint32_t)side + (uint32_t)right);
    }
}
/* frame.rs:371-389.  `/ 2` is Rust's truncating signed division. */
static void decode_mid_side(int32_t* buf, size_t bs) {
    for (size_t i = 0; i < bs; i++) {
        int32_t mid = buf[i], side = buf[bs + i];
        int32_t m = (int32_t)(((uint32_t)mid * 2u) | ((uint32_t)side & 1u));
        int32_t l = (int32_t)((uint32_t)m + (uint32_t)side);
        int32_t r = (int32_t)((uint32_t)m - (uint32_t)side);
        buf[i] = l / 2;
        buf[bs + i] = r / 2;
    }
}

typedef struct clxo_frame_info {
    int32_t  status;
    uint32_t msg;
    uint64_t time;
    uint64_t bytes_consumed;   /* reader position after the call */
    uint64_t end_bit;          /* bit offset (from frame start) just past the last subframe */
    uint32_t block_size;
    uint32_t channels;
    uint32_t bps;
    uint32_t channel_assignment;
    uint32_t sample_rate;
    uint32_t header_bytes;
} clxo_frame_info;

/* FrameReader::read_next_or_eof, frame.rs:667-779, on io::Cursor(buf[..len]).
 * `out` must hold channels*block_size i32 (<= 8*65535); `out_cap` in samples. */
int clxo_frame_decode(const uint8_t* buf, size_t len, int32_t* out, size_t out_cap,
                      int check_crc, clxo_frame_info* info) {
    pthread_once(&g_crc_once, crc_tables_init);
    memset(info, 0, sizeof *info);
    reader rd = { buf, len, 0, 0, 0, 0 };             /* Crc16Reader::new, frame.rs:673 */
    frame_header h;
    memset(&h, 0, sizeof h);
    int eof = 0;
    oerr e = read_frame_header_or_eof(&rd, &h, &eof, check_crc);
    info->bytes_consumed = rd.pos;
    if (e.status != CLX_OK) { info->status = e.status; info->msg = e.msg; return e.status; }
    if (eof) { info->status = CLX_END_OF_STREAM; return CLX_END_OF_STREAM; }
    info->header_bytes = (uint32_t)rd.pos;
    info->block_size = h.block_size;
    info->channels = h.n_channels;
    info->bps = h.bps;
    info->channel_assignment = (uint32_t)h.channel_assignment;
    info->sample_rate = h.sample_rate;
    size_t bs = h.block_size;
    size_t total = (size_t)h.n_channels * bs;
    if (total > out_cap) { info->status = CLX_API_ERROR; return CLX_API_ERROR; }
    if (h.bps == 0) { info->status = CLX_UNSUPPORTED; info->msg = CLX_MSG_NO_BPS_IN_HEADER; return CLX_UNSUPPORTED; }  /* frame.rs:687-692 */
    uint32_t bps = h.bps;
    bitstream bits;
    bs_new(&bits, &rd);
    switch (h.channel_assignment) {                    /* frame.rs:705-742 */
        case CLX_CH_INDEPENDENT:
            for (size_t ch = 0; ch < h.n_channels && e.status == CLX_OK; ch++)
                e = subframe_decode(&bits, bps, out + ch * bs, bs);
            break;
        case CLX_CH_LEFT_SIDE:
            e = subframe_decode(&bits, bps, out, bs);
            if (e.status == CLX_OK) e = subframe_decode(&bits, bps + 1, out + bs, bs);
            if (e.status == CLX_OK) decode_left_side(out, bs);
            break;
        case CLX_CH_RIGHT_SIDE:
            e = subframe_decode(&bits, bps + 1, out, bs);
            if (e.status == CLX_OK) e = subframe_decode(&bits, bps, out + bs, bs);
            if (e.status == CLX_OK) decode_right_side(out, bs);
            break;
        default:
            e = subframe_decode(&bits, bps, out, bs);
            if (e.status == CLX_OK) e = subframe_decode(&bits, bps + 1, out + bs, bs);
            if (e.status == CLX_OK) decode_mid_side(out, bs);
            break;
    }
    info->bytes_consumed = rd.pos;
    if (e.status != CLX_OK) { info->status = e.status; info->msg = e.msg; return e.status; }
    info->end_bit = (uint64_t)rd.pos * 8 - bits.bits_left;
    uint16_t computed = rd.crc16;                      /* frame.rs:753 */
    uint16_t presumed;
    if (!rd_be_u16(&rd, &presumed)) {
        info->bytes_consumed = rd.pos;
        info->status = CLX_IO_ERROR; info->msg = CLX_MSG_UNEXPECTED_EOF; return CLX_IO_ERROR;
    }
    info->bytes_consumed = rd.pos;
    if (check_crc && computed != presumed) {
        info->status = CLX_FORMAT_ERROR; info->msg = CLX_MSG_FRAME_CRC_MISMATCH; return CLX_FORMAT_ERROR;
    }
    info->time = h.variable ? h.number : (uint64_t)h.block_size * (uint64_t)(uint32_t)h.number;   /* frame.rs:771-774 */
    info->status = CLX_OK;
    return CLX_OK;
}

/* subframe::decode (subframe.rs:184) on a fresh Bitstream over Cursor(buf[..len]). */
int clxo_subframe_decode(const uint8_t* buf, size_t len, uint32_t bps, int32_t* out, size_t n,
                         uint32_t* msg, uint64_t* end_bit) {
    pthread_once(&g_crc_once, crc_tables_init);
    reader rd = { buf, len, 0, 0, 0, 0 };
    bitstream bits;
    bs_new(&bits, &rd);
    oerr e = subframe_decode(&bits, bps, out, n);
    if (msg) *msg = e.msg;
    if (end_bit) *end_bit = (uint64_t)rd.pos * 8 - bits.bits_left;
    return e.status;
}

/* ------------------------------------------------------------- lib.rs / metadata.rs (minimum for FlacReader::new) */

static int cur_u8(const uint8_t* d, size_t len, size_t* pos, uint32_t* v) { if (*pos >= len) return 0; *v = d[(*pos)++]; return 1; }
static int cur_be(const uint8_t* d, size_t len, size_t* pos, int nbytes, uint64_t* v) {
    uint64_t r = 0;
    for (int i = 0; i < nbytes; i++) { uint32_t b; if (!cur_u8(d, len, pos, &b)) return 0; r = (r << 8) | b; }
    *v = r; return 1;
}

/* ---- VORBIS_COMMENT (metadata.rs:402-513): a flat record of the parsed block for the tests ----
 * out layout: [u32 vendor_len][vendor bytes][u32 n][ n x { u32 len, u32 sep, bytes } ]  (native endian) */
static int utf8_ok(const uint8_t* p, size_t n) {           /* String::from_utf8 */
    size_t i = 0;
    while (i < n) {
        uint8_t b = p[i];
        uint32_t cp, min; size_t need;
        if (b < 0x80) { i++; continue; }
        if ((b & 0xe0) == 0xc0) { need = 1; cp = b & 0x1f; min = 0x80; }
        else if ((b & 0xf0) == 0xe0) { need = 2; cp = b & 0x0f; min = 0x800; }
        else if ((b & 0xf8) == 0xf0) { need = 3; cp = b & 0x07; min = 0x10000; }
        else return 0;
        if (i + need >= n) return 0;
        for (size_t k = 1; k <= need; k++) { if ((p[i + k] & 0xc0) != 0x80) return 0; cp = (cp << 6) | (p[i + k] & 0x3f); }
        if (cp < min || cp > 0x10ffff || (cp >= 0xd800 && cp <= 0xdfff)) return 0;
        i += need + 1;
    }
    return 1;
}
static int cur_le32(const uint8_t* d, size_t len, size_t* pos, uint32_t* v) {
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) { uint32_t b; if (!cur_u8(d, len, pos, &b)) return 0; r |= b << (8 * i); }
    *v = r; return 1;
}
static void put32(uint8_t* out, size_t cap, size_t* w, uint32_t v) { if (out && *w + 4 <= cap) memcpy(out + *w, &v, 4); *w += 4; }
static void putn(uint8_t* out, size_t cap, size_t* w, const uint8_t* p, size_t n) { if (out && *w + n <= cap) memcpy(out + *w, p, n); *w += n; }

static int read_vorbis_comment_block(const uint8_t* d, size_t len, size_t* pos, uint32_t length, uint8_t* out, size_t cap, size_t* w, uint32_t* msg) {
    if (length < 8) { *msg = CLX_MSG_VC_TOO_SHORT; return CLX_FORMAT_ERROR; }                       /* metadata.rs:403-407 */
    if (length > 10u * 1024 * 1024) { *msg = CLX_MSG_VC_TOO_LARGE; return CLX_UNSUPPORTED; }         /* 422-425 */
    uint32_t vendor_len;
    if (!cur_le32(d, len, pos, &vendor_len)) goto eof;                                               /* 430 */
    if (vendor_len > length - 8) { *msg = CLX_MSG_VC_VENDOR_TOO_LONG; return CLX_FORMAT_ERROR; }     /* 431 */
    if (vendor_len > len - *pos) goto eof;                                                           /* 438 read_into */
    if (!utf8_ok(d + *pos, vendor_len)) { *msg = CLX_MSG_VC_NOT_UTF8; return CLX_FORMAT_ERROR; }     /* 439 */
    put32(out, cap, w, vendor_len); putn(out, cap, w, d + *pos, vendor_len);
    *pos += vendor_len;
    uint32_t comments_len;
    if (!cur_le32(d, len, pos, &comments_len)) goto eof;                                             /* 446 */
    if (comments_len >= length / 4) { *msg = CLX_MSG_VC_TOO_MANY_ENTRIES; return CLX_FORMAT_ERROR; } /* 447-449 */
    size_t count_at = *w;
    put32(out, cap, w, 0);
    uint32_t n = 0, bytes_left = length - 8 - vendor_len;
    while (bytes_left >= 4 && n < comments_len) {                                                    /* 456 */
        uint32_t clen;
        if (!cur_le32(d, len, pos, &clen)) goto eof;
        bytes_left -= 4;
        if (clen > bytes_left) { *msg = CLX_MSG_VC_COMMENT_TOO_LONG; return CLX_FORMAT_ERROR; }      /* 460-462 */
        if (clen == 0) { comments_len -= 1; continue; }                                              /* 467-471 */
        if (clen > len - *pos) goto eof;                                                             /* 476 */
        const uint8_t* c = d + *pos;
        *pos += clen;
        bytes_left -= clen;
        uint32_t sep = clen;
        for (uint32_t i = 0; i < clen; i++) if (c[i] == '=') { sep = i; break; }                     /* 480 */
        if (sep == clen) { *msg = CLX_MSG_VC_NO_EQUALS; return CLX_FORMAT_ERROR; }                   /* 497-499 */
        for (uint32_t i = 0; i < sep; i++) if (c[i] < 0x20 || c[i] > 0x7d) { *msg = CLX_MSG_VC_NAME_INVALID_BYTE; return CLX_FORMAT_ERROR; }   /* 488-492 */
        if (!utf8_ok(c, clen)) { *msg = CLX_MSG_VC_NOT_UTF8; return CLX_FORMAT_ERROR; }              /* 495 */
        put32(out, cap, w, clen); put32(out, cap, w, sep); putn(out, cap, w, c, clen);
        n++;
    }
    if (bytes_left != 0) { *msg = CLX_MSG_VC_EXCESS_DATA; return CLX_FORMAT_ERROR; }                 /* 502-504 */
    if (n != comments_len) { *msg = CLX_MSG_VC_WRONG_COUNT; return CLX_FORMAT_ERROR; }               /* 506-508 */
    if (out && count_at + 4 <= cap) memcpy(out + count_at, &n, 4);
    return CLX_OK;
eof:
    *msg = CLX_MSG_UNEXPECTED_EOF;
    return CLX_IO_ERROR;
}

/* FlacReader::new_ext, lib.rs:230-307: `fLaC` marker (lib.rs:186-205), STREAMINFO first (metadata.rs:321-400), the
 * Vorbis comment block parsed (metadata.rs:402-513; a second one is an error, lib.rs:257-259), every other block
 * skipped by its length (metadata.rs:266-318) until the last-block flag or the options' early-out (lib.rs:273-277).
 * options: bit 0 metadata_only, bit 1 read_vorbis_comment = false.  tags_out (may be NULL) receives the flat record
 * described above, *tags_len its length (0: no Vorbis comment kept). */
int clxo_stream_open_ext(const uint8_t* d, size_t len, uint32_t options, clx_streaminfo* si, uint64_t* audio_off,
                         uint8_t* tags_out, size_t tags_cap, size_t* tags_len, uint32_t* msg);
int clxo_stream_open(const uint8_t* d, size_t len, clx_streaminfo* si, uint64_t* audio_off, uint32_t* msg) {
    size_t tl = 0;
    return clxo_stream_open_ext(d, len, 0, si, audio_off, NULL, 0, &tl, msg);
}
int clxo_stream_open_ext(const uint8_t* d, size_t len, uint32_t options, clx_streaminfo* si, uint64_t* audio_off,
                         uint8_t* tags_out, size_t tags_cap, size_t* tags_len, uint32_t* msg) {
    size_t pos = 0;
    uint64_t hdr;
    *msg = CLX_MSG_NONE;
    if (!cur_be(d, len, &pos, 4, &hdr)) { *msg = CLX_MSG_UNEXPECTED_EOF; return CLX_IO_ERROR; }
    if (hdr != 0x664c6143u) {
        *msg = ((hdr & 0xffffff00u) == 0x49443300u) ? CLX_MSG_ID3_HEADER : CLX_MSG_INVALID_STREAM_HEADER;
        return CLX_FORMAT_ERROR;
    }
    int first = 1, have_si = 0, have_vc = 0;
    const int metadata_only = (options & 1u) != 0;
    int want_vc = (options & 2u) == 0;
    *tags_len = 0;
    for (;;) {
        uint32_t b;
        uint64_t length;
        if (!cur_u8(d, len, &pos, &b)) { *msg = CLX_MSG_UNEXPECTED_EOF; return CLX_IO_ERROR; }
        int is_last = (b >> 7) == 1;
        uint32_t block_type = b & 0x7f;
        if (!cur_be(d, len, &pos, 3, &length)) { *msg = CLX_MSG_UNEXPECTED_EOF; return CLX_IO_ERROR; }
        if (block_type == 0) {
            if (length != 34) { *msg = CLX_MSG_STREAMINFO_LENGTH; return CLX_FORMAT_ERROR; }
            uint64_t v;
            clx_streaminfo s;
            memset(&s, 0, sizeof s);
            if (!cur_be(d, len, &pos, 2, &v)) { goto eof; }
            s.min_block_size = (uint16_t)v;
            if (!cur_be(d, len, &pos, 2, &v)) { goto eof; }
            s.max_block_size = (uint16_t)v;
            if (!cur_be(d, len, &pos, 3, &v)) { goto eof; }
            s.min_frame_size = (uint32_t)v;
            if (!cur_be(d, len, &pos, 3, &v)) { goto eof; }
            s.max_frame_size = (uint32_t)v;
            uint64_t sr_msb, sr_lsb, bps_ns, ns_lsb;
            if (!cur_be(d, len, &pos, 2, &sr_msb)) goto eof;
            if (!cur_be(d, len, &pos, 1, &sr_lsb)) goto eof;
            s.sample_rate = (uint32_t)((sr_msb << 4) | (sr_lsb >> 4));
            s.channels = (uint32_t)(((sr_lsb >> 1) & 7) + 1);
            if (!cur_be(d, len, &pos, 1, &bps_ns)) goto eof;
            s.bits_per_sample = (uint32_t)((((sr_lsb & 1) << 4) | (bps_ns >> 4)) + 1);
            if (!cur_be(d, len, &pos, 4, &ns_lsb)) goto eof;
            s.samples = ((bps_ns & 0x0f) << 32) | ns_lsb;
            if (pos + 16 > len) goto eof;
            memcpy(s.md5sum, d + pos, 16); pos += 16;
            if (s.min_block_size > s.max_block_size) { *msg = CLX_MSG_MIN_BLOCK_GT_MAX_BLOCK; return CLX_FORMAT_ERROR; }
            if (s.min_block_size < 16) { *msg = CLX_MSG_BLOCK_SIZE_LT_16; return CLX_FORMAT_ERROR; }
            if (s.min_frame_size > s.max_frame_size && s.max_frame_size != 0) { *msg = CLX_MSG_MIN_FRAME_GT_MAX_FRAME; return CLX_FORMAT_ERROR; }
            if (s.sample_rate == 0 || s.sample_rate > 655350) { *msg = CLX_MSG_INVALID_SAMPLE_RATE; return CLX_FORMAT_ERROR; }
            if (!first) { *msg = CLX_MSG_SECOND_STREAMINFO; return CLX_FORMAT_ERROR; }   /* lib.rs:267-269 */
            *si = s; have_si = 1;
        } else if (block_type == 4) {
            size_t w = 0;
            int st = read_vorbis_comment_block(d, len, &pos, (uint32_t)length, tags_out, tags_cap, &w, msg);   /* metadata.rs:291-294 */
            if (st != CLX_OK) return st;
            if (first) { *msg = CLX_MSG_STREAMINFO_MISSING; return CLX_FORMAT_ERROR; }                  /* lib.rs:244-248 */
            if (have_vc) { *msg = CLX_MSG_SECOND_VORBIS_COMMENT; return CLX_FORMAT_ERROR; }             /* lib.rs:257-259 */
            have_vc = 1; want_vc = 0; *tags_len = w;
        } else {
            if (first) {
                /* lib.rs:244-248: the first block must be streaminfo.  The block is
                 * still *read* first (metadata_iter.next()), so its own errors win. */
            }
            if (block_type == 127) { *msg = CLX_MSG_INVALID_METADATA_BLOCK_TYPE; return CLX_FORMAT_ERROR; }
            if (block_type == 2) {
                if (length < 4) { *msg = CLX_MSG_APPLICATION_BLOCK_TOO_SHORT; return CLX_FORMAT_ERROR; }
                if (length > 10u * 1024 * 1024) { *msg = CLX_MSG_APPLICATION_BLOCK_TOO_LARGE; return CLX_UNSUPPORTED; }
            }
            if (pos + length > len) goto eof;
            pos += (size_t)length;
            if (first) { *msg = CLX_MSG_STREAMINFO_MISSING; return CLX_FORMAT_ERROR; }
        }
        {
            const int was_first = first;
            first = 0;
            if (is_last) break;
            if (!was_first && metadata_only && !want_vc) break;                                          /* lib.rs:273-277 */
        }
    }
    (void)have_si;
    if (options & 2u) *tags_len = 0;                                                                  /* lib.rs:283-285 */
    *audio_off = pos;
    return CLX_OK;
eof:
    *msg = CLX_MSG_UNEXPECTED_EOF;
    return CLX_IO_ERROR;
}

/* read_metadata_block (metadata.rs:261-319) on Cursor(d[0..len)) positioned behind the block header, and
 * read_metadata_block_with_header (metadata.rs:244-248; header per 214-231).  kind: 0 StreamInfo, 1 Padding,
 * 2 Application, 4 VorbisComment, 126 Reserved (seek table / cue sheet / picture are read as Padding, 287-305).
 * app[0] = id, app[1] = offset of the data in d, app[2] = its length; tags as in clxo_stream_open_ext. */
int clxo_read_metadata_block(const uint8_t* d, size_t len, uint32_t block_type, uint32_t length, uint32_t* kind,
                             clx_streaminfo* si, uint64_t* app, uint8_t* tags_out, size_t tags_cap, size_t* tags_len,
                             size_t* consumed, uint32_t* msg) {
    size_t pos = 0;
    *msg = CLX_MSG_NONE; *tags_len = 0; *consumed = 0; *kind = 0;
    if (block_type == 0) {                                                                            /* 266-274 */
        if (length != 34) { *msg = CLX_MSG_STREAMINFO_LENGTH; return CLX_FORMAT_ERROR; }
        /* read_streaminfo_block, metadata.rs:321-400 */
        uint64_t v, sr_msb, sr_lsb, bps_ns, ns_lsb;
        clx_streaminfo s;
        memset(&s, 0, sizeof s);
        if (!cur_be(d, len, &pos, 2, &v)) goto eof;
        s.min_block_size = (uint16_t)v;
        if (!cur_be(d, len, &pos, 2, &v)) goto eof;
        s.max_block_size = (uint16_t)v;
        if (!cur_be(d, len, &pos, 3, &v)) goto eof;
        s.min_frame_size = (uint32_t)v;
        if (!cur_be(d, len, &pos, 3, &v)) goto eof;
        s.max_frame_size = (uint32_t)v;
        if (!cur_be(d, len, &pos, 2, &sr_msb)) goto eof;
        if (!cur_be(d, len, &pos, 1, &sr_lsb)) goto eof;
        if (!cur_be(d, len, &pos, 1, &bps_ns)) goto eof;
        if (!cur_be(d, len, &pos, 4, &ns_lsb)) goto eof;
        s.sample_rate = (uint32_t)((sr_msb << 4) | (sr_lsb >> 4));
        s.channels = (uint32_t)(((sr_lsb >> 1) & 7) + 1);
        s.bits_per_sample = (uint32_t)((((sr_lsb & 1) << 4) | (bps_ns >> 4)) + 1);
        s.samples = ((bps_ns & 0x0f) << 32) | ns_lsb;
        if (pos + 16 > len) goto eof;
        memcpy(s.md5sum, d + pos, 16); pos += 16;
        if (s.min_block_size > s.max_block_size) { *msg = CLX_MSG_MIN_BLOCK_GT_MAX_BLOCK; return CLX_FORMAT_ERROR; }
        if (s.min_block_size < 16) { *msg = CLX_MSG_BLOCK_SIZE_LT_16; return CLX_FORMAT_ERROR; }
        if (s.min_frame_size > s.max_frame_size && s.max_frame_size != 0) { *msg = CLX_MSG_MIN_FRAME_GT_MAX_FRAME; return CLX_FORMAT_ERROR; }
        if (s.sample_rate == 0 || s.sample_rate > 655350) { *msg = CLX_MSG_INVALID_SAMPLE_RATE; return CLX_FORMAT_ERROR; }
        *si = s; *kind = 0;
    } else if (block_type == 2) {                                                                     /* read_application_block, 525-551 */
        uint64_t id;
        if (length < 4) { *msg = CLX_MSG_APPLICATION_BLOCK_TOO_SHORT; return CLX_FORMAT_ERROR; }
        if (length > 10u * 1024 * 1024) { *msg = CLX_MSG_APPLICATION_BLOCK_TOO_LARGE; return CLX_UNSUPPORTED; }
        if (!cur_be(d, len, &pos, 4, &id)) goto eof;
        if ((size_t)(length - 4) > len - pos) goto eof;                                               /* read_into */
        app[0] = id; app[1] = pos; app[2] = length - 4;
        pos += length - 4;
        *kind = 2;
    } else if (block_type == 4) {                                                                     /* 291-294 */
        size_t w = 0;
        int st = read_vorbis_comment_block(d, len, &pos, length, tags_out, tags_cap, &w, msg);
        if (st != CLX_OK) return st;
        *tags_len = w; *kind = 4;
    } else if (block_type == 127) {                                                                   /* 303-306 */
        *msg = CLX_MSG_INVALID_METADATA_BLOCK_TYPE; return CLX_FORMAT_ERROR;
    } else {                                                                                          /* 1, 3, 5, 6: padding; the rest reserved: skip(length) */
        if ((size_t)length > len - pos) goto eof;
        pos += length;
        *kind = (block_type == 1 || block_type == 3 || block_type == 5 || block_type == 6) ? 1 : 126;
    }
    *consumed = pos;
    return CLX_OK;
eof:
    *msg = CLX_MSG_UNEXPECTED_EOF;
    return CLX_IO_ERROR;
}
int clxo_read_metadata_block_with_header(const uint8_t* d, size_t len, uint32_t* kind, uint32_t* length, int* is_last,
                                         clx_streaminfo* si, uint64_t* app, uint8_t* tags_out, size_t tags_cap, size_t* tags_len,
                                         size_t* consumed, uint32_t* msg) {
    size_t pos = 0;
    uint32_t b;
    uint64_t l24;
    *msg = CLX_MSG_NONE; *consumed = 0; *tags_len = 0; *kind = 0; *length = 0; *is_last = 0;
    if (!cur_u8(d, len, &pos, &b) || !cur_be(d, len, &pos, 3, &l24)) { *msg = CLX_MSG_UNEXPECTED_EOF; return CLX_IO_ERROR; }   /* 214-231 */
    *length = (uint32_t)l24;
    size_t used = 0;
    int st = clxo_read_metadata_block(d + pos, len - pos, b & 0x7f, (uint32_t)l24, kind, si, app, tags_out, tags_cap, tags_len, &used, msg);
    if (st != CLX_OK) return st;
    if (*kind == 2) app[1] += pos;
    *is_last = (b >> 7) == 1;
    *consumed = pos + used;
    return CLX_OK;
}

/* ------------------------------------------------------------- test hooks for the reference's unit vectors */

uint8_t  clxo_crc8(const uint8_t* p, size_t n)  { pthread_once(&g_crc_once, crc_tables_init); uint8_t s = 0;  for (size_t i = 0; i < n; i++) s = g_crc8_table[s ^ p[i]]; return s; }
uint16_t clxo_crc16(const uint8_t* p, size_t n) { pthread_once(&g_crc_once, crc_tables_init); uint16_t s = 0; for (size_t i = 0; i < n; i++) s = (uint16_t)((s << 8) ^ g_crc16_table[(uint8_t)(s >> 8) ^ p[i]]); return s; }
int32_t  clxo_extend_sign_u16(uint32_t v, uint32_t bits) { return extend_sign_u16((uint16_t)v, bits); }
int32_t  clxo_extend_sign_u32(uint32_t v, uint32_t bits) { return extend_sign_u32(v, bits); }
int32_t  clxo_rice_to_signed(uint32_t v) { return rice_to_signed(v); }
void clxo_predict_fixed(uint32_t order, int32_t* buf, size_t n) { predict_fixed(order, buf, n); }
void clxo_predict_lpc_low_order(const int16_t* c, size_t order, int32_t shift, int32_t* buf, size_t n) { predict_lpc_low_order(c, order, (int16_t)shift, buf, n); }
void clxo_predict_lpc_high_order(const int16_t* c, size_t order, int32_t shift, int32_t* buf, size_t n) { predict_lpc_high_order(c, order, (int16_t)shift, buf, n); }
void clxo_decode_left_side(int32_t* buf, size_t total) { decode_left_side(buf, total / 2); }
void clxo_decode_right_side(int32_t* buf, size_t total) { decode_right_side(buf, total / 2); }
void clxo_decode_mid_side(int32_t* buf, size_t total) { decode_mid_side(buf, total / 2); }

/* read_var_length_int on BufferedReader(Cursor(buf)), chained: `*pos` advances. */
int clxo_read_var_length_int(const uint8_t* buf, size_t len, size_t* pos, uint64_t* value, uint32_t* msg) {
    pthread_once(&g_crc_once, crc_tables_init);
    reader rd = { buf, len, *pos, 0, 0, 0 };
    oerr e = read_var_length_int(&rd, value);
    *pos = rd.pos;
    *msg = e.msg;
    return e.status;
}

/* Run a script of Bitstream calls on one Bitstream over Cursor(buf).
 * ops[i] = kind: 0 read_bit, 1 read_unary, 2 read_leq_u8, 3 read_gt_u8_leq_u16,
 * 4 read_leq_u16, 5 read_leq_u32; args[i] = bit count.  values[i] receives the
 * result, errs[i] the status (0 ok / 1 io error). */
void clxo_bitstream_script(const uint8_t* buf, size_t len, const int32_t* ops, const uint32_t* args, size_t n,
                           uint32_t* values, int32_t* errs) {
    pthread_once(&g_crc_once, crc_tables_init);
    reader rd = { buf, len, 0, 0, 0, 0 };
    bitstream bs;
    bs_new(&bs, &rd);
    for (size_t i = 0; i < n; i++) {
        oerr e = ok();
        uint32_t v = 0;
        switch (ops[i]) {
            case 0: { int b = 0; e = bs_read_bit(&bs, &b); v = (uint32_t)b; break; }
            case 1: e = bs_read_unary(&bs, &v); break;
            case 2: { uint8_t x = 0; e = bs_read_leq_u8(&bs, args[i], &x); v = x; break; }
            case 3: e = bs_read_gt_u8_leq_u16(&bs, args[i], &v); break;
            case 4: { uint16_t x = 0; e = bs_read_leq_u16(&bs, args[i], &x); v = x; break; }
            default: e = bs_read_leq_u32(&bs, args[i], &v); break;
        }
        values[i] = v;
        errs[i] = e.status;
    }
}

/* ------------------------------------------------------------- batch driver (CPU baseline timing + parity at scale) */

typedef struct {
    const uint8_t* arena; size_t arena_len;
    const uint64_t* offs; const uint32_t* max_bytes; size_t lo, hi;
    int32_t* out; const uint64_t* out_offs;   /* out may be NULL: decode into a recycled per-thread buffer */
    int32_t* statuses; uint32_t* msgs; uint64_t* end_bits;
    int check_crc;
    uint64_t samples;
} batch_job;

static void* batch_worker(void* p) {
    batch_job* j = (batch_job*)p;
    /* one recycled output buffer per thread, as examples/bench_decode.rs:55-78 */
    int32_t* scratch = NULL;
    if (!j->out) scratch = (int32_t*)malloc(sizeof(int32_t) * 8 * 65535);
    uint64_t samples = 0;
    for (size_t i = j->lo; i < j->hi; i++) {
        clxo_frame_info info;
        size_t avail = j->max_bytes ? j->max_bytes[i] : (j->arena_len - j->offs[i]);
        if (j->offs[i] + avail > j->arena_len) avail = j->arena_len - j->offs[i];
        int32_t* dst = j->out ? j->out + j->out_offs[i] : scratch;
        clxo_frame_decode(j->arena + j->offs[i], avail, dst, (size_t)8 * 65535, j->check_crc, &info);
        if (j->statuses) j->statuses[i] = info.status;
        if (j->msgs) j->msgs[i] = info.msg;
        if (j->end_bits) j->end_bits[i] = info.end_bit;
        if (info.status == CLX_OK) samples += (uint64_t)info.block_size * info.channels;
    }
    free(scratch);
    j->samples = samples;
    return NULL;
}

/* Decode frames [0,n) located at arena[offs[i]] with `nthreads` threads
 * (contiguous shards).  Returns the number of samples decoded OK. */
uint64_t clxo_decode_batch(const uint8_t* arena, size_t arena_len, const uint64_t* offs, const uint32_t* max_bytes,
                           size_t n, int32_t* out, const uint64_t* out_offs,
                           int32_t* statuses, uint32_t* msgs, uint64_t* end_bits,
                           int check_crc, int nthreads) {
    pthread_once(&g_crc_once, crc_tables_init);
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n && n > 0) nthreads = (int)n;
    batch_job* jobs = (batch_job*)calloc((size_t)nthreads, sizeof(batch_job));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int t = 0; t < nthreads; t++) {
        batch_job j = { arena, arena_len, offs, max_bytes, n * (size_t)t / (size_t)nthreads, n * (size_t)(t + 1) / (size_t)nthreads,
                        out, out_offs, statuses, msgs, end_bits, check_crc, 0 };
        jobs[t] = j;
    }
    if (nthreads == 1) batch_worker(&jobs[0]);
    else {
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    }
    uint64_t total = 0;
    for (int t = 0; t < nthreads; t++) total += jobs[t].samples;
    free(jobs); free(th);
    return total;
}

/* Timing harness for the CPU baseline (bench.py's `cpu_baseline` leg): a persistent pool of `nthreads` threads, created,
 * pinned (thread t on CPU cpus[t] when `cpus` is given) and warmed by an untimed pass over its share BEFORE the clock starts;
 * every thread owns a recycled output buffer (examples/bench_decode.rs:55-78).  Between two barriers the threads then decode
 * `passes` x n frames, taking chunks of `chunk` frames off a shared counter (what a thread-pool decoder does; a static split
 * would time the unluckiest core of a shared host instead of the machine).  The timed region is barrier to barrier -- no
 * thread creation, no allocation (tools/benchmark.sh:39-41 pins its process the same way).  Returns the samples decoded
 * inside the timed region and its duration in *seconds. */
typedef struct {
    batch_job job;
    pthread_barrier_t* bar;
    uint64_t* next;             /* shared work counter over passes * n frame slots */
    uint64_t total_slots, n;
    int cpu, chunk;
    uint64_t timed_samples;
    struct timespec t0, t1;
} bench_job;

static void* bench_worker(void* p) {
    bench_job* b = (bench_job*)p;
    if (b->cpu >= 0) {
        cpu_set_t set; CPU_ZERO(&set); CPU_SET(b->cpu, &set);
        (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
    }
    batch_job* j = &b->job;
    int32_t* scratch = (int32_t*)malloc(sizeof(int32_t) * 8 * 65535);
    clxo_frame_info info;
    for (size_t i = j->lo; i < j->hi; i++) {           /* warm-up: this thread's contiguous share, untimed */
        size_t avail = j->max_bytes ? j->max_bytes[i] : (j->arena_len - j->offs[i]);
        if (j->offs[i] + avail > j->arena_len) avail = j->arena_len - j->offs[i];
        clxo_frame_decode(j->arena + j->offs[i], avail, scratch, (size_t)8 * 65535, j->check_crc, &info);
    }
    uint64_t samples = 0;
    pthread_barrier_wait(b->bar);
    clock_gettime(CLOCK_MONOTONIC, &b->t0);
    for (;;) {
        const uint64_t s0 = __atomic_fetch_add(b->next, (uint64_t)b->chunk, __ATOMIC_RELAXED);
        if (s0 >= b->total_slots) break;
        const uint64_t s1 = s0 + (uint64_t)b->chunk < b->total_slots ? s0 + (uint64_t)b->chunk : b->total_slots;
        for (uint64_t s = s0; s < s1; s++) {
            const size_t i = (size_t)(s % b->n);
            size_t avail = j->max_bytes ? j->max_bytes[i] : (j->arena_len - j->offs[i]);
            if (j->offs[i] + avail > j->arena_len) avail = j->arena_len - j->offs[i];
            clxo_frame_decode(j->arena + j->offs[i], avail, scratch, (size_t)8 * 65535, j->check_crc, &info);
            if (info.status == CLX_OK) samples += (uint64_t)info.block_size * info.channels;
        }
    }
    pthread_barrier_wait(b->bar);
    clock_gettime(CLOCK_MONOTONIC, &b->t1);
    b->timed_samples = samples;
    free(scratch);
    return NULL;
}

uint64_t clxo_bench_batch(const uint8_t* arena, size_t arena_len, const uint64_t* offs, const uint32_t* max_bytes, size_t n,
                          int check_crc, int nthreads, int passes, const int* cpus, double* seconds) {
    pthread_once(&g_crc_once, crc_tables_init);
    if (seconds) *seconds = 0.0;
    if (n == 0) return 0;
    if (nthreads < 1) nthreads = 1;
    if (passes < 1) passes = 1;
    bench_job* jobs = (bench_job*)calloc((size_t)nthreads, sizeof(bench_job));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)nthreads);
    uint64_t next = 0;
    for (int t = 0; t < nthreads; t++) {
        batch_job j = { arena, arena_len, offs, max_bytes, n * (size_t)t / (size_t)nthreads, n * (size_t)(t + 1) / (size_t)nthreads,
                        NULL, NULL, NULL, NULL, NULL, check_crc, 0 };
        jobs[t].job = j; jobs[t].bar = &bar; jobs[t].next = &next;
        jobs[t].total_slots = (uint64_t)passes * n; jobs[t].n = n; jobs[t].chunk = 4;
        jobs[t].cpu = cpus ? cpus[t] : -1;
    }
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, bench_worker, &jobs[t]);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&bar);
    uint64_t total = 0;
    /* barrier to barrier: the earliest start stamp to the latest end stamp */
    double t0 = 1e300, t1 = 0.0;
    for (int t = 0; t < nthreads; t++) {
        total += jobs[t].timed_samples;
        const double a = (double)jobs[t].t0.tv_sec + 1e-9 * (double)jobs[t].t0.tv_nsec;
        const double b = (double)jobs[t].t1.tv_sec + 1e-9 * (double)jobs[t].t1.tv_nsec;
        if (a < t0) t0 = a;
        if (b > t1) t1 = b;
    }
    if (seconds) *seconds = t1 - t0;
    free(jobs); free(th);
    return total;
}

/* subframe::decode for n independent byte-aligned subframes (config 2). */
uint64_t clxo_decode_subframes(const uint8_t* arena, size_t arena_len, const uint64_t* offs,
                               const uint16_t* block_sizes, const uint8_t* bps, size_t n,
                               int32_t* out, const uint64_t* out_offs,
                               int32_t* statuses, uint32_t* msgs, uint64_t* end_bits) {
    pthread_once(&g_crc_once, crc_tables_init);
    uint64_t total = 0;
    int32_t* scratch = out ? NULL : (int32_t*)malloc(sizeof(int32_t) * 65536);
    for (size_t i = 0; i < n; i++) {
        uint32_t msg = 0; uint64_t eb = 0;
        int32_t* dst = out ? out + out_offs[i] : scratch;
        int st = clxo_subframe_decode(arena + offs[i], arena_len - offs[i], bps[i], dst, block_sizes[i], &msg, &eb);
        if (statuses) statuses[i] = st;
        if (msgs) msgs[i] = msg;
        if (end_bits) end_bits[i] = eb;
        if (st == CLX_OK) total += block_sizes[i];
    }
    free(scratch);
    return total;
}
