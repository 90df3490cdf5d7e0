/*
 * flacsynth.c -- a miniature FLAC *encoder* used to manufacture test/bench input.
 *
 * TEST/BENCH INFRASTRUCTURE (not product, not oracle).  It turns PCM into valid
 * FLAC frames (real CRC-8 / CRC-16, subset or non-subset on request) with the
 * coding tools the decode path has to handle: CONSTANT / VERBATIM / FIXED 0-4 /
 * LPC 1-32 subframes, Rice and Rice2 partitions of any partition order, wasted
 * bits, and independent / left-side / right-side / mid-side channel assignment.
 * Residuals are computed with the decoder's own integer formula, so
 * decode(encode(pcm)) == pcm holds by construction -- a second, independent
 * check next to the oracle.  Written from the FLAC format description; shares
 * no code with oracle/ or claxon_amd/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- bit writer (MSB first) */
typedef struct { uint8_t* buf; size_t cap; size_t bitpos; int overflow; } bitw;

static void bw_put(bitw* w, uint64_t v, unsigned n) {   /* n <= 57 */
    while (n > 0) {
        size_t byte = w->bitpos >> 3;
        unsigned used = (unsigned)(w->bitpos & 7), room = 8 - used;
        unsigned take = n < room ? n : room;
        if (byte >= w->cap) { w->overflow = 1; return; }
        uint8_t bits = (uint8_t)((v >> (n - take)) & ((1u << take) - 1));
        if (used == 0) w->buf[byte] = 0;
        w->buf[byte] |= (uint8_t)(bits << (room - take));
        w->bitpos += take;
        n -= take;
    }
}
static void bw_unary(bitw* w, uint32_t q) {   /* q zeros then a one */
    while (q >= 32) { bw_put(w, 0, 32); q -= 32; }
    bw_put(w, 1, q + 1);
}
static void bw_align(bitw* w) { if (w->bitpos & 7) bw_put(w, 0, 8 - (unsigned)(w->bitpos & 7)); }

/* ---------------------------------------------------------------- CRCs (bitwise; polynomials from the FLAC format) */
static uint8_t crc8(const uint8_t* p, size_t n) {
    uint8_t c = 0;
    for (size_t i = 0; i < n; i++) { c ^= p[i]; for (int b = 0; b < 8; b++) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : c << 1); }
    return c;
}
static uint16_t crc16(const uint8_t* p, size_t n) {
    uint16_t c = 0;
    for (size_t i = 0; i < n; i++) { c ^= (uint16_t)(p[i] << 8); for (int b = 0; b < 8; b++) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : c << 1); }
    return c;
}

/* ---------------------------------------------------------------- parameters */
enum { SF_CONSTANT = 0, SF_VERBATIM = 1, SF_FIXED = 2, SF_LPC = 3 };

typedef struct synth_subframe_params {
    int32_t type;             /* SF_* */
    int32_t order;            /* fixed 0..4, lpc 1..32 */
    int32_t qlp_precision;    /* 2..15 (lpc) */
    int32_t partition_order;  /* 0..15 */
    int32_t rice_param;       /* >=0: force this k in every partition; -1: optimal per partition */
    int32_t force_rice2;      /* 1: always 5-bit parameters (method 01) */
    int32_t wasted;           /* -1: auto-detect common trailing zero bits; >=0: force (must divide the data) */
    int32_t reserved;
} synth_subframe_params;

typedef struct synth_frame_params {
    int32_t channel_assignment;   /* 0 independent, 1 left/side, 2 right/side, 3 mid/side */
    int32_t variable_blocking;    /* 0: frame number, 1: sample number */
    uint64_t number;              /* frame number or sample number */
    synth_subframe_params sf[8];
} synth_frame_params;

/* ---------------------------------------------------------------- LPC analysis (double precision; only chooses coefficients) */
static int lpc_analyse(const int32_t* x, int n, int order, int precision, int16_t* qcoef, int* shift_out) {
    double ac[33];
    for (int lag = 0; lag <= order; lag++) {
        double s = 0;
        for (int i = lag; i < n; i++) {
            /* Welch-ish window keeps the normal equations well conditioned */
            double wi = 1.0 - pow((2.0 * i - (n - 1)) / (double)(n + 1), 2);
            double wl = 1.0 - pow((2.0 * (i - lag) - (n - 1)) / (double)(n + 1), 2);
            s += (x[i] * wi) * (x[i - lag] * wl);
        }
        ac[lag] = s;
    }
    if (ac[0] <= 0) { for (int i = 0; i < order; i++) qcoef[i] = 0; *shift_out = 0; return 0; }
    double a[33] = { 0 }, tmp[33];
    double err = ac[0];
    for (int i = 1; i <= order; i++) {
        double acc = ac[i];
        for (int j = 1; j < i; j++) acc -= a[j] * ac[i - j];
        double k = err > 0 ? acc / err : 0;
        memcpy(tmp, a, sizeof tmp);
        a[i] = k;
        for (int j = 1; j < i; j++) a[j] = tmp[j] - k * tmp[i - j];
        err *= (1 - k * k);
        if (err <= 0) err = 1e-9;
    }
    /* a[1..order]: x[i] ~ sum a[j] x[i-j]  */
    double cmax = 0;
    for (int j = 1; j <= order; j++) if (fabs(a[j]) > cmax) cmax = fabs(a[j]);
    int shift;
    if (cmax <= 0) shift = 0;
    else {
        int e; frexp(cmax, &e);               /* cmax = m * 2^e, m in [0.5,1) */
        shift = precision - 1 - e;
        if (shift > 15) shift = 15;           /* 5-bit signed field, keep it non-negative */
        if (shift < 0) shift = 0;
    }
    double errf = 0;
    int32_t qmax = (1 << (precision - 1)) - 1, qmin = -(1 << (precision - 1));
    for (int j = 1; j <= order; j++) {
        errf += a[j] * (double)(1 << shift);
        long q = lround(errf);
        if (q > qmax) q = qmax;
        if (q < qmin) q = qmin;
        errf -= (double)q;
        qcoef[j - 1] = (int16_t)q;            /* qcoef[j-1] multiplies x[i-j]  (first coded coefficient <-> newest sample) */
    }
    *shift_out = shift;
    return 1;
}

/* ---------------------------------------------------------------- residual coding */
static inline uint32_t zigzag(int32_t r) { return ((uint32_t)r << 1) ^ (uint32_t)(r >> 31); }

static int best_rice_param(const int32_t* res, int n, int kmax, uint64_t* bits_out) {
    int best = 0; uint64_t best_bits = UINT64_MAX;
    for (int k = 0; k <= kmax; k++) {
        uint64_t bits = (uint64_t)n * (uint64_t)(k + 1);
        for (int i = 0; i < n; i++) bits += zigzag(res[i]) >> k;
        if (bits < best_bits) { best_bits = bits; best = k; }
    }
    if (bits_out) *bits_out = best_bits;
    return best;
}

static void write_residual(bitw* w, const int32_t* res /* [bs-order] */, int bs, int order, const synth_subframe_params* p) {
    int porder = p->partition_order;
    int nparts = 1 << porder;
    int per = bs >> porder;
    int* kk = (int*)malloc(sizeof(int) * (size_t)nparts);
    int rice2 = p->force_rice2;
    int start = 0;
    for (int part = 0; part < nparts; part++) {
        int cnt = (part == 0) ? per - order : per;
        int k = p->rice_param >= 0 ? p->rice_param : best_rice_param(res + start, cnt, 30, NULL);
        kk[part] = k;
        if (k > 14) rice2 = 1;
        start += cnt;
    }
    bw_put(w, rice2 ? 1 : 0, 2);
    bw_put(w, (uint64_t)porder, 4);
    start = 0;
    for (int part = 0; part < nparts; part++) {
        int cnt = (part == 0) ? per - order : per;
        int k = kk[part];
        bw_put(w, (uint64_t)k, rice2 ? 5 : 4);
        for (int i = 0; i < cnt; i++) {
            uint32_t u = zigzag(res[start + i]);
            bw_unary(w, u >> k);
            if (k) bw_put(w, u & ((1u << k) - 1), (unsigned)k);
        }
        start += cnt;
    }
    free(kk);
}

static int32_t* g_scratch = NULL;   /* not thread safe; the generator is single threaded */
static size_t g_scratch_cap = 0;

/* Encode one subframe of `bs` samples `x` (each fitting `bps` bits). */
static void write_subframe(bitw* w, const int32_t* x_in, int bs, int bps, const synth_subframe_params* p) {
    if (g_scratch_cap < (size_t)bs * 2) { g_scratch_cap = (size_t)bs * 2; g_scratch = (int32_t*)realloc(g_scratch, sizeof(int32_t) * g_scratch_cap); }
    int32_t* x = g_scratch;
    int32_t* res = g_scratch + bs;
    int wasted = p->wasted;
    if (wasted < 0) {
        uint32_t orv = 0;
        for (int i = 0; i < bs; i++) orv |= (uint32_t)x_in[i];
        wasted = orv ? __builtin_ctz(orv) : 0;
        if (wasted >= bps) wasted = bps - 1;
    }
    for (int i = 0; i < bs; i++) x[i] = x_in[i] >> wasted;
    int sf_bps = bps - wasted;
    int type = p->type, order = p->order;

    bw_put(w, 0, 1);
    unsigned code;
    switch (type) {
        case SF_CONSTANT: code = 0; break;
        case SF_VERBATIM: code = 1; break;
        case SF_FIXED: code = 8u | (unsigned)order; break;
        default: code = 32u | (unsigned)(order - 1); break;
    }
    bw_put(w, code, 6);
    if (wasted) { bw_put(w, 1, 1); bw_unary(w, (uint32_t)(wasted - 1)); } else bw_put(w, 0, 1);

    uint64_t mask = sf_bps >= 32 ? 0xffffffffull : ((1ull << sf_bps) - 1);
    if (type == SF_CONSTANT) { bw_put(w, (uint64_t)(uint32_t)x[0] & mask, (unsigned)sf_bps); return; }
    if (type == SF_VERBATIM) { for (int i = 0; i < bs; i++) bw_put(w, (uint64_t)(uint32_t)x[i] & mask, (unsigned)sf_bps); return; }

    for (int i = 0; i < order; i++) bw_put(w, (uint64_t)(uint32_t)x[i] & mask, (unsigned)sf_bps);
    if (type == SF_FIXED) {
        static const int32_t fc[5][4] = { {0,0,0,0}, {1,0,0,0}, {2,-1,0,0}, {3,-3,1,0}, {4,-6,4,-1} };   /* fc[o][j] multiplies x[i-1-j] */
        for (int i = order; i < bs; i++) {
            int64_t pred = 0;
            for (int j = 0; j < order; j++) pred += (int64_t)fc[order][j] * x[i - 1 - j];
            res[i - order] = (int32_t)((int64_t)x[i] - pred);
        }
    } else {
        int16_t qc[32]; int shift = 0;
        lpc_analyse(x, bs, order, p->qlp_precision, qc, &shift);
        bw_put(w, (uint64_t)(p->qlp_precision - 1), 4);
        bw_put(w, (uint64_t)shift & 31, 5);
        for (int j = 0; j < order; j++) bw_put(w, (uint64_t)(uint16_t)qc[j] & ((1u << p->qlp_precision) - 1), (unsigned)p->qlp_precision);
        for (int i = order; i < bs; i++) {
            int64_t sum = 0;
            for (int j = 0; j < order; j++) sum += (int64_t)qc[j] * (int64_t)x[i - 1 - j];
            int64_t pred = sum >> shift;
            res[i - order] = (int32_t)((int64_t)x[i] - pred);
        }
    }
    write_residual(w, res, bs, order, p);
}

static void put_varint(bitw* w, uint64_t v) {   /* FLAC's "UTF-8"-style coding, up to 36 bits */
    if (v < 0x80) { bw_put(w, v, 8); return; }
    int extra = v < 0x800 ? 1 : v < 0x10000 ? 2 : v < 0x200000 ? 3 : v < 0x4000000 ? 4 : v < 0x80000000ull ? 5 : 6;
    unsigned lead = (unsigned)((0xff << (7 - extra)) & 0xff);
    bw_put(w, lead | (unsigned)(v >> (6 * extra)), 8);
    for (int i = extra - 1; i >= 0; i--) bw_put(w, 0x80 | ((v >> (6 * i)) & 0x3f), 8);
}

/* Encode one frame.  pcm: planar [channels][bs] (L, R for stereo).  Returns bytes written (0 on overflow). */
size_t synth_encode_frame(const int32_t* pcm, int channels, int bs, int bps, int sample_rate,
                          const synth_frame_params* fp, uint8_t* out, size_t cap) {
    bitw w = { out, cap, 0, 0 };
    bw_put(&w, 0x3ffe, 14); bw_put(&w, 0, 1); bw_put(&w, (uint64_t)(fp->variable_blocking ? 1 : 0), 1);
    int bs_code, bs_extra = 0;
    if (bs == 192) bs_code = 1;
    else if (bs == 576 || bs == 1152 || bs == 2304 || bs == 4608) bs_code = 2 + __builtin_ctz((unsigned)(bs / 576));
    else if (bs >= 256 && bs <= 32768 && (bs & (bs - 1)) == 0) bs_code = 8 + __builtin_ctz((unsigned)(bs / 256));
    else if (bs <= 256) { bs_code = 6; bs_extra = 8; }
    else { bs_code = 7; bs_extra = 16; }
    int sr_code = 0;
    switch (sample_rate) { case 88200: sr_code = 1; break; case 176400: sr_code = 2; break; case 192000: sr_code = 3; break;
        case 8000: sr_code = 4; break; case 16000: sr_code = 5; break; case 22050: sr_code = 6; break; case 24000: sr_code = 7; break;
        case 32000: sr_code = 8; break; case 44100: sr_code = 9; break; case 48000: sr_code = 10; break; case 96000: sr_code = 11; break; default: sr_code = 0; }
    bw_put(&w, (uint64_t)bs_code, 4); bw_put(&w, (uint64_t)sr_code, 4);
    int ca = fp->channel_assignment;
    bw_put(&w, (uint64_t)(ca == 0 ? channels - 1 : 7 + ca), 4);
    int bps_code = bps == 8 ? 1 : bps == 12 ? 2 : bps == 16 ? 4 : bps == 20 ? 5 : bps == 24 ? 6 : 0;
    bw_put(&w, (uint64_t)bps_code, 3); bw_put(&w, 0, 1);
    put_varint(&w, fp->number);
    if (bs_extra) bw_put(&w, (uint64_t)(bs - 1), (unsigned)bs_extra);
    bw_put(&w, crc8(out, w.bitpos >> 3), 8);

    if (ca == 0) {
        for (int c = 0; c < channels; c++) write_subframe(&w, pcm + (size_t)c * bs, bs, bps, &fp->sf[c]);
    } else {
        int32_t* t = (int32_t*)malloc(sizeof(int32_t) * (size_t)bs * 2);
        const int32_t* L = pcm; const int32_t* R = pcm + bs;
        int32_t* c0 = t; int32_t* c1 = t + bs;
        for (int i = 0; i < bs; i++) {
            int32_t side = L[i] - R[i];
            if (ca == 1) { c0[i] = L[i]; c1[i] = side; }
            else if (ca == 2) { c0[i] = side; c1[i] = R[i]; }
            else { c0[i] = (L[i] + R[i]) >> 1; c1[i] = side; }
        }
        write_subframe(&w, c0, bs, ca == 2 ? bps + 1 : bps, &fp->sf[0]);
        write_subframe(&w, c1, bs, ca == 2 ? bps : bps + 1, &fp->sf[1]);
        free(t);
    }
    bw_align(&w);
    size_t nbytes = w.bitpos >> 3;
    if (w.overflow || nbytes + 2 > cap) return 0;
    uint16_t c = crc16(out, nbytes);
    out[nbytes] = (uint8_t)(c >> 8); out[nbytes + 1] = (uint8_t)c;
    return nbytes + 2;
}

/* Encode a bare subframe (config 2: no frame header), byte aligned, zero padded. */
size_t synth_encode_subframe(const int32_t* x, int bs, int bps, const synth_subframe_params* p, uint8_t* out, size_t cap) {
    bitw w = { out, cap, 0, 0 };
    write_subframe(&w, x, bs, bps, p);
    bw_align(&w);
    return w.overflow ? 0 : (w.bitpos >> 3);
}

/* Batch: frames share (channels, bs, bps); pcm is [n][channels][bs]; fps[n].  Appends frames to
 * `arena` back to back starting at arena_pos; fills offs[n], lens[n].  Returns new arena_pos (0 on overflow). */
size_t synth_encode_frames(const int32_t* pcm, size_t n, int channels, int bs, int bps, int sample_rate,
                           const synth_frame_params* fps, uint8_t* arena, size_t arena_cap, size_t arena_pos,
                           uint64_t* offs, uint32_t* lens) {
    for (size_t f = 0; f < n; f++) {
        size_t got = synth_encode_frame(pcm + f * (size_t)channels * (size_t)bs, channels, bs, bps, sample_rate, &fps[f],
                                        arena + arena_pos, arena_cap - arena_pos);
        if (!got) return 0;
        offs[f] = arena_pos; lens[f] = (uint32_t)got;
        arena_pos += got;
    }
    return arena_pos;
}

size_t synth_encode_subframes(const int32_t* x, size_t n, int bs, int bps, const synth_subframe_params* ps,
                              uint8_t* arena, size_t arena_cap, size_t arena_pos, uint64_t* offs, uint32_t* lens) {
    for (size_t f = 0; f < n; f++) {
        size_t got = synth_encode_subframe(x + f * (size_t)bs, bs, bps, &ps[f], arena + arena_pos, arena_cap - arena_pos);
        if (!got) return 0;
        offs[f] = arena_pos; lens[f] = (uint32_t)got;
        arena_pos += got;
    }
    return arena_pos;
}

/* Re-stamp the frame number of an already encoded fixed-blocking frame whose number field
 * has the same coded length (used to tile unique frames into a long stream): rewrites the
 * varint in place and refreshes CRC-8 and CRC-16.  Returns 0 if the lengths differ. */
int synth_restamp_frame(uint8_t* frame, size_t len, uint64_t new_number) {
    uint8_t tmp[16];
    bitw w = { tmp, sizeof tmp, 0, 0 };
    put_varint(&w, new_number);
    size_t newlen = w.bitpos >> 3;
    /* old length from the leading byte */
    uint8_t first = frame[4];
    size_t oldlen = 1;
    if (first & 0x80) { oldlen = 0; for (uint8_t m = 0x80; first & m; m >>= 1) oldlen++; }
    if (oldlen != newlen) return 0;
    memcpy(frame + 4, tmp, newlen);
    size_t hdr = 4 + newlen;
    unsigned bs_code = frame[2] >> 4;
    if (bs_code == 6) hdr += 1; else if (bs_code == 7) hdr += 2;
    unsigned sr_code = frame[2] & 15;
    if (sr_code == 12) hdr += 1; else if (sr_code == 13 || sr_code == 14) hdr += 2;
    frame[hdr] = crc8(frame, hdr);
    uint16_t c = crc16(frame, len - 2);
    frame[len - 2] = (uint8_t)(c >> 8); frame[len - 1] = (uint8_t)c;
    return 1;
}

/* Tile a set of U unique frames into a long stream (">= 16 384 unique frames tiled to 1M with
 * distinct frame numbers / CRCs"): frame i of the stream is unique frame i % U re-stamped with number `number_base + i`.
 * Writes frames [lo, hi) back to back into `out` (offs[i - lo] = where frame i starts).  Returns the bytes written, 0 when
 * `out` is too small or a frame number does not fit the unique frame's number field. */
size_t synth_tile_frames(const uint8_t* uarena, const uint64_t* uoffs, const uint32_t* ulens, size_t U,
                         uint64_t lo, uint64_t hi, uint64_t number_base, uint8_t* out, size_t cap, uint64_t* offs) {
    size_t pos = 0;
    for (uint64_t i = lo; i < hi; i++) {
        const size_t u = (size_t)(i % U);
        const size_t len = ulens[u];
        if (pos + len > cap) return 0;
        memcpy(out + pos, uarena + uoffs[u], len);
        if (!synth_restamp_frame(out + pos, len, number_base + i)) return 0;
        offs[i - lo] = pos;
        pos += len;
    }
    return pos;
}
