// clx_lanes.hip -- the "many frames" path: lane-serial decode, 64 independent subframes per wavefront.
//
//   P  clx_k_scan    one lane per multi-channel frame: parses channels 0..C-2 (headers + every Rice code,
//                    no output) to find the bit at which each later subframe starts -- subframe c+1
//                    begins where subframe c ends, there is no length field (frame.rs:705-742).
//   D  clx_k_lanes   one lane per subframe: subframe header, warm-up, LPC coefficients, Rice/Rice2
//      (_hi: orders   residual decode, fixed/LPC synthesis, wasted-bits shift and stereo decorrelation
//       above 12)     (partner channel in lane^1, exchanged with DPP) fused in one pass; sixteen samples per
//                    turn leave as 64-byte row segments.  No intermediate residual buffer in HBM.
//   D2 clx_k_lanes2  the same work split over a Rice wave and a predictor/finisher wave per 64 subframes
//                    (tiles handed over through LDS): half the serial chain per wave, for batches that
//                    leave most SIMDs without a wave.
//   F  clx_k_finalize  folds the per-frame error keys into clx_frame_result (and performs the footer read).
//
// Why two paths: a lone wavefront issues one instruction every ~6 cycles, and the wave-parallel
// decoder (clx_kernels.hip) spends ~3.4 wave-instructions per code to resolve code boundaries
// speculatively; decoding serially in each lane costs ~0.6 wave-instructions per code.  With
// tens of thousands of subframes in flight the lane-serial form wins by a wide margin; with fewer
// the wave-parallel form has the lower latency.  clx_batch_run picks by batch shape.
//
// Both kernels mirror the reference's call sequence read for read (subframe.rs:29-91, 184-228,
// 236-415, 492-516, 651-721), so the first error in stream order is the one reported.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <clx_intrin.h>

#include "../../include/claxon_hip.h"
#include <type_traits>
#include "clx_device.h"

#define CLX_LERR(status, msg) (((uint32_t)(status) << 16) | (uint32_t)(msg))

// Per-lane MSB-first bit reader straight over the arena (vector L1).  Used for headers and for the rare
// "careful" steps; the per-code hot path reads through the LDS ring below.  `pos` counts bits from the
// frame's 16-byte aligned origin.
struct LaneReader {
    const uint8_t* arena;     // wave-uniform
    uint32_t origin;          // byte offset of the origin from `arena` (multiple of 16); arena_len < 4 GiB on this path
    uint32_t pos;
    uint32_t limit;           // first unreadable bit
    uint32_t err;             // CLX_LERR(...) of the first error, 0 = none
};

// 32 bits at `pos`, left aligned.  The arena allocation is padded (claxon_hip.h), so the 8-byte load
// that straddles the last readable bit stays inside it.
__device__ __forceinline__ uint32_t clx_lpeek32(const LaneReader& r, uint32_t pos) {
    const uint32_t boff = r.origin + ((pos >> 3) & ~3u);
    const uint32_t* p = reinterpret_cast<const uint32_t*>(r.arena + boff);       // 4-byte aligned: one global_load_dwordx2
    const uint64_t w = ((uint64_t)__builtin_bswap32(p[0]) << 32) | __builtin_bswap32(p[1]);
    return (uint32_t)((w << (pos & 31u)) >> 32);
}

// read_leq_u32-style field (n in 0..32); EOF sets r.err and returns 0 (input.rs:626-642)
__device__ __forceinline__ uint32_t clx_lread(LaneReader& r, uint32_t n) {
    if (r.err) return 0u;
    if (r.pos + n > r.limit) { r.err = CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF); return 0u; }
    const uint32_t v = n ? (clx_lpeek32(r, r.pos) >> (32u - n)) : 0u;
    r.pos += n;
    return v;
}
__device__ __forceinline__ int32_t clx_lread_signed(LaneReader& r, uint32_t n) {      // extend_sign_u32, subframe.rs:117-122
    if (r.err) return 0;
    if (r.pos + n > r.limit) { r.err = CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF); return 0; }
    const int32_t v = (int32_t)clx_lpeek32(r, r.pos) >> (32u - n);
    r.pos += n;
    return v;
}
// read_unary (input.rs:475-511): zeros before the next one bit
__device__ __forceinline__ uint32_t clx_lread_unary(LaneReader& r) {
    if (r.err) return 0u;
    uint32_t t = r.pos;
    uint32_t w = 0;
    while (t < r.limit) { w = clx_lpeek32(r, t); if (w) break; t += 32u; }
    if (w) t += (uint32_t)__clz((int)w);
    if (!w || t >= r.limit) { r.err = CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF); return 0u; }
    const uint32_t q = t - r.pos;
    r.pos = t + 1u;
    return q;
}

struct SfHead { uint32_t kind, order, wasted, sf_bps; };     // kind: 0 constant, 1 verbatim, 2 fixed, 3 lpc

// read_subframe_header (subframe.rs:29-91) + the wasted-bits check of subframe::decode (subframe.rs:198-204)
__device__ __forceinline__ SfHead clx_lparse_sf_header(LaneReader& r, uint32_t bps) {
    SfHead h = { 0u, 0u, 0u, bps };
    uint32_t v = clx_lread(r, 1);
    if (r.err) return h;
    if (v) { r.err = CLX_LERR(CLX_FORMAT_ERROR, CLX_MSG_SUBFRAME_HEADER_INVALID); return h; }
    v = clx_lread(r, 6);
    if (r.err) return h;
    if (v == 0u) h.kind = 0;
    else if (v == 1u) h.kind = 1;
    else if ((v & 0x3eu) == 0x02u || (v & 0x3cu) == 0x04u || (v & 0x30u) == 0x10u) {
        r.err = CLX_LERR(CLX_FORMAT_ERROR, CLX_MSG_SUBFRAME_HEADER_RESERVED); return h;
    } else if ((v & 0x38u) == 0x08u) {
        h.order = v & 7u;
        if (h.order > 4u) { r.err = CLX_LERR(CLX_FORMAT_ERROR, CLX_MSG_SUBFRAME_HEADER_RESERVED); return h; }
        h.kind = 2;
    } else { h.kind = 3; h.order = (v & 0x1fu) + 1u; }
    v = clx_lread(r, 1);
    if (r.err) return h;
    if (v) {
        const uint32_t q = clx_lread_unary(r);
        if (r.err) return h;
        h.wasted = 1u + q;
        if (h.wasted > 31u) { r.err = CLX_LERR(CLX_FORMAT_ERROR, CLX_MSG_WASTED_BITS_EXCEED_31); return h; }
    }
    if (h.wasted >= bps) { r.err = CLX_LERR(CLX_FORMAT_ERROR, CLX_MSG_NO_NON_WASTED_BITS); return h; }
    h.sf_bps = bps - h.wasted;
    return h;
}

struct ResHead { uint32_t rice2, per, n_part; };
// head of decode_residual (subframe.rs:241-277)
__device__ __forceinline__ ResHead clx_lparse_residual_header(LaneReader& r, uint32_t bs, uint32_t order) {
    ResHead h = { 0u, 0u, 0u };
    uint32_t v = clx_lread(r, 2);
    if (r.err) return h;
    if (v > 1u) { r.err = CLX_LERR(CLX_FORMAT_ERROR, CLX_MSG_RESIDUAL_RESERVED); return h; }
    h.rice2 = v;
    v = clx_lread(r, 4);
    if (r.err) return h;
    h.n_part = 1u << v;
    h.per = bs >> v;
    if ((bs & (h.n_part - 1u)) != 0u) { r.err = CLX_LERR(CLX_FORMAT_ERROR, CLX_MSG_INVALID_PARTITION_ORDER); return h; }
    if (order > h.per) { r.err = CLX_LERR(CLX_FORMAT_ERROR, CLX_MSG_INVALID_RESIDUAL); return h; }
    return h;
}
// partition parameter (subframe.rs:314-319 / 362-367); returns k
__device__ __forceinline__ uint32_t clx_lread_rice_param(LaneReader& r, uint32_t rice2) {
    const uint32_t k = clx_lread(r, rice2 ? 5u : 4u);
    if (!r.err && k == (rice2 ? 31u : 15u)) r.err = CLX_LERR(CLX_UNSUPPORTED, CLX_MSG_UNENCODED_BINARY);
    return k;
}

// One Rice code (subframe.rs:337-341): returns the folded value u = (q << k) | r and advances pos.
// Slow path (rare): unary run or code longer than the 32-bit peek.
__device__ __forceinline__ uint32_t clx_lrice_slow(LaneReader& r, uint32_t k) {
    const uint32_t q = clx_lread_unary(r);
    const uint32_t rem = clx_lread(r, k);
    return (q << k) | rem;
}

// bps at which channel `ch` of a frame is coded (frame.rs:713-741)
__device__ __forceinline__ uint32_t clx_channel_bps(const clx_dev_frame& fr, uint32_t ch) {
    const uint32_t ca = fr.channel_assignment;
    uint32_t bps = fr.bps;
    if ((ca == CLX_CH_LEFT_SIDE && ch == 1u) || (ca == CLX_CH_RIGHT_SIDE && ch == 0u) || (ca == CLX_CH_MID_SIDE && ch == 1u)) bps += 1u;
    return bps;
}

__device__ __forceinline__ void clx_report_error(uint32_t* errkey, uint32_t frame, uint32_t ch, uint32_t err) {
    // the lowest channel's error is the first one in stream order
    atomicMin(&errkey[frame], (ch << 24) | (err & 0x00ffffffu));
}

// ------------------------------------------------------------------------------------------------
// Per-lane LDS ring: the next CLX_RING dwords of the lane's stream, byte-swapped, so that the per-code
// peek is one ds_read2_b32 + a funnel shift instead of a dependent global load (an L1 hit with 64
// divergent lines costs >300 cycles; the code chain is serial in the bit position).  The ring is
// filled split-phase with 16-byte loads: a granule requested at one block boundary is written to LDS
// at the next, so HBM/L2 latency never sits on the decode chain.  Slots CLX_RING..CLX_RING+7 mirror slots
// 0..7, so reading six consecutive dwords (one block's 160-bit register window) never wraps.
// ------------------------------------------------------------------------------------------------
#define CLX_RING 32u
#define CLX_ROW (CLX_RING + 8u)       // 40 dwords: rows stay 16-byte aligned, so a granule is one ds_write_b128
struct LanesLds {
    uint32_t ring[64][CLX_ROW];
    int4 stage[64][4];         // per lane: the last (up to) 16 output samples, flushed as one 64-byte segment
};

struct Ring {
    const uint32_t* src;      // arena + origin (16-byte aligned)
    uint32_t avail_dw;        // dwords readable from src
    uint32_t fill;            // stream dwords [fill-CLX_RING, fill) are in the ring; multiple of 4
    uint32_t npend;           // granules requested at the last pump (0..3), stored in pend0 / pend1 / pend2
    uint4 pend0, pend1, pend2;
    uint32_t fast_lim;        // a fast block may start at any bit position <= fast_lim (ring coverage and EOF margin)
};

__device__ __forceinline__ uint4 clx_ring_fetch(const Ring& g, uint32_t dw) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (dw + 4u <= g.avail_dw) v = *reinterpret_cast<const uint4*>(g.src + dw);
    return v;
}
__device__ __forceinline__ void clx_ring_put(uint32_t* row, uint32_t dw, const uint4 v) {
    const uint32_t s = dw & (CLX_RING - 1u);                 // multiple of 4: granules are 16-byte aligned in the stream
    const uint4 b = make_uint4(__builtin_bswap32(v.x), __builtin_bswap32(v.y), __builtin_bswap32(v.z), __builtin_bswap32(v.w));
    *reinterpret_cast<uint4*>(row + s) = b;
    if (s < 8u) *reinterpret_cast<uint4*>(row + CLX_RING + s) = b;      // slots 32..39 mirror slots 0..7
}
// A fast block (4 codes, <= 32 bits each, + partition parameters) reads at most 160+ bits: it may start at `pos` when
// the ring holds 8 more dwords and the readable stream 160 more bits.
__device__ __forceinline__ void clx_ring_set_lim(Ring& g, uint32_t limit) {
    const uint32_t by_ring = g.fill >= 8u ? 32u * (g.fill - 8u) : 0u;
    const uint32_t by_eof = limit >= 160u ? limit - 160u : 0u;
    g.fast_lim = by_ring < by_eof ? by_ring : by_eof;
}
// The lean blocks decode first and ask afterwards: `pw` = where their last register window was loaded, `pend` = where
// they stopped.  What they decoded is what the stream holds iff that window (6 dwords) was inside the ring's filled part
// and no code reached past the end of the frame.  (Worst-case margins instead -- 16 codes <= 512 bits -- would demand
// 24 of the ring's 32 dwords ahead of the position at every turn, which lanes above 8 bits per code cannot keep.)
__device__ __forceinline__ bool clx_ring_covered(const Ring& g, uint32_t pw, uint32_t pend, uint32_t limit) {
    return (pw >> 5) + 6u <= g.fill && pend <= limit;
}
// synchronous (re)fill starting at the granule that holds dword `dw` (start of a subframe, or after a jump)
__device__ __forceinline__ void clx_ring_reset(Ring& g, uint32_t* row, uint32_t dw, uint32_t limit) {
    const uint32_t f0 = dw & ~3u;
    uint4 t[CLX_RING / 4u];
#pragma unroll
    for (uint32_t q = 0; q < CLX_RING / 4u; ++q) t[q] = clx_ring_fetch(g, f0 + 4u * q);
#pragma unroll
    for (uint32_t q = 0; q < CLX_RING / 4u; ++q) clx_ring_put(row, f0 + 4u * q, t[q]);
    g.fill = f0 + CLX_RING;
    g.npend = 0;
    clx_ring_set_lim(g, limit);
}
// every 4th block: land the granules requested last time, request new ones while the ring has room.  The two halves can be
// called apart (clx_lanes_body: the landing -- the one place that waits on the vector-memory counter -- goes in front of the
// turn's stores, a whole turn after the loads were requested, so that it never waits for a store's round trip).
__device__ __forceinline__ void clx_ring_land(Ring& g, uint32_t* row) {
    if (g.npend >= 1u) { clx_ring_put(row, g.fill, g.pend0); g.fill += 4u; }
    if (g.npend >= 2u) { clx_ring_put(row, g.fill, g.pend1); g.fill += 4u; }
    if (g.npend >= 3u) { clx_ring_put(row, g.fill, g.pend2); g.fill += 4u; }
    g.npend = 0;
}
__device__ __forceinline__ void clx_ring_request(Ring& g, uint32_t* row, uint32_t pos, uint32_t limit) {
    const uint32_t dw = pos >> 5;
    if (dw + 8u > g.fill || dw + CLX_RING < g.fill) clx_ring_reset(g, row, dw, limit);       // ran dry, or the position jumped
    else {
        const uint32_t room = CLX_RING - (g.fill - dw);                                     // dwords that may be overwritten
        if (room >= 4u) { g.pend0 = clx_ring_fetch(g, g.fill); g.npend = 1u; }
        if (room >= 8u) { g.pend1 = clx_ring_fetch(g, g.fill + 4u); g.npend = 2u; }
        if (room >= 12u) { g.pend2 = clx_ring_fetch(g, g.fill + 8u); g.npend = 3u; }        // 384 bits per 16 codes: 24 bits per code sustained
        clx_ring_set_lim(g, limit);
    }
}
__device__ __forceinline__ void clx_ring_pump(Ring& g, uint32_t* row, uint32_t pos, uint32_t limit) {
    clx_ring_land(g, row);
    clx_ring_request(g, row, pos, limit);
}
__device__ __forceinline__ uint32_t clx_ring_peek32(const uint32_t* row, uint32_t pos) {
    const uint32_t s = (pos >> 5) & (CLX_RING - 1u);
    const uint64_t w = ((uint64_t)row[s] << 32) | row[s + 1u];
    return (uint32_t)((w << (pos & 31u)) >> 32);
}

// 160-bit register window over the stream, `a` = the next 32 bits.  Loaded from the ring once per block of four
// codes (three ds_read2_b32, one LDS round trip), then advanced with funnel shifts: no memory access per code.
struct Win { uint32_t a, b, c, d, e; };
__device__ __forceinline__ Win clx_win_load(const uint32_t* row, uint32_t pos) {
    const uint32_t s = (pos >> 5) & (CLX_RING - 1u);
    const uint32_t w0 = row[s], w1 = row[s + 1u], w2 = row[s + 2u], w3 = row[s + 3u], w4 = row[s + 4u], w5 = row[s + 5u];
    const uint32_t off = pos & 31u;
    const uint32_t sh = (32u - off) & 31u;
    Win w;
    w.a = off ? clx_alignbit(w0, w1, sh) : w0; w.b = off ? clx_alignbit(w1, w2, sh) : w1;
    w.c = off ? clx_alignbit(w2, w3, sh) : w2; w.d = off ? clx_alignbit(w3, w4, sh) : w3;
    w.e = off ? clx_alignbit(w4, w5, sh) : w4;
    return w;
}
// the same with 64-bit shifts: five instructions instead of five funnel shifts + five selects for off == 0
__device__ __forceinline__ Win clx_win_load64(const uint32_t* row, uint32_t pos) {
    const uint32_t s = (pos >> 5) & (CLX_RING - 1u);
    const uint32_t w0 = row[s], w1 = row[s + 1u], w2 = row[s + 2u], w3 = row[s + 3u], w4 = row[s + 4u], w5 = row[s + 5u];
    const uint32_t off = pos & 31u;
    Win w;
    w.a = (uint32_t)(((((uint64_t)w0 << 32) | w1) << off) >> 32); w.b = (uint32_t)(((((uint64_t)w1 << 32) | w2) << off) >> 32);
    w.c = (uint32_t)(((((uint64_t)w2 << 32) | w3) << off) >> 32); w.d = (uint32_t)(((((uint64_t)w3 << 32) | w4) << off) >> 32);
    w.e = (uint32_t)(((((uint64_t)w4 << 32) | w5) << off) >> 32);
    return w;
}
// the same from five dwords with funnel shifts only: the window starts in the dword that holds bit p - 1 (p >= 1: a frame header
// precedes every subframe), so the shift count (32 - p % 32) % 32 never has to be 32 -- v_alignbit takes the low five bits of -p
__device__ __forceinline__ Win clx_win_load5(const uint32_t* row, uint32_t pos) {
    const uint32_t s = ((pos - 1u) >> 5) & (CLX_RING - 1u);
    const uint32_t w0 = row[s], w1 = row[s + 1u], w2 = row[s + 2u], w3 = row[s + 3u], w4 = row[s + 4u];
    const uint32_t sh = 0u - pos;
    Win w;
    w.a = clx_alignbit(w0, w1, sh); w.b = clx_alignbit(w1, w2, sh); w.c = clx_alignbit(w2, w3, sh); w.d = clx_alignbit(w3, w4, sh);
    w.e = 0u;
    return w;
}
// drop nb (1..32) bits
__device__ __forceinline__ void clx_win_skip(Win& w, uint32_t nb) {
    const uint32_t sh = 32u - nb;                         // alignbit uses sh & 31: nb = 32 -> whole-register move
    w.a = clx_alignbit(w.a, w.b, sh); w.b = clx_alignbit(w.b, w.c, sh); w.c = clx_alignbit(w.c, w.d, sh);
    w.d = clx_alignbit(w.d, w.e, sh); w.e = clx_alignbit(w.e, 0u, sh);
}

// ------------------------------------------------------------------------------------------------
// P: locate subframes 1..C-1 of every multi-channel frame
// ------------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(64)
void clx_k_scan_general(const clx_runs runs, const clx_dev_frame* __restrict__ frames,
                        const uint32_t* __restrict__ multi, uint32_t n_multi) {
    __shared__ struct { uint32_t ring[64][CLX_ROW]; } L;        // (the ring only: 10 KiB per wave instead of LanesLds' 14 -- more of them fit beside the decode waves)
    const clx_run& R = runs.r[blockIdx.y];
    const uint8_t* const arena = R.arena;
    const uint64_t arena_alloc_len = R.alloc_len;
    uint32_t* const sf_start = R.sf_start;
    uint32_t* const errkey = R.errkey;
    CLX_TL_BEGIN();
    const int lane = (int)threadIdx.x;
    uint32_t* const row = L.ring[lane];
    const uint32_t t = blockIdx.x * 64u + (uint32_t)lane;
    const bool active = t < n_multi;
    const uint32_t f = active ? multi[t] : 0u;
    clx_dev_frame fr;
    fr.byte_off = 0; fr.out_off = 0; fr.limit_bits = 0; fr.first_slot = 0; fr.header_bytes = 0; fr.block_size = 0;
    fr.n_channels = 0; fr.channel_assignment = 0; fr.bps = 1; fr.flags = 0;
    if (active) fr = frames[f];
    LaneReader r;
    r.arena = arena;
    r.origin = (uint32_t)(fr.byte_off & ~15ull);
    const uint32_t o = 8u * (uint32_t)(fr.byte_off & 15ull);
    r.limit = o + fr.limit_bits;
    r.pos = o + 8u * (uint32_t)fr.header_bytes;
    r.err = active ? ((r.pos > r.limit) ? CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF) : 0u) : 1u;
    Ring g;
    g.src = reinterpret_cast<const uint32_t*>(arena + r.origin);
    g.avail_dw = (uint32_t)((arena_alloc_len > r.origin ? arena_alloc_len - r.origin : 0ull) >> 2);
    g.fill = 0; g.npend = 0; g.pend0 = make_uint4(0u, 0u, 0u, 0u); g.pend1 = g.pend0; g.pend2 = g.pend0; g.fast_lim = 0;
    const uint32_t bs = fr.block_size;
    uint32_t nch = active ? (uint32_t)fr.n_channels - 1u : 0u;       // channels to scan
    uint32_t nch_max = nch;
#pragma unroll
    for (int sx = 32; sx >= 1; sx >>= 1) { const uint32_t a = __shfl_xor(nch_max, sx, 64); nch_max = a > nch_max ? a : nch_max; }

    for (uint32_t ch = 0; ch < nch_max; ++ch) {
        const bool on = ch < nch && !r.err;
        // ---- headers (per lane, generic reader)
        uint32_t codes = 0, first = 0, per = 0, parts_left = 0, rice2 = 0, order = 0;
        if (on) {
            const SfHead h = clx_lparse_sf_header(r, clx_channel_bps(fr, ch));
            if (!r.err) {
                if (h.kind == 0u) (void)clx_lread(r, h.sf_bps);
                else if (h.kind == 1u) {
                    if ((uint64_t)r.pos + (uint64_t)bs * h.sf_bps > (uint64_t)r.limit) r.err = CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
                    else r.pos += bs * h.sf_bps;
                } else {
                    if (bs < h.order) r.err = CLX_LERR(CLX_FORMAT_ERROR, h.kind == 2u ? CLX_MSG_FIXED_ORDER_GT_BLOCK : CLX_MSG_LPC_ORDER_GT_BLOCK);
                    if (!r.err) {
                        if (r.pos + h.order * h.sf_bps > r.limit) r.err = CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
                        else r.pos += h.order * h.sf_bps;
                    }
                    if (!r.err && h.kind == 3u) {
                        const uint32_t pm1 = clx_lread(r, 4);
                        if (!r.err && pm1 == 15u) r.err = CLX_LERR(CLX_FORMAT_ERROR, CLX_MSG_QLP_PRECISION_INVALID);
                        const uint32_t sh = clx_lread(r, 5);
                        if (!r.err && (sh & 0x10u)) r.err = CLX_LERR(CLX_UNSUPPORTED, CLX_MSG_NEGATIVE_QLP_SHIFT);
                        if (!r.err) {
                            if (r.pos + h.order * (pm1 + 1u) > r.limit) r.err = CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
                            else r.pos += h.order * (pm1 + 1u);
                        }
                    }
                    if (!r.err) {
                        const ResHead rh = clx_lparse_residual_header(r, bs, h.order);
                        if (!r.err) { codes = bs - h.order; first = rh.per - h.order; per = rh.per; parts_left = rh.n_part; rice2 = rh.rice2; order = h.order; }
                    }
                }
            }
        }
        // ---- all Rice codes of this subframe: only their lengths matter here
        uint32_t left = r.err ? 0u : codes;                 // codes still to skip
        uint32_t pcnt = 0, next_cnt = first, k = 0, k1 = 1;
        // one careful step (rolled where it is used): empty partitions, escape codes, long runs, EOF
        auto careful_step = [&]() {
            while (!r.err && pcnt == 0u && parts_left != 0u) {
                k = clx_lread_rice_param(r, rice2); k1 = k + 1u; parts_left -= 1u; pcnt = next_cnt; next_cnt = per;
            }
            if (!r.err) {
                const uint32_t v = clx_lpeek32(r, r.pos);
                const uint32_t nb = (uint32_t)__clz((int)v) + k1;
                if (v != 0u && nb <= 32u) {
                    r.pos += nb;
                    if (r.pos > r.limit) r.err = CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
                } else (void)clx_lrice_slow(r, k);
                pcnt -= 1u; left -= 1u;
            }
        };
        // Partitions end at multiples of their length in SAMPLES: the first (-order & 3) codes go one by one, so that the
        // blocks of four below start on multiples of 4 samples and meet partition edges only at their start -- in every
        // lane, whatever the orders of the subframes the lanes scan.
        {
            uint32_t pre = (0u - order) & 3u;
            pre = pre < left ? pre : left;
#pragma unroll 1
            for (; __any(pre != 0u); ) {
                if (pre != 0u) { careful_step(); pre = r.err ? 0u : pre - 1u; }
            }
            if (r.err) left = 0u;
        }
        uint32_t lmax = left;
#pragma unroll
        for (int sx = 32; sx >= 1; sx >>= 1) { const uint32_t a = __shfl_xor(lmax, sx, 64); lmax = a > lmax ? a : lmax; }
        if (lmax != 0u) clx_ring_reset(g, row, r.pos >> 5, r.limit);
        for (uint32_t i0 = 0; i0 < lmax; i0 += 4u) {
            if ((i0 & 12u) == 0u && i0 != 0u) clx_ring_pump(g, row, r.pos, r.limit);
            const bool busy = left >= 4u;
            // lean blocks: NB x 4 codes, each four inside ONE partition (blocks start on multiples of 4 samples, partitions on
            // multiples of their length: a partition may START with a four, its parameter is read then) -- count zeros, add,
            // shift the window, nothing else; one vote.  Sixteen codes at a time where the pump's cadence allows, else four.
            auto lean = [&](auto nb_tag) -> bool {
                constexpr uint32_t NC = 4u * (uint32_t)decltype(nb_tag)::value;
                const bool go = left >= NC;
                uint32_t p = r.pos, mx = 0, kq = k, k1q = k1, pc = pcnt, nx = next_cnt, pl = parts_left, pw = 0;
                bool bad = false;
#pragma unroll
                for (uint32_t b4 = 0; b4 < NC / 4u; ++b4) {
                    const bool at = go && pc == 0u;             // a partition starts here: its parameter comes first
                    if (__any(at)) {
                        const uint32_t pv = clx_ring_peek32(row, p);
                        if (at) {
                            const uint32_t pb = rice2 ? 5u : 4u;
                            kq = pv >> (32u - pb);
                            bad = bad || kq == (rice2 ? 31u : 15u) || pl == 0u || nx == 0u;
                            k1q = kq + 1u; p += pb; pl -= 1u; pc = nx; nx = per;
                        }
                    }
                    bad = bad || pc < 4u;                       // (a partition edge inside the four codes: the general block's)
                    pc -= 4u;
                    pw = p;
                    Win w = clx_win_load5(row, p);
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const uint32_t nb = (uint32_t)__clz((int)w.a) + k1q;   // 32 + k1 when the window is all zeros
                        mx = nb > mx ? nb : mx;
                        clx_win_skip(w, nb);
                        p += nb;
                    }
                }
                // (a lane with fewer codes left sends the wave on to the smaller blocks; the general block finishes tails)
                const bool lean_ok = go ? (!bad && clx_ring_covered(g, pw, p, r.limit) && mx <= 32u) : (left == 0u || r.err != 0u);
                const bool all = __all(lean_ok);
                if (all && go) { r.pos = p; k = kq; k1 = k1q; pcnt = pc; next_cnt = nx; parts_left = pl; left -= NC; }
                if (NC == 4u && !all) {
                    CLX_STAT(2, go && bad); CLX_STAT(4, go && !clx_ring_covered(g, pw, p, r.limit)); CLX_STAT(5, go && mx > 32u);
                    CLX_STAT(6, !go && !(left == 0u || r.err != 0u));
                }
                return all;
            };
            if ((i0 & 12u) == 0u && lean(std::integral_constant<int, 4>())) { CLX_STAT(8, 1); i0 += 12u; continue; }
            if (lean(std::integral_constant<int, 1>())) { CLX_STAT(0, 1); continue; }
            CLX_STAT(1, 1);
            // general block: 4 codes from a register window with partition parameters in between, no EOF possible, every
            // code <= 32 bits; committed only if every lane stayed on the common path
            uint32_t pos2 = r.pos, pcnt2 = pcnt, k_2 = k, k1_2 = k1, parts2 = parts_left, next2 = next_cnt;
            bool ok = !busy || r.pos <= g.fast_lim;
            {
                Win w = clx_win_load(row, pos2);
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    if (__any(busy && pcnt2 == 0u)) {
                        if (pcnt2 == 0u) {                      // partition parameter (subframe.rs:314-319 / 362-367)
                            const uint32_t pb = rice2 ? 5u : 4u;
                            k_2 = w.a >> (32u - pb);
                            if (k_2 == (rice2 ? 31u : 15u) || parts2 == 0u || next2 == 0u) ok = false;
                            k1_2 = k_2 + 1u; pos2 += pb; parts2 -= 1u; pcnt2 = next2; next2 = per;
                            clx_win_skip(w, pb);
                        }
                    }
                    const uint32_t nb = (uint32_t)__clz((int)w.a) + k1_2;      // 32 + k1 when the window is all zeros
                    if (nb > 32u) ok = false;
                    clx_win_skip(w, nb);
                    pos2 += nb; pcnt2 -= 1u;
                }
            }
            const bool all_ok = __all(ok || !busy);
            CLX_STAT(7, all_ok);
            if (all_ok && busy) { r.pos = pos2; pcnt = pcnt2; k = k_2; k1 = k1_2; parts_left = parts2; next_cnt = next2; left -= 4u; }
            const bool tail = !r.err && left != 0u && left < 4u;
            if (!all_ok || __any(tail)) {
                // careful steps (rolled): empty partitions, escape codes, long runs, EOF, tails
#pragma unroll 1
                for (int ii = 0; ii < 4; ++ii) {
                    if (!r.err && left != 0u && (!all_ok || left < 4u)) careful_step();
                }
            }
        }
        // partition parameters that belong to empty partitions at the very end (order == block size: subframe.rs:509, 706)
        if (on && !r.err) {
            while (!r.err && parts_left != 0u) { (void)clx_lread_rice_param(r, rice2); parts_left -= 1u; }
        }
        if (on) {
            if (r.err) clx_report_error(errkey, f, ch, r.err);
            else sf_start[fr.first_slot + ch + 1u] = r.pos;
        }
    }
    CLX_TL_END(2, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// D: one lane per subframe, everything fused
// ------------------------------------------------------------------------------------------------
template <int OMAX>
struct LaneState {
    LaneReader r;
    uint32_t phase;            // 0 fixed-width fields (warm-up / verbatim), 1 rice, 2 constant, 3 idle
    int32_t  cval;
    uint32_t trans_at;         // sample index at which a predicted subframe switches to residuals
    bool     transitioned;
    uint32_t order, shift;     // predictor (active from the transition on)
    int32_t  lim;              // |s| range in which the 24-bit / i32 evaluation is exact; -1: use i64
    uint32_t k, k1, pcnt, next_cnt, per, parts_left, rice2;
    int32_t  c[OMAX], hist[OMAX];
};

template <int OMAX, bool WIDE>
__device__ __forceinline__ int32_t clx_lpredict(const int32_t (&c)[OMAX], const int32_t (&hist)[OMAX], uint32_t shift) {
    if (WIDE) {
        int64_t acc = 0;
#pragma unroll
        for (int j = OMAX - 1; j >= 0; --j) acc += (int64_t)c[j] * (int64_t)hist[j];
        return (int32_t)(acc >> shift);
    } else {
        return clx_dot24z<OMAX>(c, hist) >> shift;                              // v_mad_i32_i24 chain, newest tap last
    }
}

// LPC parameters + residual header (subframe.rs:669-701, 241-277): once per predicted subframe, at the sample index
// where the warm-up ends -- also when that index is the block size (no residual samples at all: the residual
// header and its partition parameter are still in the stream, subframe.rs:509, 706).
template <int OMAX>
__device__ __forceinline__ void clx_ltransition(LaneState<OMAX>& S, const SfHead& h, uint32_t bs) {
    LaneReader& r = S.r;
    uint32_t cabs = 0;
    if (h.kind == 3u) {
        const uint32_t pm1 = clx_lread(r, 4);
        if (!r.err && pm1 == 15u) r.err = CLX_LERR(CLX_FORMAT_ERROR, CLX_MSG_QLP_PRECISION_INVALID);
        const uint32_t sh = clx_lread(r, 5);
        if (!r.err && (sh & 0x10u)) r.err = CLX_LERR(CLX_UNSUPPORTED, CLX_MSG_NEGATIVE_QLP_SHIFT);
        S.shift = sh & 0xfu;
        if (!r.err && r.pos + h.order * (pm1 + 1u) > r.limit) r.err = CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
#pragma unroll
        for (int j = 0; j < OMAX; ++j) {           // j-th coded coefficient applies to s[i-1-j] (subframe.rs:696-701)
            if ((uint32_t)j < h.order && !r.err) S.c[j] = clx_lread_signed(r, pm1 + 1u);
            cabs += (uint32_t)(S.c[j] < 0 ? -S.c[j] : S.c[j]);
        }
    } else {                                       // fixed predictors as taps on s[i-1-j] (subframe.rs:427-431)
        const uint32_t o = h.order;
        S.c[0] = o == 1u ? 1 : o == 2u ? 2 : o == 3u ? 3 : o == 4u ? 4 : 0;
        S.c[1] = o == 2u ? -1 : o == 3u ? -3 : o == 4u ? -6 : 0;
        S.c[2] = o == 3u ? 1 : o == 4u ? 4 : 0;
        S.c[3] = o == 4u ? -1 : 0;
        cabs = o == 1u ? 1u : o == 2u ? 3u : o == 3u ? 7u : o == 4u ? 15u : 0u;
    }
    S.order = h.order;
    // 24-bit evaluation (v_mad_i32_i24 chains, i32 accumulate) == the reference's i64 evaluation (subframe.rs:464-505) while
    // every history value s has |s| <= lim <= 2^23 with sum|c| * lim < 2^31: the factors fit 24 bits and no partial sum
    // wraps.  Which samples are inside is checked on the data, not assumed from the bit depth: streams whose sum|c| * 2^(bps-1)
    // reaches 2^31 (high precision, loud side channels) still run in 24 bits wherever the signal allows.
    {
        const uint32_t by_sum = cabs != 0u ? 0x7fffffffu / cabs : 0x7fffffffu;
        S.lim = (int32_t)(by_sum < (1u << 23) ? by_sum : (1u << 23));
    }
    const ResHead rh = clx_lparse_residual_header(r, bs, h.order);
    S.rice2 = rh.rice2; S.per = rh.per; S.parts_left = rh.n_part;
    S.pcnt = 0; S.next_cnt = rh.per - h.order;     // the first partition holds per - order codes (subframe.rs:283)
    S.phase = 1u;
    S.transitioned = true;
}

// One sample the careful way (generic reader over global memory, every rare case handled in line):
// returns the sample's raw value (residual, warm-up / verbatim sample, or the constant).
template <int OMAX>
__device__ __forceinline__ int32_t clx_lcareful_raw(LaneState<OMAX>& S, const SfHead& h, uint32_t bs, uint32_t i, uint32_t n) {
    LaneReader& r = S.r;
    if (n != 0u && !r.err && i == S.trans_at) clx_ltransition<OMAX>(S, h, bs);
    int32_t x = S.cval;
    if (i < n && !r.err) {
        if (S.phase == 0u) x = clx_lread_signed(r, h.sf_bps);                      // warm-up / verbatim (subframe.rs:397-415)
        else if (S.phase == 1u) {
            while (!r.err && S.pcnt == 0u && S.parts_left != 0u) {                   // partition parameter(s)
                S.k = clx_lread_rice_param(r, S.rice2); S.k1 = S.k + 1u; S.parts_left -= 1u; S.pcnt = S.next_cnt; S.next_cnt = S.per;
            }
            if (!r.err) {
                const uint32_t v = clx_lpeek32(r, r.pos);
                const uint32_t z = (uint32_t)__clz((int)v);
                const uint32_t nb = z + S.k1;
                uint32_t u;
                if (v != 0u && nb <= 32u) {
                    const uint32_t rem = (v >> ((32u - nb) & 31u)) & ((1u << S.k) - 1u);
                    u = (z << S.k) | rem;
                    r.pos += nb;
                    if (r.pos > r.limit) r.err = CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
                } else u = clx_lrice_slow(r, S.k);
                x = (int32_t)(u >> 1) ^ -(int32_t)(u & 1u);                          // rice_to_signed (subframe.rs:157-170)
                S.pcnt -= 1u;
            }
        }
    }
    return x;
}

// wasted-bits shift (subframe.rs:216-225) + stereo decorrelation (frame.rs:319-389) of one sample; wave-uniform call
struct Finish {
    uint32_t wasted, sgn;             // sgn: all ones in odd lanes ((x ^ sgn) - sgn = odd ? -x : x)
    // per-lane constants of the generic formula  v = ((P << s1 | R & bit) + (R ^ sg) - sg) >> s1  (see clx_k_predict)
    uint32_t rmask, s1, bit, sg;
    bool p_other, r_other;
    bool any_decor, all_ms, any_wasted;
    // per-lane constants of clx_decor4's form  (even & pmask) + ((((odd ^ dsg) & drm) + dc) >> s1)
    uint32_t dsg, drm, dc, pmask;
    // per-lane constants of clx_decor4_mad's form  (own * mo + other * mt + mc) >> 1, the wasted-bits shifts included (the 16-bit tier's)
    int32_t mo, mt, mc;
    bool ms_plain;                    // every lane in a mid/side pair and no wasted bits: clx_ms_short4
};
__device__ __forceinline__ Finish clx_lfinish_setup(uint32_t n, uint32_t wasted, uint32_t decor, bool pair_ok, int lane) {
    Finish F;
    const bool odd = (lane & 1) != 0;
    F.wasted = wasted; F.sgn = odd ? 0xffffffffu : 0u;
    const bool d_ms = pair_ok && decor == CLX_CH_MID_SIDE;
    const bool d_ls = pair_ok && decor == CLX_CH_LEFT_SIDE && odd;       // side channel -> right = left - side (frame.rs:327-330)
    const bool d_rs = pair_ok && decor == CLX_CH_RIGHT_SIDE && !odd;     // side channel -> left = side + right (frame.rs:352-355)
    F.p_other = (d_ms && odd) || d_ls;
    F.r_other = (d_ms && !odd) || d_rs;
    F.rmask = (d_ms || d_ls || d_rs) ? 0xffffffffu : 0u;
    F.s1 = d_ms ? 1u : 0u; F.bit = F.s1;
    F.sg = ((d_ms && odd) || d_ls) ? 0xffffffffu : 0u;
    F.drm = odd ? 0xffffffffu : (d_ms || d_rs) ? 0xffffffffu : 0u;            // (an odd lane's own value is "the odd one")
    F.dsg = (odd && (d_ms || d_ls)) ? 0xffffffffu : 0u;
    F.dc = d_ms ? (odd ? 2u : 1u) : d_ls ? 1u : 0u;
    F.pmask = (!odd || d_ms || d_ls) ? 0xffffffffu : 0u;
    F.any_decor = __any(pair_ok);
    F.all_ms = __all(n == 0u || d_ms);        // idle lanes (the tail of the last wave) do not spoil the short sequence
    F.any_wasted = __any(n != 0u && wasted != 0u);
    F.ms_plain = F.all_ms && !F.any_wasted;
    const uint32_t wo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wasted, 0xB1, 0xF, 0xF, false);      // the partner's wasted bits (lane ^ 1)
    const int32_t own2 = (int32_t)(2u << wasted), other2 = (int32_t)(2u << wo);
    F.mo = d_ms ? (odd ? -(own2 >> 1) : own2) : d_ls ? -own2 : own2;
    F.mt = d_ms ? (odd ? other2 : (other2 >> 1)) : (d_ls || d_rs) ? other2 : 0;
    F.mc = d_ms ? 1 : 0;
    return F;
}
__device__ __forceinline__ int32_t clx_lfinish(int32_t s, const Finish& F) {
    int32_t mine = F.any_wasted ? (int32_t)((uint32_t)s << F.wasted) : s;
    if (F.all_ms) {
        // every lane belongs to a mid/side pair: left = (m + side) >> 1, right = (m - side) >> 1 (frame.rs:382-384)
        return clx_ms_pair(mine, F.sgn, F.sgn & 1u, 1u);
    }
    if (F.any_decor) {
        const int32_t other = __builtin_amdgcn_update_dpp(0, mine, 0xB1, 0xF, 0xF, false);       // lane ^ 1
        const uint32_t P = (uint32_t)(F.p_other ? other : mine);
        const uint32_t R = (uint32_t)(F.r_other ? other : mine) & F.rmask;
        const uint32_t m = (P << F.s1) | (R & F.bit);
        mine = (int32_t)(m + ((R ^ F.sg) - F.sg)) >> F.s1;
    }
    return mine;
}

// ---- the lean blocks' common parts ------------------------------------------------------------------------------------
// What a lane is doing while the lean blocks run (fixed once the prologue is over): Rice codes, verbatim fields, or a
// constant.  Lanes of all three kinds run the same straight-line code; masks pick what applies.
struct LeanKind {
    bool rice, verb;
    uint32_t bitmask;        // all ones where the lane consumes bits (Rice, verbatim), 0 for a constant
    uint32_t ricemask;       // all ones for Rice lanes: only their code lengths can exceed the window
    uint32_t cor;            // the constant, OR-ed in for constant lanes
    uint32_t vbits, vsh;     // verbatim field width and the shift that sign-extends it
};
template <int OMAX>
__device__ __forceinline__ LeanKind clx_lean_kind(const LaneState<OMAX>& S, const SfHead& h) {
    LeanKind K;
    K.rice = S.phase == 1u; K.verb = S.phase == 0u;
    K.bitmask = S.phase == 2u ? 0u : 0xffffffffu;
    K.ricemask = K.rice ? 0xffffffffu : 0u;
    K.cor = S.phase == 2u ? (uint32_t)S.cval : 0u;
    K.vbits = h.sf_bps; K.vsh = (32u - h.sf_bps) & 31u;
    return K;
}
// The cursor a lean block works on (committed only if the wave's vote passes).
struct LeanCur { uint32_t p, k, k1, pcnt, next, parts; bool bad; };
// A partition that starts exactly where the block starts: its parameter is read first (subframe.rs:314-319 / 362-367).
// Partitions are a multiple of 4 (16) samples long in every stream whose block size allows it, and the prologue leaves
// every lane on a multiple of 16 -- so this is where partition boundaries fall, also when the lanes of a wave decode
// subframes of different shapes.  `any_at` (a wave vote by the caller) keeps the LDS read off the common path.
template <int OMAX>
__device__ __forceinline__ LeanCur clx_lean_begin(const LaneState<OMAX>& S, const uint32_t* ringrow, bool live, const LeanKind& K) {
    LeanCur c = { S.r.pos, S.k, S.k1, S.pcnt, S.next_cnt, S.parts_left, false };
    const bool at = live && K.rice && S.pcnt == 0u;
    if (__any(at)) {
        const uint32_t pv = clx_ring_peek32(ringrow, S.r.pos);
        if (at) {
            const uint32_t pb = S.rice2 ? 5u : 4u;
            c.k = pv >> (32u - pb);
            c.bad = c.k == (S.rice2 ? 31u : 15u) || c.parts == 0u || c.next == 0u;
            c.k1 = c.k + 1u; c.p += pb; c.parts -= 1u; c.pcnt = c.next; c.next = S.per;
        }
    }
    return c;
}
// N (a multiple of 4) values from register windows of 4.  MODE (wave-uniform): 0 every live lane reads Rice codes, 1 some
// repeat a constant, 2 some read verbatim fields -- the masks that let the kinds share one instruction stream cost four
// instructions per code, which waves of Rice lanes only (nearly all of them) need not pay.
template <int N, int MODE>
__device__ __forceinline__ void clx_lean_codes_m(const uint32_t* ringrow, const LeanKind& K, uint32_t k, uint32_t k1, uint32_t& p, uint32_t& mx,
                                                 uint32_t& pw, int32_t (&X)[N]) {
    const uint32_t kk = k & 31u;
#pragma unroll
    for (int b4 = 0; b4 < N / 4; ++b4) {
        pw = p;
        Win w = clx_win_load64(ringrow, p);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const uint32_t z = (uint32_t)__clz((int)w.a);                  // 32 when the window is all zeros
            const uint32_t nbr = z + k1;
            const uint32_t u = (z << kk) | clx_bfe(w.a, 32u - nbr, k);     // (q << k) | r, subframe.rs:337-341
            int32_t x = (int32_t)(u >> 1) ^ -(int32_t)(u & 1u);            // rice_to_signed (subframe.rs:157-170)
            uint32_t nb = nbr;
            if (MODE == 2) {                                               // verbatim rows ride along (subframe.rs:397-415)
                x = K.verb ? ((int32_t)w.a >> K.vsh) : x;
                nb = K.verb ? K.vbits : nbr;
            }
            if (MODE == 0) {
                X[4 * b4 + ii] = x;
                mx = nbr > mx ? nbr : mx;
                clx_win_skip(w, nb);
                p += nb;
            } else {
                X[4 * b4 + ii] = (int32_t)(((uint32_t)x & K.bitmask) | K.cor);
                const uint32_t nbx = nbr & K.ricemask;
                mx = nbx > mx ? nbx : mx;
                clx_win_skip(w, nb);
                p += nb & K.bitmask;
            }
        }
    }
}
template <int N>
__device__ __forceinline__ void clx_lean_codes(int mode, const uint32_t* ringrow, const LeanKind& K, uint32_t k, uint32_t k1, uint32_t& p,
                                               uint32_t& mx, uint32_t& pw, int32_t (&X)[N]) {
    if (mode == 0)      clx_lean_codes_m<N, 0>(ringrow, K, k, k1, p, mx, pw, X);
    else if (mode == 1) clx_lean_codes_m<N, 1>(ringrow, K, k, k1, p, mx, pw, X);
    else                clx_lean_codes_m<N, 2>(ringrow, K, k, k1, p, mx, pw, X);
}
// which of them a wave needs (wave-uniform)
__device__ __forceinline__ int clx_lean_mode(bool on, const LeanKind& K, uint32_t phase) {
    return __any(on && K.verb) ? 2 : __any(on && phase == 2u) ? 1 : 0;
}

template <int OMAX>
__device__ __forceinline__ void clx_lanes_body(LaneState<OMAX>& S, Ring& g, uint32_t* ringrow, int4* stage, const SfHead h, uint32_t bs, uint32_t n,
                                               uint32_t decor, bool pair_ok, int32_t* __restrict__ row, bool row_aligned,
                                               uint32_t nmax, uint32_t omax, int lane, int32_t* __restrict__ out, int32_t* __restrict__ dump) {
    LaneReader& r = S.r;
    const Finish F = clx_lfinish_setup(n, h.wasted, decor, pair_ok, lane);

    // ---- careful prologue: warm-up samples, the transition, the first residuals (rolled loop, one sample per turn)
    uint32_t i0 = (omax + 4u + 15u) & ~15u;              // multiple of 16: output segments are flushed 64 bytes at a time
    if (i0 > nmax) i0 = (nmax + 15u) & ~15u;
#pragma unroll 1
    for (uint32_t i = 0; i < i0; ++i) {
        const int32_t x = clx_lcareful_raw<OMAX>(S, h, bs, i, n);
        const int32_t pred = clx_lpredict<OMAX, true>(S.c, S.hist, S.shift);
        const uint32_t use = (S.order != 0u && i >= S.order && i >= S.trans_at) ? 0xffffffffu : 0u;
        const int32_t s = (int32_t)((uint32_t)x + ((uint32_t)pred & use));
#pragma unroll
        for (int j = OMAX - 1; j > 0; --j) S.hist[j] = S.hist[j - 1];
        S.hist[0] = s;
        const int32_t v = clx_lfinish(s, F);
        if (i < n) row[i] = v;
    }
    // ---- steady state: blocks of 4 samples through the LDS ring, rolled back to careful steps when anything is unusual
    if (i0 < nmax) clx_ring_reset(g, ringrow, r.pos >> 5, r.limit);
    // The 24-bit predictor is exact while the history it reads lies inside [-lim, lim) (see clx_lsetup_predictor).  The
    // blocks below range-check their OUTPUTS, which become the history of the next block; `good` counts how many of the most
    // recent samples were inside (saturating), so `good >= order` vouches for a block's starting history.  Blocks that
    // evaluated in 24 bits and were accepted keep it true; the i64 evaluations (prologue, wide blocks, careful steps) check
    // nothing, so `good` is recounted after them -- and the 24-bit blocks resume once the signal is back inside the range.
    uint32_t good;
    {
        bool hist_in = S.lim >= 0;
#pragma unroll
        for (int j = 0; j < OMAX; ++j) hist_in = hist_in && S.hist[j] < S.lim && S.hist[j] >= -S.lim;
        good = hist_in ? 64u : 0u;
    }
    bool wide = n != 0u && !r.err && S.order != 0u && good < S.order;    // this lane needs the i64 predictor for its next block
    bool no_lean = __any(wide);                          // wave-uniform: the lean blocks (24-bit only) cannot run now
    // rows that allow 16-byte accesses leave through the 64 B x 16 rows store shape (see K2 in clx_kernels.hip), a whole
    // turn of 16 samples at a time
    const bool al16 = __all(n == 0u || (row_aligned && (n & 3u) == 0u));
    K2Slot ms; ms.d = nullptr; ms.row = row; ms.n = n; ms.order = 0; ms.shift = 0; ms.wasted = 0; ms.decor = 0; ms.lim_log2 = 0; ms.pair_ok = false;
    K2Mover M; M.init(out, ms, lane);
    int4* const tile = stage - 4 * lane;                 // the wave's 64 x 4 staging slots seen as one tile
    const uint32_t sw = ((uint32_t)lane >> 2) & 3u;
    const LeanKind K = clx_lean_kind<OMAX>(S, h);         // (the prologue is over: no lane changes its kind any more)
    const int lean_mode = clx_lean_mode(n != 0u && !r.err, K, S.phase);
    for (uint32_t t0 = i0; t0 < nmax; t0 += 4u) {
        if ((t0 & 12u) == 0u && t0 != i0) { clx_ring_land(g, ringrow); clx_ring_request(g, ringrow, r.pos, r.limit); }   // (landed already after a lean turn)
        const bool live = (n != 0u) && !r.err && t0 < n;
        // ---- lean turn: sixteen samples at once when every live lane is in the middle of a Rice partition (or repeats a
        //      constant), nothing is near an edge and the 24-bit predictor holds: four register windows, ONE vote.
        if ((t0 & 12u) == 0u && !no_lean && al16) {
            LeanCur c = clx_lean_begin<OMAX>(S, ringrow, live, K);
            uint32_t mx = 0, pw = 0;
            int32_t Y[16];
            clx_lean_codes<16>(lean_mode, ringrow, K, c.k, c.k1, c.p, mx, pw, Y);
            int32_t hh[OMAX];
#pragma unroll
            for (int j = 0; j < OMAX; ++j) hh[j] = S.hist[j];
#pragma unroll
            for (int ii = 0; ii < 16; ++ii) {
                const int32_t pred = clx_lpredict<OMAX, false>(S.c, hh, S.shift);      // taps beyond the order are zero
                const int32_t sm = (int32_t)((uint32_t)Y[ii] + (uint32_t)pred);
#pragma unroll
                for (int j = OMAX - 1; j > 0; --j) hh[j] = hh[j - 1];
                hh[0] = sm;
                Y[ii] = sm;
            }
            int32_t hi = Y[0], lo = Y[0];
#pragma unroll
            for (int ii = 1; ii + 1 < 16; ii += 2) { hi = clx_max3(hi, Y[ii], Y[ii + 1]); lo = clx_min3(lo, Y[ii], Y[ii + 1]); }
            hi = Y[15] > hi ? Y[15] : hi; lo = Y[15] < lo ? Y[15] : lo;
            const bool in_range = S.order == 0u || (hi < S.lim && lo >= -S.lim);
            const bool ok16 = !live || ((K.rice ? (S.transitioned && !c.bad && c.pcnt >= 16u && mx <= 32u)
                                                : K.verb ? true : S.phase == 2u)
                                        && (S.phase == 2u || clx_ring_covered(g, pw, c.p, r.limit)) && t0 + 16u <= n && S.lim >= 0 && in_range);
            if (!__all(ok16)) {
                CLX_STAT(17, 1);
                CLX_STAT(18, live && K.rice && !S.transitioned); CLX_STAT(19, live && K.rice && c.bad); CLX_STAT(20, live && K.rice && c.pcnt < 16u);
                CLX_STAT(21, live && !(S.phase == 2u) && !clx_ring_covered(g, pw, c.p, r.limit)); CLX_STAT(22, live && mx > 32u); CLX_STAT(23, live && t0 + 16u > n);
                CLX_STAT(24, live && S.lim < 0); CLX_STAT(25, live && !in_range); CLX_STAT(26, live && S.phase == 3u);
            }
            if (__all(ok16)) {
                if (live) {
                    r.pos = c.p;
                    if (K.rice) { S.k = c.k; S.k1 = c.k1; S.pcnt = c.pcnt - 16u; S.next_cnt = c.next; S.parts_left = c.parts; }
#pragma unroll
                    for (int j = 0; j < OMAX; ++j) S.hist[j] = hh[j];
                }
#pragma unroll
                for (int ii = 0; ii < 16; ++ii) Y[ii] = clx_lfinish(Y[ii], F);
#pragma unroll
                for (uint32_t q = 0; q < 4u; ++q) tile[(uint32_t)lane * 4u + (q ^ sw)] = make_int4(Y[4 * q], Y[4 * q + 1], Y[4 * q + 2], Y[4 * q + 3]);
                clx_wave_sync();
                const uint32_t t = t0 + 4u * M.pc;
                const int4 w0 = tile[lane], w1 = tile[64 + lane], w2 = tile[128 + lane], w3 = tile[192 + lane];
                // The granules requested when the turn began are landed HERE (the wait on the vector-memory counter that goes
                // with it is a turn old), and the stores are an asm statement the compiler keeps no count of: otherwise the next
                // write of any register the stores read would wait for their round trip (clx_store4x16, clx_intrin.h).
                clx_ring_land(g, ringrow);
                clx_store4x16(t < M.rn[0] ? const_cast<int32_t*>(M.rp[0]) + t : dump + 0, t < M.rn[1] ? const_cast<int32_t*>(M.rp[1]) + t : dump + 4,
                              t < M.rn[2] ? const_cast<int32_t*>(M.rp[2]) + t : dump + 8, t < M.rn[3] ? const_cast<int32_t*>(M.rp[3]) + t : dump + 12,
                              w0, w1, w2, w3);
                clx_wave_sync();
                t0 += 12u;                                   // the whole turn is done
                CLX_STAT(16, 1);
                continue;
            }
        }
        int32_t y[4];
        bool lean_done = false;
        // ---- lean block: every live lane is in the middle of a Rice partition (or repeats a constant), nothing is near an
        //      edge, the 24-bit predictor holds: four codes, four predictor steps, ONE vote.  Lanes that decode subframes
        //      of the same shape meet their partition boundaries in the same block, so this is the common case by far.
        if (!no_lean) {
            LeanCur c = clx_lean_begin<OMAX>(S, ringrow, live, K);
            uint32_t mx = 0, pw = 0;
            int32_t xs[4];
            clx_lean_codes<4>(lean_mode, ringrow, K, c.k, c.k1, c.p, mx, pw, xs);
            int32_t hh[OMAX];
#pragma unroll
            for (int j = 0; j < OMAX; ++j) hh[j] = S.hist[j];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int32_t pred = clx_lpredict<OMAX, false>(S.c, hh, S.shift);      // taps beyond the order are zero
                const int32_t sm = (int32_t)((uint32_t)xs[ii] + (uint32_t)pred);
#pragma unroll
                for (int j = OMAX - 1; j > 0; --j) hh[j] = hh[j - 1];
                hh[0] = sm;
                y[ii] = sm;
            }
            int32_t hi = clx_max3(y[0], y[1], y[2]), lo = clx_min3(y[0], y[1], y[2]);
            hi = y[3] > hi ? y[3] : hi; lo = y[3] < lo ? y[3] : lo;
            const bool in_range = S.order == 0u || (hi < S.lim && lo >= -S.lim);
            const bool lean_ok = !live || ((K.rice ? (S.transitioned && !c.bad && c.pcnt >= 4u && mx <= 32u)
                                                   : K.verb ? true : S.phase == 2u)
                                           && (S.phase == 2u || clx_ring_covered(g, pw, c.p, r.limit)) && t0 + 4u <= n && S.lim >= 0 && in_range);
            if (__all(lean_ok)) {
                if (live) {
                    r.pos = c.p;
                    if (K.rice) { S.k = c.k; S.k1 = c.k1; S.pcnt = c.pcnt - 4u; S.next_cnt = c.next; S.parts_left = c.parts; }
#pragma unroll
                    for (int j = 0; j < OMAX; ++j) S.hist[j] = hh[j];
                }
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) y[ii] = clx_lfinish(y[ii], F);
                lean_done = true;
                CLX_STAT(32, 1);
            } else {
                CLX_STAT(40, live && K.rice && !S.transitioned); CLX_STAT(41, live && K.rice && c.bad); CLX_STAT(42, live && K.rice && c.pcnt < 4u);
                CLX_STAT(43, live && !(S.phase == 2u) && !clx_ring_covered(g, pw, c.p, r.limit)); CLX_STAT(44, live && mx > 32u); CLX_STAT(45, live && t0 + 4u > n);
                CLX_STAT(46, live && S.lim < 0); CLX_STAT(47, live && !in_range); CLX_STAT(48, live && S.phase == 3u);
            }
        }
        if (!lean_done) {
            CLX_STAT(33, 1); CLX_STAT(34, no_lean);
            const bool rice_on = live && S.phase == 1u;
            const bool verb_on = live && S.phase == 0u;
            bool can = true;
            if (live) {
                can = (t0 + 4u <= n) && S.phase != 3u;
                if (S.phase != 2u) can = can && r.pos <= g.fast_lim;
                if (S.phase == 1u) can = can && S.transitioned;
            }
            uint32_t pos2 = r.pos, pcnt2 = S.pcnt, k_2 = S.k, k1_2 = S.k1, parts2 = S.parts_left, next2 = S.next_cnt;
            int32_t xs[4];
            bool ok = can;
            const bool any_verb = __any(verb_on);
            {
                Win w = clx_win_load(ringrow, pos2);
                const uint32_t vsh = (32u - h.sf_bps) & 31u;
    #pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    if (__any(rice_on && pcnt2 == 0u)) {
                        if (rice_on && pcnt2 == 0u) {                    // partition parameter (subframe.rs:314-319 / 362-367)
                            const uint32_t pb = S.rice2 ? 5u : 4u;
                            k_2 = w.a >> (32u - pb);
                            if (k_2 == (S.rice2 ? 31u : 15u) || parts2 == 0u || next2 == 0u) ok = false;
                            k1_2 = k_2 + 1u; pos2 += pb; parts2 -= 1u; pcnt2 = next2; next2 = S.per;
                            clx_win_skip(w, pb);
                        }
                    }
                    // one Rice code (subframe.rs:337-341): z zeros, a one, k remainder bits
                    const uint32_t z = (uint32_t)__clz((int)w.a);        // 32 when the window is all zeros
                    uint32_t nb = z + k1_2;
                    if (rice_on && nb > 32u) ok = false;
                    const uint32_t u = (z << (k_2 & 31u)) | clx_bfe(w.a, 32u - nb, k_2);
                    int32_t x = (int32_t)(u >> 1) ^ -(int32_t)(u & 1u);  // rice_to_signed (subframe.rs:157-170)
                    if (any_verb) {                                      // verbatim rows ride along (subframe.rs:397-415)
                        if (verb_on) { x = (int32_t)w.a >> vsh; nb = h.sf_bps; }
                    }
                    if (!rice_on && !verb_on) { x = S.cval; nb = 32u; }  // constant / idle lanes: window contents are irrelevant
                    clx_win_skip(w, nb);
                    if (rice_on || verb_on) pos2 += nb;
                    pcnt2 -= 1u;
                    xs[ii] = x;
                }
            }
            const bool all_ok = __all(ok);
            CLX_STAT(35, all_ok); CLX_STAT(36, all_ok && __any(wide));
            if (all_ok) {
                if (live) { r.pos = pos2; S.pcnt = pcnt2; S.k = k_2; S.k1 = k1_2; S.parts_left = parts2; S.next_cnt = next2; }
                // predictor over the block: 24-bit evaluation, range-checked; exact i64 re-run when outside the proven range
                const bool wv = __any(wide);
                bool redo = wv;
                int32_t h0[OMAX];
    #pragma unroll
                for (int j = 0; j < OMAX; ++j) h0[j] = S.hist[j];
                if (!wv) {
    #pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const int32_t pred = clx_lpredict<OMAX, false>(S.c, S.hist, S.shift);
                        const int32_t s = (int32_t)((uint32_t)xs[ii] + (S.order != 0u ? (uint32_t)pred : 0u));
    #pragma unroll
                        for (int j = OMAX - 1; j > 0; --j) S.hist[j] = S.hist[j - 1];
                        S.hist[0] = s;
                        y[ii] = s;
                    }
                    int32_t mx = clx_max3(y[0], y[1], y[2]), mn = clx_min3(y[0], y[1], y[2]);
                    mx = y[3] > mx ? y[3] : mx; mn = y[3] < mn ? y[3] : mn;
                    const bool in_range = !live || S.order == 0u || (mx < S.lim && mn >= -S.lim);
                    if (!__all(in_range)) redo = true;
                }
                if (redo) {
    #pragma unroll
                    for (int j = 0; j < OMAX; ++j) S.hist[j] = h0[j];
    #pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const int32_t pred = clx_lpredict<OMAX, true>(S.c, S.hist, S.shift);
                        const int32_t s = (int32_t)((uint32_t)xs[ii] + (S.order != 0u ? (uint32_t)pred : 0u));
    #pragma unroll
                        for (int j = OMAX - 1; j > 0; --j) S.hist[j] = S.hist[j - 1];
                        S.hist[0] = s;
                        y[ii] = s;
                    }
                    // nobody checked these four: recount (conservatively -- one sample outside restarts the count)
                    int32_t mx = clx_max3(y[0], y[1], y[2]), mn = clx_min3(y[0], y[1], y[2]);
                    mx = y[3] > mx ? y[3] : mx; mn = y[3] < mn ? y[3] : mn;
                    const uint32_t g4 = good + 4u;
                    good = (mx < S.lim && mn >= -S.lim) ? (g4 < 64u ? g4 : 64u) : 0u;
                    wide = live && S.order != 0u && good < S.order;
                }
    #pragma unroll
                for (int ii = 0; ii < 4; ++ii) y[ii] = clx_lfinish(y[ii], F);
            } else {
                int32_t* const ys = reinterpret_cast<int32_t*>(&stage[(t0 >> 2) & 3u]);
    #pragma unroll 1
                for (uint32_t ii = 0; ii < 4u; ++ii) {
                    const uint32_t i = t0 + ii;
                    const int32_t x = clx_lcareful_raw<OMAX>(S, h, bs, i, n);
                    const int32_t pred = clx_lpredict<OMAX, true>(S.c, S.hist, S.shift);
                    const uint32_t use = (S.order != 0u && i >= S.order && i >= S.trans_at) ? 0xffffffffu : 0u;
                    const int32_t s = (int32_t)((uint32_t)x + ((uint32_t)pred & use));
    #pragma unroll
                    for (int j = OMAX - 1; j > 0; --j) S.hist[j] = S.hist[j - 1];
                    S.hist[0] = s;
                    // the careful steps predict in i64 and check nothing: recount
                    good = (s < S.lim && s >= -S.lim) ? (good < 64u ? good + 1u : 64u) : 0u;
                    ys[ii] = clx_lfinish(s, F);
                }
                wide = n != 0u && !r.err && S.order != 0u && good < S.order;
                const int4 yv = stage[(t0 >> 2) & 3u];
                y[0] = yv.x; y[1] = yv.y; y[2] = yv.z; y[3] = yv.w;
            }
            // while a lane's history holds values the 24-bit predictor cannot take, only the general block (which then
            // accumulates in i64) may run
            no_lean = __any(wide);
        }
        // ---- output: stage 16 bytes per block, write the row one full 64-byte segment at a time (scattered 16-byte
        //      stores issued microseconds apart reach HBM as partial-line writes: 2.8x write traffic when measured)
        stage[(t0 >> 2) & 3u] = make_int4(y[0], y[1], y[2], y[3]);
        if ((t0 & 12u) == 12u || t0 + 4u >= nmax) {
            const uint32_t base = t0 & ~15u;
#pragma unroll
            for (uint32_t q = 0; q < 4u; ++q) {
                const uint32_t idx = base + 4u * q;
                if (idx <= t0) {
                    const int4 v = stage[q];
                    if (row_aligned && idx + 4u <= n) *reinterpret_cast<int4*>(row + idx) = v;
                    else {
                        if (idx < n) row[idx] = v.x;
                        if (idx + 1u < n) row[idx + 1u] = v.y;
                        if (idx + 2u < n) row[idx + 2u] = v.z;
                        if (idx + 3u < n) row[idx + 3u] = v.w;
                    }
                }
            }
        }
    }
    // ---- a subframe whose warm-up fills the whole block switches after its last sample; trailing parameters of
    //      empty partitions are consumed (they are part of the stream: they move the next subframe / the CRC)
    if (n != 0u && !r.err && S.trans_at != 0xffffffffu && S.trans_at == n && !S.transitioned) clx_ltransition<OMAX>(S, h, bs);
    if (n != 0u && !r.err && S.transitioned) {
        while (!r.err && S.parts_left != 0u) { (void)clx_lread_rice_param(r, S.rice2); S.parts_left -= 1u; }
    }
}

template <int OMAX>
__device__ __forceinline__ void clx_lanes_run(LaneReader r, Ring& g, uint32_t* ringrow, int4* stage, const SfHead h, uint32_t bs, uint32_t decor, bool pair_ok,
                                              int32_t* __restrict__ row, bool row_aligned, uint32_t nmax, uint32_t omax, int lane,
                                              uint32_t* end_pos, uint32_t* err_out, int32_t* __restrict__ out, int32_t* __restrict__ dump) {
    LaneState<OMAX> S;
    S.r = r;
#pragma unroll
    for (int j = 0; j < OMAX; ++j) { S.c[j] = 0; S.hist[j] = 0; }
    const uint32_t n = r.err ? 0u : bs;              // a lane that failed in its header produces nothing
    S.phase = 3u; S.cval = 0; S.trans_at = 0xffffffffu; S.transitioned = false;
    S.order = 0; S.shift = 0; S.lim = 0x7fffffff;
    S.k = 0; S.k1 = 1; S.pcnt = 0; S.next_cnt = 0; S.per = 0; S.parts_left = 0; S.rice2 = 0;
    if (n) {
        if (h.kind == 0u) { S.cval = clx_lread_signed(S.r, h.sf_bps); S.phase = 2u; }              // decode_constant (subframe.rs:382-394)
        else if (h.kind == 1u) S.phase = 0u;
        else {
            if (bs < h.order) S.r.err = CLX_LERR(CLX_FORMAT_ERROR, h.kind == 2u ? CLX_MSG_FIXED_ORDER_GT_BLOCK : CLX_MSG_LPC_ORDER_GT_BLOCK);
            else { S.phase = 0u; S.trans_at = h.order; }
        }
    }
    clx_lanes_body<OMAX>(S, g, ringrow, stage, h, bs, n, decor, pair_ok, row, row_aligned, nmax, omax, lane, out, dump);
    *end_pos = S.r.pos;
    *err_out = S.r.err;
}

// Narrow output: this lane's staging row (channel `ch` of frame `fr`, n samples; n = 0: nothing) into the run's interleaved output --
// sample t of channel c at byte (out_off + t * channels + c) * bytes of `out` (lib.rs:473-520's order), the low 2 or 3 bytes of every
// sample, little-endian (as clx_k_interleave gives them).  A row is written by several lanes of the wave (the turns' transposing
// stores): what they wrote is made visible to the lane that narrows it first.  Slow and simple: the odd group the tiers left.
__device__ __forceinline__ void clx_narrow_row(const clx_run& R, const clx_dev_frame& fr, uint32_t ch, uint32_t n, const int32_t* row, uint32_t narrow) {
    clx_group_fence();
    const uint32_t sb = (narrow & CLX_RUN_PCM24) ? 3u : 2u, step = (uint32_t)fr.n_channels * sb;
    uint8_t* d = reinterpret_cast<uint8_t*>(R.out) + (fr.out_off + (uint64_t)ch) * sb;
#pragma unroll 1
    for (uint32_t t = 0; t < n; ++t, d += step) {
        const uint32_t v = (uint32_t)row[t];
        if (sb == 2u) *reinterpret_cast<uint16_t*>(d) = (uint16_t)v;
        else { d[0] = (uint8_t)v; d[1] = (uint8_t)(v >> 8); d[2] = (uint8_t)(v >> 16); }
    }
}

// Two kernels: waves whose highest predictor order is <= 12, and the rest (the 32-tap predictor state would otherwise
// cost every wave its occupancy: 256 VGPRs = one wave per SIMD).  Both are launched over all slots; a wave leaves at once
// when its subframes belong to the other kernel.
// One group of 64 slots (`grp`) of run R.
template <bool HI>
__device__ __forceinline__ void clx_lanes_group(LanesLds& L, const clx_run& R, const clx_dev_frame* __restrict__ frames,
                 uint32_t n_slots, int32_t* __restrict__ dump_all, const uint32_t grp) {
    const uint8_t* const arena = R.arena;
    const uint64_t arena_alloc_len = R.alloc_len;
    const uint32_t* const sf_start = R.sf_start;
    // Narrow output (CLX_OUT_PCM16 / _PCM24): the group is decoded into staging rows of this workgroup's own -- 64 rows of `stride`
    // samples, re-used for every group the workgroup takes -- and every lane then narrows ITS row into the run's interleaved output
    // (clx_narrow_row).  (Until round 6 every run in flight had a planar scratch as large as the planar output for this, and a
    // kernel of its own behind: ADVICE r05.)
    const uint32_t narrow = R.flags & (CLX_RUN_PCM16 | CLX_RUN_PCM24);
    const uint32_t stride = CLX_RUN_STAGE_STRIDE(R.flags);
    int32_t* const out = narrow ? R.planar + (size_t)blockIdx.x * 64u * stride : R.out;
    uint32_t* const errkey = R.errkey;
    uint64_t* const end_bits = R.end_bits;
    CLX_TL_BEGIN();
    const int lane = (int)threadIdx.x;
    const uint32_t slot = grp * 64u + (uint32_t)lane;
    // (the run's slot map: the plan's, or what clx_k_compose dealt -- clx_lean.hip; the scan's results are indexed by the plan's slot)
    uint32_t f = 0xffffffffu;
    if (slot < n_slots) f = R.slot_frame[slot];
    clx_dev_frame fr;
    fr.byte_off = 0; fr.out_off = 0; fr.limit_bits = 0; fr.first_slot = 0; fr.header_bytes = 0; fr.block_size = 0;
    fr.n_channels = 0; fr.channel_assignment = 0; fr.bps = 1; fr.flags = 0;
    if (f != 0xffffffffu) fr = frames[f];
    const uint32_t ch = (f != 0xffffffffu) ? slot - R.first_slot[f] : 0u;
    uint32_t bs = fr.block_size;

    LaneReader r;
    r.arena = arena;
    r.origin = (uint32_t)(fr.byte_off & ~15ull);
    const uint32_t o = 8u * (uint32_t)(fr.byte_off & 15ull);
    r.limit = o + fr.limit_bits;
    r.pos = o + 8u * (uint32_t)fr.header_bytes;
    r.err = 0u;
    bool active = (f != 0xffffffffu);
    if (active && ch != 0u) {
        const uint32_t sp = sf_start[fr.first_slot + ch];
        if (sp == 0xffffffffu) active = false;           // an earlier channel failed (the scan reported it)
        else r.pos = sp;
    }
    if (active && r.pos > r.limit) r.err = CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
    if (!active) { bs = 0; r.err = 1u; }                 // idle lane: produces nothing, reports nothing
    Ring g;
    g.src = reinterpret_cast<const uint32_t*>(arena + r.origin);
    g.avail_dw = (uint32_t)((arena_alloc_len > r.origin ? arena_alloc_len - r.origin : 0ull) >> 2);
    g.fill = 0; g.npend = 0; g.pend0 = make_uint4(0u, 0u, 0u, 0u); g.pend1 = g.pend0; g.pend2 = g.pend0; g.fast_lim = 0;

    SfHead h = { 1u, 0u, 0u, 1u };
    if (active && !r.err) h = clx_lparse_sf_header(r, clx_channel_bps(fr, ch));

    // stereo pairing: both lanes of a decorrelated frame sit at (even, odd) slots of one wave
    const uint32_t decor = active ? fr.channel_assignment : 0u;
    const uint32_t pbs = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(active ? bs : 0u), 0xB1, 0xF, 0xF, false);
    const uint32_t pd = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)decor, 0xB1, 0xF, 0xF, false);
    const bool pair_ok = active && decor != CLX_CH_INDEPENDENT && pbs == bs && pd == decor;

    uint32_t nmax = active ? bs : 0u;
    uint32_t omax = (active && !r.err && h.kind >= 2u) ? h.order : 0u;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        uint32_t a = __shfl_xor(nmax, s, 64); nmax = a > nmax ? a : nmax;
        uint32_t b = __shfl_xor(omax, s, 64); omax = b > omax ? b : omax;
    }
    int32_t* const row = narrow ? out + (size_t)lane * stride : out + (active ? fr.out_off + (uint64_t)ch * fr.block_size : 0ull);
    const bool row_aligned = (((uintptr_t)row) & 15u) == 0u;
    uint32_t end_pos = r.pos, err = r.err;
    int32_t* const dump = dump_all + (size_t)slot * 16u;                   // 64 bytes per lane for stores that fall outside a row
    if ((omax > 12u) != HI) return;                          // the other kernel's wave (it reports this wave's lanes)
    if (nmax != 0u) {
        if (HI)               clx_lanes_run<32>(r, g, L.ring[lane], L.stage[lane], h, bs, decor, pair_ok, row, row_aligned, nmax, omax, lane, &end_pos, &err, out, dump);
        else if (omax <= 4u)  clx_lanes_run<4>(r, g, L.ring[lane], L.stage[lane], h, bs, decor, pair_ok, row, row_aligned, nmax, omax, lane, &end_pos, &err, out, dump);
        else if (omax <= 8u)  clx_lanes_run<8>(r, g, L.ring[lane], L.stage[lane], h, bs, decor, pair_ok, row, row_aligned, nmax, omax, lane, &end_pos, &err, out, dump);
        else                  clx_lanes_run<12>(r, g, L.ring[lane], L.stage[lane], h, bs, decor, pair_ok, row, row_aligned, nmax, omax, lane, &end_pos, &err, out, dump);
    }
    if (narrow) clx_narrow_row(R, fr, ch, active ? bs : 0u, row, narrow);
    if (active) {
        if (err) clx_report_error(errkey, f, ch, err);
        else if (ch + 1u == fr.n_channels) end_bits[f] = (uint64_t)(end_pos - o);
    }
    CLX_TL_END(3, grp);
}
// Behind the lean tiers (R.taken != null) the general kernels do not look at every group: clx_k_left has listed the groups the tiers
// left -- usually none -- and a grid that may be much smaller than the number of groups loops over that list.  (Round 4's form,
// one workgroup of 14 KiB LDS per group that only read its group's mark, waited for room behind the other stream's decode kernel:
// 3 756 workgroups per merged launch, median 0.46 ms, and the stream's next scan queued behind them.)
// The list: R.taken[n_groups] = how many, R.taken[n_groups + 1 ..] = which, in no particular order.
template <bool HI>
__device__ __forceinline__ void clx_lanes_fused(const clx_runs& runs, const clx_dev_frame* __restrict__ frames,
                 uint32_t n_slots, int32_t* __restrict__ dump_all) {
    __shared__ LanesLds L;
    const clx_run& R = runs.r[blockIdx.y];
    if (R.taken == nullptr) { clx_lanes_group<HI>(L, R, frames, n_slots, dump_all, blockIdx.x); return; }
    const uint32_t* const left = R.taken + (n_slots + 63u) / 64u;
    const uint32_t n_left = left[0];
#pragma unroll 1
    for (uint32_t i = blockIdx.x; i < n_left; i += gridDim.x) {
        clx_lanes_group<HI>(L, R, frames, n_slots, dump_all, left[1u + i]);
        clx_wave_sync();                                     // (the next group's ring and stage start from scratch: one wave, program order)
    }
}
// (most_left: the longest list any run of the batch has had -- the host reads it back, without waiting for it, to size the
//  general kernels' grid of LATER launches: what is left for reasons only the stream knows -- 16-bit audio of more than 12 taps in a
//  batch without the split tier, waves that give up -- is left again in the next run of the same batch)
extern "C" __global__ __launch_bounds__(256)
void clx_k_left(const clx_runs runs, uint32_t n_groups, uint32_t* __restrict__ most_left, uint32_t* __restrict__ most_left_host) {
    const clx_run& R = runs.r[blockIdx.y];
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n_groups || R.taken == nullptr) return;
    if (R.taken[g] != R.gen) {
        uint32_t* const left = R.taken + n_groups;
        const uint32_t i = atomicAdd(&left[0], 1u);
        left[1u + i] = g;
        // (the host's copy, in pinned memory the device writes to: a plain store by whoever raised the maximum -- a hint that sizes
        //  later launches' grids.  Round 5 copied it back in the stream behind the general kernels: 0.1 ms per launch in front of
        //  clx_k_finalize, profiles/r06_pipelined_trace.txt)
        if (most_left != nullptr && atomicMax(most_left, i + 1u) < i + 1u && most_left_host != nullptr) *(volatile uint32_t*)most_left_host = i + 1u;
    }
}
extern "C" __global__ __launch_bounds__(64)
void clx_k_lanes(const clx_runs runs, const clx_dev_frame* __restrict__ frames, uint32_t n_slots,
                 int32_t* __restrict__ dump_all) {
    clx_lanes_fused<false>(runs, frames, n_slots, dump_all);
}
extern "C" __global__ __launch_bounds__(64)
void clx_k_lanes_hi(const clx_runs runs, const clx_dev_frame* __restrict__ frames, uint32_t n_slots,
                    int32_t* __restrict__ dump_all) {
    clx_lanes_fused<true>(runs, frames, n_slots, dump_all);
}

// ------------------------------------------------------------------------------------------------
// D2 (uses K2Predictor / K2Finisher / K2Mover of clx_kernels.hip, which every translation unit includes first):
// the same work split over two waves per 64 subframes, as K2 is: a lone wave issues one
// instruction every ~6 ticks and a lane-serial kernel lasts exactly as long as one wave's instruction stream, so the
// stream is cut in two.
//   wave R  (Rice):       header, warm-up, LPC parameters, Rice / Rice2 decode -> the raw value x of every sample
//                         (residual, warm-up / verbatim sample, or the constant), 16 samples of each row per turn, into
//                         an LDS tile; the predictor's parameters into an LDS descriptor as soon as they are parsed
//   wave PF (predictor +  x from the tile -> recurrence (K2Predictor) -> wasted shift + decorrelation (K2Finisher) ->
//            finisher):   tile -> HBM in the 64 B x 16 rows store shape (K2Mover)
// One workgroup barrier per turn; PF works on tile T while R fills tile T+1 (two tiles per pair).  Workgroup = 4 waves
// = two (R, PF) pairs, so that a CU's four SIMDs are filled by construction.
// ------------------------------------------------------------------------------------------------
struct Lanes2Lds {
    uint32_t ring[64][CLX_ROW];      // R's bitstream ring
    int4 tile[2][4][64];             // x / y tiles, layout as in K2 (int4 [row][pos], pos = piece ^ ((row >> 2) & 3))
    clx_sf_desc desc[64];            // what PF needs to know about each row's predictor (written by R)
};

// publish the predictor of a row (after the transition) / its shape (at the start)
__device__ __forceinline__ void clx_l2_publish(clx_sf_desc* d, const LaneState<32>& S, const SfHead& h, uint32_t n, bool with_coefs) {
    d->n = (uint16_t)n;
    d->order = (uint8_t)((n != 0u && h.kind >= 2u) ? h.order : 0u);
    d->wasted = (uint8_t)h.wasted;
    d->flags = 0; d->decor = 0; d->out_base = 0;
    if (!with_coefs) {
        // until the transition nothing is predicted: no taps, and a range limit that lets warm-up samples through
        d->shift = 0; d->lim_log2 = 23;
#pragma unroll
        for (int j = 0; j < 32; ++j) d->coef[j] = 0;
    } else {
        d->shift = (uint8_t)S.shift;
        d->lim_log2 = (uint8_t)(S.lim < 0 ? 0xffu : (uint32_t)(31 - __clz(S.lim)));
#pragma unroll
        for (int j = 0; j < 32; ++j) d->coef[j] = (int16_t)S.c[j];
    }
}

// the Rice wave.  Returns through S.r (position, error) like clx_lanes_run.
__device__ __forceinline__ void clx_rice_wave(LaneState<32>& S, Ring& g, uint32_t* ringrow, int4 (*tile)[4][64], clx_sf_desc* desc, const SfHead h,
                                              uint32_t bs, uint32_t n, uint32_t nmax, uint32_t omax, int lane) {
    LaneReader& r = S.r;
    const uint32_t sw = ((uint32_t)lane >> 2) & 3u;
    bool published = false;
    uint32_t i0 = (omax + 4u + 15u) & ~15u;              // careful prologue: warm-up samples, the transition, the first residuals
    if (i0 > nmax) i0 = (nmax + 15u) & ~15u;
    const uint32_t nturn = (nmax + 15u) >> 4;
    bool ring_ready = false;
    for (uint32_t T = 0; T < nturn; ++T) {
        int4* const out4 = &tile[T & 1u][0][0];
        // ---- lean turn: sixteen codes of one partition in every live lane (or sixteen repeats of a constant), four register
        //      windows, ONE vote.  Anything else -- a partition edge, verbatim rows, a long code, the ring running low, the
        //      prologue -- takes the turn through the four blocks below.
        if (ring_ready) {
            const uint32_t tb = 16u * T;
            clx_ring_pump(g, ringrow, r.pos, r.limit);
            const bool live = (n != 0u) && !r.err && tb < n;
            const LeanKind K = clx_lean_kind<32>(S, h);
            const int lean_mode = clx_lean_mode(live, K, S.phase);
            LeanCur c = clx_lean_begin<32>(S, ringrow, live, K);
            uint32_t mx = 0, pw = 0;
            int32_t X[16];
            clx_lean_codes<16>(lean_mode, ringrow, K, c.k, c.k1, c.p, mx, pw, X);
            const bool ok16 = !live || ((K.rice ? (S.transitioned && !c.bad && c.pcnt >= 16u && mx <= 32u)
                                                : K.verb ? true : S.phase == 2u)
                                        && (S.phase == 2u || clx_ring_covered(g, pw, c.p, r.limit)) && tb + 16u <= n);
            if (__all(ok16)) {
                if (live) {
                    r.pos = c.p;
                    if (K.rice) { S.k = c.k; S.k1 = c.k1; S.pcnt = c.pcnt - 16u; S.next_cnt = c.next; S.parts_left = c.parts; }
                }
#pragma unroll
                for (uint32_t q = 0; q < 4u; ++q) out4[(uint32_t)lane * 4u + (q ^ sw)] = make_int4(X[4 * q], X[4 * q + 1], X[4 * q + 2], X[4 * q + 3]);
                clx_wg_barrier();
                continue;
            }
        }
        for (uint32_t q = 0; q < 4u; ++q) {
            const uint32_t t0 = 16u * T + 4u * q;
            int32_t xs[4] = { 0, 0, 0, 0 };
            if (t0 < i0) {
#pragma unroll 1
                for (uint32_t ii = 0; ii < 4u; ++ii) {
                    const int32_t x = clx_lcareful_raw<32>(S, h, bs, t0 + ii, n);
                    xs[0] = ii == 0u ? x : xs[0]; xs[1] = ii == 1u ? x : xs[1]; xs[2] = ii == 2u ? x : xs[2]; xs[3] = ii == 3u ? x : xs[3];
                    if (S.transitioned && !published) { clx_l2_publish(desc, S, h, n, true); published = true; }
                }
            } else {
                if (!ring_ready) { clx_ring_reset(g, ringrow, r.pos >> 5, r.limit); ring_ready = true; }      // (wave-uniform: t0 is)
                // (a turn that started with the ring in place was pumped by the lean attempt above)
                const bool live = (n != 0u) && !r.err && t0 < n;
                bool lean_done = false;
                // lean block: see clx_lanes_body -- here without the predictor
                {
                    const LeanKind K = clx_lean_kind<32>(S, h);
                    const int lean_mode = clx_lean_mode(live, K, S.phase);
                    LeanCur c = clx_lean_begin<32>(S, ringrow, live, K);
                    uint32_t mx = 0, pw = 0;
                    clx_lean_codes<4>(lean_mode, ringrow, K, c.k, c.k1, c.p, mx, pw, xs);
                    const bool lean_ok = !live || ((K.rice ? (S.transitioned && !c.bad && c.pcnt >= 4u && mx <= 32u)
                                                           : K.verb ? true : S.phase == 2u)
                                                   && (S.phase == 2u || clx_ring_covered(g, pw, c.p, r.limit)) && t0 + 4u <= n);
                    if (__all(lean_ok)) {
                        if (live) {
                            r.pos = c.p;
                            if (K.rice) { S.k = c.k; S.k1 = c.k1; S.pcnt = c.pcnt - 4u; S.next_cnt = c.next; S.parts_left = c.parts; }
                        }
                        lean_done = true;
                    }
                }
                if (!lean_done) {
                    // general block: partition parameters between the codes, verbatim rows; careful steps when anything is unusual
                    const bool rice_on = live && S.phase == 1u;
                    const bool verb_on = live && S.phase == 0u;
                    bool can = true;
                    if (live) {
                        can = (t0 + 4u <= n) && S.phase != 3u;
                        if (S.phase != 2u) can = can && r.pos <= g.fast_lim;
                        if (S.phase == 1u) can = can && S.transitioned;
                    }
                    uint32_t pos2 = r.pos, pcnt2 = S.pcnt, k_2 = S.k, k1_2 = S.k1, parts2 = S.parts_left, next2 = S.next_cnt;
                    bool ok = can;
                    const bool any_verb = __any(verb_on);
                    {
                        Win w = clx_win_load(ringrow, pos2);
                        const uint32_t vsh = (32u - h.sf_bps) & 31u;
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) {
                            if (__any(rice_on && pcnt2 == 0u)) {
                                if (rice_on && pcnt2 == 0u) {                    // partition parameter (subframe.rs:314-319 / 362-367)
                                    const uint32_t pb = S.rice2 ? 5u : 4u;
                                    k_2 = w.a >> (32u - pb);
                                    if (k_2 == (S.rice2 ? 31u : 15u) || parts2 == 0u || next2 == 0u) ok = false;
                                    k1_2 = k_2 + 1u; pos2 += pb; parts2 -= 1u; pcnt2 = next2; next2 = S.per;
                                    clx_win_skip(w, pb);
                                }
                            }
                            const uint32_t z = (uint32_t)__clz((int)w.a);
                            uint32_t nb = z + k1_2;
                            if (rice_on && nb > 32u) ok = false;
                            const uint32_t u = (z << (k_2 & 31u)) | clx_bfe(w.a, 32u - nb, k_2);
                            int32_t x = (int32_t)(u >> 1) ^ -(int32_t)(u & 1u);
                            if (any_verb) {
                                if (verb_on) { x = (int32_t)w.a >> vsh; nb = h.sf_bps; }
                            }
                            if (!rice_on && !verb_on) { x = S.cval; nb = 32u; }
                            clx_win_skip(w, nb);
                            if (rice_on || verb_on) pos2 += nb;
                            pcnt2 -= 1u;
                            xs[ii] = x;
                        }
                    }
                    if (__all(ok)) {
                        if (live) { r.pos = pos2; S.pcnt = pcnt2; S.k = k_2; S.k1 = k1_2; S.parts_left = parts2; S.next_cnt = next2; }
                    } else {
#pragma unroll 1
                        for (uint32_t ii = 0; ii < 4u; ++ii) {
                            const int32_t x = clx_lcareful_raw<32>(S, h, bs, t0 + ii, n);
                            xs[0] = ii == 0u ? x : xs[0]; xs[1] = ii == 1u ? x : xs[1]; xs[2] = ii == 2u ? x : xs[2]; xs[3] = ii == 3u ? x : xs[3];
                            if (S.transitioned && !published) { clx_l2_publish(desc, S, h, n, true); published = true; }
                        }
                    }
                }
            }
            out4[(uint32_t)lane * 4u + (q ^ sw)] = make_int4(xs[0], xs[1], xs[2], xs[3]);
        }
        clx_wg_barrier();
    }
    // a subframe whose warm-up fills the whole block switches after its last sample; trailing parameters of empty
    // partitions are part of the stream (they move the next subframe / the CRC)
    if (n != 0u && !r.err && S.trans_at != 0xffffffffu && S.trans_at == n && !S.transitioned) clx_ltransition<32>(S, h, bs);
    if (n != 0u && !r.err && S.transitioned) {
        while (!r.err && S.parts_left != 0u) { (void)clx_lread_rice_param(r, S.rice2); S.parts_left -= 1u; }
    }
}

// the predictor + finisher wave
template <int OMAX, int MODE, bool ALIGNED>
__device__ __forceinline__ void clx_pf_wave(int4 (*tile)[4][64], int32_t* __restrict__ out, const K2Slot& S, const K2Finisher& F,
                                            int32_t* __restrict__ dump, uint32_t nturn, int lane) {
    K2Predictor<OMAX> P; P.init(S);
    K2Mover M; M.init(out, S, lane);
    const uint32_t sw = ((uint32_t)lane >> 2) & 3u;
    for (uint32_t T = 0; T < nturn; ++T) {
        clx_wg_barrier();                                  // tile T is complete (and so is the descriptor of every row that needs it by now)
        if (T <= 2u) {                                     // coefficients arrive with the transition, at sample `order` <= 32
            const clx_sf_desc* d = S.d;
#pragma unroll
            for (int j = 0; j < OMAX; ++j) P.c[j] = (P.n != 0u && (uint32_t)j < P.order) ? (int32_t)d->coef[j] : 0;
            P.shift = d->shift;
            const uint32_t ll = d->lim_log2;
            P.lim = (ll <= 23u) ? (int32_t)(1u << ll) : -1;
            bool in = P.lim >= 0;                          // the 24-bit evaluation needs every tap's history inside the proven range
#pragma unroll
            for (int j = 0; j < OMAX; ++j) in = in && P.hist[j] < P.lim && P.hist[j] >= -P.lim;
            P.h_ok = in || P.trivial;
        }
        int4* const t4 = &tile[T & 1u][0][0];
        int32_t x[CLX_BLK], y[CLX_BLK];
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
            const int4 v = t4[(uint32_t)lane * 4u + (q ^ sw)];
            x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
        P.block(x, y, T * CLX_BLK, K2NoHook());
        F.template block<MODE>(y, K2NoHook());
        if (ALIGNED) {
#pragma unroll
            for (uint32_t q = 0; q < 4u; ++q) t4[(uint32_t)lane * 4u + (q ^ sw)] = make_int4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
            clx_wave_sync();
            const uint32_t t = T * CLX_BLK + 4u * M.pc;
            const int4 w0 = t4[lane], w1 = t4[64 + lane], w2 = t4[128 + lane], w3 = t4[192 + lane];
            *reinterpret_cast<int4*>(t < M.rn[0] ? const_cast<int32_t*>(M.rp[0]) + t : dump + 0) = w0;
            *reinterpret_cast<int4*>(t < M.rn[1] ? const_cast<int32_t*>(M.rp[1]) + t : dump + 4) = w1;
            *reinterpret_cast<int4*>(t < M.rn[2] ? const_cast<int32_t*>(M.rp[2]) + t : dump + 8) = w2;
            *reinterpret_cast<int4*>(t < M.rn[3] ? const_cast<int32_t*>(M.rp[3]) + t : dump + 12) = w3;
        } else {
#pragma unroll
            for (uint32_t i = 0; i < (uint32_t)CLX_BLK; ++i) if (T * CLX_BLK + i < S.n) S.row[T * CLX_BLK + i] = y[i];
        }
    }
}

template <int OMAX, bool ALIGNED>
__device__ __forceinline__ void clx_pf_wave_mode(int4 (*tile)[4][64], int32_t* __restrict__ out, const K2Slot& S, int32_t* __restrict__ dump,
                                                 uint32_t nturn, int lane) {
    K2Finisher F; F.init(S, lane);
    const int mode = F.mode();
    if (mode == 0)      clx_pf_wave<OMAX, 0, ALIGNED>(tile, out, S, F, dump, nturn, lane);
    else if (mode == 1) clx_pf_wave<OMAX, 1, ALIGNED>(tile, out, S, F, dump, nturn, lane);
    else                clx_pf_wave<OMAX, 2, ALIGNED>(tile, out, S, F, dump, nturn, lane);
}

extern "C" __global__ __launch_bounds__(256)
void clx_k_lanes2(const uint8_t* __restrict__ arena, uint64_t arena_alloc_len,
                  const clx_dev_frame* __restrict__ frames,
                  const uint32_t* __restrict__ slot_frame, uint32_t n_slots,
                  const uint32_t* __restrict__ sf_start, int32_t* __restrict__ out,
                  uint32_t* __restrict__ errkey, uint64_t* __restrict__ end_bits, int32_t* __restrict__ dump_all) {
    __shared__ Lanes2Lds L2[2];
    const int lane = (int)threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t pair = wave & 1u;
    const bool is_pf = wave >= 2u;                 // wave-uniform
    Lanes2Lds& L = L2[pair];
    const uint32_t group = blockIdx.x * 2u + pair;
    const uint32_t slot = group * 64u + (uint32_t)lane;
    uint32_t f = 0xffffffffu;
    if (slot < n_slots) f = slot_frame[slot];
    clx_dev_frame fr;
    fr.byte_off = 0; fr.out_off = 0; fr.limit_bits = 0; fr.first_slot = 0; fr.header_bytes = 0; fr.block_size = 0;
    fr.n_channels = 0; fr.channel_assignment = 0; fr.bps = 1; fr.flags = 0;
    if (f != 0xffffffffu) fr = frames[f];
    const uint32_t ch = (f != 0xffffffffu) ? slot - fr.first_slot : 0u;
    // both waves of a pair see the same 64 slots: nmax (the number of turns) is the same in both.  A row that turns out
    // to be unreadable (the scan failed on an earlier channel) still takes part with garbage -- its frame is reported failed.
    uint32_t nmax = (f != 0xffffffffu) ? fr.block_size : 0u;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { const uint32_t a = __shfl_xor(nmax, s, 64); nmax = a > nmax ? a : nmax; }
    if (nmax == 0u) return;
    const uint32_t nturn = (nmax + 15u) >> 4;

    if (!is_pf) {
        uint32_t bs = fr.block_size;
        LaneReader r;
        r.arena = arena;
        r.origin = (uint32_t)(fr.byte_off & ~15ull);
        const uint32_t o = 8u * (uint32_t)(fr.byte_off & 15ull);
        r.limit = o + fr.limit_bits;
        r.pos = o + 8u * (uint32_t)fr.header_bytes;
        r.err = 0u;
        bool active = (f != 0xffffffffu);
        if (active && ch != 0u) {
            const uint32_t sp = sf_start[slot];
            if (sp == 0xffffffffu) active = false;           // an earlier channel failed (the scan reported it)
            else r.pos = sp;
        }
        if (active && r.pos > r.limit) r.err = CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
        if (!active) { bs = 0; r.err = 1u; }                 // idle lane: produces nothing, reports nothing
        Ring g;
        g.src = reinterpret_cast<const uint32_t*>(arena + r.origin);
        g.avail_dw = (uint32_t)((arena_alloc_len > r.origin ? arena_alloc_len - r.origin : 0ull) >> 2);
        g.fill = 0; g.npend = 0; g.pend0 = make_uint4(0u, 0u, 0u, 0u); g.pend1 = g.pend0; g.pend2 = g.pend0; g.fast_lim = 0;
        SfHead h = { 1u, 0u, 0u, 1u };
        if (active && !r.err) h = clx_lparse_sf_header(r, clx_channel_bps(fr, ch));
        uint32_t omax = (active && !r.err && h.kind >= 2u) ? h.order : 0u;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) { const uint32_t b = __shfl_xor(omax, s, 64); omax = b > omax ? b : omax; }
        LaneState<32> S;
        S.r = r;
#pragma unroll
        for (int j = 0; j < 32; ++j) { S.c[j] = 0; S.hist[j] = 0; }
        const uint32_t n = r.err ? 0u : bs;              // a lane that failed in its header produces nothing
        S.phase = 3u; S.cval = 0; S.trans_at = 0xffffffffu; S.transitioned = false;
        S.order = 0; S.shift = 0; S.lim = 0x7fffffff;
        S.k = 0; S.k1 = 1; S.pcnt = 0; S.next_cnt = 0; S.per = 0; S.parts_left = 0; S.rice2 = 0;
        if (n) {
            if (h.kind == 0u) { S.cval = clx_lread_signed(S.r, h.sf_bps); S.phase = 2u; }              // decode_constant (subframe.rs:382-394)
            else if (h.kind == 1u) S.phase = 0u;
            else {
                if (bs < h.order) S.r.err = CLX_LERR(CLX_FORMAT_ERROR, h.kind == 2u ? CLX_MSG_FIXED_ORDER_GT_BLOCK : CLX_MSG_LPC_ORDER_GT_BLOCK);
                else { S.phase = 0u; S.trans_at = h.order; }
            }
        }
        clx_l2_publish(&L.desc[lane], S, h, S.r.err ? 0u : n, false);
        clx_wg_barrier();                                    // PF may read the rows' shapes
        clx_rice_wave(S, g, L.ring[lane], L.tile, &L.desc[lane], h, bs, S.r.err ? 0u : n, nmax, omax, lane);
        if (active) {
            if (S.r.err) clx_report_error(errkey, f, ch, S.r.err);
            else if (ch + 1u == fr.n_channels) end_bits[f] = (uint64_t)(S.r.pos - o);
        }
    } else {
        clx_wg_barrier();                                    // the rows' shapes are in L.desc
        K2Slot S;
        S.d = &L.desc[lane];
        const bool valid = f != 0xffffffffu;
        S.n = valid ? fr.block_size : 0u;
        S.order = valid ? S.d->order : 0u; S.shift = 0; S.wasted = valid ? S.d->wasted : 0u;
        S.decor = valid ? fr.channel_assignment : 0u; S.lim_log2 = 0;
        S.row = out + (valid ? fr.out_off + (uint64_t)ch * fr.block_size : 0ull);
        if (S.n == 0u) { S.order = 0; S.wasted = 0; S.decor = 0; }
        const uint32_t pn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)S.n, 0xB1, 0xF, 0xF, false);
        const uint32_t pd = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)S.decor, 0xB1, 0xF, 0xF, false);
        S.pair_ok = (S.decor != CLX_CH_INDEPENDENT) && pn == S.n && pd == S.decor && S.n != 0u;
        uint32_t omax = S.order;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) { const uint32_t b = __shfl_xor(omax, s, 64); omax = b > omax ? b : omax; }
        int32_t* const dump = dump_all + (size_t)(group * 64u + (uint32_t)lane) * CLX_BLK;
        const bool al = (S.n == 0u) || ((((uintptr_t)S.row) & 15u) == 0u && (S.n & 3u) == 0u);
        if (__all(al)) {
            if (omax <= 4u)       clx_pf_wave_mode<4, true>(L.tile, out, S, dump, nturn, lane);
            else if (omax <= 8u)  clx_pf_wave_mode<8, true>(L.tile, out, S, dump, nturn, lane);
            else if (omax <= 12u) clx_pf_wave_mode<12, true>(L.tile, out, S, dump, nturn, lane);
            else                  clx_pf_wave_mode<32, true>(L.tile, out, S, dump, nturn, lane);
        } else {
            if (omax <= 4u)       clx_pf_wave_mode<4, false>(L.tile, out, S, dump, nturn, lane);
            else if (omax <= 8u)  clx_pf_wave_mode<8, false>(L.tile, out, S, dump, nturn, lane);
            else if (omax <= 12u) clx_pf_wave_mode<12, false>(L.tile, out, S, dump, nturn, lane);
            else                  clx_pf_wave_mode<32, false>(L.tile, out, S, dump, nturn, lane);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// F: error keys -> clx_frame_result
// ------------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256)
void clx_k_finalize(const clx_runs runs, const clx_dev_frame* __restrict__ frames, uint32_t n_frames, uint32_t n_slots) {
    const clx_run& R = runs.r[blockIdx.y];
    const uint32_t f = blockIdx.x * 256u + threadIdx.x;
    if (f >= n_frames) return;
    if (f == 0u && R.taken != nullptr) R.taken[(n_slots + 63u) / 64u] = 0u;     // (the list of groups the tiers left: empty for the next run)
    const uint32_t key = R.errkey[f];
    const clx_dev_frame fr = frames[f];
    clx_frame_result r;
    if (key == 0xffffffffu) {
        r.status = CLX_OK; r.msg = CLX_MSG_NONE; r.end_bit = R.end_bits[f];
        // the footer is read whether or not it is compared (frame.rs:754; under cfg(fuzzing) only the comparison goes away)
        if (!(fr.flags & 1u) && ((r.end_bit + 7ull) & ~7ull) + 16ull > (uint64_t)fr.limit_bits) {
            r.status = CLX_IO_ERROR; r.msg = CLX_MSG_UNEXPECTED_EOF;
        }
    }
    else { r.status = (int32_t)((key >> 16) & 0xffu); r.msg = key & 0xffffu; r.end_bit = 0; }
    // the frame's CRC-16 (frame.rs:752-763) from what the lean kernels' lanes gathered of it (clx_crct.h): every subframe's lane has
    // left its share's remainder for THIS run, the shares tile the frame, and the frame ends where its descriptor says -- otherwise
    // (a group the general kernels decoded, a descriptor that only bounds the frame) clx_k_crc16_runs checks the frame
    if (R.flags & CLX_RUN_CRC) {
        uint32_t todo = 0u;
        if (r.status == CLX_OK && !(fr.flags & 1u)) {
            todo = 1u;
            if (((r.end_bit + 7ull) & ~7ull) + 16ull == (uint64_t)fr.limit_bits) {
                bool all = true;
                uint32_t sum = 0u, par = 0u, at = 0u;
                const uint32_t nch = fr.n_channels;
                const uint32_t dend = R.crc_part[fr.first_slot + nch - 1u].db;
                for (uint32_t c = 0; c < nch; ++c) {
                    const clx_crc_part p = R.crc_part[fr.first_slot + c];
                    all = all && p.gen == R.gen && p.da == at && p.db >= p.da && p.db <= dend;
                    if (!all) break;
                    at = p.db;
                    sum ^= clx_crct_shift(p.rx & 0x7fffu, dend - p.db);
                    par ^= p.rx >> 31;
                }
                if (all) {
                    todo = 0u;
                    if (sum != 0u || par != 0u) { r.status = CLX_FORMAT_ERROR; r.msg = CLX_MSG_FRAME_CRC_MISMATCH; }
                }
            }
        }
        R.crc_todo[f] = todo;
    }
    R.results[f] = r;
    // leave the run's scratch as the next run expects to find it (nothing else reads it after this kernel): no error yet, no later
    // subframe located yet -- the host clears it once, when it is allocated
    R.errkey[f] = 0xffffffffu;
    for (uint32_t c = 1; c < fr.n_channels; ++c) R.sf_start[fr.first_slot + c] = 0xffffffffu;
}
