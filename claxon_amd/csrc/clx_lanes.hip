// clx_lanes.hip -- the "many frames" path: lane-serial decode, 64 independent subframes per wavefront.
//
//   P  clx_k_scan    one lane per multi-channel frame: parses channels 0..C-2 (headers + every Rice code,
//                    no output) to find the bit at which each later subframe starts -- subframe c+1
//                    begins where subframe c ends, there is no length field (frame.rs:705-742).
//   D  clx_k_lanes   one lane per subframe: subframe header, warm-up, LPC coefficients, Rice/Rice2
//                    residual decode, fixed/LPC synthesis, wasted-bits shift and stereo decorrelation
//                    (partner channel in lane^1, exchanged with DPP) fused in one pass; each lane streams
//                    its own output row with 16-byte stores.  No intermediate residual buffer in HBM.
//   F  clx_k_finalize  folds the per-frame error keys into clx_frame_result.
//
// Why two paths: a wavefront issues at most one instruction every ~4 cycles, and the wave-parallel
// decoder (clx_kernels.hip) spends ~10 wave-instructions per code to resolve code boundaries
// speculatively; decoding serially in each lane costs ~0.7 wave-instructions per code.  With
// thousands of frames in flight the lane-serial form wins by a wide margin; with a handful of
// frames the wave-parallel form has the lower latency.  clx_batch_run picks by batch shape.
//
// Both kernels mirror the reference's call sequence read for read (subframe.rs:29-91, 184-228,
// 236-415, 492-516, 651-721), so the first error in stream order is the one reported.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/claxon_hip.h"
#include "clx_device.h"

#define CLX_LERR(status, msg) (((uint32_t)(status) << 16) | (uint32_t)(msg))

// Per-lane MSB-first bit reader straight over the arena (served by the vector L1: a lane re-reads the
// same 64/128-byte line for ~100 codes).  `pos` counts bits from the frame's 4-byte aligned origin.
struct LaneReader {
    const uint8_t* arena;     // wave-uniform
    uint32_t origin;          // byte offset of the origin from `arena` (multiple of 4); arena_len < 4 GiB on this path
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
// P: locate subframes 1..C-1 of every multi-channel frame
// ------------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(64)
void clx_k_scan(const uint8_t* __restrict__ arena, const clx_dev_frame* __restrict__ frames,
                const uint32_t* __restrict__ multi, uint32_t n_multi,
                uint32_t* __restrict__ sf_start, uint32_t* __restrict__ errkey) {
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    if (t >= n_multi) return;
    const uint32_t f = multi[t];
    const clx_dev_frame fr = frames[f];
    LaneReader r;
    r.arena = arena;
    r.origin = (uint32_t)(fr.byte_off & ~3ull);
    const uint32_t o = 8u * (uint32_t)(fr.byte_off & 3ull);
    r.limit = o + fr.limit_bits;
    r.pos = o + 8u * (uint32_t)fr.header_bytes;
    r.err = (r.pos > r.limit) ? CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF) : 0u;
    const uint32_t bs = fr.block_size;
    for (uint32_t ch = 0; ch + 1u < fr.n_channels; ++ch) {
        const SfHead h = clx_lparse_sf_header(r, clx_channel_bps(fr, ch));
        if (!r.err) {
            if (h.kind == 0u) {                                    // decode_constant
                (void)clx_lread(r, h.sf_bps);
            } else if (h.kind == 1u) {                             // decode_verbatim
                if ((uint64_t)r.pos + (uint64_t)bs * h.sf_bps > (uint64_t)r.limit) r.err = CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
                else r.pos += bs * h.sf_bps;
            } else {
                if (bs < h.order) r.err = CLX_LERR(CLX_FORMAT_ERROR, h.kind == 2u ? CLX_MSG_FIXED_ORDER_GT_BLOCK : CLX_MSG_LPC_ORDER_GT_BLOCK);
                if (!r.err) {
                    if (r.pos + h.order * h.sf_bps > r.limit) r.err = CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
                    else r.pos += h.order * h.sf_bps;                // warm-up samples
                }
                if (!r.err && h.kind == 3u) {                      // decode_lpc, subframe.rs:669-701
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
                    uint32_t len = rh.per - h.order;
                    for (uint32_t part = 0; part < rh.n_part && !r.err; ++part) {
                        const uint32_t k = clx_lread_rice_param(r, rh.rice2);
                        const uint32_t k1 = k + 1u;
                        for (uint32_t i = 0; i < len && !r.err; ++i) {
                            const uint32_t v = clx_lpeek32(r, r.pos);
                            const uint32_t n = (uint32_t)__clz((int)v) + k1;
                            if (v != 0u && n <= 32u) {
                                r.pos += n;
                                if (r.pos > r.limit) r.err = CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
                            } else (void)clx_lrice_slow(r, k);
                        }
                        len = rh.per;
                    }
                }
            }
        }
        if (r.err) { clx_report_error(errkey, f, ch, r.err); return; }
        sf_start[fr.first_slot + ch + 1u] = r.pos;
    }
}

// ------------------------------------------------------------------------------------------------
// D: one lane per subframe, everything fused
// ------------------------------------------------------------------------------------------------
#define CLX_LB 8          // samples per block (two 16-byte stores per lane per block)

template <int OMAX, bool WIDE>
__device__ __forceinline__ int32_t clx_lpredict(const int32_t (&c)[OMAX], const int32_t (&hist)[OMAX], uint32_t shift) {
    if (WIDE) {
        int64_t acc = 0;
#pragma unroll
        for (int j = OMAX - 1; j >= 0; --j) acc += (int64_t)c[j] * (int64_t)hist[j];
        return (int32_t)(acc >> shift);
    } else {
        int32_t acc = 0;
#pragma unroll
        for (int j = OMAX - 1; j >= 0; --j) acc = __mul24(c[j], hist[j]) + acc;
        return acc >> shift;
    }
}

template <int OMAX, bool ALIGNED>
__device__ __forceinline__ void clx_lanes_body(LaneReader& r, const SfHead h, uint32_t bs, uint32_t decor, bool pair_ok,
                                               int32_t* __restrict__ row, uint32_t nmax, int lane,
                                               uint32_t* end_pos_out) {
    int32_t c[OMAX], hist[OMAX];
#pragma unroll
    for (int j = 0; j < OMAX; ++j) { c[j] = 0; hist[j] = 0; }
    const bool odd = (lane & 1) != 0;
    const bool any_decor = __any(pair_ok);
    const uint32_t n = r.err ? 0u : bs;              // a lane that failed in its header produces nothing

    // phases: 0 fixed-width fields (warm-up / verbatim), 1 rice, 2 constant, 3 idle
    uint32_t phase = 3u;
    int32_t cval = 0;
    uint32_t trans_at = 0xffffffffu;                 // sample index at which a predicted subframe switches to residuals
    if (n) {
        if (h.kind == 0u) { cval = clx_lread_signed(r, h.sf_bps); phase = 2u; }
        else if (h.kind == 1u) phase = 0u;
        else {
            if (bs < h.order) r.err = CLX_LERR(CLX_FORMAT_ERROR, h.kind == 2u ? CLX_MSG_FIXED_ORDER_GT_BLOCK : CLX_MSG_LPC_ORDER_GT_BLOCK);
            else { phase = 0u; trans_at = h.order; }
        }
    }
    uint32_t order = 0, shift = 0;                   // predictor becomes active at the transition
    uint32_t k = 0, k1 = 1, pcnt = 0, per = 0, parts_left = 0, rice2 = 0;
    int32_t lim = 0x7fffffff;                        // |s| range in which the 24-bit / i32 evaluation is exact; -1: use i64
    bool wide = false;                               // wave-uniform, sticky
    bool transitioned = false;

    // LPC parameters + residual header (subframe.rs:669-701, 241-277): executed once per predicted subframe, at the
    // sample index where the warm-up ends -- also when that index is the block size (no residual samples at all:
    // the residual header and its single partition parameter are still read, subframe.rs:509, 706).
    auto transition = [&]() {
        uint32_t cabs = 0;
        if (h.kind == 3u) {
            const uint32_t pm1 = clx_lread(r, 4);
            if (!r.err && pm1 == 15u) r.err = CLX_LERR(CLX_FORMAT_ERROR, CLX_MSG_QLP_PRECISION_INVALID);
            const uint32_t sh = clx_lread(r, 5);
            if (!r.err && (sh & 0x10u)) r.err = CLX_LERR(CLX_UNSUPPORTED, CLX_MSG_NEGATIVE_QLP_SHIFT);
            shift = sh & 0xfu;
            if (!r.err && r.pos + h.order * (pm1 + 1u) > r.limit) r.err = CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
#pragma unroll
            for (int j = 0; j < OMAX; ++j) {       // j-th coded coefficient applies to s[i-1-j] (subframe.rs:696-701)
                if ((uint32_t)j < h.order && !r.err) c[j] = clx_lread_signed(r, pm1 + 1u);
                cabs += (uint32_t)(c[j] < 0 ? -c[j] : c[j]);
            }
        } else {                                   // fixed predictors as taps on s[i-1-j] (subframe.rs:427-431)
            const uint32_t o = h.order;
            c[0] = o == 1u ? 1 : o == 2u ? 2 : o == 3u ? 3 : o == 4u ? 4 : 0;
            c[1] = o == 2u ? -1 : o == 3u ? -3 : o == 4u ? -6 : 0;
            c[2] = o == 3u ? 1 : o == 4u ? 4 : 0;
            c[3] = o == 4u ? -1 : 0;
            cabs = o == 1u ? 1u : o == 2u ? 3u : o == 3u ? 7u : o == 4u ? 15u : 0u;
        }
        order = h.order;
        lim = (h.sf_bps <= 24u && ((uint64_t)cabs << (h.sf_bps - 1u)) < (1ull << 31)) ? (int32_t)(1u << (h.sf_bps - 1u)) : -1;
        const ResHead rh = clx_lparse_residual_header(r, bs, h.order);
        rice2 = rh.rice2; per = rh.per; parts_left = rh.n_part;
        pcnt = 0;
        phase = 1u;
        transitioned = true;
        // the first partition holds per - order codes (subframe.rs:283); an empty first partition
        // (per == order) still carries its parameter
        if (!r.err) {
            k = clx_lread_rice_param(r, rice2); k1 = k + 1u; parts_left -= 1u; pcnt = per - h.order;
            while (!r.err && pcnt == 0u && parts_left != 0u) {
                k = clx_lread_rice_param(r, rice2); k1 = k + 1u; parts_left -= 1u; pcnt = per;
            }
        }
    };

    int32_t y[CLX_LB], xs[CLX_LB];
    for (uint32_t t0 = 0; t0 < nmax; t0 += CLX_LB) {
        int32_t h0[OMAX];
#pragma unroll
        for (int j = 0; j < OMAX; ++j) h0[j] = hist[j];
#pragma unroll
        for (int ii = 0; ii < CLX_LB; ++ii) {
            const uint32_t i = t0 + (uint32_t)ii;
            const bool live = (n != 0u) && !r.err;          // the transition may fall on i == n (see above)
            // ---- transition from warm-up to residuals
            if (__any(live && i == trans_at)) {
                if (live && i == trans_at) transition();
            }
            if (lim < 0 && live && i < n) wide = true;
            // ---- the sample's raw value
            int32_t x = cval;
            const bool livenow = (i < n) && !r.err;
            if (__any(livenow && phase == 0u)) {
                if (livenow && phase == 0u) x = clx_lread_signed(r, h.sf_bps);          // warm-up / verbatim (subframe.rs:397-415)
            }
            if (livenow && phase == 1u) {
                // one Rice code (subframe.rs:337-341)
                const uint32_t v = clx_lpeek32(r, r.pos);
                const uint32_t z = (uint32_t)__clz((int)v);
                const uint32_t nb = z + k1;
                uint32_t u;
                if (v != 0u && nb <= 32u) {
                    const uint32_t rem = k ? ((v << (z + 1u)) >> (32u - k)) : 0u;
                    u = (z << k) | rem;
                    r.pos += nb;
                    if (r.pos > r.limit) r.err = CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
                } else u = clx_lrice_slow(r, k);
                x = (int32_t)(u >> 1) ^ -(int32_t)(u & 1u);                              // rice_to_signed (subframe.rs:157-170)
                pcnt -= 1u;
                while (!r.err && pcnt == 0u && parts_left != 0u) {                       // next partition's parameter
                    k = clx_lread_rice_param(r, rice2); k1 = k + 1u; parts_left -= 1u; pcnt = per;
                }
            }
            xs[ii] = x;
        }
        // ---- predictor over the block: fast 24-bit evaluation, verified; exact i64 re-run when out of the proven range
        wide = __any(wide);
        bool redo = wide;
        if (!wide) {
#pragma unroll
            for (int ii = 0; ii < CLX_LB; ++ii) {
                const uint32_t i = t0 + (uint32_t)ii;
                const uint32_t ord_i = (i >= trans_at) ? order : 0u;
                const int32_t pred = clx_lpredict<OMAX, false>(c, hist, shift);
                const uint32_t use = (ord_i != 0u && i >= ord_i) ? 0xffffffffu : 0u;
                const int32_t s = (int32_t)((uint32_t)xs[ii] + ((uint32_t)pred & use));
#pragma unroll
                for (int j = OMAX - 1; j > 0; --j) hist[j] = hist[j - 1];
                hist[0] = s;
                y[ii] = s;
            }
            int32_t mx = y[0], mn = y[0];
#pragma unroll
            for (int ii = 1; ii < CLX_LB; ++ii) { mx = y[ii] > mx ? y[ii] : mx; mn = y[ii] < mn ? y[ii] : mn; }
            const bool in_range = (order == 0u) || t0 >= n || r.err != 0u || (mx < lim && mn >= -lim);
            if (!__all(in_range)) { redo = true; wide = true; }
        }
        if (redo) {
#pragma unroll
            for (int j = 0; j < OMAX; ++j) hist[j] = h0[j];
#pragma unroll
            for (int ii = 0; ii < CLX_LB; ++ii) {
                const uint32_t i = t0 + (uint32_t)ii;
                const uint32_t ord_i = (i >= trans_at) ? order : 0u;
                const int32_t pred = clx_lpredict<OMAX, true>(c, hist, shift);
                const uint32_t use = (ord_i != 0u && i >= ord_i) ? 0xffffffffu : 0u;
                const int32_t s = (int32_t)((uint32_t)xs[ii] + ((uint32_t)pred & use));
#pragma unroll
                for (int j = OMAX - 1; j > 0; --j) hist[j] = hist[j - 1];
                hist[0] = s;
                y[ii] = s;
            }
        }
        // ---- wasted-bits shift (subframe.rs:216-225), stereo decorrelation (frame.rs:319-389), store
#pragma unroll
        for (int ii = 0; ii < CLX_LB; ++ii) y[ii] = (int32_t)((uint32_t)y[ii] << h.wasted);
        if (any_decor) {
#pragma unroll
            for (int ii = 0; ii < CLX_LB; ++ii) {
                const int32_t mine = y[ii];
                const int32_t other = __builtin_amdgcn_update_dpp(0, mine, 0xB1, 0xF, 0xF, false);   // lane ^ 1
                const int32_t a = odd ? other : mine;
                const int32_t bb = odd ? mine : other;
                int32_t v = mine;
                if (pair_ok) {
                    if (decor == CLX_CH_LEFT_SIDE) { if (odd) v = (int32_t)((uint32_t)a - (uint32_t)bb); }
                    else if (decor == CLX_CH_RIGHT_SIDE) { if (!odd) v = (int32_t)((uint32_t)a + (uint32_t)bb); }
                    else {
                        const int32_t m = (int32_t)(((uint32_t)a << 1) | ((uint32_t)bb & 1u));
                        v = odd ? ((int32_t)((uint32_t)m - (uint32_t)bb) >> 1) : ((int32_t)((uint32_t)m + (uint32_t)bb) >> 1);
                    }
                }
                y[ii] = v;
            }
        }
        if (ALIGNED) {
#pragma unroll
            for (int q = 0; q < CLX_LB / 4; ++q)
                if (t0 + 4u * q < n) *reinterpret_cast<int4*>(row + t0 + 4 * q) = make_int4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
        } else {
#pragma unroll
            for (int ii = 0; ii < CLX_LB; ++ii) if (t0 + (uint32_t)ii < n) row[t0 + ii] = y[ii];
        }
    }
    // subframes whose warm-up fills the whole block (order == block size) switch after the last sample
    {
        const bool late = (n != 0u) && !r.err && trans_at != 0xffffffffu && trans_at == n && !transitioned;
        if (__any(late)) { if (late) transition(); }
    }
    *end_pos_out = r.pos;
}

extern "C" __global__ __launch_bounds__(64)
void clx_k_lanes(const uint8_t* __restrict__ arena, const clx_dev_frame* __restrict__ frames,
                 const uint32_t* __restrict__ slot_frame, uint32_t n_slots,
                 const uint32_t* __restrict__ sf_start, int32_t* __restrict__ out,
                 uint32_t* __restrict__ errkey, uint64_t* __restrict__ end_bits) {
    const int lane = (int)threadIdx.x;
    const uint32_t slot = blockIdx.x * 64u + (uint32_t)lane;
    uint32_t f = 0xffffffffu;
    if (slot < n_slots) f = slot_frame[slot];
    clx_dev_frame fr;
    fr.byte_off = 0; fr.out_off = 0; fr.limit_bits = 0; fr.first_slot = 0; fr.header_bytes = 0; fr.block_size = 0;
    fr.n_channels = 0; fr.channel_assignment = 0; fr.bps = 1; fr.flags = 0;
    if (f != 0xffffffffu) fr = frames[f];
    const uint32_t ch = (f != 0xffffffffu) ? slot - fr.first_slot : 0u;
    uint32_t bs = fr.block_size;

    LaneReader r;
    r.arena = arena;
    r.origin = (uint32_t)(fr.byte_off & ~3ull);
    const uint32_t o = 8u * (uint32_t)(fr.byte_off & 3ull);
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
    int32_t* const row = out + (active ? fr.out_off + (uint64_t)ch * fr.block_size : 0ull);
    const bool al = !active || ((((uintptr_t)row) & 15u) == 0u && (bs & 3u) == 0u);
    uint32_t end_pos = r.pos;
    if (nmax != 0u) {
        if (__all(al)) {
            if (omax <= 4u)       clx_lanes_body<4, true>(r, h, bs, decor, pair_ok, row, nmax, lane, &end_pos);
            else if (omax <= 8u)  clx_lanes_body<8, true>(r, h, bs, decor, pair_ok, row, nmax, lane, &end_pos);
            else if (omax <= 12u) clx_lanes_body<12, true>(r, h, bs, decor, pair_ok, row, nmax, lane, &end_pos);
            else                  clx_lanes_body<32, true>(r, h, bs, decor, pair_ok, row, nmax, lane, &end_pos);
        } else {
            if (omax <= 4u)       clx_lanes_body<4, false>(r, h, bs, decor, pair_ok, row, nmax, lane, &end_pos);
            else if (omax <= 8u)  clx_lanes_body<8, false>(r, h, bs, decor, pair_ok, row, nmax, lane, &end_pos);
            else if (omax <= 12u) clx_lanes_body<12, false>(r, h, bs, decor, pair_ok, row, nmax, lane, &end_pos);
            else                  clx_lanes_body<32, false>(r, h, bs, decor, pair_ok, row, nmax, lane, &end_pos);
        }
    }
    if (active) {
        if (r.err) clx_report_error(errkey, f, ch, r.err);
        else if (ch + 1u == fr.n_channels) end_bits[f] = (uint64_t)(end_pos - o);
    }
}

// ------------------------------------------------------------------------------------------------
// F: error keys -> clx_frame_result
// ------------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256)
void clx_k_finalize(const uint32_t* __restrict__ errkey, const uint64_t* __restrict__ end_bits, uint32_t n_frames,
                    clx_frame_result* __restrict__ results) {
    const uint32_t f = blockIdx.x * 256u + threadIdx.x;
    if (f >= n_frames) return;
    const uint32_t key = errkey[f];
    clx_frame_result r;
    if (key == 0xffffffffu) { r.status = CLX_OK; r.msg = CLX_MSG_NONE; r.end_bit = end_bits[f]; }
    else { r.status = (int32_t)((key >> 16) & 0xffu); r.msg = key & 0xffffu; r.end_bit = 0; }
    results[f] = r;
}
