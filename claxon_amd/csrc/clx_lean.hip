// clx_lean.hip -- the lean builds of the fused lane kernel (D): what 16-bit (and 24-bit) audio is made of, and nothing else.
//
//   DL  clx_k_lean    one lane per subframe like clx_k_lanes (clx_lanes.hip), for the waves of 64 subframes in which every live
//                     lane decodes a subframe of <= 16-bit audio (the side channel has 17) with at most 12 taps, in 16-byte aligned
//                     rows of one common block size that is a multiple of 16.  Such a wave marks its group as taken; the kernels
//                     behind it skip taken groups.
//   DL4 clx_k_lean24  the same for <= 24-bit audio (25) and <= 32 taps, on the groups clx_k_lean left: the split form of the
//                     turn (two v_dot2 chains on 12 / 16-bit pieces of every sample instead of 64-bit multiply-adds).
//
// Why more builds: clx_k_lanes carries every tier of every case in one register allocation (195 VGPRs: two waves per SIMD) and
// spends ~40 instructions per sample (148 with the i64 predictor).  These keep ONE fast tier and one slow one:
//   * lean turn, 16 samples: four register windows of four Rice codes each (one LDS read per window, funnel shifts between
//     the codes); the predictor on 16-bit packed history, two taps per v_dot2_i32_i16 (subframe.rs:559-582's loop; exact while
//     every history sample lies in [-2^15, 2^15) and sum|c| * 2^15 < 2^31 -- checked on the data per turn, not assumed);
//     stereo decorrelation through DPP; two turns' 64 x 16 samples leave together as whole 128-byte lines, 8 rows per store
//     instruction.  It decodes first and asks afterwards: one wave vote per turn.
//   * slow turn, 16 samples: a rolled per-sample loop over the generic reader with the i64 predictor (subframe.rs:586-614's
//     arithmetic), which handles every rare case in line -- partition edges inside a four, escape codes, codes longer than
//     32 bits, the end of the frame, history outside the 16-bit range.  The wave returns to lean turns as soon as every
//     lane's history is back inside.
// The history lives in ONE register array that holds packed pairs (lean) or i32 samples (slow), converted where the tier
// changes; coefficients stay packed (the slow tier unpacks them per tap).  Per-lane LDS: a ring of CLN_RING stream dwords
// (+ 4 mirrored), 128 bytes of output stage and the row's address.
//
// Mirrors the reference read for read like clx_k_lanes: subframe.rs:29-91, 184-228, 236-380, 492-516, 651-721; frame.rs:319-389.
#ifndef CLN_RING
#define CLN_RING 24u                    // stream dwords per lane in the ring (a multiple of 4).  32 would make the slot a bit field, but the
                                        // 2 KiB more per wave cost more than the two instructions per window (0.208 against 0.202 ms per step)
#endif
#define CLN_ROW (CLN_RING + 4u)
#define CLN_SCHED_FENCE() CLX_SCHED_BARRIER()         // slots RING .. RING+3 mirror slots 0 .. 3: a window of five dwords never wraps

// The ring is slot-major -- ring[slot][lane] -- so that a lane's dwords all sit in the lane's own LDS bank (bank = lane mod 32):
// the lanes of a wave read at unrelated slots (their streams advance at their own pace), and in a lane-major layout those reads
// collide three to four deep (measured: 56 % of the LDS's active cycles were bank conflicts).
// 7 168 + 8 192 = 15 360 bytes per wave: TEN waves per CU.  (LDS is handed out in granules: with the wave's 64 row addresses beside them --
// 15 872 bytes, rounds 3 and 4 -- the per-wave timeline shows nine, not the ten that 160 KiB / 15.5 KiB promises: profiles/r05_timeline_lean.txt.
// The rows' places travel in registers now, LMover.)
struct LeanLds {
    uint32_t ring[CLN_ROW][64];
    int4 stage[2][64][4];               // two turns' 64 x 16 output samples: int4 [tile][row ^ tile][piece ^ swizzle] (cln_mine)
};
#define CLN_AT(col, slot) ((col)[(slot) * 64u])        // dword `slot` of the lane whose column `col` is

// s in [0, 2 * RING) -> s mod RING
__device__ __forceinline__ uint32_t cln_wrap(uint32_t s) {
    if ((CLN_RING & (CLN_RING - 1u)) == 0u) return s & (CLN_RING - 1u);
    const uint32_t t = s - CLN_RING;
    return t < s ? t : s;
}

struct LRing {
    uint32_t origin;        // byte offset of the lane's 16-byte aligned stream origin in the arena
    uint32_t fill, fs;      // stream dwords [fill - RING, fill) are in the ring; fs = fill mod RING (the slot `fill` goes to)
    uint32_t np;            // granules requested at the last pump (0..3), in pa / pb / pc
    uint4 pa, pb, pc;
};
// ---- the frame's CRC-16, gathered by the decode lanes from the words they put into their rings anyway (clx_crct.h: the frame's
// polynomial modulo x^15 + x + 1 and its parity, eight cheap instructions per word, no table).  A lane's share is the words
// [da, db) of its frame -- from the granule in which its subframe starts to the granule in which the next one does; the first
// channel's from the frame's first granule, the last channel's to the frame's end -- so that the shares of a frame's lanes tile
// the frame.  Granules are taken in order, each once: `next` is the word the share has been taken up to.  What does not come
// through the ring (the prologue's bytes, what a slow turn walked over, the end of the share) is fetched again (cln_crc_catchup).
#define CLN_CRC_NONE 0xfffffff0u
struct LCrc {
    clx_crct c;
    uint32_t next;          // [da, next) has been taken (a multiple of 4)
    uint32_t db;            // the end of the share (a multiple of 4: whole granules); CLN_CRC_NONE: the lane gathers nothing that is kept
};
__device__ __forceinline__ void cln_crc_take(LCrc& C, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) {
    clx_crct_word(C.c, b0); clx_crct_word(C.c, b1); clx_crct_word(C.c, b2); clx_crct_word(C.c, b3);
}
// the granules [next, upto) straight from the arena
__device__ __forceinline__ void cln_crc_catchup(const clx_buf& buf, uint32_t origin, LCrc& C, uint32_t upto) {
#pragma unroll 1
    while (C.next < upto) {
        const uint4 v = clx_buf_load16(buf, origin + 4u * C.next);
        cln_crc_take(C, __builtin_bswap32(v.x), __builtin_bswap32(v.y), __builtin_bswap32(v.z), __builtin_bswap32(v.w));
        C.next += 4u;
    }
}
// one granule of which only the bytes [lo, hi) count (the frame's first and last granule: its neighbours' bytes read as zeros)
__device__ __forceinline__ void cln_crc_take_masked(LCrc& C, const uint4 v, uint32_t lo, uint32_t hi) {
    const uint32_t w[4] = { __builtin_bswap32(v.x), __builtin_bswap32(v.y), __builtin_bswap32(v.z), __builtin_bswap32(v.w) };
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) {
        // bytes 4j .. 4j+3 of the granule are the word's bytes from the top down
        const uint32_t a = lo > 4u * j ? (lo - 4u * j < 4u ? lo - 4u * j : 4u) : 0u;        // leading bytes to drop
        const uint32_t b = hi > 4u * j ? (hi - 4u * j < 4u ? hi - 4u * j : 4u) : 0u;        // bytes up to which to keep
        const uint32_t ma = a >= 4u ? 0u : 0xffffffffu >> (8u * a);
        const uint32_t mb = b >= 4u ? 0xffffffffu : ~(0xffffffffu >> (8u * b));
        clx_crct_word(C.c, w[j] & ma & mb);
    }
}
// mode: 0 no CRC; 1 the granule is the share's next one and lies inside it (the wave has checked that for all that land this turn);
// 2 ask per granule.  f: the granule's first word.
__device__ __forceinline__ void cln_put(uint32_t* row, uint32_t s, const uint4 v, LCrc& C, uint32_t f, int mode) {
    const uint32_t b0 = __builtin_bswap32(v.x), b1 = __builtin_bswap32(v.y), b2 = __builtin_bswap32(v.z), b3 = __builtin_bswap32(v.w);
    CLN_AT(row, s) = b0; CLN_AT(row, s + 1u) = b1; CLN_AT(row, s + 2u) = b2; CLN_AT(row, s + 3u) = b3;
    if (s == 0u) { CLN_AT(row, CLN_RING) = b0; CLN_AT(row, CLN_RING + 1u) = b1; CLN_AT(row, CLN_RING + 2u) = b2; CLN_AT(row, CLN_RING + 3u) = b3; }
    if (mode == 1) cln_crc_take(C, b0, b1, b2, b3);
    else if (mode == 2) {
        if (f == C.next && f + 4u <= C.db) { cln_crc_take(C, b0, b1, b2, b3); C.next = f + 4u; }
    }
}
// synchronous fill from the granule that holds dword `d` (the start of the steady state, and after a slow turn)
__device__ __forceinline__ void cln_reset(const clx_buf& buf, LRing& g, uint32_t* row, uint32_t d, LCrc& C, bool crc) {
    const uint32_t f0 = d & ~3u;
    const uint32_t s0 = f0 % CLN_RING;
    if (crc) {
        // what lies between the share's taken part and the ring's new start never comes through the ring
        if (C.db == CLN_CRC_NONE) C.next = f0;
        else cln_crc_catchup(buf, g.origin, C, f0 < C.db ? f0 : C.db);
    }
#pragma unroll
    for (uint32_t h = 0; h < CLN_RING / 4u; h += 3u) {          // three granules at a time: the loads' registers are short-lived
        uint4 t[3];
#pragma unroll
        for (uint32_t q = 0; q < 3u; ++q) if (h + q < CLN_RING / 4u) t[q] = clx_buf_load16(buf, g.origin + 4u * (f0 + 4u * (h + q)));
#pragma unroll
        for (uint32_t q = 0; q < 3u; ++q) if (h + q < CLN_RING / 4u) cln_put(row, cln_wrap(s0 + 4u * (h + q)), t[q], C, f0 + 4u * (h + q), crc ? 2 : 0);
    }
    g.fill = f0 + CLN_RING; g.fs = s0; g.np = 0;
}
// once per turn: land what was requested a turn ago, request what fits now (three granules: 24 bits per code sustained -- a
// verbatim subframe of 17-bit samples takes 272 bits per turn)
// (The landing waits for EVERYTHING the wave has in flight -- the compiler cannot count across the branches: s_waitcnt vmcnt(0) --,
// the turn's tile stores included, which are issued right in front of it.  Landing in front of the stores instead, when all that is
// in flight is a whole turn old, makes one run alone 4 % faster and a saturated machine 9 % slower (one merged launch of nine runs:
// 1.27 -> 1.39 ms; tools/gpu_ab_sat.sh): the wait is what paces the waves' stores.)
// (MAXP: granules a pump may have in flight -- three; two in the 12-tap build of the 16-bit tier, whose loop has no registers for the third:
//  cln_body)
template <int MAXP = 3>
__device__ __forceinline__ void cln_land(LRing& g, uint32_t* row, LCrc& C, bool crc) {
    // (one vote per turn instead of a question per granule: up to three granules land, all of them the shares' next ones)
    const int mode = !crc ? 0 : __all(C.next == g.fill && g.fill + 4u * (uint32_t)MAXP <= C.db) ? 1 : 2;
    if (g.np >= 1u) { cln_put(row, g.fs, g.pa, C, g.fill, mode); g.fill += 4u; g.fs = cln_wrap(g.fs + 4u); }
    if (g.np >= 2u) { cln_put(row, g.fs, g.pb, C, g.fill, mode); g.fill += 4u; g.fs = cln_wrap(g.fs + 4u); }
    if (MAXP >= 3 && g.np >= 3u) { cln_put(row, g.fs, g.pc, C, g.fill, mode); g.fill += 4u; g.fs = cln_wrap(g.fs + 4u); }
    if (mode == 1) C.next = g.fill;
    g.np = 0;
}
// `more` (wave-uniform): a second and a third granule may be asked for.  Calm waves (cln_pump_now) say so at every fourth pump
// only: their lanes use 1.3 granules per pump, out of step with each other, so the second landing block and its CRC step ran at
// every pump for the quarter of the lanes that had two in flight -- with the second granules asked for TOGETHER it runs at a quarter
// of the pumps for most of them.  In between a lane falls at most 1.5 granules behind (6 bits per sample: the calm limit), which a
// ring of 24 dwords holds beside the two turns' worth it must have landed.
template <int MAXP = 3>
__device__ __forceinline__ void cln_request(const clx_buf& buf, LRing& g, uint32_t p, bool more) {
    const uint32_t d = (p - 1u) >> 5;                                   // the oldest dword a window may still read
    const int32_t room = (int32_t)(CLN_RING + d - g.fill);                // slots that hold dwords before d
    g.np = 0;
    if (room >= 4) { g.pa = clx_buf_load16(buf, g.origin + 4u * g.fill); g.np = 1u; }
    if (!more) return;
    if (room >= 8) { g.pb = clx_buf_load16(buf, g.origin + 4u * g.fill + 16u); g.np = 2u; }
    if (MAXP >= 3 && room >= 12) { g.pc = clx_buf_load16(buf, g.origin + 4u * g.fill + 32u); g.np = 3u; }
}
// The ring is pumped (landing + requests) every OTHER turn in CALM waves: at 5 bits per code a lane uses a granule in 1.6 turns,
// and the wave paid for a landing, a CRC step and three request blocks per turn as soon as ONE lane had something in flight.
// Calm is decided once, where the subframes' sizes are known (cln_kernel: every lane's subframe holds at most 6 bits per
// sample; the scan: its frames do): such lanes keep two turns' worth landed beyond their cursors in a ring of 24 dwords.  A burst
// inside a calm subframe runs the ring dry and is refilled on the spot, as always.  Deciding it turn by turn from the lanes' rates
// was measured too (tools/r04_call8.sh, r04_call9.sh): the same gain on config 3, but the bookkeeping cost the waves that are never
// calm (configs 4 and 5) 1.5-5 %.
__device__ __forceinline__ bool cln_pump_now(bool calm, bool even) { return !calm || even; }
// slot of stream dword d, for d in [fill - RING, fill)
__device__ __forceinline__ uint32_t cln_slot(const LRing& g, uint32_t d) {
    const uint32_t t = g.fs + d - g.fill;                                 // fs - (fill - d), negative (wrapped) when it wraps
    const uint32_t t2 = t + CLN_RING;
    return t2 < t ? t2 : t;                                               // (unsigned: exactly one of them is in range)
}
// 32 bits at bit position p (p >= 1), left aligned, for the partition parameters
__device__ __forceinline__ uint32_t cln_peek32(const uint32_t* row, const LRing& g, uint32_t p) {
    const uint32_t s = cln_slot(g, (p - 1u) >> 5);
    return clx_alignbit(CLN_AT(row, s), CLN_AT(row, s + 1u), 0u - p);
}

// the part of the subframe's state the turns work on
struct LCur { uint32_t p, k, pcnt, next, parts; };
// (the lean turn also carries c1 = 31 - k, kept opaque so that 31 - k - z stays one subtraction per code)

// What a lane does in the steady state (fixed once the prologue is over): Rice codes, verbatim fields (subframe.rs:397-415) or a
// constant (382-394).  The three kinds share one instruction stream; the masks that make them do so are compiled in only for
// waves that hold a lane of the rarer kinds (MODE 1).
struct LKind {
    bool rice, verb;
    uint32_t bitmask;        // all ones where the lane consumes bits (Rice, verbatim), 0 for a constant
    uint32_t ricemask;       // all ones for Rice lanes: only their codes can outgrow the window register
    uint32_t verbmask;       // all ones for verbatim lanes
    uint32_t cor;            // the constant, OR-ed in for constant lanes
    uint32_t vsh;            // 32 - (width of a verbatim field)
    uint32_t vshm;           // vsh for verbatim lanes, 0 for the others: what is left of the window register behind a field
};

// One sample's raw value the careful way (generic reader over global memory; every rare case in line): partition parameters
// (subframe.rs:314-319 / 362-367), Rice codes of any length, the end of the frame; a verbatim field; the constant.  Steady state
// only (after the transition).
__device__ __forceinline__ int32_t cln_careful_code(LaneReader& r, LCur& c, uint32_t per, uint32_t rice2, const LKind& K) {
    r.pos = c.p;
    if (!K.rice) {
        int32_t x = (int32_t)K.cor;
        if (K.verb) { x = clx_lread_signed(r, 32u - K.vsh); c.p = r.pos; }
        return x;
    }
    while (!r.err && c.pcnt == 0u && c.parts != 0u) {
        c.k = clx_lread_rice_param(r, rice2); c.parts -= 1u; c.pcnt = c.next; c.next = per;
    }
    int32_t x = 0;
    if (!r.err) {
        const uint32_t v = clx_lpeek32(r, r.pos);
        const uint32_t z = (uint32_t)__clz((int)v);
        const uint32_t nb = z + c.k + 1u;
        uint32_t u;
        if (v != 0u && nb <= 32u) {
            const uint32_t rem = (v >> ((32u - nb) & 31u)) & ((1u << c.k) - 1u);
            u = (z << c.k) | rem;
            r.pos += nb;
            if (r.pos > r.limit) r.err = CLX_LERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
        } else u = clx_lrice_slow(r, c.k);
        x = (int32_t)(u >> 1) ^ -(int32_t)(u & 1u);                      // rice_to_signed (subframe.rs:157-170)
        c.pcnt -= 1u;
    }
    c.p = r.pos;
    return x;
}

// ---- P: the scan in the lean kernels' form --------------------------------------------------------------------------------
// clx_k_scan: one lane per multi-channel frame walks channels 0 .. C-2 (headers + the LENGTH of every Rice code, no output) to find
// the bit at which each later subframe starts -- subframe c+1 begins where subframe c ends, there is no length field
// (frame.rs:705-742).  The round-2 build (clx_k_scan_general, clx_lanes.hip: lane-major ring, blocks of four codes with a vote
// each) cost 14 vector instructions per code; this one runs the lean kernels' turn without its predictor and output: the
// slot-major ring, sixteen codes per turn (four register windows of four codes; per code v_ffbh, one subtraction, half a
// v_max3, one addition and one and a half v_alignbit), one vote per turn, the careful reader for everything rare.
// One turn of sixteen code lengths.  Returns 1 (cursor advanced for the live lanes), 0 (a rare case: the careful steps'), -2
// (nothing but the ring having run dry).
// ALL: every lane of the wave is live (the caller's vote): the masks that let lanes ride along are compiled out.
template <bool EDGE, bool ALL>
__device__ __forceinline__ int cln_scan_turn(const uint32_t* row, const LRing& g, LCur& cur, uint32_t per, uint32_t rice2, uint32_t limit, bool live_) {
    const bool live = ALL || live_;
    LCur c = cur;
    if (!live) c.pcnt = 0x7fffff00u;                  // (lanes that skip nothing never meet a partition edge)
    uint32_t c1 = 31u - c.k;
    CLX_OPAQUE(c1);
    bool bad = false;
    uint32_t zmax = 0;                                // the longest run of zeros (v_ffbh as it comes: all ones for an empty register) -- more
                                                      // than 31 - k: a code > 32 bits (round 5; the lean turn's comment)
    uint32_t pw = c.p;
    const uint32_t pb = 4u + rice2, esc = rice2 ? 31u : 15u;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (EDGE) {                                   // a partition that starts exactly here: its parameter comes first
            const bool at = c.pcnt == 0u;
            if (clx_any(at)) {
                const uint32_t pv = cln_peek32(row, g, c.p);
                if (at) {
                    c.k = pv >> (32u - pb);
                    bad = bad || c.k == esc || c.parts == 0u || c.next == 0u;
                    c.p += pb; c.parts -= 1u; c.pcnt = c.next; c.next = per;
                    c1 = 31u - c.k;
                }
                CLX_OPAQUE(c1);
            }
            bad = bad || c.pcnt < 4u;                 // (a partition edge inside the four codes: the careful steps')
        }
        c.pcnt -= 4u;
        pw = c.p;
        uint32_t wa, wb, wc, wd;
        {
            const uint32_t s = cln_slot(g, (c.p - 1u) >> 5);
            const uint32_t w0 = CLN_AT(row, s), w1 = CLN_AT(row, s + 1u), w2 = CLN_AT(row, s + 2u), w3 = CLN_AT(row, s + 3u), w4 = CLN_AT(row, s + 4u);
            const uint32_t sh = 0u - c.p;
            wa = clx_alignbit(w0, w1, sh); wb = clx_alignbit(w1, w2, sh); wc = clx_alignbit(w2, w3, sh); wd = clx_alignbit(w3, w4, sh);
        }
        uint32_t shsum = 0;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const uint32_t z = clx_ffbh(wa);
            const int32_t sh = (int32_t)(c1 - z);     // 32 - (z + 1 + k): what is left of the window behind the code
            zmax = z > zmax ? z : zmax;
            shsum += (uint32_t)sh;
            wa = clx_alignbit(wa, wb, (uint32_t)sh); wb = clx_alignbit(wb, wc, (uint32_t)sh); wc = clx_alignbit(wc, wd, (uint32_t)sh); wd = clx_alignbit(wd, 0u, (uint32_t)sh);
        }
        if (EDGE) { bad = bad || zmax > c1; zmax = 0; }      // (the parameter may change with the next four)
        c.p += 128u - shsum;
    }
    const bool toolong = !EDGE && zmax > c1;          // (one parameter for the whole turn)
    const bool covered = ((pw - 1u) >> 5) + 5u <= g.fill && c.p <= limit;
    const bool ok = !live || (!bad && !toolong && covered);
    if (__all(ok)) {
        const uint32_t p_in = cur.p;
        cur = c;
        if (!live) cur.p = p_in;
        return 1;
    }
    if (__any(live && (bad || toolong || c.p > limit))) return 0;
    return -2;
}

// The same sixteen code lengths out of four 32-BIT windows (round 5).  The scan needs no remainder -- only where each code's
// terminating one sits -- so one window register serves four codes as long as their four TERMINATORS lie in its first 31 bits, whatever
// lies behind the last of them: per code v_ffbh, one addition (z + k + 1) and one v_lshl_or (the window moved on, a sentinel one
// OR-ed in at bit 0 so that the register is never zero: v_ffbh needs no clamp), against v_ffbh, v_min, a subtraction and one and a half
// funnel shifts over four registers above; per window two ring dwords instead of five.  The sentinel marks the end of what the
// register knows: a terminator found AT or behind bit 31 is not believed -- the sum of the lengths then says so (bits in front of the
// fourth terminator > 30) and the turn is left to the 128-bit form.  Worth trying while the codes are short (k <= CLN_SHORT_K: the caller's vote).
// Returns 1 / 0 / -2 like cln_scan_turn, and -1: a window was too short, nothing else was wrong.
#ifndef CLN_SHORT_K
#define CLN_SHORT_K 5u
#endif
template <bool EDGE, bool ALL>
__device__ __forceinline__ int cln_scan_turn_short(const uint32_t* row, const LRing& g, LCur& cur, uint32_t per, uint32_t rice2, uint32_t limit, bool live_) {
    const bool live = ALL || live_;
    LCur c = cur;
    if (!live) c.pcnt = 0x7fffff00u;                  // (lanes that skip nothing never meet a partition edge)
    uint32_t k1 = c.k + 1u;
    bool bad = false;
    uint32_t far = 0;                                 // the most bits in front of a window's fourth terminator
    uint32_t pw = c.p;
    const uint32_t pb = 4u + rice2, esc = rice2 ? 31u : 15u;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (EDGE) {                                   // a partition that starts exactly here: its parameter comes first
            const bool at = c.pcnt == 0u;
            if (clx_any(at)) {
                const uint32_t pv = cln_peek32(row, g, c.p);
                if (at) {
                    c.k = pv >> (32u - pb);
                    bad = bad || c.k == esc || c.parts == 0u || c.next == 0u;
                    c.p += pb; c.parts -= 1u; c.pcnt = c.next; c.next = per;
                    k1 = c.k + 1u;
                }
            }
            bad = bad || c.pcnt < 4u;                 // (a partition edge inside the four codes: the careful steps')
        }
        c.pcnt -= 4u;
        pw = c.p;
        const uint32_t s = cln_slot(g, (c.p - 1u) >> 5);
        uint32_t w = clx_alignbit(CLN_AT(row, s), CLN_AT(row, s + 1u), 0u - c.p) | 1u;
        const uint32_t l1 = (uint32_t)__builtin_clz(w) + k1; w = (w << (l1 & 31u)) | 1u;       // (w is never zero: the sentinel)
        const uint32_t l2 = (uint32_t)__builtin_clz(w) + k1; w = (w << (l2 & 31u)) | 1u;
        const uint32_t l3 = (uint32_t)__builtin_clz(w) + k1; w = (w << (l3 & 31u)) | 1u;
        const uint32_t e4 = l1 + l2 + l3 + (uint32_t)__builtin_clz(w);                          // bits in front of the fourth terminator
        far = e4 > far ? e4 : far;
        c.p += e4 + k1;
    }
    const bool fits = far <= 30u;
    const bool covered = ((pw - 1u) >> 5) + 2u <= g.fill && c.p <= limit;
    const bool ok = !live || (!bad && fits && covered);
    if (__all(ok)) {
        const uint32_t p_in = cur.p;
        cur = c;
        if (!live) cur.p = p_in;
        return 1;
    }
    if (__any(live && !bad && !fits)) return -1;      // (what was parsed behind a window that was too short means nothing)
    if (__any(live && (bad || c.p > limit))) return 0;
    return -2;
}

// one turn of the scan, in the form the wave's codes allow: four codes out of a 32-bit window while every live lane's parameter
// is small (cln_scan_turn_short), else -- or when a window was too short for its four -- out of 128 bits
template <bool ALL>
__device__ __forceinline__ int cln_scan_turns(const uint32_t* row, const LRing& g, LCur& cur, uint32_t per, uint32_t rice2, uint32_t limit, bool live) {
    const bool edge = clx_any((ALL || live) && cur.pcnt < 16u);
    int done = -1;
    if (__all((!ALL && !live) || cur.k <= CLN_SHORT_K)) {
        if (edge) done = cln_scan_turn_short<true, ALL>(row, g, cur, per, rice2, limit, live);
        else      done = cln_scan_turn_short<false, ALL>(row, g, cur, per, rice2, limit, live);
        CLX_STAT(27, done > 0); CLX_STAT(28, done == -1);
    }
    if (done == -1) {
        if (edge) done = cln_scan_turn<true, ALL>(row, g, cur, per, rice2, limit, live);
        else      done = cln_scan_turn<false, ALL>(row, g, cur, per, rice2, limit, live);
        CLX_STAT(29, done > 0);
    }
    return done;
}

// The scan of one wave's 64 multi-channel frames (wave `bx` of run R); ring0: a ring of CLN_ROW x 64 dwords of LDS.  clx_k_scan's body, and
// the scan tickets' of clx_k_pool.
__device__ __forceinline__ void cln_scan_wave(uint32_t* ring0, const clx_run& R, const clx_dev_frame* __restrict__ frames,
                                              const uint32_t* __restrict__ multi, uint32_t n_multi, uint32_t bx, uint32_t run_idx, int lane) {
    CLX_TL_BEGIN();
    const uint8_t* const arena = R.arena;
    const uint64_t arena_alloc_len = R.alloc_len;
    uint32_t* const sf_start = R.sf_start;
    uint32_t* const errkey = R.errkey;
    uint32_t* const row = ring0 + lane;
    const uint32_t t = bx * 64u + (uint32_t)lane;
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
    // (the ring addresses the arena with 32-bit offsets through a buffer descriptor: a frame whose window could reach beyond them
    //  takes the careful steps only -- the lane path is not selected for arenas of 4 GiB anyway)
    const bool ring_able = (uint64_t)r.origin + 4ull * ((uint64_t)r.limit / 32ull + 16ull) < 0xffffffffull;
    const clx_buf buf = clx_make_buf(arena, (uint32_t)arena_alloc_len);
    LRing g;
    g.origin = r.origin; g.fill = 0; g.fs = 0; g.np = 0; g.pa = make_uint4(0u, 0u, 0u, 0u); g.pb = g.pa; g.pc = g.pa;
    LCrc NC;                                             // (the scan gathers no CRC: the decode lanes read the same bytes again)
    NC.c.r = 0u; NC.c.x = 0u; NC.next = 0u; NC.db = CLN_CRC_NONE;
    LKind KR;                                            // (the careful reader's view: every lane skips Rice codes)
    KR.rice = true; KR.verb = false; KR.bitmask = 0xffffffffu; KR.ricemask = 0xffffffffu; KR.verbmask = 0u; KR.cor = 0u; KR.vsh = 0u; KR.vshm = 0u;
    const uint32_t bs = fr.block_size;
    // (a calm wave -- no frame above 6 bits per sample over all its channels: the ring is pumped every other turn, cln_pump_now)
    const bool calm = CLN_RING >= 24u && __all(!active || fr.limit_bits <= 6u * bs * (uint32_t)fr.n_channels);      // (a shorter ring is pumped at every turn)
    uint32_t nch = active ? (uint32_t)fr.n_channels - 1u : 0u;       // channels to scan
    uint32_t nch_max = nch;
#pragma unroll
    for (int sx = 32; sx >= 1; sx >>= 1) { const uint32_t a = __shfl_xor(nch_max, sx, 64); nch_max = a > nch_max ? a : nch_max; }

    bool k_special = false;                              // the frame's content class for clx_k_compose (CLX_FKEY): a constant / verbatim
    uint32_t k_omax = 0;                                 // subframe among its channels, the highest predictor order
    for (uint32_t ch = 0; ch < nch_max; ++ch) {
        const bool on = ch < nch && !r.err;
        // ---- headers (per lane, generic reader)
        uint32_t codes = 0, first = 0, per = 0, parts_left = 0, rice2 = 0, order = 0;
        if (on) {
            const SfHead h = clx_lparse_sf_header(r, clx_channel_bps(fr, ch));
            if (!r.err) { k_special = k_special || h.kind < 2u; k_omax = h.order > k_omax ? h.order : k_omax; }
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
        LCur cur = { on ? r.pos : 32u, 0u, 0u, first, parts_left };      // (a lane that scans nothing rides along from a harmless position)
        // (one careful step: partition parameters, codes of any length, the end of the frame -- cln_careful_code's)
        // Partitions end at multiples of their length in SAMPLES: the first (-order mod 16) codes go one by one, so that the turns
        // below start on multiples of 16 samples -- in every lane, whatever the orders of the subframes the lanes scan.
        {
            uint32_t pre = (0u - order) & 15u;
            pre = pre < left ? pre : left;
#pragma unroll 1
            for (; __any(pre != 0u); ) {
                if (pre != 0u) { (void)cln_careful_code(r, cur, per, rice2, KR); left -= 1u; pre = r.err ? 0u : pre - 1u; }
            }
            if (r.err) left = 0u;
        }
        bool ring_ok = false, even = false, fresh = false;
        uint32_t pumps = 0;
#pragma unroll 1
        while (__any(left >= 16u && !r.err)) {
            const bool has = left >= 16u && !r.err;          // this lane has sixteen codes to skip
            const bool live = has && ring_able;
            if (!ring_ok) { cln_reset(buf, g, row, (cur.p - 1u) >> 5, NC, false); ring_ok = true; even = false; fresh = true; }
            else {
                fresh = false;
                if (cln_pump_now(calm, even)) { cln_land(g, row, NC, false); cln_request(buf, g, cur.p, !calm || (pumps & 3u) == 0u); ++pumps; }
            }
            even = !even;
            int done = 0;
            if (__all(live)) done = cln_scan_turns<true>(row, g, cur, per, rice2, r.limit, true);        // (the usual case: no lane rides along)
            else if (__all(live || !has)) {
                const LCur keep = cur;                       // (a lane without sixteen codes rides along where it is: its ring stays consistent)
                done = cln_scan_turns<false>(row, g, cur, per, rice2, r.limit, live);
                if (!has) cur = keep;                        // (its tail is still to come: nothing of the ride may stick)
            }
            if (done > 0) { if (has) left -= 16u; continue; }
            // nothing but the ring having run dry: it is filled again at the top and the turn taken once more
            if (done == -2 && !fresh) { ring_ok = false; continue; }
            CLX_STAT(30, 1);
            // sixteen careful steps (rolled) for the lanes that have them
#pragma unroll 1
            for (int ii = 0; ii < 16; ++ii) {
                if (has && !r.err) { (void)cln_careful_code(r, cur, per, rice2, KR); left -= 1u; }
            }
            if (r.err) left = 0u;
            ring_ok = false;                                // the positions moved without the ring
        }
        // the tail (fewer than sixteen codes), one by one
#pragma unroll 1
        while (__any(left != 0u && !r.err)) {
            if (left != 0u && !r.err) { (void)cln_careful_code(r, cur, per, rice2, KR); left -= 1u; }
        }
        if (on && !r.err) r.pos = cur.p;
        parts_left = cur.parts;
        // partition parameters that belong to empty partitions at the very end (order == block size: subframe.rs:509, 706)
        if (on && !r.err) {
            while (!r.err && parts_left != 0u) { (void)clx_lread_rice_param(r, rice2); parts_left -= 1u; }
        }
        if (on) {
            if (r.err) clx_report_error(errkey, f, ch, r.err);
            else clx_poke_u32(&sf_start[fr.first_slot + ch + 1u], r.pos);      // (past the caches: a decode wave of the same kernel may be waiting for it, clx_k_pool)
        }
    }
    // ---- the frame's content class, with the last channel's header (the cursor stands in front of it)
    if (active && R.fkey != nullptr) {
        uint32_t key = CLX_COMPOSE_KEYS - 1u;            // (a frame that does not parse: the last class)
        if (!r.err) {
            LaneReader q = r;
            const SfHead h = clx_lparse_sf_header(q, clx_channel_bps(fr, nch));
            if (!q.err) {
                k_special = k_special || h.kind < 2u; k_omax = h.order > k_omax ? h.order : k_omax;
                const uint32_t oc = (k_omax <= 4u && !k_special) ? 0u : k_omax <= 8u ? 1u : k_omax <= 12u ? 2u : 3u;      // (clx_k_lean's builds)
                key = CLX_FKEY(k_special, oc, fr.channel_assignment);
            }
        }
        R.fkey[f] = key;
    }
    CLX_TL_END_SEQ(2, ((uint64_t)R.gen << 32) | (run_idx << 20) | bx);
    (void)run_idx;
}

extern "C" __global__ __launch_bounds__(64)
void clx_k_scan(const clx_runs runs, const clx_dev_frame* __restrict__ frames, const uint32_t* __restrict__ multi, uint32_t n_multi) {
    __shared__ struct { uint32_t ring[CLN_ROW][64]; } L;        // (the ring only: 7 KiB per wave -- they fit beside the decode waves)
    cln_scan_wave(&L.ring[0][0], runs.r[blockIdx.y], frames, multi, n_multi, blockIdx.x, blockIdx.y, (int)threadIdx.x);
}

// ---- C: waves composed by content ---------------------------------------------------------------------------------------------
// clx_k_compose: one workgroup per window (clx_plan_windows: up to 16 384 consecutive stereo frames of one block size), behind the
// scan.  A stable counting sort of the window's frames by content class (clx_k_scan's key): frame of rank r takes the physical
// slots s_lo + 2r, s_lo + 2r + 1 -- so a wave of 64 subframes holds one class (one predictor build, plain Rice turns, one stereo
// form) instead of a sample of all of them.  Only the run's slot maps change (clx_run::slot_frame / first_slot); sf_start, crc_part
// and the frames' places in `out` are indexed by what the PLAN says, as before.
extern "C" __global__ __launch_bounds__(CLX_COMPOSE_THREADS)
void clx_k_compose(const clx_runs runs, const clx_window* __restrict__ windows) {
    __shared__ uint8_t keys[CLX_COMPOSE_WINDOW];
    __shared__ uint16_t cnt[CLX_COMPOSE_KEYS][CLX_COMPOSE_THREADS];      // cnt[k][t]: frames of class k among thread t's, then the rank of its next one
    __shared__ uint16_t wsum[CLX_COMPOSE_THREADS / 64][CLX_COMPOSE_KEYS];       // per wave: its threads' frames of each class
    const clx_run& R = runs.r[blockIdx.y];
    const clx_window W = windows[blockIdx.x];
    uint32_t* const slot_frame = const_cast<uint32_t*>(R.slot_frame);
    uint32_t* const first_slot = const_cast<uint32_t*>(R.first_slot);
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, n = W.f_hi - W.f_lo;
    // (the classes, coalesced; eight loads in flight per thread: one after the other they would cost a memory round trip each)
    for (uint32_t i0 = t; i0 < n; i0 += 8u * CLX_COMPOSE_THREADS) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t q = 0; q < 8u; ++q) { const uint32_t i = i0 + CLX_COMPOSE_THREADS * q; v[q] = i < n ? R.fkey[W.f_lo + i] : 0u; }
#pragma unroll
        for (uint32_t q = 0; q < 8u; ++q) { const uint32_t i = i0 + CLX_COMPOSE_THREADS * q; if (i < n) keys[i] = (uint8_t)(v[q] & (CLX_COMPOSE_KEYS - 1u)); }
    }
#pragma unroll
    for (uint32_t k = 0; k < CLX_COMPOSE_KEYS; ++k) cnt[k][t] = 0;
    __syncthreads();
    // thread t owns the frames [lo, hi) of the window: a stable sort keeps them in this order inside every class
    const uint32_t chunk = (n + CLX_COMPOSE_THREADS - 1u) / CLX_COMPOSE_THREADS, lo = t * chunk < n ? t * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
    for (uint32_t i = lo; i < hi; ++i) cnt[keys[i]][t] += 1u;           // (its own column: no other thread touches it)
    // exclusive prefix over (class-major, thread-minor): inside the wave by shuffles, across the waves through LDS
    uint32_t pre[CLX_COMPOSE_KEYS];
#pragma unroll
    for (uint32_t k = 0; k < CLX_COMPOSE_KEYS; ++k) {
        const uint32_t c = cnt[k][t];
        uint32_t x = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, d, 64); if ((int)lane >= d) x += y; }
        pre[k] = x - c;
        if (lane == 63u) wsum[wave][k] = (uint16_t)x;
    }
    __syncthreads();
    uint32_t run = 0;                                     // frames of the classes before k
#pragma unroll
    for (uint32_t k = 0; k < CLX_COMPOSE_KEYS; ++k) {
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < CLX_COMPOSE_THREADS / 64u; ++w) { const uint32_t p = wsum[w][k]; before += w < wave ? p : 0u; total += p; }
        cnt[k][t] = (uint16_t)(pre[k] + run + before);
        run += total;
    }
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t k = keys[i];
        const uint32_t rank = cnt[k][t];
        cnt[k][t] = (uint16_t)(rank + 1u);
        const uint32_t s = W.s_lo + 2u * rank, f = W.f_lo + i;
        first_slot[f] = s;
        *reinterpret_cast<uint2*>(slot_frame + s) = make_uint2(f, f);       // (s is even: the window starts at its first frame's slot)
    }
}

// ---- the output side ------------------------------------------------------------------------------------------------------
// A turn leaves 64 bytes per row in the stage; TWO turns' tiles leave for HBM together, as whole 128-byte lines: eight adjacent
// lanes write one row's line, a store instruction covers eight rows, eight instructions the pair of tiles.  (Measured, 180 000 rows
// of 16 KiB written by 2 813 waves: 64 bytes x 16 rows per instruction 3.15 TB/s, 128 bytes x 8 rows 4.95 TB/s -- and the
// saturated kernel was within 20 % of the former's time: profiles/r03_ubench_storeshape.txt)
// Stage layout: tile t's row r sits at row position r ^ t, its 16-byte piece p at p ^ ((r >> 1) & 3): the lanes' writes (one row
// each) and the movers' reads (eight lanes = pieces 0-3 of tile 0 and of tile 1 of one row) both touch every bank group once.
__device__ __forceinline__ int4* cln_mine(int4* stage0, uint32_t t0, int lane) {        // this lane's row in the tile of sample index t0
    const uint32_t t = (t0 >> 4) & 1u;
    return stage0 + t * 256u + (((uint32_t)lane ^ t) * 4u);
}
// Where the wave's rows are: a wave-uniform base (the lowest row's address) and, per lane, its own row's distance from it in bytes
// (0xffffffff: the lane has no row).  A mover fetches the distance of the row it writes from the lane that owns it (ds_bpermute: no
// LDS storage), and the store takes base + distance as scalar base + 32-bit vector offset.  (A wave whose rows span 4 GiB or more
// is not taken: cln_kernel.)
struct LMover {
    uint64_t base;             // wave-uniform
    uint32_t rowoff;           // this lane's row (bytes from `base`), or CLN_NO_ROW
    bool all_real;             // wave-uniform: every row of the wave exists
    uint32_t pcm16;            // wave-uniform: interleaved 16-bit output (CLX_RUN_PCM16) -- 1: stereo frames, rowoff is the FRAME's place, the same in
                               // both lanes of its pair; 2: mono frames, every lane its own (round 6); 3: packed 24-bit output of stereo frames
                               // (CLX_RUN_PCM24, round 6; rowoff as for 1); 0: planar i32
    bool ms;                   // wave-uniform: the stage holds mid and side as decoded -- the movers turn them into left and right (cln_ms4)
};
#define CLN_NO_ROW 0xffffffffu
// Mid/side in the movers (round 6).  A lane that decodes a subframe holds ONE channel, so undoing mid/side there takes four
// instructions per sample in each lane of the pair (the partner's value comes through DPP, the rounding through a per-lane constant).
// A mover holds the same four samples of BOTH rows of a frame, and there
//     left = ((mid << 1 | side & 1) + side) >> 1 = mid + ((side + 1) >> 1)        right = left - side        (frame.rs:382-384)
// is four plain instructions per PAIR of samples.  So in waves in which every lane belongs to a mid/side pair without wasted bits the
// turns stage what they decoded and the movers do this: two instructions per sample less.  a: mid -> left, b: side -> right.
// (The short form is the reference's wrapping arithmetic while nothing wraps: mid and side inside [-2^29, 2^29).  The lean and wide turns' range
// checks are far stricter; the prologue and the slow turn, which stage whatever a stream decodes to, look at what they stage and give the
// group up beyond it -- a damaged or crafted stream: the general kernels' form of the stereo step is exact for every value.)
#define CLN_MS_RANGE (1 << 29)
// (gathered per lane in a register -- bit 31 of the sum: some sample had bit 30 or 31 set in s + 2^29, i.e. lay outside the range -- and voted on once
//  behind the loop.  As a bool carried through the prologue's loop, whose trip count the compiler takes for divergent, the HEAD of round 6 faulted on
//  the GPU -- every wave's rows written through a bad offset -- while the simulator agreed with the oracle; this form does not.)
__device__ __forceinline__ uint32_t cln_ms_wild(int32_t s) { const uint32_t t = (uint32_t)s + (uint32_t)CLN_MS_RANGE; return t | (t << 1); }
__device__ __forceinline__ void cln_ms4(int4& a, int4& b) {
    const int32_t tx = (b.x + 1) >> 1, ty = (b.y + 1) >> 1, tz = (b.z + 1) >> 1, tw = (b.w + 1) >> 1;
    a.x += tx; a.y += ty; a.z += tz; a.w += tw;
    b.x = a.x - b.x; b.y = a.y - b.y; b.z = a.z - b.z; b.w = a.w - b.w;
}
// Narrow output (CLX_OUT_PCM16, round 5): the wave's rows are the two channels of 32 stereo frames (lanes 2F, 2F + 1), and a pair of tiles
// holds 32 samples of each -- one 128-byte line of interleaved 16-bit PCM per frame.  Eight adjacent lanes write one frame's line
// (lane q: sample pairs 4q .. 4q + 3, i.e. piece q & 3 of tile q >> 2 of BOTH rows, packed low halves left | right), a store
// instruction covers eight frames, FOUR instructions the pair of tiles (planar i32: eight).  n_tiles: 2, or 1 for a lone last tile.
__device__ __forceinline__ void cln_store_pcm16(const int4* stage0, const LMover& M, uint32_t t0, int lane, uint32_t n_tiles) {
    clx_wave_sync();
    // (the lane's places in the stage are worked out HERE, every time: left to itself the compiler keeps eight addresses per lane alive
    //  across the whole decode loop for this secondary mode -- and spills them, 168 registers being what they are)
    uint32_t ln = (uint32_t)lane;
    CLX_OPAQUE(ln);
    const uint32_t q = ln & 7u, t = q >> 2, p = q & 3u;
#pragma unroll
    for (uint32_t i = 0; i < 4u; ++i) {
        const uint32_t F = 8u * i + (ln >> 3);                                // the frame (row pair) whose line this lane helps to write
        const uint32_t r0 = 2u * F, r1 = 2u * F + 1u;
        int4 a = stage0[t * 256u + ((r0 ^ t) * 4u) + (p ^ ((r0 >> 1) & 3u))];
        int4 b = stage0[t * 256u + ((r1 ^ t) * 4u) + (p ^ ((r1 >> 1) & 3u))];
        const uint32_t o = (uint32_t)__shfl((int)M.rowoff, (int)r0, 64);
        if (M.ms) cln_ms4(a, b);
        int4 w;                                                              // (left: low half, right: high half of each dword)
        w.x = (int32_t)clx_perm((uint32_t)b.x, (uint32_t)a.x, 0x05040100u); w.y = (int32_t)clx_perm((uint32_t)b.y, (uint32_t)a.y, 0x05040100u);
        w.z = (int32_t)clx_perm((uint32_t)b.z, (uint32_t)a.z, 0x05040100u); w.w = (int32_t)clx_perm((uint32_t)b.w, (uint32_t)a.w, 0x05040100u);
        if (o != CLN_NO_ROW && t < n_tiles) clx_store1x16_s(M.base, o + 4u * t0 + 16u * q, w);      // (a frame's sample t0 sits 4 t0 bytes into its block)
    }
    clx_wave_sync();
}
// The same for MONO frames (round 6): a pair of tiles holds 32 samples of every row -- 64 bytes of 16-bit PCM.  Four adjacent lanes write
// one row's 64 bytes (lane q: samples 8q .. 8q + 7, i.e. pieces 2 (q & 1) and 2 (q & 1) + 1 of tile q >> 1, low halves packed), a store
// instruction covers sixteen rows, four instructions the pair of tiles.
__device__ __forceinline__ void cln_store_pcm16_mono(const int4* stage0, const LMover& M, uint32_t t0, int lane, uint32_t n_tiles) {
    clx_wave_sync();
    uint32_t ln = (uint32_t)lane;
    CLX_OPAQUE(ln);                                                          // (as above: nothing of this is kept across the decode loop)
    const uint32_t q = ln & 3u, t = q >> 1, p = 2u * (q & 1u);
#pragma unroll
    for (uint32_t i = 0; i < 4u; ++i) {
        const uint32_t r = 16u * i + (ln >> 2);                              // the row whose 64 bytes this lane helps to write
        const int4 a = stage0[t * 256u + ((r ^ t) * 4u) + (p ^ ((r >> 1) & 3u))];
        const int4 b = stage0[t * 256u + ((r ^ t) * 4u) + ((p + 1u) ^ ((r >> 1) & 3u))];
        const uint32_t o = (uint32_t)__shfl((int)M.rowoff, (int)r, 64);
        int4 w;                                                              // (eight samples' low halves, in order)
        w.x = (int32_t)clx_perm((uint32_t)a.y, (uint32_t)a.x, 0x05040100u); w.y = (int32_t)clx_perm((uint32_t)a.w, (uint32_t)a.z, 0x05040100u);
        w.z = (int32_t)clx_perm((uint32_t)b.y, (uint32_t)b.x, 0x05040100u); w.w = (int32_t)clx_perm((uint32_t)b.w, (uint32_t)b.z, 0x05040100u);
        if (o != CLN_NO_ROW && t < n_tiles) clx_store1x16_s(M.base, o + 2u * t0 + 16u * q, w);      // (a mono frame's sample t0 sits 2 t0 bytes into its block)
    }
    clx_wave_sync();
}
// Packed 24-bit output (CLX_OUT_PCM24, round 6): a stereo frame's 32 sample pairs of a pair of tiles are 192 bytes -- L0 R0 L1 R1 ..., three
// bytes each, little-endian -- twelve 16-byte pieces.  Sixteen adjacent lanes take one frame (twelve of them a piece each), a store
// instruction covers four frames, eight instructions the pair of tiles.  A piece is four dwords; dword j of the block (j = 3m + k) holds
// bytes of two consecutive samples of the interleaved sequence S[i] = (i even ? left : right)[i / 2], i = 4m + k:
//     k = 0: S[i] bytes 0-2, S[i+1] byte 0     k = 1: S[i] bytes 1-2, S[i+1] bytes 0-1     k = 2: S[i] byte 2, S[i+1] bytes 0-2
// -- one v_perm_b32 each.  Where S[i] sits in the stage depends on the lane (its piece) and, through a constant, on the frame: the
// eight addresses are worked out per call (kept across the decode loop they would cost the split tier eight registers it does not
// have) and the four frames-of-four steps differ by an immediate offset.  n_tiles: 2, or 1 for a lone last tile (six pieces).
__device__ __forceinline__ void cln_store_pcm24(const int4* stage0, const LMover& M, uint32_t t0, int lane, uint32_t n_tiles) {
    clx_wave_sync();
    uint32_t ln = (uint32_t)lane;
    CLX_OPAQUE(ln);
    const uint32_t c = ln & 15u, fsub = ln >> 4;                            // the piece (0 .. 11; 12 .. 15: idle) and the frame among four
    const int32_t* const st32 = reinterpret_cast<const int32_t*>(stage0);
    uint32_t at[4][2], sel[4];                                              // where S[i], S[i+1] of the piece's four dwords sit for frame `fsub` (in int32)
#pragma unroll
    for (uint32_t d = 0; d < 4u; ++d) {
        const uint32_t j = 4u * c + d, m = (j * 43u) >> 7, k = j - 3u * m;  // (j / 3 for j < 48)
        sel[d] = k == 0u ? 0x04020100u : k == 1u ? 0x05040201u : 0x06050402u;
#pragma unroll
        for (uint32_t h = 0; h < 2u; ++h) {
            const uint32_t i = 4u * m + k + h, par = i & 1u, sm = (i >> 1) & 31u, t = sm >> 4, p = (sm >> 2) & 3u, e = sm & 3u;
            at[d][h] = 4u * (t * 256u + ((2u * fsub + (par ^ t)) * 4u) + (p ^ (fsub & 3u))) + e;      // (frame F = 4 it + fsub: row 2F + par, + 128 int32 per `it`)
        }
    }
#pragma unroll
    for (uint32_t it = 0; it < 8u; ++it) {
        const uint32_t o = (uint32_t)__shfl((int)M.rowoff, (int)(2u * (4u * it + fsub)), 64);
        int4 w;
        w.x = (int32_t)clx_perm((uint32_t)st32[at[0][1] + 128u * it], (uint32_t)st32[at[0][0] + 128u * it], sel[0]);
        w.y = (int32_t)clx_perm((uint32_t)st32[at[1][1] + 128u * it], (uint32_t)st32[at[1][0] + 128u * it], sel[1]);
        w.z = (int32_t)clx_perm((uint32_t)st32[at[2][1] + 128u * it], (uint32_t)st32[at[2][0] + 128u * it], sel[2]);
        w.w = (int32_t)clx_perm((uint32_t)st32[at[3][1] + 128u * it], (uint32_t)st32[at[3][0] + 128u * it], sel[3]);
        if (o != CLN_NO_ROW && c < 6u * n_tiles) clx_store1x16_s(M.base, o + 6u * t0 + 16u * c, w);      // (a frame's sample t0 sits 6 t0 bytes into its block)
    }
    clx_wave_sync();
}
// the pair of tiles that starts at sample index t0 (a multiple of 32)
// (P24: the kernel writes packed 24-bit output too -- clx_k_lean24 alone: in a batch with that output it takes the 16-bit frames as well,
//  and clx_k_lean's register file has no room for a fourth form of the store)
template <bool P24>
__device__ __forceinline__ void cln_store_pair(const int4* stage0, const LMover& M, uint32_t t0, int lane) {
    if (P24 && M.pcm16 == 3u) { cln_store_pcm24(stage0, M, t0, lane, 2u); return; }
    if (M.pcm16 == 2u) { cln_store_pcm16_mono(stage0, M, t0, lane, 2u); return; }
    if (M.pcm16) { cln_store_pcm16(stage0, M, t0, lane, 2u); return; }
    if (M.ms) {
        // both rows of a frame per lane (cln_ms4): lane 8 fh + q takes piece q & 3 of tile q >> 2 of frames fh, fh + 8, fh + 16, fh + 24 --
        // as many reads and stores as below, four more instructions per pair of samples
        clx_wave_sync();
        uint32_t ln = (uint32_t)lane;
        CLX_OPAQUE(ln);                                   // (worked out per call, as in cln_store_pcm16: nothing of this is kept across the decode loop)
        const uint32_t fh = ln >> 3, q = ln & 7u, t = q >> 2;
        const uint32_t ia = t * 256u + (((2u * fh) ^ t) * 4u) + ((q & 3u) ^ (fh & 3u));     // row position 2 fh ^ t; + 64 int4 per eight frames
        const int4* const sa = stage0 + ia;
        const int4* const sb = stage0 + (ia ^ 4u);                                            // row position (2 fh + 1) ^ t
        const uint32_t toff = 4u * t0 + 16u * q, pa = 8u * fh;                               // (pa: lane 2 fh's place for ds_bpermute)
        if (M.all_real) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                int4 a0 = sa[128 * half], b0 = sb[128 * half], a1 = sa[128 * half + 64], b1 = sb[128 * half + 64];
                uint32_t o[4];
                if (half == 0) clx_bperm4<0, 4, 64, 68>(pa, M.rowoff, o);
                else           clx_bperm4<128, 132, 192, 196>(pa, M.rowoff, o);
                cln_ms4(a0, b0); cln_ms4(a1, b1);
                clx_store4x16_s(M.base, o[0] + toff, o[1] + toff, o[2] + toff, o[3] + toff, a0, b0, a1, b1);
            }
        } else {
            uint32_t o[4][2];
            clx_bperm2<0, 4>(pa, M.rowoff, o[0]); clx_bperm2<64, 68>(pa, M.rowoff, o[1]); clx_bperm2<128, 132>(pa, M.rowoff, o[2]); clx_bperm2<192, 196>(pa, M.rowoff, o[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int4 a = sa[64 * i], b = sb[64 * i];
                cln_ms4(a, b);
                if (o[i][0] != CLN_NO_ROW) clx_store1x16_s(M.base, o[i][0] + toff, a);       // (rows that do not exist are not written)
                if (o[i][1] != CLN_NO_ROW) clx_store1x16_s(M.base, o[i][1] + toff, b);
            }
        }
        clx_wave_sync();
        return;
    }
    clx_wave_sync();
    uint32_t ln = (uint32_t)lane;
    CLX_OPAQUE(ln);                                       // (worked out per call, as in cln_store_pcm16: the decode loop keeps the places of ONE planar mover)
    const uint32_t h = ln >> 3, q = ln & 7u, t = q >> 2;
    const int4* const src = stage0 + t * 256u + ((h ^ t) * 4u) + ((q & 3u) ^ ((h >> 1) & 3u));      // + 32 int4 per instruction (8 rows)
    const uint32_t toff = 4u * t0 + 16u * q;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        int4 w[4];
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = src[32 * (4 * half + j)];
        if (half == 0) clx_bperm4<0, 32, 64, 96>(4u * h, M.rowoff, o);                      // (rows h + 8 k: lanes' places 4 h + 32 k)
        else           clx_bperm4<128, 160, 192, 224>(4u * h, M.rowoff, o);
#ifndef CLN_NO_STORES
        if (M.all_real) clx_store4x16_s(M.base, o[0] + toff, o[1] + toff, o[2] + toff, o[3] + toff, w[0], w[1], w[2], w[3]);
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (o[j] != CLN_NO_ROW) clx_store1x16_s(M.base, o[j] + toff, w[j]);      // (rows that do not exist are not written)
        }
#else       // (measurement: what the write path costs -- everything but the store instructions; wrong output)
        asm volatile("s_nop 1" :: "v"(o[0] + toff), "v"(o[1] + toff), "v"(o[2] + toff), "v"(o[3] + toff), "v"(w[0].x), "v"(w[1].x), "v"(w[2].x), "v"(w[3].x) : "memory");
#endif
    }
    clx_wave_sync();
}
// a lone tile 0 (the block's last 16 samples when the block size is an odd multiple of 16): 64 bytes x 16 rows per instruction
template <bool P24>
__device__ __forceinline__ void cln_store_single(const int4* stage0, const LMover& M, uint32_t t0, int lane) {
    if (P24 && M.pcm16 == 3u) { cln_store_pcm24(stage0, M, t0, lane, 1u); return; }
    if (M.pcm16 == 2u) { cln_store_pcm16_mono(stage0, M, t0, lane, 1u); return; }
    if (M.pcm16) { cln_store_pcm16(stage0, M, t0, lane, 1u); return; }
    if (M.ms) {                                           // (cln_ms4: lane 4 fh + q takes piece q of frames fh and fh + 16)
        clx_wave_sync();
        const uint32_t fh = (uint32_t)lane >> 2, q = (uint32_t)lane & 3u;
#pragma unroll
        for (uint32_t i = 0; i < 2u; ++i) {
            const uint32_t r0 = 2u * (fh + 16u * i), r1 = r0 + 1u, sz = q ^ (fh & 3u);
            int4 a = stage0[r0 * 4u + sz], b = stage0[r1 * 4u + sz];
            const uint32_t oa = (uint32_t)__shfl((int)M.rowoff, (int)r0, 64), ob = (uint32_t)__shfl((int)M.rowoff, (int)r1, 64);
            cln_ms4(a, b);
            if (oa != CLN_NO_ROW) clx_store1x16_s(M.base, oa + 4u * t0 + 16u * q, a);
            if (ob != CLN_NO_ROW) clx_store1x16_s(M.base, ob + 4u * t0 + 16u * q, b);
        }
        clx_wave_sync();
        return;
    }
    clx_wave_sync();
    const uint32_t h = (uint32_t)lane >> 2, q = (uint32_t)lane & 3u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t r = 16u * (uint32_t)k + h;
        const int4 w = stage0[r * 4u + (q ^ ((r >> 1) & 3u))];
        const uint32_t o = (uint32_t)__shfl((int)M.rowoff, (int)r, 64);
        if (o != CLN_NO_ROW) clx_store1x16_s(M.base, o + 4u * t0 + 16u * q, w);
    }
    clx_wave_sync();
}
// A finished pair leaves at the START of the turn after it (and what is left when the kernel ends): the stores then sit right in
// front of the ring's landing, whose wait (s_waitcnt vmcnt(0): the compiler cannot count across the branches) paces them -- see
// cln_land.  n: tiles in the stage that have not left (0, 1: tile 0 of a pair, 2), t0: the first one's sample index.
struct LTile { uint32_t n; uint32_t t0; };
__device__ __forceinline__ void cln_done(LTile& T, uint32_t t0) {                 // the tile of sample index t0 is in the stage
    if ((t0 & 16u) == 0u) { T.n = 1u; T.t0 = t0; } else T.n = 2u;
}
template <bool P24>
__device__ __forceinline__ void cln_flush(LTile& T, const int4* stage0, const LMover& M, int lane) {
    if (T.n == 2u) { cln_store_pair<P24>(stage0, M, T.t0, lane); T.n = 0u; }          // (wave-uniform)
}
template <bool P24>
__device__ __forceinline__ void cln_flush_all(LTile& T, const int4* stage0, const LMover& M, int lane) {
    if (T.n == 2u) cln_store_pair<P24>(stage0, M, T.t0, lane);
    else if (T.n == 1u) cln_store_single<P24>(stage0, M, T.t0, lane);
    T.n = 0u;
}

// j-th coefficient (applies to s[i-1-j]) out of the packed form: C[q] = (c[2q] << 16) | (c[2q+1] & 0xffff)
template <int NP>
__device__ __forceinline__ int32_t cln_coef(const uint32_t (&C)[NP], int j) {
    return (j & 1) ? (int32_t)(int16_t)(C[j >> 1] & 0xffffu) : ((int32_t)C[j >> 1] >> 16);
}

// stereo decorrelation of four of the turn's samples (the four `b` of the turn: clx_lfinish, clx_lanes.hip) into the stage.  Four at a
// time, right behind the four's predictor steps: the outputs do not stay in registers until the end of the turn (round 5: twelve
// registers less across the turn -- clx_k_lean24 and clx_k_lean sit at the edge of their register files).  (The stage is written
// before the turn's vote: a turn that is taken again writes the same places again.)
__device__ __forceinline__ void cln_finish4(const int32_t (&s0)[4], const Finish& F, int4* mine, uint32_t sw, uint32_t b) {
    int32_t m[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = s0[i];
    if (F.any_wasted) {                                  // wasted-bits shift (subframe.rs:216-225): a wave with such a lane only
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = (int32_t)((uint32_t)m[i] << F.wasted);
    }
    int32_t y[4];
    if (F.all_ms) clx_ms_short4(m, y, F.sgn, 1u + (F.sgn & 1u));                       // (exact below 2^29: part of the turn's range check)
    else if (F.any_decor) clx_decor4(m, y, F.dsg, F.drm, F.dc, F.s1, F.pmask);        // (exact below 2^29: part of the turn's range check)
    else {
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = m[i];
    }
    mine[b ^ sw] = make_int4(y[0], y[1], y[2], y[3]);
}

// the 16-bit tier: every four of a turn goes into the stage as decoded, right behind its predictor steps (nothing of the turn's output stays in
// registers); a wave with stereo forms other than plain mid/side pairs, or with wasted bits, takes the tile out of the stage again at the END of
// the turn, eight samples per statement (clx_decor8_mad: any mix of forms, the shift included, four instructions per sample -- exact while a
// sample and its shifted value fit 24 bits: the turn's range check, whose limit cln_run lowers by the lane's wasted bits), and puts it back.  Waves
// of plain mid/side pairs have no stereo here at all: their movers undo it (cln_ms4).  The turn stays one basic block up to this wave-uniform
// choice, and the asm statements of the stereo form stay out of the compiler's way while it schedules the Rice and predictor work.
__device__ __forceinline__ void cln_finish16(const Finish& F, int4* mine, uint32_t sw) {
    if (F.any_decor || F.any_wasted) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int4 u = mine[(uint32_t)(2 * h) ^ sw], v = mine[(uint32_t)(2 * h + 1) ^ sw];
            const int32_t m[8] = { u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w };
            int32_t y[8];
            clx_decor8_mad(m, y, F.mo, F.mt, F.mc);
            mine[(uint32_t)(2 * h) ^ sw] = make_int4(y[0], y[1], y[2], y[3]);
            mine[(uint32_t)(2 * h + 1) ^ sw] = make_int4(y[4], y[5], y[6], y[7]);
        }
    }
}

// ---- the lean turn -------------------------------------------------------------------------------------------------------
// H on entry: packed pairs, H[j] = (lo: s[t0-2-j], hi: s[t0-1-j]) for j = 0 .. 2NP-2.  Returns the wave's vote; on success the
// cursor and H are advanced by sixteen samples (live lanes) and the tile is in the stage.
// EDGE: some lane's partition ends inside this turn (a parameter may have to be read in front of a four).  Without it the turn is
// ONE basic block: the compiler then overlaps the LDS round trip of a four's window with the predictor work of the four before.
// WIDE: the same turn for waves in which a lane's signal is outside the 16-bit range (a loud side channel): H holds i32 samples
// (H[j] = s[t0-1-j]) and the predictor is a chain of v_mad_i32_i24 on the unpacked coefficients CW -- exact while every history
// sample lies in [-lim, lim) with lim <= 2^23 and sum|c| * lim < 2^31 (clx_ltransition's S.lim; checked on the data like the packed
// form's).  Twice the predictor instructions of the packed form, everything else the same.
// FORM 2 (SPLIT, clx_k_lean24): the same turn for audio of more than 16 bits and / or more than 12 taps.  Every sample is kept as
// two 16-bit pieces, s = hi * 4096 + lo with lo = s & 0xfff and hi = s >> 12, each in its own packed history (H = NH pairs of lo
// pieces, then NH pairs of hi pieces); the predictor is two v_dot2_i32_i16 chains and
//     (sum c*s) >> shift  =  ((A_hi << e1) + (A_lo >> e2)) >> e3        e1 = max(12 - shift, 0), e2 = min(shift, 12), e3 = max(shift - 12, 0)
// -- the reference's i64 evaluation (subframe.rs:586-614) exactly, while sum|c| < 2^19 (any 15-bit coefficients of <= 32 taps) and
// every history sample lies in [-lim, lim) with lim <= (floor((2^31 - 1) / sum|c|) - 1) * 4096: then neither chain wraps, A_hi * 4096
// is a multiple of 2^e2, and A_hi + (A_lo >> 12) stays inside 32 bits.  No 64-bit instruction per sample.
template <int NP, int MODE, bool EDGE, int FORM, int HN>
__device__ __forceinline__ int cln_lean_turn(const uint32_t* row, const LRing& g, LCur& cur, uint32_t (&H)[HN], const uint32_t (&C)[NP],
                                              uint32_t shift, uint32_t e1, uint32_t e3, int32_t lim, uint32_t per, uint32_t rice2,
                                              uint32_t limit, bool live, const LKind& K, const Finish& F, int4* mine, uint32_t sw) {
    constexpr bool WIDE = FORM == 1, SPLIT = FORM == 2;
    constexpr int NH = 2 * NP - 1;                    // pairs carried from turn to turn
    static_assert(HN == (SPLIT ? 2 * NH : 2 * NP), "history registers");
    uint32_t P[NH + 16];                              // P[NH + m] = pair that ends at sample m of the turn (m = -NH .. 15); SPLIT: of lo pieces
    uint32_t PH[SPLIT ? NH + 16 : 1];                 // SPLIT: the same of hi pieces
    int32_t hw[2 * NP + 16];                          // WIDE: hw[2NP - 1 - j + i] = s[i - 1 - j]: the samples in time order, the turn's own appended
    int32_t cw[WIDE ? 2 * NP : 1];                    // WIDE: the coefficients unpacked -- per turn: kept across the decode loop they cost 2 NP registers
    if (WIDE) {                                       // that the packed turns, the usual ones, have no use for (round 6)
#pragma unroll
        for (int j = 0; j < 2 * NP; ++j) cw[WIDE ? j : 0] = cln_coef<NP>(C, j);
    }
    if (!WIDE) {
#pragma unroll
        for (int j = 0; j < NH; ++j) P[NH - 1 - j] = H[j];
        if (SPLIT) {
#pragma unroll
            for (int j = 0; j < NH; ++j) PH[NH - 1 - j] = H[(SPLIT ? NH : 0) + j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 2 * NP; ++j) hw[2 * NP - 1 - j] = (int32_t)H[j];
    }
    LCur c = cur;
    if (MODE == 0 ? !live : !(live && K.rice)) c.pcnt = 0x7fffff00u;      // (lanes without Rice codes never meet a partition edge)
    uint32_t c1 = 31u - c.k;
    CLX_OPAQUE(c1);
    bool bad = false;
    int32_t msh = 0;                                  // the smallest "bits left of the window after the code" -- negative: a code > 32 bits
    uint32_t zmax = 0;                                // MODE 0: the longest run of zeros instead (v_ffbh as it comes: all ones for an empty register)
    int32_t hi = -0x7fffffff - 1, lo = 0x7fffffff;
    uint32_t pw = c.p;
    const uint32_t pb = 4u + rice2, esc = rice2 ? 31u : 15u;
    // The four's Rice codes are decoded first (that fixes where the next four starts), the next four's window is requested from the
    // ring, and only then does the predictor run over the four samples: the LDS round trip hides behind it.
    uint32_t wa, wb, wc, wd;
    // a partition that starts exactly at a four: its parameter comes first (subframe.rs:314-319 / 362-367)
    auto edge = [&]() {
        if (EDGE) {
            const bool at = c.pcnt == 0u;
            if (clx_any(at)) {
                const uint32_t pv = cln_peek32(row, g, c.p);
                if (at) {
                    c.k = pv >> (32u - pb);
                    bad = bad || c.k == esc || c.parts == 0u || c.next == 0u;
                    c.p += pb; c.parts -= 1u; c.pcnt = c.next; c.next = per;
                    c1 = 31u - c.k;
                }
                CLX_OPAQUE(c1);
            }
            bad = bad || c.pcnt < 4u;                 // (a partition edge inside the four codes: the slow turn's)
        }
        c.pcnt -= 4u;
    };
    // register window: 128 bits from bit c.p on
    auto window = [&]() {
        pw = c.p;
        const uint32_t s = cln_slot(g, (c.p - 1u) >> 5);
        const uint32_t w0 = CLN_AT(row, s), w1 = CLN_AT(row, s + 1u), w2 = CLN_AT(row, s + 2u), w3 = CLN_AT(row, s + 3u), w4 = CLN_AT(row, s + 4u);
        const uint32_t sh = 0u - c.p;                 // v_alignbit takes the low five bits: (32 - p % 32) % 32
        wa = clx_alignbit(w0, w1, sh); wb = clx_alignbit(w1, w2, sh); wc = clx_alignbit(w2, w3, sh); wd = clx_alignbit(w3, w4, sh);
    };
    edge(); window();
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint32_t kk = c.k & 31u, k4 = c.k, c14 = c1;
        uint32_t shsum = 0;
        uint32_t X[4];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            // one Rice code (subframe.rs:337-341): z zeros, a one, k remainder bits = 32 - sh bits
            // __clz is v_ffbh + v_min (32 for a register of zeros), and the smallest "bits left behind the code" of the turn said whether
            // a code was longer than its window: v_ffbh, v_min, half a v_min3 per code.  Round 5: where every lane has Rice codes (MODE 0)
            // the count is taken as the instruction gives it -- all ones for a register of zeros -- and the longest count is compared
            // with what the window holds (unsigned: all ones is "too long" like 32 is), once per turn, or per four where the parameter may
            // change: v_ffbh and half a v_max3 per code.  (Round 4 had the same idea with an asm volatile statement, which the compiler
            // schedules around blindly: +15 %.  A plain asm statement is scheduled like any other instruction.)
            const uint32_t z = MODE == 0 ? clx_ffbh(wa) : (uint32_t)__clz((int)wa);
            int32_t sh = (int32_t)(c14 - z);
            const uint32_t u = (z << kk) | clx_bfe(wa, (uint32_t)sh, k4);
            uint32_t xr = (u >> 1) ^ (0u - (u & 1u));                      // rice_to_signed (subframe.rs:157-170)
            if (MODE == 0) zmax = z > zmax ? z : zmax;
            else {
                // constants and verbatim fields (subframe.rs:382-415) ride along under masks -- no branch, no select on a condition
                const int32_t shr = (int32_t)((uint32_t)sh & K.ricemask);   // (only a Rice code can be too long)
                msh = shr < msh ? shr : msh;
                const uint32_t xv = (uint32_t)((int32_t)wa >> K.vsh);
                xr = ((xr & K.ricemask) | (xv & K.verbmask)) | K.cor;
                sh = (int32_t)((uint32_t)shr | K.vshm);
            }
            shsum += (uint32_t)sh;
            wa = clx_alignbit(wa, wb, (uint32_t)sh); wb = clx_alignbit(wb, wc, (uint32_t)sh); wc = clx_alignbit(wc, wd, (uint32_t)sh); wd = clx_alignbit(wd, 0u, (uint32_t)sh);
            X[ii] = xr;
        }
        if (MODE == 0) {
            if (EDGE) { bad = bad || zmax > c14; zmax = 0; }      // (the parameter may change with the next four)
        }
        else CLX_OPAQUE(msh);                                     // (folded per four: no shift count of the block stays live for the vote)
        c.p += MODE == 0 ? 128u - shsum : ((128u - shsum) & K.bitmask);
        if (b < 3) { edge(); window(); }
        CLN_SCHED_FENCE();
        int32_t S4[4];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = 4 * b + ii;
            const uint32_t xr = X[ii];
            // predictor, oldest tap first: only the last term depends on the sample before
            int32_t acc = 0;
            int32_t pred;
            if (SPLIT) {
                // (the oldest pair's term starts each chain: the VOP3P form with a constant 0 addend -- no v_mov per chain)
                acc = clx_sdot2_first(C[NP - 1], P[NH + i - 1 - 2 * (NP - 1)]);
                int32_t ah = clx_sdot2_first(C[NP - 1], PH[(SPLIT ? NH + i - 1 - 2 * (NP - 1) : 0)]);
#pragma unroll
                for (int q = NP - 2; q >= 0; --q) {
                    acc = clx_sdot2(C[q], P[NH + i - 1 - 2 * q], acc);
                    ah = clx_sdot2(C[q], PH[(SPLIT ? NH + i - 1 - 2 * q : 0)], ah);
                }
                pred = (int32_t)(((uint32_t)ah << e1) + (uint32_t)(acc >> shift)) >> e3;
            } else if (!WIDE) {
                acc = clx_sdot2_first(C[NP - 1], P[NH + i - 1 - 2 * (NP - 1)]);
#pragma unroll
                for (int q = NP - 2; q >= 0; --q) acc = clx_sdot2(C[q], P[NH + i - 1 - 2 * q], acc);
                pred = acc >> shift;
            } else {
#pragma unroll
                for (int j = 2 * NP - 1; j >= 0; --j) acc = __mul24(cw[j], hw[2 * NP - 1 - j + i]) + acc;      // c[j] * s[i-1-j]: v_mad_i32_i24
                pred = acc >> shift;
            }
            const int32_t s = (int32_t)(xr + (uint32_t)pred);                                       // + prediction (wrapping)
            if (SPLIT) {
                P[NH + i] = clx_perm((uint32_t)s & 0xfffu, P[NH + i - 1], 0x05040302u);             // (lo: the sample before's piece, hi: this one's)
                PH[(SPLIT ? NH + i : 0)] = clx_perm((uint32_t)(s >> 12), PH[(SPLIT ? NH + i - 1 : 0)], 0x05040302u);
            }
            else if (!WIDE) P[NH + i] = clx_perm((uint32_t)s, P[NH + i - 1], 0x05040302u);          // (lo: the sample before, hi: this one)
            else hw[2 * NP + i] = s;
            hi = s > hi ? s : hi; lo = s < lo ? s : lo;
            S4[ii] = s;
        }
        CLX_OPAQUE(hi); CLX_OPAQUE(lo);
        // the stage, four by four: the split tier with its stereo form, the 16-bit tier as decoded (cln_finish16)
        if (SPLIT) cln_finish4(S4, F, mine, sw, (uint32_t)b);
        else mine[(uint32_t)b ^ sw] = make_int4(S4[0], S4[1], S4[2], S4[3]);
    }
    if constexpr (!SPLIT) cln_finish16(F, mine, sw);
    // what was decoded is what the stream holds iff no code was longer than its window register, the last window lay inside the
    // ring's filled part and nothing reached past the end of the frame; the predictor was exact iff the outputs (the next turn's
    // history) stayed inside the range
    if (MODE == 0 && !EDGE) msh = zmax > c1 ? -1 : 0;          // (one parameter for the whole turn)
    const bool covered = (((pw - 1u) >> 5) + 5u <= g.fill && c.p <= limit) || (MODE != 0 && K.bitmask == 0u);
    const bool ok = !live || (!bad && msh >= 0 && covered && hi < lim && lo >= -lim);
    const bool all = __all(ok);
    if (!all) { CLX_STAT(53, live && bad); CLX_STAT(54, live && msh < 0); CLX_STAT(55, live && !covered); CLX_STAT(56, live && !(hi < lim && lo >= -lim)); }
    if (all) {
        // every lane takes the turn's end state -- no select per register: what a lane that decodes nothing computed is never
        // looked at; only its position must stay where its ring is
        const uint32_t p_in = cur.p;
        cur = c;
        if (!live) cur.p = p_in;
        if (!WIDE) {
#pragma unroll
            for (int j = 0; j < NH; ++j) H[j] = P[NH + 15 - j];
            if (SPLIT) {
#pragma unroll
                for (int j = 0; j < NH; ++j) H[(SPLIT ? NH : 0) + j] = PH[(SPLIT ? NH + 15 - j : 0)];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2 * NP; ++j) H[j] = (uint32_t)hw[2 * NP + 15 - j];
        }
    }
    if (all) return 1;
    // why not: 0 a rare case the slow turn handles (a partition edge inside a four, an escape code, a code longer than 32 bits, the
    // end of the frame), -1 some lane's signal left the range of this evaluation, -2 nothing but the ring having run dry
    if (__any(live && (bad || msh < 0 || c.p > limit))) return 0;
    if (__any(live && !(hi < lim && lo >= -lim))) return -1;
    return -2;
}

// Returns false when the wave gives the group up: the slow turn costs about five lean turns, so a wave that keeps needing it
// (a lane whose signal stays outside the 16-bit range: loud side channels) is better off in clx_k_lanes, whose 24-bit tier takes
// such lanes at full speed.  The group is then decoded again from its start by that kernel (rows are simply rewritten; nothing
// has been reported for it yet).
#ifndef CLN_SLOW_BUDGET
#define CLN_SLOW_BUDGET 6u
#endif
template <int NP>
__device__ __forceinline__ bool cln_body(const clx_buf& buf, LaneReader& r, LRing& g, uint32_t* row, int4* stage, LCur& cur, uint32_t (&H)[2 * NP],
                                         const uint32_t (&C)[NP], uint32_t order, uint32_t shift, int32_t lim, int32_t lim24,
                                         uint32_t per, uint32_t rice2,
                                         uint32_t n, uint32_t i0, uint32_t nmax, const LKind& K, int mode, const Finish& F, const LMover& M, LTile& T, int lane,
                                         LCrc& CR, bool crc, bool calm) {

    const uint32_t sw = ((uint32_t)lane >> 1) & 3u;      // (eight neighbouring lanes' 16-byte stage stores then cover all 32 banks)
    bool slow = true;                                    // H holds i32 samples (the prologue leaves them so)
    bool ring_ok = false;
    uint32_t nslow = 0;
    for (uint32_t t0 = i0; t0 < nmax; t0 += 16u) {
        const bool live = n != 0u && !r.err;
#ifndef CLN_LAND_FIRST
        cln_flush<false>(T, stage, M, lane);             // the pair of tiles before, once it is complete
#endif
        int4* const mine = cln_mine(stage, t0, lane);
        if (slow) {
            // back to the lean turns as soon as every live lane's history fits the packed form
            bool in = true;
#pragma unroll
            for (int j = 0; j < 2 * NP; ++j) in = in && (int32_t)H[j] < lim && (int32_t)H[j] >= -lim;
            if (__all(in || !live || order == 0u)) {
#pragma unroll
                for (int j = 0; j < 2 * NP - 1; ++j) H[j] = clx_perm(H[j], H[j + 1], 0x05040100u);      // (lo: s[-2-j], hi: s[-1-j])
                slow = false;
            }
        }
        if (!ring_ok) { cln_reset(buf, g, row, (cur.p - 1u) >> 5, CR, crc); ring_ok = true; }
        else if (cln_pump_now(calm, (t0 & 16u) == 0u)) {      // (the turns at whose start a pair of tiles leaves: the stores stay right in front of the landing's wait)
            // (calm: a second and a third granule at every fourth pump -- the pumps are at the even turns.  The 12-tap build has two granules in
            //  flight at most -- with a third the compiler spills it across the turn, 168 registers being what they are -- and asks for the second
            //  at every other pump: 1.5 granules per pump either way, what a calm lane uses at most)
            constexpr int MAXP = NP == 6 ? 2 : 3;
            cln_land<MAXP>(g, row, CR, crc); cln_request<MAXP>(buf, g, cur.p, !calm || (t0 & (NP == 6 ? 32u : 96u)) == 0u);
            CLX_STAT(46, 1);
        }
#ifdef CLN_LAND_FIRST
        cln_flush<false>(T, stage, M, lane);             // (measurement: the ring lands in FRONT of the tile stores)
#endif
        const bool was_slow = slow;                      // (no lean turn is tried: the history does not fit the packed form)
        int done = 0;
        bool refilled = false;                           // the ring was refilled on the spot once in this turn (a lane outran it)
        if (!slow) {
          again_lean:
            if (mode == 0 || NP == 2) {                // (NP == 2 is only run with mode 0)
                // a partition edge inside the turn?  (lanes that decode nothing never say yes; cur.pcnt of the others is exact)
                if (clx_any(live && cur.pcnt < 16u)) done = cln_lean_turn<NP, 0, true, 0>(row, g, cur, H, C, shift, 0u, 0u, lim, per, rice2, r.limit, live, K, F, mine, sw);
                else                                 done = cln_lean_turn<NP, 0, false, 0>(row, g, cur, H, C, shift, 0u, 0u, lim, per, rice2, r.limit, live, K, F, mine, sw);
            }
            else                      done = cln_lean_turn<NP, 1, true, 0>(row, g, cur, H, C, shift, 0u, 0u, lim, per, rice2, r.limit, live, K, F, mine, sw);
            if (done > 0) {
                cln_done(T, t0);
                CLX_STAT(50, 1);
                continue;
            }
            if (done == -2 && !refilled) { cln_reset(buf, g, row, (cur.p - 1u) >> 5, CR, crc); refilled = true; CLX_STAT(59, 1); goto again_lean; }
            // not this way: the turn is taken again from where it started, on the unpacked history (sign-extended halves)
            uint32_t U[2 * NP];
#pragma unroll
            for (int j = 0; j < 2 * NP - 1; ++j) U[j] = (uint32_t)((int32_t)H[j] >> 16);
            U[2 * NP - 1] = (uint32_t)(int32_t)(int16_t)(H[2 * NP - 2] & 0xffffu);
#pragma unroll
            for (int j = 0; j < 2 * NP; ++j) H[j] = U[j];
            slow = true;
        }
        // ---- wide turn: a lane's signal is outside the 16-bit range (a loud side channel usually stays there for a while) but
        //      inside the 24-bit one: the same turn with v_mad_i32_i24 on the i32 history.  (After a lean turn that failed for
        //      another reason -- a long code, a partition edge, the ring -- it would fail the same way: the slow turn's.)
        if (was_slow || done < 0) {
            bool in24 = true;
#pragma unroll
            for (int j = 0; j < 2 * NP; ++j) in24 = in24 && (int32_t)H[j] < lim24 && (int32_t)H[j] >= -lim24;
            if (__all(in24 || !live || order == 0u)) {
              again_wide:
                const int dw = cln_lean_turn<NP, 1, true, 1>(row, g, cur, H, C, shift, 0u, 0u, lim24, per, rice2, r.limit, live, K, F, mine, sw);
                if (dw > 0) {
                    cln_done(T, t0);
                    CLX_STAT(58, 1);
                    continue;
                }
                if (dw == -2 && !refilled) { cln_reset(buf, g, row, (cur.p - 1u) >> 5, CR, crc); refilled = true; CLX_STAT(59, 1); goto again_wide; }
            }
        }
        CLX_STAT(51, 1);
        if (++nslow > CLN_SLOW_BUDGET && t0 + 16u * 4u * CLN_SLOW_BUDGET < nmax) return false;      // (wave-uniform; not when the end is near anyway)
        // ---- slow turn: sixteen samples one by one, generic reader, i64 predictor (taps beyond the order are zero)
        int32_t* const ys = reinterpret_cast<int32_t*>(mine);
        uint32_t wild = 0u;
#pragma unroll 1
        for (uint32_t ii = 0; ii < 16u; ++ii) {
            int32_t x = 0;
            if (live) x = cln_careful_code(r, cur, per, rice2, K);
            int64_t acc = 0;
#pragma unroll
            for (int j = 2 * NP - 1; j >= 0; --j) acc += (int64_t)cln_coef<NP>(C, j) * (int64_t)(int32_t)H[j];
            const int32_t s = (int32_t)((uint32_t)x + (uint32_t)(int32_t)(acc >> shift));
#pragma unroll
            for (int j = 2 * NP - 1; j > 0; --j) H[j] = H[j - 1];
            H[0] = (uint32_t)s;
            const int32_t v = clx_lfinish(s, F);             // (a wave of plain mid/side pairs has no stereo form here: its movers', cln_ms4)
            wild |= cln_ms_wild(s);
            ys[((ii >> 2) ^ sw) * 4u + (ii & 3u)] = v;
        }
        if (M.ms && __any(live && (wild >> 31) != 0u)) return false;       // (however near the end: the movers' short form is not the reference's out there)
        cln_done(T, t0);
        ring_ok = false;                                  // the position moved without the ring
    }
    return true;
}

// LPC / fixed parameters of a lane after the prologue, in the lean kernel's form
template <int NP>
__device__ __forceinline__ bool cln_run(const clx_buf& buf, LaneState<12>& S, LRing& g, uint32_t* row, int4* stage, uint32_t n, uint32_t i0, uint32_t nmax,
                                        const LKind& K, int mode, const Finish& F, const LMover& M, LTile& T, int lane, LCrc& CR, bool crc, bool calm) {
    uint32_t C[NP], H[2 * NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) C[q] = ((uint32_t)S.c[2 * q] << 16) | ((uint32_t)S.c[2 * q + 1] & 0xffffu);
#pragma unroll
    for (int j = 0; j < 2 * NP; ++j) H[j] = (uint32_t)S.hist[j];
    LCur cur = { S.r.pos, S.k, S.pcnt, S.next_cnt, S.parts_left };
    if (n == 0u) cur.p = 32u;            // a lane that decodes nothing rides along from a harmless position (never committed)
    // range in which the packed evaluation is exact: 16-bit factors, and no partial sum of the taps wraps 32 bits
    // (S.lim = min(2^23, (2^31 - 1) / sum|c|), clx_ltransition); a subframe without taps has nothing to keep in range
    // (a subframe without taps has no history to keep in range; 2^23 before and after the wasted-bits shift is what the stereo
    //  forms need: clx_decor4_mad.  16-bit audio -- 17 bits in a side channel -- never comes near it)
    const int32_t cap = (1 << 23) >> (int)F.wasted;       // (what is shifted left by the wasted bits must still fit)
    const int32_t lim0 = S.order == 0u ? (1 << 29) : S.lim < 32768 ? S.lim : 32768;
    const int32_t lim = lim0 < cap ? lim0 : cap;
    // the 24-bit evaluation's range (clx_ltransition), under the same cap for subframes without taps
    const int32_t lim24a = S.order == 0u ? (1 << 29) : S.lim;
    const int32_t lim24 = lim24a < cap ? lim24a : cap;
    const bool done = cln_body<NP>(buf, S.r, g, row, stage, cur, H, C, S.order, S.shift, lim, lim24, S.per, S.rice2, n, i0, nmax, K, mode, F, M, T, lane, CR, crc, calm);
    S.r.pos = cur.p; S.k = cur.k; S.pcnt = cur.pcnt; S.next_cnt = cur.next; S.parts_left = cur.parts;
    return done;
}

// ---- clx_k_lean24: the split form --------------------------------------------------------------------------------------------
// history as 12-bit lo / 16-bit hi pieces in packed pairs (cln_lean_turn, FORM 2): H[j] = pieces of (lo half: s[-2-j], hi half:
// s[-1-j]) for the lo pieces, H[NH + j] the same for the hi pieces; U[j] = s[-1-j]
template <int NP>
__device__ __forceinline__ void cln_pack12(const int32_t (&U)[2 * NP], uint32_t (&H)[2 * (2 * NP - 1)]) {
    constexpr int NH = 2 * NP - 1;
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        H[j] = (((uint32_t)U[j] & 0xfffu) << 16) | ((uint32_t)U[j + 1] & 0xfffu);
        H[NH + j] = ((uint32_t)(U[j] >> 12) << 16) | ((uint32_t)(U[j + 1] >> 12) & 0xffffu);
    }
}
template <int NP>
__device__ __forceinline__ void cln_unpack12(const uint32_t (&H)[2 * (2 * NP - 1)], int32_t (&U)[2 * NP]) {
    constexpr int NH = 2 * NP - 1;
#pragma unroll
    for (int j = 0; j < NH; ++j) U[j] = (int32_t)((uint32_t)((int32_t)H[NH + j] >> 16) << 12) + (int32_t)(H[j] >> 16);
    U[2 * NP - 1] = (int32_t)((uint32_t)(int32_t)(int16_t)(H[2 * NH - 1] & 0xffffu) << 12) + (int32_t)(H[NH - 1] & 0xffffu);
}

// The steady state of clx_k_lean24: split turns; the slow turn (as cln_body's) for what they leave -- and for as long as a lane's
// history is outside the range in which the split evaluation is exact.  Returns false when the wave gives the group up.
template <int NP, int OMAX>
__device__ __forceinline__ bool cln_body24(const clx_buf& buf, LaneReader& r, LRing& g, uint32_t* row, int4* stage, LCur& cur, const int32_t (&hist0)[OMAX],
                                           const uint32_t (&C)[NP], uint32_t order, uint32_t shift, int32_t lim, uint32_t per, uint32_t rice2,
                                           uint32_t n, uint32_t i0, uint32_t nmax, const LKind& K, const Finish& F, const LMover& M, LTile& T, int lane,
                                           LCrc& CR, bool crc, bool calm) {
    constexpr int NH = 2 * NP - 1;
    const uint32_t sw = ((uint32_t)lane >> 1) & 3u;
    const uint32_t e1 = shift <= 12u ? 12u - shift : 0u, e2 = shift <= 12u ? shift : 12u, e3 = shift <= 12u ? 0u : shift - 12u;
    uint32_t H[2 * NH];
    {   // (a history outside the range cannot even be kept in the pieces: such a wave -- garbage, or audio beyond 24 bits -- gives
        //  the group up at once, here and behind a slow turn)
        int32_t U[2 * NP];
        bool in = true;
#pragma unroll
        for (int j = 0; j < 2 * NP; ++j) { U[j] = hist0[j]; in = in && U[j] < lim && U[j] >= -lim; }
        if (!__all(in || n == 0u || r.err != 0u || order == 0u)) return false;
        cln_pack12<NP>(U, H);
    }
    bool ring_ok = false;
    uint32_t nslow = 0;
    for (uint32_t t0 = i0; t0 < nmax; t0 += 16u) {
        const bool live = n != 0u && !r.err;
        cln_flush<true>(T, stage, M, lane);              // the pair of tiles before, once it is complete
        int4* const mine = cln_mine(stage, t0, lane);
        if (!ring_ok) { cln_reset(buf, g, row, (cur.p - 1u) >> 5, CR, crc); ring_ok = true; }
        else if (cln_pump_now(calm, (t0 & 16u) == 0u)) { cln_land(g, row, CR, crc); cln_request(buf, g, cur.p, !calm || (t0 & 96u) == 0u); }
        {
            bool refilled = false;
          again:
            const int done = cln_lean_turn<NP, 1, true, 2>(row, g, cur, H, C, e2, e1, e3, lim, per, rice2, r.limit, live, K, F, mine, sw);
            if (done > 0) {
                cln_done(T, t0);
                CLX_STAT(9, 1);
                continue;
            }
            if (done == -2 && !refilled) { cln_reset(buf, g, row, (cur.p - 1u) >> 5, CR, crc); refilled = true; CLX_STAT(12, 1); goto again; }
        }
        CLX_STAT(10, 1);
        if (++nslow > CLN_SLOW_BUDGET && t0 + 16u * 4u * CLN_SLOW_BUDGET < nmax) return false;      // (wave-uniform; not when the end is near anyway)
        // ---- slow turn: sixteen samples one by one, generic reader, i64 predictor (taps beyond the order are zero)
        int32_t U[2 * NP];
        cln_unpack12<NP>(H, U);
        int32_t* const ys = reinterpret_cast<int32_t*>(mine);
        uint32_t wild = 0u;
#pragma unroll 1
        for (uint32_t ii = 0; ii < 16u; ++ii) {
            int32_t x = 0;
            if (live) x = cln_careful_code(r, cur, per, rice2, K);
            int64_t acc = 0;
#pragma unroll
            for (int j = 2 * NP - 1; j >= 0; --j) acc += (int64_t)cln_coef<NP>(C, j) * (int64_t)U[j];
            const int32_t s = (int32_t)((uint32_t)x + (uint32_t)(int32_t)(acc >> shift));
#pragma unroll
            for (int j = 2 * NP - 1; j > 0; --j) U[j] = U[j - 1];
            U[0] = s;
            const int32_t v = clx_lfinish(s, F);
            wild |= cln_ms_wild(s);
            ys[((ii >> 2) ^ sw) * 4u + (ii & 3u)] = v;
        }
        if (M.ms && __any(live && (wild >> 31) != 0u)) return false;       // (as cln_body's: the movers' short form is not the reference's out there)
        bool in = true;
#pragma unroll
        for (int j = 0; j < 2 * NP; ++j) in = in && U[j] < lim && U[j] >= -lim;
        if (!__all(in || !live || order == 0u)) return false;
        cln_pack12<NP>(U, H);
        cln_done(T, t0);
        ring_ok = false;                                  // the position moved without the ring
    }
    return true;
}

template <int NP, int OMAX>
__device__ __forceinline__ bool cln_run24(const clx_buf& buf, LaneState<OMAX>& S, LRing& g, uint32_t* row, int4* stage, uint32_t n, uint32_t i0, uint32_t nmax,
                                          const LKind& K, const Finish& F, const LMover& M, LTile& T, int lane, LCrc& CR, bool crc, bool calm) {
    static_assert(2 * NP <= OMAX, "taps");
    uint32_t C[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) C[q] = ((uint32_t)S.c[2 * q] << 16) | ((uint32_t)S.c[2 * q + 1] & 0xffffu);
    LCur cur = { S.r.pos, S.k, S.pcnt, S.next_cnt, S.parts_left };
    if (n == 0u) cur.p = 32u;            // a lane that decodes nothing rides along from a harmless position (never committed)
    // the range in which the split evaluation is exact (cln_lean_turn): (floor((2^31 - 1) / sum|c|) - 1) * 4096, at most 2^27 (the
    // hi piece is a 16-bit factor); a subframe without taps has no history to keep in range: 2^29 there and as the cap (what the
    // short mid/side form and the wasted-bits shift still hold)
    const int32_t cap = (1 << 29) >> (int)F.wasted;
    const int32_t lim0 = S.order == 0u ? (1 << 29) : S.lim > (1 << 15) ? (1 << 27) : (int32_t)((uint32_t)(S.lim - 1) << 12);
    const int32_t lim = lim0 < cap ? lim0 : cap;
    const bool done = cln_body24<NP, OMAX>(buf, S.r, g, row, stage, cur, S.hist, C, S.order, S.shift, lim, S.per, S.rice2, n, i0, nmax, K, F, M, T, lane, CR, crc, calm);
    S.r.pos = cur.p; S.k = cur.k; S.pcnt = cur.pcnt; S.next_cnt = cur.next; S.parts_left = cur.parts;
    return done;
}

// The kernels' common body.  SPLIT = false: clx_k_lean (<= 16-bit audio, <= 12 taps); true: clx_k_lean24 (<= 24-bit audio -- a side
// channel has 25 --, <= 32 taps, the groups clx_k_lean left).
// (bx: the group of 64 slots of run R that the wave decodes -- the workgroup's index in clx_k_lean / clx_k_lean24, a decode ticket of clx_k_pool)
template <bool SPLIT>
__device__ __forceinline__ void cln_kernel(LeanLds& L, const clx_run& R, const clx_dev_frame* __restrict__ frames,
                                           uint32_t n_slots, int32_t* __restrict__ dump_all, uint32_t bx, uint32_t run_idx, int lane) {
    constexpr int OMAX = SPLIT ? 32 : 12;
    if (SPLIT && R.taken[bx] == R.gen) return;               // clx_k_lean has decoded this group
    CLX_TL_BEGIN();                                          // (-DCLX_TIMELINE builds only: tools/timeline_lean.py)
    const uint8_t* const arena = R.arena;
    const uint64_t arena_alloc_len = R.alloc_len;
    const uint32_t* const sf_start = R.sf_start;
    int32_t* const out = R.out;
    uint32_t* const errkey = R.errkey;
    uint64_t* const end_bits = R.end_bits;
    uint32_t* const taken = R.taken;
    const uint32_t gen = R.gen;
    const uint32_t slot = bx * 64u + (uint32_t)lane;
    // (which frame this lane decodes: the run's slot map -- the plan's, or what clx_k_compose dealt; what the scan and the CRC parts
    //  are indexed by is the PLAN's slot of the subframe, cslot)
    uint32_t f = 0xffffffffu;
    if (slot < n_slots) f = R.slot_frame[slot];
    clx_dev_frame fr;
    fr.byte_off = 0; fr.out_off = 0; fr.limit_bits = 0; fr.first_slot = 0; fr.header_bytes = 0; fr.block_size = 0;
    fr.n_channels = 0; fr.channel_assignment = 0; fr.bps = 1; fr.flags = 0;
    if (f != 0xffffffffu) fr = frames[f];
    const uint32_t ch = (f != 0xffffffffu) ? slot - R.first_slot[f] : 0u;
    const uint32_t cslot = fr.first_slot + ch;
    const uint32_t bs = fr.block_size;

    LaneReader r;
    r.arena = arena;
    r.origin = (uint32_t)(fr.byte_off & ~15ull);
    const uint32_t o = 8u * (uint32_t)(fr.byte_off & 15ull);
    r.limit = o + fr.limit_bits;
    r.pos = o + 8u * (uint32_t)fr.header_bytes;
    r.err = 0u;
    bool active = (f != 0xffffffffu);
    if (active && ch != 0u) {
        const uint32_t sp = clx_peek_u32(&sf_start[cslot]);      // (past the caches: the scan wave that wrote it may belong to this very kernel, clx_k_pool)
        if (sp == 0xffffffffu) active = false;             // an earlier channel failed (the scan reported it): decodes nothing
        else r.pos = sp;
    }
    const uint32_t pos0 = r.pos;                           // where the lane's subframe starts
    // ---- does this wave qualify?  Every live lane: <= 16-bit audio, a FIXED / LPC subframe of at most 12 taps whose header
    //      parses, the wave's common block size (a multiple of 16, beyond the prologue), a 16-byte aligned row.
    // (narrow output: `out` holds interleaved 16-bit PCM, a lane's "row" is its FRAME's block there -- both lanes of a pair point to it)
    const bool pcm16 = (R.flags & CLX_RUN_PCM16) != 0u, pcm24 = (R.flags & CLX_RUN_PCM24) != 0u;      // wave-uniform
    int32_t* const rowp = pcm16 ? reinterpret_cast<int32_t*>(reinterpret_cast<int16_t*>(out) + (active ? fr.out_off : 0ull))
                        : pcm24 ? reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(out) + (active ? 3ull * fr.out_off : 0ull))
                                : out + (active ? fr.out_off + (uint64_t)ch * fr.block_size : 0ull);
    // (where the wave's rows are: the lowest one's address, wave-uniform, and every lane's distance from it -- LMover)
    uint64_t row_lo = active ? (uint64_t)(uintptr_t)rowp : ~0ull;
#pragma unroll
    for (int sx = 32; sx >= 1; sx >>= 1) {
        const uint64_t other = ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(row_lo >> 32), sx, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)row_lo, sx, 64);
        row_lo = other < row_lo ? other : row_lo;
    }
    const uint64_t row_far = active ? (uint64_t)(uintptr_t)rowp - row_lo + 4ull * bs : 0ull;      // (its last byte's distance, + 1)
    bool good = !active;
    SfHead h = { 1u, 0u, 0u, 1u };
    uint32_t bs0 = 0;
    {
        const unsigned long long am = __ballot(active);
        if (am == 0ull) return;                            // nothing to decode here
        bs0 = (uint32_t)__shfl((int)bs, (int)__ffsll((long long)am) - 1, 64);
    }
    if (active) {
        good = fr.bps <= (SPLIT ? 24u : 16u) && bs == bs0 && (bs & 15u) == 0u && bs >= (SPLIT ? 64u : 32u) && (((uintptr_t)rowp) & 15u) == 0u && row_far < 0xffffffffull &&
               (!pcm16 || (!SPLIT && ((fr.n_channels == 2u && ch == (slot & 1u)) || fr.n_channels == 1u))) &&       // (narrow output: stereo frames, channel c in lane parity c -- or mono frames)
               (!pcm24 || (SPLIT && fr.n_channels == 2u && ch == (slot & 1u))) &&        // (packed 24-bit output: stereo frames, the split tier -- 16-bit frames too)
               r.pos <= r.limit && (uint64_t)r.origin + 4ull * ((uint64_t)r.limit / 32ull + 16ull) < 0xffffffffull;
        if (good) {
            h = clx_lparse_sf_header(r, clx_channel_bps(fr, ch));
            good = !r.err && h.order <= (uint32_t)OMAX;    // (constant and verbatim subframes have order 0)
        }
    }
    // (narrow output: one kind of frame per wave -- the movers write stereo lines or mono rows)
    const uint32_t nch0 = (uint32_t)__shfl((int)(uint32_t)fr.n_channels, (int)__ffsll((long long)__ballot(active)) - 1, 64);
    if ((pcm16 || pcm24) && active && fr.n_channels != nch0) good = false;
    if (!__all(good)) {                                    // clx_k_lanes / clx_k_lanes_hi decode this group
        if (!SPLIT) {
            CLX_STAT(60, 1); CLX_STAT(61, active && (fr.bps > 16u || bs != bs0 || (bs & 15u) != 0u || bs < 32u)); CLX_STAT(62, active && r.err != 0u);
            CLX_STAT(63, active && !r.err && h.order > 12u);
        }
        return;
    }
    if (lane == 0) taken[bx] = gen;

    // ---- the lane's share of its frame's CRC-16 (LCrc): from the granule in which its subframe starts to the granule in which
    //      the next one does (the scan has said where: sf_start), the last channel's to the end of the frame as the descriptor
    //      gives it; the first granule of the frame is the first channel's, taken here with the bytes in front of the frame masked
    const clx_buf buf = clx_make_buf(arena, (uint32_t)arena_alloc_len);
    const bool crc = (R.flags & CLX_RUN_CRC) != 0u;       // wave-uniform
    // (where the next subframe starts: the end of this lane's share of the frame's CRC, and how many bits its subframe holds)
    const bool last_ch = active && ch + 1u == (uint32_t)fr.n_channels;
    const uint32_t next_sp = (active && !last_ch) ? clx_peek_u32(&sf_start[cslot + 1u]) : 0xffffffffu;
    // a calm wave: no lane's subframe holds more than 6 bits per sample -- its ring is pumped every other turn (cln_pump_now)
    const uint32_t sf_bits = !active ? 0u : last_ch ? o + fr.limit_bits - pos0 : next_sp - pos0;       // (a failed or unbounded one: huge)
    const bool calm = CLN_RING >= 24u && __all(sf_bits <= 6u * bs);
    LCrc CR;
    CR.c.r = 0u; CR.c.x = 0u; CR.next = 0u; CR.db = CLN_CRC_NONE;
    bool crc_mine = false, crc_last = false;
    uint32_t crc_da = 0u;
    if (crc && active && !(fr.flags & 1u)) {               // (a bare subframe has no footer)
        crc_mine = true;
        crc_last = last_ch;
        const uint32_t eb = (o + fr.limit_bits) >> 3;      // the frame's end, in bytes from the origin
        uint32_t db = (eb >> 4) << 2;                      // (the last channel: the frame's whole granules; the rest is taken at the end)
        if (!crc_last) {
            if (next_sp == 0xffffffffu) crc_mine = false;  // (this subframe does not parse: the frame fails)
            db = (next_sp >> 7) << 2;
        }
        if (db < 4u) { if (crc_last) crc_mine = false; db = 4u; }      // (a frame that ends inside its first granule: the stand-alone kernel's)
        crc_da = ch == 0u ? 0u : (pos0 >> 7) << 2;
        if (crc_da < 4u && ch != 0u) crc_da = 4u;
        if (crc_mine) {
            CR.db = db; CR.next = crc_da;
            if (ch == 0u) { cln_crc_take_masked(CR, clx_buf_load16(buf, r.origin), (uint32_t)(fr.byte_off & 15ull), 16u); CR.next = 4u; }
        }
    }

    const uint32_t decor = active ? fr.channel_assignment : 0u;
    const uint32_t pbs = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(active ? bs : 0u), 0xB1, 0xF, 0xF, false);
    const uint32_t pd = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)decor, 0xB1, 0xF, 0xF, false);
    const bool pair_ok = active && decor != CLX_CH_INDEPENDENT && pbs == bs && pd == decor;
    uint32_t omax = active ? h.order : 0u;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { const uint32_t b = __shfl_xor(omax, s, 64); omax = b > omax ? b : omax; }
    const uint32_t nmax = bs0;
    const uint32_t n = active ? bs : 0u;

    // ---- where the wave's rows are (cln_store_pair), dump slots for rows that do not exist
    LMover M;
    M.base = ((uint64_t)clx_uniform((uint32_t)(row_lo >> 32)) << 32) | clx_uniform((uint32_t)row_lo);
    M.rowoff = active ? (uint32_t)((uint64_t)(uintptr_t)rowp - row_lo) : CLN_NO_ROW;
    M.all_real = __all(active);
    M.pcm16 = pcm24 ? 3u : !pcm16 ? 0u : nch0 == 1u ? 2u : 1u;
    (void)dump_all;                                         // (rows that do not exist are not written: no dump slots here)
    Finish F = clx_lfinish_setup(n, h.kind == 0u ? 0u : h.wasted, decor, pair_ok, lane);      // (a constant's wasted bits are folded into it below)
    // every lane in a mid/side pair, no wasted bits (the 16-bit tier; planar or interleaved 16-bit output): the stage takes mid and side as
    // they are decoded and the movers undo the pair (cln_ms4) -- to the turns, the slow turn and the prologue such a wave has no stereo at all
    M.ms = F.ms_plain && M.pcm16 <= 1u;
    if (M.ms) { F.all_ms = false; F.any_decor = false; F.ms_plain = false; }

    // ---- careful prologue (as clx_lanes_body's): warm-up samples, the transition, the first residuals -- one sample per turn of
    //      a rolled loop, i64 predictor; leaves every lane on a multiple of 16 samples, past its transition
    LaneState<OMAX> S;
    S.r = r;
#pragma unroll
    for (int j = 0; j < OMAX; ++j) { S.c[j] = 0; S.hist[j] = 0; }
    S.phase = 3u; S.cval = 0; S.trans_at = 0xffffffffu; S.transitioned = false;
    S.order = 0; S.shift = 0; S.lim = 0x7fffffff;
    S.k = 0; S.k1 = 1; S.pcnt = 0; S.next_cnt = 0; S.per = 0; S.parts_left = 0; S.rice2 = 0;
    if (n) {                                               // (bs > order: "order larger than block size" cannot happen here)
        if (h.kind == 0u) {                                // decode_constant (subframe.rs:382-394) + its wasted-bits shift (216-225)
            S.cval = (int32_t)((uint32_t)clx_lread_signed(S.r, h.sf_bps) << h.wasted); S.phase = 2u;
        }
        else if (h.kind == 1u) S.phase = 0u;                                               // decode_verbatim (397-415)
        else { S.phase = 0u; S.trans_at = h.order; }
    }
    const uint32_t i0 = (omax + 4u + 15u) & ~15u;           // 16 or 32 (SPLIT: up to 48) (<= bs)
    LTile T = { 0u, 0u };
    int4* const stage0 = &L.stage[0][0][0];
    const uint32_t sw = ((uint32_t)lane >> 1) & 3u;
    uint32_t wild = 0u;                                     // M.ms: a sample outside the range in which cln_ms4 is the reference's step (bit 31)
#pragma unroll 1
    for (uint32_t i = 0; i < i0; ++i) {
        const int32_t x = clx_lcareful_raw<OMAX>(S, h, bs, i, n);
        const int32_t pred = clx_lpredict<OMAX, true>(S.c, S.hist, S.shift);
        const uint32_t use = (S.order != 0u && i >= S.order && i >= S.trans_at) ? 0xffffffffu : 0u;
        const int32_t s = (int32_t)((uint32_t)x + ((uint32_t)pred & use));
#pragma unroll
        for (int j = OMAX - 1; j > 0; --j) S.hist[j] = S.hist[j - 1];
        S.hist[0] = s;
        wild |= cln_ms_wild(s);
        // (the split tier's prologue is up to three tiles long and flushes a pair here; the 16-bit tier's is ONE tile and never does -- the call stays
        //  all the same: round 6's builds without it faulted on the GPU in clx_k_pool, for no reason found in the code)
        if ((i & 15u) == 0u) cln_flush<SPLIT>(T, stage0, M, lane);
        reinterpret_cast<int32_t*>(cln_mine(stage0, i, lane))[(((i >> 2) & 3u) ^ sw) * 4u + (i & 3u)] = clx_lfinish(s, F);
        if ((i & 15u) == 15u) cln_done(T, i & ~15u);
    }
    // (M.ms and a prologue sample out there -- a damaged or crafted stream: the group is given up below, clx_k_lanes' stereo step is exact for every value)
    const bool tame = !(M.ms && __any(n != 0u && (wild >> 31) != 0u));
    // ---- steady state
    LRing g;
    g.origin = r.origin; g.fill = 0; g.fs = 0; g.np = 0; g.pa = make_uint4(0u, 0u, 0u, 0u); g.pb = g.pa; g.pc = g.pa;
    bool done = tame;
    if (tame && i0 < nmax) {
        // what every lane does from here on (the prologue is over: predicted subframes have switched to residuals)
        LKind K;
        K.rice = S.phase == 1u; K.verb = S.phase == 0u;
        K.bitmask = S.phase == 2u ? 0u : 0xffffffffu;
        K.ricemask = K.rice ? 0xffffffffu : 0u;
        K.verbmask = K.verb ? 0xffffffffu : 0u;
        K.cor = S.phase == 2u ? (uint32_t)S.cval : 0u;
        K.vsh = (32u - h.sf_bps) & 31u;
        K.vshm = K.verb ? K.vsh : 0u;
        const bool lv = n != 0u && !S.r.err;
        const int mode = __any(lv && !K.rice) ? 1 : 0;                                    // wave-uniform: the mixed form or the plain one
        if constexpr (SPLIT) {
            // the split evaluation needs sum|c| < 2^19 (S.lim >= 4096) -- any <= 32 coefficients of <= 15 bits but the all -2^14 row
            if (__any(lv && S.order != 0u && S.lim < 4096)) done = false;
            else if (omax <= 12u) done = cln_run24<6, OMAX>(buf, S, g, &L.ring[0][lane], stage0, n, i0, nmax, K, F, M, T, lane, CR, crc, calm);
            else                  done = cln_run24<16, OMAX>(buf, S, g, &L.ring[0][lane], stage0, n, i0, nmax, K, F, M, T, lane, CR, crc, calm);
        } else {
            if (omax <= 4u && mode == 0) done = cln_run<2>(buf, S, g, &L.ring[0][lane], stage0, n, i0, nmax, K, mode, F, M, T, lane, CR, crc, calm);
            else if (omax <= 8u)         done = cln_run<4>(buf, S, g, &L.ring[0][lane], stage0, n, i0, nmax, K, mode, F, M, T, lane, CR, crc, calm);
            else                         done = cln_run<6>(buf, S, g, &L.ring[0][lane], stage0, n, i0, nmax, K, mode, F, M, T, lane, CR, crc, calm);
        }
    }
    if (!done) {                                           // given up: clx_k_lanes decodes the group
        if (lane == 0) taken[bx] = 0u;
        CLX_STAT(SPLIT ? 11 : 57, 1);
        return;
    }
    cln_flush_all<SPLIT>(T, stage0, M, lane);               // what is still in the stage
    // ---- trailing parameters of empty partitions are part of the stream (they move the next subframe / the CRC)
    if (n != 0u && !S.r.err && S.transitioned) {
        while (!S.r.err && S.parts_left != 0u) { (void)clx_lread_rice_param(S.r, S.rice2); S.parts_left -= 1u; }
    }
    if (active) {
        if (S.r.err) clx_report_error(errkey, f, ch, S.r.err);
        else if (ch + 1u == fr.n_channels) end_bits[f] = (uint64_t)(S.r.pos - o);
    }
    // ---- the share's remainder goes to clx_k_finalize -- the last channel's only when the frame ends where its descriptor says
    //      (a descriptor may say "at most": then the stand-alone kernel checks the frame, from the end bit)
    if (crc) {
        bool keep = crc_mine && !S.r.err;
        if (keep && crc_last) keep = ((S.r.pos + 7u) & ~7u) + 16u == o + fr.limit_bits;
        if (!keep) CR.db = 0u;                              // (nothing to catch up with)
        cln_crc_catchup(buf, r.origin, CR, CR.db);
        if (keep) {
            uint32_t db = CR.db;
            const uint32_t te = ((o + fr.limit_bits) >> 3) & 15u;
            if (crc_last && te != 0u) { cln_crc_take_masked(CR, clx_buf_load16(buf, r.origin + 4u * db), 0u, te); db += 4u; }
            clx_crc_part part;
            part.gen = gen; part.rx = clx_crct_reduced(CR.c.r) | ((uint32_t)__popc(CR.c.x) << 31); part.da = crc_da; part.db = db;
            R.crc_part[cslot] = part;
        }
    }
    CLX_TL_END_SEQ(3, ((uint64_t)R.gen << 32) | (run_idx << 20) | bx);
    (void)run_idx;
}

#ifndef CLN_WAVES
#define CLN_WAVES 3
#endif
extern "C" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CLN_WAVES)))
void clx_k_lean(const clx_runs runs, const clx_dev_frame* __restrict__ frames, uint32_t n_slots, int32_t* __restrict__ dump_all) {
    __shared__ LeanLds L;
    cln_kernel<false>(L, runs.r[blockIdx.y], frames, n_slots, dump_all, blockIdx.x, blockIdx.y, (int)threadIdx.x);
}

// ---- Q: the scan and the 16-bit tier of one merged launch as tickets (round 6) ---------------------------------------------------
// clx_k_pool: a grid of waves that STAY (as many as the machine holds at once, or as there are tickets), each taking tickets off a
// counter until none is left -- first the launch's scan waves (run by run), then its groups of 64 slots (run by run).  Why: a launch
// of clx_k_lean over thousands of workgroups shares the machine's wave slots EVENLY with the other stream's launch, so that both run
// out of workgroups -- and drain -- at the same time (profiles/r05_timeline_region_before.txt: the streams fall into step), and the
// scan in front of it is a kernel boundary at which the stream's waves are gone.  Waves that stay hold their slots until their
// launch's tickets are gone: the launches of the two streams take the machine one after the other, the next one's waves moving in
// one by one as this one's leave, and a launch's decode waves start where its scan waves end.
// A decode ticket needs its RUN's scan (sf_start): scan tickets are taken before any decode ticket, so whoever holds one is a
// resident wave and the wait ends; all the same it is bounded -- a wave that waits longer than any scan can take leaves its group
// unmarked, i.e. to the general kernels behind (which run when this kernel, and with it every scan ticket, is over).
// order: a test hook (the wave simulator takes the tickets in a given order); null on the device.
// What the scan waves hand to the decode waves is sf_start alone (errors go through atomics): written and read PAST the caches
// (clx_poke_u32 / clx_peek_u32) instead of fenced -- an agent-scope acquire in front of every decode ticket invalidates the XCD's whole
// L2, three times a microsecond, under the decode lanes' streams (eight 16-byte granules to a line): measured, the pool's launch of
// twelve runs 2.29 ms with the fences against 2.0 ms for round 5's two kernels.
#ifndef CLN_POOL_FENCES
#define CLN_POOL_FENCES 0
#endif
#ifndef CLN_POOL_SPIN
#define CLN_POOL_SPIN (1u << 18)        // looks at the counter before a decode wave gives up (about a second)
#endif
extern "C" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CLN_WAVES)))
void clx_k_pool(const clx_pool_args args) {
    __shared__ LeanLds L;
    (void)args;
#pragma unroll 1
    for (;;) {
        // (the arguments are read again for every ticket, through a pointer the compiler cannot see through: whatever it hoisted out
        //  of this loop -- the arguments, what the bodies derive from them and from the lane number -- would stay in registers across
        //  the decode loop, which has none to spare)
        const clx_pool_args* A = CLX_KERNARGS(clx_pool_args);
        CLX_OPAQUE_PTR(A);
        const uint32_t scan_w = (A->n_multi + 63u) / 64u, groups = (A->n_slots + 63u) / 64u;
        const uint32_t n_scan = A->n_runs * scan_w, total = n_scan + A->n_runs * groups;
        clx_pool_state* const ps = A->ps;
        int lane = (int)threadIdx.x;
        CLX_OPAQUE(lane);                                     // (nothing derived from the lane number is carried from ticket to ticket)
        uint32_t t = 0;
        if (threadIdx.x == 0) t = atomicAdd(&ps->next, 1u);
        t = clx_readlane(t, 0u);
        if (t >= total) break;
        if (A->order != nullptr) t = A->order[t];
        if (t < n_scan) {
            const uint32_t run = t / scan_w, w = t - run * scan_w;
            cln_scan_wave(&L.ring[0][0], A->runs.r[run], A->frames, A->multi, A->n_multi, w, run, lane);
#if CLN_POOL_FENCES
            clx_release();                                    // (every lane's sf_start / errkey / fkey stores)
#else
            clx_stores_done();                                // (every lane's sf_start stores went past the caches and have arrived; errkey is atomics)
#endif
            if (threadIdx.x == 0) atomicAdd(&ps->scan_done[run], 1u);
        } else {
            const uint32_t u = t - n_scan, run = u / groups, g = u - run * groups;
            bool ready = scan_w == 0u;
#pragma unroll 1
            for (uint32_t looks = 0; !ready && looks < CLN_POOL_SPIN; ++looks) {
                ready = clx_readlane(clx_peek_u32(&ps->scan_done[run]), 0u) >= scan_w;
                if (!ready) clx_pause();
            }
            if (ready) {
#if CLN_POOL_FENCES
                clx_acquire();
#endif
                cln_kernel<false>(L, A->runs.r[run], A->frames, A->n_slots, A->dump_all, g, run, lane);
            }
            else if (threadIdx.x == 0) atomicAdd(&ps->stuck, 1u);
        }
        clx_wave_sync();                                      // (the next ticket's ring and stage start from scratch: one wave, program order)
    }
}

extern "C" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
void clx_k_lean24(const clx_runs runs, const clx_dev_frame* __restrict__ frames, uint32_t n_slots, int32_t* __restrict__ dump_all) {
    __shared__ LeanLds L;
    cln_kernel<true>(L, runs.r[blockIdx.y], frames, n_slots, dump_all, blockIdx.x, blockIdx.y, (int)threadIdx.x);
}
