// clx_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the batched FLAC frame decoder.
//
//   K1  clx_k_residual   one wavefront per frame: subframe headers, warm-up samples, LPC coefficients,
//                        Rice/Rice2 partitioned residual decode (wave-parallel), CONSTANT / VERBATIM fill.
//                        Replaces subframe.rs:29-91, 236-415, 492-511, 651-708 and the channel dispatch
//                        of frame.rs:705-742 (one bit cursor, subframes in sequence).
//   K2  clx_k_predict    one lane per channel, two waves per 64 channels: fixed / LPC synthesis as an integer
//                        IIR (i64 accumulate, arithmetic >> shift, wrapping i32) in the predictor wave;
//                        wasted-bits shift, left/side, right/side, mid/side decorrelation and the write-back
//                        in the finisher wave.  Replaces subframe.rs:216-225, 417-474, 524-614, frame.rs:319-389.
//   K3  clx_k_crc16      one wavefront per frame: CRC-16 of the frame's bytes against its footer
//                        (frame.rs:752-763, crc.rs:109-112) as a GF(2) fold of per-lane partial CRCs.
//   K4  clx_k_interleave planar i32 -> channel-interleaved little-endian PCM (lib.rs:473-520).
//   K5-K7 clx_k_find_headers / clx_k_span_crc16 / clx_k_gather_headers: frame indexer for raw streams (8 f2).
//
// No MFMA anywhere: this is bit-serial / integer-recurrence work bounded by HBM traffic and issue
// latency, not a dense contraction.  All data-dependent control flow around cross-lane operations
// (ballot / shuffle / barrier) is kept wave-uniform.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <clx_intrin.h>
#include <clx_k2_dot2.h>

#include "../../include/claxon_hip.h"
#include "clx_device.h"

#define CLX_NW 512u            // LDS bitstream window: 512 big-endian-normalised dwords = 2 KiB per wave
#define CLX_NPOS 1152u         // code start positions of one span (u16, span relative); shares LDS with the exit tables

// ------------------------------------------------------------------------------------------------
// Bit window: a 2 KiB slice of the frame's bytes staged in LDS with coalesced 16-byte loads.  Dwords
// are byte-swapped on the way in so that FLAC's MSB-first bit order (input.rs:447-468) becomes plain
// shifts.  Bit positions (`pos`) are relative to `origin` = the arena's 16-byte granule holding the
// frame's first byte, so every global access is a naturally aligned dword / dwordx4.
// ------------------------------------------------------------------------------------------------
struct K1Lds {
    uint32_t Wfront[4];        // never written: lets the extraction read the dword in front of W[0] (its bits are masked off)
    uint32_t W[CLX_NW + 4];
    // The two phases of a span never overlap: first the exit-state tables resolve where every lane's chunk is entered,
    // then the positions of the codes are listed.  4.9 KiB per wave in all = 8 waves per SIMD.
    union {
        struct {
            // tab[lane][s]: exit state of that lane's chunk for entry state s, one byte each, so that a step of the walk
            // is "add the state to the row's address, read a byte"; rows are padded to 36 bytes (9 dwords): the eight
            // groups of lanes that read their rows at the same time then hit different LDS banks
            uint8_t  tab[64][36];
            uint8_t  gtab[8][32];  // gtab[g][s]: exit state of lane group g (8 lanes) for entry state s
        } t;
        uint16_t P[CLX_NPOS];      // P[i]: bit position (relative to the span) where code i of the span starts
    } u;
    // lut[nib] = {f, g} = {F, G}: what four stream bits `nib` (MSB first) do to a walk, for EVERY state the walk can be in and for the
    // Rice parameter of the partition being decoded (k <= CLX_LUT_KMAX; copied from the ROM when k changes).  States are carried
    // multiplied by five (s5 = 5 * state, state <= k + 1 <= 5): field s5 (5 bits wide) of F is 5 * (the state behind the four
    // bits), of G which of the four positions are code starts (bit i = the i-th bit).  A step of a walk is then ONE v_bfe_u32
    // with the state as its offset -- no address arithmetic, no LDS access that depends on the state: the table rows of a
    // chunk's nibbles are fetched ahead of the walks.
    struct alignas(8) Row { uint32_t f, g; } lut[16];
    uint8_t gent[8];           // entry state of each group of 8 chunks / of each chunk: every lane writes the entries of the
    uint8_t ent[64];           // chains it walks (lanes of a group walk the same chain) and reads back its own
};

// The transition tables of K1Lds::lut for the Rice parameters that use them (k <= CLX_LUT_KMAX: k + 2 states of 5 bits each fill
// a dword), generated at compile time and kept in constant memory: switching the LDS copy to another parameter is one 4-byte
// load + store in 32 lanes.
#define CLX_LUT_KMAX 4u
struct K1LutRom { uint32_t w[CLX_LUT_KMAX + 1u][32]; };
constexpr K1LutRom clx_make_lut_rom() {
    K1LutRom r{};
    for (uint32_t k = 0; k <= CLX_LUT_KMAX; ++k) {
        const uint32_t SC = k + 1u;
        for (uint32_t nib = 0; nib < 16u; ++nib) {
            uint32_t F = 0, G = 0;
            for (uint32_t s = 0; s <= SC; ++s) {
                uint32_t st = s, starts = 0;
                for (uint32_t i = 0; i < 4u; ++i) {
                    if (st == 0u) starts |= 1u << i;                                   // a code starts at this bit
                    const bool rem = st >= 1u && st <= k;                              // a remainder bit: count it down
                    st = rem ? st - 1u : (((nib >> (3u - i)) & 1u) ? k : SC);          // a one ends the run, a zero continues it
                }
                F |= (5u * st) << (5u * s);
                G |= starts << (5u * s);
            }
            r.w[k][2u * nib] = F; r.w[k][2u * nib + 1u] = G;
        }
    }
    return r;
}
__constant__ K1LutRom clx_lut_rom = clx_make_lut_rom();

struct BitSrc {
    const uint32_t* origin;    // arena + (byte_off & ~15)
    uint32_t avail_dw;         // dwords readable from origin (arena allocation is padded to 16 B)
    uint32_t win_dw;           // first dword (relative to origin) held in L.W
};

__device__ __forceinline__ uint32_t clx_gload_dw(const BitSrc& b, uint32_t dw) {
    return dw < b.avail_dw ? __builtin_bswap32(b.origin[dw]) : 0u;
}

// (Re)stage the window so that it starts at the 16-byte granule containing bit `pos`.  Wave-uniform.
__device__ __forceinline__ void clx_window_load(K1Lds& L, BitSrc& b, uint32_t pos, int lane) {
    __syncthreads();
    b.win_dw = (pos >> 5) & ~3u;
#pragma unroll
    for (int r = 0; r < (int)(CLX_NW / 256); ++r) {
        uint32_t dw = b.win_dw + 4u * (uint32_t)(lane + 64 * r);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (dw + 4u <= b.avail_dw) v = *reinterpret_cast<const uint4*>(b.origin + dw);
        else if (dw < b.avail_dw) {           // ragged tail (never happens with a 16-byte padded arena)
            v.x = b.origin[dw];
            if (dw + 1 < b.avail_dw) v.y = b.origin[dw + 1];
            if (dw + 2 < b.avail_dw) v.z = b.origin[dw + 2];
        }
        uint32_t* dst = &L.W[4 * (lane + 64 * r)];
        dst[0] = __builtin_bswap32(v.x); dst[1] = __builtin_bswap32(v.y);
        dst[2] = __builtin_bswap32(v.z); dst[3] = __builtin_bswap32(v.w);
    }
    if (lane < 4) L.W[CLX_NW + lane] = clx_gload_dw(b, b.win_dw + CLX_NW + (uint32_t)lane);
    __syncthreads();
}

// Make sure bits [pos, pos+nbits) (+ one dword of slack) are inside the LDS window.  Wave-uniform.
__device__ __forceinline__ void clx_window_ensure(K1Lds& L, BitSrc& b, uint32_t pos, uint32_t nbits, int lane) {
    uint32_t first = pos >> 5, last = (pos + nbits + 63u) >> 5;
    if (first < b.win_dw || last >= b.win_dw + CLX_NW) clx_window_load(L, b, pos, lane);
}

// 32 bits starting at bit `pos`, left aligned.  LDS when inside the window, else (rare) global memory.
__device__ __forceinline__ uint32_t clx_peek32(const K1Lds& L, const BitSrc& b, uint32_t pos) {
    uint32_t dw = pos >> 5, off = pos & 31u;
    uint32_t i = dw - b.win_dw;
    uint32_t hi, lo;
    if (i < CLX_NW + 3u) { hi = L.W[i]; lo = L.W[i + 1]; }
    else { hi = clx_gload_dw(b, dw); lo = clx_gload_dw(b, dw + 1); }
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (uint32_t)((v << off) >> 32);
}

// The same for a position the caller knows to be inside the window (clx_window_ensure): no bounds test, no fallback.
__device__ __forceinline__ uint32_t clx_peek32_win(const K1Lds& L, const BitSrc& b, uint32_t pos) {
    const uint32_t i = (pos >> 5) - b.win_dw;
    const uint64_t v = ((uint64_t)L.W[i] << 32) | L.W[i + 1];
    return (uint32_t)((v << (pos & 31u)) >> 32);
}

// `bits` (1..32) bits at `pos`, right aligned; 0 bits -> 0 (read_leq_u8(0) == 0, input.rs:706).
__device__ __forceinline__ uint32_t clx_peek_bits(const K1Lds& L, const BitSrc& b, uint32_t pos, uint32_t bits) {
    return bits ? (clx_peek32(L, b, pos) >> (32u - bits)) : 0u;
}
// sign-extended `bits`-wide field (extend_sign_u32, subframe.rs:117-122)
__device__ __forceinline__ int32_t clx_peek_signed(const K1Lds& L, const BitSrc& b, uint32_t pos, uint32_t bits) {
    return (int32_t)clx_peek32(L, b, pos) >> (32u - bits);
}

__device__ __forceinline__ uint32_t clx_wave_excl_scan(uint32_t v, int lane, uint32_t* total) {
    // DPP scan: four shifted adds inside each row of 16 lanes, then the row totals are broadcast down (row_bcast:15 into
    // rows 1 and 3, row_bcast:31 into rows 2 and 3): six instructions, no LDS traffic
    (void)lane;
    int incl = (int)v;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);
    *total = clx_readlane((uint32_t)incl, 63u);
    return (uint32_t)incl - v;
}

// value of `v` in the (unique) lane where `pred` holds; `dflt` when there is none.  Wave-uniform result.
__device__ __forceinline__ uint32_t clx_pick(uint32_t v, bool pred, uint32_t dflt) {
    unsigned long long m = __ballot(pred);
    if (m == 0ull) return dflt;
    return clx_readlane(v, (uint32_t)__ffsll((long long)m) - 1u);
}

// Exit state of one chunk for one entry state.  A chunk is `B` stream bits held left-aligned in `c`
// (low 32-B bits zero).  States: m in [0,k] = "m remainder bits still to skip, then a code starts";
// SC = k+1 = "inside a unary run that started earlier".  subframe.rs:337-341 per code:
// q zeros, a one, k remainder bits.
// SENT (B < 32): `c` carries a sentinel one right behind the chunk, so a walk never runs off it and the step is three
// instructions -- shift, count leading zeros, add; a code whose terminator turns out to be the sentinel is a run that
// continues into the next chunk.
template <bool SENT>
__device__ __forceinline__ uint32_t clx_chunk_exit(uint32_t c, uint32_t B, uint32_t k, uint32_t m) {
    const uint32_t SC = k + 1u;
    uint32_t p = (m == SC) ? 0u : m;
    if (SENT) {
        const uint32_t k1 = k + 1u;
        uint32_t end = 0xffffffffu;                      // position behind the last code walked (terminator + 1 + k)
        while (p < B) { p += (uint32_t)__clz((int)(c << p)) + k1; end = p; }
        return (end == B + k1) ? SC : p - B;             // terminator == B: the sentinel
    }
    for (;;) {
        if (p >= B) return p - B;
        uint32_t rest = c << p;
        if (rest == 0u) return SC;
        p += (uint32_t)__clz((int)rest) + 1u + k;
    }
}

#define CLX_ERR_NONE 0u
#define CLX_MKERR(status, msg) (((uint32_t)(status) << 16) | (uint32_t)(msg))

// Decode `count` Rice codes with parameter k starting at bit `pos` into dst[0..count).
// Returns the bit position after the last code; *err != 0 on EOF.  Wave-uniform control flow.
//
// Per span of 64 lanes x B bits: (1) every lane finds the state in which its chunk is left for every state it may be
// entered in -- four bits per step, all states at once, through the transition table in LDS for short codes
// (k <= CLX_LUT_KMAX), a walk of shift / count-leading-zeros / add per code for longer ones; (2) every lane's true entry state: chunks whose exit does
// not depend on their entry hand it to their successor directly, the rest follow by DPP wave shifts (table path), or a
// three-level walk over the 64 chunks' exit tables in LDS (8 groups of 8; long codes); (3) lanes mark the codes that
// *start* in their chunk, a wave prefix sum gives output indices; (4) the start positions are listed in LDS: a code
// ends where the next one starts, so extraction is balanced over the lanes (code i by lane i mod 64), reads the
// remainder with one v_alignbit + v_bfe from the window and writes the residuals to HBM coalesced.
__device__ uint32_t clx_rice_partition(K1Lds& L, BitSrc& b, uint32_t pos, uint32_t k, uint32_t count,
                                       int32_t* dst, uint32_t limit, uint32_t* err, uint32_t& lut_k, int lane CLX_TL_PH_PARAM) {
    // the cursor and the partition's shape are the same in every lane: keep them (and all that follows from them: chunk
    // width, window tests, loop control) on the scalar unit
    pos = clx_uniform(pos); k = clx_uniform(k); count = clx_uniform(count); limit = clx_uniform(limit);
    const uint32_t SC = k + 1u, ns = k + 2u;
    CLX_TL_PHASE(5);                       // everything outside the residual decode: headers, warm-up, descriptors
    // Transition table of the code's bit automaton, four bits at a time: a state m in [1,k] just counts a remainder
    // bit down; at a code start (0) or inside a run (SC) a one ends the run and leaves k remainder bits, a zero
    // continues it.  A table row holds what a nibble does to EVERY state (K1Lds::lut), so the exit states of a chunk for all
    // entry states are B/4 row fetches that depend on nothing but the chunk, and one v_bfe_u32 per state and step -- instead
    // of a walk of shift / count-leading-zeros / add steps per code and state.
    const bool use_lut = k <= CLX_LUT_KMAX;                  // wave-uniform.  Longer codes: few per chunk, the walks are short
    if (use_lut && lut_k != k) {
        if (lane < 32) reinterpret_cast<uint32_t*>(L.lut)[lane] = clx_lut_rom.w[k][lane];
        lut_k = k;
        __syncthreads();
    }
    uint32_t done = 0;
    while (done < count) {
        const uint32_t remaining = count - done;
        // chunk width: sized so that one span of 64 chunks usually covers the whole partition (a code with an
        // optimal parameter averages ~k+2.2 bits; k+3 leaves headroom) and every lane has work
        uint32_t B = (remaining * (k + 3u) + 63u) >> 6;
        if (use_lut) B = (B + 3u) & ~3u;                    // whole nibbles (the transition table's step)
        B = B < 4u ? 4u : B > 32u ? 32u : B;
        if (k == 0u && B > 16u) B = 16u;                    // at most 1024 <= CLX_NPOS codes per span
        clx_window_ensure(L, b, pos, 64u * B + 64u, lane);

        const uint32_t cpos = pos + B * (uint32_t)lane;
        // my chunk and the 32 bits after it (codes that start in my chunk are extracted from this 64-bit register window):
        // three dwords of the LDS window, which clx_window_ensure has just made sure of
        uint32_t c_raw, c_next;
        {
            const uint32_t wi = (cpos >> 5) - b.win_dw, off = cpos & 31u;
            const uint32_t w0 = L.W[wi], w1 = L.W[wi + 1u], w2 = L.W[wi + 2u];
            c_raw = (uint32_t)(((((uint64_t)w0 << 32) | w1) << off) >> 32);
            c_next = (uint32_t)(((((uint64_t)w1 << 32) | w2) << off) >> 32);
        }
        uint32_t c = c_raw;
        if (B < 32u) c &= ~(0xffffffffu >> B);

        CLX_TL_PHASE(0);                   // span set-up, window
        // (1) exit-state tables
        const bool sent = B < 32u;                           // wave-uniform
        const uint32_t cs = sent ? (c | (0x80000000u >> B)) : c;
        uint32_t my_entry;                                   // (table path: times five)
        if (use_lut) {
            // Five walks at once, one per entry state 0 .. 4 (states are carried times five, see K1Lds::lut): a step is the table
            // row of the next nibble and one v_bfe_u32 per walk.  States above k + 1 do not exist: their walks run through zero
            // fields and are never looked at.  The steps are unrolled -- this kernel is bound by the number of instructions its
            // waves issue, scalar loop control included (measured: eight extra s_add per step cost more than eight extra v_add).
            uint32_t e0 = 0u, e1 = 5u, e2 = 10u, e3 = 15u, e4 = 20u;
#define CLX_K1_STEP5(i) { const uint32_t F = L.lut[(c >> (28u - 4u * (i))) & 15u].f; \
                          e0 = clx_bfe(F, e0, 5u); e1 = clx_bfe(F, e1, 5u); e2 = clx_bfe(F, e2, 5u); e3 = clx_bfe(F, e3, 5u); e4 = clx_bfe(F, e4, 5u); }
            const uint32_t nb = B >> 2;                      // 1 .. 8 (wave-uniform): straight-line steps, a scalar test in front of each
            CLX_K1_STEP5(0)
            if (nb > 1u) { CLX_K1_STEP5(1)
            if (nb > 2u) { CLX_K1_STEP5(2)
            if (nb > 3u) { CLX_K1_STEP5(3)
            if (nb > 4u) { CLX_K1_STEP5(4)
            if (nb > 5u) { CLX_K1_STEP5(5)
            if (nb > 6u) { CLX_K1_STEP5(6)
            if (nb > 7u) { CLX_K1_STEP5(7) } } } } } } }
#undef CLX_K1_STEP5
            // the chunk's exit states in one register, field s5 = exit for entry state s (the field of SC = k + 1 holds state 0's:
            // entering inside a run walks exactly like entering at a code start -- its own walk said so for k <= 3, for k = 4 it
            // is put there)
            const uint32_t X = (e0 * 0x02000001u) | (e1 << 5) | (e2 << 10) | (e3 << 15) | (e4 << 20);
            CLX_TL_PHASE(1);               // exit tables
            // (2) entry states.  Rice codes resynchronise within a few codes, so for most chunks every entry state leads to
            // the same exit state: such a chunk tells its successor where it starts without knowing its own entry state.
            // What is known is handed to the next lane (DPP wave shift) until every lane knows its entry state: one round
            // plus one per chunk of the longest run of chunks whose exit does depend on their entry -- no LDS traffic.
            const uint32_t fields = 0xffffffffu >> (22u - 5u * k);                   // the fields of states 0 .. k + 1 (wave-uniform)
            const bool same = (X & fields) == e0 * (0x02108421u & fields);
            uint32_t ent = (lane == 0) ? 0u : 0xffu;         // a span always begins at a code start; 0xff: not known yet
            uint32_t outv = same ? e0 : 0xffu;
            CLX_STAT(56, lane == 0);                         // (simulator statistics: spans on the table path, rounds below)
            for (;;) {
                CLX_STAT(57, lane == 0);
                const uint32_t cand = clx_bfe(X, ent, 5u);   // (offset taken mod 32: garbage while ent is unknown, not used then)
                if (outv == 0xffu && ent != 0xffu) outv = cand;
                const uint32_t pv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)outv, 0x138, 0xF, 0xF, true);   // wave_shr:1
                if (ent == 0xffu) ent = pv;                  // (lane 0 never takes this)
                if (__all(ent != 0xffu)) break;
            }
            my_entry = ent;
        } else {
        CLX_STAT(58, lane == 0);                             // spans on the arithmetic-walk path
        const uint32_t ex0 = sent ? clx_chunk_exit<true>(cs, B, k, 0u) : clx_chunk_exit<false>(c, B, k, 0u);
        for (uint32_t g = 0; 4u * g < ns; ++g) {
            uint32_t packed = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t m = 4u * g + j;
                // entering inside a run (SC) walks exactly like entering at a code start (state 0)
                const uint32_t ex = (m == 0u || m == SC) ? ex0 : (m >= ns) ? 0u
                                  : sent ? clx_chunk_exit<true>(cs, B, k, m) : clx_chunk_exit<false>(c, B, k, m);
                packed |= ex << (8u * j);
            }
            *reinterpret_cast<uint32_t*>(&L.u.t.tab[lane][4u * g]) = packed;     // states 4g .. 4g+3 (little endian)
        }
        __syncthreads();
        CLX_TL_PHASE(1);                   // exit tables
        // (2a) group tables: lane (g8, e) walks group g8's 8 chunks for entry states e, e+8, ...
        {
            const uint32_t g8 = (uint32_t)lane >> 3;
            for (uint32_t st = (uint32_t)lane & 7u; st < ns; st += 8u) {
                uint32_t m = st;
#pragma unroll
                for (uint32_t i = 0; i < 8; ++i) m = L.u.t.tab[g8 * 8u + i][m];
                L.u.t.gtab[g8][st] = (uint8_t)m;
            }
        }
        __syncthreads();
        // (2b) across groups, (2c) inside each group.  Both walks are chains of LDS look-ups whose address is the previous result; the state in front of every step is
        // dropped into LDS (no VALU work) and each lane picks up the one that is its own afterwards.
        {
            uint32_t m = 0;                                  // a span always begins at a code start
#pragma unroll
            for (uint32_t g = 0; g < 8; ++g) { L.gent[g] = (uint8_t)m; m = L.u.t.gtab[g][m]; }
            const uint32_t g8 = (uint32_t)lane >> 3;
            m = L.gent[g8];
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i) {
                L.ent[g8 * 8u + i] = (uint8_t)m;
                if (i < 7u) m = L.u.t.tab[g8 * 8u + i][m];
            }
            my_entry = L.ent[lane];
        }
        }
        CLX_TL_PHASE(2);                   // entry states
        // (3) starts in my chunk
        uint32_t S = 0;
        if (use_lut) {
            // the same table, walked once from the true entry state: every step hands over the four start flags of its bits
            uint32_t st = my_entry;
#define CLX_K1_STEP1(i) { const K1Lds::Row row = L.lut[(c >> (28u - 4u * (i))) & 15u]; \
                          S = clx_alignbit(row.g >> st, S, 4u); st = clx_bfe(row.f, st, 5u); }     // S = (S >> 4) | (flags << 28)
            const uint32_t nb = B >> 2;
            CLX_K1_STEP1(0)
            if (nb > 1u) { CLX_K1_STEP1(1)
            if (nb > 2u) { CLX_K1_STEP1(2)
            if (nb > 3u) { CLX_K1_STEP1(3)
            if (nb > 4u) { CLX_K1_STEP1(4)
            if (nb > 5u) { CLX_K1_STEP1(5)
            if (nb > 6u) { CLX_K1_STEP1(6)
            if (nb > 7u) { CLX_K1_STEP1(7) } } } } } } }
#undef CLX_K1_STEP1
            S >>= (32u - B) & 31u;
        } else {
            uint32_t p = (my_entry == SC) ? 0u : my_entry;
            if (sent) {
                const uint32_t k1 = k + 1u;
                while (p < B) { S |= 0x1u << p; p += (uint32_t)__clz((int)(cs << p)) + k1; }
                if (my_entry == SC) S &= ~1u;                // position 0 continued a run: it is not a start
            } else {
                bool fresh = (my_entry != SC);
                while (p < B) {
                    if (fresh) S |= 1u << p;
                    fresh = true;
                    uint32_t rest = c << p;
                    if (rest == 0u) break;
                    p += (uint32_t)__clz((int)rest) + 1u + k;
                }
            }
        }
        const uint32_t cnt = (uint32_t)__popc(S);
        uint32_t total;
        const uint32_t prefix = clx_wave_excl_scan(cnt, lane, &total);
        const uint32_t ntake = total < remaining ? total : remaining;

        CLX_TL_PHASE(3);                   // starts + prefix sum
        // (4a) list the span-relative start position of every code (ascending inside a lane, lanes in order)
        {
            uint32_t Sit = S, idx = prefix;
            while (Sit != 0u) {
                const uint32_t p = (uint32_t)__ffs((int)Sit) - 1u;
                Sit &= Sit - 1u;
                L.u.P[idx] = (uint16_t)(B * (uint32_t)lane + p);
                ++idx;
            }
        }
        __syncthreads();
        // (4b) where the taken codes end.  A code ends where the next one starts; only the span's very last code has to
        // be delimited the slow way (zeros, a one, k bits), by the one lane that holds its start.
        uint32_t newpos;
        if (total <= remaining) {
            const bool owner = cnt != 0u && prefix + cnt == total;
            uint32_t e = 0;
            if (owner) {
                const uint32_t p = 31u - (uint32_t)__clz((int)S);
                const uint32_t s = cpos + p;
                uint32_t v = p ? clx_alignbit(c_raw, c_next, 32u - p) : c_raw;                // 32 bits at the code's start
                uint32_t t = s;
                if (v == 0u) {                               // long unary run (subframe.rs:326-328: rare)
                    t = s + 32u;
                    while (t < limit) { v = clx_peek32(L, b, t); if (v != 0u) break; t += 32u; }
                }
                e = (v == 0u) ? limit + 1u : t + (uint32_t)__clz((int)v) + 1u + k;
            }
            newpos = clx_pick(e, owner, limit + 1u);
        } else newpos = pos + clx_uniform((uint32_t)L.u.P[remaining]);      // start of the first code that is not taken
        CLX_TL_PHASE(4);                   // position list + the span's last code
        // (4c) extraction, balanced over the lanes and written straight to HBM (coalesced): code i has
        // q = start(i+1) - start(i) - 1 - k zeros and its k remainder bits end where code i+1 starts
        const uint32_t end_rel = newpos - pos;
        const bool far = end_rel > 64u * B + 32u;            // the span's last code runs past what the window is known to hold
        // the list ends with the end of the last taken code (already there when codes were left over: the first of those
        // starts at it); every lane writes the same value and reads only after its own write
        if (total <= remaining) L.u.P[ntake] = (uint16_t)end_rel;
        if (!far) {
            // every remainder lies inside the window: its k bits are the low bits of the 64-bit pair (dword in front of the
            // one holding its last bit, that dword) shifted right until that last bit is bit 0 -- one v_alignbit, one v_bfe
            const uint32_t bias = pos - 1u - 32u * b.win_dw;                                   // wave-uniform
            for (uint32_t i = (uint32_t)lane; i < ntake; i += 64u) {
                const uint32_t s = L.u.P[i];
                const uint32_t e = L.u.P[i + 1u];
                const uint32_t q = e - s - 1u - k;
                const uint32_t last = e + bias;              // window-relative position of the code's last bit
                const uint32_t* w = &L.W[last >> 5];
                const uint32_t r = clx_bfe(clx_alignbit(w[-1], w[0], ~last), 0u, k);         // k = 0: no bits, r = 0
                const uint32_t u = (q << k) | r;             // u32 wrapping shift, subframe.rs:340
                dst[done + i] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1u);                       // rice_to_signed, subframe.rs:157-170
            }
        } else
        for (uint32_t i = (uint32_t)lane; i < ntake; i += 64u) {
            const uint32_t s = L.u.P[i];
            const uint32_t e = (i + 1u < ntake) ? (uint32_t)L.u.P[i + 1u] : end_rel;
            const uint32_t q = e - s - 1u - k;
            uint32_t r = 0;
            if (k != 0u) {
                const uint32_t at = pos + e - k;
                const uint32_t v = (i + 1u == ntake) ? clx_peek32(L, b, at) : clx_peek32_win(L, b, at);
                r = v >> (32u - k);
            }
            const uint32_t u = (q << k) | r;                 // u32 wrapping shift, subframe.rs:340
            dst[done + i] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1u);                           // rice_to_signed, subframe.rs:157-170
        }
        // ends grow with the code index: the last taken code is past the limit iff any is (input.rs EOF)
        if (newpos > limit) { *err = CLX_MKERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF); return newpos; }
        done += ntake;
        pos = newpos;
        __syncthreads();
        CLX_TL_PHASE(6);                   // extraction + stores
    }
    return pos;
}

// Sequential header reader: mirrors the reference's sequence of Bitstream calls one for one, so that
// the FIRST error in stream order wins (EOF vs format error), exactly as `try!` does.
struct HdrReader {
    uint32_t pos, limit, err;
};
__device__ __forceinline__ bool clx_hdr_bits(const K1Lds& L, const BitSrc& b, HdrReader& h, uint32_t n, uint32_t* v) {
    if (h.pos + n > h.limit) { h.err = CLX_MKERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF); return false; }
    *v = clx_peek_bits(L, b, h.pos, n);
    h.pos += n;
    return true;
}

extern "C" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7, 8)))
void clx_k_residual(const uint8_t* __restrict__ arena, uint64_t arena_alloc_len,
                    const clx_dev_frame* __restrict__ frames, uint32_t n_frames,
                    int32_t* __restrict__ out, clx_sf_desc* __restrict__ sfd,
                    clx_frame_result* __restrict__ results) {
    __shared__ K1Lds L;
    const int lane = (int)threadIdx.x;
    const uint32_t f = blockIdx.x;
    if (f >= n_frames) return;
    CLX_TL_BEGIN();
    const clx_dev_frame fr = frames[f];

    BitSrc b;
    const uint64_t origin_byte = fr.byte_off & ~15ull;
    b.origin = reinterpret_cast<const uint32_t*>(arena + origin_byte);
    {
        uint64_t a = (arena_alloc_len > origin_byte) ? ((arena_alloc_len - origin_byte) >> 2) : 0ull;
        b.avail_dw = a > 0xfffffff0ull ? 0xfffffff0u : (uint32_t)a;
    }
    b.win_dw = 0;
    const uint32_t o = 8u * (uint32_t)(fr.byte_off & 15ull);          // bit offset of the frame inside its granule
    HdrReader h;
    h.limit = o + fr.limit_bits;
    h.pos = o + 8u * (uint32_t)fr.header_bytes;
    h.err = CLX_ERR_NONE;
    if (h.pos > h.limit) h.err = CLX_MKERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
    if (lane < 32) reinterpret_cast<uint32_t*>(L.lut)[lane] = 0u;
    uint32_t lut_k = 0xffffffffu;                                      // the Rice parameter the table is built for: none yet
    clx_window_load(L, b, h.pos < h.limit ? h.pos : o, lane);

    const uint32_t bs = fr.block_size;
    const uint32_t ca = fr.channel_assignment;

    uint32_t n_handed = 0;                   // channels whose descriptor has been handed to K2
    for (uint32_t ch = 0; ch < fr.n_channels && h.err == CLX_ERR_NONE; ++ch) {
        // side channels carry one extra bit (frame.rs:713-741)
        uint32_t bps = fr.bps;
        if ((ca == CLX_CH_LEFT_SIDE && ch == 1) || (ca == CLX_CH_RIGHT_SIDE && ch == 0) ||
            (ca == CLX_CH_MID_SIDE && ch == 1)) bps += 1u;
        int32_t* const chan = out + fr.out_off + (uint64_t)ch * bs;
        clx_window_ensure(L, b, h.pos, 128u, lane);

        // ---- read_subframe_header, subframe.rs:29-91
        uint32_t v;
        if (!clx_hdr_bits(L, b, h, 1, &v)) break;
        if (v) { h.err = CLX_MKERR(CLX_FORMAT_ERROR, CLX_MSG_SUBFRAME_HEADER_INVALID); break; }
        if (!clx_hdr_bits(L, b, h, 6, &v)) break;
        uint32_t kind, order = 0;               // kind: 0 constant, 1 verbatim, 2 fixed, 3 lpc
        if (v == 0u) kind = 0;
        else if (v == 1u) kind = 1;
        else if ((v & 0x3eu) == 0x02u || (v & 0x3cu) == 0x04u || (v & 0x30u) == 0x10u) {
            h.err = CLX_MKERR(CLX_FORMAT_ERROR, CLX_MSG_SUBFRAME_HEADER_RESERVED); break;
        } else if ((v & 0x38u) == 0x08u) {
            order = v & 7u;
            if (order > 4u) { h.err = CLX_MKERR(CLX_FORMAT_ERROR, CLX_MSG_SUBFRAME_HEADER_RESERVED); break; }
            kind = 2;
        } else { kind = 3; order = (v & 0x1fu) + 1u; }
        if (!clx_hdr_bits(L, b, h, 1, &v)) break;
        uint32_t wasted = 0;
        if (v) {                                 // 1 + read_unary (subframe.rs:73-77)
            uint32_t t = h.pos; uint32_t w = 0; bool found = false;
            while (t < h.limit) {
                w = clx_peek32(L, b, t);
                if (w != 0u) { found = true; break; }
                t += 32u;
            }
            if (found) { t += (uint32_t)__clz((int)w); if (t >= h.limit) found = false; }
            if (!found) { h.err = CLX_MKERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF); break; }
            wasted = 1u + (t - h.pos);
            h.pos = t + 1u;
            if (wasted > 31u) { h.err = CLX_MKERR(CLX_FORMAT_ERROR, CLX_MSG_WASTED_BITS_EXCEED_31); break; }
        }
        // ---- subframe::decode, subframe.rs:198-211
        if (wasted >= bps) { h.err = CLX_MKERR(CLX_FORMAT_ERROR, CLX_MSG_NO_NON_WASTED_BITS); break; }
        const uint32_t sf_bps = bps - wasted;

        uint32_t qshift = 0;
        int32_t my_coef = 0;                     // lane j holds coef[j] (j < order)

        if (kind == 0u) {                        // decode_constant, subframe.rs:382-394
            if (h.pos + sf_bps > h.limit) { h.err = CLX_MKERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF); break; }
            const int32_t s = clx_peek_signed(L, b, h.pos, sf_bps);
            h.pos += sf_bps;
            for (uint32_t i = (uint32_t)lane; i < bs; i += 64u) chan[i] = s;
        } else if (kind == 1u) {                 // decode_verbatim, subframe.rs:397-415
            if ((uint64_t)h.pos + (uint64_t)bs * sf_bps > (uint64_t)h.limit) {
                h.err = CLX_MKERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF); break;
            }
            for (uint32_t i0 = 0; i0 < bs; i0 += 64u) {
                clx_window_ensure(L, b, h.pos + i0 * sf_bps, 64u * sf_bps + 32u, lane);
                const uint32_t i = i0 + (uint32_t)lane;
                if (i < bs) chan[i] = clx_peek_signed(L, b, h.pos + i * sf_bps, sf_bps);
            }
            h.pos += bs * sf_bps;
        } else {
            // decode_fixed (subframe.rs:492-516) / decode_lpc (subframe.rs:651-721)
            if (bs < order) {
                h.err = CLX_MKERR(CLX_FORMAT_ERROR, kind == 2u ? CLX_MSG_FIXED_ORDER_GT_BLOCK : CLX_MSG_LPC_ORDER_GT_BLOCK);
                break;
            }
            // warm-up samples: `order` verbatim fields, one per lane
            if (h.pos + order * sf_bps > h.limit) { h.err = CLX_MKERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF); break; }
            clx_window_ensure(L, b, h.pos, order * sf_bps + 32u * 15u + 64u, lane);
            if ((uint32_t)lane < order) chan[lane] = clx_peek_signed(L, b, h.pos + (uint32_t)lane * sf_bps, sf_bps);
            h.pos += order * sf_bps;
            if (kind == 3u) {
                if (!clx_hdr_bits(L, b, h, 4, &v)) break;
                const uint32_t precision = v + 1u;
                if (v == 15u) { h.err = CLX_MKERR(CLX_FORMAT_ERROR, CLX_MSG_QLP_PRECISION_INVALID); break; }
                if (!clx_hdr_bits(L, b, h, 5, &v)) break;
                if (v & 0x10u) { h.err = CLX_MKERR(CLX_UNSUPPORTED, CLX_MSG_NEGATIVE_QLP_SHIFT); break; }
                qshift = v;
                if (h.pos + order * precision > h.limit) { h.err = CLX_MKERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF); break; }
                // j-th coded coefficient applies to s[i-1-j] (stored reversed in the reference, subframe.rs:696-701)
                if ((uint32_t)lane < order) my_coef = clx_peek_signed(L, b, h.pos + (uint32_t)lane * precision, precision);
                h.pos += order * precision;
            } else {
                // fixed predictors as IIR taps on s[i-1-j] (subframe.rs:427-431, oldest-to-newest there)
                my_coef = (order == 1u) ? (lane == 0 ? 1 : 0)
                        : (order == 2u) ? (lane == 0 ? 2 : lane == 1 ? -1 : 0)
                        : (order == 3u) ? (lane == 0 ? 3 : lane == 1 ? -3 : lane == 2 ? 1 : 0)
                        : (order == 4u) ? (lane == 0 ? 4 : lane == 1 ? -6 : lane == 2 ? 4 : lane == 3 ? -1 : 0) : 0;
            }
            // ---- decode_residual, subframe.rs:236-304
            if (!clx_hdr_bits(L, b, h, 2, &v)) break;
            if (v > 1u) { h.err = CLX_MKERR(CLX_FORMAT_ERROR, CLX_MSG_RESIDUAL_RESERVED); break; }
            const uint32_t rice2 = v;
            if (!clx_hdr_bits(L, b, h, 4, &v)) break;
            const uint32_t porder = v;
            const uint32_t n_part = 1u << porder;
            const uint32_t per = bs >> porder;
            if ((bs & (n_part - 1u)) != 0u) { h.err = CLX_MKERR(CLX_FORMAT_ERROR, CLX_MSG_INVALID_PARTITION_ORDER); break; }
            if (order > per) { h.err = CLX_MKERR(CLX_FORMAT_ERROR, CLX_MSG_INVALID_RESIDUAL); break; }
            uint32_t start = order;
            uint32_t len = per - order;
            for (uint32_t part = 0; part < n_part; ++part) {
                clx_window_ensure(L, b, h.pos, 64u, lane);
                if (!clx_hdr_bits(L, b, h, rice2 ? 5u : 4u, &v)) break;
                if (v == (rice2 ? 31u : 15u)) { h.err = CLX_MKERR(CLX_UNSUPPORTED, CLX_MSG_UNENCODED_BINARY); break; }
                uint32_t perr = CLX_ERR_NONE;
                h.pos = clx_rice_partition(L, b, h.pos, v, len, chan + start, h.limit, &perr, lut_k, lane CLX_TL_PH_ARG);
                if (perr != CLX_ERR_NONE) { h.err = perr; break; }
                start += len;
                len = per;
            }
            if (h.err != CLX_ERR_NONE) break;
        }
        // ---- hand the predictor to K2
        {
            const uint32_t slot = fr.first_slot + ch;
            clx_sf_desc* d = &sfd[slot];
            if ((uint32_t)lane < 32u) d->coef[lane] = (int16_t)((uint32_t)lane < order ? my_coef : 0);
            // sum|c| over the taps: K2 may evaluate the recurrence in 32 bits with 24-bit factors while the history stays
            // inside [-lim, lim), lim <= 2^23 and sum|c| * lim < 2^31 (exact for every in-range history; K2 checks the data)
            uint32_t cabs = (uint32_t)(my_coef < 0 ? -my_coef : my_coef);
            if ((uint32_t)lane >= order || kind < 2u) cabs = 0;
            uint32_t csum;
            (void)clx_wave_excl_scan(cabs, lane, &csum);             // (its total: a DPP reduction, no LDS round trips)
            if (lane == 0) {
                // the largest power of two that keeps sum|c| * lim < 2^31 (and 24-bit factors): K2 checks the data against it
                const uint32_t by_sum = csum != 0u ? 0x7fffffffu / csum : 0x7fffffffu;
                const uint32_t ll = 31u - (uint32_t)__clz((int)(by_sum | 1u));
                // the 16 bytes in front of the coefficients as ONE store (clx_sf_desc's layout: static_asserts in clx_device.h)
                const uint64_t base = fr.out_off + (uint64_t)ch * bs;
                uint4 hdr16;
                hdr16.x = (uint32_t)base; hdr16.y = (uint32_t)(base >> 32);
                hdr16.z = (bs & 0xffffu) | ((ll < 23u ? ll : 23u) << 16) | ((fr.bps <= 16u ? CLX_SF_NARROW : 0u) << 24);   // n, lim_log2, flags
                hdr16.w = ((kind >= 2u) ? order : 0u) | (qshift << 8) | (wasted << 16) | (ca << 24);       // order, shift, wasted, decor
                *reinterpret_cast<uint4*>(d) = hdr16;
            }
            n_handed = ch + 1u;
        }
    }
    // the slots of subframes that were never reached (the frame failed before them) are empty for K2: every slot of the frame
    // is written by this wave on every run, so the descriptors need no clearing between runs
    if ((uint32_t)lane < fr.n_channels - n_handed) sfd[fr.first_slot + n_handed + (uint32_t)lane].n = 0;
    // the footer is read whether or not it is compared (frame.rs:754; under cfg(fuzzing) only the comparison goes away)
    if (!h.err && !(fr.flags & 1u) && (uint64_t)(((h.pos - o) + 7u) & ~7u) + 16u > (uint64_t)fr.limit_bits)
        h.err = CLX_MKERR(CLX_IO_ERROR, CLX_MSG_UNEXPECTED_EOF);
    if (lane == 0) {                         // status, msg, end_bit (64 bits, upper half 0) as one 16-byte store
        *reinterpret_cast<uint4*>(&results[f]) = make_uint4(h.err >> 16, h.err & 0xffffu, h.pos - o, 0u);
    }
    CLX_TL_PHASE(5);
    CLX_TL_END(0, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// K2: predictor synthesis + wasted-bits shift + stereo decorrelation.  One lane per subframe
// ("predictor slot"), 64 slots per pair of waves, BLK = 16 samples per turn: a tile of 64 rows x 64 bytes
// travels HBM -> LDS (DMA) -> predictor wave -> LDS -> finisher wave -> HBM (the memory side and the
// two-wave schedule are described further down, above clx_predict_wave).  The recurrence
//     s[i] = x[i] + ((sum_j c[j]*s[i-1-j]) >> shift)          subframe.rs:559-566, 575-582, 606-613
// needs an i64 accumulator in general.  When K1 proved  sum|c| * 2^(sf_bps-1) < 2^31  and
// sf_bps <= 24  (clx_sf_desc::lim_log2), the sum of a VALID stream fits i32 and every factor fits
// 24 bits, so the block is first run with v_mad_i32_i24 (full rate) and its outputs are range-checked;
// a block in which any lane leaves the proven range (corrupt streams only) is re-run with
// v_mad_i64_i32, which is exact for any input -- garbage in, the reference's garbage out.
// The stereo partner sits in lane^1 (host aligns decorrelated pairs to even slots) and enters the
// finisher's arithmetic as a DPP operand; decorrelation (frame.rs:319-389) is off the recurrence's
// critical path, in a different wave.
// ------------------------------------------------------------------------------------------------
#define CLX_BLK 16

struct K2NoHook { __device__ __forceinline__ void operator()(int) const {} };

// `hook(i)` runs after sample i: the caller's memory instructions are issued in between the arithmetic instead of in
// one burst (a burst of scattered 64-lane VMEM instructions stalls the wave's issue for hundreds of cycles).
template <int OMAX, bool WIDE, bool MASKED, typename Hook>
__device__ __forceinline__ void clx_iir_block(const int32_t (&x)[CLX_BLK], int32_t (&y)[CLX_BLK], int32_t (&hist)[OMAX],
                                              const int32_t (&c)[OMAX], uint32_t t0, uint32_t order, uint32_t shift, Hook&& hook) {
#pragma unroll
    for (int i = 0; i < CLX_BLK; ++i) {
        int32_t pred;
        if (WIDE) {
            int64_t acc = 0;
#pragma unroll
            for (int j = OMAX - 1; j >= 0; --j) acc += (int64_t)c[j] * (int64_t)hist[j];   // newest tap last: shortest dependent chain
            pred = (int32_t)(acc >> shift);
        } else {
            const int32_t acc = clx_dot24z<OMAX>(c, hist);                               // v_mad_i32_i24 chain, newest tap last
            pred = acc >> shift;
        }
        int32_t s;
        if (MASKED) {   // warm-up samples (i < order) pass through; branch-free so the block stays one straight line of code
            const uint32_t use = (t0 + (uint32_t)i >= order) ? 0xffffffffu : 0u;
            s = (int32_t)((uint32_t)x[i] + ((uint32_t)pred & use));
        } else s = (int32_t)((uint32_t)x[i] + (uint32_t)pred);
#pragma unroll
        for (int j = OMAX - 1; j > 0; --j) hist[j] = hist[j - 1];
        hist[0] = s;
        y[i] = s;
        hook(i);
    }
}

template <bool ALIGNED>
__device__ __forceinline__ void clx_row_load(const int32_t* __restrict__ row, uint32_t t, uint32_t n, int32_t (&v)[CLX_BLK]) {
    // Unconditional loads from clamped indices (no control flow => the compiler can count them and keep the
    // prefetch in flight); samples past the row's end are never stored, so what they hold does not matter.
    if (ALIGNED) {
        const uint32_t last = n >= 4u ? n - 4u : 0u;
#pragma unroll
        for (int q = 0; q < CLX_BLK / 4; ++q) {
            const uint32_t idx = t + 4u * q < last ? t + 4u * q : last;
            const int4 w = *reinterpret_cast<const int4*>(row + idx);
            v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
        }
    } else {
        const uint32_t last = n >= 1u ? n - 1u : 0u;
#pragma unroll
        for (int i = 0; i < CLX_BLK; ++i) { const uint32_t idx = t + (uint32_t)i < last ? t + (uint32_t)i : last; v[i] = row[idx]; }
    }
}
// Unconditional stores: samples past the row's end go to a per-lane dump area instead of being branched around
// (any branch around a memory operation makes the compiler fall back to s_waitcnt vmcnt(0), which ties every use
// of prefetched data to the completion of all older stores).
template <bool ALIGNED>
__device__ __forceinline__ void clx_row_store(int32_t* __restrict__ row, int32_t* __restrict__ dump, uint32_t t, uint32_t n,
                                              const int32_t (&v)[CLX_BLK]) {
    if (ALIGNED) {
#pragma unroll
        for (int q = 0; q < CLX_BLK / 4; ++q) {
            int32_t* p = (t + 4u * q < n) ? row + t + 4 * q : dump + 4 * q;
            *reinterpret_cast<int4*>(p) = make_int4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < CLX_BLK; ++i) { int32_t* p = (t + (uint32_t)i < n) ? row + t + i : dump + i; *p = v[i]; }
    }
}

// What one lane knows about its predictor slot (a row of `out`).
struct K2Slot {
    const clx_sf_desc* d;
    int32_t* row;              // first sample of the row; empty slots point at out[0] (read, never written)
    uint32_t n, order, shift, wasted, decor, lim_log2;
    bool pair_ok;
};

// The recurrence of one row, a block of CLX_BLK samples at a time: 24-bit fast evaluation with a range check, exact i64
// re-run of a block that leaves the proven range (see the header comment of this section).
template <int OMAX>
__device__ __forceinline__ void clx_dot2_block_any(const int32_t (&x)[CLX_BLK], int32_t (&y)[CLX_BLK], int32_t (&pr)[OMAX > 1 ? OMAX - 1 : 1],
                                                   const int32_t (&C)[OMAX / 2], uint32_t shift, int32_t prev) {
    if constexpr (OMAX <= 12) clx_dot2_block<OMAX>(x, y, pr, C, shift, prev);
}

// Three evaluations of the same recurrence, fastest first; each is exact on the inputs it accepts and a block that turns out
// to leave an evaluation's range is re-run with the next one from the saved history:
//   16-bit : history as packed pairs, two taps per v_dot2_i32_i16 (clx_k2_dot2.h): every sample inside [-lim16, lim16),
//            lim16 = min(lim, 2^15) -- 16-bit audio, the common case (mid is 16 bits wide; side is 17 but small)
//   24-bit : v_mad_i32_i24 chain: every sample inside [-lim, lim), lim <= 2^23 with sum|c| * lim < 2^31 (from K1)
//   exact  : v_mad_i64_i32, any input -- garbage in, the reference's garbage out
// LEAN: without the 24-bit evaluation (16-bit, else exact): what the kernel for 16-bit audio uses -- fewer values live at once.
template <int OMAX, bool LEAN = false>
struct K2Predictor {
    static constexpr bool HAS16 = OMAX <= 12;
    int32_t c[OMAX], hist[OMAX];
    int32_t C2[OMAX / 2];      // (c[2p] << 16) | (c[2p+1] & 0xffff)
    int32_t lim, lim16;
    uint32_t n, order, shift;
    bool trivial, h_ok, h16_ok;
    __device__ __forceinline__ void init(const K2Slot& S) {
        n = S.n; order = S.order; shift = S.shift;
#pragma unroll
        for (int j = 0; j < OMAX; ++j) { c[j] = (n != 0u && (uint32_t)j < order) ? (int32_t)S.d->coef[j] : 0; hist[j] = 0; }
#pragma unroll
        for (int p = 0; p < OMAX / 2; ++p) C2[p] = (int32_t)(((uint32_t)c[2 * p] << 16) | ((uint32_t)c[2 * p + 1] & 0xffffu));
        // |s| <= lim proves the i32/i24 evaluation exact; lanes K1 could not prove (lim_log2 = 0xff) force the wide path
        // (range is [-lim, lim-1]: the 24-bit signed factor range when lim = 2^23)
        lim = (S.lim_log2 <= 23u) ? (int32_t)(1u << S.lim_log2) : -1;
        lim16 = lim > 32768 ? 32768 : lim;
        trivial = (n == 0u) || (order == 0u);                 // nothing is predicted: any evaluation is exact
        h_ok = (lim >= 0) || trivial;
        h16_ok = h_ok;                                        // (the history starts as zeros)
    }
    // x: residuals / warm-up samples of samples t0 .. t0+15  ->  y: the channel's samples before shift / decorrelation
    // hook(i), i = 0..15, is called exactly once per block, after sample i of the first evaluation
    template <typename Hook>
    __device__ __forceinline__ void block(const int32_t (&x)[CLX_BLK], int32_t (&y)[CLX_BLK], uint32_t t0, Hook&& hook) {
        bool done = false, hooked = false;
        if (HAS16 && t0 >= (uint32_t)OMAX && __all(h16_ok)) {     // (past every lane's warm-up: nothing to mask)
            int32_t pr[OMAX > 1 ? OMAX - 1 : 1];
#pragma unroll
            for (int j = 0; j + 1 < OMAX; ++j) pr[j] = (int32_t)clx_perm((uint32_t)hist[j], (uint32_t)hist[j + 1], 0x05040100u);     // (lo: the older sample, hi: the newer)
            clx_dot2_block_any<OMAX>(x, y, pr, C2, shift, hist[0]);
            int32_t mx = y[0], mn = y[0];
#pragma unroll
            for (int i = 1; i + 1 < CLX_BLK; i += 2) { mx = clx_max3(mx, y[i], y[i + 1]); mn = clx_min3(mn, y[i], y[i + 1]); }
            mx = y[CLX_BLK - 1] > mx ? y[CLX_BLK - 1] : mx; mn = y[CLX_BLK - 1] < mn ? y[CLX_BLK - 1] : mn;
            const bool in16 = trivial || t0 >= n || (mx < lim16 && mn >= -lim16);
            if (__all(in16)) {
#pragma unroll
                for (int j = 0; j < OMAX; ++j) hist[j] = y[CLX_BLK - 1 - j];
#pragma unroll
                for (int i = 0; i < CLX_BLK; ++i) hook(i);
                return;
            }
            h16_ok = in16;             // the lanes that left the range sit out until their history is back inside (below)
        }
        if (!LEAN && __all(h_ok)) {
            hooked = true;
            int32_t h0[OMAX];
#pragma unroll
            for (int j = 0; j < OMAX; ++j) h0[j] = hist[j];
            if (t0 < (uint32_t)OMAX) clx_iir_block<OMAX, false, true>(x, y, hist, c, t0, order, shift, hook);
            else                     clx_iir_block<OMAX, false, false>(x, y, hist, c, t0, order, shift, hook);      // every lane is past its warm-up
            int32_t mx = y[0], mn = y[0];
#pragma unroll
            for (int i = 1; i + 1 < CLX_BLK; i += 2) { mx = clx_max3(mx, y[i], y[i + 1]); mn = clx_min3(mn, y[i], y[i + 1]); }
            mx = y[CLX_BLK - 1] > mx ? y[CLX_BLK - 1] : mx; mn = y[CLX_BLK - 1] < mn ? y[CLX_BLK - 1] : mn;
            const bool in_range = trivial || t0 >= n || (mx < lim && mn >= -lim);
            if (__all(in_range)) {
                done = true;
                if (HAS16) h16_ok = trivial || t0 >= n || (mx < lim16 && mn >= -lim16);
            } else {
#pragma unroll
                for (int j = 0; j < OMAX; ++j) hist[j] = h0[j];
            }
        }
        if (!done) {
            if (hooked) clx_iir_block<OMAX, true, true>(x, y, hist, c, t0, order, shift, K2NoHook());
            else        clx_iir_block<OMAX, true, true>(x, y, hist, c, t0, order, shift, hook);
            bool ok = lim >= 0, ok16 = lim16 >= 0;
#pragma unroll
            for (int j = 0; j < OMAX; ++j) { ok = ok && hist[j] < lim && hist[j] >= -lim; ok16 = ok16 && hist[j] < lim16 && hist[j] >= -lim16; }
            h_ok = ok || trivial || t0 + CLX_BLK >= n;
            h16_ok = ok16 || trivial || t0 + CLX_BLK >= n;
        }
    }
};

// wasted-bits shift (subframe.rs:216-225) and stereo decorrelation (frame.rs:319-389) of a finished block
struct K2Finisher {
    uint32_t wasted, sgn, nsg, rmask, s1, bit, sg;
    bool p_other, r_other, any_decor, all_ms, any_wasted;
    __device__ __forceinline__ void init(const K2Slot& S, int lane) {
        const bool odd = (lane & 1) != 0;
        wasted = S.wasted;
        sgn = odd ? 0xffffffffu : 0u;                          // (x ^ sgn) - sgn = odd ? -x : x
        nsg = odd ? 1u : 0u;
        // per-lane constants of the generic formula  v = ((P << s1 | R & bit) + (R ^ sg) - sg) >> s1
        //   mid/side  even: P = mid (mine),  R = side (other), s1 = 1              frame.rs:382-384
        //             odd : P = mid (other), R = side (mine),  s1 = 1, minus
        //   left/side odd : right = left (other) - side (mine)                     frame.rs:327-330
        //   right/side even: left = side (mine) + right (other)                    frame.rs:352-355
        //   anything else : v = mine
        const bool d_ms = S.pair_ok && S.decor == CLX_CH_MID_SIDE;
        const bool d_ls = S.pair_ok && S.decor == CLX_CH_LEFT_SIDE && odd;
        const bool d_rs = S.pair_ok && S.decor == CLX_CH_RIGHT_SIDE && !odd;
        p_other = (d_ms && odd) || d_ls;
        r_other = (d_ms && !odd) || d_rs;
        rmask = (d_ms || d_ls || d_rs) ? 0xffffffffu : 0u;
        s1 = d_ms ? 1u : 0u; bit = s1;
        sg = ((d_ms && odd) || d_ls) ? 0xffffffffu : 0u;
        any_decor = __any(S.decor != CLX_CH_INDEPENDENT && S.pair_ok);
        // the common case gets a shorter instruction sequence; empty slots (the tail of the last wave) do not spoil it
        all_ms = __all(S.n == 0u || d_ms);
        any_wasted = __any(wasted != 0u);
    }
    // hook(i), i = 0..15, is called exactly once, after sample i is finished.  MODE (wave-uniform, picked once per
    // wave so that no branch surrounds the hook's memory instructions -- hipcc answers a memory operation under a
    // branch with s_waitcnt vmcnt(0), which would drain the prefetch ring): 0 all mid/side, 1 mixed, 2 no decorrelation
    template <int MODE, typename Hook>
    __device__ __forceinline__ void block(int32_t (&y)[CLX_BLK], Hook&& hook) const {
        if (any_wasted) {
#pragma unroll
            for (int i = 0; i < CLX_BLK; ++i) y[i] = (int32_t)((uint32_t)y[i] << wasted);
        }
#pragma unroll
        for (int i = 0; i < CLX_BLK; ++i) {
            if (MODE == 0) {
                // every lane belongs to a mid/side pair: left = (m + side) >> 1, right = (m - side) >> 1 with
                // m = (mid << 1) | (side & 1)  (frame.rs:382-384; m +- side is even)
                y[i] = clx_ms_pair(y[i], sgn, nsg, 1u);
            } else if (MODE == 1) {
                const int32_t mine = y[i];
                const int32_t other = __builtin_amdgcn_update_dpp(0, mine, 0xB1, 0xF, 0xF, false);   // lane ^ 1
                const uint32_t P = (uint32_t)(p_other ? other : mine);
                const uint32_t R = (uint32_t)(r_other ? other : mine) & rmask;
                const uint32_t m = (P << s1) | (R & bit);
                y[i] = (int32_t)(m + ((R ^ sg) - sg)) >> s1;
            }
            hook(i);
        }
    }
    // the same for samples LO .. HI-1 only, without the wasted-bits shift (shift_all does all 16 at once)
    __device__ __forceinline__ void shift_all(int32_t (&y)[CLX_BLK]) const {
        if (any_wasted) {
#pragma unroll
            for (int i = 0; i < CLX_BLK; ++i) y[i] = (int32_t)((uint32_t)y[i] << wasted);
        }
    }
    template <int MODE, int LO, int HI>
    __device__ __forceinline__ void decorrelate(int32_t (&y)[CLX_BLK]) const {
#pragma unroll
        for (int i = LO; i < HI; ++i) {
            if (MODE == 0) y[i] = clx_ms_pair(y[i], sgn, nsg, 1u);
            else if (MODE == 1) {
                const int32_t mine = y[i];
                const int32_t other = __builtin_amdgcn_update_dpp(0, mine, 0xB1, 0xF, 0xF, false);   // lane ^ 1
                const uint32_t P = (uint32_t)(p_other ? other : mine);
                const uint32_t R = (uint32_t)(r_other ? other : mine) & rmask;
                const uint32_t m = (P << s1) | (R & bit);
                y[i] = (int32_t)(m + ((R ^ sg) - sg)) >> s1;
            }
        }
    }
    __device__ __forceinline__ int mode() const { return all_ms ? 0 : any_decor ? 1 : 2; }
    template <typename Hook>
    __device__ __forceinline__ void block(int32_t (&y)[CLX_BLK], Hook&& hook) const {
        if (all_ms) block<0>(y, hook); else if (any_decor) block<1>(y, hook); else block<2>(y, hook);
    }
};

// Rows that are not 16-byte aligned / a multiple of 4 samples long (block sizes that are not a multiple of 4, odd
// sample offsets): one wave, per-lane element accesses, three row buffers rotating through a loop unrolled by three
// (the block computed in turn t was requested in turn t-2).
template <int OMAX>
__device__ __forceinline__ void clx_predict_unaligned(const K2Slot& S, int32_t* __restrict__ dump, uint32_t nmax, int lane) {
    K2Predictor<OMAX> P; P.init(S);
    K2Finisher F; F.init(S, lane);
    int32_t y[CLX_BLK];
    int32_t bufA[CLX_BLK], bufB[CLX_BLK], bufC[CLX_BLK];
    clx_row_load<false>(S.row, 0u, S.n, bufA);
    clx_row_load<false>(S.row, CLX_BLK, S.n, bufB);
    for (uint32_t t0 = 0; t0 < nmax; t0 += 3u * CLX_BLK) {
        clx_row_load<false>(S.row, t0 + 2u * CLX_BLK, S.n, bufC);
        P.block(bufA, y, t0, K2NoHook()); F.block(y, K2NoHook()); clx_row_store<false>(S.row, dump, t0, S.n, y);
        clx_row_load<false>(S.row, t0 + 3u * CLX_BLK, S.n, bufA);
        P.block(bufB, y, t0 + CLX_BLK, K2NoHook()); F.block(y, K2NoHook()); clx_row_store<false>(S.row, dump, t0 + CLX_BLK, S.n, y);
        clx_row_load<false>(S.row, t0 + 4u * CLX_BLK, S.n, bufB);
        P.block(bufC, y, t0 + 2u * CLX_BLK, K2NoHook()); F.block(y, K2NoHook()); clx_row_store<false>(S.row, dump, t0 + 2u * CLX_BLK, S.n, y);
    }
}

// ---- aligned rows: two waves per 64 rows -----------------------------------------------------------------------------
// A lone wave issues one instruction every ~6 shader ticks (tools/ubench/valu_lat.hip), and K2 has only n_slots/64 waves
// for 1024 SIMDs: the kernel's duration is ONE wave's instruction count.  So the per-sample work is split over two waves
// of a workgroup that hand blocks over through LDS:
//   wave 0 "predictor": x from the tile -> recurrence + range check -> y back into the tile          (~12 instr/sample)
//   wave 1 "finisher" : y from the tile -> wasted shift, decorrelation -> tile -> HBM                 (~10 instr/sample)
// Memory side (tools/ubench/storeshape.hip, rowrmw.hip measure the shapes on MI355X): a store instruction whose 64 lanes
// write 16 B to 64 different rows moves 0.8 TB/s, 64 B to 16 rows moves 3.9 TB/s.  So a block (16 samples = 64 B of each
// of the 64 rows = one 4 KiB tile) crosses HBM as four instructions of "4 adjacent lanes = one row's 64 B, 16 rows":
//   in : global_load_lds_dwordx4 (LDS-DMA: no VGPRs, asynchronous, counted by the finisher's vmcnt) into a ring of
//        DEPTH tiles -- with ~2 us of memory latency the ring is what keeps DEPTH x 4 KiB per workgroup in flight
//   out: four ds_read_b128 + global_store_dwordx4 in the same shape; the tile is refilled two turns later
// Tile layout: int4 [row][pos], pos = piece ^ ((row >> 2) & 3): both the lane = row view (64-byte stride) and the
// instruction view (contiguous) are LDS-bank-conflict free.
// Schedule (one workgroup barrier per turn; barrier i ends turn i; a tile is touched by one wave per turn):
//   predictor, turn i: read x(i+1) | recurrence on x(i) | write y(i)
//   finisher,  turn i: read y(i-1) | shift, decorrelate -- and between the samples: store block i-2 (held in registers),
//                      DMA(i-2+DEPTH) into its tile | tile | transposed read of block i-1 | wait until DMA(i+2) landed
// The finisher's memory instructions go out one at a time between the arithmetic: a burst of scattered 64-lane VMEM
// instructions blocks the wave's issue for hundreds of cycles.  Its vmcnt counts 4 stores + 4 DMAs per turn, the turn's
// last one a DMA (every turn, also the first two, whose stores go to the dump area): DMA(i+2), issued in turn
// i+4-DEPTH, is followed by 8*(DEPTH-4) younger operations when turn i ends.

#ifndef CLX_K2_DEPTH
#define CLX_K2_DEPTH 12       // tiles in a group's ring (see clx_load_wave for what it has to cover)
#endif

// what a lane needs to move "its" 16 bytes of every tile: instruction k moves rows 16k .. 16k+15, 4 lanes per row
struct K2Mover {
    const int32_t* rp[4];
    uint32_t rn[4];
    uint32_t pc;               // which 16-byte piece of the row's 64 bytes this lane moves
    __device__ __forceinline__ void init(const int32_t* out, const K2Slot& S, int lane) {
        pc = ((uint32_t)lane & 3u) ^ (((uint32_t)lane >> 4) & 3u);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int src = k * 16 + (lane >> 2);
            const uint64_t ro = __shfl((unsigned long long)(S.row - out), src, 64);
            rp[k] = out + ro; rn[k] = __shfl(S.n, src, 64);
        }
    }
};

template <int OMAX, int DEPTH = CLX_K2_DEPTH, bool LEAN = false>
__device__ __forceinline__ void clx_predict_wave(int4 (*ring)[4][64], const K2Slot& S, uint32_t nblk, int lane CLX_TL_PARAM) {
    K2Predictor<OMAX, LEAN> P; P.init(S);
    const uint32_t sw = ((uint32_t)lane >> 2) & 3u;                               // this lane's row swizzle (lane = row view)
    auto fetch = [&](int32_t (&x)[CLX_BLK], uint32_t blk) __attribute__((always_inline)) {
        const int4* tile = &ring[blk % DEPTH][0][0];
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
            const int4 w = tile[(uint32_t)lane * 4u + (q ^ sw)];
            x[4 * q] = w.x; x[4 * q + 1] = w.y; x[4 * q + 2] = w.z; x[4 * q + 3] = w.w;
        }
    };
    int32_t xa[CLX_BLK], xb[CLX_BLK], y[CLX_BLK];
    // (raising this wave's issue priority with s_setprio -- its chain is the kernel's duration -- changed nothing measurable,
    //  alone or beside the Rice waves of a pipelined submission: 0.369 against 0.371 ms per step)
    clx_wg_barrier();                          // tiles 0 and 1 have landed
    fetch(xa, 0u);
    // two turns per trip so that the "current" and "next" blocks alternate between xa and xb without copies
    auto turn = [&](int32_t (&xc)[CLX_BLK], int32_t (&xn)[CLX_BLK], uint32_t i) __attribute__((always_inline)) {
        if (i < nblk) {
            fetch(xn, i + 1u);
            P.block(xc, y, i * CLX_BLK, K2NoHook());
            int4* tile = &ring[i % DEPTH][0][0];
#pragma unroll
            for (uint32_t q = 0; q < 4u; ++q) tile[(uint32_t)lane * 4u + (q ^ sw)] = make_int4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
        }
        CLX_TL_WAIT(clx_wg_barrier());
    };
    // nblk + 2 barriers after the first one, matched by the finisher (which lags: it stores block i-2 in turn i)
    const uint32_t nturn = nblk + 2u;
    for (uint32_t i = 0; i < nturn; i += 2u) { turn(xa, xb, i); if (i + 1u < nturn) turn(xb, xa, i + 1u); }
}

// ---- aligned rows, four waves per 64 rows (the latency build's current form) -----------------------------------------------
// A two-wave schedule (predictor + one finisher that also moved the data) left both waves with ~240 instructions per turn, and a
// turn lasts as long as the busier one.  Only the recurrence has to be one wave's serial chain: finishing a tile is independent
// of finishing the next one, and moving tiles is independent of both.  So per 64 rows:
//   predictor P, turn i : read x(i+1) | recurrence on x(i) | write y(i)                                  (clx_predict_wave)
//   finisher F(i mod 2) : turn i+1: read y(i) | wasted shift | decorrelate samples 0..7
//                         turn i+2: decorrelate samples 8..15 | tile | transposed read | 4 stores          (two turns per tile)
//   loader L, turn t    : DMA of block t-3+DEPTH into the tile that F finished reading in turn t-1 | wait until block t+2 landed
// and a turn is the predictor's chain and nothing else.  The loader's vmcnt counts LDS-DMA loads only (the finishers issue the
// stores): block j, requested in turn j+3-DEPTH, is followed by the DEPTH-5 younger blocks (4 loads each) when the turn before
// the predictor's read of it ends.  (One wave that mixes stores and DMAs under a counted vmcnt decoded wrongly under load with
// less than ~3 us between request and use.)
template <int MODE, int DEPTH = CLX_K2_DEPTH>
__device__ __forceinline__ void clx_finish_wave_alt(int4 (*ring)[4][64], int32_t* __restrict__ out, const K2Slot& S, const K2Finisher& F,
                                                    int32_t* __restrict__ dump, uint32_t nblk, uint32_t parity, int lane CLX_TL_PARAM) {
    K2Mover M; M.init(out, S, lane);
    const uint32_t sw = ((uint32_t)lane >> 2) & 3u;                               // lane = row view
    clx_wg_barrier();                          // = the predictor's first barrier: tiles 0 and 1 are there
    const uint32_t nturn = nblk + 2u;          // barriers after the first one (clx_predict_wave)
    uint32_t done = 0;                         // barriers passed
    // turns before this wave's first tile is written: 0 (and 1 for the odd finisher)
    for (uint32_t i = 0; i <= parity; ++i) { CLX_TL_WAIT(clx_wg_barrier()); ++done; }
    for (uint32_t k = parity; k < nblk; k += 2u) {
        int4* const tile = &ring[k % DEPTH][0][0];
        int32_t y[CLX_BLK];
        // ---- turn k+1
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
            const int4 v = tile[(uint32_t)lane * 4u + (q ^ sw)];
            y[4 * q] = v.x; y[4 * q + 1] = v.y; y[4 * q + 2] = v.z; y[4 * q + 3] = v.w;
        }
        F.shift_all(y);
        F.template decorrelate<MODE, 0, CLX_BLK / 2>(y);
        CLX_TL_WAIT(clx_wg_barrier()); ++done;
        // ---- turn k+2
        F.template decorrelate<MODE, CLX_BLK / 2, CLX_BLK>(y);
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) tile[(uint32_t)lane * 4u + (q ^ sw)] = make_int4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
        clx_wave_sync();
        const int4 w0 = tile[lane], w1 = tile[64 + lane], w2 = tile[128 + lane], w3 = tile[192 + lane];
        const uint32_t t_st = k * CLX_BLK + 4u * M.pc;
        auto move = [&](int kk, const int4& wk) __attribute__((always_inline)) {
            int32_t* p = (t_st < M.rn[kk]) ? const_cast<int32_t*>(M.rp[kk]) + t_st : dump + 4 * kk;
            *reinterpret_cast<int4*>(p) = wk;
        };
        move(0, w0); move(1, w1); move(2, w2); move(3, w3);
        CLX_TL_WAIT(clx_wg_barrier()); ++done;             // (also waits for the LDS reads: the loader may refill the tile next turn)
    }
    while (done < nturn) { CLX_TL_WAIT(clx_wg_barrier()); ++done; }
}

template <int DEPTH = CLX_K2_DEPTH>
__device__ __forceinline__ void clx_load_wave(int4 (*ring)[4][64], const int32_t* __restrict__ out, const K2Slot& S, uint32_t nblk, int lane CLX_TL_PARAM) {
    static_assert(DEPTH >= 6 && 4 * DEPTH < 64, "vmcnt is a 6-bit counter: the whole ring is requested at once at the start");
    K2Mover M; M.init(out, S, lane);
    auto dma = [&](uint32_t blk) __attribute__((always_inline)) {
        const uint32_t t = blk * CLX_BLK + 4u * M.pc;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t last = M.rn[k] >= 4u ? M.rn[k] - 4u : 0u;              // clamped: what lies past a row's end is never stored
            clx_glds16(M.rp[k] + (t < last ? t : last), clx_lds_addr(&ring[blk % DEPTH][k][0]));
        }
    };
    for (uint32_t i = 0; i < (uint32_t)DEPTH; ++i) dma(i);
    clx_wait_vmcnt<4 * (DEPTH - 2)>();         // tiles 0 and 1
    clx_wg_barrier();
    const uint32_t nturn = nblk + 2u;
    for (uint32_t t = 0; t < nturn; ++t) {
        if (t >= 3u) dma(t - 3u + (uint32_t)DEPTH);            // (wave-uniform; the loader has no other memory operation to disturb)
        clx_wait_vmcnt<4 * (DEPTH - 5)>();                      // block t+2 has landed: the predictor reads it next turn
        CLX_TL_WAIT(clx_wg_barrier());
    }
    clx_wait_vmcnt<0>();
}

// ---- aligned rows, throughput build: one wave per 64 rows ----------------------------------------------------------------
// The two-wave schedule above buys latency: it halves ONE wave's instruction stream, which is what a small batch waits
// for.  A large batch (several workgroups per CU) is bound by issue slots and LDS instead, and the barrier per turn
// turns into convoys between workgroups that share SIMDs (measured: 80 000 rows take 0.77 ms in the two-wave build,
// although their instructions fit in 0.25 ms).  This build keeps everything in one wave -- same tiles, same shapes, no
// barrier -- with a short ring (the other waves of the SIMD hide the memory latency).
//   turn i: wait DMA(i+1) | read x(i+1) | recurrence + finish on x(i) | tile | transposed read | 4 stores | DMA(i+DEPTH)
//   VMEM ops younger than DMA(i+1) when turn i starts: turns i+2-DEPTH .. i-1, 8 each = 8*(DEPTH-2)
template <int OMAX, int MODE, int DEPTH>
__device__ __forceinline__ void clx_predict_single(int4 (*ring)[4][64], int32_t* __restrict__ out, const K2Slot& S, const K2Finisher& F,
                                                   int32_t* __restrict__ dump, uint32_t nblk, int lane) {
    K2Predictor<OMAX> P; P.init(S);
    K2Mover M; M.init(out, S, lane);
    const uint32_t sw = ((uint32_t)lane >> 2) & 3u;                               // lane = row view
    auto dma = [&](uint32_t blk) __attribute__((always_inline)) {
        const uint32_t t = blk * CLX_BLK + 4u * M.pc;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t last = M.rn[k] >= 4u ? M.rn[k] - 4u : 0u;              // clamped: what lies past a row's end is never stored
            clx_glds16(M.rp[k] + (t < last ? t : last), clx_lds_addr(&ring[blk % DEPTH][k][0]));
        }
    };
    auto fetch = [&](int32_t (&x)[CLX_BLK], uint32_t blk) __attribute__((always_inline)) {
        const int4* tile = &ring[blk % DEPTH][0][0];
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
            const int4 v = tile[(uint32_t)lane * 4u + (q ^ sw)];
            x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
    };
    int32_t xa[CLX_BLK], xb[CLX_BLK], y[CLX_BLK];
    for (uint32_t i = 0; i < (uint32_t)DEPTH; ++i) dma(i);
    clx_wait_vmcnt<4 * (DEPTH - 1)>();
    clx_wave_sync();
    fetch(xa, 0u);
    auto turn = [&](int32_t (&xc)[CLX_BLK], int32_t (&xn)[CLX_BLK], uint32_t i) __attribute__((always_inline)) {
        if (i + 2u <= (uint32_t)DEPTH) clx_wait_vmcnt<4 * (DEPTH - 2)>(); else clx_wait_vmcnt<8 * (DEPTH - 2)>();
        clx_wave_sync();
        fetch(xn, i + 1u);
        P.block(xc, y, i * CLX_BLK, K2NoHook());
        F.template block<MODE>(y, K2NoHook());
        int4* tile = &ring[i % DEPTH][0][0];
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) tile[(uint32_t)lane * 4u + (q ^ sw)] = make_int4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
        clx_wave_sync();
        const uint32_t t = i * CLX_BLK + 4u * M.pc;
        int4 w0 = tile[lane], w1 = tile[64 + lane], w2 = tile[128 + lane], w3 = tile[192 + lane];
        *reinterpret_cast<int4*>(t < M.rn[0] ? const_cast<int32_t*>(M.rp[0]) + t : dump + 0) = w0;
        *reinterpret_cast<int4*>(t < M.rn[1] ? const_cast<int32_t*>(M.rp[1]) + t : dump + 4) = w1;
        *reinterpret_cast<int4*>(t < M.rn[2] ? const_cast<int32_t*>(M.rp[2]) + t : dump + 8) = w2;
        *reinterpret_cast<int4*>(t < M.rn[3] ? const_cast<int32_t*>(M.rp[3]) + t : dump + 12) = w3;
        clx_wait_lds();                          // the tile has been read: the DMA may overwrite it
        dma(i + DEPTH);
    };
    for (uint32_t i = 0; i < nblk; i += 2u) { turn(xa, xb, i); turn(xb, xa, i + 1u); }     // (a turn past the last block stores to the dump area)
    clx_wait_vmcnt<0>();
}

template <int OMAX, int DEPTH>
__device__ __forceinline__ void clx_predict_single_mode(int4 (*ring)[4][64], int32_t* __restrict__ out, const K2Slot& S, int32_t* __restrict__ dump,
                                                        uint32_t nblk, int lane) {
    K2Finisher F; F.init(S, lane);
    const int mode = F.mode();
    if (mode == 0)      clx_predict_single<OMAX, 0, DEPTH>(ring, out, S, F, dump, nblk, lane);
    else if (mode == 1) clx_predict_single<OMAX, 1, DEPTH>(ring, out, S, F, dump, nblk, lane);
    else                clx_predict_single<OMAX, 2, DEPTH>(ring, out, S, F, dump, nblk, lane);
}

// Workgroup = 8 waves = two groups of 64 rows, each with a predictor, two finishers and a loader.  Waves go to the CU's four
// SIMDs in cyclic order: waves w and w+4 share one.  The predictors (waves 0, 1), whose chains decide the kernel's duration,
// share theirs with the loaders (waves 4, 5: a handful of instructions per turn); the finishers (2, 3, 6, 7) share the other two.
// The groups share nothing but the barrier.
// Two kernels.  Every wave of a kernel is given the registers its hungriest role and path needs -- 203 with the unaligned rows,
// the 24-bit and exact evaluations of up to 32 taps -- and while a workgroup of the general kernel sits on a CU its eight waves
// keep them: 1 600 of the CU's 2 048 vector registers and 96 KB of its LDS, one workgroup per CU.  clx_k_predict16 takes the
// groups of 64 rows that are all 16-byte aligned, of 16-bit audio (K1's flag in the descriptor) and of at most 8 taps -- what a
// 16-bit stream is made of -- with the 16-bit evaluation and the exact one behind it, one group = four waves per workgroup and a
// 7-tile ring: 102 registers per wave, 28 KB per workgroup.  The predictor stages of several submissions in flight
// (clx_batch_submit) then are resident side by side, four or five workgroups to a CU, instead of queueing for whole CUs.
// clx_k_predict takes every other group with everything; both leave the other's groups alone.
#define CLX_K2_DEPTH16 7
template <bool FAST, int DEPTH, int GROUPS>
__device__ __forceinline__ void clx_predict_groups(int4 (*ring2)[DEPTH][4][64], int32_t* __restrict__ out, const clx_sf_desc* __restrict__ sfd,
                                                   uint32_t n_slots, int32_t* __restrict__ dump_all) {
    CLX_TL_BEGIN();
    const int lane = (int)threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t pair = GROUPS == 2 ? (wave & 1u) : 0u;
    const uint32_t role = GROUPS == 2 ? (wave >> 1) : wave;   // 0: predictor, 1: finisher of the even tiles, 2: loader, 3: finisher of the odd tiles
    const bool finisher = (role & 1u) != 0u;       // wave-uniform
    int4 (*ring)[4][64] = ring2[pair];
    const uint32_t group = blockIdx.x * (uint32_t)GROUPS + pair;
    const uint32_t slot = group * 64u + (uint32_t)lane;
    K2Slot S;
    S.d = &sfd[slot < n_slots ? slot : 0];
    S.n = 0; S.order = 0; S.shift = 0; S.wasted = 0; S.decor = 0; S.lim_log2 = 0;
    uint64_t base = 0;
    uint32_t narrow = 1u;                          // the slot's samples are at most 16 bits wide (K1's flag; empty slots do not object)
    if (slot < n_slots) {
        S.n = S.d->n; S.order = S.d->order; S.shift = S.d->shift; S.wasted = S.d->wasted; S.decor = S.d->decor; base = S.d->out_base;
        S.lim_log2 = S.d->lim_log2;
        narrow = S.d->flags & CLX_SF_NARROW;
    }
    if (S.n == 0u) { S.order = 0; S.shift = 0; S.wasted = 0; S.decor = 0; S.lim_log2 = 0; base = 0; narrow = 1u; }
    S.row = out + base;
    // a decorrelated pair is only formed when both of its subframes decoded (same block size, same mode)
    const uint32_t pn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)S.n, 0xB1, 0xF, 0xF, false);
    const uint32_t pd = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)S.decor, 0xB1, 0xF, 0xF, false);
    S.pair_ok = (S.decor != CLX_CH_INDEPENDENT) && pn == S.n && pd == S.decor && S.n != 0u;
    uint32_t nmax = S.n, omax = S.order;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        uint32_t a = __shfl_xor(nmax, s, 64); nmax = a > nmax ? a : nmax;
        uint32_t o = __shfl_xor(omax, s, 64); omax = o > omax ? o : omax;
    }
    int32_t* const dump = dump_all + (size_t)(group * 64u + (uint32_t)lane) * CLX_BLK;      // 64 bytes per lane (finishers / unaligned only)
    const bool work = (S.order != 0u) || (S.wasted != 0u) || S.pair_ok;
    // rows of nothing but CONSTANT / VERBATIM / FIXED-0 mono subframes without wasted bits are already final
    // (the three waves of a group see the same 64 slots, so every decision below is the same in all of them; a wave that
    //  returns no longer takes part in the workgroup's barriers)
    if (nmax == 0u || !__any(work)) return;
    // 16-byte row accesses need 16-byte aligned rows whose length is a multiple of 4 samples
    const bool al = (S.n == 0u) || ((((uintptr_t)S.row) & 15u) == 0u && (S.n & 3u) == 0u);
    const bool all_al = __all(al);
    if ((all_al && omax <= 8u && __all(narrow != 0u)) != FAST) return;      // the other kernel's group
    if (all_al) {
        const uint32_t nblk = (nmax + CLX_BLK - 1u) / CLX_BLK;
        if (finisher) {
            K2Finisher F; F.init(S, lane);
            const int mode = F.mode();
            const uint32_t parity = role >> 1;
            if (mode == 0)      clx_finish_wave_alt<0, DEPTH>(ring, out, S, F, dump, nblk, parity, lane CLX_TL_ARG);
            else if (mode == 1) clx_finish_wave_alt<1, DEPTH>(ring, out, S, F, dump, nblk, parity, lane CLX_TL_ARG);
            else                clx_finish_wave_alt<2, DEPTH>(ring, out, S, F, dump, nblk, parity, lane CLX_TL_ARG);
        }
        else if (role == 2u)  clx_load_wave<DEPTH>(ring, out, S, nblk, lane CLX_TL_ARG);
        else if (FAST) {
            if (omax <= 4u)   clx_predict_wave<4, DEPTH, true>(ring, S, nblk, lane CLX_TL_ARG);
            else              clx_predict_wave<8, DEPTH, true>(ring, S, nblk, lane CLX_TL_ARG);
        }
        else if (omax <= 4u)  clx_predict_wave<4, DEPTH>(ring, S, nblk, lane CLX_TL_ARG);
        else if (omax <= 8u)  clx_predict_wave<8, DEPTH>(ring, S, nblk, lane CLX_TL_ARG);
        else if (omax <= 12u) clx_predict_wave<12, DEPTH>(ring, S, nblk, lane CLX_TL_ARG);
        else                  clx_predict_wave<32, DEPTH>(ring, S, nblk, lane CLX_TL_ARG);
    } else if (!FAST && role == 0u) {
        if (omax <= 4u)       clx_predict_unaligned<4>(S, dump, nmax, lane);
        else if (omax <= 8u)  clx_predict_unaligned<8>(S, dump, nmax, lane);
        else if (omax <= 12u) clx_predict_unaligned<12>(S, dump, nmax, lane);
        else                  clx_predict_unaligned<32>(S, dump, nmax, lane);
    }
    CLX_TL_END(1, blockIdx.x * 8u + (threadIdx.x >> 6));
}
extern "C" __global__ __launch_bounds__(256)
void clx_k_predict16(int32_t* __restrict__ out, const clx_sf_desc* __restrict__ sfd, uint32_t n_slots, int32_t* __restrict__ dump_all) {
    __shared__ int4 ring2[1][CLX_K2_DEPTH16][4][64];  // 28 KiB: 7 tiles x 64 B of each of the group's 64 rows
    clx_predict_groups<true, CLX_K2_DEPTH16, 1>(ring2, out, sfd, n_slots, dump_all);
}
extern "C" __global__ __launch_bounds__(512)
void clx_k_predict(int32_t* __restrict__ out, const clx_sf_desc* __restrict__ sfd, uint32_t n_slots, int32_t* __restrict__ dump_all) {
    __shared__ int4 ring2[2][CLX_K2_DEPTH][4][64];    // per group 48 KiB: 12 tiles x 64 B of each of its 64 rows
    clx_predict_groups<false, CLX_K2_DEPTH, 2>(ring2, out, sfd, n_slots, dump_all);
}

// K2, throughput build (see clx_predict_single): one wave per 64 rows, picked by the host for large batches.
// Two kernels: groups whose highest predictor order is <= 12, and the rest.  The 32-tap predictor needs ~190 VGPRs; in a
// kernel of its own it does not halve the occupancy of the common case.
// `skip_fast`: the groups clx_k_predict16 takes (it ran before) are left alone.
template <bool HI>
__device__ __forceinline__ void clx_predict_1w_groups(int32_t* __restrict__ out, const clx_sf_desc* __restrict__ sfd, uint32_t n_slots, int32_t* __restrict__ dump_all,
                                                      uint32_t skip_fast) {
    constexpr int DEPTH = 3;
    __shared__ int4 ring[DEPTH][4][64];           // 12 KiB
    const int lane = (int)threadIdx.x;
    const uint32_t group = blockIdx.x;
    const uint32_t slot = group * 64u + (uint32_t)lane;
    K2Slot S;
    S.d = &sfd[slot < n_slots ? slot : 0];
    S.n = 0; S.order = 0; S.shift = 0; S.wasted = 0; S.decor = 0; S.lim_log2 = 0;
    uint64_t base = 0;
    uint32_t narrow = 1u;
    if (slot < n_slots) {
        S.n = S.d->n; S.order = S.d->order; S.shift = S.d->shift; S.wasted = S.d->wasted; S.decor = S.d->decor; base = S.d->out_base;
        S.lim_log2 = S.d->lim_log2;
        narrow = S.d->flags & CLX_SF_NARROW;
    }
    if (S.n == 0u) { S.order = 0; S.shift = 0; S.wasted = 0; S.decor = 0; S.lim_log2 = 0; base = 0; narrow = 1u; }
    S.row = out + base;
    const uint32_t pn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)S.n, 0xB1, 0xF, 0xF, false);
    const uint32_t pd = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)S.decor, 0xB1, 0xF, 0xF, false);
    S.pair_ok = (S.decor != CLX_CH_INDEPENDENT) && pn == S.n && pd == S.decor && S.n != 0u;
    uint32_t nmax = S.n, omax = S.order;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        uint32_t a = __shfl_xor(nmax, s, 64); nmax = a > nmax ? a : nmax;
        uint32_t o = __shfl_xor(omax, s, 64); omax = o > omax ? o : omax;
    }
    int32_t* const dump = dump_all + (size_t)(group * 64u + (uint32_t)lane) * CLX_BLK;
    const bool work = (S.order != 0u) || (S.wasted != 0u) || S.pair_ok;
    if (nmax == 0u || !__any(work)) return;
    if ((omax > 12u) != HI) return;                // the other kernel's group
    const bool al = (S.n == 0u) || ((((uintptr_t)S.row) & 15u) == 0u && (S.n & 3u) == 0u);
    if (skip_fast != 0u && __all(al) && omax <= 8u && __all(narrow != 0u)) return;       // clx_k_predict16's group
    if (__all(al)) {
        const uint32_t nblk = (nmax + CLX_BLK - 1u) / CLX_BLK;
        if (HI)               clx_predict_single_mode<32, DEPTH>(ring, out, S, dump, nblk, lane);
        else if (omax <= 4u)  clx_predict_single_mode<4, DEPTH>(ring, out, S, dump, nblk, lane);
        else if (omax <= 8u)  clx_predict_single_mode<8, DEPTH>(ring, out, S, dump, nblk, lane);
        else                  clx_predict_single_mode<12, DEPTH>(ring, out, S, dump, nblk, lane);
    } else {
        if (HI)               clx_predict_unaligned<32>(S, dump, nmax, lane);
        else if (omax <= 4u)  clx_predict_unaligned<4>(S, dump, nmax, lane);
        else if (omax <= 8u)  clx_predict_unaligned<8>(S, dump, nmax, lane);
        else                  clx_predict_unaligned<12>(S, dump, nmax, lane);
    }
}
extern "C" __global__ __launch_bounds__(64)
void clx_k_predict_1w(int32_t* __restrict__ out, const clx_sf_desc* __restrict__ sfd, uint32_t n_slots, int32_t* __restrict__ dump_all, uint32_t skip_fast) {
    clx_predict_1w_groups<false>(out, sfd, n_slots, dump_all, skip_fast);
}
extern "C" __global__ __launch_bounds__(64)
void clx_k_predict_1w_hi(int32_t* __restrict__ out, const clx_sf_desc* __restrict__ sfd, uint32_t n_slots, int32_t* __restrict__ dump_all, uint32_t skip_fast) {
    clx_predict_1w_groups<true>(out, sfd, n_slots, dump_all, skip_fast);
}

// ------------------------------------------------------------------------------------------------
// K3: CRC-16 (poly 0x8005, init 0, MSB first; crc.rs:69, 109-112) of each successfully decoded
// frame's bytes [byte_off, byte_off + ceil(end_bit/8)) against the big-endian footer that follows
// (frame.rs:752-763).  CRC with init 0 is linear: crc(A||B) = crc(A)*x^(8|B|) mod P  xor  crc(B).
// Each lane folds a contiguous slice, then the slices are combined right to left.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t clx_crc16_byte(uint32_t crc, uint32_t byte) {
    crc ^= byte << 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) crc = (crc & 0x8000u) ? ((crc << 1) ^ 0x8005u) & 0xffffu : (crc << 1) & 0xffffu;
    return crc;
}
// (a * b) mod P over GF(2), 16-bit polynomials
__host__ __device__ __forceinline__ uint32_t clx_gf_mulmod(uint32_t a, uint32_t b) {
    uint32_t r = 0;
#pragma unroll
    for (int i = 15; i >= 0; --i) {
        r = (r & 0x8000u) ? ((r << 1) ^ 0x8005u) & 0xffffu : (r << 1) & 0xffffu;
        if ((b >> i) & 1u) r ^= a;
    }
    return r;
}
// x^(8*nbytes) mod P
__device__ __forceinline__ uint32_t clx_xpow8(uint32_t nbytes) {
    uint32_t result = 1u;            // x^0
    uint32_t base = 0x0100u;         // x^8
    while (nbytes) {
        if (nbytes & 1u) result = clx_gf_mulmod(result, base);
        base = clx_gf_mulmod(base, base);
        nbytes >>= 1;
    }
    return result;
}
__host__ __device__ __forceinline__ uint32_t clx_xpow8_64(uint64_t nbytes) {
    uint32_t result = 1u, base = 0x0100u;
    while (nbytes) {
        if (nbytes & 1ull) result = clx_gf_mulmod(result, base);
        base = clx_gf_mulmod(base, base);
        nbytes >>= 1;
    }
    return result;
}

// K3 tables, generated at compile time (constant memory), copied to LDS by every workgroup:
//   t[k][v]  state after byte v followed by k zero bytes, from state 0 ("slicing by 4": a dword of the message is four
//            independent look-ups instead of four dependent ones)
//   gap[h][v] (v << 8h) * x^(8*1008) mod P: carries a lane's running CRC over the 1008 bytes that the other lanes cover
//            before its next 16
//   lane[L]  x^(8*16*(63-L)) mod P: what lane L's CRC is multiplied by in the final sum (its bytes are followed by the 63-L
//            later lanes' 16 each)
struct K3Rom { uint16_t t[4][256]; uint16_t gap[2][256]; uint16_t lane[64]; };
constexpr uint32_t clx_c_mulmod(uint32_t a, uint32_t b) {
    uint32_t r = 0;
    for (int i = 15; i >= 0; --i) {
        r = (r & 0x8000u) ? ((r << 1) ^ 0x8005u) & 0xffffu : (r << 1) & 0xffffu;
        if ((b >> i) & 1u) r ^= a;
    }
    return r;
}
constexpr uint32_t clx_c_xpow8(uint32_t nbytes) {
    uint32_t result = 1u, base = 0x0100u;
    while (nbytes) { if (nbytes & 1u) result = clx_c_mulmod(result, base); base = clx_c_mulmod(base, base); nbytes >>= 1; }
    return result;
}
constexpr K3Rom clx_make_k3_rom() {
    K3Rom r{};
    for (uint32_t v = 0; v < 256u; ++v) {
        uint32_t c = v << 8;
        for (int i = 0; i < 8; ++i) c = (c & 0x8000u) ? ((c << 1) ^ 0x8005u) & 0xffffu : (c << 1) & 0xffffu;
        r.t[0][v] = (uint16_t)c;
    }
    for (uint32_t k = 1; k < 4u; ++k)
        for (uint32_t v = 0; v < 256u; ++v) { const uint32_t c = r.t[k - 1][v]; r.t[k][v] = (uint16_t)(((c << 8) & 0xffffu) ^ r.t[0][c >> 8]); }
    const uint32_t g = clx_c_xpow8(1008u);
    for (uint32_t v = 0; v < 256u; ++v) { r.gap[0][v] = (uint16_t)clx_c_mulmod(v, g); r.gap[1][v] = (uint16_t)clx_c_mulmod(v << 8, g); }
    for (uint32_t L = 0; L < 64u; ++L) r.lane[L] = (uint16_t)clx_c_xpow8(16u * (63u - L));
    return r;
}
__constant__ K3Rom clx_k3_rom = clx_make_k3_rom();

// One wavefront per frame (four per workgroup, which share the tables; a wave takes frames f, f + waves, ...).  The frame's bytes
// are taken in rounds of 1 KiB that END at the frame's end -- lane L the 16 bytes [1024 r + 16 L, +16) of the round -- so the
// first round starts in front of the frame: those bytes count as zeros, which a CRC with initial value 0 does not see
// (crc.rs:109-112).  Every position then has a multiplier that does not depend on the frame's length.
// `todo` (lane path): which frames are still to be checked -- the lean kernels' lanes gather the CRC of the frames they decode
// themselves (clx_crct.h), clx_k_finalize has judged those.
__device__ __forceinline__ void clx_crc16_frames(const uint8_t* __restrict__ arena, const clx_dev_frame* __restrict__ frames, uint32_t n_frames,
                                                 clx_frame_result* __restrict__ results, const uint32_t* __restrict__ todo) {
    __shared__ K3Rom T;
    __shared__ uint32_t any_todo;
    const int lane = (int)threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6, n_waves = gridDim.x * 4u;
    if (todo != nullptr) {
        // (the usual case on the lane path: the decode lanes have settled every frame -- a workgroup that finds nothing to do leaves
        //  before it copies the tables; in a full machine these workgroups wait for room behind the other stream's decode kernel)
        if (threadIdx.x == 0) any_todo = 0u;
        __syncthreads();
        bool mine = false;
        for (uint32_t f = blockIdx.x * 4u + wave + n_waves * (uint32_t)lane; f < n_frames; f += n_waves * 64u) mine = mine || todo[f] != 0u;
        if (mine) any_todo = 1u;
        __syncthreads();
        if (any_todo == 0u) return;
    }
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&clx_k3_rom);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&T);
        for (uint32_t i = threadIdx.x; i < sizeof(K3Rom) / 4u; i += 256u) dst[i] = src[i];
    }
    __syncthreads();
    for (uint32_t f = blockIdx.x * 4u + wave; f < n_frames; f += n_waves) {
        if (todo != nullptr && todo[f] == 0u) continue;                      // wave-uniform
        const clx_dev_frame fr = frames[f];
        const clx_frame_result r = results[f];
        if (r.status != CLX_OK || (fr.flags & 1u)) continue;                 // wave-uniform
        const uint32_t nbytes = (uint32_t)((r.end_bit + 7u) >> 3);
        if ((uint64_t)nbytes * 8u + 16u > (uint64_t)fr.limit_bits) {           // read_be_u16 fails: frame.rs:754
            if (lane == 0) { results[f].status = CLX_IO_ERROR; results[f].msg = CLX_MSG_UNEXPECTED_EOF; }
            continue;
        }
        const uint8_t* const p = arena + fr.byte_off;
        const uint32_t rounds = (nbytes + 1023u) >> 10;
        // byte offset (relative to the frame's first byte) of this lane's 16 bytes in round 0; negative in front of the frame
        int32_t o = (int32_t)nbytes - (int32_t)(rounds << 10) + 16 * lane;
        uint32_t crc = 0;
        for (uint32_t rd = 0; rd < rounds; ++rd, o += 1024) {
            // the 16 message bytes as four little-endian dwords (byte 0 first = lowest byte of w[0])
            uint32_t w[4];
            if (o >= 0) {
                const uintptr_t adr = (uintptr_t)(p + o);
                const uint32_t* q = reinterpret_cast<const uint32_t*>(adr & ~(uintptr_t)3);
                const uint32_t sh = 8u * (uint32_t)(adr & 3u);
                const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];   // (the arena is padded by 16 bytes: claxon_hip.h)
                w[0] = clx_alignbit(d1, d0, sh); w[1] = clx_alignbit(d2, d1, sh); w[2] = clx_alignbit(d3, d2, sh); w[3] = clx_alignbit(d4, d3, sh);
            } else {
                // (only in round 0, only the lanes in front of the frame's first byte) byte by byte, zeros in front
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t v = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int32_t at = o + 4 * k + j; if (at >= 0) v |= (uint32_t)p[at] << (8 * j); }
                    w[k] = v;
                }
            }
            if (rd != 0u) crc = (uint32_t)T.gap[1][crc >> 8] ^ (uint32_t)T.gap[0][crc & 0xffu];      // over the other lanes' 1008 bytes
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // state folded into the first two bytes, then four independent look-ups
                const uint32_t t = w[k] ^ (crc >> 8) ^ ((crc & 0xffu) << 8);
                crc = (uint32_t)T.t[3][t & 0xffu] ^ (uint32_t)T.t[2][(t >> 8) & 0xffu] ^ (uint32_t)T.t[1][(t >> 16) & 0xffu] ^ (uint32_t)T.t[0][t >> 24];
            }
        }
        // crc(A || B) = crc(A) * x^(8|B|) xor crc(B): every lane's bytes are followed by the later lanes' 16 each
        uint32_t contrib = clx_gf_mulmod(crc, (uint32_t)T.lane[lane]);
#pragma unroll
        for (int s2 = 32; s2 >= 1; s2 >>= 1) contrib ^= __shfl_xor(contrib, s2, 64);
        if (lane == 0) {
            const uint32_t presumed = ((uint32_t)p[nbytes] << 8) | (uint32_t)p[nbytes + 1];
            if (contrib != presumed) { results[f].status = CLX_FORMAT_ERROR; results[f].msg = CLX_MSG_FRAME_CRC_MISMATCH; }
        }
    }
}
extern "C" __global__ __launch_bounds__(256)
void clx_k_crc16(const uint8_t* __restrict__ arena, const clx_dev_frame* __restrict__ frames, uint32_t n_frames,
                 clx_frame_result* __restrict__ results) {
    clx_crc16_frames(arena, frames, n_frames, results, nullptr);
}
// the same for the runs of a merged lane-path launch (blockIdx.y picks the run)
extern "C" __global__ __launch_bounds__(256)
void clx_k_crc16_runs(const clx_runs runs, const clx_dev_frame* __restrict__ frames, uint32_t n_frames) {
    const clx_run& R = runs.r[blockIdx.y];
    clx_crc16_frames(R.arena, frames, n_frames, R.results, (R.flags & CLX_RUN_CRC) ? R.crc_todo : nullptr);
}

// ------------------------------------------------------------------------------------------------
// K4: interleave / narrow output stage.  What every caller of the reference does right after the
// hot path: walk the block channel-interleaved (FlacSamples, lib.rs:473-520; Block::stereo_samples -> i16 WAV,
// examples/decode.rs:48-62).  Frame f's planar samples planar[out_off + c*bs + i] become little-endian two's
// complement PCM of `sb` bytes per sample at byte (out_off + i*C + c) * sb of `dst` -- the byte order the STREAMINFO
// MD5 is defined over (metadata.rs:52-53).  Frames that failed to decode are skipped (they expose nothing).
// One workgroup per frame; HBM-bound: reads 4 B, writes sb B per sample, both coalesced on the common shapes.
// ------------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256)
void clx_k_interleave(const int32_t* __restrict__ planar, const clx_dev_frame* __restrict__ frames,
                      const clx_frame_result* __restrict__ results, uint32_t n_frames,
                      uint8_t* __restrict__ dst, uint32_t sb) {
    const uint32_t f = blockIdx.x;
    if (f >= n_frames) return;
    if (results && results[f].status != CLX_OK) return;
    const clx_dev_frame fr = frames[f];
    const uint32_t C = fr.n_channels, bs = fr.block_size;
    const int32_t* __restrict__ src = planar + fr.out_off;
    uint8_t* __restrict__ d = dst + fr.out_off * (uint64_t)sb;
    const bool even = (fr.out_off & 1ull) == 0ull && ((uintptr_t)dst & 7u) == 0u;
    if (C == 2u && sb == 2u && even) {                 // 16-bit stereo: one dword per sample pair
        uint32_t* d32 = reinterpret_cast<uint32_t*>(d);
        for (uint32_t i = threadIdx.x; i < bs; i += 256u)
            d32[i] = ((uint32_t)src[i] & 0xffffu) | ((uint32_t)src[bs + i] << 16);
    } else if (C == 2u && sb == 4u && even) {
        int2* d64 = reinterpret_cast<int2*>(d);
        for (uint32_t i = threadIdx.x; i < bs; i += 256u) d64[i] = make_int2(src[i], src[bs + i]);
    } else if (sb == 4u && ((uintptr_t)dst & 3u) == 0u) {
        int32_t* d32 = reinterpret_cast<int32_t*>(d);
        for (uint32_t i = threadIdx.x; i < bs; i += 256u)
            for (uint32_t c = 0; c < C; ++c) d32[i * C + c] = src[c * bs + i];
    } else if (sb == 2u && ((uintptr_t)dst & 1u) == 0u) {
        uint16_t* d16 = reinterpret_cast<uint16_t*>(d);
        for (uint32_t i = threadIdx.x; i < bs; i += 256u)
            for (uint32_t c = 0; c < C; ++c) d16[i * C + c] = (uint16_t)src[c * bs + i];
    } else {                                           // 8 / 24-bit packing, odd alignments: byte stores
        for (uint32_t i = threadIdx.x; i < bs; i += 256u)
            for (uint32_t c = 0; c < C; ++c) {
                const uint32_t v = (uint32_t)src[c * bs + i];
                uint8_t* q = d + (size_t)(i * C + c) * sb;
                for (uint32_t k = 0; k < sb; ++k) q[k] = (uint8_t)(v >> (8u * k));
            }
    }
}

// K4b: what a failed frame left in the planar output is cleared -- the reference drops the buffer of a frame that fails
// (frame.rs:667: Err consumes it), so nothing of it may be observable in a buffer handed back to the host.
extern "C" __global__ __launch_bounds__(256)
void clx_k_clear_failed(int32_t* __restrict__ planar, const clx_dev_frame* __restrict__ frames,
                        const clx_frame_result* __restrict__ results, uint32_t n_frames) {
    const uint32_t f = blockIdx.x;
    if (f >= n_frames || results[f].status == CLX_OK) return;
    const clx_dev_frame fr = frames[f];
    const uint32_t total = (uint32_t)fr.n_channels * fr.block_size;
    for (uint32_t i = threadIdx.x; i < total; i += 256u) planar[fr.out_off + i] = 0;
}

// ------------------------------------------------------------------------------------------------
// K5 / K6: frame indexer for raw streams (header grammar frame.rs:131-316; the reference has no resync,
// frame.rs:601-602).  The byte work is data-parallel and lives here; the (tiny) chain logic lives in the host
// (clx_index_frames_device, clx_api.hip), which also re-parses every candidate with the authoritative host parser.
//   K5 clx_k_find_headers: every byte position is tested for "sync code + a header that obeys the grammar's length
//      rules + matching CRC-8" (a superset of the valid headers); hits are appended to a list (unordered).
//   K6 clx_k_span_crc16: CRC-16 (as K3) of the bytes between consecutive sorted candidates.  CRC-16 with init 0 and
//      no final xor is linear and a frame followed by its own footer has CRC 0, so the host confirms a frame
//      [p_i, p_j) by folding the spans in between: crc(A||B) = crc(A) * x^(8|B|) xor crc(B).
// ------------------------------------------------------------------------------------------------
// length of a CRC-8-valid frame header at d (including the CRC byte), 0 if there is none; reads < 17 bytes
__device__ __forceinline__ uint32_t clx_header_probe(const uint8_t* __restrict__ d, uint64_t avail) {
    if (avail < 6u) return 0u;
    const uint32_t b2 = d[2], b3 = d[3], b4 = d[4];
    const uint32_t bn = b2 >> 4, sn = b2 & 15u;
    if (bn == 0u || sn == 15u) return 0u;
    if ((b3 >> 4) > 10u || (b3 & 1u)) return 0u;
    const uint32_t bc = (b3 >> 1) & 7u;
    if (bc == 3u || bc == 7u) return 0u;
    uint32_t ones = (uint32_t)__clz((int)(~(b4 << 24)));          // leading one bits of the first varint byte (read_var_length_int, frame.rs:64-105)
    if (ones > 8u) ones = 8u;
    if (ones == 1u) return 0u;
    const uint32_t extra = ones ? ones - 1u : 0u;
    const uint32_t len = 5u + extra + (bn == 6u ? 1u : bn == 7u ? 2u : 0u) + (sn == 12u ? 1u : (sn == 13u || sn == 14u) ? 2u : 0u);
    if ((uint64_t)len + 1u > avail) return 0u;
    for (uint32_t i = 0; i < extra; ++i) if ((d[5u + i] & 0xc0u) != 0x80u) return 0u;
    uint32_t crc = 0;                                              // CRC-8, poly 0x07, init 0 (crc.rs:13-31)
    for (uint32_t i = 0; i < len; ++i) {
        crc ^= d[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) crc = (crc & 0x80u) ? ((crc << 1) ^ 0x07u) & 0xffu : (crc << 1) & 0xffu;
    }
    return crc == d[len] ? len + 1u : 0u;
}

extern "C" __global__ __launch_bounds__(256)
void clx_k_find_headers(const uint8_t* __restrict__ data, uint64_t len, uint64_t start, uint64_t* __restrict__ cand,
                        uint32_t cap, uint32_t* __restrict__ count) {
    // 16 positions per thread: one aligned 16-byte load + the byte after it
    const uint64_t base = (start & ~15ull) + ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 16u;
    if (base >= len) return;
    const uint4 v = *reinterpret_cast<const uint4*>(data + base);          // (allocation is padded to 16 bytes + 16)
    const uint32_t nxt = data[base + 16u];
    const uint32_t w[5] = { v.x, v.y, v.z, v.w, nxt };
#pragma unroll
    for (uint32_t i = 0; i < 16u; ++i) {
        const uint32_t b0 = (w[i >> 2] >> (8u * (i & 3u))) & 0xffu;
        const uint32_t b1 = (w[(i + 1u) >> 2] >> (8u * ((i + 1u) & 3u))) & 0xffu;
        if (b0 == 0xffu && (b1 & 0xfeu) == 0xf8u) {
            const uint64_t p = base + i;
            if (p >= start && p + 2u <= len && clx_header_probe(data + p, len - p) != 0u) {
                const uint32_t k = atomicAdd(count, 1u);
                if (k < cap) cand[k] = p;
            }
        }
    }
}

// span j = [pos[j], pos[j+1]) for j < n_spans (pos has n_spans + 1 entries, ascending); one wave per span
extern "C" __global__ __launch_bounds__(64)
void clx_k_span_crc16(const uint8_t* __restrict__ data, const uint64_t* __restrict__ pos, uint32_t n_spans, uint16_t* __restrict__ crc_out) {
    const int lane = (int)threadIdx.x;
    const uint32_t j = blockIdx.x;
    if (j >= n_spans) return;
    const uint8_t* p = data + pos[j];
    const uint64_t nbytes = pos[j + 1] - pos[j];
    const uint64_t per = (nbytes + 63u) / 64u;
    const uint64_t lo = (uint64_t)lane * per < nbytes ? (uint64_t)lane * per : nbytes;
    const uint64_t hi = lo + per < nbytes ? lo + per : nbytes;
    uint32_t crc = 0;
    for (uint64_t i = lo; i < hi; ++i) crc = clx_crc16_byte(crc, p[i]);
    const uint64_t tail = nbytes - hi;
    uint32_t contrib = (hi > lo) ? clx_gf_mulmod(crc, clx_xpow8_64(tail)) : 0u;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) contrib ^= __shfl_xor(contrib, s, 64);
    if (lane == 0) crc_out[j] = (uint16_t)contrib;
}

// K7: the first 20 bytes of every candidate, gathered for the host's authoritative header parse (zero padded at the end)
extern "C" __global__ __launch_bounds__(256)
void clx_k_gather_headers(const uint8_t* __restrict__ data, uint64_t len, const uint64_t* __restrict__ pos, uint32_t n, uint8_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint64_t p = pos[i];
    for (uint32_t k = 0; k < 20u; ++k) out[(size_t)i * 20u + k] = p + k < len ? data[p + k] : 0u;
}
