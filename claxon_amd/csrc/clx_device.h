// clx_device.h -- device-side records shared by the host library and the HIP kernels.
//
// Data layout in HBM:
//   arena   : the compressed frames, byte for byte as in the FLAC stream (16-byte aligned base)
//   frames  : one clx_dev_frame per frame (32 B) -- what read_frame_header_or_eof (frame.rs:131-316)
//             yields, plus where the decode goes
//   sfdesc  : one clx_sf_desc per predictor slot (80 B): the parsed subframe header K1 hands to K2
//   out     : planar i32 PCM; frame i's channel c at out[out_off + c*block_size ...) (frame.rs:409-410)
//   results : one clx_frame_result per frame
#ifndef CLX_DEVICE_H
#define CLX_DEVICE_H

// tier statistics: compiled in by the wave simulator only (tools/sim_tiers.py), nothing on the device
#ifndef CLX_STAT
#define CLX_STAT(i, n) do {} while (0)
#endif

#include <stdint.h>
#include <stddef.h>

#include "clx_crct.h"

struct clx_dev_frame {
    uint64_t byte_off;       // frame start (sync code) in the arena
    uint64_t out_off;        // sample index of channel 0 in `out`
    uint32_t limit_bits;     // readable bits from byte_off: 8*min(max_bytes, arena_len-byte_off), capped at 2^31
    uint32_t first_slot;     // predictor slot of subframe 0 (even for stereo-decorrelated frames)
    uint16_t header_bytes;   // first subframe starts at byte_off+header_bytes
    uint16_t block_size;
    uint8_t  n_channels;
    uint8_t  channel_assignment;   // CLX_CH_*
    uint8_t  bps;
    uint8_t  flags;          // bit0: bare subframe (no CRC-16 footer)
};

#define CLX_SF_NARROW 1u
struct clx_sf_desc {
    uint64_t out_base;       // sample index of this subframe's first sample in `out`
    uint16_t n;              // block size; 0 = empty slot / frame failed before this subframe was parsed
    uint8_t  lim_log2;       // 0..23: K1 proved |s| <= 2^lim_log2 makes 32-bit/24-bit-factor evaluation exact; 0xff: use i64
    uint8_t  flags;          // CLX_SF_NARROW: the frame's samples are at most 16 bits wide (the side channel has one more)
    uint8_t  order;          // IIR taps (0: constant/verbatim/fixed-0; fixed 1..4; lpc 1..32)
    uint8_t  shift;          // qlp shift (0 for fixed predictors)
    uint8_t  wasted;         // wasted bits per sample: final left shift (subframe.rs:216-225)
    uint8_t  decor;          // CLX_CH_* of the frame (0 = independent): stereo decorrelation role
    int16_t  coef[32];       // coef[j] multiplies s[i-1-j] (first coded coefficient <-> newest sample)
};

// Lane path: one launch may decode several RUNS of the same planned batch -- the same frame descriptors against different arenas
// and output buffers (consecutive submissions merged into one grid: blockIdx.y picks the run).  What differs between the runs:
#ifndef CLX_MAX_MERGE
#define CLX_MAX_MERGE 12
#endif
struct clx_run {
    const uint8_t* arena;    // the run's compressed frames
    uint64_t alloc_len;      // bytes readable from `arena` (the padded allocation)
    int32_t* out;            // the run's planar output
    uint32_t* sf_start;      // scratch: start bit of every later subframe (clx_k_scan -> decode kernels)
    uint32_t* errkey;        // scratch: first-error key per frame
    uint64_t* end_bits;      // scratch: end bit per frame
    uint32_t* taken;         // per group of 64 slots: == gen when clx_k_lean decoded the group in this run
    clx_frame_result* results;
    const uint32_t* slot_frame;  // which frame the lane at each (physical) slot decodes -- the plan's map, or this run's own when waves are
    const uint32_t* first_slot;  // composed by content (clx_k_compose) -- and, per frame, the physical slot of its first subframe
    uint32_t* fkey;          // scratch, per frame: the content class clx_k_scan found (clx_k_compose's sort key); null: no composition
    clx_crc_part* crc_part;  // scratch, per predictor slot: what the lean kernels' lanes found of their frame's CRC-16 (clx_crct.h)
    uint32_t* crc_todo;      // scratch, per frame: clx_k_finalize -> clx_k_crc16_runs: 1 = the stand-alone kernel has to check this frame
    int32_t* planar;         // narrow output only: the general kernels' staging rows for this run -- 64 rows of CLX_RUN_STAGE_STRIDE samples per workgroup
    uint32_t gen;
    uint32_t flags;          // CLX_RUN_CRC: the frames' CRC-16 is verified (the decode lanes gather it, clx_k_finalize judges it)
};
#define CLX_RUN_CRC 1u
#define CLX_RUN_PCM16 2u     // `out` holds interleaved 16-bit PCM (claxon_hip.h: CLX_OUT_PCM16)
#define CLX_RUN_PCM24 4u     // `out` holds interleaved packed 24-bit PCM (CLX_OUT_PCM24)
// narrow output: bits 16..31 of the flags = the length of a staging row in units of four samples (>= the batch's largest block size)
#define CLX_RUN_STAGE_STRIDE(flags) (((flags) >> 16) * 4u)
#define CLX_RUN_STAGE_BITS(stride) ((((uint32_t)(stride) + 3u) / 4u) << 16)
struct clx_runs { clx_run r[CLX_MAX_MERGE]; };       // passed to the kernels by value

// clx_k_pool (clx_lean.hip): the scan waves and the 16-bit tier's decode waves of one merged launch as TICKETS that a grid of resident
// waves takes off a counter -- first every run's scan waves, then every run's groups of 64 slots.  One of these per internal stream
// (launches of one stream follow each other); zeroed in front of every launch.
struct clx_pool_state {
    uint32_t next;                       // the next ticket
    uint32_t stuck;                      // decode tickets whose wave gave up waiting for its run's scan (their groups go to the general kernels)
    uint32_t scan_done[CLX_MAX_MERGE];   // per run of the launch: scan tickets that are done
};
struct clx_pool_args {                   // clx_k_pool's arguments (one struct by value: the kernel reads it again for every ticket)
    clx_runs runs;
    const clx_dev_frame* frames;
    const uint32_t* multi;               // the multi-channel frames (the scan's lanes)
    clx_pool_state* ps;
    const uint32_t* order;               // test hook: ticket i of the counter stands for ticket order[i] (null: itself)
    int32_t* dump_all;
    uint32_t n_runs, n_slots, n_multi, pad;
};

// clx_k_compose: a window of consecutive stereo frames of one block size whose lanes are dealt by content class (clx_plan.h)
#define CLX_COMPOSE_WINDOW 16384u
#define CLX_COMPOSE_KEYS 32u
#define CLX_COMPOSE_THREADS 512u
struct clx_window { uint32_t f_lo, f_hi, s_lo, pad; };   // frames [f_lo, f_hi), physical slots from s_lo (= the first frame's canonical slot) on
// content class of a stereo frame (clx_k_scan): bit 4 a constant / verbatim subframe, bits 3-2 the predictor order class, bits 1-0 the
// channel assignment.  Sorting by it puts waves of one predictor build side by side.
#define CLX_FKEY(special, order_class, assignment) (((special) ? 16u : 0u) | ((uint32_t)(order_class) << 2) | ((uint32_t)(assignment) & 3u))

#ifdef __cplusplus
static_assert(sizeof(clx_run) == 120, "clx_run layout");
static_assert(sizeof(clx_dev_frame) == 32, "clx_dev_frame layout");
static_assert(sizeof(clx_sf_desc) == 80, "clx_sf_desc layout");
// K1 writes the 16 bytes in front of `coef` as one store: {out_base | n, lim_log2, flags | order, shift, wasted, decor}
static_assert(offsetof(clx_sf_desc, out_base) == 0 && offsetof(clx_sf_desc, n) == 8 && offsetof(clx_sf_desc, lim_log2) == 10 &&
              offsetof(clx_sf_desc, flags) == 11 && offsetof(clx_sf_desc, order) == 12 && offsetof(clx_sf_desc, shift) == 13 &&
              offsetof(clx_sf_desc, wasted) == 14 && offsetof(clx_sf_desc, decor) == 15 && offsetof(clx_sf_desc, coef) == 16, "clx_sf_desc layout");

// Debug aid (tools/timeline.py): with -DCLX_TIMELINE every wave of an instrumented kernel records when it started and
// ended (s_memrealtime, 100 MHz), its shader clock ticks and where it ran.  Not part of the product build.
#if defined(CLX_TIMELINE) && defined(__HIPCC__)
#define CLX_TL_WAVES 65536
__device__ uint64_t clx_timeline_buf[4][CLX_TL_WAVES][14];
#define CLX_TL_BEGIN() const uint64_t tl_r0 = __builtin_amdgcn_s_memrealtime(), tl_c0 = __builtin_amdgcn_s_memtime(); uint64_t tl_wait = 0; (void)tl_wait; uint64_t tl_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t tl_mark = tl_c0; (void)tl_ph; (void)tl_mark
#define CLX_TL_END(kid, wave) do { \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        if ((threadIdx.x & 63u) == 0 && (wave) < CLX_TL_WAVES) { \
            uint32_t hwid, xcc; \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); \
            uint64_t* tl = clx_timeline_buf[kid][wave]; \
            tl[0] = tl_r0; tl[1] = __builtin_amdgcn_s_memrealtime(); tl[2] = tl_c0; tl[3] = __builtin_amdgcn_s_memtime(); \
            tl[4] = ((uint64_t)xcc << 32) | hwid; tl[5] = tl_wait; for (int tl_i = 0; tl_i < 8; ++tl_i) tl[6 + tl_i] = tl_ph[tl_i]; \
        } } while (0)
/* the same with the wave's record taken off a counter (waves of several launches side by side: clx_debug_timeline_reset zeroes it); `tag` goes to field 5 */
__device__ uint32_t clx_timeline_n[4];
#define CLX_TL_END_SEQ(kid, tag) do { \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        if ((threadIdx.x & 63u) == 0) { \
            const uint32_t tl_w = atomicAdd(&clx_timeline_n[kid], 1u); \
            if (tl_w < CLX_TL_WAVES) { \
                uint32_t hwid, xcc; \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); \
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); \
                uint64_t* tl = clx_timeline_buf[kid][tl_w]; \
                tl[0] = tl_r0; tl[1] = __builtin_amdgcn_s_memrealtime(); tl[2] = tl_c0; tl[3] = __builtin_amdgcn_s_memtime(); \
                tl[4] = ((uint64_t)xcc << 32) | hwid; tl[5] = (uint64_t)(tag); \
            } } } while (0)
#define CLX_TL_PARAM , uint64_t& tl_wait
#define CLX_TL_ARG , tl_wait
#define CLX_TL_PH_PARAM , uint64_t (&tl_ph)[8], uint64_t& tl_mark
#define CLX_TL_PH_ARG , tl_ph, tl_mark
/* ticks since the previous mark go to phase `i` */
#define CLX_TL_PHASE(i) do { const uint64_t tl_n = __builtin_amdgcn_s_memtime(); tl_ph[i] += tl_n - tl_mark; tl_mark = tl_n; } while (0)
#define CLX_TL_WAIT(stmt) do { const uint64_t tl_a = __builtin_amdgcn_s_memtime(); stmt; tl_wait += __builtin_amdgcn_s_memtime() - tl_a; } while (0)
#else
#define CLX_TL_PARAM
#define CLX_TL_ARG
#define CLX_TL_PH_PARAM
#define CLX_TL_PH_ARG
#define CLX_TL_PHASE(i) do {} while (0)
#define CLX_TL_WAIT(stmt) do { stmt; } while (0)
#define CLX_TL_BEGIN() do {} while (0)
#define CLX_TL_END(kid, wave) do {} while (0)
#define CLX_TL_END_SEQ(kid, tag) do {} while (0)
#endif

#endif

#endif
