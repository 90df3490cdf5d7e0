// clx_plan.h -- host-side batch planning shared by the library (clx_api.hip) and the CPU wave
// simulator used in tests: turns the caller's clx_frame_desc list into clx_dev_frame records.
#ifndef CLX_PLAN_H
#define CLX_PLAN_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../../include/claxon_hip.h"
#include "clx_device.h"

// Returns -1 on success, else the index of the first invalid descriptor.
// Predictor slots: one per subframe, in stream order; a stereo-decorrelated frame starts on an even
// slot so that K2 finds the partner channel at row^1 of the same wave.
static inline long clx_plan_frames(const clx_frame_desc* frames, size_t n, const uint64_t* out_sample_offsets,
                                   clx_dev_frame* dev, uint64_t* n_slots) {
    uint64_t slot = 0;
    for (size_t i = 0; i < n; ++i) {
        const clx_frame_desc* f = &frames[i];
        if (f->n_channels < 1 || f->n_channels > 8 || f->block_size == 0 || f->channel_assignment > CLX_CH_MID_SIDE ||
            (f->channel_assignment != CLX_CH_INDEPENDENT && f->n_channels != 2) || f->bps == 0 || f->bps > 32)
            return (long)i;
        clx_dev_frame* d = &dev[i];
        memset(d, 0, sizeof *d);
        d->byte_off = f->byte_off;
        d->out_off = out_sample_offsets[i];
        if (f->channel_assignment != CLX_CH_INDEPENDENT && (slot & 1ull)) ++slot;
        d->first_slot = (uint32_t)slot;
        slot += f->n_channels;
        d->header_bytes = f->header_bytes;
        d->block_size = f->block_size;
        d->n_channels = f->n_channels;
        d->channel_assignment = f->channel_assignment;
        d->bps = f->bps;
        d->flags = (uint8_t)(f->reserved[0] & 1u);
        d->limit_bits = f->max_bytes;            // bytes for now; clx_plan_limits converts to clamped bits
    }
    *n_slots = slot;
    return -1;
}

// Which kernels decode a batch when the caller does not say (measured on MI355X: tools/bench_configs.py).
// The wave-per-frame kernels scale with the work in the batch; the lane-per-subframe kernels last as long as one lane's serial
// chain however few subframes there are, and win once the batch fills the machine -- earlier the more work a subframe carries:
//   24-bit audio (order-32 predictors, Rice2: the wave path's slow cases)            lanes from ~3 000 subframes
//   16-bit, ~9.5 compressed bits per sample (mixed real-world shapes, config 5)      lanes from ~20 000 subframes
//   16-bit, ~5 bits per sample (configs 2 and 3)                                     lanes from ~52 000 subframes (28 000 when mono:
//                                                                                    no scan pass for the later channels)
// and between the two lane builds the two-wave one while its workgroups still get a CU each (longer for 24-bit audio).
// `pipelined`: the batch is one of several in flight (clx_batch_submit).  There up to twelve runs of the fused lane kernels go out
// as ONE grid (round 3: merged launches), and that form is ahead of the wave kernels' four in flight at every size measured
// (tools/path_threshold_sweep.sh, profiles/r03_path_sweep.txt: 600 .. 5 000 frames of configs 2 and 3, 3.6x .. 4.4x; round 2's
// unmerged runs had lost to the wave kernels below ~10 000 subframes of short codes): the lane kernels always.
// `bytes` = sum of the frames' max_bytes when those are real frame lengths (0 = unknown, e.g. "to the end of the stream").
struct clx_path_choice { bool lanes, lanes_split; };
static inline clx_path_choice clx_select_path(uint64_t slots, uint64_t samples, uint64_t bytes, bool heavy, bool all_mono, bool pipelined = false) {
    clx_path_choice c;
    double rate = (samples && bytes) ? 8.0 * (double)bytes / (double)samples : 7.5;      // compressed bits per sample
    if (rate > 32.0) rate = 7.5;                                                          // not frame lengths: unknown
    double threshold;
    if (pipelined) threshold = 0.0;
    else if (heavy) threshold = 3000.0;                    // (a quarter or more of the samples are wider than 16 bits)
    else {
        // at <= 6.5 / >= 8.5 bits per sample
        const double lo = all_mono ? 28000.0 : 52000.0;
        const double hi = all_mono ? 14000.0 : 20000.0;
        const double t = rate <= 6.5 ? 0.0 : rate >= 8.5 ? 1.0 : (rate - 6.5) / 2.0;
        threshold = lo + (hi - lo) * t;
    }
    c.lanes = (double)slots >= threshold;
    c.lanes_split = slots <= 40000 || (heavy && slots <= 80000);
    return c;
}

// limit_bits = 8 * min(max_bytes, arena_len - byte_off), capped at 2^31 bits (no frame needs more).
static inline void clx_plan_limits(const clx_frame_desc* frames, size_t n, size_t arena_len, clx_dev_frame* dev) {
    for (size_t i = 0; i < n; ++i) {
        uint64_t avail = frames[i].byte_off < arena_len ? (uint64_t)arena_len - frames[i].byte_off : 0;
        uint64_t mb = frames[i].max_bytes < avail ? frames[i].max_bytes : avail;
        if (mb > (1ull << 28)) mb = 1ull << 28;
        dev[i].limit_bits = (uint32_t)(mb * 8ull);
    }
}

// Lane path: slot -> frame map (0xffffffff for alignment padding) and the list of multi-channel frames.
// slot_frame must hold n_slots entries, multi up to n; returns the number of multi-channel frames.
static inline size_t clx_plan_lanes(const clx_dev_frame* dev, size_t n, uint64_t n_slots, uint32_t* slot_frame, uint32_t* multi) {
    for (uint64_t s = 0; s < n_slots; ++s) slot_frame[s] = 0xffffffffu;
    size_t n_multi = 0;
    for (size_t i = 0; i < n; ++i) {
        for (uint32_t c = 0; c < dev[i].n_channels; ++c) slot_frame[dev[i].first_slot + c] = (uint32_t)i;
        if (dev[i].n_channels > 1) multi[n_multi++] = (uint32_t)i;
    }
    return n_multi;
}

// How many workgroups per run the general lane kernels get BEHIND the lean tiers, where they loop over the list of groups the tiers
// left (clx_k_left).  What the descriptors say about a group of 64 slots is what the tiers ask first -- at most 24 bits, one block
// size that is a multiple of 16 and at least 32 (64 beyond 16 bits), rows that start on 16 bytes (taking the output's base as
// aligned) -- and a group that fails there is left for certain; what only the stream says (an order beyond the tier's, a header
// that does not parse, a wave that gives its group up) is rare: an eighth of the groups is held ready for it.  (Waves composed by
// content hold frames of one block size and width class: the count is the same in stream order.)
// flags: the batch's (CLX_OUT_PCM16: the 16-bit tier writes narrow output for waves of stereo frames whose channel c sits in a lane of
// parity c and whose blocks start on 16 bytes of int16, or of mono frames alone; CLX_OUT_PCM24: both tiers, stereo frames whose
// blocks start on 16 bytes -- everything else is left for certain too).
// sure_out (optional): the groups that are left for certain.
static inline unsigned clx_plan_general_grid(const clx_dev_frame* dev, const uint32_t* slot_frame, uint64_t n_slots, uint32_t flags = 0, uint64_t* sure_out = nullptr) {
    const uint64_t groups = (n_slots + 63) / 64;
    uint64_t sure = 0;
    for (uint64_t g = 0; g < groups; ++g) {
        bool left = false;
        uint32_t bs0 = 0, ch0 = 0;
        for (uint64_t s = 64 * g; s < 64 * g + 64 && s < n_slots && !left; ++s) {
            const uint32_t f = slot_frame[s];
            if (f == 0xffffffffu) continue;
            const clx_dev_frame& d = dev[f];
            const uint32_t bs = d.block_size;
            if (!bs0) { bs0 = bs; ch0 = d.n_channels; }
            const uint64_t row = d.out_off + (uint64_t)(s - d.first_slot) * bs;
            left = d.bps > 24u || bs != bs0 || (bs & 15u) != 0u || bs < (d.bps > 16u ? 64u : 32u) || (row & 3ull) != 0ull;
            if (flags & CLX_OUT_PCM24)                       // (packed 24-bit output: stereo frames whose blocks start on 16 bytes)
                left = left || d.n_channels != 2u || (d.out_off & 15ull) != 0ull || ((s - d.first_slot) & 1u) != (s & 1u);
            if (flags & CLX_OUT_PCM16)
                left = left || d.n_channels > 2u || d.n_channels != ch0 || (d.out_off & 7ull) != 0ull || (d.n_channels == 2u && ((s - d.first_slot) & 1u) != (s & 1u));
        }
        sure += left ? 1u : 0u;
    }
    if (sure_out) *sure_out = sure;
    const uint64_t want = 2 * sure + groups / 8 + 16;
    return (unsigned)(want < groups ? want : groups);
}

// Windows of clx_k_compose: maximal runs of consecutive stereo frames of one block size and one width class (<= 16 bits or not),
// cut every CLX_COMPOSE_WINDOW frames.  mode: 0 by content of the descriptors (a window is composed when it holds >= 256 frames and
// its channel assignments differ), 1 every such window, -1 none.  `win` must hold n entries; returns their number.
static inline size_t clx_plan_windows(const clx_dev_frame* dev, size_t n, int mode, clx_window* win) {
    size_t nw = 0;
    if (mode < 0) return 0;
    size_t i = 0;
    while (i < n) {
        if (dev[i].n_channels != 2) { ++i; continue; }
        size_t j = i;
        bool mixed = false;
        while (j < n && j - i < CLX_COMPOSE_WINDOW && dev[j].n_channels == 2 && dev[j].block_size == dev[i].block_size &&
               (dev[j].bps <= 16) == (dev[i].bps <= 16) && dev[j].first_slot == dev[i].first_slot + 2u * (uint32_t)(j - i)) {
            mixed = mixed || dev[j].channel_assignment != dev[i].channel_assignment;
            ++j;
        }
        if (j - i >= 64 && (dev[i].first_slot & 1u) == 0u && (mode > 0 || (mixed && j - i >= 256))) {       // (an even first slot: pairs stay pairs)
            win[nw].f_lo = (uint32_t)i; win[nw].f_hi = (uint32_t)j; win[nw].s_lo = dev[i].first_slot; win[nw].pad = 0;
            ++nw;
        }
        i = j;
    }
    return nw;
}

#endif
