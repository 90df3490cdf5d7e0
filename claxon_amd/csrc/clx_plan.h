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

#endif
