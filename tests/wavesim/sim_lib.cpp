// TEST INFRASTRUCTURE: runs the production kernel source under the wave simulator.
#include <vector>
#include <algorithm>
#include <cstring>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
extern "C" { uint64_t sim_stats[64]; }
static const void* sim_kernargs = nullptr;       // where CLX_KERNARGS finds the running kernel's argument block
#define CLX_STAT(i, n) (sim_stats[i] += (uint64_t)(n))
#define CLN_POOL_SPIN 3u        // (lanes run one at a time here: a scan that is not over when a decode ticket asks will not be over later)
#include "clx_kernels.hip"
#include "clx_lanes.hip"
#include "clx_lean.hip"
#include "clx_plan.h"

// The lane path of one planned batch: the plan and ONE set of scratch (what a flight of the library holds), used by consecutive
// runs -- sf_start / errkey as clx_k_finalize leaves them, the groups' marks and the CRC parts tagged with generation numbers, the
// run's own slot maps that clx_k_compose re-deals run by run.
struct SimLanes {
    std::vector<clx_dev_frame> dev;
    uint64_t n_slots = 0;
    size_t n = 0, n_multi = 0, n_windows = 0;
    uint32_t flags = 0, gen = 6u;
    std::vector<uint32_t> slot_frame, multi, sf_start, errkey, taken, crc_todo, first_slot, fkey, slot_frame_plan, first_slot_plan;
    std::vector<uint64_t> endbits;
    std::vector<clx_crc_part> crc_part;
    std::vector<clx_window> windows;
    std::vector<int32_t> planar;       // narrow output: the general kernels' staging rows of the run (64 per workgroup, clx_lanes_group)
    uint32_t stage_stride = 0;
    bool lean = false;
    bool plan(const clx_frame_desc* frames, size_t n_, const uint64_t* out_offs, size_t arena_len, uint32_t flags_) {
        n = n_; flags = flags_;
        dev.assign(n ? n : 1, clx_dev_frame{});
        if (clx_plan_frames(frames, n, out_offs, dev.data(), &n_slots) >= 0) return false;
        clx_plan_limits(frames, n, arena_len, dev.data());
        slot_frame.assign(n_slots ? n_slots : 1, 0u); multi.assign(n ? n : 1, 0u);
        sf_start.assign(n_slots ? n_slots : 1, 0xffffffffu); errkey.assign(n ? n : 1, 0xffffffffu); endbits.assign(n ? n : 1, 0);
        n_multi = clx_plan_lanes(dev.data(), n, n_slots, slot_frame.data(), multi.data());
        taken.assign(2 * ((n_slots + 63) / 64) + 2, 0u);    // (the groups' marks, then the list clx_k_left makes: a count and the groups)
        lean = (flags & CLX_LANES_FUSED) && !(flags & CLX_LANES_GENERAL);
        crc_part.assign(n_slots ? n_slots : 1, clx_crc_part{});
        crc_todo.assign(n ? n : 1, 0xa5a5a5a5u);
        first_slot.assign(n ? n : 1, 0u); fkey.assign(n ? n : 1, 0u);
        for (size_t i = 0; i < n; ++i) first_slot[i] = dev[i].first_slot;
        windows.assign(n ? n : 1, clx_window{});
        const int cmode = (!lean || (flags & CLX_NO_COMPOSE)) ? -1 : (flags & CLX_COMPOSE) ? 1 : 0;
        n_windows = clx_plan_windows(dev.data(), n, cmode, windows.data());
        slot_frame_plan = slot_frame; first_slot_plan = first_slot;      // (what must stay untouched)
        if (flags & (CLX_OUT_PCM16 | CLX_OUT_PCM24)) {
            if (!lean) return false;
            uint32_t bs_max = 1;
            for (size_t i = 0; i < n; ++i) bs_max = std::max<uint32_t>(bs_max, frames[i].block_size);
            stage_stride = (bs_max + 3u) & ~3u;
        }
        return true;
    }
    // the run that decodes `arena` into `out` with this set of scratch, as launch_pending makes it (clx_api.hip): a new generation
    // number, the wrap handled as the library does
    clx_run make_run(const uint8_t* arena, size_t arena_len, int32_t* out, clx_frame_result* results) {
        const uint64_t alloc_len = ((uint64_t)arena_len + 15ull) & ~15ull;
        if (++gen == 0u) {       // (the generation number wrapped: nothing stale may look current)
            std::fill(taken.begin(), taken.begin() + (n_slots + 63) / 64, 0u);      // (the marks; the list behind them is empty between runs)
            memset(crc_part.data(), 0, crc_part.size() * sizeof(clx_crc_part));
            gen = 1u;
        }
        clx_run R;
        memset(&R, 0, sizeof R);
        R.arena = arena; R.alloc_len = alloc_len + 16; R.out = out; R.sf_start = sf_start.data();
        R.errkey = errkey.data(); R.end_bits = endbits.data(); R.taken = lean ? taken.data() : nullptr;
        R.results = results; R.gen = gen;
        R.crc_part = crc_part.data(); R.crc_todo = crc_todo.data();
        R.flags = ((flags & CLX_VERIFY_CRC16) ? CLX_RUN_CRC : 0u) | ((flags & CLX_OUT_PCM16) ? CLX_RUN_PCM16 : 0u) | ((flags & CLX_OUT_PCM24) ? CLX_RUN_PCM24 : 0u);
        R.planar = nullptr;            // (narrow output: handed out with the general kernels' grid, as the library does)
        // the run's slot maps: the plan's, or its own when waves are composed by content (as the library does: clx_plan_windows)
        R.slot_frame = slot_frame.data(); R.first_slot = first_slot.data(); R.fkey = n_windows ? fkey.data() : nullptr;
        return R;
    }
    // the library's rule (launch_lanes): the scan and the 16-bit tier as clx_k_pool's tickets unless the waves are composed by content
    bool pooled() const { return lean && (flags & CLX_POOL) && !(flags & CLX_OUT_PCM24) && !(n_windows && n_multi); }
    // one run by itself: a launch of one run
    int run(const uint8_t* arena, size_t arena_len, int32_t* out, clx_frame_result* results) {
        clx_runs runs;
        memset(&runs, 0, sizeof runs);
        runs.r[0] = make_run(arena, arena_len, out, results);
        if (pooled()) {      // (tickets in their own order; two workgroups: the second finds the counter run out)
            clx_pool_state ps;
            memset(&ps, 0, sizeof ps);
            clx_pool_args A;
            memset(&A, 0, sizeof A);
            A.runs = runs; A.frames = dev.data(); A.multi = multi.data(); A.ps = &ps; A.order = nullptr; A.dump_all = nullptr;
            A.n_runs = 1u; A.n_slots = (uint32_t)n_slots; A.n_multi = (uint32_t)n_multi;
            sim_kernargs = &A;
            SIM_LAUNCH(clx_k_pool, 2, 64, A);
            sim_kernargs = nullptr;
            if (ps.stuck != 0u) return CLX_API_ERROR;
            sim_stats[31] += 1;      // (launches through clx_k_pool)
        }
        return back(runs, true);
    }
    // what follows the pool's tickets, or the whole launch when there is no pool: `front_done`: the scan and clx_k_lean have run
    int back(const clx_runs& runs, bool maybe_front_done) {
        const bool front_done = maybe_front_done && pooled();
        clx_frame_result* const results = runs.r[0].results;
        const uint8_t* const arena = runs.r[0].arena;
        int32_t* const out = runs.r[0].out;
        const uint64_t alloc_len = runs.r[0].alloc_len - 16;
        if (n_multi && !front_done) {
            if (flags & CLX_LANES_GENERAL) SIM_LAUNCH(clx_k_scan_general, (n_multi + 63) / 64, 64, runs, dev.data(), multi.data(), (uint32_t)n_multi);
            else SIM_LAUNCH(clx_k_scan, (n_multi + 63) / 64, 64, runs, dev.data(), multi.data(), (uint32_t)n_multi);
        }
        // CLX_LANES_FUSED: the fused kernels; otherwise the two-wave one
        if (flags & CLX_LANES_FUSED) {
            std::vector<int32_t> dump(((n_slots + 127) / 128) * 128 * 32 + 16);
            if (n_windows && n_multi) {
                SIM_LAUNCH(clx_k_compose, n_windows, CLX_COMPOSE_THREADS, runs, windows.data());
                sim_stats[48] += n_windows;
                // what it dealt is a permutation of the plan's pairs inside each window, and nothing outside them moved
                std::vector<uint8_t> seen(n ? n : 1, 0);
                for (size_t wi = 0; wi < n_windows; ++wi)
                    for (uint32_t f = windows[wi].f_lo; f < windows[wi].f_hi; ++f) {
                        const uint32_t s0 = first_slot[f];
                        if (s0 < windows[wi].s_lo || s0 + 1u >= windows[wi].s_lo + 2u * (windows[wi].f_hi - windows[wi].f_lo) || ((s0 - windows[wi].s_lo) & 1u) ||
                            slot_frame[s0] != f || slot_frame[s0 + 1u] != f) return CLX_API_ERROR;
                        seen[f] = 1;
                    }
                for (size_t i = 0; i < n; ++i) if (!seen[i] && first_slot[i] != first_slot_plan[i]) return CLX_API_ERROR;
                for (uint64_t sl = 0; sl < n_slots; ++sl) {
                    const uint32_t f = slot_frame[sl];
                    if (f == 0xffffffffu ? slot_frame_plan[sl] != 0xffffffffu : (sl < first_slot[f] || sl >= first_slot[f] + dev[f].n_channels)) return CLX_API_ERROR;
                }
            }
            // the lean kernel first (it marks the groups it decodes with this run's generation number), unless the caller
            // asks for the general kernels alone (CLX_LANES_GENERAL: the pre-round-3 form, kept as a test target)
            if (lean && !front_done && !(flags & CLX_OUT_PCM24)) SIM_LAUNCH(clx_k_lean, (n_slots + 63) / 64, 64, runs, dev.data(), (uint32_t)n_slots, dump.data());
            for (size_t gi = 0; gi < (n_slots + 63) / 64; ++gi) sim_stats[52] += taken[gi] == runs.r[0].gen;
            if (lean) {     // the split tier on what is left (the library launches it when the batch holds frames of more than 16 bits)
                uint64_t before = 0, after = 0;
                for (size_t gi = 0; gi < (n_slots + 63) / 64; ++gi) before += taken[gi] == runs.r[0].gen;
                SIM_LAUNCH(clx_k_lean24, (n_slots + 63) / 64, 64, runs, dev.data(), (uint32_t)n_slots, dump.data());
                for (size_t gi = 0; gi < (n_slots + 63) / 64; ++gi) after += taken[gi] == runs.r[0].gen;
                sim_stats[13] += after - before;
            }
            // behind the tiers the general kernels loop over the list of groups that were left, with a grid smaller than the list
            // is long whenever it can be (a third of what is left: every workgroup takes several groups)
            size_t ggrid = (n_slots + 63) / 64;
            if (lean) {
                SIM_LAUNCH(clx_k_left, (ggrid + 255) / 256, 256, runs, (uint32_t)ggrid, (uint32_t*)nullptr, (uint32_t*)nullptr);
                const uint32_t n_left = taken[ggrid];
                sim_stats[49] += n_left;
                if (n_left > ggrid) return CLX_API_ERROR;
                ggrid = n_left >= 3 ? n_left / 3 : 1;
            }
            clx_runs gruns = runs;
            if (flags & (CLX_OUT_PCM16 | CLX_OUT_PCM24)) {       // (the staging rows: 64 per workgroup, stale between launches)
                planar.assign(ggrid * 64 * (size_t)stage_stride + 16, 0x2b2b2b2b);
                gruns.r[0].planar = planar.data();
                gruns.r[0].flags |= CLX_RUN_STAGE_BITS(stage_stride);
            }
            SIM_LAUNCH(clx_k_lanes, ggrid, 64, gruns, dev.data(), (uint32_t)n_slots, dump.data());
            SIM_LAUNCH(clx_k_lanes_hi, ggrid, 64, gruns, dev.data(), (uint32_t)n_slots, dump.data());
        } else {
            std::vector<int32_t> dump(((n_slots + 127) / 128) * 128 * 16 + 16);
            SIM_LAUNCH(clx_k_lanes2, (n_slots + 127) / 128, 256, arena, alloc_len + 16, dev.data(), slot_frame.data(), (uint32_t)n_slots, sf_start.data(), out,
                       errkey.data(), endbits.data(), dump.data());
        }
        SIM_LAUNCH(clx_k_finalize, (n + 255) / 256, 256, runs, dev.data(), (uint32_t)n, (uint32_t)n_slots);
        if (lean && taken[(n_slots + 63) / 64] != 0u) return CLX_API_ERROR;      // (the list is left empty for the next run)
        // (the scratch is left ready for a next run)
        for (size_t i = 0; i < n; ++i) if (errkey[i] != 0xffffffffu) return CLX_API_ERROR;
        if ((flags & CLX_LANES_FUSED)) for (uint64_t sl = 0; sl < n_slots; ++sl) if (sf_start[sl] != 0xffffffffu) return CLX_API_ERROR;
        if (flags & CLX_VERIFY_CRC16) {
            // (how many frames the lean kernels' lanes settled themselves, how many the stand-alone kernel has to check)
            for (size_t i = 0; i < n; ++i) { if (crc_todo[i] > 1u) return CLX_API_ERROR; sim_stats[14 + crc_todo[i]] += 1;
                if (getenv("SIM_CRC_DEBUG") && crc_todo[i]) { fprintf(stderr, "todo frame %zu st %d:", i, results[i].status); for (uint32_t c = 0; c < dev[i].n_channels; ++c) { const clx_crc_part& q = crc_part[dev[i].first_slot + c]; fprintf(stderr, " [gen %u rx %08x da %u db %u]", q.gen, q.rx, q.da, q.db); } fprintf(stderr, " endbit %llu limit %u off %llu\n", (unsigned long long)results[i].end_bit, dev[i].limit_bits, (unsigned long long)dev[i].byte_off); } }
            SIM_LAUNCH(clx_k_crc16_runs, (n + 3) / 4, 256, runs, dev.data(), (uint32_t)n);
        }
        return CLX_OK;
    }
};

// Consecutive runs of ONE planned batch on ONE set of scratch (the multi-run state of clx_batch_submit: generation-tagged marks and
// CRC parts, slot maps re-dealt run by run, scratch left cleared by clx_k_finalize): run r decodes arenas[r] (all of arena_len
// bytes, the same frame layout -- what differs is damage) into outs[r] / results[r].  first_gen: the generation number of run 0
// (0xffffffff makes run 1 wrap).
extern "C" int sim_decode_frames_runs(const uint8_t* const* arenas, size_t arena_len, size_t n_runs, const clx_frame_desc* frames, size_t n,
                                      int32_t* const* outs, const uint64_t* out_offs, clx_frame_result* const* results, uint32_t flags, uint32_t first_gen) {
    if (!(flags & CLX_PATH_LANES)) return CLX_API_ERROR;
    SimLanes L;
    if (!L.plan(frames, n, out_offs, arena_len, flags)) return CLX_API_ERROR;
    L.gen = first_gen - 1u;
    for (size_t r = 0; r < n_runs; ++r) {
        const int st = L.run(arenas[r], arena_len, outs[r], results[r]);
        if (st != CLX_OK) return st;
    }
    return CLX_OK;
}

// ONE merged launch of n_runs (<= CLX_MAX_MERGE) runs of one planned batch through clx_k_pool, every run on a scratch set of its own
// (the flights of the library), the tickets taken in the order `order` gives (a permutation of [0, tickets); null: as they come) by
// `workers` workgroups one after the other; then every run's kernels behind the pool.  stuck_out: decode tickets that gave up
// waiting for their run's scan (taken before it in `order`: their groups must come out right all the same, through the general kernels).
extern "C" int sim_decode_frames_pool(const uint8_t* const* arenas, size_t arena_len, size_t n_runs, const clx_frame_desc* frames, size_t n,
                                      int32_t* const* outs, const uint64_t* out_offs, clx_frame_result* const* results, uint32_t flags,
                                      const uint32_t* order, uint32_t workers, uint32_t* stuck_out) {
    if (!(flags & CLX_PATH_LANES) || n_runs < 1 || n_runs > CLX_MAX_MERGE) return CLX_API_ERROR;
    std::vector<SimLanes> L(n_runs);
    clx_runs runs;
    memset(&runs, 0, sizeof runs);
    for (size_t r = 0; r < n_runs; ++r) {
        if (!L[r].plan(frames, n, out_offs, arena_len, flags)) return CLX_API_ERROR;
        L[r].gen = 40u + (uint32_t)r;
        runs.r[r] = L[r].make_run(arenas[r], arena_len, outs[r], results[r]);
    }
    if (!L[0].pooled()) return CLX_API_ERROR;
    clx_pool_state ps;
    memset(&ps, 0, sizeof ps);
    clx_pool_args A;
    memset(&A, 0, sizeof A);
    A.runs = runs; A.frames = L[0].dev.data(); A.multi = L[0].multi.data(); A.ps = &ps; A.order = order; A.dump_all = nullptr;
    A.n_runs = (uint32_t)n_runs; A.n_slots = (uint32_t)L[0].n_slots; A.n_multi = (uint32_t)L[0].n_multi;
    sim_kernargs = &A;
    SIM_LAUNCH(clx_k_pool, workers ? workers : 1u, 64, A);
    sim_kernargs = nullptr;
    if (stuck_out) *stuck_out = ps.stuck;
    const uint32_t scan_w = (uint32_t)((L[0].n_multi + 63) / 64);
    for (size_t r = 0; r < n_runs; ++r) if (ps.scan_done[r] != scan_w) return CLX_API_ERROR;
    for (size_t r = 0; r < n_runs; ++r) {
        clx_runs one;
        memset(&one, 0, sizeof one);
        one.r[0] = runs.r[r];
        const int st = L[r].back(one, true);
        if (st != CLX_OK) return st;
    }
    return CLX_OK;
}

// flags: the ABI's (claxon_hip.h) and nothing else.  stop_after_k1 (the harness's own switch -- a flag bit of its own until round 6,
// where it collided with CLX_K2_LATENCY): residual inspection, the wave path only.
// The wave path's predictor build: CLX_K2_LATENCY / CLX_K2_THROUGHPUT as the library takes them (launch_waves, clx_api.hip); with
// neither, every build gets its turn by the parity of the slot count (the library would go by the batch's size).
extern "C" int sim_decode_frames(const uint8_t* arena, size_t arena_len, const clx_frame_desc* frames, size_t n,
                                 int32_t* out, const uint64_t* out_offs, clx_frame_result* results, uint32_t flags,
                                 clx_sf_desc* sfd_out /* optional, n_slots entries */, uint64_t* n_slots_out, int stop_after_k1) {
    if ((flags & CLX_K2_LATENCY) && (flags & CLX_K2_THROUGHPUT)) return CLX_API_ERROR;
    if ((flags & (CLX_K2_LATENCY | CLX_K2_THROUGHPUT)) && (flags & CLX_PATH_LANES)) return CLX_API_ERROR;      // (a build of the WAVE path's predictor)
    if (stop_after_k1 && (flags & CLX_PATH_LANES)) return CLX_API_ERROR;
    if (flags & CLX_PATH_LANES) {
        SimLanes L;
        if (!L.plan(frames, n, out_offs, arena_len, flags)) return CLX_API_ERROR;
        if (n_slots_out) *n_slots_out = L.n_slots;
        return L.run(arena, arena_len, out, results);
    }
    std::vector<clx_dev_frame> dev(n ? n : 1);
    uint64_t n_slots = 0;
    if (clx_plan_frames(frames, n, out_offs, dev.data(), &n_slots) >= 0) return CLX_API_ERROR;
    clx_plan_limits(frames, n, arena_len, dev.data());
    std::vector<clx_sf_desc> sfd(n_slots ? n_slots : 1);
    memset(sfd.data(), 0, sfd.size() * sizeof(clx_sf_desc));
    const uint64_t alloc_len = ((uint64_t)arena_len + 15ull) & ~15ull;
    SIM_LAUNCH(clx_k_residual, n, 64, arena, alloc_len, dev.data(), (uint32_t)n, out, sfd.data(), results);
    if (n_slots_out) *n_slots_out = n_slots;
    if (sfd_out) memcpy(sfd_out, sfd.data(), n_slots * sizeof(clx_sf_desc));
    if (stop_after_k1) return CLX_OK;      // (residual inspection)
    std::vector<int32_t> dump(((n_slots + 127) / 128) * 128 * 16 + 16);
    bool all_narrow_aligned = true;        // (the library's rule for what follows clx_k_predict16: batch_plan_)
    for (size_t i = 0; i < n; ++i) all_narrow_aligned = all_narrow_aligned && frames[i].bps <= 16 && (frames[i].block_size & 3u) == 0u && (out_offs[i] & 3ull) == 0ull;
    const bool one_wave = (flags & CLX_K2_THROUGHPUT) ? true : (flags & CLX_K2_LATENCY) ? false : (n_slots & 1) != 0;
    const bool behind16_1w = (flags & CLX_K2_LATENCY) ? all_narrow_aligned : (n_slots & 2) != 0;
    // (without a flag every build of K2 is exercised: the one-wave one for odd slot counts, the multi-wave ones for even ones)
    if (one_wave) {
        SIM_LAUNCH(clx_k_predict_1w, (n_slots + 63) / 64, 64, out, sfd.data(), (uint32_t)n_slots, dump.data(), 0u);
        SIM_LAUNCH(clx_k_predict_1w_hi, (n_slots + 63) / 64, 64, out, sfd.data(), (uint32_t)n_slots, dump.data(), 0u);
    }
    else {
        // the fast kernel first; what it leaves goes to the general kernel, or (slot counts 2 mod 4) to the one-wave kernels
        // told to skip its groups -- the library does the latter when every frame is 16-bit and aligned
        SIM_LAUNCH(clx_k_predict16, (n_slots + 63) / 64, 256, out, sfd.data(), (uint32_t)n_slots, dump.data());
        if (behind16_1w) {
            SIM_LAUNCH(clx_k_predict_1w, (n_slots + 63) / 64, 64, out, sfd.data(), (uint32_t)n_slots, dump.data(), 1u);
            SIM_LAUNCH(clx_k_predict_1w_hi, (n_slots + 63) / 64, 64, out, sfd.data(), (uint32_t)n_slots, dump.data(), 1u);
        }
        else SIM_LAUNCH(clx_k_predict, (n_slots + 127) / 128, 512, out, sfd.data(), (uint32_t)n_slots, dump.data());
    }
    if (flags & CLX_VERIFY_CRC16)
        SIM_LAUNCH(clx_k_crc16, (n + 3) / 4, 256, arena, dev.data(), (uint32_t)n, results);
    return CLX_OK;
}

// the library's kernel selection rule (clx_plan.h), for tests/test_select_path.py: returns lanes | lanes_split << 1
extern "C" int sim_select_path(uint64_t slots, uint64_t samples, uint64_t bytes, int heavy, int all_mono, int pipelined) {
    const clx_path_choice c = clx_select_path(slots, samples, bytes, heavy != 0, all_mono != 0, pipelined != 0);
    return (c.lanes ? 1 : 0) | (c.lanes_split ? 2 : 0);
}

extern "C" int sim_interleave(const int32_t* planar, const clx_frame_desc* frames, size_t n, const uint64_t* out_offs,
                              const clx_frame_result* results, uint8_t* pcm, uint32_t sample_bytes) {
    std::vector<clx_dev_frame> dev(n ? n : 1);
    uint64_t n_slots = 0;
    if (clx_plan_frames(frames, n, out_offs, dev.data(), &n_slots) >= 0) return CLX_API_ERROR;
    if (n) SIM_LAUNCH(clx_k_interleave, n, 256, planar, dev.data(), results, (uint32_t)n, pcm, sample_bytes);
    return CLX_OK;
}

// frame indexer kernels (K5, K6) under simulation; `data` must be 16-byte aligned and padded by >= 32 bytes
extern "C" int sim_find_headers(const uint8_t* data, uint64_t len, uint64_t start, uint64_t* cand, uint32_t cap, uint32_t* count) {
    *count = 0;
    if (start >= len) return CLX_OK;
    const uint64_t scan0 = start & ~15ull;
    const uint64_t n_threads = (len - scan0 + 15) / 16;
    SIM_LAUNCH(clx_k_find_headers, (n_threads + 255) / 256, 256, data, len, start, cand, cap, count);
    return CLX_OK;
}
extern "C" int sim_span_crc16(const uint8_t* data, const uint64_t* pos, uint32_t n_spans, uint16_t* crc) {
    if (n_spans) SIM_LAUNCH(clx_k_span_crc16, n_spans, 64, data, pos, n_spans, crc);
    return CLX_OK;
}
