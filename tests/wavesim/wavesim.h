// wavesim.h -- a tiny wave64 SIMT simulator on ucontext fibers (TEST INFRASTRUCTURE).
//
// One fiber per lane.  Cross-lane operations (__shfl*, __ballot, __any) and __syncthreads() are
// rendezvous points: a lane parks until every live lane of its wave (block) has arrived, then all
// exchange values.  Lanes that have returned from the kernel count as inactive, like exec-masked
// lanes.  A rendezvous reached from two different call sites at once -- i.e. a cross-lane operation
// under divergent control flow, which the kernels promise not to do -- or a lane that never arrives
// aborts with a diagnostic instead of silently "working".
#ifndef WAVESIM_H
#define WAVESIM_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __constant__ static const
#define __launch_bounds__(...)

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint2 { uint32_t x, y; } __attribute__((aligned(8)));
static inline uint2 make_uint2(uint32_t a, uint32_t b) { uint2 v; v.x = a; v.y = b; return v; }
struct uint4 { uint32_t x, y, z, w; } __attribute__((aligned(16)));
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { uint4 v; v.x = a; v.y = b; v.z = c; v.w = d; return v; }

namespace wavesim {

enum { kMaxThreads = 1024, kStack = 1 << 18 };

struct State {
    ucontext_t sched;
    ucontext_t fiber[kMaxThreads];
    char* stacks = nullptr;
    int nthreads = 0;
    int cur = -1;
    bool done[kMaxThreads];
    // rendezvous bookkeeping (per wave for cross-lane ops, per block for barriers)
    int wave_arrived[kMaxThreads / 64];
    unsigned wave_gen[kMaxThreads / 64];
    const void* wave_site[kMaxThreads / 64];
    int wave_live[kMaxThreads / 64];
    int blk_arrived = 0; unsigned blk_gen = 0; int blk_live = 0;
    uint64_t slot[2][kMaxThreads];      // exchange buffers, double buffered by generation parity
    unsigned long long progress = 0, spins = 0;
    void (*entry)(void*) = nullptr;
    void* entry_arg = nullptr;
};

inline State& S() { static State s; return s; }

struct Idx { unsigned x, y, z; };

}  // namespace wavesim

static wavesim::Idx threadIdx, blockIdx, blockDim, gridDim;

namespace wavesim {

inline void yield_() {
    State& s = S();
    int me = s.cur;
    swapcontext(&s.fiber[me], &s.sched);
}

inline void die(const char* what) { fprintf(stderr, "wavesim: %s (block %u, thread %d)\n", what, blockIdx.x, S().cur); abort(); }

// park until all live lanes of my wave arrived at the same site
inline unsigned wave_rendezvous(const void* site) {
    State& s = S();
    const int me = s.cur, w = me / 64;
    if (s.wave_arrived[w] == 0) s.wave_site[w] = site;
    else if (s.wave_site[w] != site) die("cross-lane operation reached from divergent control flow");
    const unsigned gen = s.wave_gen[w];
    if (++s.wave_arrived[w] >= s.wave_live[w]) { s.wave_arrived[w] = 0; ++s.wave_gen[w]; ++s.progress; }
    else while (s.wave_gen[w] == gen) yield_();
    return gen;
}

inline void lane_exit() {
    State& s = S();
    const int me = s.cur, w = me / 64;
    s.done[me] = true;
    --s.wave_live[w]; --s.blk_live;
    ++s.progress;
    if (s.wave_live[w] > 0 && s.wave_arrived[w] >= s.wave_live[w]) { s.wave_arrived[w] = 0; ++s.wave_gen[w]; }
    if (s.blk_live > 0 && s.blk_arrived >= s.blk_live) { s.blk_arrived = 0; ++s.blk_gen; }
}

inline void trampoline() {
    State& s = S();
    s.entry(s.entry_arg);
    lane_exit();
    yield_();
    die("resumed a finished lane");
}

// run one block of `nthreads` threads; entry(arg) is the kernel body bound to its arguments
inline void run_block(unsigned bx, unsigned grid_x, int nthreads, void (*entry)(void*), void* arg) {
    State& s = S();
    if (nthreads > kMaxThreads) die("block too large");
    if (!s.stacks) s.stacks = (char*)malloc((size_t)kMaxThreads * kStack);
    s.nthreads = nthreads; s.entry = entry; s.entry_arg = arg;
    blockIdx = { bx, 0, 0 }; blockDim = { (unsigned)nthreads, 1, 1 }; gridDim = { grid_x, 1, 1 };
    const int nw = (nthreads + 63) / 64;
    for (int w = 0; w < nw; ++w) { s.wave_arrived[w] = 0; s.wave_gen[w] = 0; s.wave_live[w] = (w == nw - 1 && nthreads % 64) ? nthreads % 64 : 64; }
    s.blk_arrived = 0; s.blk_gen = 0; s.blk_live = nthreads;
    for (int t = 0; t < nthreads; ++t) {
        s.done[t] = false;
        getcontext(&s.fiber[t]);
        s.fiber[t].uc_stack.ss_sp = s.stacks + (size_t)t * kStack;
        s.fiber[t].uc_stack.ss_size = kStack;
        s.fiber[t].uc_link = nullptr;
        makecontext(&s.fiber[t], (void (*)())trampoline, 0);
    }
    int live = nthreads;
    unsigned long long last_progress = s.progress; unsigned long long idle_rounds = 0;
    while (live > 0) {
        live = 0;
        for (int t = 0; t < nthreads; ++t) {
            if (s.done[t]) continue;
            ++live;
            s.cur = t; threadIdx = { (unsigned)t, 0, 0 };
            swapcontext(&s.sched, &s.fiber[t]);
        }
        if (s.progress == last_progress) { if (++idle_rounds > 4) { s.cur = -1; die("deadlock: lanes wait at a rendezvous that not every live lane reaches"); } }
        else { idle_rounds = 0; last_progress = s.progress; }
    }
    s.cur = -1;
}

template <typename T> inline uint64_t to_bits(T v) { uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template <typename T> inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }

// all-to-all exchange inside a wave: returns a pointer to the 64 published values
template <typename T> inline const uint64_t* exchange(T v, const void* site, uint64_t neutral = 0) {
    State& s = S();
    const int me = s.cur, w = me / 64;
    const unsigned par = s.wave_gen[w] & 1u;
    s.slot[par][me] = to_bits(v);
    // lanes that exited BEFORE this rendezvous publish the operation's neutral value (filled in once, by the first
    // arriver); a lane that exits after taking part keeps what it published
    if (s.wave_arrived[w] == 0) for (int l = 0; l < 64; ++l) if (w * 64 + l >= s.nthreads || s.done[w * 64 + l]) s.slot[par][w * 64 + l] = neutral;
    wave_rendezvous(site);
    return &s.slot[par][w * 64];
}

}  // namespace wavesim

// A rendezvous "site" is the source line of the call (not a machine address: the host compiler is free to
// duplicate code, which must not look like divergence).  Two lanes meeting from different lines = divergence.
#define WAVESIM_SITE(line) ((const void*)(uintptr_t)(line))

__attribute__((noinline)) static void __syncthreads() {
    using namespace wavesim;
    State& s = S();
    const unsigned gen = s.blk_gen;
    if (++s.blk_arrived >= s.blk_live) { s.blk_arrived = 0; ++s.blk_gen; ++s.progress; }
    else while (s.blk_gen == gen) yield_();
}

namespace wavesim {
template <typename T> static T shfl_(int line, T v, int src, int width = 64) {
    (void)width;
    const uint64_t* all = exchange(v, WAVESIM_SITE(line));
    return from_bits<T>(all[src & 63]);
}
template <typename T> static T shfl_up_(int line, T v, unsigned delta, int width = 64) {
    (void)width;
    const int lane = S().cur & 63;
    const uint64_t* all = exchange(v, WAVESIM_SITE(line));
    return lane >= (int)delta ? from_bits<T>(all[lane - (int)delta]) : v;
}
template <typename T> static T shfl_xor_(int line, T v, int mask, int width = 64) {
    (void)width;
    const int lane = S().cur & 63;
    const uint64_t* all = exchange(v, WAVESIM_SITE(line));
    return from_bits<T>(all[(lane ^ mask) & 63]);
}
static unsigned long long ballot_(int line, int pred) {
    const uint64_t* all = exchange<uint32_t>(pred ? 1u : 0u, WAVESIM_SITE(line));
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) if (all[l]) m |= 1ull << l;
    return m;
}
static int any_(int line, int pred) {
    const uint64_t* all = exchange<uint32_t>(pred ? 1u : 0u, WAVESIM_SITE(line));
    for (int l = 0; l < 64; ++l) if (all[l]) return 1;
    return 0;
}
static int all_(int line, int pred) {
    // inactive (exited) lanes do not veto: their neutral value is 1
    const uint64_t* all = exchange<uint32_t>(pred ? 1u : 0u, WAVESIM_SITE(line), 1);
    for (int l = 0; l < 64; ++l) if (!all[l]) return 0;
    return 1;
}
// DPP controls the kernels use, with gfx9 semantics: quad_perm (0x00-0xFF): lane l reads lane (l & ~3) | perm[l & 3];
// row_shr:n (0x111-0x11F): lane l reads lane l-n of its row of 16, lanes without a source get 0 (bound_ctrl) or `old`;
// row_bcast:15 (0x142) / row_bcast:31 (0x143): lane 15 of the previous row / lane 31 of the previous half goes to a whole
// row.  Rows not selected by row_mask keep `old`.  (bank_mask must be 0xF.)
static int update_dpp_(int line, int old, int src, int dpp_ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    if (bank_mask != 0xf) die("only bank_mask 0xF is simulated");
    const int lane = S().cur & 63;
    const uint64_t* all = exchange(src, WAVESIM_SITE(line));
    int from = -1;                                       // source lane, -1: none
    if (dpp_ctrl >= 0 && dpp_ctrl <= 0xff) from = (lane & ~3) | ((dpp_ctrl >> (2 * (lane & 3))) & 3);
    else if (dpp_ctrl >= 0x111 && dpp_ctrl <= 0x11f) { const int n = dpp_ctrl - 0x110; from = (lane & 15) >= n ? lane - n : -1; }
    else if (dpp_ctrl == 0x138) from = lane - 1;                                    // wave_shr:1 (GFX9)
    else if (dpp_ctrl == 0x142) from = (lane >= 16) ? ((lane & ~15) - 1) : -1;
    else if (dpp_ctrl == 0x143) from = (lane >= 32) ? 31 : -1;
    else die("this DPP control is not simulated");
    if (!((row_mask >> (lane >> 4)) & 1)) return old;
    if (from < 0) return bound_ctrl ? 0 : old;
    return from_bits<int>(all[from]);
}
}  // namespace wavesim

#define __shfl(...) wavesim::shfl_(__LINE__, __VA_ARGS__)
#define __shfl_up(...) wavesim::shfl_up_(__LINE__, __VA_ARGS__)
#define __shfl_xor(...) wavesim::shfl_xor_(__LINE__, __VA_ARGS__)
#define __ballot(...) wavesim::ballot_(__LINE__, __VA_ARGS__)
#define __any(...) wavesim::any_(__LINE__, __VA_ARGS__)
#define __all(...) wavesim::all_(__LINE__, __VA_ARGS__)
#define __builtin_amdgcn_update_dpp(...) wavesim::update_dpp_(__LINE__, __VA_ARGS__)

static inline unsigned atomicMin(unsigned* p, unsigned v) { unsigned old = *p; if (v < old) *p = v; return old; }   // lanes run one at a time
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned old = *p; *p = old + v; return old; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned old = *p; if (v > old) *p = v; return old; }
static inline int __mul24(int a, int b) { return (int)((unsigned)((a << 8) >> 8) * (unsigned)((b << 8) >> 8)); }
struct int4 { int x, y, z, w; } __attribute__((aligned(16)));
static inline int4 make_int4(int a, int b, int c, int d) { int4 v; v.x = a; v.y = b; v.z = c; v.w = d; return v; }
struct int2 { int x, y; } __attribute__((aligned(8)));
static inline int2 make_int2(int a, int b) { int2 v; v.x = a; v.y = b; return v; }

static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }

// SIM_LAUNCH(kernel, grid, block, args...)
#define SIM_LAUNCH(kernel, grid, block, ...)                                                     \
    do {                                                                                          \
        auto body_ = [&]() { kernel(__VA_ARGS__); };                                              \
        using Body_ = decltype(body_);                                                            \
        for (unsigned bx_ = 0; bx_ < (unsigned)(grid); ++bx_)                                     \
            wavesim::run_block(bx_, (unsigned)(grid), (int)(block), [](void* p) { (*(Body_*)p)(); }, &body_); \
    } while (0)

#endif
