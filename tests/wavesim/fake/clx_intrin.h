// TEST INFRASTRUCTURE: host definitions of the instruction wrappers in claxon_amd/csrc/intrin/clx_intrin.h,
// with the gfx950 instructions' documented semantics, for the wave simulator build.
#ifndef CLX_INTRIN_H
#define CLX_INTRIN_H
#include <stdint.h>
static inline uint32_t clx_alignbit(uint32_t hi, uint32_t lo, uint32_t shift) {
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (shift & 31u));
}
static inline uint32_t clx_bfe(uint32_t src, uint32_t offset, uint32_t width) {
    offset &= 31u; width &= 31u;
    if (width == 0u) return 0u;
    return (src >> offset) & (width >= 32u ? 0xffffffffu : ((1u << width) - 1u));
}
static inline uint32_t clx_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) { const uint32_t b = (sel >> (8 * i)) & 0xffu; r |= (b < 8u ? (uint32_t)((v >> (8 * b)) & 0xffu) : (b >= 13u ? 0xffu : 0u)) << (8 * i); }
    return r;
}
static inline int32_t clx_mad24(int32_t a, int32_t b, int32_t c) {
    return (int32_t)((uint32_t)((a << 8) >> 8) * (uint32_t)((b << 8) >> 8) + (uint32_t)c);
}
static inline int32_t clx_max3(int32_t a, int32_t b, int32_t c) { int32_t m = a > b ? a : b; return m > c ? m : c; }
static inline int32_t clx_min3(int32_t a, int32_t b, int32_t c) { int32_t m = a < b ? a : b; return m < c ? m : c; }
template <int N> static inline int32_t clx_dot24(const int32_t* c, const int32_t* h, int32_t acc) {
    for (int j = N - 1; j >= 0; --j) acc = clx_mad24(c[j], h[j], acc);
    return acc;
}
template <int N> static inline int32_t clx_dot24z(const int32_t* c, const int32_t* h) { return clx_dot24<N>(c, h, 0); }
static inline int32_t clx_ms_pair_(int line, int32_t y, uint32_t sgn, uint32_t nsg, uint32_t one) {
    const uint32_t side = (uint32_t)wavesim::update_dpp_(line, 0, y, 0xF5, 0xF, 0xF, false);
    const uint32_t mid = (uint32_t)wavesim::update_dpp_(line, 0, y, 0xA0, 0xF, 0xF, false);
    const uint32_t m = (mid << 1) | (side & one);
    return (int32_t)(m + (side ^ sgn) + nsg) >> 1;
}
#define clx_ms_pair(y, sgn, nsg, one) clx_ms_pair_(__LINE__, (y), (sgn), (nsg), (one))
static inline int32_t clx_sdot2(uint32_t a, uint32_t b, int32_t acc);
static inline int32_t clx_sdot2_first(uint32_t a, uint32_t b) { return clx_sdot2(a, b, 0); }
static inline int32_t clx_sdot2(uint32_t a, uint32_t b, int32_t acc) {
    const int32_t lo = (int32_t)(int16_t)(a & 0xffffu) * (int32_t)(int16_t)(b & 0xffffu);
    const int32_t hi = (int32_t)(int16_t)(a >> 16) * (int32_t)(int16_t)(b >> 16);
    return (int32_t)((uint32_t)lo + (uint32_t)hi + (uint32_t)acc);
}
struct clx_buf { const uint8_t* base; uint32_t bytes; };
static inline clx_buf clx_make_buf(const void* base, uint32_t bytes) { clx_buf b; b.base = (const uint8_t*)base; b.bytes = bytes; return b; }
static inline uint4 clx_buf_load16(const clx_buf& b, uint32_t byte_off) {     // dwords past the end of the buffer read as zero
    uint32_t w[4] = { 0u, 0u, 0u, 0u };
    for (int i = 0; i < 4; ++i) if ((uint64_t)byte_off + 4u * (uint32_t)i + 4u <= (uint64_t)b.bytes) memcpy(&w[i], b.base + byte_off + 4u * (uint32_t)i, 4);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
#define clx_ms_pair4(y, out, sgn, nsg, one) do { for (int q_ = 0; q_ < 4; ++q_) (out)[q_] = clx_ms_pair_(__LINE__, (y)[q_], (sgn), (nsg), (one)); } while (0)
#define clx_any(p) (wavesim::any_(__LINE__, (p) ? 1 : 0) != 0)
#define CLX_OPAQUE(x) ((void)(x))
#define CLX_OPAQUE_PTR(p) ((void)(p))
#define CLX_KERNARGS(T) ((const T*)sim_kernargs)      // (sim_lib.cpp points it at the argument block of the kernel it launches)
#define CLX_SCHED_BARRIER() ((void)0)
static inline void clx_store4x16(int32_t* p0, int32_t* p1, int32_t* p2, int32_t* p3, const int4& w0, const int4& w1, const int4& w2, const int4& w3) {
    *reinterpret_cast<int4*>(p0) = w0; *reinterpret_cast<int4*>(p1) = w1; *reinterpret_cast<int4*>(p2) = w2; *reinterpret_cast<int4*>(p3) = w3;
}
static inline void clx_store1x16_s(uint64_t base, uint32_t o, const int4& w) { *reinterpret_cast<int4*>((uintptr_t)(base + o)) = w; }
template <int O0, int O1, int O2, int O3>
static inline void clx_bperm4(uint32_t byte_addr, uint32_t v, uint32_t (&r)[4]) {
    r[0] = (uint32_t)__shfl(v, (int)(((byte_addr + O0) >> 2) & 63u), 64); r[1] = (uint32_t)__shfl(v, (int)(((byte_addr + O1) >> 2) & 63u), 64);
    r[2] = (uint32_t)__shfl(v, (int)(((byte_addr + O2) >> 2) & 63u), 64); r[3] = (uint32_t)__shfl(v, (int)(((byte_addr + O3) >> 2) & 63u), 64);
}
template <int O0, int O1>
static inline void clx_bperm2(uint32_t byte_addr, uint32_t v, uint32_t (&r)[2]) {
    r[0] = (uint32_t)__shfl(v, (int)(((byte_addr + O0) >> 2) & 63u), 64); r[1] = (uint32_t)__shfl(v, (int)(((byte_addr + O1) >> 2) & 63u), 64);
}
static inline void clx_store4x16_s(uint64_t base, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3, const int4& w0, const int4& w1, const int4& w2, const int4& w3) {
    clx_store1x16_s(base, o0, w0); clx_store1x16_s(base, o1, w1); clx_store1x16_s(base, o2, w2); clx_store1x16_s(base, o3, w3);
}
static inline int32_t clx_ms_short_(int line, int32_t y, uint32_t sgn, uint32_t c) {
    const uint32_t side = (uint32_t)wavesim::update_dpp_(line, 0, y, 0xF5, 0xF, 0xF, false);
    const uint32_t mid = (uint32_t)wavesim::update_dpp_(line, 0, y, 0xA0, 0xF, 0xF, false);
    return (int32_t)(mid + (uint32_t)((int32_t)((side ^ sgn) + c) >> 1));
}
#define clx_ms_short4(y, out, sgn, c) do { for (int q_ = 0; q_ < 4; ++q_) (out)[q_] = clx_ms_short_(__LINE__, (y)[q_], (sgn), (c)); } while (0)
static inline uint32_t clx_ffbh(uint32_t x) { return x ? (uint32_t)__builtin_clz(x) : 0xffffffffu; }
static inline int32_t clx_mad24_(int32_t a, int32_t b, int32_t c) {      // v_mad_i32_i24: the low 24 bits of a and b, sign-extended
    const int64_t x = (int64_t)((int32_t)((uint32_t)a << 8) >> 8), y = (int64_t)((int32_t)((uint32_t)b << 8) >> 8);
    return (int32_t)(uint32_t)((uint64_t)(x * y) + (uint64_t)(int64_t)c);
}
static inline int32_t clx_decor_mad_(int line, int32_t y, int32_t mo, int32_t mt, int32_t c) {
    const int32_t other = wavesim::update_dpp_(line, 0, y, 0xB1, 0xF, 0xF, false);                // quad_perm [1,0,3,2]
    return clx_mad24_(other, mt, clx_mad24_(y, mo, c)) >> 1;
}
#define clx_decor8_mad(y, out, mo, mt, c) do { for (int q_ = 0; q_ < 8; ++q_) (out)[q_] = clx_decor_mad_(__LINE__, (y)[q_], (mo), (mt), (c)); } while (0)
#define clx_decor4_mad(y, out, mo, mt, c) do { for (int q_ = 0; q_ < 4; ++q_) (out)[q_] = clx_decor_mad_(__LINE__, (y)[q_], (mo), (mt), (c)); } while (0)
static inline int32_t clx_decor_(int line, int32_t y, uint32_t sg, uint32_t rmask, uint32_t c, uint32_t s1, uint32_t pmask) {
    const uint32_t odd = (uint32_t)wavesim::update_dpp_(line, 0, y, 0xF5, 0xF, 0xF, false);       // quad_perm [1,1,3,3]
    const uint32_t even = (uint32_t)wavesim::update_dpp_(line, 0, y, 0xA0, 0xF, 0xF, false);      // quad_perm [0,0,2,2]
    return (int32_t)((even & pmask) + (uint32_t)((int32_t)(((odd ^ sg) & rmask) + c) >> s1));
}
#define clx_decor4(y, out, sg, rmask, c, s1, pmask) do { for (int q_ = 0; q_ < 4; ++q_) (out)[q_] = clx_decor_(__LINE__, (y)[q_], (sg), (rmask), (c), (s1), (pmask)); } while (0)
// LDS-DMA in the simulator: synchronous copy; the "LDS address" is simply the host pointer of the shared object.
static inline uintptr_t clx_lds_addr(const void* p) { return (uintptr_t)p; }
static inline void clx_glds16(const void* gsrc, uintptr_t lds_base) {
    memcpy((char*)lds_base + 16 * (wavesim::S().cur & 63), gsrc, 16);
}
template <int N> static inline void clx_wait_vmcnt() {}
static inline void clx_wait_lds() {}
#define clx_wave_sync() ((void)__ballot(1))     // a per-wave rendezvous (a block barrier would pair up with the other wave)
#define clx_wg_barrier() __syncthreads()
// clx_uniform: the kernels claim the value is the same in every live lane -- checked here
static inline uint32_t clx_uniform_(int line, uint32_t v) {
    const unsigned long long live = wavesim::ballot_(line, 1);
    const uint32_t first = wavesim::shfl_(line, v, __builtin_ctzll(live), 64);
    if (wavesim::any_(line, v != first)) { fprintf(stderr, "wavesim: clx_uniform() of a value that differs between lanes (line %d)\n", line); abort(); }
    return first;
}
#define clx_uniform(v) clx_uniform_(__LINE__, (v))
static inline uint32_t clx_peek_u32(const uint32_t* p) { return *(const volatile uint32_t*)p; }
static inline void clx_poke_u32(uint32_t* p, uint32_t v) { *(volatile uint32_t*)p = v; }
static inline void clx_stores_done() {}
#define clx_group_fence() ((void)__ballot(1))       // (a per-wave rendezvous: every lane has written its share)
static inline void clx_pause() {}
static inline void clx_release() {}
static inline void clx_acquire() {}
#define clx_readlane(v, idx) ((uint32_t)__shfl((uint32_t)(v), (int)(idx), 64))
#endif
