// clx_intrin.h -- gfx950 instruction wrappers used by the kernels (resolved via -I .../csrc/intrin).
#ifndef CLX_INTRIN_H
#define CLX_INTRIN_H
#include <stdint.h>

// v_alignbit_b32: low 32 bits of ({hi,lo} >> (shift & 31))
__device__ __forceinline__ uint32_t clx_alignbit(uint32_t hi, uint32_t lo, uint32_t shift) {
    return __builtin_amdgcn_alignbit(hi, lo, shift);
}
// v_bfe_u32: (src >> offset) & ((1 << width) - 1); width 0 -> 0
__device__ __forceinline__ uint32_t clx_bfe(uint32_t src, uint32_t offset, uint32_t width) {
    return __builtin_amdgcn_ubfe(src, offset, width);
}
// v_mad_i32_i24: low 32 bits of sext24(a)*sext24(b) + c.  Inline asm so that the accumulation stays a chain in
// the order written (oldest tap first, newest last): only the last mad then depends on the newest sample.
__device__ __forceinline__ int32_t clx_mad24(int32_t a, int32_t b, int32_t c) {
    int32_t d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int32_t clx_max3(int32_t a, int32_t b, int32_t c) {
    int32_t d;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int32_t clx_min3(int32_t a, int32_t b, int32_t c) {
    int32_t d;
    asm("v_min3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
#endif
