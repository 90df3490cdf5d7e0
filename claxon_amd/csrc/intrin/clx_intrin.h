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
#endif
