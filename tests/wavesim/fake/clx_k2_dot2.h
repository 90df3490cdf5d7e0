// TEST INFRASTRUCTURE: host definition of clx_dot2_block (claxon_amd/csrc/intrin/clx_k2_dot2.h) with the documented semantics of
// v_dot2_i32_i16 / v_perm_b32 (history pairs truncated to 16 bits, products of sign-extended halves, wrapping 32-bit sums), for
// the wave simulator build.
#ifndef CLX_K2_DOT2_H
#define CLX_K2_DOT2_H
#include <stdint.h>
template <int OMAX> static inline void clx_dot2_block(const int32_t (&x)[16], int32_t (&y)[16], int32_t (&pr)[OMAX - 1], const int32_t (&C)[OMAX / 2],
                                                      uint32_t shift, int32_t prev) {
    // pair_k for k = -(OMAX-1) .. 15, stored at index k + OMAX - 1
    uint32_t pair[OMAX - 1 + 16];
    for (int j = 0; j < OMAX - 1; ++j) pair[OMAX - 2 - j] = (uint32_t)pr[j];            // pr[j] = pair_{-1-j}
    int32_t sprev = prev;
    for (int i = 0; i < 16; ++i) {
        uint32_t acc = 0;
        for (int p = OMAX / 2 - 1; p >= 0; --p) {
            const uint32_t h = pair[(i - 1 - 2 * p) + OMAX - 1], c = (uint32_t)C[p];
            acc += (uint32_t)((int32_t)(int16_t)(c & 0xffffu) * (int32_t)(int16_t)(h & 0xffffu))
                 + (uint32_t)((int32_t)(int16_t)(c >> 16) * (int32_t)(int16_t)(h >> 16));
        }
        const int32_t s = (int32_t)((uint32_t)((int32_t)acc >> (shift & 31u)) + (uint32_t)x[i]);
        y[i] = s;
        pair[i + OMAX - 1] = ((uint32_t)s << 16) | ((uint32_t)sprev & 0xffffu);
        sprev = s;
    }
    for (int j = 0; j < OMAX - 1; ++j) pr[j] = 0;                                           // (clobbered on the device)
}
#endif
