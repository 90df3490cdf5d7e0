// clx_crct.h -- the frame CRC-16 (frame.rs:752-763, crc.rs:109-112) as the decode lanes compute it: without tables.
//
// FLAC's CRC-16 is the remainder of the message polynomial times x^16 by P = x^16 + x^15 + x^2 + 1, first bit = highest degree,
// initial value 0, nothing reflected or inverted.  A frame's footer IS that remainder, so a frame is intact exactly when the
// polynomial of ALL its bytes, footer included, is a multiple of P.  P factors: P = (x + 1) * T with T = x^15 + x + 1 (a primitive
// trinomial), and the two factors are coprime, so
//     frame intact  <=>  frame(x) = 0 mod T   and   frame(1) = 0  (an even number of one bits).
// Both halves are cheap per 32-bit word W (big-endian, as the decode lanes hold their stream in the LDS ring):
//   * mod T by Horner's rule, r' = r * x^32 + W, with x^15 = x + 1: x^32 = x^4 + x^2, so r * x^32 is two shifts and an XOR, and
//     the 32-bit sum folds back to 18 bits with hi * (x + 1): shift, AND, shift, two XORs.  r is kept lazily reduced (< 2^18);
//   * the parity of the frame is the parity of the XOR of its words: one XOR per word.
// Eight cheap vector instructions per word, no look-up, no LDS -- against four dependent table look-ups per word in the
// stand-alone kernel (clx_k_crc16), whose tables would cost the decode waves LDS they do not have.
// Zero words in front of a message do not change its remainder, zero words behind it multiply it by a power of x (which keeps a
// zero remainder zero): bytes outside the frame inside its first and last 16-byte granule are masked to zero.
// A frame's subframes are decoded by different lanes: lane c holds the remainder r_c of the words [Da_c, Db_c) of the frame;
// clx_k_finalize sums r_c * x^(32 * (Dend - Db_c)) (clx_crct_shift) and looks at the sum and the parity.
#ifndef CLX_CRCT_H
#define CLX_CRCT_H
#include <stdint.h>

#ifndef CLX_HD
#if defined(__HIPCC__)
#define CLX_HD __host__ __device__ __forceinline__
#else
#define CLX_HD static inline
#endif
#endif

struct clx_crct { uint32_t r, x; };       // r: the words so far mod T, lazily reduced (< 2^18); x: their XOR

// one more word (most significant bit first)
CLX_HD void clx_crct_word(clx_crct& c, uint32_t w) {
    const uint32_t b = (c.r << 4) ^ (c.r << 2) ^ w;
    const uint32_t h = b >> 15;
    c.r = (b & 0x7fffu) ^ h ^ (h << 1);
    c.x ^= w;
}
// the remainder fully reduced (15 bits)
CLX_HD uint32_t clx_crct_reduced(uint32_t r) {
    const uint32_t h = r >> 15;
    return (r & 0x7fffu) ^ h ^ (h << 1);       // (h < 8: the result is below 2^15)
}
// a * b mod T (15-bit polynomials)
CLX_HD uint32_t clx_crct_mulmod(uint32_t a, uint32_t b) {
    uint32_t acc = 0;
    for (int i = 0; i < 15; ++i) acc ^= ((b >> i) & 1u) ? (a << i) : 0u;        // < 2^29
    uint32_t h = acc >> 15;
    acc = (acc & 0x7fffu) ^ h ^ (h << 1);                                       // < 2^16
    h = acc >> 15;
    return (acc & 0x7fffu) ^ h ^ (h << 1);
}
// r * x^(32 n) mod T: the remainder of a run of words that n more words follow (square and multiply on x^32 = x^4 + x^2)
CLX_HD uint32_t clx_crct_shift(uint32_t r, uint32_t n) {
    uint32_t base = 0x14u;
    r = clx_crct_reduced(r);
    while (n) {
        if (n & 1u) r = clx_crct_mulmod(r, base);
        base = clx_crct_mulmod(base, base);
        n >>= 1;
    }
    return r;
}
// What a lane leaves for clx_k_finalize (16 bytes per predictor slot, clx_run::crc_part): valid for the run whose generation
// number it carries.
struct clx_crc_part {
    uint32_t gen;            // the run's generation number (clx_run::gen)
    uint32_t rx;             // bits 0-14: the words [da, db) mod T, reduced; bit 31: their parity
    uint32_t da, db;         // word indices from the frame's 16-byte aligned origin (multiples of 4)
};
#endif
