// TEST INFRASTRUCTURE: the decode lanes' CRC-16 arithmetic (claxon_amd/csrc/clx_crct.h: the frame's polynomial modulo x^15 + x + 1
// and its parity) against the byte-wise CRC-16 of crc.rs:109-112, on random frames placed at random byte offsets inside 16-byte
// granules, split into up to four shares at random granule boundaries and recombined as clx_k_finalize does.  g++, host only.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../claxon_amd/csrc/clx_crct.h"

static uint16_t crc16_bytewise(const uint8_t* p, size_t n) {      // x^16 + x^15 + x^2 + 1, initial value 0, most significant bit first
    uint32_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        c ^= (uint32_t)p[i] << 8;
        for (int b = 0; b < 8; ++b) c = (c & 0x8000u) ? ((c << 1) ^ 0x8005u) & 0xffffu : (c << 1) & 0xffffu;
    }
    return (uint16_t)c;
}
static uint32_t rnd(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); }

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 3000;
    uint64_t seed = 12345;
    int intact = 0, damaged = 0;
    for (int t = 0; t < trials; ++t) {
        const size_t n = 1 + rnd(seed) % 700, lead = rnd(seed) % 16;
        std::vector<uint8_t> buf(lead + n + 2, 0);
        for (size_t i = 0; i < n; ++i) buf[lead + i] = (uint8_t)rnd(seed);
        const uint16_t c = crc16_bytewise(buf.data() + lead, n);
        buf[lead + n] = (uint8_t)(c >> 8); buf[lead + n + 1] = (uint8_t)c;
        const int kind = t % 3;
        if (kind == 1) buf[lead + n + (rnd(seed) & 1)] ^= (uint8_t)(1u << (rnd(seed) % 8));            // a damaged footer
        if (kind == 2) buf[lead + rnd(seed) % n] ^= (uint8_t)(1u << (rnd(seed) % 8));                 // a damaged byte
        const bool good = crc16_bytewise(buf.data() + lead, n) == (((uint32_t)buf[lead + n] << 8) | buf[lead + n + 1]);
        buf.resize((buf.size() + 15) / 16 * 16, 0);                  // (bytes outside the frame inside its first / last granule: zero)
        const uint32_t nd = (uint32_t)(buf.size() / 4), ng = nd / 4;
        uint32_t cuts[5] = { 0, rnd(seed) % (ng + 1), rnd(seed) % (ng + 1), rnd(seed) % (ng + 1), ng };
        for (int a = 1; a < 4; ++a) for (int b = a + 1; b < 4; ++b) if (cuts[b] < cuts[a]) { const uint32_t x = cuts[a]; cuts[a] = cuts[b]; cuts[b] = x; }
        uint32_t sum = 0, par = 0;
        for (int k = 0; k < 4; ++k) {
            clx_crct cs = { 0u, 0u };
            for (uint32_t d = 4 * cuts[k]; d < 4 * cuts[k + 1]; ++d)
                clx_crct_word(cs, ((uint32_t)buf[4 * d] << 24) | ((uint32_t)buf[4 * d + 1] << 16) | ((uint32_t)buf[4 * d + 2] << 8) | buf[4 * d + 3]);
            if (cs.r >= (1u << 18)) { printf("lazy remainder out of range\n"); return 1; }
            sum ^= clx_crct_shift(cs.r, nd - 4 * cuts[k + 1]);
            par ^= (uint32_t)__builtin_popcount(cs.x) & 1u;
        }
        const bool ok = sum == 0u && par == 0u;
        if (ok != good) { printf("trial %d: trinomial form says %d, byte-wise CRC says %d\n", t, (int)ok, (int)good); return 1; }
        intact += good; damaged += !good;
    }
    printf("ok %d intact %d damaged\n", intact, damaged);
    return (intact > trials / 4 && damaged > trials / 4) ? 0 : 1;
}
