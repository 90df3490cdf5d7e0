// dot2_hazard.hip -- what has to sit between a v_dot2_i32_i16 and a VALU instruction that reads its result on gfx950 (the
// predictor's 16-bit tier puts `s_nop 2` there, once per sample): nothing, s_nop 0 / 1 / 2, or independent vector instructions
// that do useful work.  Each variant runs a dependent chain and is compared with the host's arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define STR2(x) #x
#define STR(x) STR2(x)
template <int V>
__global__ void k(int32_t* o, int32_t p0, int32_t c1, int32_t c2, int iters) {
    int32_t p = p0 + (int32_t)threadIdx.x, acc, r = 0, m = 0, n = 0x7fffffff, q = p ^ 0x1234;
    for (int i = 0; i < iters; ++i) {
#define HEAD "v_dot2_i32_i16 %0, %4, %5, 0\n\tv_dot2_i32_i16 %0, %1, %6, %0\n\t"
#define TAIL "v_ashrrev_i32 %1, 1, %0\n\tv_add_u32 %1, %1, %4\n\tv_perm_b32 %4, %1, %4, %7"
        if (V == 0)      asm volatile(HEAD TAIL : "=&v"(acc), "+v"(r), "+v"(m), "+v"(n), "+v"(p) : "v"(c1), "v"(c2), "v"(0x05040100), "v"(q));
        else if (V == 1) asm volatile(HEAD "s_nop 0\n\t" TAIL : "=&v"(acc), "+v"(r), "+v"(m), "+v"(n), "+v"(p) : "v"(c1), "v"(c2), "v"(0x05040100), "v"(q));
        else if (V == 2) asm volatile(HEAD "s_nop 1\n\t" TAIL : "=&v"(acc), "+v"(r), "+v"(m), "+v"(n), "+v"(p) : "v"(c1), "v"(c2), "v"(0x05040100), "v"(q));
        else if (V == 3) asm volatile(HEAD "s_nop 2\n\t" TAIL : "=&v"(acc), "+v"(r), "+v"(m), "+v"(n), "+v"(p) : "v"(c1), "v"(c2), "v"(0x05040100), "v"(q));
        else if (V == 4) asm volatile(HEAD "v_max3_i32 %2, %2, %4, %8\n\t" TAIL : "=&v"(acc), "+v"(r), "+v"(m), "+v"(n), "+v"(p) : "v"(c1), "v"(c2), "v"(0x05040100), "v"(q));
        else if (V == 5) asm volatile(HEAD "v_max3_i32 %2, %2, %4, %8\n\tv_min3_i32 %3, %3, %4, %8\n\t" TAIL : "=&v"(acc), "+v"(r), "+v"(m), "+v"(n), "+v"(p) : "v"(c1), "v"(c2), "v"(0x05040100), "v"(q));
        else if (V == 6) asm volatile(HEAD "v_max3_i32 %2, %2, %4, %8\n\ts_nop 0\n\t" TAIL : "=&v"(acc), "+v"(r), "+v"(m), "+v"(n), "+v"(p) : "v"(c1), "v"(c2), "v"(0x05040100), "v"(q));
        else             asm volatile(HEAD "v_max3_i32 %2, %2, %4, %8\n\ts_nop 1\n\t" TAIL : "=&v"(acc), "+v"(r), "+v"(m), "+v"(n), "+v"(p) : "v"(c1), "v"(c2), "v"(0x05040100), "v"(q));
    }
    o[threadIdx.x] = p ^ r;
}
static int32_t dot2(int32_t a, int32_t b, int32_t c) { return (int32_t)(int16_t)(a & 0xffff) * (int16_t)(b & 0xffff) + (int32_t)(int16_t)(a >> 16) * (int16_t)(b >> 16) + c; }
static int32_t host(int32_t p0, int lane, int32_t c1, int32_t c2, int iters) {
    int32_t p = p0 + lane, r = 0;
    for (int i = 0; i < iters; ++i) {
        int32_t acc = dot2(p, c1, 0); acc = dot2(r, c2, acc);
        r = (acc >> 1) + p;
        // v_perm_b32 D, S0 = r, S1 = p, sel 0x05040100: bytes {S1.b0, S1.b1, S0.b0, S0.b1} -> (r.lo16 << 16) | p.lo16
        p = (int32_t)(((uint32_t)r << 16) | ((uint32_t)p & 0xffffu));
    }
    return p ^ r;
}
int main() {
    int32_t* d; (void)hipMalloc(&d, 256);
    const int32_t p0 = 0x00030005, c1 = (int32_t)0x0002fffd, c2 = 0x00010003; const int iters = 2000;
    int32_t h[64];
    const char* names[8] = { "nothing", "s_nop 0", "s_nop 1", "s_nop 2", "one independent v_max3", "v_max3 + v_min3", "v_max3 + s_nop 0", "v_max3 + s_nop 1" };
#define RUN(V) { k<V><<<1, 64>>>(d, p0, c1, c2, iters); (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); int bad = 0; \
                 for (int l = 0; l < 64; ++l) bad += h[l] != host(p0, l, c1, c2, iters); printf("%-26s: %s\n", names[V], bad ? "RESULT DIFFERS" : "ok"); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    return 0;
}
