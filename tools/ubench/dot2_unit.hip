// dot2_unit.hip -- what v_dot2_i32_i16 / v_perm_b32 compute on gfx950 (unit check for the packed-history predictor)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(int32_t* o, const int32_t* a) {
    int32_t x = a[0], y = a[1], z = a[2], r, p, r2;
    asm volatile("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    asm volatile("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r2) : "v"(x), "v"(y));
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(p) : "v"(a[3]), "v"(a[4]), "v"(0x05040100));
    o[0] = r; o[1] = r2; o[2] = p;
}
int main() {
    int32_t h[5] = { (int32_t)((uint32_t)(-3 & 0xffff) << 16 | (5 & 0xffff)), (int32_t)((uint32_t)(7 & 0xffff) << 16 | ((uint32_t)-11 & 0xffff)), 1000,
                     (int32_t)0xAABB1234, (int32_t)0xCCDD5678 };
    int32_t *d, *o, r[3];
    (void)hipMalloc(&d, 20); (void)hipMalloc(&o, 12);
    (void)hipMemcpy(d, h, 20, hipMemcpyHostToDevice);
    k<<<1, 1>>>(o, d); (void)hipMemcpy(r, o, 12, hipMemcpyDeviceToHost);
    printf("dot2((-3,5),(7,-11),1000) = %d  (expect lo*lo + hi*hi + c = 5*-11 + -3*7 + 1000 = %d)\n", r[0], 5 * -11 + -3 * 7 + 1000);
    printf("dot2(..., 0) = %d (expect %d)\n", r[1], 5 * -11 + -3 * 7);
    printf("perm(S0=0xAABB1234, S1=0xCCDD5678, 0x05040100) = 0x%08x (expect 0x12345678)\n", (unsigned)r[2]);
    return 0;
}
