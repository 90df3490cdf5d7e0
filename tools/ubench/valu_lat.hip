// valu_lat.hip -- issue cost of dependent vs independent VALU chains for ONE wave per SIMD on gfx950 (s_memtime ticks =
// shader cycles).  Decides whether K2's per-row recurrence is bound by issue slots or by dependent-instruction latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define KERNEL(name, body, ninstr) \
__global__ __launch_bounds__(64) void name(uint64_t* out, int a0, int b0) { \
    int a = a0 + threadIdx.x, b = b0, c = a0 * 3, d = b0 * 5, e = a0 ^ 77, f = b0 + 9; long long w = a0; \
    uint64_t t0 = __builtin_amdgcn_s_memtime(); \
    for (int it = 0; it < 64; ++it) { REP64(body) } \
    asm volatile("s_nop 0" ::: "memory"); \
    uint64_t t1 = __builtin_amdgcn_s_memtime(); \
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = (uint64_t)(a + c + d + e + f + (int)w); } \
} \
static const int name##_n = ninstr;
KERNEL(k_mad24_dep,  asm volatile("v_mad_i32_i24 %0, %1, %0, %0" : "+v"(a) : "v"(b));, 1)
KERNEL(k_mad24_dep2, asm volatile("v_mad_i32_i24 %0, %2, %0, %0\n\tv_mad_i32_i24 %1, %2, %1, %1" : "+v"(a), "+v"(c) : "v"(b));, 2)
KERNEL(k_mad24_dep4, asm volatile("v_mad_i32_i24 %0, %4, %0, %0\n\tv_mad_i32_i24 %1, %4, %1, %1\n\tv_mad_i32_i24 %2, %4, %2, %2\n\tv_mad_i32_i24 %3, %4, %3, %3" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));, 4)
KERNEL(k_add_dep,    asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "v"(b));, 1)
KERNEL(k_add_dep2,   asm volatile("v_add_u32 %0, %2, %0\n\tv_add_u32 %1, %2, %1" : "+v"(a), "+v"(c) : "v"(b));, 2)
KERNEL(k_madu24_dep, asm volatile("v_mad_u32_u24 %0, %1, %0, %0" : "+v"(a) : "v"(b));, 1)
KERNEL(k_mad64_dep,  asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w) : "v"(b), "v"(c) : "vcc");, 1)
KERNEL(k_mullo_dep,  asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a) : "v"(b));, 1)
KERNEL(k_dpp_dep,    asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32 %0, %1, %0" : "+v"(a) : "v"(b));, 2)
KERNEL(k_ashr_add,   asm volatile("v_ashrrev_i32 %0, 3, %0\n\tv_add_u32 %0, %1, %0" : "+v"(a) : "v"(b));, 2)
KERNEL(k_max3,       asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));, 1)
int main() {
    uint64_t* d; if (hipMalloc(&d, 4096 * 16) != hipSuccess) return 1;
    uint64_t h[2];
#define RUN(name, grid) { name<<<grid, 64>>>(d, 3, 5); name<<<grid, 64>>>(d, 3, 5); (void)hipDeviceSynchronize(); \
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); \
        printf("%-14s grid %5d: %7.2f cycles per instruction (%d per body)\n", #name, grid, (double)h[0] / (64.0 * 64 * name##_n), name##_n); }
    RUN(k_mad24_dep, 1) RUN(k_mad24_dep2, 1) RUN(k_mad24_dep4, 1) RUN(k_add_dep, 1) RUN(k_add_dep2, 1) RUN(k_madu24_dep, 1)
    RUN(k_mad64_dep, 1) RUN(k_mullo_dep, 1) RUN(k_dpp_dep, 1) RUN(k_ashr_add, 1) RUN(k_max3, 1)
    RUN(k_mad24_dep, 313) RUN(k_mad24_dep2, 313) RUN(k_mad24_dep, 2048) RUN(k_mad24_dep2, 2048)
    RUN(k_mad24_dep4, 1024) RUN(k_mad24_dep4, 2048) RUN(k_mad24_dep4, 4096) RUN(k_mad24_dep4, 8192) RUN(k_mad24_dep, 8192) RUN(k_add_dep2, 8192)
    return 0;
}
