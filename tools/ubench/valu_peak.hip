// valu_peak.hip -- how many VALU wave-instructions a SIMD of gfx950 issues per cycle when it has 1, 2, 4, 8 waves to pick from
// (independent chains of v_mad_i32_i24 / v_add_u32, nothing else in the loop).  The denominator of "K1 runs at x % of the VALU
// issue rate" (DESIGN.md section 4.4 / 5).  Time from HIP events; shader clock from s_memtime against s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
template <int KIND>
__global__ __launch_bounds__(64) void k_busy(uint64_t* out, int a0, int b0, int iters) {
    int a = a0 + threadIdx.x, b = b0, c = a0 * 3, d = b0 * 5, e = a0 ^ 77;
    uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) { REP16(asm volatile("v_mad_i32_i24 %0, %4, %0, %0\n\tv_mad_i32_i24 %1, %4, %1, %1\n\tv_mad_i32_i24 %2, %4, %2, %2\n\tv_mad_i32_i24 %3, %4, %3, %3" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));) }
        else           { REP16(asm volatile("v_add_u32 %0, %4, %0\n\tv_xor_b32 %1, %4, %1\n\tv_add_u32 %2, %4, %2\n\tv_xor_b32 %3, %4, %3" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));) }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (a + c + d + e == 0x7fffffff) out[2] = 1;
}
template <int KIND> void run(uint64_t* d, const char* name, int waves_per_simd, int iters) {
    const int grid = 1024 * waves_per_simd;                    // 256 CUs x 4 SIMDs
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_busy<KIND><<<grid, 64>>>(d, 3, 5, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k_busy<KIND><<<grid, 64>>>(d, 3, 5, iters); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double mhz = (double)h[0] / (double)h[1] * 100.0;
    const double instr_per_simd = (double)waves_per_simd * iters * 64.0;
    const double cycles = ms * 1e-3 * mhz * 1e6;
    printf("%-8s %d wave(s) per SIMD: %.3f ms, shader %.0f MHz, wave 0 alone saw %.2f cycles per instruction; SIMD: %.2f cycles per wave-instruction\n",
           name, waves_per_simd, ms, mhz, (double)h[0] / (iters * 64.0), cycles / instr_per_simd);
}
int main() {
    uint64_t* d; if (hipMalloc(&d, 64) != hipSuccess) return 1;
    for (int w : {1, 2, 4, 8}) run<0>(d, "mad24", w, 4000);
    for (int w : {1, 2, 4, 8}) run<1>(d, "add/xor", w, 4000);
    return 0;
}
