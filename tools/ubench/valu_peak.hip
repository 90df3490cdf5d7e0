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
        else if (KIND == 1) { REP16(asm volatile("v_add_u32 %0, %4, %0\n\tv_xor_b32 %1, %4, %1\n\tv_add_u32 %2, %4, %2\n\tv_xor_b32 %3, %4, %3" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));) }
#define FOUR(op) REP16(asm volatile(op " %0, %0, %4, 5\n\t" op " %1, %1, %4, 5\n\t" op " %2, %2, %4, 5\n\t" op " %3, %3, %4, 5" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));)
        else if (KIND == 2) { FOUR("v_bfe_u32") }
        else if (KIND == 3) { FOUR("v_alignbit_b32") }
        else if (KIND == 4) { FOUR("v_and_or_b32") }
        else if (KIND == 5) { FOUR("v_lshl_or_b32") }
        else if (KIND == 6) { REP16(asm volatile("v_lshrrev_b32 %0, 3, %0\n\tv_and_b32 %1, %4, %1\n\tv_lshlrev_b32 %2, 1, %2\n\tv_or_b32 %3, %4, %3" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));) }
        else if (KIND == 7) { REP16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b) : "vcc");) }
        else if (KIND == 8) { REP16(asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));) }
        else if (KIND == 9) { REP16(asm volatile("v_add_u32 %0, %4, %0\n\ts_add_u32 %5, %5, 1\n\tv_xor_b32 %1, %4, %1\n\ts_add_u32 %5, %5, 1\n\tv_add_u32 %2, %4, %2\n\ts_add_u32 %5, %5, 1\n\tv_xor_b32 %3, %4, %3\n\ts_add_u32 %5, %5, 1" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(b), "+s"(a0) : : "scc");) }
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
    run<2>(d, "bfe", 8, 4000); run<3>(d, "alignbit", 8, 4000); run<4>(d, "and_or", 8, 4000); run<5>(d, "lshl_or", 8, 4000);
    run<6>(d, "shift/and/or (VOP2)", 8, 4000); run<7>(d, "cndmask", 8, 4000); run<8>(d, "mov_dpp wave_shr", 8, 4000);
    run<9>(d, "add/xor + one s_add each (per VALU instruction)", 8, 4000); run<9>(d, "add/xor + one s_add each (per VALU instruction)", 1, 4000);
    return 0;
}
