// vgpr_bank.hip -- does the SIMD time of a three-source vector instruction depend on WHICH registers its sources are?
// (VGPR banks = register number mod 4 on GCN-lineage parts.)  Explicit registers through asm clobbers; 8 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define CLOB : : : "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59"
template <int KIND>
__global__ __launch_bounds__(64) void k_busy(uint64_t* out, int iters) {
    asm volatile("v_mov_b32 v44, 1\n v_mov_b32 v45, 2\n v_mov_b32 v46, 3\n v_mov_b32 v47, 5\n v_mov_b32 v48, 7\n v_mov_b32 v49, 9\n v_mov_b32 v50, 11\n v_mov_b32 v51, 13\n v_mov_b32 v52, 17\n v_mov_b32 v53, 19\n v_mov_b32 v54, 23\n v_mov_b32 v55, 29" CLOB);
    uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        // sources v44 v48 v52: all bank 0
        if (KIND == 0) { REP16(asm volatile("v_alignbit_b32 v40, v44, v48, v52\n v_alignbit_b32 v41, v44, v48, v52\n v_alignbit_b32 v42, v44, v48, v52\n v_alignbit_b32 v43, v44, v48, v52" CLOB);) }
        // sources v44 v49 v54: banks 0 1 2
        else if (KIND == 1) { REP16(asm volatile("v_alignbit_b32 v40, v44, v49, v54\n v_alignbit_b32 v41, v44, v49, v54\n v_alignbit_b32 v42, v44, v49, v54\n v_alignbit_b32 v43, v44, v49, v54" CLOB);) }
        // two sources same bank, third another
        else if (KIND == 2) { REP16(asm volatile("v_alignbit_b32 v40, v44, v48, v53\n v_alignbit_b32 v41, v44, v48, v53\n v_alignbit_b32 v42, v44, v48, v53\n v_alignbit_b32 v43, v44, v48, v53" CLOB);) }
        // VOP2 with two sources in one bank / in two
        else if (KIND == 3) { REP16(asm volatile("v_max_i32 v40, v44, v48\n v_max_i32 v41, v44, v48\n v_max_i32 v42, v44, v48\n v_max_i32 v43, v44, v48" CLOB);) }
        else if (KIND == 4) { REP16(asm volatile("v_max_i32 v40, v44, v49\n v_max_i32 v41, v44, v49\n v_max_i32 v42, v44, v49\n v_max_i32 v43, v44, v49" CLOB);) }
        // dot2c: acc (dst) + two sources
        else if (KIND == 5) { REP16(asm volatile("v_dot2c_i32_i16 v40, v44, v48\n v_dot2c_i32_i16 v41, v45, v49\n v_dot2c_i32_i16 v42, v46, v50\n v_dot2c_i32_i16 v43, v47, v51" CLOB);) }
        else if (KIND == 6) { REP16(asm volatile("v_dot2c_i32_i16 v40, v45, v50\n v_dot2c_i32_i16 v41, v46, v51\n v_dot2c_i32_i16 v42, v47, v48\n v_dot2c_i32_i16 v43, v44, v49" CLOB);) }
        // dependent chain of three-source instructions (each reads the one before)
        else if (KIND == 7) { REP16(asm volatile("v_alignbit_b32 v40, v40, v49, v54\n v_alignbit_b32 v40, v40, v49, v54\n v_alignbit_b32 v40, v40, v49, v54\n v_alignbit_b32 v40, v40, v49, v54" CLOB);) }
        // the same, two chains interleaved
        else if (KIND == 8) { REP16(asm volatile("v_alignbit_b32 v40, v40, v49, v54\n v_alignbit_b32 v41, v41, v50, v55\n v_alignbit_b32 v40, v40, v49, v54\n v_alignbit_b32 v41, v41, v50, v55" CLOB);) }
        // add/sub VOP2 (the cheap class), sources in one bank / two banks
        else if (KIND == 9) { REP16(asm volatile("v_add_u32 v40, v44, v48\n v_add_u32 v41, v44, v48\n v_add_u32 v42, v44, v48\n v_add_u32 v43, v44, v48" CLOB);) }
        else if (KIND == 10) { REP16(asm volatile("v_add_u32 v40, v44, v49\n v_add_u32 v41, v44, v49\n v_add_u32 v42, v44, v49\n v_add_u32 v43, v44, v49" CLOB);) }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}
template <int KIND> void run(uint64_t* d, const char* name, int waves_per_simd, int iters) {
    const int grid = 1024 * waves_per_simd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_busy<KIND><<<grid, 64>>>(d, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k_busy<KIND><<<grid, 64>>>(d, iters); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double mhz = (double)h[0] / (double)h[1] * 100.0;
    printf("%-44s %d wave(s)/SIMD: wave 0 saw %.2f cycles per instruction; SIMD: %.2f cycles per wave-instruction\n",
           name, waves_per_simd, (double)h[0] / (iters * 64.0), ms * 1e-3 * mhz * 1e6 / ((double)waves_per_simd * iters * 64.0));
}
#define BOTH(K, name) run<K>(d, name, 8, 2000); run<K>(d, name, 3, 2000); run<K>(d, name, 1, 2000);
int main() {
    uint64_t* d; if (hipMalloc(&d, 64) != hipSuccess) return 1;
    BOTH(0, "alignbit, 3 sources in ONE bank") BOTH(1, "alignbit, 3 sources in three banks") BOTH(2, "alignbit, two of three in one bank")
    BOTH(3, "v_max_i32, 2 sources one bank") BOTH(4, "v_max_i32, 2 sources two banks") BOTH(5, "dot2c, sources one bank") BOTH(6, "dot2c, sources spread")
    BOTH(7, "alignbit dependent chain") BOTH(8, "alignbit two dependent chains") BOTH(9, "v_add_u32 one bank") BOTH(10, "v_add_u32 two banks")
    return 0;
}
