// ldsgran.hip -- how many one-wave workgroups of a given LDS size does a CU of this GPU hold?  (Round 5: the per-wave timeline showed 9
// decode waves per CU where 160 KiB / 15.5 KiB says 10 -- LDS is handed out in granules.)  Prints the occupancy the runtime
// reports for dynamic LDS sizes around the kernels' footprints, and measures it: a kernel whose waves record the CU they ran on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
__global__ __launch_bounds__(64) void k_hold(uint32_t* where, int spin) {
    extern __shared__ uint32_t lds[];
    lds[threadIdx.x] = blockIdx.x;
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)spin) __builtin_amdgcn_s_sleep(8);       // hold the slot ~spin x 10 ns
    if (threadIdx.x == 0) where[blockIdx.x] = ((xcc & 0xf) << 16) | ((hwid >> 8) & 0xffff);         // XCC | SE, SH, CU
}
int main() {
    const int sizes[] = { 12288, 12800, 13312, 13653, 14080, 14336, 14592, 14848, 15360, 15616, 15872, 16384, 16640, 17920, 18432 };
    uint32_t* d; const int grid = 256 * 16;
    if (hipMalloc(&d, grid * 4) != hipSuccess) return 1;
    for (int sz : sizes) {
        (void)hipFuncSetAttribute((const void*)k_hold, hipFuncAttributeMaxDynamicSharedMemorySize, sz);
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_hold, 64, (size_t)sz);
        // measured: launch 16 workgroups per CU that each hold their slot for 200 us; the kernel's duration / 200 us = rounds
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k_hold<<<grid, 64, sz>>>(d, 20000); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); k_hold<<<grid, 64, sz>>>(d, 20000); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("LDS %5d B per one-wave workgroup: runtime says %2d per CU; 16 per CU holding 0.2 ms each took %.2f ms = %.1f rounds -> %.1f resident per CU\n",
               sz, nb, ms, ms / 0.2, 16.0 / (ms / 0.2));
    }
    return 0;
}
