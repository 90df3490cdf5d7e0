// ldsbeside.hip -- what fits BESIDE the decode waves?  Round 6: clx_k_lanes (14 336 B of LDS, 108 VGPRs per one-wave workgroup) waits half a
// millisecond and more for room on CUs that hold ten decode waves of 15 360 B and 168 VGPRs (profiles/r05_config3_pipelined_trace.txt).  What may a
// workgroup ask for and still start at once there?  Kernel A: ten one-wave workgroups per CU of `sa` bytes of LDS (and 168 VGPRs in the
// second part) that hold their slot for 2 ms; 0.3 ms later kernel B on a second stream: one workgroup per CU of `sb` bytes (and 16 / 112 /
// 144 VGPRs) that returns at once.  B's completion time says whether it found room (~0.3 ms) or waited for A's waves to leave (~2 ms).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
#define BODY \
    extern __shared__ uint32_t lds[]; \
    lds[threadIdx.x] = blockIdx.x; \
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime(); \
    while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)spin) __builtin_amdgcn_s_sleep(8); \
    if (threadIdx.x == 0) out[blockIdx.x] = lds[0];
__global__ __launch_bounds__(64) void k_hold(uint32_t* out, int spin) { BODY }
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(168))) void k_hold168(uint32_t* out, int spin) { BODY }
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(112))) void k_probe112(uint32_t* out, int spin) { BODY }
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(144))) void k_probe144(uint32_t* out, int spin) { BODY }
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(24))) void k_probe24(uint32_t* out, int spin) { BODY }
typedef void (*kern)(uint32_t*, int);
int main() {
    uint32_t* d; if (hipMalloc(&d, 65536 * 4) != hipSuccess) return 1;
    hipStream_t s1, s2; (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (kern k : { (kern)k_hold, (kern)k_hold168, (kern)k_probe112, (kern)k_probe144, (kern)k_probe24 })
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    auto run = [&](kern ka, const char* na, int sa, int per_cu, kern kb, const char* nb, int sb, int nbw) {
        (void)hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        ka<<<256 * per_cu, 64, sa, s1>>>(d, 200000);                 // (more than fit simply queue), 2 ms each
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.0003) {}
        kb<<<nbw, 64, sb, s2>>>(d + 32768, 100);
        (void)hipStreamSynchronize(s2);
        const double tb = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3;
        (void)hipStreamSynchronize(s1);
        const double ta = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3;
        printf("A: %2d x %5d B %-9s per CU held 2 ms | B: %4d workgroups of %5d B %-10s came back after %.2f ms (A after %.2f ms)  -> %s\n", per_cu, sa, na, nbw, sb, nb, tb, ta,
               tb < 1.5 ? "found room" : "WAITED");
    };
    const int sb_list[] = { 0, 4096, 8192, 9216, 10240, 11264, 12288, 13312, 14336, 15360 };
    for (int sa : { 15360, 15872, 16384 }) for (int sb : sb_list) run(k_hold, "few VGPRs", sa, 10, k_hold, "few VGPRs", sb, 256);
    // with the kernels' registers: 168 per decode wave (three waves fill a SIMD's 512), 112 / 144 / 24 for the kernels behind
    for (int per_cu : { 10, 12 })
        for (int sb : { 0, 3584, 10240, 14336 }) {
            run(k_hold168, "168 VGPRs", 15360, per_cu, k_probe112, "112 VGPRs", sb, 256);
            run(k_hold168, "168 VGPRs", 15360, per_cu, k_probe144, "144 VGPRs", sb, 256);
            run(k_hold168, "168 VGPRs", 15360, per_cu, k_probe24, "24 VGPRs", sb, 256);
            run(k_hold168, "168 VGPRs", 15360, per_cu, k_probe112, "112 VGPRs", sb, 96);
        }
    return 0;
}
