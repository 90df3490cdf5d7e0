// rowpipe.hip -- skeleton of K2's aligned path without the arithmetic: LDS-DMA ring in (64 B x 16 rows per instruction),
// LDS transposition, 64 B x 16 rows stores out, plus K dependent VALU instructions per 16-sample turn standing in for
// the predictor.  Answers: what does the memory structure cost alone, and does it hide under compute?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr uint32_t ROW = 4096;
__device__ __forceinline__ void glds16(const void* g, uint32_t lds) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int DEPTH, int K, bool INPLACE, bool STORE>
__global__ __launch_bounds__(64) void k_pipe(int32_t* __restrict__ a, int32_t* __restrict__ b, uint32_t nrows) {
    __shared__ int4 ring[DEPTH][4][64];
    const uint32_t lane = threadIdx.x;
    const uint32_t r0 = blockIdx.x * 64u;
    const uint32_t sw = (lane >> 2) & 3u, pc = (lane & 3u) ^ ((lane >> 4) & 3u);
    const int32_t* rp[4]; int32_t* wp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t r = r0 + k * 16 + (lane >> 2); if (r >= nrows) r = nrows - 1;
        rp[k] = a + (size_t)r * ROW + 4u * pc; wp[k] = (INPLACE ? a : b) + (size_t)r * ROW + 4u * pc;
    }
    auto dma = [&](uint32_t blk) __attribute__((always_inline)) {
        const uint32_t t = blk * 16u < ROW ? blk * 16u : ROW - 16u;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            glds16(rp[k] + t, __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&ring[blk % DEPTH][k][0]));
    };
    for (uint32_t i = 0; i < (uint32_t)DEPTH; ++i) dma(i);
    for (uint32_t i = 0; i < ROW / 16; ++i) {
        if (i + 1 < (uint32_t)DEPTH) wait_vm<4 * (DEPTH - 1)>(); else wait_vm<(STORE ? 8 : 4) * (DEPTH - 1)>();
        __builtin_amdgcn_wave_barrier();
        int4* tile = &ring[i % DEPTH][0][0];
        int4 x[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) x[q] = tile[lane * 4u + (q ^ sw)];
        int v = x[0].x ^ x[1].y ^ x[2].z ^ x[3].w;
#pragma unroll 8
        for (int j = 0; j < K; ++j) v = v * 3 + j;               // dependent chain, one VALU op (v_mad) each
        x[0].x += v;
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) tile[lane * 4u + (q ^ sw)] = x[q];
        __builtin_amdgcn_wave_barrier();
        if (STORE) {
            int4 w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = ring[i % DEPTH][k][lane];
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<int4*>(wp[k] + i * 16u) = w[k];
        }
        wait_lds();
        dma(i + DEPTH);
    }
    wait_vm<0>();
}
template <typename F> static float timeit(F f, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main(int argc, char** argv) {
    const uint32_t nrows = argc > 1 ? atoi(argv[1]) : 20000;
    const size_t bytes = (size_t)nrows * ROW * 4;
    int32_t *a, *b; if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 1, bytes);
    const uint32_t grid = (nrows + 63) / 64;
    printf("rows %u (%.0f MB), %u waves\n", nrows, bytes / 1e6, grid);
#define RUN(name, ...) { float ms = timeit([&] { __VA_ARGS__; }, 20); printf("%-44s %.3f ms\n", name, ms); }
    RUN("D8 K0   in place", (k_pipe<8, 0, true, true><<<grid, 64>>>(a, b, nrows)));
    RUN("D8 K0   a->b", (k_pipe<8, 0, false, true><<<grid, 64>>>(a, b, nrows)));
    RUN("D8 K0   load only", (k_pipe<8, 0, true, false><<<grid, 64>>>(a, b, nrows)));
    RUN("D4 K0   in place", (k_pipe<4, 0, true, true><<<grid, 64>>>(a, b, nrows)));
    RUN("D2 K0   in place", (k_pipe<2, 0, true, true><<<grid, 64>>>(a, b, nrows)));
    RUN("D8 K256 in place", (k_pipe<8, 256, true, true><<<grid, 64>>>(a, b, nrows)));
    RUN("D8 K512 in place", (k_pipe<8, 512, true, true><<<grid, 64>>>(a, b, nrows)));
    RUN("D8 K512 load only", (k_pipe<8, 512, true, false><<<grid, 64>>>(a, b, nrows)));
    RUN("D8 K512 a->b", (k_pipe<8, 512, false, true><<<grid, 64>>>(a, b, nrows)));
    RUN("D4 K512 in place", (k_pipe<4, 512, true, true><<<grid, 64>>>(a, b, nrows)));
    RUN("D2 K512 in place", (k_pipe<2, 512, true, true><<<grid, 64>>>(a, b, nrows)));
    return 0;
}
