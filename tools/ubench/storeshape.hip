// storeshape.hip -- how fast can one wave per 64 rows write 16 KiB rows, by store shape?  (write-only; K2's output side)
// G lanes cover G*16 contiguous bytes of one row; an instruction covers 64/G rows.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr uint32_t ROW = 4096;
template <int G, int NT>
__global__ __launch_bounds__(64) void k_store(int32_t* __restrict__ a, uint32_t nrows) {
    const uint32_t lane = threadIdx.x;
    const uint32_t r0 = blockIdx.x * 64u;
    constexpr uint32_t RPI = 64 / G;                 // rows per instruction
    const uint32_t sub = lane / G, col = lane % G;
    // walk: for every 'turn' (G*16 bytes of every row): RPI rows per instruction, 64/RPI = G instructions
    for (uint32_t v = 0; v < ROW / 4; v += G) {
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)G; ++k) {
            const uint32_t r = r0 + k * RPI + sub;
            if (r < nrows) {
                int4 w = make_int4(v, k, r, lane);
                int4* p = (int4*)(a + (size_t)r * ROW) + v + col;
                typedef int v4i __attribute__((ext_vector_type(4)));
                if (NT) __builtin_nontemporal_store(v4i{w.x, w.y, w.z, w.w}, (v4i*)p); else *p = w;
            }
        }
    }
}
template <int W>     // W-byte stores (4 or 8), one row per lane
__global__ __launch_bounds__(64) void k_narrow(int32_t* __restrict__ a, uint32_t nrows) {
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if (r >= nrows) return;
    int32_t* row = a + (size_t)r * ROW;
    for (uint32_t t = 0; t < ROW; t += 16) {
#pragma unroll
        for (uint32_t q = 0; q < 16; q += W / 4) {
            if (W == 4) row[t + q] = t + q; else *(int2*)(row + t + q) = make_int2(t, q);
        }
    }
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
    int32_t* a; if (hipMalloc(&a, bytes) != hipSuccess) return 1;
    (void)hipMemset(a, 1, bytes);
    const uint32_t grid = (nrows + 63) / 64;
    printf("rows %u (%.0f MB), %u waves\n", nrows, bytes / 1e6, grid);
#define RUN(name, ...) { float ms = timeit([&] { __VA_ARGS__; }, 20); printf("%-40s %.3f ms  %6.2f TB/s\n", name, ms, bytes / 1e9 / ms); }
    RUN("16 B x 64 rows / instr", (k_store<1, 0><<<grid, 64>>>(a, nrows)));
    RUN("32 B x 32 rows", (k_store<2, 0><<<grid, 64>>>(a, nrows)));
    RUN("64 B x 16 rows", (k_store<4, 0><<<grid, 64>>>(a, nrows)));
    RUN("128 B x 8 rows", (k_store<8, 0><<<grid, 64>>>(a, nrows)));
    RUN("256 B x 4 rows", (k_store<16, 0><<<grid, 64>>>(a, nrows)));
    RUN("512 B x 2 rows", (k_store<32, 0><<<grid, 64>>>(a, nrows)));
    RUN("1 KiB x 1 row", (k_store<64, 0><<<grid, 64>>>(a, nrows)));
    RUN("16 B x 64 rows nontemporal", (k_store<1, 1><<<grid, 64>>>(a, nrows)));
    RUN("64 B x 16 rows nontemporal", (k_store<4, 1><<<grid, 64>>>(a, nrows)));
    RUN("256 B x 4 rows nontemporal", (k_store<16, 1><<<grid, 64>>>(a, nrows)));
    RUN("1 KiB x 1 row nontemporal", (k_store<64, 1><<<grid, 64>>>(a, nrows)));
    RUN("4 B x 64 rows (dword stores)", (k_narrow<4><<<grid, 64>>>(a, nrows)));
    RUN("8 B x 64 rows (dwordx2 stores)", (k_narrow<8><<<grid, 64>>>(a, nrows)));
    return 0;
}
