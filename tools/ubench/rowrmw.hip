// rowrmw.hip -- microbenchmark of K2's memory pattern: every lane walks its own 16 KiB row, reading and rewriting it.
// Build: hipcc --offload-arch=gfx950 -O3 -o rowrmw rowrmw.hip ; run on the GPU box, prints ms and GB/s per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr uint32_t ROW = 4096;     // samples per row

// CH int4 per lane per turn (CH*16 bytes contiguous per row), DEPTH turns prefetched; MODE 0 rmw in place, 1 read only
// (xor-reduce), 2 write only, 3 read src / write dst (separate buffers)
template <int CH, int MODE>
__global__ __launch_bounds__(64) void k_rows(int32_t* __restrict__ a, int32_t* __restrict__ b, uint32_t nrows, int32_t* sink) {
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if (r >= nrows) return;
    int4* row = (int4*)(a + (size_t)r * ROW);
    int4* dst = MODE == 3 ? (int4*)(b + (size_t)r * ROW) : row;
    constexpr uint32_t NV = ROW / 4;       // int4 per row
    int4 cur[CH], nxt[CH];
    int acc = 0;
    if (MODE != 2) {
#pragma unroll
        for (int q = 0; q < CH; ++q) cur[q] = row[q];
    }
    for (uint32_t v = 0; v < NV; v += CH) {
        if (MODE != 2) {
            const uint32_t vn = v + CH < NV ? v + CH : v;
#pragma unroll
            for (int q = 0; q < CH; ++q) nxt[q] = row[vn + q];
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            int4 w = MODE == 2 ? make_int4(v, q, r, 1) : cur[q];
            w.x += 1; w.y ^= w.x; w.z += w.y; w.w ^= w.z;
            if (MODE == 1) acc ^= w.w; else dst[v + q] = w;
        }
        if (MODE != 2) {
#pragma unroll
            for (int q = 0; q < CH; ++q) cur[q] = nxt[q];
        }
    }
    if (MODE == 1 && acc == 0x12345678) *sink = acc;
}

// coalesced reference: a wave streams whole rows, 1 KiB per instruction
template <int MODE>
__global__ __launch_bounds__(64) void k_stream(int32_t* __restrict__ a, uint32_t nrows, int32_t* sink) {
    int acc = 0;
    for (uint32_t r = blockIdx.x * 64u; r < blockIdx.x * 64u + 64u && r < nrows; ++r) {
        int4* row = (int4*)(a + (size_t)r * ROW);
#pragma unroll 4
        for (uint32_t v = threadIdx.x; v < ROW / 4; v += 64) {
            int4 w = row[v];
            w.x += 1; w.y ^= w.x; w.z += w.y; w.w ^= w.z;
            if (MODE == 1) acc ^= w.w; else row[v] = w;
        }
    }
    if (MODE == 1 && acc == 0x12345678) *sink = acc;
}

template <typename F> static float timeit(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char** argv) {
    const uint32_t nrows = argc > 1 ? atoi(argv[1]) : 20000;
    const size_t bytes = (size_t)nrows * ROW * 4;
    int32_t *a, *b, *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    const uint32_t grid = (nrows + 63) / 64;
    const double mb = bytes / 1e6;
    printf("rows %u x 16 KiB = %.0f MB, %u waves\n", nrows, mb, grid);
#define RUN(name, traffic, ...) { float ms = timeit([&] { __VA_ARGS__; }, 20); printf("%-34s %.3f ms  %7.0f GB/s\n", name, ms, (traffic) * mb / ms / 1e3 * 1.0); }
    RUN("stream rmw (coalesced)", 2, (k_stream<0><<<grid, 64>>>(a, nrows, sink)));
    RUN("stream read", 1, (k_stream<1><<<grid, 64>>>(a, nrows, sink)));
    RUN("rows rmw   64 B/turn", 2, (k_rows<4, 0><<<grid, 64>>>(a, b, nrows, sink)));
    RUN("rows rmw  128 B/turn", 2, (k_rows<8, 0><<<grid, 64>>>(a, b, nrows, sink)));
    RUN("rows rmw  256 B/turn", 2, (k_rows<16, 0><<<grid, 64>>>(a, b, nrows, sink)));
    RUN("rows rmw  512 B/turn", 2, (k_rows<32, 0><<<grid, 64>>>(a, b, nrows, sink)));
    RUN("rows read  64 B/turn", 1, (k_rows<4, 1><<<grid, 64>>>(a, b, nrows, sink)));
    RUN("rows read 256 B/turn", 1, (k_rows<16, 1><<<grid, 64>>>(a, b, nrows, sink)));
    RUN("rows write  64 B/turn", 1, (k_rows<4, 2><<<grid, 64>>>(a, b, nrows, sink)));
    RUN("rows write 256 B/turn", 1, (k_rows<16, 2><<<grid, 64>>>(a, b, nrows, sink)));
    RUN("rows a->b   64 B/turn", 2, (k_rows<4, 3><<<grid, 64>>>(a, b, nrows, sink)));
    RUN("rows a->b  256 B/turn", 2, (k_rows<16, 3><<<grid, 64>>>(a, b, nrows, sink)));
    return 0;
}
