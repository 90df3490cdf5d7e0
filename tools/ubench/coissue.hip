// coissue.hip -- does a SIMD of gfx950 take a scalar / LDS / branch instruction of one wave WHILE its vector pipe executes another
// wave's instruction, or do they queue behind each other?  (Round 5: bench.py's issue model added flat per-instruction costs;
// the review asked for the measurement.)  Each wave runs `iters` iterations of one pattern; W waves per SIMD are enforced with
// dynamic LDS (160 KiB / 4W per one-wave workgroup) and a grid of exactly 1024 W workgroups.  Printed: SIMD cycles per iteration
// per wave (shader clock from s_memtime / s_memrealtime) -- additive costs show as T(pattern) = T(vector part) + n * cost, co-issue
// as T(pattern) = T(vector part).
//   S = v_perm_b32 (the 4.15-cycle class: v_alignbit, v_bfe, v_perm, v_dot2, v_ffbh, v_min3, ...), F = v_add_u32 / v_mov (the 2.3-2.7 class),
//   A = s_add_u32, L = ds_read_b32 (waited for at the end of the iteration), X = s_and_saveexec_b64 + s_cbranch_execz (not taken) + s_or_b64 exec,
//   R = v_readlane_b32 into an SGPR + v_writelane_b32 back (what a spilled scalar register costs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define PAT(body) asm volatile(body : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+s"(sa), "+s"(sb), "+v"(lx), "+s"(sx), "+v"(sp) : "v"(b), "s"(sc), "v"(la), "s"(full) : "scc", "vcc", "memory")
// operands: 0-3 vector chains, 4 sa, 5 sb, 6 lx, 7 sx, 8 sp, 9 b, 10 sc, 11 la, 12 full
#define LW "s_waitcnt lgkmcnt(0)\n\t"
#define S4 "v_perm_b32 %0, %0, %9, %9\n\tv_perm_b32 %1, %1, %9, %9\n\tv_perm_b32 %2, %2, %9, %9\n\tv_perm_b32 %3, %3, %9, %9\n\t"
#define F4 "v_add_u32 %0, %0, %9\n\tv_add_u32 %1, %1, %9\n\tv_add_u32 %2, %2, %9\n\tv_add_u32 %3, %3, %9\n\t"
#define F4E "v_add_u32_e64 %0, %0, %9\n\tv_add_u32_e64 %1, %1, %9\n\tv_add_u32_e64 %2, %2, %9\n\tv_add_u32_e64 %3, %3, %9\n\t"
#define G4 "v_lshrrev_b32 %0, 1, %0\n\tv_sub_u32 %1, %1, %9\n\tv_ashrrev_i32 %2, 1, %2\n\tv_xor_b32 %3, %3, %9\n\t"
#define SF4 "v_perm_b32 %0, %0, %9, %9\n\tv_add_u32 %1, %1, %9\n\tv_perm_b32 %2, %2, %9, %9\n\tv_add_u32 %3, %3, %9\n\t"
#define FS4 "v_add_u32 %0, %0, %9\n\tv_perm_b32 %1, %1, %9, %9\n\tv_add_u32 %2, %2, %9\n\tv_perm_b32 %3, %3, %9, %9\n\t"
#define SSFF4 "v_perm_b32 %0, %0, %9, %9\n\tv_perm_b32 %1, %1, %9, %9\n\tv_add_u32 %2, %2, %9\n\tv_add_u32 %3, %3, %9\n\t"
#define SD4 "v_perm_b32 %0, %0, %9, %9\n\tv_perm_b32 %0, %0, %9, %9\n\tv_perm_b32 %0, %0, %9, %9\n\tv_perm_b32 %0, %0, %9, %9\n\t"
#define A1 "s_add_u32 %4, %4, %10\n\t"
#define A2 "s_add_u32 %4, %4, %10\n\ts_add_u32 %5, %5, %10\n\t"
#define L1 "ds_read_b32 %6, %11\n\t"
#define X1 "s_and_saveexec_b64 %7, %12\n\ts_cbranch_execz 1f\n\t1:\n\ts_or_b64 exec, exec, %7\n\t"
#define R1 "v_readlane_b32 %4, %8, 3\n\ts_nop 0\n\tv_writelane_b32 %8, %4, 3\n\t"

enum { T_S16, T_S16_A4, T_S16_A8, T_S16_A16, T_S16_L1, T_S16_L2, T_S16_L4, T_F16, T_F16_A4, T_F16_A8, T_S8F8, T_MIX, T_S16_X2, T_S16_X4, T_S16_R4, T_SD16, T_SD16_A4, T_A16, T_L4, T_S8_F8, T_SFALT, T_SSFF, T_F16E, T_G16, T_S8_G8, T_COUNT };
static const char* const names[T_COUNT] = {
    "16 S", "16 S + 4 A", "16 S + 8 A", "16 S + 16 A", "16 S + 1 L", "16 S + 2 L", "16 S + 4 L", "16 F", "16 F + 4 A", "16 F + 8 A", "8 S + 8 F",
    "12 S + 4 F + 4 A + 1 L (the kernels' mix)", "16 S + 2 X (6 scalar, 2 of them branches)", "16 S + 4 X", "16 S + 4 R (readlane + writelane)",
    "16 S, ONE dependent chain", "16 S dependent + 4 A", "16 A alone", "4 L alone",
    "8 S then 8 F (runs of eight)", "S F S F ... (alternating, 8 + 8)", "S S F F ... (pairs, 8 + 8)", "16 F in the VOP3 encoding (v_add_u32_e64)",
    "16 G (lshr / sub / ashr / xor, VOP2)", "8 S then 8 G" };

template <int T>
__global__ __launch_bounds__(64) void k_pat(uint64_t* out, int a0, int b0, int iters) {
    extern __shared__ uint32_t lds[];
    int a = a0 + threadIdx.x, b = b0, c = a0 * 3, d = b0 * 5, e = a0 ^ 77, lx = 0, sp = a0;
    uint32_t sa = (uint32_t)a0, sb = (uint32_t)b0, sc = 7u;
    uint64_t sx = 0, full = ~0ull;
    const uint32_t la = 4u * threadIdx.x;
    lds[threadIdx.x] = a0;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (T == T_S16) PAT(S4 S4 S4 S4);
        else if (T == T_S16_A4) PAT(S4 A1 S4 A1 S4 A1 S4 A1);
        else if (T == T_S16_A8) PAT(S4 A2 S4 A2 S4 A2 S4 A2);
        else if (T == T_S16_A16) PAT(S4 A2 A2 S4 A2 A2 S4 A2 A2 S4 A2 A2);
        else if (T == T_S16_L1) PAT(L1 S4 S4 S4 S4 LW);
        else if (T == T_S16_L2) PAT(L1 S4 S4 L1 S4 S4 LW);
        else if (T == T_S16_L4) PAT(L1 S4 L1 S4 L1 S4 L1 S4 LW);
        else if (T == T_F16) PAT(F4 F4 F4 F4);
        else if (T == T_F16_A4) PAT(F4 A1 F4 A1 F4 A1 F4 A1);
        else if (T == T_F16_A8) PAT(F4 A2 F4 A2 F4 A2 F4 A2);
        else if (T == T_S8F8) PAT(S4 F4 S4 F4);
        else if (T == T_MIX) PAT(L1 S4 A1 F4 A1 S4 A1 S4 A1 LW);
        else if (T == T_S16_X2) PAT(S4 S4 X1 S4 S4 X1);
        else if (T == T_S16_X4) PAT(S4 X1 S4 X1 S4 X1 S4 X1);
        else if (T == T_S16_R4) PAT(S4 R1 S4 R1 S4 R1 S4 R1);
        else if (T == T_SD16) PAT(SD4 SD4 SD4 SD4);
        else if (T == T_SD16_A4) PAT(SD4 A1 SD4 A1 SD4 A1 SD4 A1);
        else if (T == T_A16) PAT(A2 A2 A2 A2 A2 A2 A2 A2);
        else if (T == T_L4) PAT(L1 L1 L1 L1 LW);
        else if (T == T_S8_F8) PAT(S4 S4 F4 F4);
        else if (T == T_SFALT) PAT(SF4 FS4 SF4 FS4);
        else if (T == T_SSFF) PAT(SSFF4 SSFF4 SSFF4 SSFF4);
        else if (T == T_F16E) PAT(F4E F4E F4E F4E);
        else if (T == T_G16) PAT(G4 G4 G4 G4);
        else if (T == T_S8_G8) PAT(S4 S4 G4 G4);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (a + c + d + e + lx + sp + (int)sa + (int)sb + (int)sx == 0x7fffffff) out[2] = 1;
}

static double results[T_COUNT][9];
template <int T> void run(uint64_t* d, int W, int iters) {
    const int grid = 1024 * W;
    const size_t lds = (size_t)(160 * 1024 / (4 * W)) & ~255u;         // at most 4 W one-wave workgroups per CU
    (void)hipFuncSetAttribute((const void*)k_pat<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_pat<T><<<grid, 64, lds>>>(d, 3, 5, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k_pat<T><<<grid, 64, lds>>>(d, 3, 5, iters); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double mhz = (double)h[0] / (double)h[1] * 100.0;          // s_memrealtime ticks at 100 MHz
    results[T][W] = ms * 1e-3 * mhz * 1e6 / ((double)W * iters);      // SIMD cycles per iteration per wave
    if (T == 0) printf("shader clock during '16 S' at W=%d: %.0f MHz\n", W, mhz);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}
template <int T> void all(uint64_t* d) {
    const int Ws[5] = { 1, 2, 3, 4, 8 };
    for (int w : Ws) run<T>(d, w, 4000);
    printf("%-46s", names[T]);
    for (int w : Ws) printf("  W=%d %7.2f", w, results[T][w]);
    printf("\n"); fflush(stdout);
    if constexpr (T + 1 < T_COUNT) all<T + 1>(d);
}
int main() {
    uint64_t* d; if (hipMalloc(&d, 64) != hipSuccess) return 1;
    printf("SIMD cycles per iteration per wave (W waves per SIMD); one iteration = the pattern named\n"); fflush(stdout);
    all<0>(d);
    return 0;
}
