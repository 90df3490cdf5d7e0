// iir_sched.hip -- how fast can ONE wave run the order-8 integer recurrence of K2's predictor, as a function of the
// instruction schedule?  (s_memtime ticks = shader cycles.)  K2's duration is one wave's serial chain, so the cycles per
// sample of this loop are the kernel's floor.
//   cur   : what K2 does today: per sample ONE dependent chain of 8 v_mad_i32_i24 (newest tap last), s_nop, ashr, add
//   toep  : "Toeplitz" form: every finished sample is multiplied into the 8 running sums of the samples that follow it;
//           only mad(c0) -> ashr -> add is on the critical path, the other 7 mads are independent of each other
//   toep2 : the same, the critical instructions spread between the independent ones by hand
//   tree  : two partial sums (taps 7..4, taps 3..1) + critical mad, compiler scheduled
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../claxon_amd/csrc/intrin/clx_intrin.h"

#define NS 4096
__device__ __forceinline__ int32_t mad24(int32_t a, int32_t b, int32_t c) { return __mul24(a, b) + c; }

template <int VAR>
__global__ __launch_bounds__(64) void k_iir(uint64_t* tim, int32_t* out, const int32_t* in, int sh) {
    int32_t c[8], h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { c[j] = in[j * 64 + threadIdx.x] >> 20; h[j] = 0; }
    int32_t P[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) P[j] = 0;
    int32_t acc = 0, sp = 0;
    int32_t pr[7] = {0, 0, 0, 0, 0, 0, 0}, C2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) C2[j] = (int32_t)(((uint32_t)c[2 * j] << 16) | ((uint32_t)c[2 * j + 1] & 0xffffu));
    const int32_t* xin = in + 512 + threadIdx.x;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < NS; t += 16) {
        int32_t x[16], y[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = xin[((t + i) & 63) * 64] >> 18;
        if (VAR == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int32_t a = clx_dot24z<8>(c, h);
                const int32_t s = x[i] + (a >> sh);
#pragma unroll
                for (int j = 7; j > 0; --j) h[j] = h[j - 1];
                h[0] = s; y[i] = s;
            }
        } else if (VAR == 1) {
            // P[d]: sum of the contributions of all finished samples to the prediction of the sample d steps ahead
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int32_t s = x[i] + (P[0] >> sh);
#pragma unroll
                for (int d = 0; d < 7; ++d) P[d] = mad24(c[d], s, P[d + 1]);
                P[7] = __mul24(c[7], s);
                y[i] = s;
            }
        } else if (VAR == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int32_t s;
                // critical: ashr, add; then the mad that completes the next prediction first, the rest behind it
                asm volatile("v_ashrrev_i32 %0, %9, %1\n\t"
                             "v_add_u32 %0, %0, %10\n\t"
                             "v_mad_i32_i24 %1, %11, %0, %2\n\t"
                             "v_mad_i32_i24 %2, %12, %0, %3\n\t"
                             "v_mad_i32_i24 %3, %13, %0, %4\n\t"
                             "v_mad_i32_i24 %4, %14, %0, %5\n\t"
                             "v_mad_i32_i24 %5, %15, %0, %6\n\t"
                             "v_mad_i32_i24 %6, %16, %0, %7\n\t"
                             "v_mad_i32_i24 %7, %17, %0, %8\n\t"
                             "v_mul_i32_i24 %8, %18, %0"
                             : "=&v"(s), "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]), "+v"(P[4]), "+v"(P[5]), "+v"(P[6]), "+v"(P[7])
                             : "v"(sh), "v"(x[i]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]));
                y[i] = s;
            }
        } else if (VAR == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int32_t A = mad24(c[4], h[4], mad24(c[5], h[5], mad24(c[6], h[6], __mul24(c[7], h[7]))));
                const int32_t B = mad24(c[1], h[1], mad24(c[2], h[2], __mul24(c[3], h[3])));
                const int32_t s = x[i] + ((mad24(c[0], h[0], A) + B) >> sh);
#pragma unroll
                for (int j = 7; j > 0; --j) h[j] = h[j - 1];
                h[0] = s; y[i] = s;
            }
        } else if (VAR == 4) {
            // two samples in flight: sample i's critical instructions interleaved with the independent mads of sample i-1
            // (s_prev's contributions to P[2..7] are not needed by sample i's prediction beyond P[0], P[1])
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int32_t s;
                asm volatile("v_ashrrev_i32 %0, %9, %1\n\t"
                             "v_add_u32 %0, %0, %10\n\t"
                             "v_mad_i32_i24 %1, %11, %0, %2\n\t"
                             "v_mad_i32_i24 %2, %12, %0, %3\n\t"
                             "v_mad_i32_i24 %3, %13, %0, %4\n\t"
                             "v_mad_i32_i24 %4, %14, %0, %5\n\t"
                             : "=&v"(s), "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]), "+v"(P[4]), "+v"(P[5]), "+v"(P[6]), "+v"(P[7])
                             : "v"(sh), "v"(x[i]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]));
                asm volatile("v_mad_i32_i24 %0, %5, %4, %1\n\t"
                             "v_mad_i32_i24 %1, %6, %4, %2\n\t"
                             "v_mad_i32_i24 %2, %7, %4, %3\n\t"
                             "v_mul_i32_i24 %3, %8, %4"
                             : "+v"(P[4]), "+v"(P[5]), "+v"(P[6]), "+v"(P[7])
                             : "v"(s), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]));
                y[i] = s;
            }
        } else if (VAR == 5) {
            // software pipelined by hand, 4 samples per statement: sample i's critical instructions (ashr, add, mad c0) alternate
            // with the updates of P3..P7 by sample i-1, which nothing waits for
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                asm volatile("v_ashrrev_i32 %0, %13, %4\n\tv_mad_i32_i24 %7, %21, %12, %8\n\tv_add_u32 %0, %0, %14\n\tv_mad_i32_i24 %8, %22, %12, %9\n\tv_mad_i32_i24 %4, %18, %0, %5\n\tv_mad_i32_i24 %9, %23, %12, %10\n\tv_mad_i32_i24 %5, %19, %0, %6\n\tv_mad_i32_i24 %10, %24, %12, %11\n\tv_mad_i32_i24 %6, %20, %0, %7\n\tv_mul_i32_i24 %11, %25, %12\n\tv_ashrrev_i32 %1, %13, %4\n\tv_mad_i32_i24 %7, %21, %0, %8\n\tv_add_u32 %1, %1, %15\n\tv_mad_i32_i24 %8, %22, %0, %9\n\tv_mad_i32_i24 %4, %18, %1, %5\n\tv_mad_i32_i24 %9, %23, %0, %10\n\tv_mad_i32_i24 %5, %19, %1, %6\n\tv_mad_i32_i24 %10, %24, %0, %11\n\tv_mad_i32_i24 %6, %20, %1, %7\n\tv_mul_i32_i24 %11, %25, %0\n\tv_ashrrev_i32 %2, %13, %4\n\tv_mad_i32_i24 %7, %21, %1, %8\n\tv_add_u32 %2, %2, %16\n\tv_mad_i32_i24 %8, %22, %1, %9\n\tv_mad_i32_i24 %4, %18, %2, %5\n\tv_mad_i32_i24 %9, %23, %1, %10\n\tv_mad_i32_i24 %5, %19, %2, %6\n\tv_mad_i32_i24 %10, %24, %1, %11\n\tv_mad_i32_i24 %6, %20, %2, %7\n\tv_mul_i32_i24 %11, %25, %1\n\tv_ashrrev_i32 %3, %13, %4\n\tv_mad_i32_i24 %7, %21, %2, %8\n\tv_add_u32 %3, %3, %17\n\tv_mad_i32_i24 %8, %22, %2, %9\n\tv_mad_i32_i24 %4, %18, %3, %5\n\tv_mad_i32_i24 %9, %23, %2, %10\n\tv_mad_i32_i24 %5, %19, %3, %6\n\tv_mad_i32_i24 %10, %24, %2, %11\n\tv_mad_i32_i24 %6, %20, %3, %7\n\tv_mul_i32_i24 %11, %25, %2\n\tv_mov_b32 %12, %3"
                             : "=&v"(y[i]), "=&v"(y[i + 1]), "=&v"(y[i + 2]), "=&v"(y[i + 3]),
                               "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]), "+v"(P[4]), "+v"(P[5]), "+v"(P[6]), "+v"(P[7]), "+v"(sp)
                             : "v"(sh), "v"(x[i]), "v"(x[i + 1]), "v"(x[i + 2]), "v"(x[i + 3]),
                               "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]));
            }
        } else if (VAR == 6) {
            // v_dot2_i32_i16: two taps per instruction on history kept as packed pairs of 16-bit samples (valid while every
            // sample fits 16 bits); 16 samples in ONE statement (no compiler padding between instructions)
            int32_t acc_t, tmp_p;
            asm volatile("v_dot2_i32_i16 %24, %45, %22, 0\n\tv_dot2_i32_i16 %24, %44, %20, %24\n\tv_dot2_i32_i16 %24, %43, %18, %24\n\tv_dot2_i32_i16 %24, %42, %16, %24\n\tv_ashrrev_i32 %24, %46, %24\n\tv_add_u32 %0, %24, %26\n\tv_perm_b32 %25, %0, %23, %47\n\tv_dot2_i32_i16 %24, %45, %21, 0\n\tv_dot2_i32_i16 %24, %44, %19, %24\n\tv_dot2_i32_i16 %24, %43, %17, %24\n\tv_dot2_i32_i16 %24, %42, %25, %24\n\tv_ashrrev_i32 %24, %46, %24\n\tv_add_u32 %1, %24, %27\n\tv_perm_b32 %22, %1, %0, %47\n\tv_dot2_i32_i16 %24, %45, %20, 0\n\tv_dot2_i32_i16 %24, %44, %18, %24\n\tv_dot2_i32_i16 %24, %43, %16, %24\n\tv_dot2_i32_i16 %24, %42, %22, %24\n\tv_ashrrev_i32 %24, %46, %24\n\tv_add_u32 %2, %24, %28\n\tv_perm_b32 %21, %2, %1, %47\n\tv_dot2_i32_i16 %24, %45, %19, 0\n\tv_dot2_i32_i16 %24, %44, %17, %24\n\tv_dot2_i32_i16 %24, %43, %25, %24\n\tv_dot2_i32_i16 %24, %42, %21, %24\n\tv_ashrrev_i32 %24, %46, %24\n\tv_add_u32 %3, %24, %29\n\tv_perm_b32 %20, %3, %2, %47\n\tv_dot2_i32_i16 %24, %45, %18, 0\n\tv_dot2_i32_i16 %24, %44, %16, %24\n\tv_dot2_i32_i16 %24, %43, %22, %24\n\tv_dot2_i32_i16 %24, %42, %20, %24\n\tv_ashrrev_i32 %24, %46, %24\n\tv_add_u32 %4, %24, %30\n\tv_perm_b32 %19, %4, %3, %47\n\tv_dot2_i32_i16 %24, %45, %17, 0\n\tv_dot2_i32_i16 %24, %44, %25, %24\n\tv_dot2_i32_i16 %24, %43, %21, %24\n\tv_dot2_i32_i16 %24, %42, %19, %24\n\tv_ashrrev_i32 %24, %46, %24\n\tv_add_u32 %5, %24, %31\n\tv_perm_b32 %18, %5, %4, %47\n\tv_dot2_i32_i16 %24, %45, %16, 0\n\tv_dot2_i32_i16 %24, %44, %22, %24\n\tv_dot2_i32_i16 %24, %43, %20, %24\n\tv_dot2_i32_i16 %24, %42, %18, %24\n\tv_ashrrev_i32 %24, %46, %24\n\tv_add_u32 %6, %24, %32\n\tv_perm_b32 %17, %6, %5, %47\n\tv_dot2_i32_i16 %24, %45, %25, 0\n\tv_dot2_i32_i16 %24, %44, %21, %24\n\tv_dot2_i32_i16 %24, %43, %19, %24\n\tv_dot2_i32_i16 %24, %42, %17, %24\n\tv_ashrrev_i32 %24, %46, %24\n\tv_add_u32 %7, %24, %33\n\tv_perm_b32 %16, %7, %6, %47\n\tv_dot2_i32_i16 %24, %45, %22, 0\n\tv_dot2_i32_i16 %24, %44, %20, %24\n\tv_dot2_i32_i16 %24, %43, %18, %24\n\tv_dot2_i32_i16 %24, %42, %16, %24\n\tv_ashrrev_i32 %24, %46, %24\n\tv_add_u32 %8, %24, %34\n\tv_perm_b32 %25, %8, %7, %47\n\tv_dot2_i32_i16 %24, %45, %21, 0\n\tv_dot2_i32_i16 %24, %44, %19, %24\n\tv_dot2_i32_i16 %24, %43, %17, %24\n\tv_dot2_i32_i16 %24, %42, %25, %24\n\tv_ashrrev_i32 %24, %46, %24\n\tv_add_u32 %9, %24, %35\n\tv_perm_b32 %22, %9, %8, %47\n\tv_dot2_i32_i16 %24, %45, %20, 0\n\tv_dot2_i32_i16 %24, %44, %18, %24\n\tv_dot2_i32_i16 %24, %43, %16, %24\n\tv_dot2_i32_i16 %24, %42, %22, %24\n\tv_ashrrev_i32 %24, %46, %24\n\tv_add_u32 %10, %24, %36\n\tv_perm_b32 %21, %10, %9, %47\n\tv_dot2_i32_i16 %24, %45, %19, 0\n\tv_dot2_i32_i16 %24, %44, %17, %24\n\tv_dot2_i32_i16 %24, %43, %25, %24\n\tv_dot2_i32_i16 %24, %42, %21, %24\n\tv_ashrrev_i32 %24, %46, %24\n\tv_add_u32 %11, %24, %37\n\tv_perm_b32 %20, %11, %10, %47\n\tv_dot2_i32_i16 %24, %45, %18, 0\n\tv_dot2_i32_i16 %24, %44, %16, %24\n\tv_dot2_i32_i16 %24, %43, %22, %24\n\tv_dot2_i32_i16 %24, %42, %20, %24\n\tv_ashrrev_i32 %24, %46, %24\n\tv_add_u32 %12, %24, %38\n\tv_perm_b32 %19, %12, %11, %47\n\tv_dot2_i32_i16 %24, %45, %17, 0\n\tv_dot2_i32_i16 %24, %44, %25, %24\n\tv_dot2_i32_i16 %24, %43, %21, %24\n\tv_dot2_i32_i16 %24, %42, %19, %24\n\tv_ashrrev_i32 %24, %46, %24\n\tv_add_u32 %13, %24, %39\n\tv_perm_b32 %18, %13, %12, %47\n\tv_dot2_i32_i16 %24, %45, %16, 0\n\tv_dot2_i32_i16 %24, %44, %22, %24\n\tv_dot2_i32_i16 %24, %43, %20, %24\n\tv_dot2_i32_i16 %24, %42, %18, %24\n\tv_ashrrev_i32 %24, %46, %24\n\tv_add_u32 %14, %24, %40\n\tv_perm_b32 %17, %14, %13, %47\n\tv_dot2_i32_i16 %24, %45, %25, 0\n\tv_dot2_i32_i16 %24, %44, %21, %24\n\tv_dot2_i32_i16 %24, %43, %19, %24\n\tv_dot2_i32_i16 %24, %42, %17, %24\n\tv_ashrrev_i32 %24, %46, %24\n\tv_add_u32 %15, %24, %41\n\tv_perm_b32 %16, %15, %14, %47"
                         : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3]), "=&v"(y[4]), "=&v"(y[5]), "=&v"(y[6]), "=&v"(y[7]),
                           "=&v"(y[8]), "=&v"(y[9]), "=&v"(y[10]), "=&v"(y[11]), "=&v"(y[12]), "=&v"(y[13]), "=&v"(y[14]), "=&v"(y[15]),
                           "+v"(pr[0]), "+v"(pr[1]), "+v"(pr[2]), "+v"(pr[3]), "+v"(pr[4]), "+v"(pr[5]), "+v"(pr[6]), "+v"(sp),
                           "=&v"(acc_t), "=&v"(tmp_p)
                         : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
                           "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]), "v"(x[13]), "v"(x[14]), "v"(x[15]),
                           "v"(C2[0]), "v"(C2[1]), "v"(C2[2]), "v"(C2[3]), "v"(sh), "v"(0x05040100));
            sp = y[15];
        } else if (VAR == 7) {
            // today's chain (8 x v_mad_i32_i24, newest tap last), but 16 samples in ONE statement: no compiler padding
            int32_t acc_t;
            asm volatile("v_mad_i32_i24 %16, %40, %48, 0\n\tv_mad_i32_i24 %16, %39, %47, %16\n\tv_mad_i32_i24 %16, %38, %46, %16\n\tv_mad_i32_i24 %16, %37, %45, %16\n\tv_mad_i32_i24 %16, %36, %44, %16\n\tv_mad_i32_i24 %16, %35, %43, %16\n\tv_mad_i32_i24 %16, %34, %42, %16\n\tv_mad_i32_i24 %16, %33, %41, %16\n\tv_ashrrev_i32 %16, %49, %16\n\tv_add_u32 %0, %16, %17\n\tv_mad_i32_i24 %16, %40, %47, 0\n\tv_mad_i32_i24 %16, %39, %46, %16\n\tv_mad_i32_i24 %16, %38, %45, %16\n\tv_mad_i32_i24 %16, %37, %44, %16\n\tv_mad_i32_i24 %16, %36, %43, %16\n\tv_mad_i32_i24 %16, %35, %42, %16\n\tv_mad_i32_i24 %16, %34, %41, %16\n\tv_mad_i32_i24 %16, %33, %0, %16\n\tv_ashrrev_i32 %16, %49, %16\n\tv_add_u32 %1, %16, %18\n\tv_mad_i32_i24 %16, %40, %46, 0\n\tv_mad_i32_i24 %16, %39, %45, %16\n\tv_mad_i32_i24 %16, %38, %44, %16\n\tv_mad_i32_i24 %16, %37, %43, %16\n\tv_mad_i32_i24 %16, %36, %42, %16\n\tv_mad_i32_i24 %16, %35, %41, %16\n\tv_mad_i32_i24 %16, %34, %0, %16\n\tv_mad_i32_i24 %16, %33, %1, %16\n\tv_ashrrev_i32 %16, %49, %16\n\tv_add_u32 %2, %16, %19\n\tv_mad_i32_i24 %16, %40, %45, 0\n\tv_mad_i32_i24 %16, %39, %44, %16\n\tv_mad_i32_i24 %16, %38, %43, %16\n\tv_mad_i32_i24 %16, %37, %42, %16\n\tv_mad_i32_i24 %16, %36, %41, %16\n\tv_mad_i32_i24 %16, %35, %0, %16\n\tv_mad_i32_i24 %16, %34, %1, %16\n\tv_mad_i32_i24 %16, %33, %2, %16\n\tv_ashrrev_i32 %16, %49, %16\n\tv_add_u32 %3, %16, %20\n\tv_mad_i32_i24 %16, %40, %44, 0\n\tv_mad_i32_i24 %16, %39, %43, %16\n\tv_mad_i32_i24 %16, %38, %42, %16\n\tv_mad_i32_i24 %16, %37, %41, %16\n\tv_mad_i32_i24 %16, %36, %0, %16\n\tv_mad_i32_i24 %16, %35, %1, %16\n\tv_mad_i32_i24 %16, %34, %2, %16\n\tv_mad_i32_i24 %16, %33, %3, %16\n\tv_ashrrev_i32 %16, %49, %16\n\tv_add_u32 %4, %16, %21\n\tv_mad_i32_i24 %16, %40, %43, 0\n\tv_mad_i32_i24 %16, %39, %42, %16\n\tv_mad_i32_i24 %16, %38, %41, %16\n\tv_mad_i32_i24 %16, %37, %0, %16\n\tv_mad_i32_i24 %16, %36, %1, %16\n\tv_mad_i32_i24 %16, %35, %2, %16\n\tv_mad_i32_i24 %16, %34, %3, %16\n\tv_mad_i32_i24 %16, %33, %4, %16\n\tv_ashrrev_i32 %16, %49, %16\n\tv_add_u32 %5, %16, %22\n\tv_mad_i32_i24 %16, %40, %42, 0\n\tv_mad_i32_i24 %16, %39, %41, %16\n\tv_mad_i32_i24 %16, %38, %0, %16\n\tv_mad_i32_i24 %16, %37, %1, %16\n\tv_mad_i32_i24 %16, %36, %2, %16\n\tv_mad_i32_i24 %16, %35, %3, %16\n\tv_mad_i32_i24 %16, %34, %4, %16\n\tv_mad_i32_i24 %16, %33, %5, %16\n\tv_ashrrev_i32 %16, %49, %16\n\tv_add_u32 %6, %16, %23\n\tv_mad_i32_i24 %16, %40, %41, 0\n\tv_mad_i32_i24 %16, %39, %0, %16\n\tv_mad_i32_i24 %16, %38, %1, %16\n\tv_mad_i32_i24 %16, %37, %2, %16\n\tv_mad_i32_i24 %16, %36, %3, %16\n\tv_mad_i32_i24 %16, %35, %4, %16\n\tv_mad_i32_i24 %16, %34, %5, %16\n\tv_mad_i32_i24 %16, %33, %6, %16\n\tv_ashrrev_i32 %16, %49, %16\n\tv_add_u32 %7, %16, %24\n\tv_mad_i32_i24 %16, %40, %0, 0\n\tv_mad_i32_i24 %16, %39, %1, %16\n\tv_mad_i32_i24 %16, %38, %2, %16\n\tv_mad_i32_i24 %16, %37, %3, %16\n\tv_mad_i32_i24 %16, %36, %4, %16\n\tv_mad_i32_i24 %16, %35, %5, %16\n\tv_mad_i32_i24 %16, %34, %6, %16\n\tv_mad_i32_i24 %16, %33, %7, %16\n\tv_ashrrev_i32 %16, %49, %16\n\tv_add_u32 %8, %16, %25\n\tv_mad_i32_i24 %16, %40, %1, 0\n\tv_mad_i32_i24 %16, %39, %2, %16\n\tv_mad_i32_i24 %16, %38, %3, %16\n\tv_mad_i32_i24 %16, %37, %4, %16\n\tv_mad_i32_i24 %16, %36, %5, %16\n\tv_mad_i32_i24 %16, %35, %6, %16\n\tv_mad_i32_i24 %16, %34, %7, %16\n\tv_mad_i32_i24 %16, %33, %8, %16\n\tv_ashrrev_i32 %16, %49, %16\n\tv_add_u32 %9, %16, %26\n\tv_mad_i32_i24 %16, %40, %2, 0\n\tv_mad_i32_i24 %16, %39, %3, %16\n\tv_mad_i32_i24 %16, %38, %4, %16\n\tv_mad_i32_i24 %16, %37, %5, %16\n\tv_mad_i32_i24 %16, %36, %6, %16\n\tv_mad_i32_i24 %16, %35, %7, %16\n\tv_mad_i32_i24 %16, %34, %8, %16\n\tv_mad_i32_i24 %16, %33, %9, %16\n\tv_ashrrev_i32 %16, %49, %16\n\tv_add_u32 %10, %16, %27\n\tv_mad_i32_i24 %16, %40, %3, 0\n\tv_mad_i32_i24 %16, %39, %4, %16\n\tv_mad_i32_i24 %16, %38, %5, %16\n\tv_mad_i32_i24 %16, %37, %6, %16\n\tv_mad_i32_i24 %16, %36, %7, %16\n\tv_mad_i32_i24 %16, %35, %8, %16\n\tv_mad_i32_i24 %16, %34, %9, %16\n\tv_mad_i32_i24 %16, %33, %10, %16\n\tv_ashrrev_i32 %16, %49, %16\n\tv_add_u32 %11, %16, %28\n\tv_mad_i32_i24 %16, %40, %4, 0\n\tv_mad_i32_i24 %16, %39, %5, %16\n\tv_mad_i32_i24 %16, %38, %6, %16\n\tv_mad_i32_i24 %16, %37, %7, %16\n\tv_mad_i32_i24 %16, %36, %8, %16\n\tv_mad_i32_i24 %16, %35, %9, %16\n\tv_mad_i32_i24 %16, %34, %10, %16\n\tv_mad_i32_i24 %16, %33, %11, %16\n\tv_ashrrev_i32 %16, %49, %16\n\tv_add_u32 %12, %16, %29\n\tv_mad_i32_i24 %16, %40, %5, 0\n\tv_mad_i32_i24 %16, %39, %6, %16\n\tv_mad_i32_i24 %16, %38, %7, %16\n\tv_mad_i32_i24 %16, %37, %8, %16\n\tv_mad_i32_i24 %16, %36, %9, %16\n\tv_mad_i32_i24 %16, %35, %10, %16\n\tv_mad_i32_i24 %16, %34, %11, %16\n\tv_mad_i32_i24 %16, %33, %12, %16\n\tv_ashrrev_i32 %16, %49, %16\n\tv_add_u32 %13, %16, %30\n\tv_mad_i32_i24 %16, %40, %6, 0\n\tv_mad_i32_i24 %16, %39, %7, %16\n\tv_mad_i32_i24 %16, %38, %8, %16\n\tv_mad_i32_i24 %16, %37, %9, %16\n\tv_mad_i32_i24 %16, %36, %10, %16\n\tv_mad_i32_i24 %16, %35, %11, %16\n\tv_mad_i32_i24 %16, %34, %12, %16\n\tv_mad_i32_i24 %16, %33, %13, %16\n\tv_ashrrev_i32 %16, %49, %16\n\tv_add_u32 %14, %16, %31\n\tv_mad_i32_i24 %16, %40, %7, 0\n\tv_mad_i32_i24 %16, %39, %8, %16\n\tv_mad_i32_i24 %16, %38, %9, %16\n\tv_mad_i32_i24 %16, %37, %10, %16\n\tv_mad_i32_i24 %16, %36, %11, %16\n\tv_mad_i32_i24 %16, %35, %12, %16\n\tv_mad_i32_i24 %16, %34, %13, %16\n\tv_mad_i32_i24 %16, %33, %14, %16\n\tv_ashrrev_i32 %16, %49, %16\n\tv_add_u32 %15, %16, %32"
                         : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3]), "=&v"(y[4]), "=&v"(y[5]), "=&v"(y[6]), "=&v"(y[7]),
                           "=&v"(y[8]), "=&v"(y[9]), "=&v"(y[10]), "=&v"(y[11]), "=&v"(y[12]), "=&v"(y[13]), "=&v"(y[14]), "=&v"(y[15]), "=&v"(acc_t)
                         : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
                           "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]), "v"(x[13]), "v"(x[14]), "v"(x[15]),
                           "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]),
                           "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]), "v"(h[4]), "v"(h[5]), "v"(h[6]), "v"(h[7]), "v"(sh));
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = y[15 - j];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc ^= y[i];
    }
    asm volatile("s_nop 0" ::: "memory");
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) tim[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int VAR> static void run(const char* name, int grid, uint64_t* d_t, int32_t* d_o, const int32_t* d_in, std::vector<int32_t>* ref) {
    k_iir<VAR><<<grid, 64>>>(d_t, d_o, d_in, 9);
    k_iir<VAR><<<grid, 64>>>(d_t, d_o, d_in, 9);
    (void)hipDeviceSynchronize();
    std::vector<uint64_t> t(grid); std::vector<int32_t> o(64);
    (void)hipMemcpy(t.data(), d_t, grid * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(o.data(), d_o, 64 * 4, hipMemcpyDeviceToHost);
    uint64_t mx = 0; double av = 0; for (auto v : t) { mx = v > mx ? v : mx; av += (double)v; }
    bool same = true;
    if (ref->empty()) *ref = o; else same = (*ref == o);
    printf("%-6s grid %5d: %7.1f cycles/sample avg, %7.1f max   %s\n", name, grid, av / grid / NS, (double)mx / NS, same ? "same result" : "RESULT DIFFERS");
}

int main() {
    uint64_t* d_t; int32_t* d_o; int32_t* d_in;
    if (hipMalloc(&d_t, 16384 * 8) != hipSuccess) return 1;
    (void)hipMalloc(&d_o, 16384 * 64 * 4); (void)hipMalloc(&d_in, (512 + 64 * 64) * 4);
    std::vector<int32_t> in(512 + 64 * 64);
    uint32_t s = 12345; for (auto& v : in) { s = s * 1664525u + 1013904223u; v = (int32_t)s; }
    (void)hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    std::vector<int32_t> ref;
    for (int grid : {1, 313, 1024, 2048}) {
        run<0>("cur", grid, d_t, d_o, d_in, &ref); run<1>("toep", grid, d_t, d_o, d_in, &ref); run<2>("toep2", grid, d_t, d_o, d_in, &ref);
        run<3>("tree", grid, d_t, d_o, d_in, &ref); run<4>("toep4", grid, d_t, d_o, d_in, &ref); run<5>("pipe", grid, d_t, d_o, d_in, &ref); run<6>("dot2", grid, d_t, d_o, d_in, &ref); run<7>("mad1s", grid, d_t, d_o, d_in, &ref);
    }
    return 0;
}
