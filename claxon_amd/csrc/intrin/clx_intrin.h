// clx_intrin.h -- gfx950 instruction wrappers used by the kernels (resolved via -I .../csrc/intrin).
#ifndef CLX_INTRIN_H
#define CLX_INTRIN_H
#include <stdint.h>

// v_alignbit_b32: low 32 bits of ({hi,lo} >> (shift & 31))
__device__ __forceinline__ uint32_t clx_alignbit(uint32_t hi, uint32_t lo, uint32_t shift) {
    return __builtin_amdgcn_alignbit(hi, lo, shift);
}
// v_bfe_u32: (src >> offset) & ((1 << width) - 1); width 0 -> 0
__device__ __forceinline__ uint32_t clx_bfe(uint32_t src, uint32_t offset, uint32_t width) {
    return __builtin_amdgcn_ubfe(src, offset, width);
}
// nothing is scheduled across this point (the compiler otherwise orders a basic block by its own latency model)
#define CLX_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// v_perm_b32: byte i of the result is byte sel[8i+7:8i] of the 8-byte value {hi, lo} (selectors 0-3: lo, 4-7: hi)
__device__ __forceinline__ uint32_t clx_perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// v_mad_i32_i24: low 32 bits of sext24(a)*sext24(b) + c.  Inline asm so that the accumulation stays a chain in
// the order written (oldest tap first, newest last): only the last mad then depends on the newest sample.
__device__ __forceinline__ int32_t clx_mad24(int32_t a, int32_t b, int32_t c) {
    int32_t d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// v_max3_i32 / v_min3_i32: plain C++ that the compiler folds into the three-operand forms (an asm statement each would make it
// pad the boundaries between dependent statements with s_nop)
__device__ __forceinline__ int32_t clx_max3(int32_t a, int32_t b, int32_t c) { const int32_t m = a > b ? a : b; return m > c ? m : c; }
__device__ __forceinline__ int32_t clx_min3(int32_t a, int32_t b, int32_t c) { const int32_t m = a < b ? a : b; return m < c ? m : c; }
// Dot product of N 24-bit factors pairs as ONE asm statement: acc = sum_j c[j]*h[j], evaluated oldest tap (j = N-1) first
// so that the newest sample h[0] is needed last.  One statement per chain because hipcc pads every boundary between
// dependent asm statements with an s_nop (it cannot see what the instruction inside is): a mad per statement costs a
// wasted issue slot for every two mads.
template <int N> __device__ __forceinline__ int32_t clx_dot24(const int32_t* c, const int32_t* h, int32_t acc);
template <> __device__ __forceinline__ int32_t clx_dot24<4>(const int32_t* c, const int32_t* h, int32_t acc) {
    asm volatile("v_mad_i32_i24 %0, %1, %2, %0\n\tv_mad_i32_i24 %0, %3, %4, %0\n\tv_mad_i32_i24 %0, %5, %6, %0\n\tv_mad_i32_i24 %0, %7, %8, %0"
                 : "+v"(acc) : "v"(c[3]), "v"(h[3]), "v"(c[2]), "v"(h[2]), "v"(c[1]), "v"(h[1]), "v"(c[0]), "v"(h[0]));
    return acc;
}
template <> __device__ __forceinline__ int32_t clx_dot24<8>(const int32_t* c, const int32_t* h, int32_t acc) {
    asm volatile("v_mad_i32_i24 %0, %1, %2, %0\n\tv_mad_i32_i24 %0, %3, %4, %0\n\tv_mad_i32_i24 %0, %5, %6, %0\n\tv_mad_i32_i24 %0, %7, %8, %0\n\t"
                 "v_mad_i32_i24 %0, %9, %10, %0\n\tv_mad_i32_i24 %0, %11, %12, %0\n\tv_mad_i32_i24 %0, %13, %14, %0\n\tv_mad_i32_i24 %0, %15, %16, %0"
                 : "+v"(acc) : "v"(c[7]), "v"(h[7]), "v"(c[6]), "v"(h[6]), "v"(c[5]), "v"(h[5]), "v"(c[4]), "v"(h[4]),
                               "v"(c[3]), "v"(h[3]), "v"(c[2]), "v"(h[2]), "v"(c[1]), "v"(h[1]), "v"(c[0]), "v"(h[0]));
    return acc;
}
template <> __device__ __forceinline__ int32_t clx_dot24<12>(const int32_t* c, const int32_t* h, int32_t acc) {
    return clx_dot24<8>(c, h, clx_dot24<4>(c + 8, h + 8, acc));
}
template <> __device__ __forceinline__ int32_t clx_dot24<32>(const int32_t* c, const int32_t* h, int32_t acc) {
    acc = clx_dot24<8>(c + 24, h + 24, acc); acc = clx_dot24<8>(c + 16, h + 16, acc);
    acc = clx_dot24<8>(c + 8, h + 8, acc);   return clx_dot24<8>(c, h, acc);
}
// The same starting from zero: the first mad takes the constant 0 as its addend, which saves the v_mov that would
// otherwise clear the accumulator for every sample.
template <int N> __device__ __forceinline__ int32_t clx_dot24z(const int32_t* c, const int32_t* h);
template <> __device__ __forceinline__ int32_t clx_dot24z<4>(const int32_t* c, const int32_t* h) {
    int32_t acc;
    asm volatile("v_mad_i32_i24 %0, %1, %2, 0\n\tv_mad_i32_i24 %0, %3, %4, %0\n\tv_mad_i32_i24 %0, %5, %6, %0\n\tv_mad_i32_i24 %0, %7, %8, %0"
                 : "=&v"(acc) : "v"(c[3]), "v"(h[3]), "v"(c[2]), "v"(h[2]), "v"(c[1]), "v"(h[1]), "v"(c[0]), "v"(h[0]));
    return acc;
}
template <> __device__ __forceinline__ int32_t clx_dot24z<8>(const int32_t* c, const int32_t* h) {
    int32_t acc;
    asm volatile("v_mad_i32_i24 %0, %1, %2, 0\n\tv_mad_i32_i24 %0, %3, %4, %0\n\tv_mad_i32_i24 %0, %5, %6, %0\n\tv_mad_i32_i24 %0, %7, %8, %0\n\t"
                 "v_mad_i32_i24 %0, %9, %10, %0\n\tv_mad_i32_i24 %0, %11, %12, %0\n\tv_mad_i32_i24 %0, %13, %14, %0\n\tv_mad_i32_i24 %0, %15, %16, %0"
                 : "=&v"(acc) : "v"(c[7]), "v"(h[7]), "v"(c[6]), "v"(h[6]), "v"(c[5]), "v"(h[5]), "v"(c[4]), "v"(h[4]),
                                "v"(c[3]), "v"(h[3]), "v"(c[2]), "v"(h[2]), "v"(c[1]), "v"(h[1]), "v"(c[0]), "v"(h[0]));
    return acc;
}
template <> __device__ __forceinline__ int32_t clx_dot24z<12>(const int32_t* c, const int32_t* h) {
    return clx_dot24<8>(c, h, clx_dot24z<4>(c + 8, h + 8));
}
template <> __device__ __forceinline__ int32_t clx_dot24z<32>(const int32_t* c, const int32_t* h) {
    int32_t acc = clx_dot24z<8>(c + 24, h + 24); acc = clx_dot24<8>(c + 16, h + 16, acc);
    acc = clx_dot24<8>(c + 8, h + 8, acc);       return clx_dot24<8>(c, h, acc);
}
// Mid/side reconstruction of one sample for a lane pair (even lane = mid -> left, odd lane = side -> right; frame.rs:371-389):
//   m = (mid << 1) | (side & 1);   y = (m + (odd ? -side : side)) >> 1        with sgn = odd ? ~0 : 0, nsg = odd ? 1 : 0
// Six instructions: the partner's value enters the AND / XOR as a DPP operand instead of through v_mov_dpp copies (which
// also need their destination cleared first).  The leading s_nop covers the two wait states a DPP read needs after a
// VALU write of the same register -- hipcc cannot see into the statement.
__device__ __forceinline__ int32_t clx_ms_pair(int32_t y, uint32_t sgn, uint32_t nsg, uint32_t one) {
    int32_t out; uint32_t t, x, mid;
    asm volatile("s_nop 1\n\t"
                 "v_and_b32_dpp %1, %4, %7 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_xor_b32_dpp %2, %4, %5 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %3, %4 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_lshl_or_b32 %3, %3, 1, %1\n\t"
                 "v_add3_u32 %0, %3, %2, %6\n\t"
                 "v_ashrrev_i32 %0, 1, %0"
                 : "=&v"(out), "=&v"(t), "=&v"(x), "=&v"(mid) : "v"(y), "v"(sgn), "v"(nsg), "v"(one));
    return out;
}
// v_dot2_i32_i16 (v_dot2c_i32_i16): a.lo16 * b.lo16 + a.hi16 * b.hi16 + acc, signed 16-bit factors, 32-bit wrapping sum.  A
// builtin, not asm: the compiler then knows the instruction (wait states behind its result, scheduling around it).
typedef short clx_short2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int32_t clx_sdot2(uint32_t a, uint32_t b, int32_t acc) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(clx_short2, a), __builtin_bit_cast(clx_short2, b), acc, false);
}
// The first term of a chain, acc = a.lo16 * b.lo16 + a.hi16 * b.hi16: the VOP3P form with the constant 0 as its addend.  The builtin
// becomes v_dot2c_i32_i16 (VOP2: the accumulator is also the destination), which costs a v_mov to clear the accumulator for
// every sample -- one instruction in thirty.  (Nothing has to sit between a v_dot2 and a reader of its result on gfx950:
// tools/ubench/dot2_hazard.hip.)
__device__ __forceinline__ int32_t clx_sdot2_first(uint32_t a, uint32_t b) {
    int32_t d;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// Raw buffer over the arena: 32-bit byte offsets per lane against one wave-uniform descriptor, and loads that reach past the
// end of the allocation return zeros instead of faulting (what a lane with a damaged frame descriptor may ask for).
typedef uint32_t clx_u32x4 __attribute__((ext_vector_type(4)));
struct clx_buf { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ clx_buf clx_make_buf(const void* base, uint32_t bytes) {
    clx_buf b; b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000); return b;
}
__device__ __forceinline__ uint4 clx_buf_load16(const clx_buf& b, uint32_t byte_off) {
    const clx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)byte_off, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// Four samples at once: one statement, so the two wait states in front of the first DPP read are paid once per four.
__device__ __forceinline__ void clx_ms_pair4(const int32_t (&y)[4], int32_t (&out)[4], uint32_t sgn, uint32_t nsg, uint32_t one) {
    uint32_t t, x, mid;
    asm volatile("s_nop 1\n\t"
                 "v_and_b32_dpp %4, %7, %13 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_xor_b32_dpp %5, %7, %11 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %6, %7 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_lshl_or_b32 %6, %6, 1, %4\n\t"
                 "v_add3_u32 %0, %6, %5, %12\n\t"
                 "v_ashrrev_i32 %0, 1, %0\n\t"
                 "v_and_b32_dpp %4, %8, %13 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_xor_b32_dpp %5, %8, %11 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %6, %8 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_lshl_or_b32 %6, %6, 1, %4\n\t"
                 "v_add3_u32 %1, %6, %5, %12\n\t"
                 "v_ashrrev_i32 %1, 1, %1\n\t"
                 "v_and_b32_dpp %4, %9, %13 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_xor_b32_dpp %5, %9, %11 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %6, %9 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_lshl_or_b32 %6, %6, 1, %4\n\t"
                 "v_add3_u32 %2, %6, %5, %12\n\t"
                 "v_ashrrev_i32 %2, 1, %2\n\t"
                 "v_and_b32_dpp %4, %10, %13 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_xor_b32_dpp %5, %10, %11 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %6, %10 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_lshl_or_b32 %6, %6, 1, %4\n\t"
                 "v_add3_u32 %3, %6, %5, %12\n\t"
                 "v_ashrrev_i32 %3, 1, %3"
                 : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(t), "=&v"(x), "=&v"(mid)
                 : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(sgn), "v"(nsg), "v"(one));
}
// Wave vote on a bool without the detour through an int (s_and with exec + a scalar branch, no vector instruction).
__device__ __forceinline__ bool clx_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// The value of x as something the optimiser cannot see through (no instruction): keeps an expression in the shape written.
#define CLX_OPAQUE(x) asm volatile("" : "+v"(x))
#define CLX_OPAQUE_PTR(p) asm volatile("" : "+s"(p))      // the same for a wave-uniform pointer
// the kernel's argument block where the dispatch packet put it (a kernel whose only argument is one T by value): read-only, scalar loads
#define CLX_KERNARGS(T) ((const T*)(const void*)__builtin_amdgcn_kernarg_segment_ptr())
// Four 16-byte stores as ONE asm statement.  hipcc makes the next write of a register that a store of its own has read wait for
// that store's COMPLETION (s_waitcnt vmcnt(0): a round trip to the L2) -- the hardware only needs the two wait states behind the
// last store of the statement (the data is read when the store issues).  An asm store is not in hipcc's vmcnt bookkeeping: its
// counted waits for LOADS can then only be longer than needed, never shorter (loads return in order among themselves), so the
// stores are placed where the next counted wait is a turn away (clx_lean.hip).
typedef int clx_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void clx_store4x16(int32_t* p0, int32_t* p1, int32_t* p2, int32_t* p3, const int4& w0, const int4& w1, const int4& w2, const int4& w3) {
    const clx_i32x4 a = { w0.x, w0.y, w0.z, w0.w }, b = { w1.x, w1.y, w1.z, w1.w }, c = { w2.x, w2.y, w2.z, w2.w }, d = { w3.x, w3.y, w3.z, w3.w };
    asm volatile("global_store_dwordx4 %0, %4, off\n\tglobal_store_dwordx4 %1, %5, off\n\tglobal_store_dwordx4 %2, %6, off\n\t"
                 "global_store_dwordx4 %3, %7, off\n\ts_nop 1"
                 :: "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(a), "v"(b), "v"(c), "v"(d) : "memory");
}
// the same with a wave-uniform base and 32-bit byte offsets (global_store ... saddr: address = scalar base + zero-extended vector offset)
// The decode kernels' output stores carry the streaming hint (`nt`, round 6): a decoded line is never read again, and while it sat in
// the L2 it pushed out the lanes' INPUT lines -- a lane takes its stream 16 bytes at a time, eight requests to a 128-byte line, tens of
// microseconds apart.  rocprofv3 FETCH_SIZE of clx_k_lean in ONE merged launch of twelve runs: 155 MB per run without the hint, 65 MB
// with it (the run's input is 51.5 MB); the pipelined steps 1.5-3 % shorter, builds alternating on two boxes
// (profiles/r06_store_policy_ab.txt).  `sc1` and `sc0 sc1 nt` were measured beside it: nothing / the same.  -DCLX_STORE_PLAIN: without.
#ifdef CLX_STORE_PLAIN
#define CLX_STORE_POLICY ""
#else
#define CLX_STORE_POLICY " nt"
#endif
__device__ __forceinline__ void clx_store4x16_s(uint64_t base, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3, const int4& w0, const int4& w1, const int4& w2, const int4& w3) {
    const clx_i32x4 a = { w0.x, w0.y, w0.z, w0.w }, b = { w1.x, w1.y, w1.z, w1.w }, c = { w2.x, w2.y, w2.z, w2.w }, d = { w3.x, w3.y, w3.z, w3.w };
    asm volatile("global_store_dwordx4 %0, %4, %8" CLX_STORE_POLICY "\n\tglobal_store_dwordx4 %1, %5, %8" CLX_STORE_POLICY "\n\tglobal_store_dwordx4 %2, %6, %8" CLX_STORE_POLICY "\n\t"
                 "global_store_dwordx4 %3, %7, %8" CLX_STORE_POLICY "\n\ts_nop 1"
                 :: "v"(o0), "v"(o1), "v"(o2), "v"(o3), "v"(a), "v"(b), "v"(c), "v"(d), "s"(base) : "memory");
}
// Four (two) ds_bpermute_b32 off ONE address register: r[k] = the value of v in lane (byte_addr + Ok) / 4, byte_addr + Ok < 256.  Written as
// address + constant the constant becomes the instruction's offset field; __shfl(v, lane term + constant) masks the lane number first, and the
// compiler then computes an address register per read and keeps all of them alive across the decode loop (the movers' eight row places: eight
// registers of the 168, some of them spilled).
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void clx_bperm4(uint32_t byte_addr, uint32_t v, uint32_t (&r)[4]) {
    r[0] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(byte_addr + (uint32_t)O0), (int)v); r[1] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(byte_addr + (uint32_t)O1), (int)v);
    r[2] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(byte_addr + (uint32_t)O2), (int)v); r[3] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(byte_addr + (uint32_t)O3), (int)v);
}
template <int O0, int O1>
__device__ __forceinline__ void clx_bperm2(uint32_t byte_addr, uint32_t v, uint32_t (&r)[2]) {
    r[0] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(byte_addr + (uint32_t)O0), (int)v); r[1] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(byte_addr + (uint32_t)O1), (int)v);
}
__device__ __forceinline__ void clx_store1x16_s(uint64_t base, uint32_t o, const int4& w) {
    const clx_i32x4 a = { w.x, w.y, w.z, w.w };
    asm volatile("global_store_dwordx4 %0, %1, %2" CLX_STORE_POLICY "\n\ts_nop 1" :: "v"(o), "v"(a), "s"(base) : "memory");
}
// Mid/side reconstruction of four samples for lane pairs (even lane = mid -> left, odd lane = side -> right), the short form:
//   left = mid + ((side + 1) >> 1),   right = mid - (side >> 1) = mid + ((-side + 1) >> 1)
// which equals frame.rs:382-384's ((2 mid | side & 1) +- side) / 2 while nothing wraps (|mid|, |side| < 2^29: the caller's range
// check).  sgn = odd lane ? ~0 : 0, c = odd lane ? 2 : 1, so that both are  mid + (((side ^ sgn) + c) >> 1).  Four instructions
// per sample; one statement per four samples (the two wait states in front of the first DPP read are paid once).
__device__ __forceinline__ void clx_ms_short4(const int32_t (&y)[4], int32_t (&out)[4], uint32_t sgn, uint32_t c) {
    uint32_t t;
    asm volatile("s_nop 1\n\t"
                 "v_xor_b32_dpp %4, %5, %9 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_u32 %4, %4, %10\n\t"
                 "v_ashrrev_i32 %4, 1, %4\n\t"
                 "v_add_u32_dpp %0, %5, %4 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_xor_b32_dpp %4, %6, %9 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_u32 %4, %4, %10\n\t"
                 "v_ashrrev_i32 %4, 1, %4\n\t"
                 "v_add_u32_dpp %1, %6, %4 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_xor_b32_dpp %4, %7, %9 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_u32 %4, %4, %10\n\t"
                 "v_ashrrev_i32 %4, 1, %4\n\t"
                 "v_add_u32_dpp %2, %7, %4 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_xor_b32_dpp %4, %8, %9 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_u32 %4, %4, %10\n\t"
                 "v_ashrrev_i32 %4, 1, %4\n\t"
                 "v_add_u32_dpp %3, %8, %4 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf"
                 : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(t)
                 : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(sgn), "v"(c));
}
// v_ffbh_u32 as it comes: the number of leading zeros, 0xffffffff for 0 (__clz adds a v_min for that case; __builtin_clz leaves it
// undefined).  Not volatile: the compiler schedules it like any other instruction.
__device__ __forceinline__ uint32_t clx_ffbh(uint32_t x) {
    uint32_t z;
    asm("v_ffbh_u32 %0, %1" : "=v"(z) : "v"(x));
    return z;
}
// Any stereo decorrelation (frame.rs:319-389) of four samples, or none, AND the wasted-bits shift (subframe.rs:216-225), by three
// per-lane constants:   out = (own * mo + other * mt + c) >> 1   (other = the value of the lane's partner, lane ^ 1)
//     mid/side    even (mid, w wasted bits; the side's: v): mo = 2 << w, mt = 1 << v, c = 1    odd (side): mo = -(1 << w), mt = 2 << v, c = 1
//                 -- (2 mid + side + 1) >> 1 and (2 mid - side + 1) >> 1 are frame.rs:382-384's ((2 mid | side & 1) +- side) / 2
//     left/side   even: mo = 2 << w, mt = 0      odd: mo = -(2 << w), mt = 2 << v  (left - side)
//     right/side  even: mo = 2 << w, mt = 2 << v (side + right)      odd: mo = 2 << w, mt = 0
//     no partner  mo = 2 << w, mt = 0
// with v_mad_i32_i24: exact while both values lie in [-2^23, 2^23) BEFORE and AFTER their shifts (the caller's range check: then
// nothing wraps here and nothing wraps in the reference's wrapping arithmetic).  Four instructions per sample, one of them DPP.
__device__ __forceinline__ void clx_decor4_mad(const int32_t (&y)[4], int32_t (&out)[4], int32_t mo, int32_t mt, int32_t c) {
    int32_t t;
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mad_i32_i24 %0, %5, %9, %11\n\t"
                 "v_mad_i32_i24 %0, %4, %10, %0\n\t"
                 "v_ashrrev_i32 %0, 1, %0\n\t"
                 "v_mov_b32_dpp %4, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mad_i32_i24 %1, %6, %9, %11\n\t"
                 "v_mad_i32_i24 %1, %4, %10, %1\n\t"
                 "v_ashrrev_i32 %1, 1, %1\n\t"
                 "v_mov_b32_dpp %4, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mad_i32_i24 %2, %7, %9, %11\n\t"
                 "v_mad_i32_i24 %2, %4, %10, %2\n\t"
                 "v_ashrrev_i32 %2, 1, %2\n\t"
                 "v_mov_b32_dpp %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mad_i32_i24 %3, %8, %9, %11\n\t"
                 "v_mad_i32_i24 %3, %4, %10, %3\n\t"
                 "v_ashrrev_i32 %3, 1, %3"
                 : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(t)
                 : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(mo), "v"(mt), "v"(c));
}
// The same for eight samples, instruction class by instruction class (round 6): the plain VOP2 adds, subtractions and shifts issue at
// twice the rate of everything else -- but only in runs of their own kind (tools/ubench/coissue.hip, profiles/r06_ubench_coissue.txt: 8 S + 8 F
// in runs of eight 60.7 cycles per wave at 8 waves per SIMD and 74.6 / 77.9 at 2 / 3, alternating 74.2 and 108.9 / 99.2) -- so the eight v_ashrrev sit in a row.  (own * mo + other * mt + c = other * mt + c, then
// + own * mo: the partner's value lands in the output register, no temporary.)
__device__ __forceinline__ void clx_decor8_mad(const int32_t (&y)[8], int32_t (&out)[8], int32_t mo, int32_t mt, int32_t c) {
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %1, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %2, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %3, %11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %4, %12 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %5, %13 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %6, %14 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %7, %15 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mad_i32_i24 %0, %0, %17, %18\n\t"
                 "v_mad_i32_i24 %1, %1, %17, %18\n\t"
                 "v_mad_i32_i24 %2, %2, %17, %18\n\t"
                 "v_mad_i32_i24 %3, %3, %17, %18\n\t"
                 "v_mad_i32_i24 %4, %4, %17, %18\n\t"
                 "v_mad_i32_i24 %5, %5, %17, %18\n\t"
                 "v_mad_i32_i24 %6, %6, %17, %18\n\t"
                 "v_mad_i32_i24 %7, %7, %17, %18\n\t"
                 "v_mad_i32_i24 %0, %8, %16, %0\n\t"
                 "v_mad_i32_i24 %1, %9, %16, %1\n\t"
                 "v_mad_i32_i24 %2, %10, %16, %2\n\t"
                 "v_mad_i32_i24 %3, %11, %16, %3\n\t"
                 "v_mad_i32_i24 %4, %12, %16, %4\n\t"
                 "v_mad_i32_i24 %5, %13, %16, %5\n\t"
                 "v_mad_i32_i24 %6, %14, %16, %6\n\t"
                 "v_mad_i32_i24 %7, %15, %16, %7\n\t"
                 "v_ashrrev_i32 %0, 1, %0\n\t"
                 "v_ashrrev_i32 %1, 1, %1\n\t"
                 "v_ashrrev_i32 %2, 1, %2\n\t"
                 "v_ashrrev_i32 %3, 1, %3\n\t"
                 "v_ashrrev_i32 %4, 1, %4\n\t"
                 "v_ashrrev_i32 %5, 1, %5\n\t"
                 "v_ashrrev_i32 %6, 1, %6\n\t"
                 "v_ashrrev_i32 %7, 1, %7"
                 : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5]), "=&v"(out[6]), "=&v"(out[7])
                 : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(mo), "v"(mt), "v"(c));
}
// Any stereo decorrelation (frame.rs:319-389) of four samples, or none, by per-lane constants: in a pair of lanes (channel 0 in the
// even one) the value that is added or subtracted is always the odd lane's and the value it is applied to the even lane's, so
//     out = (even & pmask) + ((((odd ^ sg) & rmask) + c) >> s1)
// covers mid/side (both lanes: rmask = pmask = ~0, s1 = 1, sg / c = 0 / 1 in the even lane, ~0 / 2 in the odd one), left/side
// (even: rmask = 0; odd: sg = ~0, c = 1 -- left - side), right/side (even: side + right; odd: pmask = 0 -- its own value is the
// odd one) and lanes without a partner (even: rmask = 0; odd: pmask = 0), while nothing wraps in the mid/side form (below 2^29:
// the caller's range check).  Six instructions per sample, two of them DPP.
__device__ __forceinline__ void clx_decor4(const int32_t (&y)[4], int32_t (&out)[4], uint32_t sg, uint32_t rmask, uint32_t c, uint32_t s1, uint32_t pmask) {
    uint32_t t, e;
    asm volatile("s_nop 1\n\t"
                 "v_xor_b32_dpp %4, %6, %10 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_and_b32_dpp %5, %6, %14 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_and_b32 %4, %4, %11\n\t"
                 "v_add_u32 %4, %4, %12\n\t"
                 "v_ashrrev_i32 %4, %13, %4\n\t"
                 "v_add_u32 %0, %5, %4\n\t"
                 "v_xor_b32_dpp %4, %7, %10 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_and_b32_dpp %5, %7, %14 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_and_b32 %4, %4, %11\n\t"
                 "v_add_u32 %4, %4, %12\n\t"
                 "v_ashrrev_i32 %4, %13, %4\n\t"
                 "v_add_u32 %1, %5, %4\n\t"
                 "v_xor_b32_dpp %4, %8, %10 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_and_b32_dpp %5, %8, %14 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_and_b32 %4, %4, %11\n\t"
                 "v_add_u32 %4, %4, %12\n\t"
                 "v_ashrrev_i32 %4, %13, %4\n\t"
                 "v_add_u32 %2, %5, %4\n\t"
                 "v_xor_b32_dpp %4, %9, %10 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
                 "v_and_b32_dpp %5, %9, %14 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_and_b32 %4, %4, %11\n\t"
                 "v_add_u32 %4, %4, %12\n\t"
                 "v_ashrrev_i32 %4, %13, %4\n\t"
                 "v_add_u32 %3, %5, %4"
                 : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(t), "=&v"(e)
                 : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(sg), "v"(rmask), "v"(c), "v"(s1), "v"(pmask));
}
// LDS-DMA: every lane copies 16 bytes from its own global address straight into LDS at lds_base + 16*lane (no VGPR
// round trip, asynchronous, counted by vmcnt).  Inline asm on purpose: hipcc drains vmcnt(0) before the next LDS read
// when it can see the DMA, which would serialise a prefetch ring; with asm the waits are placed by hand
// (clx_wait_vmcnt).  M0 carries the LDS base and is saved / restored inside the statement (it is compiler-reserved).
__device__ __forceinline__ void clx_glds16(const void* gsrc, uint32_t lds_byte_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
// byte address of a __shared__ object inside the workgroup's LDS allocation (wave-uniform)
__device__ __forceinline__ uint32_t clx_lds_addr(const void* p) {
    return __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p);
}
template <int N> __device__ __forceinline__ void clx_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void clx_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// wave-level ordering point for LDS traffic of a one-wave workgroup: no instruction, only stops the compiler from
// moving LDS accesses across it (the LDS queue itself is in order per wave)
__device__ __forceinline__ void clx_wave_sync() { __builtin_amdgcn_wave_barrier(); }
// workgroup barrier for waves that hand LDS data to each other, without the fences of __syncthreads() (hipcc drains
// vmcnt(0) there, which would stall the LDS-DMA prefetch ring): LDS writes are made visible by the lgkmcnt wait
__device__ __forceinline__ void clx_wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// A value the code knows to be wave-uniform, moved to a scalar register: everything computed from it afterwards runs on
// the scalar unit instead of taking VALU issue slots (K1 is VALU-issue bound).  The simulator checks the claim.
__device__ __forceinline__ uint32_t clx_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// what waves of one kernel tell each other through global memory (clx_k_pool): a counter read past the caches, a pause between two
// looks at it, and the fences on both sides of the hand-over (agent scope: gfx950 has an L2 per XCD)
__device__ __forceinline__ uint32_t clx_peek_u32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void clx_poke_u32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void clx_stores_done() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// what the lanes of ONE wave (a one-wave workgroup) wrote to global memory for each other: made visible before it is read
__device__ __forceinline__ void clx_group_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void clx_pause() { __builtin_amdgcn_s_sleep(32); }
__device__ __forceinline__ void clx_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
__device__ __forceinline__ void clx_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
// lane `idx` (wave-uniform) of v, as a scalar
__device__ __forceinline__ uint32_t clx_readlane(uint32_t v, uint32_t idx) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)idx); }
#endif
