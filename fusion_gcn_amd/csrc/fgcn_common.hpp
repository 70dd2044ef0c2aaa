// Shared host/device helpers for libfgcn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/fgcn.h"

namespace fgcn {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// ---- host: error reporting ------------------------------------------------------------------------------
char* error_buffer();  // thread-local, 512 bytes (fgcn_api.hip)

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FGCN_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return FGCN_OK;
}

// Cache policy of the kernels' streamed output stores (the aux operand of the buffer-store builtins: bit 1 = nt).  An activation of
// the 64-clip step is 245 MB, written once and read by a later kernel after as much other traffic: a plain store allocates its line
// in L2 (the matrix kernels write 64-byte row pieces per instruction, i.e. partial lines) and costs the kernel its full HBM write time
// on top of the matrix work; the non-temporal form streams them out (pw_gemm -7 .. -22 % per launch, the joint mixing -19 .. -31 %,
// the spatial forward -3 .. -11 %, the halo conv -0.3 .. -6 % in loops of identical launches; inside the step less, -0.55 ms of 59 in
// all: profiles/r03_ab_store_nt.txt).  A SMALL tensor (the 8-clip shard's 31 MB activations, its 92 MB three-wide 1x1 outputs) is
// still in L2 / Infinity Cache when its consumer starts, and the streamed form loses that (+0.04 .. +0.08 ms on the 9.7 ms step): the
// launchers choose per call from the bytes written (stream_out) between two instantiations of the kernel (template parameter STR; a
// run-time branch around the epilogue cost the tightest kernels spills).  Stores of a load-add-store (accumulating epilogues: the line is resident from the load) and of partial-sum
// rows stay plain.
#ifndef FGCN_STORE_AUX
#define FGCN_STORE_AUX 2
#endif
#define FGCN_REQUIRE(cond, code, ...) \
    do {                              \
        if (!(cond)) return ::fgcn::fail(code, __VA_ARGS__); \
    } while (0)

inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }

// Division of a row index by a run-time constant (joints per frame, rows per sample, frames) without the ~25-instruction sequence a
// 32-bit integer division costs per lane: q = (n * m) >> p with m = ceil(2^p / d), p = 29 + ceil(log2 d) -- exact for every n < 2^29
// (n * (m d - 2^p) < 2^29 d <= 2^p).  One 32 x 32 -> 64-bit multiply and a 64-bit shift.  The halo conv's index prologue had 32 such
// divisions per thread and tile: 930 vector instructions in front of a 64-channel tile's 1296 MFMAs.
struct FastDiv {
    unsigned m;
    int p;
    unsigned d;
};
inline FastDiv make_fastdiv(unsigned d) {
    int L = 0;
    while ((1ull << L) < d) ++L;
    FastDiv f;
    f.p = 29 + L;
    f.m = (unsigned)(((1ull << f.p) + d - 1) / d);
    f.d = d;
    return f;
}
#ifdef FGCN_NO_FASTDIV   // (A/B builds, tools/build_probe.py: the plain integer division)
__device__ __forceinline__ unsigned fastdiv(unsigned n, FastDiv f) { return n / f.d; }
#else
__device__ __forceinline__ unsigned fastdiv(unsigned n, FastDiv f) { return (unsigned)(((unsigned long long)n * f.m) >> f.p); }
#endif

// Kernel-variant selectors (fgcn_set_tuning): defaults are the measured-best variants; tests and tools/kbench.py
// flip them to compare.  key 0: row-GEMM tile for <= 64 output channels, key 1: for wider outputs (see fgcn_gemm.hip);
// key 4: halo conv register budget (0: 3
// workgroups per CU, 1: 2); key 5: XCD-aware workgroup order, bit 0 row GEMM (off), bit 1 halo conv (off), bit 2
// disables it for the weight gradient (on by default), bit 3 disables the column-tile-fastest grid of the row GEMM.
int tuning(int key);
// true when a call that writes `bytes` of output should stream it (tuning key 10: 0 = from 96 MiB on, 1 = never, 2 = always)
inline bool stream_out(long long bytes) {
    const int k = tuning(10);
    return k == 2 || (k == 0 && bytes >= (96ll << 20));
}
int math_mode();   // FGCN_MATH_F32 / FGCN_MATH_BF16 / FGCN_MATH_BF16X3 (fgcn_set_math_mode)
int products();    // FGCN_PRODUCTS_BF16X3 / FGCN_PRODUCTS_F16X2 inside FGCN_MATH_BF16X3 (fgcn_set_products)
inline bool f16x2_products() { return math_mode() == FGCN_MATH_BF16X3 && products() == FGCN_PRODUCTS_F16X2; }

// ---- device: MFMA 32x32x2 f32 -----------------------------------------------------------------------------
// A operand: lane l holds A[i = l & 31][k = l >> 5];  B operand: lane l holds B[k = l >> 5][j = l & 31];
// C/D: lane l, register r holds D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// ---- device: bf16-operand MFMA (FGCN_MATH_BF16) -----------------------------------------------------------------------
// v_mfma_f32_32x32x8_bf16: lane l holds A[i = l & 31][k = 4*(l >> 5) .. +3] and B[k = 4*(l >> 5) .. +3][j = l & 31] as four
// bf16 in two registers; D as for the f32 instruction.  The f32 kernels already give every lane four consecutive
// contraction indices per 16-byte fragment read (lane half h holds k = 8q + 4h + e): one bf16 MFMA replaces the four f32
// MFMAs e = 0..3, operands rounded (RNE, v_cvt_pk_bf16_f32) as the fragment is formed.
using s16x4 = __attribute__((ext_vector_type(4))) short;
__device__ __forceinline__ s16x4 pack_bf16(float a0, float a1, float a2, float a3) {
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
    using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
    const bf16x2 lo = __builtin_convertvector(f32x2{a0, a1}, bf16x2);
    const bf16x2 hi = __builtin_convertvector(f32x2{a2, a3}, bf16x2);
    const u32x2 r = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
    return __builtin_bit_cast(s16x4, r);
}
__device__ __forceinline__ s16x4 pack_bf16(f32x4 a) { return pack_bf16(a[0], a[1], a[2], a[3]); }
// bfloat16 STORAGE helpers (math mode bf16, the `_t` entry points of include/fgcn.h): four bfloat16 (8 bytes) -> four floats (exact), two
// floats -> two bfloat16 in one dword (round to nearest even, element 0 in the low half)
__device__ __forceinline__ f32x4 unpack_bf16x4(__attribute__((ext_vector_type(2))) unsigned h) {
    const unsigned a = h[0], b = h[1];       // (element -> scalar before the bit casts: hipcc 7.2 reads element 0 of a vector element otherwise)
    return f32x4{__builtin_bit_cast(float, a << 16), __builtin_bit_cast(float, a & 0xffff0000u), __builtin_bit_cast(float, b << 16),
                 __builtin_bit_cast(float, b & 0xffff0000u)};
}
// the value of lane ^ 1 (DPP quad_perm [1, 0, 3, 2]: no LDS traffic)
__device__ __forceinline__ float lane_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ unsigned pack_bf16x2(float a0, float a1) {
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a0, a1}, bf16x2));
}
__device__ __forceinline__ f32x16 mfma_bf16(s16x4 a, s16x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c, 0, 0, 0);
}


// FGCN_MATH_BF16X3: f32-accurate products on the bf16 matrix pipe.  x = x_h + x_m + x_l with three bf16 terms (8 + 8 + 8
// significand bits: the split is exact), and a.b is the sum of the six partial products down to 2^-16 of the leading one
// (h.h, h.m, m.h, m.m, h.l, l.h); the dropped ones (m.l, l.m, l.l) are below 2^-23 |a.b|, i.e. below the rounding of an
// f32 product.  Every partial product is exact in the f32 accumulator.  Six bf16 MFMAs (16 cycles each) replace four f32
// MFMAs (64 cycles each): 2.67x the f32 matrix rate at f32 accuracy.
// Frag<MM>: the operand fragment of one bf16 MFMA (4 consecutive k per lane) in math mode MM (1: rounded, 2: split).
// The production kernels of this mode use v_mfma_f32_32x32x16_bf16 (8 k per lane: lane (r, h) holds k = 8h + j, 32 cycles
// per instruction = the full bf16 rate) on operands that were split ONCE -- weights by fgcn_pack_split3 in HBM, activation
// tiles as they are staged into LDS: splitting a fragment in registers costs 22 vector instructions per 4 values, which
// only pays where the fragment feeds many MFMAs.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using u32x4v = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ f32x16 mfma_bf16_k16(u32x4v a, u32x4v b, f32x16 c) {
#ifdef FGCN_PROBE_16X16   // timing probe only (wrong numerics): the same FLOPs as two 16x16x32 instructions
    f32x4 lo = {c[0], c[1], c[2], c[3]}, hi = {c[4], c[5], c[6], c[7]};
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), hi, 0, 0, 0);
    c[0] = lo[0]; c[1] = lo[1]; c[2] = lo[2]; c[3] = lo[3]; c[4] = hi[0]; c[5] = hi[1]; c[6] = hi[2]; c[7] = hi[3];
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}
// the six partial products of a three-way split pair (index 0 = high, 1 = middle, 2 = low part), small terms first
__device__ __forceinline__ f32x16 mfma_x3_k16(const u32x4v (&a)[3], const u32x4v (&b)[3], f32x16 c) {
    c = mfma_bf16_k16(a[2], b[0], c);
    c = mfma_bf16_k16(a[0], b[2], c);
    c = mfma_bf16_k16(a[1], b[1], c);
    c = mfma_bf16_k16(a[1], b[0], c);
    c = mfma_bf16_k16(a[0], b[1], c);
    return mfma_bf16_k16(a[0], b[0], c);
}

// the same six partial products on v_mfma_f32_16x16x32_bf16 (lane (i, g = lane >> 4) holds k = 8g + j; accumulator register r of
// lane (col, g) = row 4g + r): half the accumulator traffic per FLOP of the 32x32x16 form -- under these kernels the chip
// holds its clock with it where the 32x32 form makes it throttle
__device__ __forceinline__ f32x4 mfma_bf16_k32(u32x4v a, u32x4v b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_x3_k32(const u32x4v (&a)[3], const u32x4v (&b)[3], f32x4 c) {
    c = mfma_bf16_k32(a[2], b[0], c);
    c = mfma_bf16_k32(a[0], b[2], c);
    c = mfma_bf16_k32(a[1], b[1], c);
    c = mfma_bf16_k32(a[1], b[0], c);
    c = mfma_bf16_k32(a[0], b[1], c);
    return mfma_bf16_k32(a[0], b[0], c);
}

// ---- FGCN_MATH_F16X2: f32 products from two-way f16 splits, block-scaled -----------------------------------------------------------
// x * 2^e = h + l with h = f16(x 2^e), l = f16(x 2^e - h): 11 + 11 significand bits and the sign of l give |x 2^e - h - l| <= 2^-24 |x 2^e|
// while l stays a normal f16 (|x 2^e| >= 2^-2), an absolute 2^-25 below; a.b from three products (l.h, h.l, h.h; l.l <= 2^-24 |a.b| is
// dropped) -- half of bf16x3's matrix work.  f16 has 5 exponent bits, so every operand block is scaled by a power of two (exact) that
// puts its largest magnitude into [2^14, 2^15): activations per staged (tile, channel chunk) inside the kernels, weights per packed
// form (fgcn_pack_run_scaled); the accumulator carries the scale and the epilogue removes it.
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
__device__ __forceinline__ f32x4 mfma_f16_k32(u32x4v a, u32x4v b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_h2_k32(const u32x4v (&a)[2], const u32x4v (&b)[2], f32x4 c) {
    c = mfma_f16_k32(a[1], b[0], c);
    c = mfma_f16_k32(a[0], b[1], c);
    return mfma_f16_k32(a[0], b[0], c);
}
// the same three products on v_mfma_f32_32x32x16_f16 (the fused spatial forward's shape)
__device__ __forceinline__ f32x16 mfma_f16_k16(u32x4v a, u32x4v b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_h2_k16(const u32x4v (&a)[2], const u32x4v (&b)[2], f32x16 c) {
    c = mfma_f16_k16(a[1], b[0], c);
    c = mfma_f16_k16(a[0], b[1], c);
    return mfma_f16_k16(a[0], b[0], c);
}
__device__ __forceinline__ void split_f16_pair(float a0, float a1, unsigned& h, unsigned& l) {
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
    const f16x2 hh = __builtin_convertvector(f32x2{a0, a1}, f16x2);
    const f32x2 back = __builtin_convertvector(hh, f32x2);
    const f16x2 ll = __builtin_convertvector(f32x2{a0 - back[0], a1 - back[1]}, f16x2);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, ll);
}
__device__ __forceinline__ void split2h_x4(f32x4 a, u32x2& h, u32x2& l) {
    unsigned h0, l0, h1, l1;
    split_f16_pair(a[0], a[1], h0, l0);
    split_f16_pair(a[2], a[3], h1, l1);
    h = u32x2{h0, h1};
    l = u32x2{l0, l1};
}

// power-of-two scale 2^s with amax * 2^s in [2^14, 2^15) from the float bits of amax >= 0 (0, denormals: s = 0); returned as the
// exponent s (so that callers can take min / differences) -- exp2i(s) is the float
__device__ __forceinline__ int scale_exp_for(unsigned amax_bits) {
    const int e = (int)((amax_bits >> 23) & 0xffu);           // biased exponent: amax in [2^(e-127), 2^(e-126))
    return e == 0 ? 0 : 141 - e;                              // 14 - (e - 127)
}
__device__ __forceinline__ float exp2i(int s) {               // 2^s for s in [-126, 127]
    s = s < -126 ? -126 : (s > 127 ? 127 : s);
    return __builtin_bit_cast(float, (unsigned)(s + 127) << 23);
}

// eight values (one lane's fragment) -> the two f16 parts, 16 bytes each
__device__ __forceinline__ void split2h_x8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7, float sc,
                                           u32x4v (&q)[2]) {
    u32x2 h0, l0, h1, l1;
    split2h_x4(f32x4{v0, v1, v2, v3} * sc, h0, l0);
    split2h_x4(f32x4{v4, v5, v6, v7} * sc, h1, l1);
    q[0] = u32x4v{h0[0], h0[1], h1[0], h1[1]};
    q[1] = u32x4v{l0[0], l0[1], l1[0], l1[1]};
}
__device__ __forceinline__ float wave_max_abs16(const float (&v)[16], float m) {
#pragma unroll
    for (int i = 0; i < 16; ++i) m = fmaxf(m, fabsf(v[i]));
    return m;
}
__device__ __forceinline__ float wave_reduce_max(float m) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    return m;
}

template <int MM> struct Frag;
template <> struct Frag<1> { s16x4 h; };
template <> struct Frag<2> { s16x4 h, m, l; };

__device__ __forceinline__ void split_bf16_pair(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a0, a1}, bf16x2));
    const float r0 = a0 - __builtin_bit_cast(float, h << 16), r1 = a1 - __builtin_bit_cast(float, h & 0xffff0000u);
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
    const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{q0, q1}, bf16x2));
}

// four values -> their high / middle / low bf16 parts, 8 bytes each (element order kept)
__device__ __forceinline__ void split3_x4(f32x4 a, u32x2& h, u32x2& m, u32x2& l) {
    unsigned h0, m0, l0, h1, m1, l1;
    split_bf16_pair(a[0], a[1], h0, m0, l0);
    split_bf16_pair(a[2], a[3], h1, m1, l1);
    h = u32x2{h0, h1};
    m = u32x2{m0, m1};
    l = u32x2{l0, l1};
}

// eight values (one lane's k = 8h + j fragment of v_mfma_f32_32x32x16_bf16) -> the three bf16 parts, 16 bytes each
__device__ __forceinline__ void split3_x8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7,
                                          u32x4v (&q)[3]) {
    u32x2 h0, m0, l0, h1, m1, l1;
    split3_x4(f32x4{v0, v1, v2, v3}, h0, m0, l0);
    split3_x4(f32x4{v4, v5, v6, v7}, h1, m1, l1);
    q[0] = u32x4v{h0[0], h0[1], h1[0], h1[1]};
    q[1] = u32x4v{m0[0], m0[1], m1[0], m1[1]};
    q[2] = u32x4v{l0[0], l0[1], l1[0], l1[1]};
}

// The same helpers with the number of bf16 parts as a template parameter: NP = 3 is the exact three-way split above (FGCN_MATH_BF16X3),
// NP = 1 one bfloat16 per value, round-to-nearest-even (FGCN_MATH_BF16: part 0 of the split IS the rounded value) -- one kernel source
// for both modes (the tile kernels of fgcn_emb_tile.hip).
template <int NP>
__device__ __forceinline__ void splitn_x4(f32x4 a, u32x2 (&q)[NP]) {
    if constexpr (NP == 3) split3_x4(a, q[0], q[1], q[2]);
    else q[0] = __builtin_bit_cast(u32x2, pack_bf16(a));
}
template <int NP>
__device__ __forceinline__ void splitn_x8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7, u32x4v (&q)[NP]) {
    if constexpr (NP == 3) {
        split3_x8(v0, v1, v2, v3, v4, v5, v6, v7, q);
    } else {
        const u32x2 lo = __builtin_bit_cast(u32x2, pack_bf16(v0, v1, v2, v3)), hi = __builtin_bit_cast(u32x2, pack_bf16(v4, v5, v6, v7));
        q[0] = u32x4v{lo[0], lo[1], hi[0], hi[1]};
    }
}
template <int NP>
__device__ __forceinline__ f32x4 mfma_np_k32(const u32x4v (&a)[NP], const u32x4v (&b)[NP], f32x4 c) {
    if constexpr (NP == 3) return mfma_x3_k32(a, b, c);
    else return mfma_bf16_k32(a[0], b[0], c);
}
template <int NP>
__device__ __forceinline__ f32x16 mfma_np_k16(const u32x4v (&a)[NP], const u32x4v (&b)[NP], f32x16 c) {
    if constexpr (NP == 3) return mfma_x3_k16(a, b, c);
    else return mfma_bf16_k16(a[0], b[0], c);
}

template <int MM>
__device__ __forceinline__ Frag<MM> make_frag(float a0, float a1, float a2, float a3) {
    Frag<MM> f;
    if constexpr (MM == 1) {
        f.h = pack_bf16(a0, a1, a2, a3);
    } else {
        unsigned h0, m0, l0, h1, m1, l1;
        split_bf16_pair(a0, a1, h0, m0, l0);
        split_bf16_pair(a2, a3, h1, m1, l1);
        f.h = __builtin_bit_cast(s16x4, u32x2{h0, h1});
        f.m = __builtin_bit_cast(s16x4, u32x2{m0, m1});
        f.l = __builtin_bit_cast(s16x4, u32x2{l0, l1});
    }
    return f;
}
template <int MM>
__device__ __forceinline__ Frag<MM> make_frag(f32x4 a) { return make_frag<MM>(a[0], a[1], a[2], a[3]); }

template <int MM>
__device__ __forceinline__ f32x16 mfma_frag(const Frag<MM>& a, const Frag<MM>& b, f32x16 c) {
    if constexpr (MM == 2) {          // small terms first
        c = mfma_bf16(a.l, b.h, c);
        c = mfma_bf16(a.h, b.l, c);
        c = mfma_bf16(a.m, b.m, c);
        c = mfma_bf16(a.m, b.h, c);
        c = mfma_bf16(a.h, b.m, c);
    }
    return mfma_bf16(a.h, b.h, c);
}

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// Temporal index map (see fgcn_tmap in fgcn.h): returns ti or -1.
__device__ __forceinline__ int tmap_src(int to, int tap, int ta, int tb, int tc, int td, int T_in) {
    const int num = to * ta + tap * tb + tc;
    if (num < 0) return -1;
    int ti = num;
    if (td != 1) {
        if (num % td) return -1;
        ti = num / td;
    }
    return ti < T_in ? ti : -1;
}

}  // namespace fgcn
