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

#define FGCN_REQUIRE(cond, code, ...) \
    do {                              \
        if (!(cond)) return ::fgcn::fail(code, __VA_ARGS__); \
    } while (0)

inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }

// Kernel-variant selectors (fgcn_set_tuning): defaults are the measured-best variants; tests and tools/kbench.py
// flip them to compare.  key 0: row-GEMM tile for <= 64 output channels, key 1: for wider outputs (see fgcn_gemm.hip);
// key 4: halo conv register budget (0: 3
// workgroups per CU, 1: 2); key 5: XCD-aware workgroup order, bit 0 row GEMM (off), bit 1 halo conv (off), bit 2
// disables it for the weight gradient (on by default), bit 3 disables the column-tile-fastest grid of the row GEMM.
int tuning(int key);
int math_mode();   // FGCN_MATH_F32 / FGCN_MATH_BF16 (fgcn_set_math_mode)

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
__device__ __forceinline__ f32x16 mfma_bf16(s16x4 a, s16x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c, 0, 0, 0);
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
