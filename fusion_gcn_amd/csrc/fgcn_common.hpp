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

// ---- device: MFMA 32x32x2 f32 -----------------------------------------------------------------------------
// A operand: lane l holds A[i = l & 31][k = l >> 5];  B operand: lane l holds B[k = l >> 5][j = l & 31];
// C/D: lane l, register r holds D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
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
