// Probe: does v_pk_fma_f32 work issued by a second wave of the same SIMD overlap with a wave that is saturating the f32
// matrix pipe (v_mfma_f32_32x32x2_f32)?  8 waves per workgroup (2 per SIMD); mode 0: waves 0-3 MFMA, 4-7 idle;
// mode 1: waves 0-3 idle, 4-7 packed FMA; mode 2: both.   hipcc --offload-arch=gfx950 -O3 ... && ./probe
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

__global__ __launch_bounds__(512) void probe(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode == 0 || mode == 2) {
            f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
            const float x = (float)threadIdx.x * 1e-3f, y = 1.0001f;
            for (int i = 0; i < iters; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
            }
            r = a0[0] + a1[1] + a2[2] + a3[3];
        }
    } else {
        if (mode == 1 || mode == 2) {
            f32x2 c[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) c[j] = f32x2{(float)j, (float)threadIdx.x};
            const f32x2 a = {1.0001f, 0.9999f}, b = {1e-6f, -1e-6f};
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 16; ++j) c[j] = __builtin_elementwise_fma(c[j], a, b);   // 16 independent packed FMAs
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) r += c[j][0] + c[j][1];
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    float* d;
    const int blocks = 256 * 1, iters = 20000;
    hipMalloc(&d, blocks * 512 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, d, 100, mode);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, d, iters, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfma = (mode != 1) ? 256.0 * 4 * iters * 4 * 4096.0 : 0;           // blocks*waves*iters*4 MFMAs*4096 FLOP
        const double valu = (mode != 0) ? 256.0 * 4 * iters * 16.0 * 64 * 4 : 0;        // 16 pk_fma * 64 lanes * 4 FLOP
        printf("mode %d: %.3f ms  MFMA %.1f TF  VALU %.1f TF  total %.1f TF\n", mode, ms, mfma / ms / 1e9, valu / ms / 1e9,
               (mfma + valu) / ms / 1e9);
    }
    return 0;
}
