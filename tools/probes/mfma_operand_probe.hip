// Probe: does the issue rate of v_mfma_f32_16x16x32_bf16 depend on WHICH operand stays the same across back-to-back instructions?
// Pattern A: acc[j] = mfma(a, b[j])  (A fixed, B varies: the transposed products of fgcn_spatial_bwd_tile's contraction)
// Pattern B: acc[j] = mfma(a[j], b)  (A varies, B fixed: fgcn_pw.hip / fgcn_tconv.hip, a[mt] against one weight fragment)
// four accumulators in rotation, chains of six (the split-bf16 partial products), 2 waves per SIMD.
// Measured on MI355X (round 4): 2048 / 2117 TF/s of bf16 MFMAs with A fixed, 2045 / 2159 with B fixed -- NO asymmetry: the 5-25 % that
// pw_gemm lost with transposed accumulators (profiles/r04_ab_pw_transposed_epilogue.txt) is not the matrix pipe's doing.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_operand_probe.hip -o gpurun_out/mfma_operand && gpurun_out/mfma_operand
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int PAT>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fix[3], var[4][3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            fix[p][i] = (__bf16)(float)(threadIdx.x + i + p);
#pragma unroll
            for (int j = 0; j < 4; ++j) var[j][p][i] = (__bf16)(1.0f / (float)(threadIdx.x + i + j + p + 1));
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // six partial products of a three-way split pair (fgcn_common.hpp mfma_x3_k32)
                const int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    if (PAT == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fix[pa[q]], var[j][pb[q]], acc[j], 0, 0, 0);
                    else acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(var[j][pa[q]], fix[pb[q]], acc[j], 0, 0, 0);
                    asm volatile("" : "+v"(acc[j]));
                }
            }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int PAT>
void run(float* d) {
    const int blocks = 256, threads = 512, iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<PAT>, dim3(blocks), dim3(threads), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<PAT>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * iters * 48.0 * 16384.0;
    printf("%s: %.3f ms  %.1f TF/s of bf16 MFMAs\n", PAT == 0 ? "A fixed, B varies" : "A varies, B fixed", ms, flop / ms / 1e9);
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * sizeof(float));
    run<0>(d); run<1>(d); run<0>(d); run<1>(d);
    return 0;
}
