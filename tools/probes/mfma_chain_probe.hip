// Probe: issue rate of v_mfma_f32_16x16x32_bf16 when consecutive MFMAs accumulate into the same registers (chains of the six
// split-bf16 partial products) versus 2 / 4 / 8 accumulators in rotation; 1 or 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_chain_probe.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int NACC>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(float)(threadIdx.x + i);
        b[i] = (__bf16)(1.0f / (float)(threadIdx.x + i + 1));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 48 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
                asm volatile("" : "+v"(acc[i]));      // keep the source order
            }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NACC>
void run(float* d, int threads) {
    const int blocks = 256, iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(threads), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * iters * 48.0 * 16384.0;
    printf("accumulators in rotation %d, waves per SIMD %d: %.3f ms  %.1f TF/s\n", NACC, threads / 256, ms, flop / ms / 1e9);
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * sizeof(float));
    for (int threads : {256, 512}) {
        run<1>(d, threads); run<2>(d, threads); run<3>(d, threads); run<4>(d, threads); run<8>(d, threads);
    }
    return 0;
}
