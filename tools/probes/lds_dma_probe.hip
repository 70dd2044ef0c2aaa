// Probe: where does `buffer_load_dwordx4 ... lds` put each lane's 16 bytes, and what does an out-of-range lane write?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const float* in, float* out, unsigned bytes) {
    __shared__ __attribute__((aligned(16))) float sm[512];
    using lds_ptr = __attribute__((address_space(3))) void*;
    for (int i = threadIdx.x; i < 512; i += 64) sm[i] = -1.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, bytes, 0x00020000);
    const unsigned lane = threadIdx.x;
    // lanes 0..59 read 16 bytes at lane*16; lanes 60..63 are out of range
    const unsigned off = lane < 60 ? lane * 16u : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)sm, 16, off, 0, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = sm[i];
}

int main() {
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = (float)i;
    float *din, *dout;
    hipMalloc(&din, 1024);
    hipMalloc(&dout, 2048);
    hipMemcpy(din, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, din, dout, 1024u);
    std::vector<float> o(512);
    hipMemcpy(o.data(), dout, 2048, hipMemcpyDeviceToHost);
    for (int i = 0; i < 272; ++i) printf("%g%c", o[i], (i % 16 == 15) ? '\n' : ' ');
    printf("\n");
    return 0;
}
