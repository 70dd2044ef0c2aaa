// One-launch weight re-layout for a whole model (fgcn_pack_run).
//
// After every optimizer step the kernels' packed / split weight forms must be rebuilt from the parameters: ~25 small forms
// per block, ~250 per model -- as separate launches (torch cat / permute / contiguous + fgcn_pack_split3) that is ~170
// launches of a few microseconds each, all on the step's critical path (0.7 ms of an 11 ms step at 8 clips per GPU).
// Here a form is DATA: an item says which logical matrix W[tap][k][n] it is -- a sum over up to FGCN_PACK_MAX_SEG source
// segments, each a strided window of one parameter tensor (concatenation along k or n = disjoint segments, the summed conv_d
// bias = overlapping ones, channel padding = nothing covering that range) -- and which layout the consumer streams (plain,
// k-interleaved float4, the three-way bf16 split in fragment or accumulator order).  One launch walks a workgroup -> (item,
// first unit) map over ALL items; the item table lives in device memory and is rebuilt only when the set of forms changes.
// Reference semantics of the matrices themselves: torch_src/models/mmargcn/agcn.py:41-42,71-73,77 (the Conv2d weights).
#include "fgcn_common.hpp"

namespace fgcn {

__device__ __forceinline__ float pack_fetch(const fgcn_pack_item& it, int tap, int k, int n) {
    float v = 0.f;
    for (int s = 0; s < it.nseg; ++s) {
        const fgcn_pack_seg& g = it.seg[s];
        const int dt = tap - g.t0, dk = k - g.k0, dn = n - g.n0;
        if ((unsigned)dt < (unsigned)g.tlen && (unsigned)dk < (unsigned)g.klen && (unsigned)dn < (unsigned)g.nlen)
            v += g.src[(long long)(dt * g.tap_step + g.tap0) * g.st_tap + (long long)dk * g.st_k + (long long)dn * g.st_n];
    }
    return v;
}

// contraction index of element j of k-group kg: fragment order, or the order in which a 32x32 accumulator enumerates its rows
__device__ __forceinline__ int pack_k(int mode, int kg, int j) {
    return (mode == FGCN_PACK_SPLIT3_ACC || mode == FGCN_PACK_SPLIT2H_ACC) ? 16 * (kg >> 1) + 4 * (kg & 1) + (j & 3) + 8 * (j >> 2) : 8 * kg + j;
}

__global__ __launch_bounds__(256) void pack_run_kernel(const fgcn_pack_item* items, const int* blockmap) {
    const fgcn_pack_item& it = items[blockmap[2 * blockIdx.x]];
    const long long u = (long long)blockmap[2 * blockIdx.x + 1] * 256 + threadIdx.x;
    const int N = it.N;
    if (it.mode == FGCN_PACK_PLAIN) {
        if (u >= (long long)it.taps * it.K * N) return;
        const int n = (int)(u % N);
        const long long tk = u / N;
        reinterpret_cast<float*>(it.dst)[u] = pack_fetch(it, (int)(tk / it.K), (int)(tk % it.K), n);
        return;
    }
    const int KG = it.kgroups;
    if (u >= (long long)it.taps * KG * N) return;
    const int n = (int)(u % N);
    const long long tk = u / N;
    const int kg = (int)(tk % KG), tap = (int)(tk / KG);
    if (it.mode == FGCN_PACK_K4) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = 4 * kg + j < it.K ? pack_fetch(it, tap, 4 * kg + j, n) : 0.f;
        reinterpret_cast<f32x4*>(it.dst)[u] = v;
        return;
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = pack_k(it.mode, kg, j);
        v[j] = k < it.K ? pack_fetch(it, tap, k, n) : 0.f;
    }
    if (it.mode == FGCN_PACK_SPLIT2H || it.mode == FGCN_PACK_SPLIT2H_ACC) {
        // high / low f16 parts of W * 2^s, s from the form's maximum (header word 0, written by pack_amax_kernel)
        const float sc = exp2i(min(scale_exp_for(*reinterpret_cast<const unsigned*>(it.dst)), 126));
        u32x2 h0, l0, h1, l1;
        split2h_x4(f32x4{v[0], v[1], v[2], v[3]} * sc, h0, l0);
        split2h_x4(f32x4{v[4], v[5], v[6], v[7]} * sc, h1, l1);
        unsigned short* d2 = reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(it.dst) + 16);
        const long long plane2 = (long long)it.taps * KG * N * 8;
        *reinterpret_cast<u32x4v*>(d2 + u * 8) = u32x4v{h0[0], h0[1], h1[0], h1[1]};
        *reinterpret_cast<u32x4v*>(d2 + plane2 + u * 8) = u32x4v{l0[0], l0[1], l1[0], l1[1]};
        return;
    }
    u32x4v q[3];
    split3_x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], q);
    unsigned short* dst = reinterpret_cast<unsigned short*>(it.dst);
    const long long plane = (long long)it.taps * KG * N * 8;
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4v*>(dst + p * plane + u * 8) = q[p];
}

// FGCN_PACK_SPLIT2H, pass 1 and 2: header word 0 of every such form = 0, then = float bits of max |W| (non-negative floats order like
// their bit patterns, and an integer maximum does not depend on the order of the updates: reproducible)
__global__ void pack_zero_kernel(const fgcn_pack_item* items, int n_items) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_items && (items[i].mode == FGCN_PACK_SPLIT2H || items[i].mode == FGCN_PACK_SPLIT2H_ACC))
        *reinterpret_cast<unsigned*>(items[i].dst) = 0u;
}

__global__ __launch_bounds__(256) void pack_amax_kernel(const fgcn_pack_item* items, const int* blockmap) {
    const fgcn_pack_item& it = items[blockmap[2 * blockIdx.x]];
    if (it.mode != FGCN_PACK_SPLIT2H && it.mode != FGCN_PACK_SPLIT2H_ACC) return;      // (workgroup-uniform)
    const long long u = (long long)blockmap[2 * blockIdx.x + 1] * 256 + threadIdx.x;
    const int N = it.N, KG = it.kgroups;
    float m = 0.f;
    if (u < (long long)it.taps * KG * N) {
        const int n = (int)(u % N);
        const long long tk = u / N;
        const int kg = (int)(tk % KG), tap = (int)(tk / KG);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = pack_k(it.mode, kg, j);
            if (k < it.K) m = fmaxf(m, fabsf(pack_fetch(it, tap, k, n)));
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(it.dst), __builtin_bit_cast(unsigned, m));
}

}  // namespace fgcn

using namespace fgcn;

extern "C" int fgcn_pack_kgroups(int mode, int K) {
    switch (mode) {
        case FGCN_PACK_PLAIN: return K;
        case FGCN_PACK_K4: return (K + 3) / 4;
        case FGCN_PACK_SPLIT3: return (K + 7) / 8;
        case FGCN_PACK_SPLIT2H: return (K + 7) / 8;
        case FGCN_PACK_SPLIT2H_ACC: return (K + 15) / 16 * 2;
        case FGCN_PACK_SPLIT3_ACC: return (K + 15) / 16 * 2;
        default: return -1;
    }
}

extern "C" long long fgcn_pack_units(int mode, int taps, int K, int N) {
    const int kg = fgcn_pack_kgroups(mode, K);
    return kg < 0 ? -1 : (long long)taps * kg * N;
}

extern "C" int fgcn_pack_run(const fgcn_pack_item* items_dev, const int* blockmap_dev, int n_workgroups, void* stream) {
    FGCN_REQUIRE(items_dev && blockmap_dev && n_workgroups > 0, FGCN_E_BADARG, "pack_run: bad argument (workgroups=%d)",
                 n_workgroups);
    hipLaunchKernelGGL(pack_run_kernel, dim3((unsigned)n_workgroups), dim3(256), 0, (hipStream_t)stream, items_dev,
                       blockmap_dev);
    return launch_status("pack_run");
}

extern "C" int fgcn_pack_run_scaled(const fgcn_pack_item* items_dev, const int* blockmap_dev, int n_workgroups, int n_items,
                                    void* stream) {
    FGCN_REQUIRE(items_dev && blockmap_dev && n_workgroups > 0 && n_items > 0, FGCN_E_BADARG, "pack_run_scaled: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pack_zero_kernel, dim3((unsigned)cdiv(n_items, 256)), dim3(256), 0, s, items_dev, n_items);
    hipLaunchKernelGGL(pack_amax_kernel, dim3((unsigned)n_workgroups), dim3(256), 0, s, items_dev, blockmap_dev);
    hipLaunchKernelGGL(pack_run_kernel, dim3((unsigned)n_workgroups), dim3(256), 0, s, items_dev, blockmap_dev);
    return launch_status("pack_run_scaled");
}
