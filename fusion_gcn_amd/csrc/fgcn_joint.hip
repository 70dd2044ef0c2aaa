// Joint-mixing kernels: everything in the AGCN block that contracts over the skeleton's joints (V <= 32).
//   joint_mix  : out_t (V x ch) (+)= M (V x V) . in_t (V x ch)      per sample n and frame t   (MFMA 32x32x2 f32)
//   joint_gram : G (V x V) += in1_t (V x ch) . in2_t^T (ch x V)      summed over frames and channels
//   adj_softmax_{fwd,bwd}: the column softmax that turns the joint affinity into the data-dependent adjacency.
// The per-sample V x V matrices live in LDS (zero-padded to 32 x 32, row stride 33); one wave owns one frame
// at a time, the joint index sits on the MFMA row/K dimension and 32 channels on the lanes, so global loads and
// stores are 128-byte contiguous per half-wave.  These ops are HBM-bound (each activation element feeds one
// 32x32x2 step); the fused spatial kernel (fgcn_spatial.hip) removes them from the forward pass.
#include "fgcn_common.hpp"
// joint_dagg reads every operand tile exactly once: its tile loads carry the non-temporal hint (-6 % per launch in kbench, -0.03 .. -0.2 ms on the step;
// the same hint on joint_gram's and joint_mix_vec's loads, whose rows are read by several subsets / waves, cost 20-50 %: profiles/r03_ab_store_nt.txt)
#ifndef FGCN_DAGG_LDAUX
#define FGCN_DAGG_LDAUX 2
#endif

namespace fgcn {

constexpr int MS = 33;  // LDS row stride of a padded 32 x 32 joint matrix
constexpr int MIX_MAX_MATS = 3;

struct MixP {
    const float* in;
    float* out;
    const float* mats;
    int B, T, V, ld_in, ld_out, in_ch, out_ch, n_mats, mats_batched, n_items, accumulate, t_chunk;
    unsigned in_bytes;
    fgcn_mix_item items[FGCN_MIX_MAX_ITEMS];
};

__global__ __launch_bounds__(256) void joint_mix_kernel(MixP p) {
    __shared__ float mat[MIX_MAX_MATS * 32 * MS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int n = blockIdx.y;
    const int t0 = blockIdx.x * p.t_chunk;
    const int t1 = min(t0 + p.t_chunk, p.T);
    const int V = p.V;

    const float* msrc = p.mats + (p.mats_batched ? (long long)n * p.n_mats * V * V : 0);
    for (int i = tid; i < p.n_mats * 32 * 32; i += 256) {
        const int mi = i >> 10, u = (i >> 5) & 31, w = i & 31;
        mat[(mi * 32 + u) * MS + w] = (u < V && w < V) ? msrc[(mi * V + u) * V + w] : 0.f;
    }
    __syncthreads();

    const int ksteps = (V + 1) >> 1;
    // branch-free buffer loads (lanes / joints that do not take part read zeros through an out-of-range offset), issued
    // one (item, term) step ahead of the MFMA chain that consumes them
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    auto issue = [&](int t, int it, int tr, float (&bv)[16]) {
        const fgcn_mix_item& item = p.items[it < p.n_items ? it : 0];
        const fgcn_mix_term& term = item.term[tr];
        const int c_in = l31 < 16 ? term.in_c_lo + l31 : term.in_c_hi + (l31 - 16);
        const bool take = t < t1 && it < p.n_items && ((term.mask >> (l31 >> 4)) & 1) && c_in < p.in_ch && l31 < item.width;
        const unsigned base = (unsigned)((((long long)n * p.T + (t < t1 ? t : t0)) * V) * p.ld_in + c_in) * 4u;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int k = 2 * s + h;
            bv[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                  rin, (take && s < ksteps && k < V) ? base + (unsigned)(k * p.ld_in) * 4u : OOB, 0, 0));
        }
    };
    float bcur[16], bnxt[16];
    issue(t0 + wave, 0, 0, bcur);
    for (int t = t0 + wave; t < t1; t += 4) {
        const long long row0 = ((long long)n * p.T + t) * V;
        for (int it = 0; it < p.n_items; ++it) {
            const fgcn_mix_item& item = p.items[it];
            f32x16 acc = zero16();
            for (int tr = 0; tr < item.nterms; ++tr) {
                if (tr + 1 < item.nterms) issue(t, it, tr + 1, bnxt);
                else if (it + 1 < p.n_items) issue(t, it + 1, 0, bnxt);
                else issue(t + 4, 0, 0, bnxt);
                const fgcn_mix_term& term = item.term[tr];
                const float* mrow = &mat[term.mat * 32 * MS];
                // A[i = out joint][k = in joint] = M[i][k] (or M[k][i]); B[k = in joint][j = channel]
                const int a_i = term.transpose ? 1 : MS, a_k = term.transpose ? MS : 1;
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    if (s < ksteps) acc = mfma32(mrow[l31 * a_i + (2 * s + h) * a_k], bcur[s], acc);
#pragma unroll
                for (int s = 0; s < 16; ++s) bcur[s] = bnxt[s];
            }
            const int c_out = item.out_c + l31;
            if (c_out < p.out_ch && l31 < item.width) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int u = acc_row(r, lane);
                    if (u < V) {
                        float* dst = p.out + (row0 + u) * p.ld_out + c_out;
                        *dst = p.accumulate ? *dst + acc[r] : acc[r];
                    }
                }
            }
        }
    }
}

// Channel-group form (agg recompute, dx, embedding gradients): lane j of a group owns VW consecutive channels, one
// item covers 32*VW channels, loads and stores are whole 128/256-byte rows (VW = 4 measured no faster and needs 216 VGPRs).  Built for a short instruction
// stream and high occupancy rather than software pipelining (these mixes are HBM-bound; the first version spent its
// time issuing ~1000 VALU instructions per frame with load, MFMA and store phases serialised at 1-2 waves/SIMD):
//   * every load and store is a buffer instruction whose per-lane offset (joint row, lane channels, or the
//     out-of-range sentinel for joints >= V / absent channels) is computed once per kernel; the frame / channel-group
//     base goes in the scalar offset, so the inner loops carry no address arithmetic and no exec-mask branches;
//   * the A operands (M or M^T, zero-padded) sit in LDS as [k][i] images: one ds_read with an immediate offset per step;
//   * an item whose single term reads the same input as the previous item keeps that input in registers
//     (agg_0..2 share x); the accumulate form fetches the old output before the MFMA chain, not after it.
struct MixVP {
    const float* in;
    float* out;
    const float* mats;
    int B, T, V, ld_in, ld_out, n_mats, mats_batched, n_items, t_chunk;
    float* colsum;   // optional: per-workgroup column sums of everything this launch writes, [B * chunks][ld_out]
    unsigned* amax;  // optional: receives max |value written| (integer atomic maximum of the float bits: order-independent)
    unsigned in_bytes, out_bytes;
    struct Item {  // dword fields only: the kernel reads them with scalar loads (16-bit fields went through vector memory)
        int out_c, nterms, img[3], in_c[3];
    } items[FGCN_MIX_MAX_ITEMS];
    int nch;
};

template <int VW> struct MixVec;
template <> struct MixVec<1> {
    using raw = unsigned;
    static __device__ __forceinline__ raw load(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned so) { return __builtin_amdgcn_raw_buffer_load_b32(r, v, so, 0); }
    static __device__ __forceinline__ void store(raw d, __amdgpu_buffer_rsrc_t r, unsigned v, unsigned so) { __builtin_amdgcn_raw_buffer_store_b32(d, r, v, so, 0); }
};
template <> struct MixVec<2> {
    using raw = __attribute__((ext_vector_type(2))) unsigned;
    static __device__ __forceinline__ raw load(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned so) { return __builtin_amdgcn_raw_buffer_load_b64(r, v, so, 0); }
    static __device__ __forceinline__ void store(raw d, __amdgpu_buffer_rsrc_t r, unsigned v, unsigned so) { __builtin_amdgcn_raw_buffer_store_b64(d, r, v, so, 0); }
};
constexpr int IMG = 32 * 32;  // one A-operand image, [k][i]

// KS = MFMA k-steps (joint pairs) covered: ceil(V / 2) rounded up to even; the padding steps multiply zeros (the images
// are zero-padded and absent joints load as zeros).  Compile-time so that the MFMA chains carry no branches: with a
// runtime step count hipcc moved all accumulators between AGPRs and VGPRs around every conditional step.
// (Output stores stay plain at every size: streamed (fgcn_common.hpp, stream_out) this kernel won 19-31 % in a loop of identical launches -- its
// input then survives in the Infinity Cache from one repetition to the next -- and LOST 5-13 % per launch inside the step, same box:
// profiles/r03_ab_store_nt.txt section 7.)
template <int VW, bool ACC, int KS>
__global__ __launch_bounds__(256) void joint_mix_vec_kernel(MixVP p) {
    using vec = __attribute__((ext_vector_type(VW))) float;
    using raw = typename MixVec<VW>::raw;
    __shared__ float img[2 * MIX_MAX_MATS * IMG];
    extern __shared__ float cs[];                     // [4 waves][ld_out] column sums (only with p.colsum)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int n = blockIdx.y;
    const int t0 = blockIdx.x * p.t_chunk;
    const int t1 = min(t0 + p.t_chunk, p.T);
    const int V = p.V;
    if (p.colsum)
        for (int i = tid; i < 4 * p.ld_out; i += 256) cs[i] = 0.f;

    // image 2m + tr holds A[i = out joint][k = in joint] = tr ? M_m[k][i] : M_m[i][k] at [k][i]
    const float* msrc = p.mats + (p.mats_batched ? (long long)n * p.n_mats * V * V : 0);
    for (int e = tid; e < 2 * p.n_mats * IMG; e += 256) {
        const int im = e >> 10, k = (e >> 5) & 31, i = e & 31;
        const int r = (im & 1) ? k : i, c = (im & 1) ? i : k;
        img[e] = (r < V && c < V) ? msrc[((im >> 1) * V + r) * V + c] : 0.f;
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, p.out_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const bool lane_ok = VW * l31 < p.nch;  // all items of a launch have the same width (checked on the host)
    unsigned koff[16], uoff[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int k = 2 * s + h, u = acc_row(s, lane);
        koff[s] = (lane_ok && k < V) ? (unsigned)(k * p.ld_in + VW * l31) * 4u : OOB;
        uoff[s] = (lane_ok && u < V) ? (unsigned)(u * p.ld_out + VW * l31) * 4u : OOB;
    }
    const float* arow = &img[h * 32 + l31];
    float wmax = 0.f;                                 // (p.amax) largest magnitude this lane wrote; padding rows / lanes hold exact zeros

    for (int t = t0 + wave; t < t1; t += 4) {
        const unsigned frame = (unsigned)(n * p.T + t) * (unsigned)V;
        const unsigned fin = frame * (unsigned)p.ld_in * 4u, fout = frame * (unsigned)p.ld_out * 4u;
        raw bv[KS];
        int loaded_c = -1;
        for (int it = 0; it < p.n_items; ++it) {
            const MixVP::Item& item = p.items[it];
            const unsigned so_out = __builtin_amdgcn_readfirstlane(fout + (unsigned)item.out_c * 4u);
            raw old[ACC ? 16 : 1];
            if constexpr (ACC) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & 3) + 8 * (r >> 2) < 2 * KS) old[r] = MixVec<VW>::load(rout, uoff[r], so_out);
            }
            f32x16 acc[VW];
#pragma unroll
            for (int m = 0; m < VW; ++m) acc[m] = zero16();
            for (int tr = 0; tr < item.nterms; ++tr) {
                const int in_c = item.in_c[tr];
                if (!(item.nterms == 1 && in_c == loaded_c)) {
                    const unsigned so_in = __builtin_amdgcn_readfirstlane(fin + (unsigned)in_c * 4u);
#pragma unroll
                    for (int s = 0; s < KS; ++s) bv[s] = MixVec<VW>::load(rin, koff[s], so_in);
                }
                loaded_c = item.nterms == 1 ? in_c : -1;
                const float* a = arow + item.img[tr] * IMG;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const float av = a[s * 64];
                    const vec b = __builtin_bit_cast(vec, bv[s]);
#pragma unroll
                    for (int m = 0; m < VW; ++m) acc[m] = mfma32(av, b[m], acc[m]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if ((r & 3) + 8 * (r >> 2) >= 2 * KS) continue;  // compile-time: register r only holds padding joints
                vec v;
#pragma unroll
                for (int m = 0; m < VW; ++m) v[m] = acc[m][r];
                if constexpr (ACC) v += __builtin_bit_cast(vec, old[r]);
                MixVec<VW>::store(__builtin_bit_cast(raw, v), rout, uoff[r], so_out);
#pragma unroll
                for (int m = 0; m < VW; ++m) wmax = fmaxf(wmax, fabsf(v[m]));
            }
            if constexpr (!ACC) {
                if (p.colsum) {   // wave-uniform.  Rows >= V and absent channels are exact zeros (zero-padded images / loads)
#pragma unroll
                    for (int m = 0; m < VW; ++m) {
                        float sum = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) sum += acc[m][r];
                        sum += __shfl_xor(sum, 32);
                        if (h == 0 && lane_ok) cs[wave * p.ld_out + item.out_c + VW * l31 + m] += sum;   // wave-private row
                    }
                }
            }
        }
    }
    if (p.amax) {                                     // (kernel-uniform)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, d));
        if (lane == 0) atomicMax(p.amax, __builtin_bit_cast(unsigned, wmax));
    }
    if (p.colsum) {
        __syncthreads();
        float* dst = p.colsum + ((long long)n * gridDim.x + blockIdx.x) * p.ld_out;
        for (int c = tid; c < p.ld_out; c += 256)
            dst[c] = cs[c] + cs[p.ld_out + c] + cs[2 * p.ld_out + c] + cs[3 * p.ld_out + c];
    }
}

struct GramP {
    const float* in1;
    const float* in2;
    float* partial;
    int B, T, V, ld1, ld2, t_chunk, n_items, share1;
    unsigned in1_bytes, in2_bytes;
    struct Item {  // dword fields: read with scalar loads (the 16-bit ABI fields went through vector memory + vmcnt(0))
        int c1, c2, width, mat;
    } items[FGCN_GRAM_MAX_ITEMS];
};

// The block's affinity gram (three items = three subsets, theta_k against phi_k, equal widths 8 NQ = ic): item count and width as
// compile-time constants.  In the generic kernel below both are run-time values in wave-uniform guards around loads and MFMAs inside
// the frame loop; hipcc branched there and drained vmcnt(0) at the joins (tools/kres.py: 14 full drains) -- the pattern DESIGN.md
// section 3.8 of Appendix A found in the GEMM epilogues and section 3.9 in joint_dagg.  Here a frame is 6 NQ / 4 x 4 branch-free loads, then its MFMAs,
// and the NEXT frame's first item is requested before the current frame's last MFMAs.
template <int NQ>
__global__ __launch_bounds__(256, 3) void joint_gram3_kernel(GramP p) {
    __shared__ float red[4 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int t0 = chunk * p.t_chunk;
    const int t1 = min(t0 + p.t_chunk, p.T);
    const int V = p.V;
    const bool row_ok = l31 < V;
    const int vv = row_ok ? l31 : 0;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in1, 0, p.in1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in2, 0, p.in2_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    int ic1[3], ic2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ic1[i] = __builtin_amdgcn_readfirstlane(p.items[i].c1);
        ic2[i] = __builtin_amdgcn_readfirstlane(p.items[i].c2);
    }
    f32x16 acc[3] = {zero16(), zero16(), zero16()};
    // lane (joint l31, half h) holds channels 8 q + 4 h .. + 3 of its row, q = 0 .. NQ - 1
    auto loadq = [&](const __amdgpu_buffer_rsrc_t& r, int ld, int t, int c, f32x4 (&a)[NQ]) {
        const unsigned o = (row_ok && t < t1) ? ((unsigned)((n * p.T + t) * V + vv) * (unsigned)ld + c + 4 * h) * 4u : OOB;
#pragma unroll
        for (int q = 0; q < NQ; ++q) a[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, o, 32u * q, 0));
    };
    f32x4 ac[NQ], bc[NQ];
    loadq(r1, p.ld1, t0 + wave, ic1[0], ac);
    loadq(r2, p.ld2, t0 + wave, ic2[0], bc);
    for (int t = t0 + wave; t < t1; t += 4) {
#pragma unroll
        for (int it = 0; it < 3; ++it) {
            f32x4 an[NQ], bn[NQ];                                    // the next item's rows (the next frame's first item after the last)
            loadq(r1, p.ld1, it < 2 ? t : t + 4, ic1[it < 2 ? it + 1 : 0], an);
            loadq(r2, p.ld2, it < 2 ? t : t + 4, ic2[it < 2 ? it + 1 : 0], bn);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[it] = mfma32(ac[q][e], bc[q][e], acc[it]);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                ac[q] = an[q];
                bc[q] = bn[q];
            }
        }
    }
    const int nchunk = gridDim.x;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[it][r];
        __syncthreads();
        float* dst = p.partial + (((long long)n * nchunk + chunk) * 3 + it) * 1024;
        for (int e = tid; e < 1024; e += 256) {
            const float s = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
            const int r = e >> 6, l = e & 63;
            dst[acc_row(r, l) * 32 + (l & 31)] = s;
        }
    }
}

__global__ __launch_bounds__(256, 3) void joint_gram_kernel(GramP p) {
    __shared__ float red[4 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int t0 = chunk * p.t_chunk;
    const int t1 = min(t0 + p.t_chunk, p.T);
    const int V = p.V;
    const bool row_ok = l31 < V;
    const int vv = row_ok ? l31 : 0;

    // Branch-free buffer loads (absent joints / channels read as zeros through an out-of-range offset).  Widths are
    // multiples of 4 (host check).
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in1, 0, p.in1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in2, 0, p.in2_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    // item descriptors in scalar registers (dynamic indexing of the kernel-argument array went through vector memory
    // and a full vmcnt(0) drain in front of every prefetch)
    int ic1[3], ic2[3], iw[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const bool have = i < p.n_items;
        ic1[i] = __builtin_amdgcn_readfirstlane(have ? p.items[i].c1 : 0);
        ic2[i] = __builtin_amdgcn_readfirstlane(have ? p.items[i].c2 : 0);
        iw[i] = __builtin_amdgcn_readfirstlane(have ? p.items[i].width : 0);
    }
    // No software prefetch: the kernel is HBM-bound and 4-5 waves per SIMD (about 100 VGPRs) hide the load latency better
    // than a second set of 32 staging registers at 2 waves per SIMD did (3.3 TB/s).  When every item contracts the same
    // channels of in1 (dA^_k = x^T dagg_k: x is shared by the three subsets) its fragments are loaded once per group.
    f32x16 acc[3] = {zero16(), zero16(), zero16()};
    const bool share1 = p.share1 != 0;
    auto load1 = [&](int t, int it, int q0, f32x4 (&a)[4]) {
        const int c1 = it == 0 ? ic1[0] : (it == 1 ? ic1[1] : ic1[2]);
        const int width = it == 0 ? iw[0] : (it == 1 ? iw[1] : iw[2]);
        const unsigned o1 = ((unsigned)((n * p.T + t) * V + vv) * (unsigned)p.ld1 + c1 + 4 * h) * 4u;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            a[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                 r1, (row_ok && 8 * (q0 + j) + 4 * h < width) ? o1 + 32u * (q0 + j) : OOB, 0, 0));
    };
    auto load2 = [&](int t, int it, int q0, f32x4 (&b)[4]) {
        const int c2 = it == 0 ? ic2[0] : (it == 1 ? ic2[1] : ic2[2]);
        const int width = it == 0 ? iw[0] : (it == 1 ? iw[1] : iw[2]);
        const unsigned o2 = ((unsigned)((n * p.T + t) * V + vv) * (unsigned)p.ld2 + c2 + 4 * h) * 4u;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            b[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                 r2, (row_ok && 8 * (q0 + j) + 4 * h < width) ? o2 + 32u * (q0 + j) : OOB, 0, 0));
    };
    const int nq0 = (iw[0] + 7) >> 3;
    for (int t = t0 + wave; t < t1; t += 4) {
        if (share1) {
            for (int q0 = 0; q0 < nq0; q0 += 4) {
                f32x4 ac[4];
                load1(t, 0, q0, ac);
#pragma unroll
                for (int it = 0; it < 3; ++it) {
                    if (it < p.n_items) {
                        f32x4 bc[4];
                        load2(t, it, q0, bc);
#pragma unroll
                        for (int j = 0; j < 4; ++j)   // groups past the width multiply zeros
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[it] = mfma32(ac[j][e], bc[j][e], acc[it]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                if (it < p.n_items) {
                    const int nq = (iw[it] + 7) >> 3;
                    for (int q0 = 0; q0 < nq; q0 += 4) {
                        f32x4 ac[4], bc[4];
                        load1(t, it, q0, ac);
                        load2(t, it, q0, bc);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (q0 + j < nq) {   // wave-uniform (narrow embeddings: 16 channels = 2 groups)
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[it] = mfma32(ac[j][e], bc[j][e], acc[it]);
                            }
                        }
                    }
                }
            }
        }
    }
    // deterministic cross-wave sum, one matrix at a time
    const int nchunk = gridDim.x;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        if (it < p.n_items) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[it][r];
            __syncthreads();
            float* dst = p.partial + (((long long)n * nchunk + chunk) * p.n_items + p.items[it].mat) * 1024;
            for (int e = tid; e < 1024; e += 256) {
                const float s = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
                const int r = e >> 6, l = e & 63;
                dst[acc_row(r, l) * 32 + (l & 31)] = s;
            }
        }
    }
}

// ---- fused consumer of dagg: dx (+)= sum_k dagg_k . A^_k^T  AND  dA^_k = x^T . dagg_k in one pass ---------------------------
// The unfused pair (joint_mix_vec for dx, joint_gram for dA^) read the 3-activation-wide dagg twice.  Here a wave takes a
// frame, walks it in 32-channel chunks and parks the x chunk and each subset's dagg chunk once in wave-private LDS tiles
// ([32 joints][32 + 4], whole 128-byte lines per load instruction); the gram reads the tiles joint-per-lane
// (16-byte fragments = 4 k-steps), the mix reads the same dagg tile channel-per-lane.  No workgroup barrier in the loop.
struct DaggP {
    const float* x;
    const float* dagg;
    const float* mats;
    float* dx;
    float* partial;
    int B, T, V, C, ld_x, ld_d, ld_dx, n_sub, mats_batched, t_chunk, accumulate;
    unsigned x_bytes, d_bytes;
    // up to two gated addends of dx: dx += e[i] * [bit of m[i]] -- contiguous (B, T, V, C) tensors with fgcn_bn_act's sign image
    // (the ReLU-gated gradients that reach x through the block's identity shortcuts; agcn.py:114,135)
    const float* e[2];
    const unsigned char* m[2];
    int n_extra;
};

constexpr int DTS = 36;   // tile row stride: 16-byte reads of 8 consecutive rows hit 32 distinct banks

// MB: workgroups per CU the register allocation aims at (the gated form needs ~187 VGPRs: 2, key 6 bit 3 forces 3; the split-bf16
// gram runs 3 per CU with 48 bytes of scratch -- measured 3-4 % faster than 2 per CU without -- and key 6 bit 3 selects 2)
// X3: the gram dA^_k = x^T . dagg_k on the bf16 matrix pipe at f32 accuracy (math modes bf16x3 / bf16): the x chunk is split once
// per (frame, chunk) into its three bf16 parts in registers (lane = joint, 8 consecutive channels per fragment), each dagg chunk
// once per subset, and a 32-channel contraction is 2 x 6 v_mfma_f32_32x32x16_bf16 (384 cycles) instead of 16 f32 MFMAs (1024):
// the f32 form of this kernel runs at about half the rate its matrix work allows and the gram is more than half of that work.
// NSC / ACCM: the subset count and the accumulate flag as compile-time constants (0 / -1: run-time values).  Both sit in wave-uniform guards
// around loads inside the chunk loop; as run-time values they made hipcc branch there and, having lost count of the outstanding memory
// operations at every join, drain vmcnt(0) eight or nine times per chunk -- each one a full round trip of the chunk's dx stores
// (vmcnt counts loads and stores in issue order; DESIGN.md section 3.8 found the same in the GEMM epilogues).
template <int KS, int NE, int MB = 3, bool X3 = false, int NSC = 0, int ACCM = -1>
__global__ __launch_bounds__(256, MB) void joint_dagg_kernel(DaggP p) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    constexpr int NIMG = NE > 0 ? 4 : 3;               // + the identity (slot 3): gated addends ride the mix MFMAs
    float* img = dsm;                                  // [NIMG][k = in joint w][i = out joint v] = A^_k[v][w]
    float* tiles = dsm + NIMG * IMG;                   // [4 waves][2][32][DTS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int t0 = chunk * p.t_chunk;
    const int t1 = min(t0 + p.t_chunk, p.T);
    const int V = p.V, C = p.C, NS = NSC ? NSC : p.n_sub;
    const bool accumulate = ACCM < 0 ? p.accumulate != 0 : ACCM != 0;
    float* xt = tiles + wave * 2 * 32 * DTS;
    float* dt = xt + 32 * DTS;

    const float* msrc = p.mats + (p.mats_batched ? (long long)n * NS * V * V : 0);
    for (int e = tid; e < NIMG * IMG; e += 256) {
        const int k = e >> 10, w = (e >> 5) & 31, v = e & 31;
        if (k < 3) img[e] = (k < NS && v < V && w < V) ? msrc[(k * V + v) * V + w] : 0.f;
        else img[e] = (v == w && v < V) ? 1.f : 0.f;
    }
    __syncthreads();

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dagg, 0, p.d_bytes, 0x00020000);
    // dx rows of this workgroup's frames: offsets relative to frame t0 (no tensor-size limit on dx)
    const __amdgpu_buffer_rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.dx + ((long long)n * p.T + t0) * V * p.ld_dx), 0, (unsigned)((t1 - t0) * V * p.ld_dx) * 4u, 0x00020000);

    // gated addends: resources over this workgroup's frames (element index relative to frame t0; C % 8 == 0 keeps the sign
    // image byte-aligned at every frame)
    __amdgpu_buffer_rsrc_t re[NE > 0 ? NE : 1], rm[NE > 0 ? NE : 1];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const long long e0 = ((long long)n * p.T + t0) * V * C;
        re[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.e[i] + e0), 0, (unsigned)((t1 - t0) * V * C) * 4u, 0x00020000);
        rm[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.m[i] + (e0 >> 3)), 0, (unsigned)((t1 - t0) * V * C) >> 3, 0x00020000);
    }

    // staging: lane -> (row = lane / 8 + 8 * pass, 16-byte group lane % 8); all 32 rows are written (absent joints and
    // channels load zeros), so the tiles never hold stale data.  The global loads of a tile are issued one tile AHEAD (the x
    // chunk of the next (frame, channel chunk), the next subset's dagg chunk) and parked in registers while the MFMAs of the
    // current tile run; they are written to the wave-private LDS tile right before that tile's own MFMAs.
    const int srow = lane >> 3, sg = lane & 7;
    auto loadt = [&](const __amdgpu_buffer_rsrc_t& r, unsigned frow_bytes, int ld, int c, int cw, bool valid, f32x4 (&v)[4]) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = 8 * ps + srow;
            const bool ok = valid && row < V && 4 * sg < cw;
            v[ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  r, ok ? frow_bytes + (unsigned)(row * ld + c + 4 * sg) * 4u : OOB, 0, FGCN_DAGG_LDAUX));
        }
    };
    auto storet = [&](float* tile, const f32x4 (&v)[4]) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) *reinterpret_cast<f32x4*>(&tile[(8 * ps + srow) * DTS + 4 * sg]) = v[ps];
    };

    // gated addend tiles: the same staging (whole 128-byte lines), the sign-image nibble of each 16-byte group fetched beside
    // it and applied as the tile is written to LDS; the tile then goes through the mix MFMAs against the identity image
    auto loadg = [&](int i, int t, int c, int cw, bool valid, f32x4 (&v)[4], unsigned (&mb)[4]) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = 8 * ps + srow;
            const bool ok = valid && row < V && 4 * sg < cw;
            const unsigned el = (unsigned)(((t - t0) * V + row) * C + c + 4 * sg);
            v[ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(re[i], ok ? el * 4u : OOB, 0, 0));
            mb[ps] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rm[i], ok ? el >> 3 : OOB, 0, 0) >> (el & 4u);
        }
    };
    auto storeg = [&](float* tile, const f32x4 (&v)[4], const unsigned (&mb)[4]) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            f32x4 g = v[ps];
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = (mb[ps] >> e) & 1u ? g[e] : 0.f;
            *reinterpret_cast<f32x4*>(&tile[(8 * ps + srow) * DTS + 4 * sg]) = g;
        }
    };

    f32x16 accg[3] = {zero16(), zero16(), zero16()};
    const float* xa = xt + l31 * DTS + 4 * h;          // gram fragments: lane = joint
    const float* db = dt + l31 * DTS + 4 * h;
    const float* dm = dt + h * DTS + l31;              // mix B operand: lane = channel, joints 2s + h
    const float* am = img + h * 32 + l31;              // mix A operand: image row k = w = 2s + h, lane = out joint v
    const int u0 = 4 * h;                              // accumulator register r holds out joint (r&3) + 8(r>>2) + 4h
    auto frame_bytes = [&](int t, int ld) { return (unsigned)((n * p.T + t) * V) * (unsigned)ld * 4u; };
    // PF3: one register set per subset, each refilled with the NEXT chunk's tile of its subset as soon as it has been written to LDS:
    // every load of chunk i + 1 is requested during chunk i, in front of chunk i's dx stores, and dx's old values at the top of the
    // chunk -- so no wait of the chunk loop ever includes a store's write acknowledgement (two workgroups per CU: 48 registers more).
    constexpr bool PF3 = NSC == 3 && NE == 0 && MB == 2;
    f32x4 vx[4], vd[4], vd3[PF3 ? 3 : 1][4];
    {
        const int t = t0 + wave;
        loadt(rx, frame_bytes(t, p.ld_x), p.ld_x, 0, min(32, C), t < t1, vx);
        if constexpr (PF3) {
#pragma unroll
            for (int k = 0; k < 3; ++k) loadt(rd, frame_bytes(t, p.ld_d), p.ld_d, k * C, min(32, C), t < t1, vd3[k]);
        } else {
            loadt(rd, frame_bytes(t, p.ld_d), p.ld_d, 0, min(32, C), t < t1, vd);
        }
    }
    for (int t = t0 + wave; t < t1; t += 4) {
        const unsigned fd = frame_bytes(t, p.ld_d);
        for (int c0 = 0; c0 < C; c0 += 32) {
            const int cw = min(32, C - c0);
            // the (frame, chunk) after this one
            const bool same_t = c0 + 32 < C;
            const int tn = same_t ? t : t + 4, cn = same_t ? c0 + 32 : 0;
            const int cwn = min(32, C - cn);
            const bool nvalid = tn < t1;
            storet(xt, vx);
            if constexpr (PF3) loadt(rx, frame_bytes(nvalid ? tn : t, p.ld_x), p.ld_x, nvalid ? cn : c0, nvalid ? cwn : cw, true, vx);
            else loadt(rx, frame_bytes(nvalid ? tn : t, p.ld_x), p.ld_x, cn, cwn, nvalid, vx);
            // dx chunk: rows v in the registers, 32 channels on the lanes
            const int c = c0 + l31;
            const unsigned coff = c < C ? (unsigned)(((t - t0) * V + u0) * p.ld_dx + c) * 4u : OOB;
            const unsigned rstep = (unsigned)p.ld_dx * 4u;
            float old[16];
            auto load_old = [&]() {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    old[r] = 0.f;
                    if (dr < 2 * KS)                   // compile-time: register r only holds padding joints otherwise
                        old[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                               rdx, (coff != OOB && u0 + dr < V) ? coff + dr * rstep : OOB, 0, 0));
                }
            };
            if constexpr (PF3 && ACCM == 1) load_old();
            if constexpr (PF3) __builtin_amdgcn_sched_barrier(0);
            f32x16 accx = zero16();
            unsigned gm[4] = {0u, 0u, 0u, 0u};
            u32x4v xs3[2][3];                           // X3: the x chunk's gram fragments (channels 16s + 8h + j of joint l31)
            if constexpr (X3) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(xt + l31 * DTS + 16 * s2 + 8 * h);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(xt + l31 * DTS + 16 * s2 + 8 * h + 4);
                    split3_x8(a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3], xs3[s2]);
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k < NS) {                          // wave-uniform
                    if constexpr (PF3) {
                        // (past the workgroup's last chunk the current tile is requested again, unused: a wave-uniform "no next
                        // chunk" condition on the request would come back as a branch around the loads and cost the wait counts)
                        storet(dt, vd3[k]);
                        loadt(rd, frame_bytes(nvalid ? tn : t, p.ld_d), p.ld_d, k * C + (nvalid ? cn : c0), nvalid ? cwn : cw, true, vd3[k]);
                        __builtin_amdgcn_sched_barrier(0);   // (the scheduler would sink the requests to the end of the chunk, next to the stores)
                    } else {
                        storet(dt, vd);
                        if (k + 1 < NS) loadt(rd, fd, p.ld_d, (k + 1) * C + c0, cw, true, vd);
                        else if (NE > 0) loadg(0, t, c0, cw, true, vd, gm);
                        else loadt(rd, frame_bytes(nvalid ? tn : t, p.ld_d), p.ld_d, cn, cwn, nvalid, vd);
                    }
                    if constexpr (X3) {                // dA^_k += x chunk . dagg_k chunk^T, split-bf16 (channels 16s + 8h + j)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            const f32x4 b0 = *reinterpret_cast<const f32x4*>(dt + l31 * DTS + 16 * s2 + 8 * h);
                            const f32x4 b1 = *reinterpret_cast<const f32x4*>(dt + l31 * DTS + 16 * s2 + 8 * h + 4);
                            u32x4v ds3[3];
                            split3_x8(b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3], ds3);
                            accg[k] = mfma_x3_k16(xs3[s2], ds3, accg[k]);
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {  // dA^_k += x chunk . dagg_k chunk^T (channels 8q + 4h + e)
                            const f32x4 av = *reinterpret_cast<const f32x4*>(xa + 8 * q);
                            const f32x4 bv = *reinterpret_cast<const f32x4*>(db + 8 * q);
#pragma unroll
                            for (int e = 0; e < 4; ++e) accg[k] = mfma32(av[e], bv[e], accg[k]);
                        }
                    }
#pragma unroll
                    for (int s = 0; s < KS; ++s)       // dx chunk += A^_k . dagg_k chunk (joints 2s + h)
                        accx = mfma32(am[k * IMG + s * 64], dm[2 * s * DTS], accx);
                }
            }
#pragma unroll
            for (int i = 0; i < NE; ++i) {             // dx chunk += I . (gated addend chunk)
                storeg(dt, vd, gm);
                if (i + 1 < NE) loadg(i + 1, t, c0, cw, true, vd, gm);
                else loadt(rd, frame_bytes(nvalid ? tn : t, p.ld_d), p.ld_d, cn, cwn, nvalid, vd);
#pragma unroll
                for (int s = 0; s < KS; ++s) accx = mfma32(am[3 * IMG + s * 64], dm[2 * s * DTS], accx);
            }
            if constexpr (!(PF3 && ACCM == 1)) {
                if (accumulate) {
                    load_old();
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) old[r] = 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                if (dr >= 2 * KS) continue;            // compile-time: register r only holds padding joints
                const float val = accx[r] + old[r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rdx,
                                                      (coff != OOB && u0 + dr < V) ? coff + dr * rstep : OOB, 0, 0);
            }
        }
    }
    // deterministic cross-wave sum of the gram accumulators (the tiles are free now: reuse them as [4 waves][1024])
    const int nchunk = gridDim.x;
    float* red = tiles;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < NS) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = accg[k][r];
            __syncthreads();
            float* dst = p.partial + (((long long)n * nchunk + chunk) * NS + k) * 1024;
            for (int e = tid; e < 1024; e += 256) {
                const float sum = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
                const int r = e >> 6, l = e & 63;
                dst[acc_row(r, l) * 32 + (l & 31)] = sum;
            }
        }
    }
}

// ---- fused agg recompute + conv_d weight gradient: dWd_k[c][o] = sum_{n,t,v} (x . A^_k)[(n,t,v), c] * dy[(n,t,v), o] ---------
// The unfused pair wrote agg = x . A^ (3 activations wide) with joint_mix_vec and read it back in the weight-gradient GEMM.
// Here a workgroup owns a (32 in-channel x 64 out-channel) tile of all three subsets for the frames [t0, t1) of one
// sample; a wave takes a frame: the x chunk and the dy chunk land in wave-private LDS tiles, agg_k (joints x 32 channels)
// is formed in the accumulator registers (KS MFMAs) and those registers are the A operand of the weight-gradient MFMAs
// as they stand: register r of lane (c, h) holds joint (r&3) + 8(r>>2) + 4h, which is exactly a 2-deep contraction step
// whose B operand is dy[joint][o].  Deterministic partial slabs [n * chunks + chunk][3 * Cin][Cout], summed by the caller.
struct SWgradP {
    const float* x;
    const float* dy;
    const float* mats;
    float* partial;
    int B, T, V, Cin, Cout, ld_x, ld_dy, n_sub, mats_batched, t_chunk, tiles_o;
    unsigned x_bytes, dy_bytes, p_bytes;
};

constexpr int YTS = 68;   // dy tile row stride (64 channels + 4)
constexpr int SW_AHB = 80; // MM == 2: bytes per [w] row of a split A^ plane (32 joints x bf16 + 16 pad: conflict-free b128 reads)

// MM: 0 = exact f32 MFMAs; 1 = FGCN_MATH_BF16 (operands rounded once, one bf16 MFMA per product group); 2 = FGCN_MATH_BF16X3: the
// weight-gradient contraction from exact three-way bf16 splits (six v_mfma_f32_32x32x16_bf16 per 16 joints, 384 cycles instead of
// 13 x 2 x 64 on the f32 pipe): the aggregation's accumulator registers 8 gp .. 8 gp + 7 of lane half h ARE a 16-deep k fragment
// (joints (j & 3) + 8 (j >> 2) + 4 h + 16 gp), split in registers; the dy fragments in the same joint order are split once per frame and
// serve the three subsets.  The joint mixing itself stays on the f32 MFMA (13 of the frame's 91 f32-equivalent steps).
template <int KS, int MM>
__global__ __launch_bounds__(256, 2) void spatial_wgrad_kernel(SWgradP p) {
    constexpr bool BF = MM == 1;
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float* img = wsm;                                  // [3][kk = in joint v][i = out joint w] = A^_k[v][w]
    // MM == 2: instead, the three bf16 parts of A^_k as planes [subset][part][w][SW_AHB bytes] (v contiguous: one ds_read_b128 = the 8
    // joints of a lane's fragment of the split joint mixing)
    constexpr int IMG_FLOATS = MM == 2 ? 9 * 32 * SW_AHB / 4 : 3 * IMG;
    unsigned char* ahs = reinterpret_cast<unsigned char*>(wsm);
    float* tiles = wsm + IMG_FLOATS;                   // [4 waves][32 * DTS + 32 * YTS]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x, chunk = blockIdx.y, n = blockIdx.z;
    const int tc = tile / p.tiles_o, to = tile - tc * p.tiles_o;
    const int c0 = tc * 32, o0 = to * 64;
    const int t0 = chunk * p.t_chunk;
    const int t1 = min(t0 + p.t_chunk, p.T);
    const int V = p.V, NS = p.n_sub;
    float* xt = tiles + wave * (32 * DTS + 32 * YTS);
    float* yt = xt + 32 * DTS;

    const float* msrc = p.mats + (p.mats_batched ? (long long)n * NS * V * V : 0);
    for (int e = tid; e < 3 * IMG; e += 256) {
        if constexpr (MM == 2) {
            const int k = e >> 10, w = (e >> 5) & 31, v = e & 31;
            const float a = (k < NS && v < V && w < V) ? msrc[(k * V + v) * V + w] : 0.f;
            unsigned ph, pm, pl;
            split_bf16_pair(a, 0.f, ph, pm, pl);
            unsigned short* d = reinterpret_cast<unsigned short*>(ahs + ((k * 3) * 32 + w) * SW_AHB) + v;
            d[0] = (unsigned short)ph;
            d[32 * SW_AHB / 2] = (unsigned short)pm;
            d[2 * 32 * SW_AHB / 2] = (unsigned short)pl;
        } else {
            const int k = e >> 10, v = (e >> 5) & 31, w = e & 31;
            img[e] = (k < NS && v < V && w < V) ? msrc[(k * V + v) * V + w] : 0.f;
        }
    }
    __syncthreads();

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    const int cw = min(32, p.Cin - c0), ow = min(64, p.Cout - o0);

    f32x16 accw[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k) accw[k][0] = accw[k][1] = zero16();
    const float* xb = xt + h * DTS + l31;              // mix B operand: joint 2s + h, lane = channel
    const float* am = img + h * 32 + l31;              // mix A operand: image row kk = v = 2s + h, lane = out joint
    const float* yb = yt + 4 * h * YTS + l31;          // weight-gradient B operand: joint (r&3) + 8(r>>2) + 4h, lane = o
    for (int t = t0 + wave; t < t1; t += 4) {
        const unsigned row0 = (unsigned)((n * p.T + t) * V);
        {   // x chunk: lane -> (row lane/8 + 8 pass, 16-byte group lane%8); dy chunk: (row lane/16 + 4 pass, group lane%16)
            const int xr = lane >> 3, xg = lane & 7, yr = lane >> 4, yg = lane & 15;
            f32x4 xv[4], yv[8];
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = 8 * ps + xr;
                xv[ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    rx, (row < V && 4 * xg < cw) ? ((row0 + row) * (unsigned)p.ld_x + c0 + 4 * xg) * 4u : OOB, 0, 0));
            }
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
                const int row = 4 * ps + yr;
                yv[ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    ry, (row < V && 4 * yg < ow) ? ((row0 + row) * (unsigned)p.ld_dy + o0 + 4 * yg) * 4u : OOB, 0, 0));
            }
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) *reinterpret_cast<f32x4*>(&xt[(8 * ps + xr) * DTS + 4 * xg]) = xv[ps];
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) *reinterpret_cast<f32x4*>(&yt[(4 * ps + yr) * YTS + 4 * yg]) = yv[ps];
        }
        u32x4v yb3[MM == 2 ? 2 : 1][2][3];                 // MM == 2: dy fragments of this frame, [16-joint group][32-column tile][part]
        u32x4v xf3[MM == 2 ? 2 : 1][3];                    // MM == 2: x fragments of the joint mixing (lane = channel, joints 16 s2 + 8 h + j)
        if constexpr (MM == 2) {
            const float* xq = xt + 8 * h * DTS + l31;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                split3_x8(xq[(16 * s2) * DTS], xq[(16 * s2 + 1) * DTS], xq[(16 * s2 + 2) * DTS], xq[(16 * s2 + 3) * DTS], xq[(16 * s2 + 4) * DTS],
                          xq[(16 * s2 + 5) * DTS], xq[(16 * s2 + 6) * DTS], xq[(16 * s2 + 7) * DTS], xf3[s2]);
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                if (16 * gp >= 2 * KS) continue;
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    const float* yq = yb + 16 * gp * YTS + ot * 32;
                    split3_x8(yq[0], yq[YTS], yq[2 * YTS], yq[3 * YTS], yq[8 * YTS], yq[9 * YTS], yq[10 * YTS], yq[11 * YTS], yb3[gp][ot]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k < NS) {                              // wave-uniform
                f32x16 agg = zero16();                 // agg_k chunk: rows = out joint, lanes = channel
                if constexpr (MM == 2) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        u32x4v af[3];
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            af[pl] = *reinterpret_cast<const u32x4v*>(ahs + ((k * 3 + pl) * 32 + l31) * SW_AHB + 16 * h + 32 * s2);
                        agg = mfma_x3_k16(af, xf3[s2], agg);
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < KS; ++s) agg = mfma32(am[k * IMG + s * 64], xb[2 * s * DTS], agg);
                }
                if constexpr (MM == 2) {
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        if (16 * gp >= 2 * KS) continue;
                        u32x4v a3[3];
                        split3_x8(agg[8 * gp], agg[8 * gp + 1], agg[8 * gp + 2], agg[8 * gp + 3], agg[8 * gp + 4], agg[8 * gp + 5],
                                  agg[8 * gp + 6], agg[8 * gp + 7], a3);
#pragma unroll
                        for (int ot = 0; ot < 2; ++ot) accw[k][ot] = mfma_x3_k16(a3, yb3[gp][ot], accw[k][ot]);
                    }
                } else if constexpr (BF) {
                    // 8 joints per bf16 MFMA: registers 4g..4g+3 of lane half h are joints 8g + 4h + (0..3)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (8 * g >= 2 * KS) continue;
                        const s16x4 ap = pack_bf16(agg[4 * g], agg[4 * g + 1], agg[4 * g + 2], agg[4 * g + 3]);
                        const float* yrow = yb + 8 * g * YTS;
#pragma unroll
                        for (int ot = 0; ot < 2; ++ot)
                            accw[k][ot] = mfma_bf16(ap, pack_bf16(yrow[ot * 32], yrow[YTS + ot * 32], yrow[2 * YTS + ot * 32],
                                                                  yrow[3 * YTS + ot * 32]), accw[k][ot]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        if (dr >= 2 * KS) continue;    // compile-time: register r only holds padding joints (exact zeros)
#pragma unroll
                        for (int ot = 0; ot < 2; ++ot) accw[k][ot] = mfma32(agg[r], yb[dr * YTS + ot * 32], accw[k][ot]);
                    }
                }
            }
        }
    }
    // cross-wave sum (fixed order) and the slab: partial[slab][k * Cin + c0 + c][o0 + o]
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)p.partial, 0, p.p_bytes, 0x00020000);
    const unsigned slab = (unsigned)(n * gridDim.y + chunk);
    float* red = tiles;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k >= NS) continue;
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = accw[k][ot][r];
            __syncthreads();
            for (int e = tid; e < 1024; e += 256) {
                const float sum = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
                const int r = e >> 6, l = e & 63;
                const int c = acc_row(r, l), o = ot * 32 + (l & 31);
                const unsigned off = (c < cw && o < ow)
                                         ? ((slab * (unsigned)(NS * p.Cin) + (unsigned)(k * p.Cin + c0 + c)) * (unsigned)p.Cout + o0 + o) * 4u
                                         : OOB;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum), rp, off, 0, 0);
            }
        }
    }
}

// sum of the nchunk per-chunk partial matrices at this thread's element: eight loads in flight, four running sums (the
// 75-chunk walk of an 8-clip shard was a 20 us chain of dependent loads), fixed order -> bitwise reproducible
__device__ __forceinline__ float sum_chunk_partials(const float* src, int nchunk, long long stride) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = 0;
    for (; c + 8 <= nchunk; c += 8) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = src[(long long)(c + u) * stride];
        s0 += x[0] + x[4];
        s1 += x[1] + x[5];
        s2 += x[2] + x[6];
        s3 += x[3] + x[7];
    }
    for (; c < nchunk; ++c) s0 += src[(long long)c * stride];
    return (s0 + s1) + (s2 + s3);
}

// One 32 x 32 thread block per (sample, subset) matrix: thread (v, w) sums its chunk partials (consecutive threads read
// consecutive addresses), the matrix goes through LDS and every thread reduces its own column (dim -2 of the (V, V)
// affinity).  adj_ab may be given as two addends (adj_a, adj_b) so the caller needs no separate add kernel.
__global__ __launch_bounds__(1024) void adj_softmax_fwd_kernel(const float* partial, int nchunk, float scale,
                                                               const float* adj_a, const float* adj_b, float* c_out,
                                                               float* a_hat, int K, int V, int use_softmax) {
    __shared__ float S[32][33];
    const int v = threadIdx.x >> 5, w = threadIdx.x & 31;
    const int n = blockIdx.x / K, k = blockIdx.x - n * K;
    const bool in = v < V && w < V;
    const long long o = ((long long)(n * K + k) * V + v) * V + w;
    const float ab = in ? adj_a[((long long)k * V + v) * V + w] + (adj_b ? adj_b[((long long)k * V + v) * V + w] : 0.f) : 0.f;
    if (!use_softmax) {
        if (in) a_hat[o] = ab;
        return;
    }
    const float* src = partial + ((long long)n * nchunk * K + k) * 1024 + threadIdx.x;
    float s = sum_chunk_partials(src, nchunk, (long long)K * 1024);
    s *= scale;
    S[v][w] = in ? s : -INFINITY;
    __syncthreads();
    float mx = -INFINITY;
    for (int u = 0; u < V; ++u) mx = fmaxf(mx, S[u][w]);
    float den = 0.f;
    for (int u = 0; u < V; ++u) den += expf(S[u][w] - mx);
    if (in) {
        const float c = expf(s - mx) / den;
        c_out[o] = c;
        a_hat[o] = c + ab;
    }
}

__global__ __launch_bounds__(1024) void adj_softmax_bwd_kernel(const float* partial, int nchunk, float scale,
                                                               const float* c_in, float* d_a_hat, float* d_s, int K,
                                                               int V) {
    __shared__ float P[32][33];
    const int v = threadIdx.x >> 5, w = threadIdx.x & 31;
    const int n = blockIdx.x / K, k = blockIdx.x - n * K;
    const bool in = v < V && w < V;
    const long long o = ((long long)(n * K + k) * V + v) * V + w;
    const float* src = partial + ((long long)n * nchunk * K + k) * 1024 + threadIdx.x;
    const float dc = sum_chunk_partials(src, nchunk, (long long)K * 1024);
    if (in) d_a_hat[o] = dc;
    if (!c_in || !d_s) return;
    const float cv = in ? c_in[o] : 0.f;
    P[v][w] = cv * dc;
    __syncthreads();
    float dot = 0.f;
    for (int u = 0; u < V; ++u) dot += P[u][w];
    if (in) d_s[o] = scale * cv * (dc - dot);
}

}  // namespace fgcn

using namespace fgcn;

extern "C" int fgcn_joint_mix_chunks(int B, int T);
static int pick_t_chunk(int B, int T) {
    // enough workgroups to fill 256 CUs a few times over, at least 4 frames (one per wave) per workgroup
    int chunk = 32;
    while (chunk > 4 && (long long)B * cdiv(T, chunk) < 1024) chunk >>= 1;
    return chunk;
}

extern "C" int fgcn_joint_mix(const float* in, float* out, const float* mats, int B, int T, int V,
                              int ld_in, int ld_out, int in_channels, int out_channels, int n_mats, int mats_batched,
                              const fgcn_mix_item* items, int n_items, int accumulate, void* stream) {
    FGCN_REQUIRE(in && out && mats && items, FGCN_E_BADARG, "joint_mix: null pointer");
    FGCN_REQUIRE(B > 0 && T > 0 && V > 0 && V <= FGCN_MAX_V, FGCN_E_BADARG, "joint_mix: bad B/T/V (%d,%d,%d)", B, T, V);
    FGCN_REQUIRE(n_mats >= 1 && n_mats <= MIX_MAX_MATS && n_items >= 1 && n_items <= FGCN_MIX_MAX_ITEMS,
                 FGCN_E_BADARG, "joint_mix: n_mats=%d n_items=%d out of range", n_mats, n_items);
    FGCN_REQUIRE(in_channels > 0 && out_channels > 0 && ld_in >= in_channels && ld_out >= out_channels, FGCN_E_BADARG,
                 "joint_mix: channel counts exceed row strides");
    FGCN_REQUIRE(B <= 65535, FGCN_E_BADARG, "joint_mix: B too large for grid.y");
    const long long mix_in_bytes = (long long)B * T * V * ld_in * 4;
    FGCN_REQUIRE(mix_in_bytes < 0x7FFF0000ll, FGCN_E_BADARG, "joint_mix: input must be smaller than 2 GiB");
    MixP p;
    p.in_bytes = (unsigned)mix_in_bytes;
    p.in = in; p.out = out; p.mats = mats;
    p.B = B; p.T = T; p.V = V; p.ld_in = ld_in; p.ld_out = ld_out; p.in_ch = in_channels; p.out_ch = out_channels;
    p.n_mats = n_mats; p.mats_batched = mats_batched; p.n_items = n_items; p.accumulate = accumulate;
    p.t_chunk = pick_t_chunk(B, T);
    for (int i = 0; i < n_items; ++i) {
        const fgcn_mix_item& it = items[i];
        FGCN_REQUIRE(it.nterms >= 1 && it.nterms <= 3 && it.out_c >= 0 && it.width >= 1 && it.width <= 32, FGCN_E_BADARG,
                     "joint_mix: item %d malformed", i);
        for (int t = 0; t < it.nterms; ++t)
            FGCN_REQUIRE(it.term[t].mat >= 0 && it.term[t].mat < n_mats && it.term[t].in_c_lo >= 0 &&
                             it.term[t].in_c_hi >= 0,
                         FGCN_E_BADARG, "joint_mix: item %d term %d malformed", i, t);
        p.items[i] = it;
    }
    dim3 grid((unsigned)cdiv(T, p.t_chunk), (unsigned)B);
    hipLaunchKernelGGL(joint_mix_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return launch_status("joint_mix");
}

extern "C" int fgcn_joint_gram(const float* in1, const float* in2, float* partial, int B, int T, int V,
                               int ld1, int ld2, int t_chunk, int n_mats, const fgcn_gram_item* items, int n_items,
                               void* stream) {
    FGCN_REQUIRE(in1 && in2 && partial && items, FGCN_E_BADARG, "joint_gram: null pointer");
    FGCN_REQUIRE(B > 0 && T > 0 && V > 0 && V <= FGCN_MAX_V && t_chunk > 0, FGCN_E_BADARG, "joint_gram: bad sizes");
    FGCN_REQUIRE(n_items >= 1 && n_items <= 3 && n_mats == n_items, FGCN_E_BADARG,
                 "joint_gram: needs 1..3 items, one per output matrix (n_items=%d n_mats=%d)", n_items, n_mats);
    FGCN_REQUIRE(ld1 % 4 == 0 && ld2 % 4 == 0 && aligned16(in1) && aligned16(in2), FGCN_E_ALIGN,
                 "joint_gram: strides/pointers must be 16-byte aligned");
    FGCN_REQUIRE(B <= 65535, FGCN_E_BADARG, "joint_gram: B too large for grid.y");
    const long long b1 = (long long)B * T * V * ld1 * 4, b2 = (long long)B * T * V * ld2 * 4;
    FGCN_REQUIRE(b1 < 0x7FFF0000ll && b2 < 0x7FFF0000ll, FGCN_E_BADARG, "joint_gram: operands must be smaller than 2 GiB");
    GramP p;
    p.in1_bytes = (unsigned)b1; p.in2_bytes = (unsigned)b2;
    p.in1 = in1; p.in2 = in2; p.partial = partial;
    p.B = B; p.T = T; p.V = V; p.ld1 = ld1; p.ld2 = ld2; p.t_chunk = t_chunk; p.n_items = n_items;
    for (int i = 0; i < n_items; ++i) {
        FGCN_REQUIRE(items[i].mat == i && items[i].width > 0 && items[i].width % 4 == 0 && items[i].c1 % 4 == 0 &&
                         items[i].c2 % 4 == 0 &&
                         items[i].c1 + items[i].width <= ld1 + 3 && items[i].c2 + items[i].width <= ld2 + 3,
                     FGCN_E_BADARG, "joint_gram: item %d malformed", i);
        p.items[i].c1 = items[i].c1; p.items[i].c2 = items[i].c2;
        p.items[i].width = items[i].width; p.items[i].mat = items[i].mat;
    }
    p.share1 = 1;
    for (int i = 1; i < n_items; ++i)
        if (items[i].c1 != items[0].c1 || items[i].width != items[0].width) p.share1 = 0;
    dim3 grid((unsigned)cdiv(T, t_chunk), (unsigned)B);
    // three items of one width 16 / 32 / 64 that do not share their first operand (the affinity gram theta_k^T phi_k): the specialised kernel;
    // tuning key 12 = 1 keeps the generic one (A/B)
    const int w0 = items[0].width;
    const bool three = n_items == 3 && !p.share1 && items[1].width == w0 && items[2].width == w0 && fgcn::tuning(12) != 1;
    if (three && w0 == 16) hipLaunchKernelGGL(joint_gram3_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (three && w0 == 32) hipLaunchKernelGGL(joint_gram3_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (three && w0 == 64) hipLaunchKernelGGL(joint_gram3_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(joint_gram_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return launch_status("joint_gram");
}

extern "C" int fgcn_adj_softmax_fwd(const float* partial, int nchunk, float scale, const float* adj_a,
                                    const float* adj_b, float* c_out, float* a_hat, int B, int K, int V,
                                    int use_softmax, void* stream) {
    FGCN_REQUIRE(adj_a && a_hat && B > 0 && K > 0 && V > 0 && V <= FGCN_MAX_V, FGCN_E_BADARG,
                 "adj_softmax_fwd: bad argument");
    FGCN_REQUIRE(!use_softmax || (partial && c_out && nchunk > 0), FGCN_E_BADARG, "adj_softmax_fwd: missing partials");
    hipLaunchKernelGGL(adj_softmax_fwd_kernel, dim3((unsigned)(B * K)), dim3(1024), 0, (hipStream_t)stream, partial, nchunk,
                       scale, adj_a, adj_b, c_out, a_hat, K, V, use_softmax);
    return launch_status("adj_softmax_fwd");
}

extern "C" int fgcn_adj_softmax_bwd(const float* partial, int nchunk, float scale, const float* c_in,
                                    float* d_a_hat, float* d_s, int B, int K, int V, void* stream) {
    FGCN_REQUIRE(partial && d_a_hat && nchunk > 0 && B > 0 && K > 0 && V > 0 && V <= FGCN_MAX_V, FGCN_E_BADARG,
                 "adj_softmax_bwd: bad argument");
    hipLaunchKernelGGL(adj_softmax_bwd_kernel, dim3((unsigned)(B * K)), dim3(1024), 0, (hipStream_t)stream, partial, nchunk,
                       scale, c_in, d_a_hat, d_s, K, V);
    return launch_status("adj_softmax_bwd");
}

extern "C" int fgcn_joint_mix_vec(const float* in, float* out, const float* mats, int B, int T, int V,
                                  int ld_in, int ld_out, int n_mats, int mats_batched,
                                  const fgcn_mixv_item* items, int n_items, int vw, int accumulate,
                                  float* colsum_partial, unsigned* out_amax, void* stream) {
    FGCN_REQUIRE(in && out && mats && items, FGCN_E_BADARG, "joint_mix_vec: null pointer");
    FGCN_REQUIRE(!(colsum_partial && accumulate), FGCN_E_BADARG, "joint_mix_vec: column sums only without accumulation");
    FGCN_REQUIRE(B > 0 && B <= 65535 && T > 0 && V > 0 && V <= FGCN_MAX_V, FGCN_E_BADARG,
                 "joint_mix_vec: bad B/T/V (%d,%d,%d)", B, T, V);
    FGCN_REQUIRE(n_mats >= 1 && n_mats <= MIX_MAX_MATS && n_items >= 1 && n_items <= FGCN_MIX_MAX_ITEMS,
                 FGCN_E_BADARG, "joint_mix_vec: n_mats=%d n_items=%d out of range", n_mats, n_items);
    FGCN_REQUIRE(vw == 1 || vw == 2, FGCN_E_BADARG, "joint_mix_vec: vw must be 1 or 2 (got %d)", vw);
    FGCN_REQUIRE(ld_in % 4 == 0 && ld_out % 4 == 0 && aligned16(in) && aligned16(out), FGCN_E_ALIGN,
                 "joint_mix_vec: 16-byte alignment");
    const long long in_bytes = (long long)B * T * V * ld_in * 4, out_bytes = (long long)B * T * V * ld_out * 4;
    FGCN_REQUIRE(in_bytes < 0x7FFF0000ll && out_bytes < 0x7FFF0000ll, FGCN_E_BADARG,
                 "joint_mix_vec: tensors must be smaller than 2 GiB");
    MixVP p;
    p.in_bytes = (unsigned)in_bytes; p.out_bytes = (unsigned)out_bytes;
    p.in = in; p.out = out; p.mats = mats;
    p.B = B; p.T = T; p.V = V; p.ld_in = ld_in; p.ld_out = ld_out;
    p.n_mats = n_mats; p.mats_batched = mats_batched; p.n_items = n_items;
    p.t_chunk = pick_t_chunk(B, T);
    p.colsum = colsum_partial;
    p.amax = out_amax;
    for (int i = 0; i < n_items; ++i) {
        const fgcn_mixv_item& it = items[i];
        FGCN_REQUIRE(it.nterms >= 1 && it.nterms <= 3 && it.nch >= vw && it.nch <= 32 * vw && it.nch % vw == 0 &&
                         it.nch == items[0].nch && it.out_c >= 0 && it.out_c % vw == 0 && it.out_c + it.nch <= ld_out,
                     FGCN_E_BADARG, "joint_mix_vec: item %d malformed (all items of a call share one width)", i);
        for (int t = 0; t < it.nterms; ++t)
            FGCN_REQUIRE(it.term[t].mat >= 0 && it.term[t].mat < n_mats && it.term[t].in_c >= 0 &&
                             it.term[t].in_c % vw == 0 && it.term[t].in_c + it.nch <= ld_in,
                         FGCN_E_BADARG, "joint_mix_vec: item %d term %d malformed", i, t);
        p.items[i].out_c = it.out_c;
        p.items[i].nterms = it.nterms;
        for (int t = 0; t < 3; ++t) {
            p.items[i].img[t] = t < it.nterms ? 2 * it.term[t].mat + (it.term[t].transpose ? 1 : 0) : 0;
            p.items[i].in_c[t] = t < it.nterms ? it.term[t].in_c : 0;
        }
    }
    p.nch = items[0].nch;
    dim3 grid((unsigned)cdiv(T, p.t_chunk), (unsigned)B);
    const hipStream_t st = (hipStream_t)stream;
    const int ks = (V + 3) / 4 * 2;  // k-steps, rounded up to even
    const size_t cs_lds = colsum_partial ? (size_t)4 * ld_out * sizeof(float) : 0;
    FGCN_REQUIRE(cs_lds <= 32 * 1024, FGCN_E_BADARG, "joint_mix_vec: ld_out=%d too wide for the column-sum scratch", ld_out);
#define FGCN_MIXV_KS(VW_, ACC_)                                                                                   \
    do {                                                                                                          \
        if (ks <= 10) hipLaunchKernelGGL((joint_mix_vec_kernel<VW_, ACC_, 10>), grid, dim3(256), cs_lds, st, p);      \
        else if (ks <= 12) hipLaunchKernelGGL((joint_mix_vec_kernel<VW_, ACC_, 12>), grid, dim3(256), cs_lds, st, p); \
        else if (ks <= 14) hipLaunchKernelGGL((joint_mix_vec_kernel<VW_, ACC_, 14>), grid, dim3(256), cs_lds, st, p); \
        else hipLaunchKernelGGL((joint_mix_vec_kernel<VW_, ACC_, 16>), grid, dim3(256), cs_lds, st, p);               \
    } while (0)
#define FGCN_MIXV(VW_)                            \
    do {                                          \
        if (accumulate) FGCN_MIXV_KS(VW_, true);  \
        else FGCN_MIXV_KS(VW_, false);            \
    } while (0)
    if (vw == 2) FGCN_MIXV(2);
    else FGCN_MIXV(1);
#undef FGCN_MIXV
#undef FGCN_MIXV_KS
    return launch_status("joint_mix_vec");
}

extern "C" int fgcn_joint_dagg(const float* x, const float* dagg, const float* mats, float* dx, float* partial,
                               int B, int T, int V, int C, int ld_x, int ld_dagg, int ld_dx, int n_subsets,
                               int mats_batched, int t_chunk, int accumulate, const float* extra1,
                               const unsigned char* mask1, const float* extra2, const unsigned char* mask2, void* stream) {
    FGCN_REQUIRE(x && dagg && mats && dx && partial, FGCN_E_BADARG, "joint_dagg: null pointer");
    const int n_extra = extra1 ? (extra2 ? 2 : 1) : 0;
    FGCN_REQUIRE((!extra1 || mask1) && (!extra2 || (mask2 && extra1)), FGCN_E_BADARG, "joint_dagg: a gated addend needs its sign image");
    FGCN_REQUIRE(n_extra == 0 || (C % 8 == 0 && (long long)B * T * V * C * 4 < 0x7FFF0000ll), FGCN_E_BADARG,
                 "joint_dagg: gated addends need C %% 8 == 0 (C=%d) and tensors below 2 GiB", C);
    FGCN_REQUIRE(B > 0 && B <= 65535 && T > 0 && V > 0 && V <= FGCN_MAX_V && C > 0 && t_chunk > 0, FGCN_E_BADARG,
                 "joint_dagg: bad sizes B=%d T=%d V=%d C=%d", B, T, V, C);
    FGCN_REQUIRE(n_subsets >= 1 && n_subsets <= 3, FGCN_E_BADARG, "joint_dagg: n_subsets=%d (1..3)", n_subsets);
    FGCN_REQUIRE(C % 4 == 0 && ld_x % 4 == 0 && ld_dagg % 4 == 0 && ld_x >= C && ld_dagg >= n_subsets * C && ld_dx >= C &&
                     aligned16(x) && aligned16(dagg),
                 FGCN_E_ALIGN, "joint_dagg: C and the row strides must be multiples of 4 and cover the channels");
    const long long xb = (long long)B * T * V * ld_x * 4, db = (long long)B * T * V * ld_dagg * 4;
    FGCN_REQUIRE(xb < 0x7FFF0000ll && db < 0x7FFF0000ll && (long long)t_chunk * V * ld_dx * 4 < 0x7FFF0000ll, FGCN_E_BADARG,
                 "joint_dagg: x and dagg must be smaller than 2 GiB");
    DaggP p;
    p.x = x; p.dagg = dagg; p.mats = mats; p.dx = dx; p.partial = partial;
    p.B = B; p.T = T; p.V = V; p.C = C; p.ld_x = ld_x; p.ld_d = ld_dagg; p.ld_dx = ld_dx; p.n_sub = n_subsets;
    p.mats_batched = mats_batched; p.t_chunk = t_chunk; p.accumulate = accumulate;
    p.x_bytes = (unsigned)xb; p.d_bytes = (unsigned)db;
    p.e[0] = extra1; p.m[0] = mask1; p.e[1] = extra2; p.m[1] = mask2; p.n_extra = n_extra;
    // 49,152 bytes (53,248 with the identity image of the gated form): three workgroups per CU
    const size_t lds = (size_t)((n_extra ? 4 : 3) * IMG + 4 * 2 * 32 * DTS) * sizeof(float);
    dim3 grid((unsigned)cdiv(T, t_chunk), (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    const int ks = (V + 3) / 4 * 2;
    const bool three = (fgcn::tuning(6) & 8) != 0;
    const bool x3 = fgcn::math_mode() != FGCN_MATH_F32 && !(fgcn::tuning(6) & 16);   // key 6 bit 4: the f32-MFMA gram in every mode
    const bool mb3_form = (fgcn::tuning(6) & 2048) != 0;                              // key 6 bit 11: the compile-time form at three workgroups per CU, one tile set (A/B)
    const bool plain_form = (fgcn::tuning(6) & 1024) != 0;                            // key 6 bit 10: run-time subset count / accumulate flag (A/B)
#define FGCN_DAGG(KS_)                                                                                    \
    do {                                                                                                  \
        if (n_extra == 2 && three) hipLaunchKernelGGL((joint_dagg_kernel<KS_, 2, 3>), grid, dim3(256), lds, s, p); \
        else if (n_extra == 2) hipLaunchKernelGGL((joint_dagg_kernel<KS_, 2, 2>), grid, dim3(256), lds, s, p); \
        else if (n_extra == 1) hipLaunchKernelGGL((joint_dagg_kernel<KS_, 1, 2>), grid, dim3(256), lds, s, p); \
        else if (x3 && !three && n_subsets == 3 && !plain_form && !mb3_form) {                            \
            if (accumulate) hipLaunchKernelGGL((joint_dagg_kernel<KS_, 0, 2, true, 3, 1>), grid, dim3(256), lds, s, p); \
            else hipLaunchKernelGGL((joint_dagg_kernel<KS_, 0, 2, true, 3, 0>), grid, dim3(256), lds, s, p); \
        }                                                                                                 \
        else if (x3 && !three && n_subsets == 3 && !plain_form) {                                         \
            if (accumulate) hipLaunchKernelGGL((joint_dagg_kernel<KS_, 0, 3, true, 3, 1>), grid, dim3(256), lds, s, p); \
            else hipLaunchKernelGGL((joint_dagg_kernel<KS_, 0, 3, true, 3, 0>), grid, dim3(256), lds, s, p); \
        }                                                                                                 \
        else if (x3 && !three) hipLaunchKernelGGL((joint_dagg_kernel<KS_, 0, 3, true>), grid, dim3(256), lds, s, p); \
        else if (x3) hipLaunchKernelGGL((joint_dagg_kernel<KS_, 0, 2, true>), grid, dim3(256), lds, s, p); \
        else hipLaunchKernelGGL((joint_dagg_kernel<KS_, 0>), grid, dim3(256), lds, s, p);                 \
    } while (0)
    if (ks <= 10) FGCN_DAGG(10);
    else if (ks <= 12) FGCN_DAGG(12);
    else if (ks <= 14) FGCN_DAGG(14);
    else FGCN_DAGG(16);
#undef FGCN_DAGG
    return launch_status("joint_dagg");
}

extern "C" int fgcn_joint_mix_chunks(int B, int T) { return (int)cdiv(T, pick_t_chunk(B, T)); }

extern "C" int fgcn_spatial_wgrad_chunks(int B, int T, int Cin, int Cout) {
    const long long tiles = cdiv(Cin, 32) * cdiv(Cout, 64);
    // about a thousand workgroups, at least 4 frames (one per wave) each; half as many for small batches, where a workgroup's fixed work (the
    // split A^ planes, six cross-wave sums) outweighs its frames (same-box A/B of the 8-clip step, tuning key 13: 512 -> 9.31 / 9.33 ms,
    // 1024 -> 9.40 / 9.40, 2048 -> 9.52 / 9.53, 256 -> +0.12; at 64 clips 512 and 2048 read the same)
    const int target = fgcn::tuning(13) > 0 ? fgcn::tuning(13) : (B <= 32 ? 512 : 1024);
    long long nchunk = cdiv(target, tiles * B);
    if (nchunk < 1) nchunk = 1;
    if (nchunk > cdiv(T, 4)) nchunk = cdiv(T, 4);
    return (int)cdiv(T, cdiv(T, nchunk));
}

extern "C" int fgcn_spatial_wgrad(const float* x, const float* dy, const float* mats, float* partial,
                                  int B, int T, int V, int Cin, int Cout, int ld_x, int ld_dy, int n_subsets,
                                  int mats_batched, void* stream) {
    FGCN_REQUIRE(x && dy && mats && partial, FGCN_E_BADARG, "spatial_wgrad: null pointer");
    FGCN_REQUIRE(B > 0 && B <= 65535 && T > 0 && V > 0 && V <= FGCN_MAX_V && Cin > 0 && Cout > 0, FGCN_E_BADARG,
                 "spatial_wgrad: bad sizes B=%d T=%d V=%d Cin=%d Cout=%d", B, T, V, Cin, Cout);
    FGCN_REQUIRE(n_subsets >= 1 && n_subsets <= 3, FGCN_E_BADARG, "spatial_wgrad: n_subsets=%d (1..3)", n_subsets);
    FGCN_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0 && ld_x % 4 == 0 && ld_dy % 4 == 0 && ld_x >= Cin && ld_dy >= Cout &&
                     aligned16(x) && aligned16(dy),
                 FGCN_E_ALIGN, "spatial_wgrad: channels and row strides must be multiples of 4");
    const int nchunk = fgcn_spatial_wgrad_chunks(B, T, Cin, Cout);
    const long long xb = (long long)B * T * V * ld_x * 4, yb = (long long)B * T * V * ld_dy * 4;
    const long long pb = (long long)B * nchunk * n_subsets * Cin * Cout * 4;
    FGCN_REQUIRE(xb < 0x7FFF0000ll && yb < 0x7FFF0000ll && pb < 0x7FFF0000ll, FGCN_E_BADARG,
                 "spatial_wgrad: tensors must be smaller than 2 GiB");
    SWgradP p;
    p.x = x; p.dy = dy; p.mats = mats; p.partial = partial;
    p.B = B; p.T = T; p.V = V; p.Cin = Cin; p.Cout = Cout; p.ld_x = ld_x; p.ld_dy = ld_dy; p.n_sub = n_subsets;
    p.mats_batched = mats_batched; p.t_chunk = (int)cdiv(T, nchunk); p.tiles_o = (int)cdiv(Cout, 64);
    p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)yb; p.p_bytes = (unsigned)pb;
    const bool x3m = fgcn::math_mode() == FGCN_MATH_BF16X3 && !(fgcn::tuning(6) & 512);
    // 65,536 bytes (76,288 with the split A^ planes of the bf16x3 form): two workgroups per CU
    const size_t lds = (size_t)((x3m ? 9 * 32 * SW_AHB / 4 : 3 * IMG) + 4 * (32 * DTS + 32 * YTS)) * sizeof(float);
    dim3 grid((unsigned)(cdiv(Cin, 32) * p.tiles_o), (unsigned)nchunk, (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    const bool bf = fgcn::math_mode() == FGCN_MATH_BF16;
    const bool x3 = fgcn::math_mode() == FGCN_MATH_BF16X3 && !(fgcn::tuning(6) & 512);   // key 6 bit 9: the exact-f32 form (A/B control)
    const int ks = (V + 3) / 4 * 2;
    static bool lds_opt_in = false;   // once per process; not a stream operation (stays out of graph captures)
#define FGCN_SW(KS_)                                                                                                   \
    do {                                                                                                               \
        if (bf) hipLaunchKernelGGL((spatial_wgrad_kernel<KS_, 1>), grid, dim3(256), lds, s, p);                        \
        else if (x3) hipLaunchKernelGGL((spatial_wgrad_kernel<KS_, 2>), grid, dim3(256), lds, s, p);                   \
        else hipLaunchKernelGGL((spatial_wgrad_kernel<KS_, 0>), grid, dim3(256), lds, s, p);                           \
    } while (0)
    const int max_lds = (int)((9 * 32 * SW_AHB / 4 + 4 * (32 * DTS + 32 * YTS)) * sizeof(float));   // the larger (bf16x3) form
#define FGCN_SW_ATTR(KS_)                                                                                             \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_wgrad_kernel<KS_, 0>),                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);                                  \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_wgrad_kernel<KS_, 2>),                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);                                   \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_wgrad_kernel<KS_, 1>),                          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)
    if (!lds_opt_in) {
        FGCN_SW_ATTR(10); FGCN_SW_ATTR(12); FGCN_SW_ATTR(14); FGCN_SW_ATTR(16);
        lds_opt_in = true;
    }
    if (ks <= 10) FGCN_SW(10);
    else if (ks <= 12) FGCN_SW(12);
    else if (ks <= 14) FGCN_SW(14);
    else FGCN_SW(16);
#undef FGCN_SW
#undef FGCN_SW_ATTR
    return launch_status("spatial_wgrad");
}
