// Fused spatial graph convolution, forward (north-star kernel 1):
//
//     y[(n,t,w), o] = bias[o] + sum_k sum_c Wd_k[o][c] * ( sum_v x[(n,t,v), c] * A^_k[n][v][w] )
//
// i.e. the K=3 joint aggregations x.A^_k and the three 1x1 convolutions conv_d[k] of the reference's
// SpatialGraphConv.forward (torch_src/models/mmargcn/agcn.py:103-111) in ONE kernel, with no agg tensor in HBM.
//
// Mapping (one wave = one frame t of one sample n at a time; a workgroup = 4 waves = 4 frames of the same sample):
//   step 1  agg_k^T (32 c x 32 w) = X_t^T (c x v) . A^_k (v x w)     13 x v_mfma_f32_32x32x2_f32 for V = 25
//           A operand: x[(n,t,v)][c0 + lane]  -> 128-byte contiguous global reads per half-wave, read ONCE per
//           channel tile and reused for the three subsets;  B operand: A^_k[v][w] from LDS (per-sample, zero-padded
//           32 x 32, row stride 33: conflict-free).
//   step 2  y^T (32 o x 32 w) += Wd_k (o x c) . agg_k^T (c x w)      16 MFMAs per (o tile, c tile, k)
//           the step-1 accumulator IS the B operand of step 2 (register r of lane half h holds row
//           (r&3)+8(r>>2)+4h of agg^T, exactly the k index a 32x32x2 B operand needs), so agg never leaves
//           registers; the A operand Wd_k[o][c] streams from L2 straight into registers as 16-byte buffer loads from
//           a k-interleaved packing wd4[(k*Cin+c)/4][o][4] (lane = output channel, one load = the 4 consecutive c of
//           registers 4g..4g+3), requested one step ahead.  No LDS staging of weights and no workgroup barrier in the
//           frame loop: the four waves of a workgroup run independently.
//   epilogue: each 32x32 accumulator goes through a wave-private LDS tile, so rows leave as contiguous 128-byte
//           segments (+ sum_k bd_k) and the BatchNorm partial sums (sum, sum of squares per channel) need 3 shuffle
//           steps per value instead of 5.
// The joint axis sits on 25 of the 32 MFMA columns (78 % of the f32 MFMA rate is the ceiling of this mapping);
// A^_k's zero padding makes the 7 idle columns exact zeros, so they drop out of stores and statistics.
#include "fgcn_common.hpp"

// Timing probes of spatial_fwd_x3_kernel (wrong results; tools/build_probe.py only): bit 0 = no output stores, bit 1 = no step-1 MFMAs,
// bit 2 = no step-2 MFMAs, bit 3 = the weight fragments are loaded once
#ifndef FGCN_PROBE_SP
#define FGCN_PROBE_SP 0
#endif

namespace fgcn {

constexpr int AHS = 33;

struct SpatialP {
    const float* x;
    const float* a_hat;
    const float* wd;
    const float* bias;
    float* y;
    float* stats;
    int B, T, V, Cin, Cout, ld_x, ld_y, ns, a_batched, t_chunk;
    unsigned x_bytes, w_bytes;
    unsigned w_plane_bytes;   // FGCN_MATH_BF16X3: bytes of one part of the split weights
    // per_xcd > 0 (spatial_fwd_x3_kernel): 1-D grid in XCD-aware order, column block fastest -- the Cout / 64 column blocks of one
    // (sample, frame chunk) re-read the same x rows; run back to back on one XCD they find them in its L2 (id b -> virtual
    // workgroup (b % 8) * per_xcd + b / 8, consecutive ids go round-robin over the 8 XCDs)
    int per_xcd, nchunk, ncol;
};

constexpr int TTS = 36;   // row stride of the per-wave transpose tile (32 channels + 4 pad)
using u32x4s = __attribute__((ext_vector_type(4))) unsigned int;

__device__ __forceinline__ float sp_load1(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}
__device__ __forceinline__ f32x4 sp_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// MM: math mode of step 2 (the Cin x Cout contraction); step 1 (the <= 32-joint mixing) stays f32.
//   FGCN_MATH_BF16  : operands rounded as the fragments are formed, one bf16 MFMA per four f32 ones.
//   FGCN_MATH_BF16X3: f32-accurate split products on v_mfma_f32_32x32x16_bf16 (fgcn_common.hpp).  Registers 8gp..8gp+7 of
//                     the step-1 accumulator are one operand fragment (rows 16gp + 4h + (j&3) + 8(j>>2)), split in
//                     registers once per CT_OUT column tiles; the weights come pre-split in the same row order
//                     (fgcn_pack_split3, acc_order): [part][(k*Cin + c)/16][h][o][8].  Whole 32-channel tiles only.
template <int CT_IN, int CT_OUT, int MM>
__global__ __launch_bounds__(256, ((CT_OUT <= 2 || (CT_OUT == 4 && (CT_IN <= 2 || MM == 2))) ? 2 : 1)) void spatial_fwd_kernel(SpatialP p) {
    constexpr int WROW = CT_OUT * 32;                 // padded Cout
    constexpr unsigned OOB = 0x80000000u;             // buffer offset beyond num_records: the load returns 0
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ah = smem;                                 // [3][32][33]
    float* st = smem + ((3 * 32 * AHS + 3) & ~3);     // [4 waves][2][WROW]
    float* tt = st + 4 * 2 * WROW;                    // [4 waves][32][TTS] accumulator transpose tiles
    float* bl = tt + 4 * 32 * TTS;                    // [WROW] bias (zeros where absent)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int n = blockIdx.y;
    const int V = p.V, NS = p.ns;
    const int t0 = blockIdx.x * p.t_chunk;
    const int t1 = min(t0 + p.t_chunk, p.T);

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wd, 0, p.w_bytes, 0x00020000);
    // output rows of this workgroup's frames [t0, t1): offsets relative to frame t0 (no tensor-size limit); stores are
    // branch-free buffer stores (absent joints / channels carry the out-of-range offset and are dropped) -- guarded
    // global stores made hipcc drain vmcnt(0), i.e. the next frame's x prefetch, in front of every row store
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.y + ((long long)n * p.T + t0) * V * p.ld_y), 0, (unsigned)((t1 - t0) * V * p.ld_y) * 4u, 0x00020000);

    const float* asrc = p.a_hat + (p.a_batched ? (long long)n * NS * V * V : 0);
    for (int i = tid; i < 3 * 32 * 32; i += 256) {
        const int k = i >> 10, v = (i >> 5) & 31, w = i & 31;
        ah[(k * 32 + v) * AHS + w] = (k < NS && v < V && w < V) ? asrc[(k * V + v) * V + w] : 0.f;
    }
    for (int i = tid; i < 4 * 2 * WROW; i += 256) st[i] = 0.f;
    // blockIdx.z = column block of WROW outputs (bf16x3 runs 256 outputs as two 128-column blocks: two workgroups per CU
    // and no register spills, at the price of forming agg twice)
    const int ob = blockIdx.z * WROW;
    for (int i = tid; i < WROW; i += 256) bl[i] = (p.bias && ob + i < p.Cout) ? p.bias[ob + i] : 0.f;
    __syncthreads();                                  // the only workgroup barrier before the final statistics sum

    const int ksteps = (V + 1) >> 1;
    // Weights stream from L2 straight into registers: wd4[(k*Cin + c)/4][o][4] (k-interleaved packing), lane = output
    // channel o, one 16-byte buffer load = the 4 consecutive input channels c = 8g + 4h + e that registers 4g..4g+3 of
    // the step-1 accumulator hold as the contraction index.  No LDS staging, no barriers: waves run independently.
    unsigned wvo[CT_OUT];
#pragma unroll
    for (int ot = 0; ot < CT_OUT; ++ot) {
        const int o = ob + ot * 32 + l31;
        wvo[ot] = o < p.Cout ? (unsigned)(h * p.Cout + o) * 16u : OOB;
    }
    // x of frame t, channel tile ci: 16 branch-free dword loads (joint 2s + h of channel ci*32 + lane)
    auto load_x = [&](int t, int ci, float (&xv)[16]) {
        const int c = ci * 32 + l31;
        const bool ok = t < t1 && c < p.Cin;
        const unsigned base = (unsigned)((((long long)n * p.T + (t < t1 ? t : t0)) * V) * p.ld_x + c) * 4u;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int v = 2 * s + h;
            xv[s] = sp_load1(rx, (ok && s < ksteps && v < V) ? base + (unsigned)(v * p.ld_x) * 4u : OOB);
        }
    };

    // weight fragment of step (ci, k, g): 4 consecutive input channels c = ci*32 + 8g + 4h + e for this lane's o
    auto load_w = [&](int ci, int k, int g, auto& wv) {
        const unsigned so = (unsigned)((((k * p.Cin + ci * 32) >> 2) + 2 * g) * p.Cout) * 16u;
        const bool gok = ci * 32 + 8 * g + 4 * h < p.Cin;
#pragma unroll
        for (int ot = 0; ot < CT_OUT; ++ot) wv[ot] = sp_load4(rw, gok ? wvo[ot] : OOB, so);
    };

    // LATE_X (bf16x3 with >= 4 output tiles): no second x tile -- the next tile is loaded into xcur right after the last
    // subset's step 1 consumed it (its step 2 covers the latency); 16 registers less
    constexpr bool LATE_X = MM == 2 && CT_OUT >= 4;
    float xcur[16], xnxt[LATE_X ? 1 : 16];
    // at 256 outputs a second weight set measured slower in f32 (1.33 -> 1.47 ms); with bf16 MFMAs (8 per weight group instead
    // of 32) the un-prefetched loads are pure exposed latency (2.4 ms), so that mode always prefetches
    constexpr bool BF = MM == 1;
    constexpr bool PREFETCH_W = CT_OUT <= 4 || BF;
    f32x4 wcur[MM == 2 ? 1 : CT_OUT], wnxt[(PREFETCH_W && MM != 2) ? CT_OUT : 1];
    // bf16x3: the fragment sets of one 16-channel group (3 parts x CT_OUT column tiles); a column tile's set is reloaded
    // with the next group's right after its MFMAs were issued, so CT_OUT - 1 units of MFMAs cover the load
    u32x4v w3[MM == 2 ? CT_OUT : 1][3];
    auto load_w3 = [&](int ci, int k, int gp, int ot, u32x4v (&wv)[3]) {
        const unsigned so = (unsigned)((((k * p.Cin + ci * 32) >> 4) + gp) * 2 * p.Cout) * 16u;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            wv[pl] = __builtin_amdgcn_raw_buffer_load_b128(rw, wvo[ot], so + pl * p.w_plane_bytes, 0);
    };
    load_x(t0 + wave, 0, xcur);
    if constexpr (MM == 2) {
#pragma unroll
        for (int ot = 0; ot < CT_OUT; ++ot) load_w3(0, 0, 0, ot, w3[ot]);
    } else if constexpr (PREFETCH_W) {
        load_w(0, 0, 0, wcur);
    }
    for (int tg = t0; tg < t1; tg += 4) {
        const int t = tg + wave;
        const bool tv = t < t1;
        f32x16 acc[CT_OUT];
#pragma unroll
        for (int i = 0; i < CT_OUT; ++i) acc[i] = zero16();

#pragma unroll 1
        for (int ci = 0; ci < CT_IN; ++ci) {
            // next tile's x (next channel tile, or the first tile of this wave's next frame) flies during the MFMAs
            if constexpr (!LATE_X) {
                if (ci + 1 < CT_IN) load_x(t, ci + 1, xnxt);
                else load_x(t + 4, 0, xnxt);
            }
            const int cleft = p.Cin - ci * 32;          // valid input channels in this tile
#pragma unroll 1
            for (int k = 0; k < NS; ++k) {
                // step 1: agg^T tile (32 c x 32 w)
                f32x16 agg = zero16();
                const float* ak = ah + (k * 32) * AHS + l31;
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    if (s < ksteps) agg = mfma32(xcur[s], ak[(2 * s + h) * AHS], agg);
                if constexpr (LATE_X) {
                    if (k == NS - 1) {
                        if (ci + 1 < CT_IN) load_x(t, ci + 1, xcur);
                        else load_x(t + 4, 0, xcur);
                    }
                }
                // step 2: y^T tiles; registers 4g..4g+3 of agg are contraction rows c = 8g + 4h + (0..3).
                // The weights of the NEXT step (next group, next subset, next channel tile, or the next frame's first
                // step) are requested before this step's MFMAs.
                const int nG = cleft >= 32 ? 4 : (cleft + 7) >> 3;   // channel groups that exist (Cin = 4: one)
                if constexpr (MM == 2) {
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        int nci = ci, nk = k, ngp = 1;               // the group after this one (wave-uniform)
                        if (gp == 1) {
                            ngp = 0;
                            if (k + 1 < NS) nk = k + 1;
                            else if (ci + 1 < CT_IN) nci = ci + 1, nk = 0;
                            else nci = 0, nk = 0;
                        }
                        u32x2 h0, m0, l0, h1, m1, l1;
                        split3_x4(f32x4{agg[8 * gp], agg[8 * gp + 1], agg[8 * gp + 2], agg[8 * gp + 3]}, h0, m0, l0);
                        split3_x4(f32x4{agg[8 * gp + 4], agg[8 * gp + 5], agg[8 * gp + 6], agg[8 * gp + 7]}, h1, m1, l1);
                        const u32x4v b3[3] = {u32x4v{h0[0], h0[1], h1[0], h1[1]}, u32x4v{m0[0], m0[1], m1[0], m1[1]},
                                              u32x4v{l0[0], l0[1], l1[0], l1[1]}};
#pragma unroll
                        for (int ot = 0; ot < CT_OUT; ++ot) {
                            acc[ot] = mfma_x3_k16(w3[ot], b3, acc[ot]);
                            load_w3(nci, nk, ngp, ot, w3[ot]);
                        }
                    }
                } else
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g < nG) {                       // wave-uniform
                        if constexpr (PREFETCH_W) {
                            if (g + 1 < nG) load_w(ci, k, g + 1, wnxt);
                            else if (k + 1 < NS) load_w(ci, k + 1, 0, wnxt);
                            else if (ci + 1 < CT_IN) load_w(ci + 1, 0, 0, wnxt);
                            else load_w(0, 0, 0, wnxt);
                        } else {
                            load_w(ci, k, g, wcur);     // 256 output channels: no registers left for a second set
                        }
                        if constexpr (BF) {
                            const s16x4 bp = pack_bf16(agg[4 * g], agg[4 * g + 1], agg[4 * g + 2], agg[4 * g + 3]);
#pragma unroll
                            for (int ot = 0; ot < CT_OUT; ++ot) acc[ot] = mfma_bf16(pack_bf16(wcur[ot]), bp, acc[ot]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
#pragma unroll
                                for (int ot = 0; ot < CT_OUT; ++ot) acc[ot] = mfma32(wcur[ot][e], agg[4 * g + e], acc[ot]);
                        }
                        if constexpr (PREFETCH_W) {
#pragma unroll
                            for (int ot = 0; ot < CT_OUT; ++ot) wcur[ot] = wnxt[ot];
                        }
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < 16; ++s)
                if constexpr (!LATE_X) xcur[s] = xnxt[s];
        }

        // ---- epilogue for this frame: transpose each 32(o) x 32(w) accumulator through a wave-private LDS tile so
        //      rows leave as contiguous 128-byte segments and the per-channel sums need 3 shuffle steps, not 5 x 16 ---
        float* T = tt + wave * 32 * TTS;
        const int rr = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
        for (int ot = 0; ot < CT_OUT; ++ot) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(&T[l31 * TTS + 8 * g + 4 * h]) =
                    f32x4{acc[ot][4 * g], acc[ot][4 * g + 1], acc[ot][4 * g + 2], acc[ot][4 * g + 3]};
            const int ol = ot * 32 + c4, o = ob + ol;   // column inside this block / in the tensor
            const bool ook = o < p.Cout;                // Cout % 4 == 0: the quad is all-in or all-out
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(&bl[ol]);
            f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = rr + 8 * i;
                const f32x4 val = *reinterpret_cast<const f32x4*>(&T[r * TTS + c4]) + b4;
                const bool keep = tv && r < V && ook;
                const unsigned off = keep ? (unsigned)(((t - t0) * V + r) * p.ld_y + o) * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, val), ry, off, 0, 0);
                const f32x4 kept = keep ? val : f32x4{0.f, 0.f, 0.f, 0.f};
                s1 += kept;
                s2 += kept * kept;
            }
            if (p.stats) {
#pragma unroll
                for (int m = 8; m <= 32; m <<= 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s1[e] += __shfl_xor(s1[e], m);
                        s2[e] += __shfl_xor(s2[e], m);
                    }
                }
                if (lane < 8 && ook) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        st[(wave * 2 + 0) * WROW + ol + e] += s1[e];
                        st[(wave * 2 + 1) * WROW + ol + e] += s2[e];
                    }
                }
            }
        }
    }

    if (p.stats) {
        __syncthreads();
        const long long wg = (long long)blockIdx.y * gridDim.x + blockIdx.x;
        for (int i = tid; i < 2 * WROW; i += 256) {
            const int which = i / WROW, o = i - which * WROW;
            if (ob + o < p.Cout)
                p.stats[(wg * 2 + which) * p.Cout + ob + o] = st[(0 * 2 + which) * WROW + o] + st[(1 * 2 + which) * WROW + o] +
                                                          st[(2 * 2 + which) * WROW + o] + st[(3 * 2 + which) * WROW + o];
        }
    }
}


// ---- FGCN_MATH_BF16X3, whole 32-channel input tiles --------------------------------------------------------------------
// The kernel above (MM = 2) hands every weight fragment to ONE frame (32 MFMA columns): at 6 bytes per weight and six
// 32-cycle MFMAs per fragment set each wave pulls 16 B/clk of weights, 64 B/clk per CU -- the L2 -> CU limit -- and its step 1
// still runs on the f32 MFMA (13 x 64 cycles per tile and subset).  Here
//   * a wave owns TWO frames: every weight fragment set feeds both (half the weight stream per FLOP);
//   * step 1 is split-bf16 too: x is split once per (frame, channel tile) as it arrives (lane = channel, 8 consecutive joints
//     per 16-byte fragment), A^_k is split once per workgroup into LDS as [subset][part][w][v] bf16 (one ds_read_b128 = the 8
//     joints of a lane's fragment): 2 x 6 MFMAs of 32 cycles instead of 13 of 64;
//   * a workgroup covers 64 output columns (blockIdx.z = column block; agg is re-formed per block -- at 384 cycles per tile
//     and subset that is cheaper than the registers a wider block would need for two frames).
// Same accumulator-as-operand chain, weight format (fgcn_pack_split3, acc_order), epilogue and statistics as above.
constexpr int AHB = 80;    // bytes per [w] row of a split A^ plane (32 joints x bf16 + 16 pad: conflict-free b128 reads)

// NP = 3: three bf16 parts per operand, six products (FGCN_PRODUCTS_BF16X3).  NP = 2: two f16 parts, three products
// (FGCN_PRODUCTS_F16X2, fgcn_common.hpp) with power-of-two block scales, all of them wave-private because the waves of this kernel are
// independent: x per (frame pair, channel tile) from the wave's largest |x| (ex), A^ per workgroup as it is written to LDS (eA), the
// aggregation per (tile, subset) from the wave's largest accumulator of step 1 (e2), the weights per packed form (ew, header of
// FGCN_PACK_SPLIT2H_ACC).  The step-2 accumulators carry S = ex + eA + e2 + ew; when a tile's natural S differs from the one in
// force they are rescaled (a power of two: exact; upwards only while a running bound of log2 |acc| stays below 120, the remainder goes
// into e2) and the epilogue multiplies 2^-S back out.
template <int CT_IN, int NP = 3, bool STR = false>              // STR: non-temporal output stores (fgcn_common.hpp, stream_out)
__global__ __launch_bounds__(256, 2) void spatial_fwd_x3_kernel(SpatialP p) {
    static_assert(NP == 2 || NP == 3, "three bf16 parts or two f16 parts");
    constexpr int CT_OUT = 2, WROW = CT_OUT * 32;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char* ahs = reinterpret_cast<unsigned char*>(smem);      // [3 subsets][NP parts][32 w][AHB]
    float* st = smem + (9 * 32 * AHB) / 4;                            // [4 waves][2][WROW]
    float* tt = st + 4 * 2 * WROW;                                    // [4 waves][32][TTS]
    float* bl = tt + 4 * 32 * TTS;                                    // [WROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.per_xcd > 0) {
        const int vid = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
        if (vid >= p.nchunk * p.B * p.ncol) return;
        bz = vid % p.ncol;
        const int rest = vid / p.ncol;
        bx = rest % p.nchunk;
        by = rest / p.nchunk;
    }
    const int n = by;
    const int V = p.V, NS = p.ns;
    const int t0 = bx * p.t_chunk;
    const int t1 = min(t0 + p.t_chunk, p.T);

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const unsigned char*>(p.wd) + (NP == 2 ? 16 : 0)), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.y + ((long long)n * p.T + t0) * V * p.ld_y), 0, (unsigned)((t1 - t0) * V * p.ld_y) * 4u, 0x00020000);

    const float* asrc = p.a_hat + (p.a_batched ? (long long)n * NS * V * V : 0);
    int eA = 0;                                                       // NP == 2: scale of this sample's A^ images
    if constexpr (NP == 2) {
        float m = 0.f;
        for (int i = tid; i < 3 * 32 * 32; i += 256) {
            const int k = i >> 10, w = (i >> 5) & 31, v = i & 31;
            m = fmaxf(m, (k < NS && v < V && w < V) ? fabsf(asrc[(k * V + v) * V + w]) : 0.f);
        }
        m = wave_reduce_max(m);
        if (lane == 0) bl[wave] = m;                                  // (bl is written again below, behind the barrier)
        __syncthreads();
        eA = __builtin_amdgcn_readfirstlane(min(scale_exp_for(__builtin_bit_cast(unsigned, fmaxf(fmaxf(bl[0], bl[1]), fmaxf(bl[2], bl[3])))), 126));
        __syncthreads();
    }
    const float scA = exp2i(eA);
    for (int i = tid; i < 3 * 32 * 32; i += 256) {
        const int k = i >> 10, w = (i >> 5) & 31, v = i & 31;
        const float a = (k < NS && v < V && w < V) ? asrc[(k * V + v) * V + w] : 0.f;
        unsigned short* d = reinterpret_cast<unsigned short*>(ahs + ((k * NP) * 32 + w) * AHB) + v;
        if constexpr (NP == 2) {
            unsigned ph, pl;
            split_f16_pair(a * scA, 0.f, ph, pl);
            d[0] = (unsigned short)ph;
            d[32 * AHB / 2] = (unsigned short)pl;
        } else {
            unsigned ph, pm, pl;
            split_bf16_pair(a, 0.f, ph, pm, pl);
            d[0] = (unsigned short)ph;
            d[32 * AHB / 2] = (unsigned short)pm;
            d[2 * 32 * AHB / 2] = (unsigned short)pl;
        }
    }
    for (int i = tid; i < 4 * 2 * WROW; i += 256) st[i] = 0.f;
    const int ob = bz * WROW;
    for (int i = tid; i < WROW; i += 256) bl[i] = (p.bias && ob + i < p.Cout) ? p.bias[ob + i] : 0.f;
    __syncthreads();

    unsigned wvo[CT_OUT];
#pragma unroll
    for (int ot = 0; ot < CT_OUT; ++ot) {
        const int o = ob + ot * 32 + l31;
        wvo[ot] = o < p.Cout ? (unsigned)(h * p.Cout + o) * 16u : OOB;
    }
    // x of frame t, channel tile ci: lane = channel, register 8s + j = joint 16s + 8h + j (the k order of the 32x32x16 fragment);
    // the joint's row offset splits into a per-lane part (8h rows) and a scalar part (16s + j rows, the instruction's soffset)
    const unsigned row_b = (unsigned)p.ld_x * 4u;
    auto load_x = [&](int t, int ci, float (&xv)[16]) {
        const bool ok = t < t1;
        const unsigned base = (unsigned)((((long long)n * p.T + (ok ? t : t0)) * V + 8 * h) * p.ld_x + ci * 32 + l31) * 4u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int vs = 16 * (r >> 3) + (r & 7);
            xv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (ok && vs + 8 * h < V) ? base : OOB,
                                                                                   (unsigned)vs * row_b, 0));
        }
    };
    u32x4v w3[CT_OUT][NP];
    auto load_w3 = [&](int ci, int k, int gp, int ot, u32x4v (&wv)[NP]) {
        const unsigned so = (unsigned)((((k * p.Cin + ci * 32) >> 4) + gp) * 2 * p.Cout) * 16u;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            wv[pl] = __builtin_amdgcn_raw_buffer_load_b128(rw, wvo[ot], so + pl * p.w_plane_bytes, 0);
    };
    const unsigned char* af_lane = ahs + l31 * AHB + 16 * h;           // + (k*NP + part) * 32 * AHB + 32 * s
    const int ew = NP == 2 ? min(scale_exp_for(*reinterpret_cast<const unsigned*>(p.wd)), 126) : 0;
    constexpr int S_NONE = 100000;

    float xr[2][16];
    load_x(t0 + 2 * wave, 0, xr[0]);
    load_x(t0 + 2 * wave + 1, 0, xr[1]);
#pragma unroll
    for (int ot = 0; ot < CT_OUT; ++ot) load_w3(0, 0, 0, ot, w3[ot]);

    for (int tg = t0; tg < t1; tg += 8) {
        const int tA = tg + 2 * wave;
        f32x16 acc[2][CT_OUT];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int i = 0; i < CT_OUT; ++i) acc[f][i] = zero16();
        int S = S_NONE, abound = 0;                                   // NP == 2: scale exponent the accumulators carry, log2 bound of |acc|

#pragma unroll 1
        for (int ci = 0; ci < CT_IN; ++ci) {
            u32x4v xs[2][2][NP];                                      // [frame][16-joint step][part]
            int ex = 0;
            if constexpr (NP == 2) {
                const float m = wave_reduce_max(wave_max_abs16(xr[1], wave_max_abs16(xr[0], 0.f)));
                ex = __builtin_amdgcn_readfirstlane(min(scale_exp_for(__builtin_bit_cast(unsigned, m)), 126));
                const float scx = exp2i(ex);
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
                        split2h_x8(xr[f][8 * s2], xr[f][8 * s2 + 1], xr[f][8 * s2 + 2], xr[f][8 * s2 + 3], xr[f][8 * s2 + 4],
                                   xr[f][8 * s2 + 5], xr[f][8 * s2 + 6], xr[f][8 * s2 + 7], scx, xs[f][s2]);
            } else {
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
                        split3_x8(xr[f][8 * s2], xr[f][8 * s2 + 1], xr[f][8 * s2 + 2], xr[f][8 * s2 + 3], xr[f][8 * s2 + 4],
                                  xr[f][8 * s2 + 5], xr[f][8 * s2 + 6], xr[f][8 * s2 + 7], xs[f][s2]);
            }
            // one subset: step 1 for both frames, then step 2; `last` (compile time) = the tile's final subset, after whose
            // step 1 the split x is dead and the next tile's x is requested into the same registers
            auto subset = [&](int k, auto last) {
                f32x16 agg[2] = {zero16(), zero16()};
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4v af[NP];
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        af[pl] = *reinterpret_cast<const u32x4v*>(af_lane + (k * NP + pl) * 32 * AHB + 32 * s2);
                    if constexpr ((FGCN_PROBE_SP & 2) != 0) {
                        agg[0][s2] += __builtin_bit_cast(float, xs[0][s2][0][0] ^ af[0][0]);
                        agg[1][s2] += __builtin_bit_cast(float, xs[1][s2][0][0] ^ af[0][1]);
                    } else if constexpr (NP == 2) {
                        agg[0] = mfma_h2_k16(xs[0][s2], af, agg[0]);
                        agg[1] = mfma_h2_k16(xs[1][s2], af, agg[1]);
                    } else {
                        agg[0] = mfma_x3_k16(xs[0][s2], af, agg[0]);
                        agg[1] = mfma_x3_k16(xs[1][s2], af, agg[1]);
                    }
                }
                if constexpr (decltype(last)::value) {
                    if (ci + 1 < CT_IN) {
                        load_x(tA, ci + 1, xr[0]);
                        load_x(tA + 1, ci + 1, xr[1]);
                    } else {
                        load_x(tA + 8, 0, xr[0]);
                        load_x(tA + 9, 0, xr[1]);
                    }
                }
                // NP == 2: the aggregation (in units of 2^-(ex + eA)) is split at 2^e2, e2 from the wave's largest accumulator; the
                // step-2 accumulators then want S = ex + eA + e2 + ew -- they follow (exactly), or e2 gives way where they cannot
                float sc2 = 1.f;
                if constexpr (NP == 2) {
                    float m = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) m = fmaxf(m, fmaxf(fabsf(agg[0][r]), fabsf(agg[1][r])));
                    const unsigned mb = __builtin_bit_cast(unsigned, wave_reduce_max(m));
                    if ((mb >> 23) != 0u) {                           // (wave-uniform; an all-zero aggregation adds nothing)
                        int e2 = __builtin_amdgcn_readfirstlane(min(scale_exp_for(mb), 126));
                        const int want = ex + eA + e2 + ew;
                        if (S == S_NONE) S = want;
                        else if (want != S) {
                            int d = want - S;
                            if (d > 120 - abound) d = 120 - abound;
                            if (d != 0) {
                                const float fsc = exp2i(d);
#pragma unroll
                                for (int f = 0; f < 2; ++f)
#pragma unroll
                                    for (int ot = 0; ot < CT_OUT; ++ot) acc[f][ot] *= fsc;
                                S += d;
                                abound += d;
                            }
                            e2 -= want - S;                           // what the accumulators could not follow
                        }
                        sc2 = exp2i(e2);                              // (|e2| <= 126 unless the tiles of one frame pair span > 2^120)
                        abound = (abound > 44 ? abound : 44) + 1;
                    }
                }
                // step 2: registers 8gp..8gp+7 of agg are contraction rows c = 16gp + 4h + (j&3) + 8(j>>2)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    int nci = ci, nk = k, ngp = 1;                    // the group after this one (wave-uniform)
                    if (gp == 1) {
                        ngp = 0;
                        if (!decltype(last)::value) nk = k + 1;
                        else if (ci + 1 < CT_IN) nci = ci + 1, nk = 0;
                        else nci = 0, nk = 0;
                    }
                    u32x4v b3[2][NP];
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        if constexpr (NP == 2) {
                            split2h_x8(agg[f][8 * gp], agg[f][8 * gp + 1], agg[f][8 * gp + 2], agg[f][8 * gp + 3], agg[f][8 * gp + 4],
                                       agg[f][8 * gp + 5], agg[f][8 * gp + 6], agg[f][8 * gp + 7], sc2, b3[f]);
                        } else {
                            u32x2 h0, m0, l0, h1, m1, l1;
                            split3_x4(f32x4{agg[f][8 * gp], agg[f][8 * gp + 1], agg[f][8 * gp + 2], agg[f][8 * gp + 3]}, h0, m0, l0);
                            split3_x4(f32x4{agg[f][8 * gp + 4], agg[f][8 * gp + 5], agg[f][8 * gp + 6], agg[f][8 * gp + 7]}, h1, m1, l1);
                            b3[f][0] = u32x4v{h0[0], h0[1], h1[0], h1[1]};
                            b3[f][1] = u32x4v{m0[0], m0[1], m1[0], m1[1]};
                            b3[f][NP - 1] = u32x4v{l0[0], l0[1], l1[0], l1[1]};
                        }
                    }
#pragma unroll
                    for (int ot = 0; ot < CT_OUT; ++ot) {
                        if constexpr ((FGCN_PROBE_SP & 4) != 0) {
                            acc[0][ot][gp] += __builtin_bit_cast(float, w3[ot][0][0] ^ b3[0][0][0] ^ b3[0][NP - 1][3]);
                            acc[1][ot][gp] += __builtin_bit_cast(float, w3[ot][NP - 1][3] ^ b3[1][0][0] ^ b3[1][NP - 1][3]);
                        } else if constexpr (NP == 2) {
                            acc[0][ot] = mfma_h2_k16(w3[ot], b3[0], acc[0][ot]);
                            acc[1][ot] = mfma_h2_k16(w3[ot], b3[1], acc[1][ot]);
                        } else {
                            acc[0][ot] = mfma_x3_k16(w3[ot], b3[0], acc[0][ot]);
                            acc[1][ot] = mfma_x3_k16(w3[ot], b3[1], acc[1][ot]);
                        }
                        if constexpr ((FGCN_PROBE_SP & 8) == 0) load_w3(nci, nk, ngp, ot, w3[ot]);
                    }
                }
            };
#pragma unroll 1
            for (int k = 0; k + 1 < NS; ++k) subset(k, std::false_type{});
            subset(NS - 1, std::true_type{});
        }
        if constexpr (NP == 2) {                                       // back to true units (two factors: |S| may exceed 126)
            if (S != S_NONE) {
                const float u0 = exp2i(-(S / 4)), u1 = exp2i(-(S - 3 * (S / 4)));   // (four factors: S is a sum of four exponents)
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int ot = 0; ot < CT_OUT; ++ot) acc[f][ot] = ((acc[f][ot] * u0) * u0) * u0 * u1;
            }
        }

        // ---- epilogue, one frame after the other through the wave-private transpose tile ------------------------------------
        float* T = tt + wave * 32 * TTS;
        const int rr = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int t = tA + f;
            const bool tv = t < t1;
#pragma unroll
            for (int ot = 0; ot < CT_OUT; ++ot) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(&T[l31 * TTS + 8 * g + 4 * h]) =
                        f32x4{acc[f][ot][4 * g], acc[f][ot][4 * g + 1], acc[f][ot][4 * g + 2], acc[f][ot][4 * g + 3]};
                const int ol = ot * 32 + c4, o = ob + ol;
                const bool ook = o < p.Cout;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(&bl[ol]);
                f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = rr + 8 * i;
                    const f32x4 val = *reinterpret_cast<const f32x4*>(&T[r * TTS + c4]) + b4;
                    const bool keep = tv && r < V && ook;
                    const unsigned off = (keep && (!(FGCN_PROBE_SP & 1) || val[0] == 123.456f)) ? (unsigned)(((t - t0) * V + r) * p.ld_y + o) * 4u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, val), ry, off, 0, STR ? FGCN_STORE_AUX : 0);
                    const f32x4 kept = keep ? val : f32x4{0.f, 0.f, 0.f, 0.f};
                    s1 += kept;
                    s2 += kept * kept;
                }
                if (p.stats) {
#pragma unroll
                    for (int m = 8; m <= 32; m <<= 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            s1[e] += __shfl_xor(s1[e], m);
                            s2[e] += __shfl_xor(s2[e], m);
                        }
                    }
                    if (lane < 8 && ook) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            st[(wave * 2 + 0) * WROW + ol + e] += s1[e];
                            st[(wave * 2 + 1) * WROW + ol + e] += s2[e];
                        }
                    }
                }
            }
        }
    }

    if (p.stats) {
        __syncthreads();
        const long long wg = (long long)by * (p.per_xcd > 0 ? p.nchunk : (int)gridDim.x) + bx;
        for (int i = tid; i < 2 * WROW; i += 256) {
            const int which = i / WROW, o = i - which * WROW;
            if (ob + o < p.Cout)
                p.stats[(wg * 2 + which) * p.Cout + ob + o] = st[(0 * 2 + which) * WROW + o] + st[(1 * 2 + which) * WROW + o] +
                                                          st[(2 * 2 + which) * WROW + o] + st[(3 * 2 + which) * WROW + o];
        }
    }
}

}  // namespace fgcn

using namespace fgcn;

template <int CI>
static void launch_spatial_x3(const SpatialP& p, hipStream_t s) {
    const size_t lds = (size_t)9 * 32 * AHB + (4 * 2 * 64 + 4 * 32 * TTS + 64) * sizeof(float);
    dim3 grid((unsigned)cdiv(p.T, p.t_chunk), (unsigned)p.B, (unsigned)cdiv(p.Cout, 64));
    SpatialP q = p;
    q.nchunk = (int)grid.x;
    q.ncol = (int)grid.z;
    q.per_xcd = 0;
    const long long total = (long long)grid.x * grid.y * grid.z;
    if (q.ncol > 1 && !(fgcn::tuning(5) & 16) && total < (1ll << 30)) {     // key 5 bit 4: the plain 3-D grid
        q.per_xcd = (int)cdiv(total, 8);
        grid = dim3((unsigned)(q.per_xcd * 8));
    }
    const bool str = fgcn::stream_out((long long)p.B * p.T * p.V * p.Cout * 4);
    if (fgcn::f16x2_products()) {
        if (str) hipLaunchKernelGGL((spatial_fwd_x3_kernel<CI, 2, true>), grid, dim3(256), lds, s, q);
        else hipLaunchKernelGGL((spatial_fwd_x3_kernel<CI, 2>), grid, dim3(256), lds, s, q);
    } else {
        if (str) hipLaunchKernelGGL((spatial_fwd_x3_kernel<CI, 3, true>), grid, dim3(256), lds, s, q);
        else hipLaunchKernelGGL((spatial_fwd_x3_kernel<CI, 3>), grid, dim3(256), lds, s, q);
    }
}

static int spatial_t_chunk(int B, int T) {
    int chunk = 32;
    // (not below 8: the two-frames-per-wave kernel walks 8 frames per workgroup iteration; a 4-frame chunk left half its waves idle)
    while (chunk > 8 && (long long)B * cdiv(T, chunk) < 1024) chunk >>= 1;
    return chunk;
}

extern "C" int fgcn_spatial_tiles(int B, int T) { return (int)(B * cdiv(T, spatial_t_chunk(B, T))); }

template <int CI, int CO>
static void launch_spatial(const SpatialP& p, hipStream_t s) {
    const size_t lds = (((3 * 32 * AHS + 3) & ~3) + 4 * 2 * CO * 32 + 4 * 32 * TTS + CO * 32) * sizeof(float);
    dim3 grid((unsigned)cdiv(p.T, p.t_chunk), (unsigned)p.B, (unsigned)cdiv(p.Cout, CO * 32));
    static bool lds_opt_in = false;  // once per instantiation (not a stream operation: keep it out of graph captures)
    if (!lds_opt_in && lds > 48 * 1024) {  // gfx950 has 160 KiB of LDS per CU; opt in beyond the default dynamic limit
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_fwd_kernel<CI, CO, 0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_fwd_kernel<CI, CO, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_fwd_kernel<CI, CO, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        lds_opt_in = true;
    }
    const int mm = fgcn::math_mode();
    if (mm == FGCN_MATH_BF16X3 && p.Cin % 32 == 0)      // (narrower inputs: the exact f32 kernel, f32 pack_k4 weights)
        hipLaunchKernelGGL((spatial_fwd_kernel<CI, CO, 2>), grid, dim3(256), lds, s, p);
    else if (mm == FGCN_MATH_BF16)
        hipLaunchKernelGGL((spatial_fwd_kernel<CI, CO, 1>), grid, dim3(256), lds, s, p);
    else
        hipLaunchKernelGGL((spatial_fwd_kernel<CI, CO, 0>), grid, dim3(256), lds, s, p);
}

template <int CI>
static int dispatch_out(int co, const SpatialP& p, hipStream_t s) {
    switch (co) {
        case 1: launch_spatial<CI, 1>(p, s); return 0;
        case 2: launch_spatial<CI, 2>(p, s); return 0;
        case 4: launch_spatial<CI, 4>(p, s); return 0;
        case 8:
            if (fgcn::math_mode() == FGCN_MATH_BF16X3 && p.Cin % 32 == 0) launch_spatial<CI, 4>(p, s);   // two column blocks
            else launch_spatial<CI, 8>(p, s);
            return 0;
    }
    return -1;
}

extern "C" int fgcn_spatial_fwd(const float* x, const float* a_hat, const float* wd, const float* bias_sum, float* y,
                                float* stat_partials, int B, int T, int V, int Cin, int Cout, int ld_x, int ld_y,
                                int n_subsets, int a_hat_batched, void* stream) {
    FGCN_REQUIRE(x && a_hat && wd && y, FGCN_E_BADARG, "spatial_fwd: null pointer");
    FGCN_REQUIRE(B > 0 && B <= 65535 && T > 0 && V > 0 && V <= FGCN_MAX_V && Cin > 0 && Cout > 0, FGCN_E_BADARG,
                 "spatial_fwd: bad sizes B=%d T=%d V=%d Cin=%d Cout=%d", B, T, V, Cin, Cout);
    FGCN_REQUIRE(n_subsets >= 1 && n_subsets <= 3, FGCN_E_BADARG, "spatial_fwd: n_subsets=%d (1..3)", n_subsets);
    FGCN_REQUIRE(Cin % 4 == 0, FGCN_E_ALIGN, "spatial_fwd: Cin must be a multiple of 4 (pad the 3-channel input), got %d", Cin);
    FGCN_REQUIRE(Cout % 4 == 0 && ld_y % 4 == 0 && ld_y >= Cout && ld_x >= Cin, FGCN_E_ALIGN,
                 "spatial_fwd: Cout and ld_y must be multiples of 4, strides must cover the channels");
    FGCN_REQUIRE(aligned16(y) && aligned16(wd) && (!bias_sum || aligned16(bias_sum)), FGCN_E_ALIGN,
                 "spatial_fwd: 16-byte alignment");
    auto tiles = [](int c) { int t = (c + 31) / 32; return t <= 1 ? 1 : t <= 2 ? 2 : t <= 4 ? 4 : t <= 8 ? 8 : -1; };
    const int ci = tiles(Cin), co = tiles(Cout);
    FGCN_REQUIRE(ci > 0 && co > 0, FGCN_E_BADARG, "spatial_fwd: at most 256 channels (Cin=%d Cout=%d)", Cin, Cout);
    const bool split = fgcn::math_mode() == FGCN_MATH_BF16X3 && Cin % 32 == 0;   // wd: fgcn_pack_split3(acc_order) form
    const long long x_bytes = (long long)B * T * V * ld_x * 4;
    const long long w_bytes = (long long)n_subsets * Cin * Cout * (split ? (fgcn::f16x2_products() ? 4 : 6) : 4);
    FGCN_REQUIRE(x_bytes < 0x7FFF0000ll, FGCN_E_BADARG, "spatial_fwd: x must be smaller than 2 GiB (32-bit buffer offsets)");
    FGCN_REQUIRE(aligned16(x) || true, FGCN_E_ALIGN, "spatial_fwd: alignment");
    SpatialP p{x, a_hat, wd, bias_sum, y, stat_partials, B, T, V, Cin, Cout, ld_x, ld_y, n_subsets, a_hat_batched,
               spatial_t_chunk(B, T), (unsigned)x_bytes, (unsigned)w_bytes,
               (unsigned)((long long)n_subsets * Cin * Cout * 2), 0, 0, 0};
    hipStream_t s = (hipStream_t)stream;
    int rc = -1;
    if (split && !(fgcn::tuning(7) & 1)) {     // two frames per wave, split-bf16 aggregation (tuning key 7 bit 0: the older form)
        switch (ci) {
            case 1: launch_spatial_x3<1>(p, s); break;
            case 2: launch_spatial_x3<2>(p, s); break;
            case 4: launch_spatial_x3<4>(p, s); break;
            case 8: launch_spatial_x3<8>(p, s); break;
        }
        return launch_status("spatial_fwd");
    }
    switch (ci) {
        case 1: rc = dispatch_out<1>(co, p, s); break;
        case 2: rc = dispatch_out<2>(co, p, s); break;
        case 4: rc = dispatch_out<4>(co, p, s); break;
        case 8: rc = dispatch_out<8>(co, p, s); break;
    }
    FGCN_REQUIRE(rc == 0, FGCN_E_BADARG, "spatial_fwd: unsupported tile combination");
    return launch_status("spatial_fwd");
}
