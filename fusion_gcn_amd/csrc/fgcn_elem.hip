// BatchNorm statistics / apply / backward and the residual + ReLU epilogues of the AGCN block, channels-last.
// All of these are pure HBM streams: 16-byte loads, channel = (flat index * 4) % C, per-channel vectors
// (mean, rstd, scale, shift) read through L1/L2.  Reductions produce per-tile partials that a second tiny
// kernel (fgcn_reduce_sum / bn_finalize) sums in a fixed order: results are bitwise reproducible.
#include "fgcn_common.hpp"

namespace fgcn {

constexpr int ELEM_ROWS_PER_TILE = 64;    // small tiles keep >= 8 workgroups per CU busy even at 8 clips per GPU
constexpr int ELEM_MAX_TILES = 1024;    // partial rows of the reduction kernels (summed by reduce_sum, 22 launches per step whose time follows the
                                        // row count): same-box step A/B 2048 -> 1024: 60.95 -> 60.66 ms; 512: 61.4-61.7 (the reduce kernels lose bandwidth)

// ---- BatchNorm finalize ------------------------------------------------------------------------------------
// block = 8 channels x 128 partial-groups (C/8 workgroups, 4 loads in flight per thread: the 7500-tile partial arrays
// of the headline shape were a 48 us serial walk with 32 groups on C/32 workgroups); double accumulation of the float
// tile sums, fixed summation order.
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* partials, int P, long long count,
                                                           const float* gamma, const float* beta, float* rmean,
                                                           float* rvar, float momentum, float eps, float* out, int C) {
    __shared__ double s1[128][9], s2[128][9];
    const int cl = threadIdx.x & 7, g = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    double a = 0.0, b = 0.0;
    if (c < C) {
        int i = g;
        for (; i + 896 < P; i += 1024) {                 // eight row groups (sixteen loads) in flight: the walk is latency-bound
            float x0[8], x1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0[u] = partials[((long long)(i + 128 * u) * 2 + 0) * C + c];
                x1[u] = partials[((long long)(i + 128 * u) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a += (double)x0[u];
                b += (double)x1[u];
            }
        }
        for (; i + 384 < P; i += 512) {
            float x0[4], x1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x0[u] = partials[((long long)(i + 128 * u) * 2 + 0) * C + c];
                x1[u] = partials[((long long)(i + 128 * u) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a += (double)x0[u];
                b += (double)x1[u];
            }
        }
        for (; i < P; i += 128) {
            a += (double)partials[((long long)i * 2 + 0) * C + c];
            b += (double)partials[((long long)i * 2 + 1) * C + c];
        }
    }
    s1[g][cl] = a;
    s2[g][cl] = b;
    __syncthreads();
    // the 128 group sums of a channel: a fixed tree (seven steps) instead of one thread's walk over 127 dependent LDS reads (~4 us of a ~19 us
    // launch that runs 26 times per step, and whose time does not shrink with the batch)
#ifndef FGCN_FINALIZE_TREE
#define FGCN_FINALIZE_TREE 1            // 0: the serial walk (A/B builds)
#endif
#if FGCN_FINALIZE_TREE
#pragma unroll
    for (int stride = 64; stride >= 1; stride >>= 1) {
        if (g < stride) {
            s1[g][cl] += s1[g + stride][cl];
            s2[g][cl] += s2[g + stride][cl];
        }
        __syncthreads();
    }
#endif
    if (g == 0 && c < C) {
#if FGCN_FINALIZE_TREE
        a = s1[0][cl];
        b = s2[0][cl];
#else
        for (int i = 1; i < 128; ++i) {
            a += s1[i][cl];
            b += s2[i][cl];
        }
#endif
        const double mean = a / (double)count;
        double var = b / (double)count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float scale = gamma[c] * rstd;
        out[0 * C + c] = (float)mean;
        out[1 * C + c] = rstd;
        out[2 * C + c] = scale;
        out[3 * C + c] = beta[c] - (float)mean * scale;
        if (rmean && rvar) {
            const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
        }
    }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rmean, const float* rvar,
                                      float eps, float* out, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float rstd = 1.f / sqrtf(rvar[c] + eps);
    const float scale = gamma[c] * rstd;
    out[0 * C + c] = rmean[c];
    out[1 * C + c] = rstd;
    out[2 * C + c] = scale;
    out[3 * C + c] = beta[c] - rmean[c] * scale;
}

// 16 bytes per lane, plain or non-temporal (`stream`: a kernel argument, fgcn_common.hpp stream_out -- large activations are streamed
// past L2: the backward apply pass -9 .. -12 % at the 64-clip step's sizes)
// ... and the same for an input that is read here for the last time in a long while (bn_act's operands: -15 % per launch with both;
// bn_act_bwd_apply re-reads what the reduce pass in front of it has just brought in and is better off with plain loads)
__device__ __forceinline__ f32x4 load4(const float* ptr, int stream) {
    return stream ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ptr)) : *reinterpret_cast<const f32x4*>(ptr);
}
__device__ __forceinline__ void store4(float* ptr, f32x4 val, int stream) {
    if (stream) __builtin_nontemporal_store(val, reinterpret_cast<f32x4*>(ptr));
    else *reinterpret_cast<f32x4*>(ptr) = val;
}
// four values -> four bfloat16 (round to nearest even: v_cvt_pk_bf16_f32, the same rounding the bf16 kernels apply when they stage an
// f32 tensor), one 8-byte store.  HALF-PRECISION STORAGE of a tensor that only bf16 MFMA staging reads (math mode bf16: G, the temporal
// conv's input, and dU, its output gradient): the consumer then copies the bytes instead of converting them, and moves half of them.
__device__ __forceinline__ void store4_bf16(unsigned short* ptr, f32x4 val, int stream) {
    const u32x2 h = __builtin_bit_cast(u32x2, pack_bf16(val));
    if (stream) __builtin_nontemporal_store(h, reinterpret_cast<u32x2*>(ptr));
    else *reinterpret_cast<u32x2*>(ptr) = h;
}
// four elements at element offset e of a tensor that is float32 or (half) bfloat16 -- the typed forms of the BatchNorm passes (template
// parameter TY, kernel argument hm: one bit per activation operand; math mode bf16 with half-precision activation storage, the `_t` entry
// points).  The flag is a kernel argument: a uniform branch, eight bytes per lane instead of sixteen.
__device__ __forceinline__ f32x4 ldx4(const float* p, long long e, bool half, int stream) {
    if (half) {
        const u32x2* q = reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(p) + e);
        return unpack_bf16x4(stream ? __builtin_nontemporal_load(q) : *q);
    }
    return load4(p + e, stream);
}
// ---- forward epilogue ----------------------------------------------------------------------------------------
// MASK: also writes the sign bits of the result (bit e%8 of byte e/8 = [out[e] > 0]) for the backward passes, which then
// read 1/32 of an activation instead of `out`; a lane pair shares a byte (n4 is even, host check).
// O16: `out` is a bfloat16 tensor (store4_bf16; ld_out == C); the sign image is that of the f32 values (a value that rounds to zero
// keeps its sign bit: the backward gates exactly as with f32 storage).
// TY: typed operands -- bit 0 of hm: `a` is bfloat16, bit 1: `b` is
template <int RES, bool MASK, bool O16 = false, bool TY = false>
__global__ __launch_bounds__(256) void bn_act_kernel(const float* a, const float* va, const float* b, const float* vb,
                                                     float* out, unsigned char* mask, long long n4, int C, int relu, int stream, int ld_out, int hm) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(((unsigned)i * 4u) % (unsigned)C);   // n4 < 2^30 (host check): 32-bit modulo
        const f32x4 x = TY ? ldx4(a, i * 4, hm & 1, stream) : load4(a + i * 4, stream);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(va + 2 * C + c);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(va + 3 * C + c);
        f32x4 y = x * sc + sh;
        if (RES == 1) {
            y += TY ? ldx4(b, i * 4, hm & 2, stream) : load4(b + i * 4, stream);
        } else if (RES == 2) {
            const f32x4 r = TY ? ldx4(b, i * 4, hm & 2, stream) : load4(b + i * 4, stream);
            y += r * *reinterpret_cast<const f32x4*>(vb + 2 * C + c) + *reinterpret_cast<const f32x4*>(vb + 3 * C + c);
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
        }
        // (ld_out != C: the result goes into a channel window of a wider tensor -- MS-G3D's branch concatenation)
        if constexpr (O16) store4_bf16(reinterpret_cast<unsigned short*>(out) + i * 4, y, stream);
        else store4(ld_out == C ? out + i * 4 : out + (i * 4 / C) * ld_out + c, y, stream);
        if (MASK) {
            const int nib = (y[0] > 0.f ? 1 : 0) | (y[1] > 0.f ? 2 : 0) | (y[2] > 0.f ? 4 : 0) | (y[3] > 0.f ? 8 : 0);
            const int other = __shfl_xor(nib, 1);
            if (!(threadIdx.x & 1)) mask[i >> 1] = (unsigned char)(nib | (other << 4));
        }
    }
}

// [out > 0] of the four elements at offset o (multiple of 4) from the sign-bit image or from `out` itself
template <bool MASKED>
__device__ __forceinline__ void relu_gate(f32x4& dp, const float* out, const unsigned char* mask, long long o) {
    if (MASKED) {
        const int nib = mask[o >> 3] >> (int)(o & 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) dp[e] = (nib >> e) & 1 ? dp[e] : 0.f;
    } else {
        const f32x4 y = *reinterpret_cast<const f32x4*>(out + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) dp[e] = y[e] > 0.f ? dp[e] : 0.f;
    }
}

// ---- backward pass 1: per-channel reductions ---------------------------------------------------------------
// blockDim = (C/4, ny): thread (x, y) owns channels 4x..4x+3 and rows y, y+ny, ... of its tile.
// TY: typed operands -- bit 0 of hm: `dout` is bfloat16, bit 1: `a`, bit 2: `b`
template <int RES, bool MASKED, bool TY = false>
__global__ void bn_act_bwd_reduce_kernel(const float* dout, const float* out, const unsigned char* mask, const float* a,
                                         const float* va, const float* b, const float* vb, float* partials,
                                         long long rows, long long rows_per_tile, int C, int relu, int ld_dout, int grp_rows, int hm) {
    extern __shared__ float red[];  // [ny][3][Cw], Cw = the channel window of this block: 4 * blockDim.x channels from c0
    const int Cw = blockDim.x * 4, c0 = blockIdx.y * Cw, cl = threadIdx.x * 4;
    const int c = c0 + cl;
    const bool cok = c < C;                              // (C % 4 == 0: a quad is all-in or all-out)
    const long long r0 = (long long)blockIdx.x * rows_per_tile;
    const long long r1 = min(r0 + rows_per_tile, rows);
    f32x4 mean_a = {0.f, 0.f, 0.f, 0.f}, rstd_a = mean_a, mean_b = mean_a, rstd_b = mean_a;
    if (cok) {
        mean_a = *reinterpret_cast<const f32x4*>(va + c);
        rstd_a = *reinterpret_cast<const f32x4*>(va + C + c);
        if (RES == 2) {
            mean_b = *reinterpret_cast<const f32x4*>(vb + c);
            rstd_b = *reinterpret_cast<const f32x4*>(vb + C + c);
        }
    }
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, s3 = s1;
    if (cok) {
        // (ld_dout > C: a channel window of a wider gradient; grp_rows > 0: dout is one row per GROUP of grp_rows consecutive rows -- the
        // gradient of the pooled output of the model's last block, fgcn_bn_act_pool -- and every row of a group reads its group's row)
        auto ld_d = [&](long long r) {
            const long long dr = grp_rows ? (long long)((unsigned)r / (unsigned)grp_rows) : r;
            return TY ? ldx4(dout, dr * ld_dout + c, hm & 1, 0) : *reinterpret_cast<const f32x4*>(dout + dr * ld_dout + c);
        };
        auto ld_a = [&](long long o) { return TY ? ldx4(a, o, hm & 2, 0) : *reinterpret_cast<const f32x4*>(a + o); };
        auto ld_b = [&](long long o) { return TY ? ldx4(b, o, hm & 4, 0) : *reinterpret_cast<const f32x4*>(b + o); };
        // the gate of the four elements at offset o: the sign image's nibble, or the four values of `out` themselves
        auto ld_g = [&](long long o) -> f32x4 {
            if (MASKED) return f32x4{__builtin_bit_cast(float, (int)(mask[o >> 3] >> (int)(o & 4))), 0.f, 0.f, 0.f};
            return *reinterpret_cast<const f32x4*>(out + o);
        };
        auto add = [&](f32x4 dp, f32x4 gv, f32x4 av, f32x4 bv) {
            if (relu) {
                if (MASKED) {
                    const float g0 = gv[0];
                    const int nib = __builtin_bit_cast(int, g0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dp[e] = (nib >> e) & 1 ? dp[e] : 0.f;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dp[e] = gv[e] > 0.f ? dp[e] : 0.f;
                }
            }
            const f32x4 ah = (av - mean_a) * rstd_a;
            s1 += dp;
            s2 += dp * ah;
            if (RES == 2) s3 += dp * ((bv - mean_b) * rstd_b);
        };
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        // four rows in flight (two with a second BatchNorm branch; all their loads first: the loop was one memory round trip per row); the sums
        // still run over the thread's rows in ascending order -- same bits as one by one
        constexpr int UNR = RES == 2 ? 2 : 4;                // (inside the 128 registers that four workgroups per CU leave a thread)
        const long long ny = blockDim.y;
        long long r = r0 + threadIdx.y;
        for (; r + (UNR - 1) * ny < r1; r += UNR * ny) {
            f32x4 dpv[UNR], gvv[UNR], avv[UNR], bvv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long long o = (r + u * ny) * C + c;
                dpv[u] = ld_d(r + u * ny);
                gvv[u] = relu ? ld_g(o) : z4;
                avv[u] = ld_a(o);
                bvv[u] = RES == 2 ? ld_b(o) : z4;
            }
            __builtin_amdgcn_sched_barrier(0);               // (the loads of all rows first: see the eight-wide kernel)
#pragma unroll
            for (int u = 0; u < UNR; ++u) add(dpv[u], gvv[u], avv[u], bvv[u]);
        }
        for (; r < r1; r += ny) {
            const long long o = r * C + c;
            add(ld_d(r), relu ? ld_g(o) : z4, ld_a(o), RES == 2 ? ld_b(o) : z4);
        }
    }
    float* mine = red + (long long)threadIdx.y * 3 * Cw;
    *reinterpret_cast<f32x4*>(mine + cl) = s1;
    *reinterpret_cast<f32x4*>(mine + Cw + cl) = s2;
    *reinterpret_cast<f32x4*>(mine + 2 * Cw + cl) = s3;
    __syncthreads();
    const int nthreads = blockDim.x * blockDim.y;
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    for (int i = t; i < 3 * Cw; i += nthreads) {
        const int which = i / Cw, ci = i - which * Cw;
        float s = 0.f;
        for (int y = 0; y < (int)blockDim.y; ++y) s += red[y * 3 * Cw + i];
        if (c0 + ci < C) partials[(long long)blockIdx.x * 3 * C + which * C + c0 + ci] = s;
    }
}

// ---- backward pass 2: apply -----------------------------------------------------------------------------------
// O16: `da` is a bfloat16 tensor (the gradient of the temporal conv's output, read only by bf16 MFMA staging)
// TY: typed operands -- bit 0 of hm: `dout` is bfloat16, bit 1: `a`, bit 2: `b` (da: O16; db stays float32)
template <int RES, bool MASKED, bool O16 = false, bool TY = false>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const float* dout, const float* out,
                                                               const unsigned char* mask, const float* a,
                                                               const float* va, const float* b, const float* vb,
                                                               const float* sums, float* da, float* db, long long n4,
                                                               int C, int relu, int train, float inv_m,
                                                               int db_accumulate, int stream, int ld_dout, int grp_rows, int hm) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(((unsigned)i * 4u) % (unsigned)C);   // n4 < 2^30 (host check): 32-bit modulo
        f32x4 dp;
        if (grp_rows) dp = *reinterpret_cast<const f32x4*>(dout + (long long)(((unsigned)i * 4u / (unsigned)C) / (unsigned)grp_rows) * ld_dout + c);   // (one row per group: float32)
        else if (TY) dp = ldx4(dout, ld_dout == C ? i * 4 : (i * 4 / C) * ld_dout + c, hm & 1, 0);
        else dp = *reinterpret_cast<const f32x4*>(ld_dout == C ? dout + i * 4 : dout + (i * 4 / C) * ld_dout + c);
        if (relu) relu_gate<MASKED>(dp, out, mask, i * 4);
        const f32x4 sc_a = *reinterpret_cast<const f32x4*>(va + 2 * C + c);
        f32x4 ga = dp;
        if (train) {
            const f32x4 ah = ((TY ? ldx4(a, i * 4, hm & 2, 0) : *reinterpret_cast<const f32x4*>(a + i * 4)) - *reinterpret_cast<const f32x4*>(va + c)) *
                             *reinterpret_cast<const f32x4*>(va + C + c);
            ga = dp - *reinterpret_cast<const f32x4*>(sums + c) * inv_m -
                 ah * (*reinterpret_cast<const f32x4*>(sums + C + c) * inv_m);
        }
        if constexpr (O16) store4_bf16(reinterpret_cast<unsigned short*>(da) + i * 4, ga * sc_a, stream);
        else store4(da + i * 4, ga * sc_a, stream);
        if (RES != 0 && db) {
            f32x4 gb = dp;
            if (RES == 2) {
                const f32x4 sc_b = *reinterpret_cast<const f32x4*>(vb + 2 * C + c);
                if (train) {
                    const f32x4 bh =
                        ((TY ? ldx4(b, i * 4, hm & 4, 0) : *reinterpret_cast<const f32x4*>(b + i * 4)) - *reinterpret_cast<const f32x4*>(vb + c)) *
                        *reinterpret_cast<const f32x4*>(vb + C + c);
                    gb = dp - *reinterpret_cast<const f32x4*>(sums + c) * inv_m -
                         bh * (*reinterpret_cast<const f32x4*>(sums + 2 * C + c) * inv_m);
                }
                gb = gb * sc_b;
            }
            if (db_accumulate) gb += *reinterpret_cast<const f32x4*>(db + i * 4);
            store4(db + i * 4, gb, db_accumulate ? 0 : stream);
        }
    }
}

// ---- the last block's epilogue with the global average pooling behind it ---------------------------------------------------
// out = relu(BN(a) + r) of the LAST block is read once, by the pooling in front of the classifier (agcn.py:196-197): here the pass
// that would write it sums it per (group, channel) instead and writes only the sign image the backward's ReLU gate reads -- one
// activation write and one activation read less per step.  grid = (splits, groups, channel windows of 4 blockDim.x), block =
// (C/4 up to 64, ny): thread (x, y) owns channels 4x .. 4x+3 and rows y, y + ny, ... of its split; fixed-order sums.
// TY: typed operands -- bit 0 of hm: `a` is bfloat16, bit 1: `b`
template <int RES, bool TY = false>
__global__ __launch_bounds__(256) void bn_act_pool_kernel(const float* a, const float* va, const float* b, const float* vb,
                                                         unsigned char* mask, float* partial, int grp_rows, int per, int C, int hm) {
    extern __shared__ float red[];                       // [ny][Cw]
    const int Cw = blockDim.x * 4, c = blockIdx.z * Cw + threadIdx.x * 4;
    const bool cok = c < C;                              // (C % 8 == 0: a quad -- and its lane pair -- is all-in or all-out)
    const int g = blockIdx.y, r0 = blockIdx.x * per, r1 = min(r0 + per, grp_rows);
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc, sc_b = sc, sh_b = sc, sum = sc;
    if (cok) {
        sc = *reinterpret_cast<const f32x4*>(va + 2 * C + c);
        sh = *reinterpret_cast<const f32x4*>(va + 3 * C + c);
        if (RES == 2) {
            sc_b = *reinterpret_cast<const f32x4*>(vb + 2 * C + c);
            sh_b = *reinterpret_cast<const f32x4*>(vb + 3 * C + c);
        }
    }
    auto one = [&](f32x4 x, f32x4 r, long long o) {
        f32x4 y = x * sc + sh;
        if (RES == 1) y += r;
        else if (RES == 2) y += r * sc_b + sh_b;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
        sum += y;
        const int nib = (y[0] > 0.f ? 1 : 0) | (y[1] > 0.f ? 2 : 0) | (y[2] > 0.f ? 4 : 0) | (y[3] > 0.f ? 8 : 0);
        const int other = __shfl_xor(nib, 1);
        if (!(threadIdx.x & 1)) mask[o >> 3] = (unsigned char)(nib | (other << 4));
    };
    if (cok) {                                           // (uniform per lane pair)
        const long long base = (long long)g * grp_rows;
        int r = r0 + threadIdx.y;
        const int ny = blockDim.y;
        for (; r + 3 * ny < r1; r += 4 * ny) {           // four rows (eight 16-byte loads) in flight
            f32x4 x[4], q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long o = (base + r + u * ny) * C + c;
                x[u] = TY ? ldx4(a, o, hm & 1, 1) : __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a + o));
                q[u] = RES != 0 ? (TY ? ldx4(b, o, hm & 2, 1) : __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(b + o))) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) one(x[u], q[u], (base + r + u * ny) * C + c);
        }
        for (; r < r1; r += ny) {
            const long long o = (base + r) * C + c;
            one(TY ? ldx4(a, o, hm & 1, 0) : *reinterpret_cast<const f32x4*>(a + o),
                RES != 0 ? (TY ? ldx4(b, o, hm & 2, 0) : *reinterpret_cast<const f32x4*>(b + o)) : f32x4{0.f, 0.f, 0.f, 0.f}, o);
        }
    }
    *reinterpret_cast<f32x4*>(red + threadIdx.y * Cw + threadIdx.x * 4) = sum;
    __syncthreads();
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    for (int i = t; i < Cw; i += blockDim.x * blockDim.y) {
        float s = 0.f;
        for (int y = 0; y < (int)blockDim.y; ++y) s += red[y * Cw + i];
        if (blockIdx.z * Cw + i < C) partial[((long long)g * gridDim.x + blockIdx.x) * C + blockIdx.z * Cw + i] = s;
    }
}

// ---- the typed passes, EIGHT elements per thread (C % 8 == 0) -----------------------------------------------------------------------------
// With bfloat16 operands a four-element thread moves 8 bytes per operand and the passes stop following the bytes (measured in the bf16 step,
// profiles/r06_ab_bf16_half_activations.txt: bn_act_bwd_apply 104.6 -> 95.5 us for 0.61 -> 0.37 GB): the thread count, not the traffic, bounds
// them.  Here a thread owns eight consecutive channels: one 16-byte load per bfloat16 operand (two per float32 one), one 16-byte store of
// eight bfloat16, one byte of the sign image.  Same arithmetic per element as the four-wide kernels: same bits.
struct f32x8 {
    f32x4 lo, hi;
};
__device__ __forceinline__ f32x8 ldx8(const float* p, long long e, bool half, int stream) {
    f32x8 r;
    if (half) {
        const u32x4v* q = reinterpret_cast<const u32x4v*>(reinterpret_cast<const unsigned short*>(p) + e);
        const u32x4v h = stream ? __builtin_nontemporal_load(q) : *q;
        r.lo = unpack_bf16x4(u32x2{h[0], h[1]});
        r.hi = unpack_bf16x4(u32x2{h[2], h[3]});
    } else {
        r.lo = load4(p + e, stream);
        r.hi = load4(p + e + 4, stream);
    }
    return r;
}
__device__ __forceinline__ void stx8(float* p, long long e, f32x8 v, bool half, int stream) {
    if (half) {
        const u32x2 a = __builtin_bit_cast(u32x2, pack_bf16(v.lo)), b = __builtin_bit_cast(u32x2, pack_bf16(v.hi));
        const u32x4v h = u32x4v{a[0], a[1], b[0], b[1]};
        u32x4v* q = reinterpret_cast<u32x4v*>(reinterpret_cast<unsigned short*>(p) + e);
        if (stream) __builtin_nontemporal_store(h, q);
        else *q = h;
    } else {
        store4(p + e, v.lo, stream);
        store4(p + e + 4, v.hi, stream);
    }
}
__device__ __forceinline__ f32x8 ldv8(const float* v, int c) { return f32x8{*reinterpret_cast<const f32x4*>(v + c), *reinterpret_cast<const f32x4*>(v + c + 4)}; }
__device__ __forceinline__ int sign_byte(f32x8 y) {
    return (y.lo[0] > 0.f ? 1 : 0) | (y.lo[1] > 0.f ? 2 : 0) | (y.lo[2] > 0.f ? 4 : 0) | (y.lo[3] > 0.f ? 8 : 0) | (y.hi[0] > 0.f ? 16 : 0) |
           (y.hi[1] > 0.f ? 32 : 0) | (y.hi[2] > 0.f ? 64 : 0) | (y.hi[3] > 0.f ? 128 : 0);
}
__device__ __forceinline__ void gate8(f32x8& d, int bits) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        d.lo[e] = (bits >> e) & 1 ? d.lo[e] : 0.f;
        d.hi[e] = (bits >> (4 + e)) & 1 ? d.hi[e] : 0.f;
    }
}

// The storage types are TEMPLATE parameters here: behind a run-time flag every load sits in a branch of its own with its conversion, and the
// compiler waits for each before it requests the next (the row-unrolled reduce below gained nothing until the flags were compile time).
// HM: bit 0 = `a` is bfloat16, bit 1 = `b`, bit 2 = `out`
template <int RES, bool MASK, int HM>
__global__ __launch_bounds__(256) void bn_act8_kernel(const float* a, const float* va, const float* b, const float* vb, float* out,
                                                      unsigned char* mask, long long n8, int C, int relu, int stream) {
    constexpr int hm = HM;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(((unsigned)i * 8u) % (unsigned)C);       // n8 < 2^29 (host check)
        const f32x8 x = ldx8(a, i * 8, hm & 1, stream), sc = ldv8(va + 2 * C, c), sh = ldv8(va + 3 * C, c);
        f32x8 y{x.lo * sc.lo + sh.lo, x.hi * sc.hi + sh.hi};
        if (RES == 1) {
            const f32x8 r = ldx8(b, i * 8, hm & 2, stream);
            y.lo += r.lo;
            y.hi += r.hi;
        } else if (RES == 2) {
            const f32x8 r = ldx8(b, i * 8, hm & 2, stream), sb = ldv8(vb + 2 * C, c), hb = ldv8(vb + 3 * C, c);
            y.lo += r.lo * sb.lo + hb.lo;
            y.hi += r.hi * sb.hi + hb.hi;
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y.lo[e] = fmaxf(y.lo[e], 0.f);
                y.hi[e] = fmaxf(y.hi[e], 0.f);
            }
        }
        stx8(out, i * 8, y, hm & 4, stream);
        if (MASK) mask[i] = (unsigned char)sign_byte(y);
    }
}

// HM: bit 0 = `dout` is bfloat16, bit 1 = `a`, bit 2 = `b`; the ReLU gate is the sign image (or none); blockDim = (C/8 up to 128, ny)
template <int RES, int HM>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce8_kernel(const float* dout, const unsigned char* mask, const float* a, const float* va, const float* b,
                                          const float* vb, float* partials, long long rows, long long rows_per_tile, int C, int relu,
                                          int grp_rows) {
    constexpr int hm = HM;
    extern __shared__ float red[];  // [ny][3][Cw]
    const int Cw = blockDim.x * 8, c0 = blockIdx.y * Cw, cl = threadIdx.x * 8;
    const int c = c0 + cl;
    const bool cok = c < C;
    const long long r0 = (long long)blockIdx.x * rows_per_tile;
    const long long r1 = min(r0 + rows_per_tile, rows);
    f32x8 zero{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x8 mean_a = zero, rstd_a = zero, mean_b = zero, rstd_b = zero, s1 = zero, s2 = zero, s3 = zero;
    if (cok) {
        mean_a = ldv8(va, c);
        rstd_a = ldv8(va + C, c);
        if (RES == 2) {
            mean_b = ldv8(vb, c);
            rstd_b = ldv8(vb + C, c);
        }
        // without a ReLU there is no sign image: the byte is then read from `a` (any valid memory) and ignored
        const unsigned char* mp = relu ? mask : reinterpret_cast<const unsigned char*>(a);
        // several rows in flight (all their loads first); the sums still run over the thread's rows in ascending order: same bits as one by one
        auto add = [&](f32x8 dp, int bits, f32x8 av, f32x8 bv) {
            gate8(dp, relu ? bits : 0xff);               // (a select, not a branch: the sign-byte loads stay with the others)
            const f32x4 ahl = (av.lo - mean_a.lo) * rstd_a.lo, ahh = (av.hi - mean_a.hi) * rstd_a.hi;
            s1.lo += dp.lo;
            s1.hi += dp.hi;
            s2.lo += dp.lo * ahl;
            s2.hi += dp.hi * ahh;
            if (RES == 2) {
                s3.lo += dp.lo * ((bv.lo - mean_b.lo) * rstd_b.lo);
                s3.hi += dp.hi * ((bv.hi - mean_b.hi) * rstd_b.hi);
            }
        };
        auto ld_d = [&](long long r, long long o) {
            // (one float32 row per group -- never together with a bfloat16 dout, host check: an offset select, not a branch around the load)
            if constexpr ((HM & 1) != 0) return ldx8(dout, o, true, 0);
            else return ldx8(dout, grp_rows ? (long long)((unsigned)r / (unsigned)grp_rows) * C + c : o, false, 0);
        };
        // (at most 256 threads per block: the launch bound lets the compiler keep four rows of sixteen-byte loads in registers -- with 128-thread
        // blocks, four per CU, the kernel had half the bytes in flight that the HBM latency asks for: 0.26 GB in 98.7 us)
        constexpr int UNR = RES == 2 ? 2 : 4;
        const long long ny = blockDim.y;
        long long r = r0 + threadIdx.y;
        for (; r + (UNR - 1) * ny < r1; r += UNR * ny) {
            f32x8 dpv[UNR], avv[UNR], bvv[UNR];
            int bits[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const long long o = (r + u * ny) * C + c;
                dpv[u] = ld_d(r + u * ny, o);
                bits[u] = mp[o >> 3];                    // (unconditional: a load behind `if (relu)` would sit in a branch of its own)
                avv[u] = ldx8(a, o, hm & 2, 0);
                bvv[u] = RES == 2 ? ldx8(b, o, hm & 4, 0) : zero;
            }
            __builtin_amdgcn_sched_barrier(0);               // (left alone the scheduler interleaves the rows' loads with the sums: one row in flight)
#pragma unroll
            for (int u = 0; u < UNR; ++u) add(dpv[u], bits[u], avv[u], bvv[u]);
        }
        for (; r < r1; r += ny) {
            const long long o = r * C + c;
            add(ld_d(r, o), mp[o >> 3], ldx8(a, o, hm & 2, 0), RES == 2 ? ldx8(b, o, hm & 4, 0) : zero);
        }
    }
    float* mine = red + (long long)threadIdx.y * 3 * Cw;
    *reinterpret_cast<f32x4*>(mine + cl) = s1.lo;
    *reinterpret_cast<f32x4*>(mine + cl + 4) = s1.hi;
    *reinterpret_cast<f32x4*>(mine + Cw + cl) = s2.lo;
    *reinterpret_cast<f32x4*>(mine + Cw + cl + 4) = s2.hi;
    *reinterpret_cast<f32x4*>(mine + 2 * Cw + cl) = s3.lo;
    *reinterpret_cast<f32x4*>(mine + 2 * Cw + cl + 4) = s3.hi;
    __syncthreads();
    const int nthreads = blockDim.x * blockDim.y;
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    for (int i = t; i < 3 * Cw; i += nthreads) {
        const int which = i / Cw, ci = i - which * Cw;
        float s = 0.f;
        for (int y = 0; y < (int)blockDim.y; ++y) s += red[y * 3 * Cw + i];
        if (c0 + ci < C) partials[(long long)blockIdx.x * 3 * C + which * C + c0 + ci] = s;
    }
}

// HM: bit 0 = `dout` is bfloat16, bit 1 = `a`, bit 2 = `b`, bit 3 = `da`; db follows `da` when db_half (a kernel argument: it selects a store, no load)
// TRAIN (compile time) and the unconditional sign-image load keep every load of a thread in ONE basic block: with `if (relu)` / `if (train)`
// around them the kernel made three memory round trips per thread, one after the other (dout; the sign byte; a and the vectors)
template <int RES, int HM, bool TRAIN>
__global__ __launch_bounds__(256) void bn_act_bwd_apply8_kernel(const float* dout, const unsigned char* mask, const float* a, const float* va,
                                                                const float* b, const float* vb, const float* sums, float* da, float* db,
                                                                long long n8, int C, int relu, float inv_m, int db_accumulate,
                                                                int stream, int grp_rows, int db_half) {
    constexpr int hm = HM;
    constexpr bool train = TRAIN;
    const unsigned char* mp = relu ? mask : reinterpret_cast<const unsigned char*>(dout);     // (no ReLU: any valid byte, ignored)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(((unsigned)i * 8u) % (unsigned)C);
        f32x8 dp;
        if constexpr ((HM & 1) != 0) dp = ldx8(dout, i * 8, true, 0);
        else dp = ldx8(dout, grp_rows ? (long long)(((unsigned)i * 8u / (unsigned)C) / (unsigned)grp_rows) * C + c : i * 8, false, 0);   // (a select, not a branch)
        const int bits = mp[i];
        f32x8 av{dp.lo, dp.hi};
        if constexpr (TRAIN) av = ldx8(a, i * 8, hm & 2, 0);
        gate8(dp, relu ? bits : 0xff);                   // (a select: behind `if (relu)` the compiler sinks the sign-byte load into the branch)
        const f32x8 sc_a = ldv8(va + 2 * C, c);
        f32x8 ga = dp;
        f32x8 s0, s1;
        if constexpr (TRAIN) {
            const f32x8 mean = ldv8(va, c), rstd = ldv8(va + C, c);
            s0 = ldv8(sums, c);
            s1 = ldv8(sums + C, c);
            ga.lo = dp.lo - s0.lo * inv_m - ((av.lo - mean.lo) * rstd.lo) * (s1.lo * inv_m);
            ga.hi = dp.hi - s0.hi * inv_m - ((av.hi - mean.hi) * rstd.hi) * (s1.hi * inv_m);
        }
        stx8(da, i * 8, f32x8{ga.lo * sc_a.lo, ga.hi * sc_a.hi}, hm & 8, stream);
        if (RES != 0 && db) {
            f32x8 gb = dp;
            if (RES == 2) {
                const f32x8 sc_b = ldv8(vb + 2 * C, c);
                if (train) {
                    const f32x8 bv = ldx8(b, i * 8, hm & 4, 0), mean = ldv8(vb, c), rstd = ldv8(vb + C, c), s2 = ldv8(sums + 2 * C, c);
                    gb.lo = dp.lo - s0.lo * inv_m - ((bv.lo - mean.lo) * rstd.lo) * (s2.lo * inv_m);
                    gb.hi = dp.hi - s0.hi * inv_m - ((bv.hi - mean.hi) * rstd.hi) * (s2.hi * inv_m);
                }
                gb.lo = gb.lo * sc_b.lo;
                gb.hi = gb.hi * sc_b.hi;
            }
            if (db_accumulate) {
                gb.lo += *reinterpret_cast<const f32x4*>(db + i * 8);
                gb.hi += *reinterpret_cast<const f32x4*>(db + i * 8 + 4);
            }
            stx8(db, i * 8, gb, db_half != 0, db_accumulate ? 0 : stream);       // (db_half: never with db_accumulate, host check)
        }
    }
}

// ---- column sums ------------------------------------------------------------------------------------------------
__global__ void col_sum_kernel(const float* x, float* partials, long long rows, long long rows_per_tile, int C, int ld) {
    extern __shared__ float red[];  // [ny][C4*4]
    const int c = threadIdx.x * 4;
    const int CP = blockDim.x * 4;
    const long long r0 = (long long)blockIdx.x * rows_per_tile;
    const long long r1 = min(r0 + rows_per_tile, rows);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (long long r = r0 + threadIdx.y; r < r1; r += blockDim.y) s += *reinterpret_cast<const f32x4*>(x + r * ld + c);
    *reinterpret_cast<f32x4*>(red + (long long)threadIdx.y * CP + c) = s;
    __syncthreads();
    const int nthreads = blockDim.x * blockDim.y;
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    for (int i = t; i < C; i += nthreads) {
        float acc = 0.f;
        for (int y = 0; y < (int)blockDim.y; ++y) acc += red[y * CP + i];
        partials[(long long)blockIdx.x * C + i] = acc;
    }
}

// ---- global average pooling: out[g][c] = mean over the R rows of group g (agcn.py:196-197, x.mean(3).mean(1)) -----------
// Two fixed-order stages, no atomics / semaphores (torch's multi-block mean did not survive HIP-graph replay inside the
// captured step: wrong pooled features from the second replay on at 8 clips).
__global__ __launch_bounds__(1024) void group_sum_kernel(const float* x, float* partial, int rows, int C, int ld, int splits) {
    __shared__ float red[16][65];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int g = blockIdx.y, sp = blockIdx.z;
    const int c = blockIdx.x * 64 + tx;
    const int per = (rows + splits - 1) / splits;
    const int r0 = sp * per, r1 = min(r0 + per, rows);
    float s = 0.f;
    if (c < C)
        for (int r = r0 + ty; r < r1; r += 16) s += x[((long long)g * rows + r) * ld + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += red[k][tx];
        partial[((long long)g * splits + sp) * C + c] = a;
    }
}

__global__ void group_mean_finish_kernel(const float* partial, float* out, int groups, int C, int splits, float inv_rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * C) return;
    const int g = i / C, c = i - g * C;
    float a = 0.f;
    for (int sp = 0; sp < splits; ++sp) a += partial[((long long)g * splits + sp) * C + c];
    out[i] = a * inv_rows;
}

}  // namespace fgcn

using namespace fgcn;

extern "C" int fgcn_elem_tiles(long long rows) {
    long long t = cdiv(rows, ELEM_ROWS_PER_TILE);
    return (int)(t < ELEM_MAX_TILES ? t : ELEM_MAX_TILES);
}

static long long rows_per_tile_for(long long rows) { return cdiv(rows, fgcn_elem_tiles(rows)); }

// One 16-byte group per thread (the cap only bounds the grid): with 8192 blocks a thread of the 64-clip step walked seven groups, and every
// trip of that loop waits for its loads behind the previous trip's store (vmcnt counts in issue order) -- in-step, per kernel, same box
// (profiles/r03_ab_elem_blocks.txt): bn_act 1.80 -> 1.62 ms per step, bn_act_bwd_apply 2.72 -> 2.51 (identity residual), 1.62 -> 1.49 (down).
#ifndef FGCN_ELEM_BLOCKS
#define FGCN_ELEM_BLOCKS (1 << 22)
#endif
static unsigned stream_blocks(long long n4) {
    const long long b = cdiv(n4, 256);
    return (unsigned)(b < FGCN_ELEM_BLOCKS ? b : FGCN_ELEM_BLOCKS);
}

extern "C" int fgcn_bn_finalize(const float* partials, int n_partials, long long count, const float* gamma,
                                const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                float* out_vec, int C, void* stream) {
    FGCN_REQUIRE(partials && gamma && beta && out_vec && n_partials > 0 && count > 0 && C > 0, FGCN_E_BADARG,
                 "bn_finalize: bad argument");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)cdiv(C, 8)), dim3(1024), 0, (hipStream_t)stream, partials,
                       n_partials, count, gamma, beta, running_mean, running_var, momentum, eps, out_vec, C);
    return launch_status("bn_finalize");
}

extern "C" int fgcn_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, float* out_vec, int C, void* stream) {
    FGCN_REQUIRE(gamma && beta && running_mean && running_var && out_vec && C > 0, FGCN_E_BADARG,
                 "bn_eval_coeffs: bad argument");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((unsigned)cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, gamma,
                       beta, running_mean, running_var, eps, out_vec, C);
    return launch_status("bn_eval_coeffs");
}

static int check_elem(const char* what, long long rows, int C, int res_mode, const void* b, const void* vb) {
    FGCN_REQUIRE(rows > 0 && C > 0 && C % 4 == 0, FGCN_E_BADARG, "%s: rows=%lld C=%d (C must be a multiple of 4)", what,
                 rows, C);
    FGCN_REQUIRE(res_mode >= 0 && res_mode <= 2, FGCN_E_BADARG, "%s: res_mode=%d", what, res_mode);
    FGCN_REQUIRE(res_mode == 0 || b, FGCN_E_BADARG, "%s: residual operand missing", what);
    FGCN_REQUIRE(res_mode != 2 || vb, FGCN_E_BADARG, "%s: residual BatchNorm vector missing", what);
    return FGCN_OK;
}

static int bn_act_impl(const float* a, const float* vec_a, const float* b, const float* vec_b, float* out,
                       unsigned char* sign_mask, long long rows, int C, int res_mode, int relu, int ld_out, void* stream, bool o16 = false, int hm = 0) {
    FGCN_REQUIRE(a && vec_a && out, FGCN_E_BADARG, "bn_act: null pointer");
    FGCN_REQUIRE((!o16 && !hm) || fgcn::math_mode() == FGCN_MATH_BF16, FGCN_E_BADARG, "bn_act: bfloat16 tensors need math mode bf16");
    FGCN_REQUIRE(!o16 || ld_out == C, FGCN_E_BADARG, "bn_act: a bfloat16 output is contiguous (ld_out == C)");
    if (int e = check_elem("bn_act", rows, C, res_mode, b, vec_b)) return e;
    FGCN_REQUIRE(aligned16(a) && aligned16(out) && aligned16(vec_a) && (!b || aligned16(b)), FGCN_E_ALIGN,
                 "bn_act: 16-byte alignment");
    const long long n4 = rows * C / 4;
    FGCN_REQUIRE(n4 < (1ll << 30), FGCN_E_BADARG, "elementwise kernel: tensor too large (>= 2^32 elements)");
    hipStream_t s = (hipStream_t)stream;
    FGCN_REQUIRE(!sign_mask || n4 % 2 == 0, FGCN_E_BADARG, "bn_act: a sign mask needs rows*C to be a multiple of 8");
    dim3 g(stream_blocks(n4)), blk(256);
    const int str = fgcn::stream_out(n4 * 16) ? 1 : 0;
    if (hm && C % 8 == 0 && ld_out == C) {      // typed operands: eight elements per thread
        const long long n8 = n4 / 2;
        const int hm8 = hm | (o16 ? 4 : 0);
        dim3 g8(stream_blocks(n8));
#define FGCN_BN_ACT8H(RES_, HM_)                                                                                                     \
    do {                                                                                                                             \
        if (sign_mask) hipLaunchKernelGGL((bn_act8_kernel<RES_, true, HM_>), g8, blk, 0, s, a, vec_a, b, vec_b, out, sign_mask, n8, C, relu, str); \
        else hipLaunchKernelGGL((bn_act8_kernel<RES_, false, HM_>), g8, blk, 0, s, a, vec_a, b, vec_b, out, sign_mask, n8, C, relu, str);          \
    } while (0)
#define FGCN_BN_ACT8(RES_)                                                                                                           \
    do switch (RES_ == 0 ? (hm8 & ~2) : hm8) {                                                                                       \
        case 1: FGCN_BN_ACT8H(RES_, 1); break;                                                                                       \
        case 2: FGCN_BN_ACT8H(RES_, 2); break;                                                                                       \
        case 3: FGCN_BN_ACT8H(RES_, 3); break;                                                                                       \
        case 4: FGCN_BN_ACT8H(RES_, 4); break;                                                                                       \
        case 5: FGCN_BN_ACT8H(RES_, 5); break;                                                                                       \
        case 6: FGCN_BN_ACT8H(RES_, 6); break;                                                                                       \
        default: FGCN_BN_ACT8H(RES_, 7); break;                                                                                      \
    } while (0)
        if (res_mode == 0) FGCN_BN_ACT8(0);
        else if (res_mode == 1) FGCN_BN_ACT8(1);
        else FGCN_BN_ACT8(2);
#undef FGCN_BN_ACT8
#undef FGCN_BN_ACT8H
        return launch_status("bn_act");
    }
#define FGCN_BN_ACT4(RES_, M_, O_, TY_) \
    hipLaunchKernelGGL((bn_act_kernel<RES_, M_, O_, TY_>), g, blk, 0, s, a, vec_a, b, vec_b, out, sign_mask, n4, C, relu, str, ld_out, hm)
#define FGCN_BN_ACT3(RES_, M_, O_)                                                                                \
    do {                                                                                                          \
        if (hm) FGCN_BN_ACT4(RES_, M_, O_, true);                                                                 \
        else FGCN_BN_ACT4(RES_, M_, O_, false);                                                                   \
    } while (0)
#define FGCN_BN_ACT(RES_)                                                                                         \
    do {                                                                                                          \
        if (o16 && sign_mask) FGCN_BN_ACT3(RES_, true, true);                                                     \
        else if (o16) FGCN_BN_ACT3(RES_, false, true);                                                            \
        else if (sign_mask) FGCN_BN_ACT3(RES_, true, false);                                                      \
        else FGCN_BN_ACT3(RES_, false, false);                                                                    \
    } while (0)
    if (res_mode == 0) FGCN_BN_ACT(0);
    else if (res_mode == 1) FGCN_BN_ACT(1);
    else FGCN_BN_ACT(2);
#undef FGCN_BN_ACT
#undef FGCN_BN_ACT3
#undef FGCN_BN_ACT4
    return launch_status("bn_act");
}

static int reduce_block(int C, dim3* blk) {
    int cx = C / 4;
    if (cx < 1) return -1;
    if (cx > 256) cx = 256;                    // wider tensors: 1024-channel windows on gridDim.y
    int ny = 256 / cx;
    if (ny > 32) ny = 32;
    *blk = dim3((unsigned)cx, (unsigned)ny);
    return 0;
}

static int bn_act_bwd_reduce_impl(const float* dout, const float* out, const unsigned char* sign_mask,
                                  const float* a, const float* vec_a, const float* b, const float* vec_b,
                                  float* partials, int n_tiles, long long rows, int C, int res_mode, int relu,
                                  int ld_dout, void* stream, int grp_rows = 0, int hm = 0) {
    FGCN_REQUIRE(grp_rows >= 0 && (grp_rows == 0 || (rows % grp_rows == 0 && ld_dout == C)), FGCN_E_BADARG,
                 "bn_act_bwd_reduce: %lld rows are not whole groups of %d", rows, grp_rows);
    FGCN_REQUIRE(!hm || (fgcn::math_mode() == FGCN_MATH_BF16 && !((hm & 1) && grp_rows) && (!relu || sign_mask)), FGCN_E_BADARG,
                 "bn_act_bwd_reduce: bfloat16 tensors need math mode bf16, the sign image as the ReLU gate and a float32 per-group gradient");
    FGCN_REQUIRE(dout && a && vec_a && partials && (!relu || out || sign_mask), FGCN_E_BADARG,
                 "bn_act_bwd_reduce: null pointer");
    if (int e = check_elem("bn_act_bwd_reduce", rows, C, res_mode, b, vec_b)) return e;
    FGCN_REQUIRE(n_tiles == fgcn_elem_tiles(rows), FGCN_E_BADARG, "bn_act_bwd_reduce: n_tiles must be %d",
                 fgcn_elem_tiles(rows));
    dim3 blk;
    FGCN_REQUIRE(reduce_block(C, &blk) == 0, FGCN_E_BADARG, "bn_act_bwd_reduce: C=%d unsupported", C);
    const size_t lds = (size_t)blk.y * 3 * blk.x * 4 * sizeof(float) * ((hm && (fgcn::tuning(26) & 1)) ? 2 : 1);
    FGCN_REQUIRE(lds <= 64 * 1024, FGCN_E_BADARG, "bn_act_bwd_reduce: C=%d needs too much LDS", C);
    const long long rpt = rows_per_tile_for(rows);
    hipStream_t s = (hipStream_t)stream;
    dim3 g((unsigned)n_tiles, (unsigned)cdiv(C, (int)blk.x * 4));
    if (hm && C % 8 == 0 && ld_dout == C && blk.x % 2 == 0) {
        // typed operands: eight channels per thread, the same rows per thread as the four-wide kernel (same sums, bit for bit)
        const dim3 blk8(blk.x / 2, (fgcn::tuning(26) & 1) ? blk.y * 2 : blk.y);      // (key 26 bit 0, A/B: 256 threads -- other sums in the last bits)
#define FGCN_BN_RED8(RES_, HM_) \
    hipLaunchKernelGGL((bn_act_bwd_reduce8_kernel<RES_, HM_>), g, blk8, lds, s, dout, sign_mask, a, vec_a, b, vec_b, partials, rows, rpt, C, relu, grp_rows)
        if (res_mode == 2) {
            switch (hm) {
                case 1: FGCN_BN_RED8(2, 1); break;
                case 2: FGCN_BN_RED8(2, 2); break;
                case 3: FGCN_BN_RED8(2, 3); break;
                case 4: FGCN_BN_RED8(2, 4); break;
                case 5: FGCN_BN_RED8(2, 5); break;
                case 6: FGCN_BN_RED8(2, 6); break;
                default: FGCN_BN_RED8(2, 7); break;
            }
        } else {
            switch (hm & 3) {
                case 1: FGCN_BN_RED8(0, 1); break;
                case 2: FGCN_BN_RED8(0, 2); break;
                default: FGCN_BN_RED8(0, 3); break;
            }
        }
#undef FGCN_BN_RED8
        return launch_status("bn_act_bwd_reduce");
    }
#define FGCN_BN_RED3(RES_, M_, TY_)                                                                                \
    hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<RES_, M_, TY_>), g, blk, lds, s, dout, out, sign_mask, a, vec_a, b, vec_b, \
                       partials, rows, rpt, C, relu, ld_dout, grp_rows, hm)
#define FGCN_BN_RED(RES_, M_)                                                                                      \
    do {                                                                                                           \
        if (hm) FGCN_BN_RED3(RES_, M_, true);                                                                      \
        else FGCN_BN_RED3(RES_, M_, false);                                                                        \
    } while (0)
    if (res_mode == 2) {
        if (sign_mask) FGCN_BN_RED(2, true); else FGCN_BN_RED(2, false);
    } else {
        if (sign_mask) FGCN_BN_RED(0, true); else FGCN_BN_RED(0, false);
    }
#undef FGCN_BN_RED
#undef FGCN_BN_RED3
    return launch_status("bn_act_bwd_reduce");
}

static int bn_act_bwd_apply_impl(const float* dout, const float* out, const unsigned char* sign_mask,
                                 const float* a, const float* vec_a, const float* b, const float* vec_b,
                                 const float* sums, float* da, float* db,
                                 long long rows, int C, int res_mode, int relu, int train, int db_accumulate,
                                 int ld_dout, void* stream, int grp_rows = 0, bool o16 = false, int hm = 0, int db16 = 0) {
    FGCN_REQUIRE(grp_rows >= 0 && (grp_rows == 0 || (rows % grp_rows == 0 && ld_dout == C)), FGCN_E_BADARG,
                 "bn_act_bwd_apply: %lld rows are not whole groups of %d", rows, grp_rows);
    FGCN_REQUIRE((!hm && !o16) || (fgcn::math_mode() == FGCN_MATH_BF16 && !((hm & 1) && grp_rows) && (!hm || !relu || sign_mask)), FGCN_E_BADARG,
                 "bn_act_bwd_apply: bfloat16 tensors need math mode bf16, the sign image as the ReLU gate and a float32 per-group gradient");
    FGCN_REQUIRE(dout && vec_a && da && (!relu || out || sign_mask) && (!train || (a && sums)), FGCN_E_BADARG,
                 "bn_act_bwd_apply: null pointer");
    FGCN_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && res_mode >= 0 && res_mode <= 2, FGCN_E_BADARG,
                 "bn_act_bwd_apply: bad shape/res_mode");
    FGCN_REQUIRE(res_mode != 2 || !db || (vec_b && (!train || b)), FGCN_E_BADARG,
                 "bn_act_bwd_apply: residual BatchNorm inputs missing");
    const long long n4 = rows * C / 4;
    FGCN_REQUIRE(n4 < (1ll << 30), FGCN_E_BADARG, "elementwise kernel: tensor too large (>= 2^32 elements)");
    const float inv_m = 1.f / (float)rows;
    hipStream_t s = (hipStream_t)stream;
    dim3 g(stream_blocks(n4)), blk(256);
    const int str = fgcn::stream_out(n4 * 16) ? 1 : 0;
    FGCN_REQUIRE(!db16 || (hm && C % 8 == 0 && ld_dout == C && !db_accumulate && db && (!relu || sign_mask)), FGCN_E_BADARG,
                 "bn_act_bwd_apply_t: a bfloat16 db needs the eight-wide typed kernel (C %% 8 == 0) and no accumulation");
    if (hm && C % 8 == 0 && ld_dout == C && (!relu || sign_mask)) {      // typed operands: eight elements per thread
        const long long n8 = n4 / 2;
        const int hm8 = hm | (o16 ? 8 : 0);
        dim3 g8(stream_blocks(n8));
        const int rm = (res_mode == 0 || !db) ? 0 : res_mode;
#define FGCN_BN_APP8H(RES_, HM_)                                                                                                          \
    do {                                                                                                                                  \
        if (train) hipLaunchKernelGGL((bn_act_bwd_apply8_kernel<RES_, HM_, true>), g8, blk, 0, s, dout, sign_mask, a, vec_a, b, vec_b, sums, da, db, n8, C, \
                                      relu, inv_m, db_accumulate, str, grp_rows, db16);                                                   \
        else hipLaunchKernelGGL((bn_act_bwd_apply8_kernel<RES_, HM_, false>), g8, blk, 0, s, dout, sign_mask, a, vec_a, b, vec_b, sums, da, db, n8, C,      \
                                relu, inv_m, db_accumulate, str, grp_rows, db16);                                                         \
    } while (0)
#define FGCN_BN_APP8(RES_)                                                                                                                \
    do switch (RES_ == 2 ? hm8 : (hm8 & ~4)) {                                                                                            \
        case 1: FGCN_BN_APP8H(RES_, 1); break;                                                                                            \
        case 2: FGCN_BN_APP8H(RES_, 2); break;                                                                                            \
        case 3: FGCN_BN_APP8H(RES_, 3); break;                                                                                            \
        case 4: FGCN_BN_APP8H(RES_, 4); break;                                                                                            \
        case 5: FGCN_BN_APP8H(RES_, 5); break;                                                                                            \
        case 6: FGCN_BN_APP8H(RES_, 6); break;                                                                                            \
        case 7: FGCN_BN_APP8H(RES_, 7); break;                                                                                            \
        case 8: FGCN_BN_APP8H(RES_, 8); break;                                                                                            \
        case 9: FGCN_BN_APP8H(RES_, 9); break;                                                                                            \
        case 10: FGCN_BN_APP8H(RES_, 10); break;                                                                                          \
        case 11: FGCN_BN_APP8H(RES_, 11); break;                                                                                          \
        case 12: FGCN_BN_APP8H(RES_, 12); break;                                                                                          \
        case 13: FGCN_BN_APP8H(RES_, 13); break;                                                                                          \
        case 14: FGCN_BN_APP8H(RES_, 14); break;                                                                                          \
        default: FGCN_BN_APP8H(RES_, 15); break;                                                                                          \
    } while (0)
        if (rm == 0) FGCN_BN_APP8(0);
        else if (rm == 1) FGCN_BN_APP8(1);
        else FGCN_BN_APP8(2);
#undef FGCN_BN_APP8
#undef FGCN_BN_APP8H
        return launch_status("bn_act_bwd_apply");
    }
#define FGCN_BN_APP4(RES_, M_, O_, TY_)                                                                            \
    hipLaunchKernelGGL((bn_act_bwd_apply_kernel<RES_, M_, O_, TY_>), g, blk, 0, s, dout, out, sign_mask, a, vec_a, b, vec_b, sums, \
                       da, db, n4, C, relu, train, inv_m, db_accumulate, str, ld_dout, grp_rows, hm)
#define FGCN_BN_APP(RES_, M_)                                                                                      \
    do {                                                                                                           \
        if (o16 && hm) FGCN_BN_APP4(RES_, M_, true, true);                                                         \
        else if (o16) FGCN_BN_APP4(RES_, M_, true, false);                                                         \
        else if (hm) FGCN_BN_APP4(RES_, M_, false, true);                                                          \
        else FGCN_BN_APP4(RES_, M_, false, false);                                                                 \
    } while (0)
    if (res_mode == 0 || !db) {
        if (sign_mask) FGCN_BN_APP(0, true); else FGCN_BN_APP(0, false);
    } else if (res_mode == 1) {
        if (sign_mask) FGCN_BN_APP(1, true); else FGCN_BN_APP(1, false);
    } else {
        if (sign_mask) FGCN_BN_APP(2, true); else FGCN_BN_APP(2, false);
    }
#undef FGCN_BN_APP
#undef FGCN_BN_APP4
    return launch_status("bn_act_bwd_apply");
}

extern "C" int fgcn_col_sum(const float* x, float* partials, long long rows, int C, int ld, void* stream) {
    FGCN_REQUIRE(x && partials && rows > 0 && C > 0 && ld % 4 == 0 && ld >= ((C + 3) & ~3) && aligned16(x), FGCN_E_BADARG,
                 "col_sum: bad argument (rows=%lld C=%d ld=%d)", rows, C, ld);
    const int cx = (C + 3) / 4;
    FGCN_REQUIRE(cx <= 256, FGCN_E_BADARG, "col_sum: C=%d too wide", C);
    int ny = 256 / cx;
    if (ny > 32) ny = 32;
    dim3 blk((unsigned)cx, (unsigned)ny);
    const size_t lds = (size_t)ny * cx * 4 * sizeof(float);
    hipLaunchKernelGGL(col_sum_kernel, dim3((unsigned)fgcn_elem_tiles(rows)), blk, lds, (hipStream_t)stream, x, partials,
                       rows, rows_per_tile_for(rows), C, ld);
    return launch_status("col_sum");
}

extern "C" int fgcn_group_mean_splits(int groups, int rows) {
    int splits = 1;
    while (splits < 32 && (long long)groups * splits < 512 && rows / (splits * 2) >= 64) splits *= 2;
    return splits;
}

extern "C" int fgcn_group_mean(const float* x, float* partial, float* out, int groups, int rows, int C, int ld,
                               void* stream) {
    FGCN_REQUIRE(x && partial && out && groups > 0 && groups <= 65535 && rows > 0 && C > 0 && ld >= C, FGCN_E_BADARG,
                 "group_mean: bad argument (groups=%d rows=%d C=%d ld=%d)", groups, rows, C, ld);
    const int splits = fgcn_group_mean_splits(groups, rows);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(group_sum_kernel, dim3((unsigned)cdiv(C, 64), (unsigned)groups, (unsigned)splits), dim3(64, 16), 0, s, x,
                       partial, rows, C, ld, splits);
    hipLaunchKernelGGL(group_mean_finish_kernel, dim3((unsigned)cdiv((long long)groups * C, 256)), dim3(256), 0, s, partial,
                       out, groups, C, splits, 1.f / (float)rows);
    return launch_status("group_mean");
}

// ---- batched transpose (node-major <-> feature-major images of the 1-D graph convolutions) -------
// out[b][c][r] = in[b][r][c] for r < R, c < C; in rows have stride ld_in, out rows stride ld_out >= R, and the columns
// [R, ld_out) of every out row are zero-filled (the padded contraction index of the adjacency product).
namespace fgcn {
__global__ __launch_bounds__(256) void transpose_kernel(const float* in, float* out, int R, int C, int ld_in, int ld_out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = in + (long long)b * R * ld_in;
    float* dst = out + (long long)b * C * ld_out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < R && c < C) ? src[(long long)r * ld_in + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < C && r < ld_out) dst[(long long)c * ld_out + r] = tile[tx][ty + 8 * i];   // r >= R: the zero padding
    }
}
}  // namespace fgcn

extern "C" int fgcn_transpose(const float* in, float* out, int B, int R, int C, int ld_in, int ld_out, void* stream) {
    FGCN_REQUIRE(in && out && B > 0 && R > 0 && C > 0, FGCN_E_BADARG, "transpose: null pointer or empty shape");
    FGCN_REQUIRE(ld_in >= C && ld_out >= R && B <= 65535 && (C + 31) / 32 <= 65535, FGCN_E_BADARG,
                 "transpose: strides must cover the rows (ld_in=%d C=%d ld_out=%d R=%d)", ld_in, C, ld_out, R);
    dim3 grid((unsigned)((ld_out + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)B);
    hipLaunchKernelGGL(fgcn::transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, R, C, ld_in, ld_out);
    return fgcn::launch_status("transpose");
}

// ---- row softmax of the transposed joint affinity on large graphs (AGCNGraphConvolution, graph_convolution.py:95-100) -------------
// The reference takes softmax over dim -2 of S = theta^T phi / ic  (N, V, V); the host forms S^T (row w, column v), so the
// softmax runs along the contiguous axis: one wave per row, V up to a few thousand.
//   fwd:  c = softmax(scale * st[row][0:V]);  a = c + adj_t[(row % (K*V))][0:V];  columns [V, ld) of c and a are zero-filled
//   bwd:  ds = scale * c .* (da - sum_v c .* da)
namespace fgcn {
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__global__ __launch_bounds__(256) void row_softmax_fwd_kernel(const float* st, const float* adj_t, float* c_out, float* a_out,
                                                              long long rows, int V, int ld, int KV, float scale) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* s = st + row * ld;
    float mx = -INFINITY;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, s[v] * scale);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int v = lane; v < V; v += 64) sum += expf(s[v] * scale - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    const float* ad = adj_t + (row % KV) * (long long)ld;
    for (int v = lane; v < ld; v += 64) {
        const float c = v < V ? expf(s[v] * scale - mx) * inv : 0.f;
        c_out[row * ld + v] = c;
        a_out[row * ld + v] = v < V ? c + ad[v] : 0.f;
    }
}

__global__ __launch_bounds__(256) void row_softmax_bwd_kernel(const float* da, const float* c, float* ds, long long rows, int V,
                                                              int ld, float scale) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* d = da + row * ld;
    const float* cc = c + row * ld;
    float dot = 0.f;
    for (int v = lane; v < V; v += 64) dot += cc[v] * d[v];
    dot = wave_sum(dot);
    for (int v = lane; v < ld; v += 64) ds[row * ld + v] = v < V ? scale * cc[v] * (d[v] - dot) : 0.f;
}
}  // namespace fgcn

extern "C" int fgcn_row_softmax_fwd(const float* st, const float* adj_t, float* c_out, float* a_out, long long rows, int V,
                                    int ld, int KV, float scale, void* stream) {
    FGCN_REQUIRE(st && adj_t && c_out && a_out && rows > 0 && V > 0 && ld >= V && KV > 0, FGCN_E_BADARG,
                 "row_softmax_fwd: bad argument (rows=%lld V=%d ld=%d)", rows, V, ld);
    hipLaunchKernelGGL(fgcn::row_softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, st, adj_t,
                       c_out, a_out, rows, V, ld, KV, scale);
    return fgcn::launch_status("row_softmax_fwd");
}

extern "C" int fgcn_row_softmax_bwd(const float* da, const float* c, float* ds, long long rows, int V, int ld, float scale,
                                    void* stream) {
    FGCN_REQUIRE(da && c && ds && rows > 0 && V > 0 && ld >= V, FGCN_E_BADARG, "row_softmax_bwd: bad argument");
    hipLaunchKernelGGL(fgcn::row_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, da, c, ds,
                       rows, V, ld, scale);
    return fgcn::launch_status("row_softmax_bwd");
}

extern "C" int fgcn_bn_act(const float* a, const float* vec_a, const float* b, const float* vec_b, float* out,
                           unsigned char* sign_mask, long long rows, int C, int res_mode, int relu, void* stream) {
    return bn_act_impl(a, vec_a, b, vec_b, out, sign_mask, rows, C, res_mode, relu, C, stream);
}
extern "C" int fgcn_bn_act_h(const float* a, const float* vec_a, const float* b, const float* vec_b, unsigned short* out_h,
                             unsigned char* sign_mask, long long rows, int C, int res_mode, int relu, void* stream) {
    return bn_act_impl(a, vec_a, b, vec_b, reinterpret_cast<float*>(out_h), sign_mask, rows, C, res_mode, relu, C, stream, true);
}
extern "C" int fgcn_bn_act_bwd_apply_h(const float* dout, int grp_rows, const float* out, const unsigned char* sign_mask,
                                       const float* a, const float* vec_a, const float* b, const float* vec_b,
                                       const float* sums, unsigned short* da_h, float* db,
                                       long long rows, int C, int res_mode, int relu, int train, int db_accumulate, void* stream) {
    FGCN_REQUIRE(grp_rows >= 0, FGCN_E_BADARG, "bn_act_bwd_apply_h: grp_rows=%d", grp_rows);
    return bn_act_bwd_apply_impl(dout, out, sign_mask, a, vec_a, b, vec_b, sums, reinterpret_cast<float*>(da_h), db, rows, C, res_mode, relu,
                                 train, db_accumulate, C, stream, grp_rows, true);
}
extern "C" int fgcn_bn_act_bwd_reduce(const float* dout, const float* out, const unsigned char* sign_mask,
                                      const float* a, const float* vec_a, const float* b, const float* vec_b,
                                      float* partials, int n_tiles, long long rows, int C, int res_mode, int relu,
                                      void* stream) {
    return bn_act_bwd_reduce_impl(dout, out, sign_mask, a, vec_a, b, vec_b, partials, n_tiles, rows, C, res_mode, relu, C, stream);
}
extern "C" int fgcn_bn_act_bwd_apply(const float* dout, const float* out, const unsigned char* sign_mask,
                                     const float* a, const float* vec_a, const float* b, const float* vec_b,
                                     const float* sums, float* da, float* db,
                                     long long rows, int C, int res_mode, int relu, int train, int db_accumulate,
                                     void* stream) {
    return bn_act_bwd_apply_impl(dout, out, sign_mask, a, vec_a, b, vec_b, sums, da, db, rows, C, res_mode, relu, train, db_accumulate, C,
                                 stream);
}

// fgcn_bn_act followed by fgcn_group_mean without the tensor between them (the last block of the model): see bn_act_pool_kernel
extern "C" int fgcn_bn_act_pool_splits(int groups, int grp_rows) {
    int splits = 1;
    while (splits < 64 && (long long)groups * splits < 1024 && grp_rows / (splits * 2) >= 32) splits *= 2;
    return splits;
}

static int bn_act_pool_impl(const float* a, const float* vec_a, const float* b, const float* vec_b, unsigned char* sign_mask,
                            float* partial, float* pooled, int groups, int grp_rows, int C, int res_mode, void* stream, int hm) {
    FGCN_REQUIRE(a && vec_a && sign_mask && partial && pooled, FGCN_E_BADARG, "bn_act_pool: null pointer");
    FGCN_REQUIRE(!hm || fgcn::math_mode() == FGCN_MATH_BF16, FGCN_E_BADARG, "bn_act_pool: bfloat16 tensors need math mode bf16");
    FGCN_REQUIRE(groups > 0 && groups <= 65535 && grp_rows > 0 && C > 0 && C % 8 == 0, FGCN_E_BADARG,
                 "bn_act_pool: groups=%d grp_rows=%d C=%d (C must be a multiple of 8)", groups, grp_rows, C);
    if (int e = check_elem("bn_act_pool", (long long)groups * grp_rows, C, res_mode, b, vec_b)) return e;
    FGCN_REQUIRE(aligned16(a) && aligned16(vec_a) && (!b || aligned16(b)), FGCN_E_ALIGN, "bn_act_pool: 16-byte alignment");
    const int splits = fgcn_bn_act_pool_splits(groups, grp_rows);
    const int per = (int)cdiv(grp_rows, splits);
    int cx = C / 4;
    if (cx > 64) cx = 64;
    const int ny = 256 / cx > 16 ? 16 : 256 / cx;
    const dim3 grid((unsigned)splits, (unsigned)groups, (unsigned)cdiv(C, cx * 4)), blk((unsigned)cx, (unsigned)ny);
    const size_t lds = (size_t)ny * cx * 4 * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define FGCN_BN_POOL(RES_)                                                                                                          \
    do {                                                                                                                            \
        if (hm) hipLaunchKernelGGL((bn_act_pool_kernel<RES_, true>), grid, blk, lds, s, a, vec_a, b, vec_b, sign_mask, partial, grp_rows, per, C, hm); \
        else hipLaunchKernelGGL((bn_act_pool_kernel<RES_, false>), grid, blk, lds, s, a, vec_a, b, vec_b, sign_mask, partial, grp_rows, per, C, 0);    \
    } while (0)
    if (res_mode == 0) FGCN_BN_POOL(0);
    else if (res_mode == 1) FGCN_BN_POOL(1);
    else FGCN_BN_POOL(2);
#undef FGCN_BN_POOL
    hipLaunchKernelGGL(group_mean_finish_kernel, dim3((unsigned)cdiv((long long)groups * C, 256)), dim3(256), 0, s, partial, pooled, groups,
                       C, splits, 1.f / (float)grp_rows);
    return launch_status("bn_act_pool");
}

extern "C" int fgcn_bn_act_pool(const float* a, const float* vec_a, const float* b, const float* vec_b, unsigned char* sign_mask,
                                float* partial, float* pooled, int groups, int grp_rows, int C, int res_mode, void* stream) {
    return bn_act_pool_impl(a, vec_a, b, vec_b, sign_mask, partial, pooled, groups, grp_rows, C, res_mode, stream, 0);
}

// ---- typed forms (`_t`): `half_mask` says which activation operands are bfloat16 tensors (math mode bf16 with half-precision activation
// storage: the reference's autocast semantics, session/procedures/step.py:55-78); bit order = argument order, see include/fgcn.h ------------
extern "C" int fgcn_bn_act_t(const void* a, const float* vec_a, const void* b, const float* vec_b, void* out, unsigned char* sign_mask,
                             long long rows, int C, int res_mode, int relu, int half_mask, void* stream) {
    FGCN_REQUIRE((half_mask & ~7) == 0, FGCN_E_BADARG, "bn_act_t: half_mask=%d", half_mask);
    return bn_act_impl(static_cast<const float*>(a), vec_a, static_cast<const float*>(b), vec_b, static_cast<float*>(out), sign_mask, rows, C,
                       res_mode, relu, C, stream, (half_mask & 4) != 0, half_mask & 3);
}
extern "C" int fgcn_bn_act_pool_t(const void* a, const float* vec_a, const void* b, const float* vec_b, unsigned char* sign_mask,
                                  float* partial, float* pooled, int groups, int grp_rows, int C, int res_mode, int half_mask, void* stream) {
    FGCN_REQUIRE((half_mask & ~3) == 0, FGCN_E_BADARG, "bn_act_pool_t: half_mask=%d", half_mask);
    return bn_act_pool_impl(static_cast<const float*>(a), vec_a, static_cast<const float*>(b), vec_b, sign_mask, partial, pooled, groups,
                            grp_rows, C, res_mode, stream, half_mask);
}
extern "C" int fgcn_bn_act_bwd_reduce_t(const void* dout, int grp_rows, const float* out, const unsigned char* sign_mask, const void* a,
                                        const float* vec_a, const void* b, const float* vec_b, float* partials, int n_tiles, long long rows,
                                        int C, int res_mode, int relu, int half_mask, void* stream) {
    FGCN_REQUIRE((half_mask & ~7) == 0 && grp_rows >= 0, FGCN_E_BADARG, "bn_act_bwd_reduce_t: half_mask=%d grp_rows=%d", half_mask, grp_rows);
    return bn_act_bwd_reduce_impl(static_cast<const float*>(dout), out, sign_mask, static_cast<const float*>(a), vec_a, static_cast<const float*>(b),
                                  vec_b, partials, n_tiles, rows, C, res_mode, relu, C, stream, grp_rows, half_mask);
}
extern "C" int fgcn_bn_act_bwd_apply_t(const void* dout, int grp_rows, const float* out, const unsigned char* sign_mask, const void* a,
                                       const float* vec_a, const void* b, const float* vec_b, const float* sums, void* da, void* db,
                                       long long rows, int C, int res_mode, int relu, int train, int db_accumulate, int half_mask, void* stream) {
    FGCN_REQUIRE((half_mask & ~31) == 0 && grp_rows >= 0, FGCN_E_BADARG, "bn_act_bwd_apply_t: half_mask=%d grp_rows=%d", half_mask, grp_rows);
    return bn_act_bwd_apply_impl(static_cast<const float*>(dout), out, sign_mask, static_cast<const float*>(a), vec_a, static_cast<const float*>(b),
                                 vec_b, sums, static_cast<float*>(da), static_cast<float*>(db), rows, C, res_mode, relu, train, db_accumulate, C, stream,
                                 grp_rows, (half_mask & 8) != 0, half_mask & 7, (half_mask & 16) ? 1 : 0);
}

// The two backward passes with the gradient of a POOLED output (fgcn_bn_act_pool): dout is float[rows / grp_rows][C], one row per group of
// grp_rows consecutive rows (already divided by the group size), read in place of the rows x C broadcast of it
extern "C" int fgcn_bn_act_bwd_reduce_g(const float* dout_g, int grp_rows, const float* out, const unsigned char* sign_mask,
                                        const float* a, const float* vec_a, const float* b, const float* vec_b,
                                        float* partials, int n_tiles, long long rows, int C, int res_mode, int relu, void* stream) {
    FGCN_REQUIRE(grp_rows > 0, FGCN_E_BADARG, "bn_act_bwd_reduce_g: grp_rows=%d", grp_rows);
    return bn_act_bwd_reduce_impl(dout_g, out, sign_mask, a, vec_a, b, vec_b, partials, n_tiles, rows, C, res_mode, relu, C, stream, grp_rows);
}
extern "C" int fgcn_bn_act_bwd_apply_g(const float* dout_g, int grp_rows, const float* out, const unsigned char* sign_mask,
                                       const float* a, const float* vec_a, const float* b, const float* vec_b,
                                       const float* sums, float* da, float* db,
                                       long long rows, int C, int res_mode, int relu, int train, int db_accumulate, void* stream) {
    FGCN_REQUIRE(grp_rows > 0, FGCN_E_BADARG, "bn_act_bwd_apply_g: grp_rows=%d", grp_rows);
    return bn_act_bwd_apply_impl(dout_g, out, sign_mask, a, vec_a, b, vec_b, sums, da, db, rows, C, res_mode, relu, train, db_accumulate, C,
                                 stream, grp_rows);
}

// The same three passes for a plain BatchNorm (no residual, no activation) whose RESULT is a channel window of a wider tensor -- one
// of the six branches of MS-G3D's multi-scale temporal convolution, concatenated on the channel axis (ms_tcn.py:88-109): the forward
// writes rows of stride ld_out >= C into the window (out = window base), the backward reads the window of the concatenation's
// gradient (dout = window base, stride ld_dout) -- no torch.cat, no contiguous copies of its backward slices.
extern "C" int fgcn_bn_apply_ld(const float* a, const float* vec_a, float* out, long long rows, int C, int ld_out, void* stream) {
    FGCN_REQUIRE(ld_out >= C && ld_out % 4 == 0, FGCN_E_ALIGN, "bn_apply_ld: ld_out=%d must cover C=%d and be a multiple of 4", ld_out, C);
    return bn_act_impl(a, vec_a, nullptr, nullptr, out, nullptr, rows, C, 0, 0, ld_out, stream);
}
extern "C" int fgcn_bn_bwd_reduce_ld(const float* dout, int ld_dout, const float* a, const float* vec_a, float* partials, int n_tiles,
                                     long long rows, int C, void* stream) {
    FGCN_REQUIRE(ld_dout >= C && ld_dout % 4 == 0 && aligned16(dout), FGCN_E_ALIGN, "bn_bwd_reduce_ld: ld_dout=%d must cover C=%d, multiples of 4", ld_dout, C);
    return bn_act_bwd_reduce_impl(dout, nullptr, nullptr, a, vec_a, nullptr, nullptr, partials, n_tiles, rows, C, 0, 0, ld_dout, stream);
}
extern "C" int fgcn_bn_bwd_apply_ld(const float* dout, int ld_dout, const float* a, const float* vec_a, const float* sums, float* da,
                                    long long rows, int C, int train, void* stream) {
    FGCN_REQUIRE(ld_dout >= C && ld_dout % 4 == 0 && aligned16(dout), FGCN_E_ALIGN, "bn_bwd_apply_ld: ld_dout=%d must cover C=%d, multiples of 4", ld_dout, C);
    return bn_act_bwd_apply_impl(dout, nullptr, nullptr, a, vec_a, nullptr, nullptr, sums, da, nullptr, rows, C, 0, 0, train, 0, ld_dout, stream);
}
