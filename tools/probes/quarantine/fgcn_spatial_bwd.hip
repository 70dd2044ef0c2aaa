// Fused spatial graph convolution, backward w.r.t. the input and the adjacency (autograd of agcn.py:103-111):
//
//     dagg_k[(n,t,w), c] = sum_o dy[(n,t,w), o] * Wd_k[o][c]                     (never written to HBM)
//     dx[(n,t,v), c]    (+)= sum_k sum_w A^_k[n][v][w] * dagg_k[(n,t,w), c]
//     dA^_k[n][v][w]      = sum_t sum_c x[(n,t,v), c] * dagg_k[(n,t,w), c]       (per T-chunk partials)
//
// Replaces three launches (row GEMM K=Cout -> 3*Cin, joint_mix, joint_gram) and the 3-activation-wide dagg tensor
// they exchanged through HBM.  One wave = one frame at a time, waves are independent (no workgroup barrier in the
// frame loop); per channel tile ci (32 input channels on the lanes) and subset k:
//   (a) D = dagg_k tile (32 w x 32 c): A = dy rows (lane = joint w, one 16-byte buffer load = 4 consecutive o,
//       k-permuted like the row GEMM), B = Wd_k streamed from L2 as k-interleaved float4 (wdt4[k][o/4][c][4], lane =
//       input channel c); Cout/2 MFMAs, loads prefetched one 8-channel step ahead.
//   (b) dx tile (32 v x 32 c) += A^_k (LDS) . D: the accumulator D is the B operand as it stands (16 MFMAs).
//   (c) dA^_k (32 v x 32 w) += x_t (v x c) . D^T: D goes through a wave-private LDS tile to put c on the contraction
//       index (16 ds_write_b32 + 4 ds_read_b128), x fragments are 16-byte row loads; 16 MFMAs.
// Joints sit on 25 of 32 MFMA rows/columns, so the ceiling is 78 % of the f32 MFMA rate, as in the forward kernel.
#include "fgcn_common.hpp"

namespace fgcn {

constexpr int BAHS = 33;   // LDS row stride of a padded joint matrix
constexpr int BTTS = 36;   // row stride of the per-wave transpose tile

struct SpatialBwdP {
    const float* dy;
    const float* x;
    const float* a_hat;
    const float* wdt4;
    float* dx;
    float* partial;
    int B, T, V, Cin, Cout, ld_dy, ld_x, ld_dx, ns, a_batched, t_chunk, accumulate;
    unsigned dy_bytes, x_bytes, w_bytes;
};

__device__ __forceinline__ f32x4 sb_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

__global__ __launch_bounds__(256) void spatial_bwd_kernel(SpatialBwdP p) {
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ah = smem;                                   // [3][32][33]
    float* tt = smem + ((3 * 32 * BAHS + 3) & ~3);      // [4 waves][32][BTTS]; reused as [4][1024] for the final sum

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int V = p.V, NS = p.ns;
    const int t0 = chunk * p.t_chunk;
    const int t1 = min(t0 + p.t_chunk, p.T);

    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wdt4, 0, p.w_bytes, 0x00020000);

    const float* asrc = p.a_hat + (p.a_batched ? (long long)n * NS * V * V : 0);
    for (int i = tid; i < 3 * 32 * 32; i += 256) {
        const int k = i >> 10, v = (i >> 5) & 31, w = i & 31;
        ah[(k * 32 + v) * BAHS + w] = (k < NS && v < V && w < V) ? asrc[(k * V + v) * V + w] : 0.f;
    }
    __syncthreads();

    float* T = tt + wave * 32 * BTTS;
    const int nq = p.Cout >> 3;                          // 8 output channels of dy per step (Cout % 8 == 0)
    const int K4 = p.Cout >> 2;
    // wave-private image of this frame's dy rows: [joint][Cout + 4] (conflict-free 16-byte A-fragment reads).  The frame
    // slice is contiguous in HBM, so it is filled with fully coalesced 16-byte loads ONCE per frame instead of one
    // per-row gather per (subset, channel tile) pass.
    const int DYS = p.Cout + 4;
    float* dyl = tt + 4 * 32 * BTTS + wave * V * DYS;      // V rows per wave (rows >= V are never read)
    const int row4 = p.Cout >> 2;                        // float4 per dy row
    const int n4 = V * row4;                             // float4 in the frame slice

    const int rsh = 31 - __clz(row4);                    // row4 is a power of two (host check)
    const bool small = n4 <= 512;
    f32x4 pf[8];
    if (small) {
        const unsigned row_first = (unsigned)(((long long)n * p.T + min(t0 + wave, p.T - 1)) * V);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = lane + 64 * j;
            pf[j] = sb_load4(rdy, (i < n4 && t0 + wave < t1) ? ((row_first + (i >> rsh)) * (unsigned)p.ld_dy + 4 * (i & (row4 - 1))) * 4u : OOB, 0);
        }
    }

    f32x16 acca[3] = {zero16(), zero16(), zero16()};     // dA^_k, summed over this wave's frames
    for (int t = t0 + wave; t < t1; t += 4) {
        const unsigned rowb = (unsigned)(((long long)n * p.T + t) * V);
        // lane-per-row offsets: joint l31 of this frame (absent joints read zeros)
        // fill the image: 8 coalesced 16-byte loads in flight per lane, then 8 LDS stores.  When the whole slice fits in
        // one batch (Cout <= 64 ... n4 <= 512) it was already requested during the previous frame (register prefetch).
        if (small) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = lane + 64 * j;
                if (i < n4) *reinterpret_cast<f32x4*>(&dyl[(i >> rsh) * DYS + 4 * (i & (row4 - 1))]) = pf[j];
            }
            const int tn = t + 4;                        // this wave's next frame
            const unsigned rown = (unsigned)(((long long)n * p.T + (tn < t1 ? tn : t)) * V);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = lane + 64 * j;
                pf[j] = sb_load4(rdy, (i < n4 && tn < t1) ? ((rown + (i >> rsh)) * (unsigned)p.ld_dy + 4 * (i & (row4 - 1))) * 4u : OOB, 0);
            }
        } else {
            for (int i0 = lane; i0 < n4; i0 += 512) {
                f32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = i0 + 64 * j;
                    v[j] = sb_load4(rdy, i < n4 ? ((rowb + (i >> rsh)) * (unsigned)p.ld_dy + 4 * (i & (row4 - 1))) * 4u : OOB, 0);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = i0 + 64 * j;
                    if (i < n4) *reinterpret_cast<f32x4*>(&dyl[(i >> rsh) * DYS + 4 * (i & (row4 - 1))]) = v[j];
                }
            }
        }
        const float* dya = dyl + (l31 < V ? l31 : 0) * DYS + 4 * h;   // lanes w >= V reuse row 0 and are zeroed at use
        const bool wok = l31 < V;
        const unsigned xo = l31 < V ? ((rowb + l31) * (unsigned)p.ld_x + 4 * h) * 4u : OOB;
        const int nci = (p.Cin + 31) >> 5;
#pragma unroll 1
        for (int ci = 0; ci < nci; ++ci) {               // runtime loop: the dx tile of a channel tile only sums over k
            f32x16 accx = zero16();
            const int c = ci * 32 + l31;
            const unsigned wvo = c < p.Cin ? (unsigned)(h * p.Cin + c) * 16u : OOB;
            // x fragments of this channel tile for step (c): x[(t, v = lane)][ci*32 + 8q + 4h + e]
            f32x4 xa[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                xa[q] = sb_load4(rx, (ci * 32 + 8 * q + 4 * h < p.Cin) ? xo : OOB, (unsigned)(ci * 32 + 8 * q) * 4u);
#pragma unroll
            for (int k = 0; k < 3; ++k) {                // static index into acca[]; NS <= 3
                if (k >= NS) break;
                // ---- (a) D = dy_t . Wd_k[:, ci tile] -------------------------------------------------------------
                f32x16 D = zero16();
                const unsigned wso = (unsigned)(k * K4 * p.Cin) * 16u;
                // weight fragments ride a 4-deep register ring (one 16-byte L2 load feeds only 4 MFMAs = 256 cycles,
                // less than the L2 latency, so three to four loads must be in flight)
                f32x4 bq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bq[j] = sb_load4(rw, wvo, wso + (unsigned)(2 * j * p.Cin) * 16u);
                // every load below is unconditional (the tail asks for an out-of-range offset and gets zeros): a load in
                // a conditional block makes hipcc drain the whole ring with vmcnt(0); the A fragment of step q+1 is read
                // from LDS before the MFMAs of step q.  nq % 4 == 0 (Cout % 32 == 0, host check).
                f32x4 an = *reinterpret_cast<const f32x4*>(dya);
                for (int q0 = 0; q0 < nq; q0 += 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int q = q0 + j;
                        f32x4 a0 = an;
                        if (!wok) a0 = f32x4{0.f, 0.f, 0.f, 0.f};
                        an = *reinterpret_cast<const f32x4*>(dya + 8 * min(q + 1, nq - 1));
#pragma unroll
                        for (int e = 0; e < 4; ++e) D = mfma32(a0[e], bq[j][e], D);
                        bq[j] = sb_load4(rw, q + 4 < nq ? wvo : OOB, wso + (unsigned)(2 * (q + 4) * p.Cin) * 16u);
                    }
                }
                // ---- (b) dx tile += A^_k . D : register r of D is contraction row w = (r&3) + 8(r>>2) + 4h -------
                const float* ak = ah + (k * 32 + l31) * BAHS + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) accx = mfma32(ak[(r & 3) + 8 * (r >> 2)], D[r], accx);
                // ---- (c) dA^_k += x_t[:, ci tile] . D^T through the wave-private tile T[w][c] --------------------
#pragma unroll
                for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * h) * BTTS + l31] = D[r];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 tb = *reinterpret_cast<const f32x4*>(&T[l31 * BTTS + 8 * q + 4 * h]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acca[k] = mfma32(xa[q][e], tb[e], acca[k]);
                }
            }
            // ---- dx tile of this channel tile: rows v in the registers, channels on the lanes (128 B per half-wave) --
            if (c < p.Cin) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int v = acc_row(r, lane);
                    if (v < V) {
                        float* dst = p.dx + ((long long)rowb + v) * p.ld_dx + c;
                        *dst = p.accumulate ? *dst + accx[r] : accx[r];
                    }
                }
            }
        }
    }
    // ---- deterministic cross-wave sum of the three dA^ matrices, one at a time ---------------------------------------
    float* red = tt;
    const int nchunk = gridDim.x;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < NS) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acca[k][r];
            __syncthreads();
            float* dst = p.partial + (((long long)n * nchunk + chunk) * NS + k) * 1024;
            for (int e = tid; e < 1024; e += 256) {
                const float s = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
                const int r = e >> 6, l = e & 63;
                dst[acc_row(r, l) * 32 + (l & 31)] = s;
            }
        }
    }
}

}  // namespace fgcn

using namespace fgcn;

static int spatial_bwd_t_chunk(int B, int T) {
    int chunk = 32;
    while (chunk > 4 && (long long)B * cdiv(T, chunk) < 1024) chunk >>= 1;
    return chunk;
}

extern "C" int fgcn_spatial_bwd_chunks(int B, int T) { return (int)cdiv(T, spatial_bwd_t_chunk(B, T)); }

extern "C" int fgcn_spatial_bwd(const float* dy, const float* x, const float* a_hat, const float* wdt4, float* dx,
                                float* partial, int B, int T, int V, int Cin, int Cout, int ld_dy, int ld_x, int ld_dx,
                                int n_subsets, int a_hat_batched, int accumulate, void* stream) {
    FGCN_REQUIRE(dy && x && a_hat && wdt4 && dx && partial, FGCN_E_BADARG, "spatial_bwd: null pointer");
    FGCN_REQUIRE(B > 0 && B <= 65535 && T > 0 && V > 0 && V <= FGCN_MAX_V && Cin > 0 && Cout > 0, FGCN_E_BADARG,
                 "spatial_bwd: bad sizes B=%d T=%d V=%d Cin=%d Cout=%d", B, T, V, Cin, Cout);
    FGCN_REQUIRE(n_subsets >= 1 && n_subsets <= 3, FGCN_E_BADARG, "spatial_bwd: n_subsets=%d (1..3)", n_subsets);
    FGCN_REQUIRE((Cout & (Cout - 1)) == 0 && Cout >= 32, FGCN_E_BADARG, "spatial_bwd: Cout must be a power of two >= 32 (got %d)", Cout);
    FGCN_REQUIRE(Cin % 4 == 0 && Cout % 16 == 0 && ld_dy % 4 == 0 && ld_x % 4 == 0 && ld_dy >= Cout && ld_x >= Cin &&
                     ld_dx >= Cin,
                 FGCN_E_ALIGN, "spatial_bwd: Cin %% 4, Cout %% 16 and 4-float row strides required (Cin=%d Cout=%d)", Cin,
                 Cout);
    FGCN_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(wdt4), FGCN_E_ALIGN, "spatial_bwd: 16-byte alignment");
    const long long dyb = (long long)B * T * V * ld_dy * 4, xb = (long long)B * T * V * ld_x * 4;
    FGCN_REQUIRE(dyb < 0x7FFF0000ll && xb < 0x7FFF0000ll, FGCN_E_BADARG, "spatial_bwd: tensors must be smaller than 2 GiB");
    const int ci = (Cin + 31) / 32;
    FGCN_REQUIRE(ci <= 8, FGCN_E_BADARG, "spatial_bwd: at most 256 input channels (Cin=%d)", Cin);
    SpatialBwdP p{dy, x, a_hat, wdt4, dx, partial, B, T, V, Cin, Cout, ld_dy, ld_x, ld_dx, n_subsets, a_hat_batched,
                  spatial_bwd_t_chunk(B, T), accumulate, (unsigned)dyb, (unsigned)xb,
                  (unsigned)((long long)n_subsets * Cout * Cin * 4)};
    // joint matrices + transpose tiles (4608 floats, reused as the 4096-float final-sum scratch) + 4 dy frame images
    const size_t lds = (((3 * 32 * BAHS + 3) & ~3) + 4 * 32 * BTTS + 4 * V * (Cout + 4)) * sizeof(float);
    FGCN_REQUIRE(lds <= 160 * 1024, FGCN_E_BADARG, "spatial_bwd: V=%d x Cout=%d needs %zu bytes of LDS (> 160 KiB)", V, Cout,
                 lds);
    dim3 grid((unsigned)cdiv(T, p.t_chunk), (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    static bool lds_opt_in = false;   // once per process: allow the full 160 KiB of gfx950 LDS as dynamic shared memory
    if (!lds_opt_in) {
        const int max_lds = 160 * 1024;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        lds_opt_in = true;
    }
    FGCN_REQUIRE(Cout <= 256, FGCN_E_BADARG, "spatial_bwd: at most 256 output channels (Cout=%d)", Cout);
    hipLaunchKernelGGL(spatial_bwd_kernel, grid, dim3(256), lds, s, p);
    return launch_status("spatial_bwd");
}
