// Forward of the attention embeddings with the affinity gram on chip (split-bf16 math modes):
//
//     emb[(n,t,v), j]   = sum_c x[(n,t,v), c] Wemb[c][j] + bemb[j],     j over [th0 ph0 th1 ph1 th2 ph2], each ic wide
//     S_k[n][v][w]      = sum_{t,e} emb[(n,t,v), th_k + e] emb[(n,t,w), ph_k + e]        (partial sums per row segment)
//
// reference: A1 = conv_a[k](x), A2 = conv_b[k](x), torch.matmul(A1, A2) of SpatialGraphConv.forward, torch_src/models/mmargcn/agcn.py:104-106
// (the 1 / (ic T) scale and the softmax stay in fgcn_adj_softmax_fwd).  Until round 5 the stacked 1x1 product wrote emb and fgcn_joint_gram
// read all 1.5 activations of it back for ten V x V matrices per sample; emb has to be written for the backward, but the gram can be formed
// from the tile while it is on chip.
//
// A workgroup (4 waves; 2 x 2 over channels x rows) owns a contiguous range of frame tiles (F = 128 / V whole frames) of ONE sample and CW
// embedding channels: all 6 ic of them for ic = 16 / 32, one subset's th_k | ph_k (2 ic = 128) for ic = 64 (three workgroups per row range).
//   * The product is formed TRANSPOSED: emb^T (CW x 128 rows) = Wemb^T . x^T with the pre-split weights (fgcn_pack_split3 of the Cin x 6 ic
//     matrix; streamed from L2 through a two-slot ring) as the A operand and the x tile -- staged in 32-channel chunks as bf16 planes, split once,
//     the next chunk parked in registers during the MFMAs -- as the B operand.  An accumulator lane then holds four consecutive CHANNELS of one
//     row: 16-byte stores of emb, 8-byte pieces of the gram's image.
//   * Per subset the tile's th_k | ph_k channels (bias included: what the reference multiplies) are split into an LDS image [row][2 ic]; wave
//     (vt, wt) owns the 16 x 16 tile (v in 16 vt .., w in 16 wt ..) of S_k and adds, frame by frame, th_kf (A operand: image rows of the frame,
//     8 channels per lane) . ph_kf^T (B operand: the same rows, the phi half) on 16x16x32 MFMAs; joints >= V are masked to zero.  The gram
//     accumulators live across the workgroup's tiles: one (3, 32, 32) partial per row segment (fgcn_adj_softmax_fwd sums them).
// NP = 3: exact three-way bf16 splits (FGCN_MATH_BF16X3, either product form); NP = 1: operands rounded to bfloat16 once (FGCN_MATH_BF16).
// Every sum has a fixed order (bitwise reproducible).
#include <algorithm>

#include "fgcn_common.hpp"

namespace fgcn {

struct EmbFwP {
    const float* x;
    const void* w3;                     // fgcn_pack_split3 form of the Cin x (6 ic) matrix: [part][c / 8][j][8] bf16
    const float* bias;                  // float[6 ic]
    float* emb;
    float* partial;                     // float[B][nseg][3][32][32]
    int B, T, V, Cin, ic, Ce, ld_x, ld_e;
    int F, tiles_t, tps, nseg, ncol;    // frames per tile, tiles per sample, tiles per segment, segments per sample, column workgroups
    unsigned x_bytes, e_bytes, w_plane_bytes, p_bytes;
};

constexpr unsigned EF_OOB = 0x80000000u;
constexpr int EF_XS = 64;               // bytes per row and part of the x image (32 channels x bf16), 32-byte blocks XOR-swizzled by row bit 2
constexpr int EF_XPLANE = 128 * EF_XS;
// gram image of one pass: [part][128 rows][(W theta + W phi channels) x bf16 + 16 pad bytes], W = min(ic, 32) -- one 32-channel slice of a
// subset's theta | phi at a time, so that the 64-channel groups of the 256-output blocks keep two workgroups per CU (all 128 channels at
// once: 104 KB, one workgroup of four waves per CU, 0.83 against 0.46 ms for the unfused pair)
constexpr int ef_gw(int ic) { return ic < 32 ? ic : 32; }
constexpr int ef_gs(int ic) { return 4 * ef_gw(ic) + 16; }
template <int NP> constexpr int ef_lds(int ic) { return std::max(NP * EF_XPLANE, NP * 128 * ef_gs(ic)); }

// MU: 16-channel units per wave (CW = 32 MU channels per workgroup); IC: channels per group (16, 32, 64); NSUB: subsets of the workgroup
// (3 when it holds all 6 ic channels, 1 when it holds th_k | ph_k of one subset)
// E16 (NP = 1): emb is written as BFLOAT16 (half-precision storage: only the bf16 staging of fgcn_emb_dx_tile_h / fgcn_emb_wgrad_tile_h reads
// it, and that staging rounds to bfloat16 anyway -- the same values, half the bytes; ld_e in elements)
// H16 bit 0 = that (emb bfloat16), bit 1 = x is a BFLOAT16 tensor too (half-precision activation storage, the `_t` entry point; ld_x in
// elements): its rows are copied into the image, 8 bytes per four channels -- the staged bytes of the float32 tensor of the same values
template <int NP, int MU, int IC, int NSUB, int H16 = 0>
__global__ __launch_bounds__(256, 2) void emb_fwd_tile_kernel(EmbFwP p) {
    static_assert(!H16 || NP == 1, "bfloat16 tensors: the one-part kernel");
    constexpr bool E16 = (H16 & 1) != 0, X16 = (H16 & 2) != 0;
    constexpr int CW = 32 * MU, NR = 4, GW = ef_gw(IC), GS = ef_gs(IC), GPLANE = 128 * GS, KS = IC >= 32 ? IC / 32 : 1;
    static_assert(CW == (NSUB == 3 ? 6 * IC : 2 * IC), "workgroup channels");
    auto swz = [](int r) -> unsigned { return (unsigned)(r & 4) << 3; };
    extern __shared__ __attribute__((aligned(16))) unsigned char ef_lds_raw[];
    unsigned char* Xh = ef_lds_raw;                                  // x image [NP][128 rows][64 B] / gram image [NP][128 rows][GS]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wc = wave & 1, wr = wave >> 1;
    const int col_wg = blockIdx.x % p.ncol;
    const int rs = blockIdx.x / p.ncol;
    const int n = rs / p.nseg, seg = rs - n * p.nseg;
    const int V = p.V, F = p.F;
    const int tile_lo = seg * p.tps, tile_hi = min(tile_lo + p.tps, p.tiles_t);
    const int cbase = col_wg * CW;                                   // first embedding channel of the workgroup
    const int ch0 = cbase + wc * 16 * MU;                            // ... of this wave (+ 16 mu)

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, p.w_plane_bytes * NP, 0x00020000);
    const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)p.emb, 0, p.e_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, (unsigned)p.Ce * 4u, 0x00020000);

    // bias of this lane's channels: unit mu, channels ch0 + 16 mu + 4 g4 .. + 3
    f32x4 bv[MU];
#pragma unroll
    for (int mu = 0; mu < MU; ++mu) bv[mu] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)(ch0 + 16 * mu + 4 * g4) * 4u, 0, 0));

    // weight fragment of (unit mu, 32-channel chunk kc): lane (channel l15, g4) <- Wemb[kc + 8 g4 .. + 7][ch0 + 16 mu + l15]
    const int nchunks = p.Cin >> 5;
    unsigned wvoff[MU];
#pragma unroll
    for (int mu = 0; mu < MU; ++mu) wvoff[mu] = (unsigned)(((long long)g4 * p.Ce + ch0 + 16 * mu + l15) * 16);
    auto load_w = [&](u32x4v (&dst)[NP], int mu, int c) {
        if (c >= nchunks) c = 0;                                     // past the last chunk: the first one of the next tile
        const unsigned so = (unsigned)(((long long)(4 * c) * p.Ce) * 16);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) dst[pl] = __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff[mu], so + pl * p.w_plane_bytes, 0);
    };
    // x rows of a chunk: thread (row tid / 8 + 32 i, channels 4 (tid % 8) .. + 3)
    const int srow = tid >> 3, sg = tid & 7;
    f32x4 stg[4];
    auto fetch = [&](int tile, int c) {
        const int t0_ = tile * F;
        const int nrows_ = tile < tile_hi ? min(F, p.T - t0_) * V : 0;
        const unsigned row0_ = (unsigned)((n * p.T + t0_) * V);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = srow + 32 * i;
            const unsigned off = r < nrows_ ? ((row0_ + (unsigned)r) * (unsigned)p.ld_x + (unsigned)(32 * c + 4 * sg)) * (X16 ? 2u : 4u) : EF_OOB;
            if constexpr (X16) {                                     // four bfloat16 = 8 bytes, parked in the first two components
                const u32x2 h = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, off, 0, 0));
                const unsigned b0 = h[0], b1 = h[1];                 // (element -> scalar before a bit cast: hipcc 7.2 reads element 0 otherwise)
                stg[i] = f32x4{__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1), 0.f, 0.f};
            } else {
                stg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
            }
        }
    };
    auto deposit = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = srow + 32 * i;
            u32x2 parts[NP];
            if constexpr (X16) {                                     // already bfloat16: a copy
                const float e0 = stg[i][0], e1 = stg[i][1];
                parts[0] = u32x2{__builtin_bit_cast(unsigned, e0), __builtin_bit_cast(unsigned, e1)};
            } else {
                splitn_x4<NP>(stg[i], parts);
            }
            unsigned char* dst = Xh + r * EF_XS + ((unsigned)(sg * 8) ^ swz(r));
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<u32x2*>(dst + pl * EF_XPLANE) = parts[pl];
        }
    };
    const int xrow = wr * 64 + l15;                                  // + 16 nt
    auto load_x = [&](u32x4v (&dst)[NP], int nt) {
        const int r = xrow + 16 * nt;
        const unsigned char* src = Xh + r * EF_XS + ((unsigned)(16 * g4) ^ swz(r));
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) dst[pl] = *reinterpret_cast<const u32x4v*>(src + pl * EF_XPLANE);
    };

    // gram accumulators: this wave's 16 x 16 tile (v tile vt, w tile wt) of every subset the workgroup holds
    const int vt = wave >> 1, wt = wave & 1;
    f32x4 gacc[NSUB];
#pragma unroll
    for (int k = 0; k < NSUB; ++k) gacc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    // weight ring slots: a divisor of MU, so that unit 0 of the next chunk lands in slot 0; fragments are requested WD units ahead.  One part
    // (FGCN_MATH_BF16): one MFMA per fragment and row tile instead of six -- one unit ahead leaves the L2 latency exposed (fgcn_tconv.hip,
    // FGCN_HALO_RING_NP1): a deeper ring there
#ifndef FGCN_EF_RING_NP1
#define FGCN_EF_RING_NP1 1
#endif
    constexpr int RS = (NP == 1 && FGCN_EF_RING_NP1) ? (MU % 4 == 0 ? 4 : 3) : (MU % 2 == 0 ? 2 : 3);
    constexpr int WD = (NP == 1 && FGCN_EF_RING_NP1) ? RS - 1 : 1;
    static_assert(MU % RS == 0 && WD < RS && WD <= MU, "ring");
    u32x4v wq[RS][NP];
#pragma unroll
    for (int d = 0; d < WD; ++d) load_w(wq[d], d, 0);
    fetch(tile_lo, 0);
    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        const int t0 = tile * F;
        const int nf = min(F, p.T - t0);
        const int nrows = nf * V;
        const unsigned m0 = (unsigned)((n * p.T + t0) * V);
        f32x4 acc[MU][NR];
#pragma unroll
        for (int mu = 0; mu < MU; ++mu)
#pragma unroll
            for (int nt = 0; nt < NR; ++nt) acc[mu][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---- emb^T tile = Wemb^T . x^T ----------------------------------------------------------------------------------------------
        for (int c = 0; c < nchunks; ++c) {
            __syncthreads();                                         // the image (x chunk / gram image) is free
            deposit();
            __syncthreads();
            if (c + 1 < nchunks) fetch(tile, c + 1);                 // lands during the MFMAs below
            else fetch(tile + 1, 0);                                 // (past the segment: nothing is read)
            u32x4v xf[NR][NP];
#pragma unroll
            for (int nt = 0; nt < NR; ++nt) load_x(xf[nt], nt);
#pragma unroll
            for (int mu = 0; mu < MU; ++mu) {
                if (mu + WD < MU) load_w(wq[(mu + WD) % RS], mu + WD, c);
                else load_w(wq[(mu + WD - MU) % RS], mu + WD - MU, c + 1);
#pragma unroll
                for (int nt = 0; nt < NR; ++nt) acc[mu][nt] = mfma_np_k32<NP>(wq[mu % RS], xf[nt], acc[mu][nt]);
            }
        }
        // ---- bias, 16-byte stores of emb (lane = row, four consecutive channels per unit) ----------------------------------------------
        // (row tiles outside, channel units inside: consecutive stores fill a row's 64-byte pieces in address order, so the halves of a 128-byte
        // line reach the L2 back to back -- fgcn_spatial_tile.hip's epilogue has the measurement)
#pragma unroll
        for (int nt = 0; nt < NR; ++nt)
#pragma unroll
            for (int mu = 0; mu < MU; ++mu) {
                acc[mu][nt] += bv[mu];
                const int R = wr * 64 + 16 * nt + l15;
                const unsigned off = R < nrows ? ((m0 + (unsigned)R) * (unsigned)p.ld_e + (unsigned)(ch0 + 16 * mu + 4 * g4)) * 4u : EF_OOB;
                if constexpr (E16) {
                    const u32x2 h = __builtin_bit_cast(u32x2, pack_bf16(acc[mu][nt]));
                    __builtin_amdgcn_raw_buffer_store_b64(h, re, off == EF_OOB ? EF_OOB : off >> 1, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, acc[mu][nt]), re, off, 0, 0);
                }
            }
        // ---- the gram of every subset from the tile: per pass one GW-channel slice of theta_k | phi_k goes into the image -------------------
#pragma unroll
        for (int ks = 0; ks < NSUB; ++ks) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                __syncthreads();                                     // the previous image's reads are done
#pragma unroll
                for (int mu = 0; mu < MU; ++mu) {
                    const int cw = wc * 16 * MU + 16 * mu;           // unit's first channel inside the workgroup
                    const int grp = cw / IC, sub = NSUB == 3 ? (grp >> 1) : 0, side = grp & 1;   // (wave-uniform)
                    const int cg = cw - grp * IC;                    // channel inside its group
                    if (sub != ks || cg / GW != s) continue;
                    const int ci = side * GW + (cg - s * GW);        // channel inside the pass's theta | phi image
#pragma unroll
                    for (int nt = 0; nt < NR; ++nt) {
                        const int R = wr * 64 + 16 * nt + l15;
                        u32x2 parts[NP];
                        splitn_x4<NP>(acc[mu][nt], parts);
                        unsigned char* dst = Xh + R * GS + (ci + 4 * g4) * 2;
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<u32x2*>(dst + pl * GPLANE) = parts[pl];
                    }
                }
                __syncthreads();
                for (int f = 0; f < nf; ++f) {
                    const bool a_ok = 16 * vt + l15 < V && (IC >= 32 || g4 < 2), b_ok = 16 * wt + l15 < V && (IC >= 32 || g4 < 2);
                    const int ra = a_ok ? f * V + 16 * vt + l15 : 0, rbw = b_ok ? f * V + 16 * wt + l15 : 0;   // (absent joints / channels: row 0, then zeroed)
                    u32x4v af[NP], bf[NP];
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) {
                        const u32x4v a = *reinterpret_cast<const u32x4v*>(Xh + pl * GPLANE + ra * GS + (8 * g4) * 2);
                        const u32x4v b = *reinterpret_cast<const u32x4v*>(Xh + pl * GPLANE + rbw * GS + (GW + 8 * g4) * 2);
                        af[pl] = a_ok ? a : u32x4v{0u, 0u, 0u, 0u};
                        bf[pl] = b_ok ? b : u32x4v{0u, 0u, 0u, 0u};
                    }
                    gacc[ks] = mfma_np_k32<NP>(af, bf, gacc[ks]);
                }
            }
        }
    }

    // ---- the segment's partial matrices: lane (w = 16 wt + l15, g4), register r -> v = 16 vt + 4 g4 + r ----------------------------------
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)p.partial, 0, p.p_bytes, 0x00020000);
#pragma unroll
    for (int ks = 0; ks < NSUB; ++ks) {
        const int k = NSUB == 3 ? ks : col_wg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned off = ((((unsigned)(n * p.nseg + seg) * 3u + (unsigned)k) * 32u + (unsigned)(16 * vt + 4 * g4 + r)) * 32u + (unsigned)(16 * wt + l15)) * 4u;
            const float val = gacc[ks][r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rp, off, 0, 0);
        }
    }
}

struct EfGeom {
    int F, tiles_t, tps, nseg, ncol;
};
static EfGeom ef_geom(int B, int T, int V, int ic) {
    EfGeom g;
    g.F = 128 / V;
    g.tiles_t = (int)cdiv(T, g.F);
    g.ncol = ic == 64 ? 3 : 1;
    // resident workgroups: two per CU (FGCN_MATH_BF16: twice as many, shorter segments -- the one-part kernel's tiles are too short to cover a
    // workgroup's prologue: 26.47 -> 26.34 ms per bf16 step); tuning key 22 overrides the target
    const int slots = fgcn::tuning(22) > 0 ? fgcn::tuning(22) : (fgcn::math_mode() == FGCN_MATH_BF16 ? 1024 : 512);
    const int want = std::max(1, slots / (B * g.ncol));              // segments per sample
    g.tps = (int)cdiv(g.tiles_t, std::min(g.tiles_t, want));
    g.nseg = (int)cdiv(g.tiles_t, g.tps);
    return g;
}

static bool emb_fwd_mode_ok() { return fgcn::math_mode() == FGCN_MATH_BF16X3 || fgcn::math_mode() == FGCN_MATH_BF16; }
static bool emb_fwd_sizes_ok(int V, int ic, int Cin) {
    return V >= 16 && V <= FGCN_MAX_V && (ic == 16 || ic == 32 || ic == 64) && Cin >= 32 && Cin % 32 == 0;
}

}  // namespace fgcn

using namespace fgcn;

// 1 when fgcn_emb_fwd_tile runs these sizes in the current math mode (FGCN_MATH_BF16X3 with either product form -- the kernel always multiplies
// three-way bf16 splits there -- or FGCN_MATH_BF16; 16 .. 32 joints; ic 16, 32 or 64; Cin a multiple of 32)
extern "C" int fgcn_emb_fwd_tile_available(int V, int ic, int Cin) { return (emb_fwd_mode_ok() && emb_fwd_sizes_ok(V, ic, Cin)) ? 1 : 0; }

// row segments per sample = partial matrices per sample (0: sizes the kernel does not take)
extern "C" int fgcn_emb_fwd_tile_segments(int B, int T, int V, int ic) {
    if (B <= 0 || T <= 0 || V < 16 || V > FGCN_MAX_V || !(ic == 16 || ic == 32 || ic == 64)) return 0;
    return ef_geom(B, T, V, ic).nseg;
}

static int emb_fwd_tile_impl(const float* x, const void* w3, const float* bias, float* emb, float* partial, int B, int T, int V, int Cin,
                             int ic, int ld_x, int ld_e, void* stream, int e16);

extern "C" int fgcn_emb_fwd_tile(const float* x, const void* w3, const float* bias, float* emb, float* partial, int B, int T, int V, int Cin,
                                 int ic, int ld_x, int ld_e, void* stream) {
    return emb_fwd_tile_impl(x, w3, bias, emb, partial, B, T, V, Cin, ic, ld_x, ld_e, stream, 0);
}

// emb written as BFLOAT16 (math mode bf16 only; ld_e in elements): its only readers, fgcn_emb_dx_tile_h / fgcn_emb_wgrad_tile_h, copy instead of
// convert -- bit-identical results, half the bytes of the 1.5-activation-wide tensor
extern "C" int fgcn_emb_fwd_tile_h(const float* x, const void* w3, const float* bias, unsigned short* emb_h, float* partial, int B, int T, int V,
                                   int Cin, int ic, int ld_x, int ld_e, void* stream) {
    FGCN_REQUIRE(emb_h, FGCN_E_BADARG, "emb_fwd_tile_h: null pointer");
    return emb_fwd_tile_impl(x, w3, bias, reinterpret_cast<float*>(emb_h), partial, B, T, V, Cin, ic, ld_x, ld_e, stream, 1);
}

// typed form (math mode bf16): half_mask bit 0 = x is a bfloat16 tensor, bit 1 = emb is written as bfloat16 (emb may be NULL)
extern "C" int fgcn_emb_fwd_tile_t(const void* x, const void* w3, const float* bias, void* emb, float* partial, int B, int T, int V, int Cin,
                                   int ic, int ld_x, int ld_e, int half_mask, void* stream) {
    FGCN_REQUIRE((half_mask & ~3) == 0, FGCN_E_BADARG, "emb_fwd_tile_t: half_mask=%d", half_mask);
    return emb_fwd_tile_impl(static_cast<const float*>(x), w3, bias, static_cast<float*>(emb), partial, B, T, V, Cin, ic, ld_x, ld_e, stream,
                             ((half_mask & 2) ? 1 : 0) | ((half_mask & 1) ? 2 : 0));
}

static int emb_fwd_tile_impl(const float* x, const void* w3, const float* bias, float* emb, float* partial, int B, int T, int V, int Cin,
                             int ic, int ld_x, int ld_e, void* stream, int e16) {      // e16: bit 0 = emb bfloat16, bit 1 = x bfloat16 (1 or 3)
    FGCN_REQUIRE(!e16 || fgcn::math_mode() == FGCN_MATH_BF16, FGCN_E_BADARG, "emb_fwd_tile_h: bfloat16 tensors need math mode bf16");
    // emb == NULL (inference: nothing reads the embeddings after the gram): the kernel's stores of emb go to an empty buffer descriptor and are
    // dropped by the hardware -- the 1.5-activation-wide tensor is never written
    const bool write_emb = emb != nullptr;
    if (!write_emb) emb = partial;
    FGCN_REQUIRE(x && w3 && bias && partial, FGCN_E_BADARG, "emb_fwd_tile: null pointer");
    FGCN_REQUIRE(B > 0 && T > 0, FGCN_E_BADARG, "emb_fwd_tile: bad sizes B=%d T=%d", B, T);
    FGCN_REQUIRE(fgcn_emb_fwd_tile_available(V, ic, Cin), FGCN_E_BADARG,
                 "emb_fwd_tile: needs math mode bf16x3 or bf16, 16 <= V <= %d, ic 16 / 32 / 64, Cin %% 32 == 0 (V=%d ic=%d Cin=%d, mode %d)", FGCN_MAX_V,
                 V, ic, Cin, fgcn::math_mode());
    const int Ce = 6 * ic;
    FGCN_REQUIRE(ld_x % 4 == 0 && ld_e % 4 == 0 && ld_x >= Cin && ld_e >= Ce, FGCN_E_ALIGN, "emb_fwd_tile: row strides");
    FGCN_REQUIRE(aligned16(x) && aligned16(w3) && aligned16(emb) && aligned16(bias) && (reinterpret_cast<uintptr_t>(partial) & 3u) == 0, FGCN_E_ALIGN,
                 "emb_fwd_tile: 16-byte alignment");
    const long long x_bytes = (long long)B * T * V * ld_x * ((e16 & 2) ? 2 : 4), e_bytes = (long long)B * T * V * ld_e * ((e16 & 1) ? 2 : 4);
    const long long plane = (long long)Cin * Ce * 2;
    FGCN_REQUIRE(x_bytes < 0x7FFF0000ll && e_bytes < 0x7FFF0000ll && plane * 3 < 0x7FFF0000ll, FGCN_E_BADARG,
                 "emb_fwd_tile: tensors must be smaller than 2 GiB (32-bit buffer offsets)");
    const EfGeom g = ef_geom(B, T, V, ic);
    EmbFwP p;
    p.x = x; p.w3 = w3; p.bias = bias; p.emb = emb; p.partial = partial;
    p.B = B; p.T = T; p.V = V; p.Cin = Cin; p.ic = ic; p.Ce = Ce; p.ld_x = ld_x; p.ld_e = ld_e;
    p.F = g.F; p.tiles_t = g.tiles_t; p.tps = g.tps; p.nseg = g.nseg; p.ncol = g.ncol;
    p.x_bytes = (unsigned)x_bytes; p.e_bytes = write_emb ? (unsigned)e_bytes : 0u; p.w_plane_bytes = (unsigned)plane;
    p.p_bytes = (unsigned)((long long)B * g.nseg * 3 * 1024 * 4);
    const dim3 grid((unsigned)(B * g.nseg * g.ncol));
    hipStream_t s = (hipStream_t)stream;
    const int np = fgcn::math_mode() == FGCN_MATH_BF16 ? 1 : 3;
#define FGCN_EF(NP_, MU_, IC_, NSUB_)                                                                                       \
    do {                                                                                                                    \
        static bool opted = false;   /* once per instantiation; not a stream operation (stays out of graph captures) */    \
        constexpr int lds_ = ef_lds<NP_>(IC_);                                                                              \
        if (!opted) {                                                                                                       \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&emb_fwd_tile_kernel<NP_, MU_, IC_, NSUB_>),            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_);                                    \
            opted = true;                                                                                                   \
        }                                                                                                                   \
        hipLaunchKernelGGL((emb_fwd_tile_kernel<NP_, MU_, IC_, NSUB_>), grid, dim3(256), lds_, s, p);                       \
    } while (0)
#define FGCN_EF16(MU_, IC_, NSUB_, H_)                                                                                      \
    do {                                                                                                                    \
        static bool opted16 = false;                                                                                        \
        constexpr int lds_ = ef_lds<1>(IC_);                                                                                \
        if (!opted16) {                                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&emb_fwd_tile_kernel<1, MU_, IC_, NSUB_, H_>),          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_);                                    \
            opted16 = true;                                                                                                 \
        }                                                                                                                   \
        hipLaunchKernelGGL((emb_fwd_tile_kernel<1, MU_, IC_, NSUB_, H_>), grid, dim3(256), lds_, s, p);                     \
    } while (0)
#define FGCN_EF_NP(MU_, IC_, NSUB_)                    \
    do {                                               \
        if (np == 3) FGCN_EF(3, MU_, IC_, NSUB_);      \
        else if (e16 == 3) FGCN_EF16(MU_, IC_, NSUB_, 3); \
        else if (e16 == 2) FGCN_EF16(MU_, IC_, NSUB_, 2); \
        else if (e16 == 1) FGCN_EF16(MU_, IC_, NSUB_, 1); \
        else FGCN_EF(1, MU_, IC_, NSUB_);              \
    } while (0)
    if (ic == 16) FGCN_EF_NP(3, 16, 3);
    else if (ic == 32) FGCN_EF_NP(6, 32, 3);
    else FGCN_EF_NP(4, 64, 1);
#undef FGCN_EF_NP
#undef FGCN_EF16
#undef FGCN_EF
    return launch_status("emb_fwd_tile");
}
