// Fused spatial graph convolution, BACKWARD, tile form (north-star kernel 1, data-gradient side, split-bf16 math mode):
//
//     dagg_k[(n,t,w), c] = sum_o dy[(n,t,w), o] Wd_k[o][c]                      (never written to HBM)
//     dx[(n,t,v), c]    (+)= sum_k sum_w dagg_k[(n,t,w), c] A^_k[n][v][w]
//     dA^_k[n][v][w]      = sum_{t,c} x[(n,t,v), c] dagg_k[(n,t,w), c]          (partial sums per workgroup)
//
// reference: the autograd backward of SpatialGraphConv.forward, torch_src/models/mmargcn/agcn.py:103-111.
//
// The unfused chain wrote the three-activation-wide dagg with a 1x1 row GEMM (fgcn_pw_gemm) and read it back in fgcn_joint_dagg:
// 1.47 GB of HBM round trip per 245 MB activation.  Here a workgroup (8 waves, one per CU) owns F = 128 / V whole frames of one
// sample -- the forward tile kernel's geometry (fgcn_spatial_tile.hip) -- and walks the input channels in groups of 64:
//   1. dagg^T tile (192 = 3 subsets x 64 channels, by 128 rows) = Wd^T . dY^T on the matrix pipe, the dY tile staged through LDS in
//      32-channel steps as three bf16 planes (split once), the pre-split weights (fgcn_pack_split3 of the Cout x 3 Cin matrix)
//      streamed from L2.  The product is formed TRANSPOSED (weights = A operand): an accumulator lane then holds four consecutive
//      channels of one row, i.e. 8-byte pieces of a row-major image.
//   2. per 32-channel half: the owners split their accumulators and write the all-subset image [k][row][32 ch] (three bf16 planes);
//      * gram units (frame f, 16-joint tile vt of v) -> wave 2 (f mod 4) + vt: dA^_k (v tile x w) += x_f (registers, requested ahead of
//        the image barrier, split once) . dagg_kf^T (image rows, ds_read_b128); 24 accumulator registers per wave, summed over its frames;
//      * mix:  (frame, 16-channel tile) units dealt to the waves by a host-made table: dx^T (c x v) = sum_k dagg_kf^T (transposing
//        LDS reads, ds_read_b64_tr_b16: the contraction runs along image rows) . A^_k^T (split planes in LDS), stored (or added) to dx
//        as 16-byte row pieces.
// Every sum has a fixed order (bitwise reproducible); rows / joints beyond the tile or V are staged as zeros or meet zero columns
// of the A^ planes.
// One workgroup per CU runs its waves in lockstep, so every request is made a phase or more ahead: the dY rows of a contraction step two
// steps before (two register sets; 0.18 of 0.90 ms were exposed waits with one), the next group's first two steps and first weight
// fragment before the current group's halves, dx's old values before the gram (they are the mix accumulators' start), the x rows ahead of
// the image barrier.  Measured and not adopted (profiles/r04_kbench_spatial_bwd_tile.log): a second staging buffer with ONE barrier per
// step (the rows are then requested one step less ahead: 0-6 % slower); the four-wave form below.
#include <algorithm>
#include <type_traits>

#include "fgcn_common.hpp"

// Timing probes (wrong results; tools/build_probe.py only): bit 0 = no gram MFMAs, bit 1 = no mix MFMAs, bit 2 = no MFMAs in the
// contraction steps, bit 3 = x rows / old dx values not requested, bit 4 = the dY rows requested for the first step only,
// bit 5 = no image writes (the compiler then drops the contraction too), bit 6 = no dx stores, bit 7 = weight fragments requested
// once per workgroup, bit 8 = the dY rows deposited WITHOUT the three-way split (the bits of the f32 values stand in for pre-split
// planes: what the kernel would cost if its producer had written dY as bf16 planes -- VERDICT r05 item 1 -- minus the 1.5x bytes),
// bit 9 = the same for the x rows of the gram (no in-register split)
#ifndef FGCN_PROBE_SB
#define FGCN_PROBE_SB 0
#endif

namespace fgcn {

struct SpBwdP {
    const float* dy;
    const float* x;
    const float* a_hat;
    const void* w3;                     // fgcn_pack_split3 form of the Cout x (3 Cin) matrix [o][k * Cin + c] = Wd_k[o][c]
    float* dx;
    float* partial;                     // float[B][nseg][3][32][32]
    int B, T, V, Cin, Cout, ld_dy, ld_x, ld_dx, a_batched;
    int F, tiles_t, tps, nseg;          // frames per tile, tiles per sample, tiles per segment, segments per sample
    unsigned dy_bytes, x_bytes, dx_bytes, w_plane_bytes;
    // two gated addends of dx (eight-wave form, not accumulating): dx = mix + e[0] * [bit of m[0]] + e[1] * [bit of m[1]] -- contiguous
    // (B, T, V, Cin) tensors with fgcn_bn_act's sign images: the ReLU-gated gradients that reach x through the block's two identity
    // shortcuts (agcn.py:114,135), which the BatchNorm-backward kernels then neither write nor read-modify-write into dx
    const float* e[2];
    const unsigned char* m[2];
    unsigned e_bytes, m_bytes;          // bytes of a gated addend / of a sign image (one bit per element)
    // e0_grp > 0: e[0] is float[B / e0_grp][Cin], one row per GROUP of e0_grp consecutive samples, added to every row of the group's samples (the
    // gradient of the pooled output of the model's last block, fgcn_bn_act_pool: a per-clip vector instead of its (B, T, V, Cin) broadcast)
    int e0_grp;
    unsigned e0_bytes;
    int mix_wave[16];                   // mix unit u = 2 f + (16-channel tile of the 32-channel half) -> wave (eight-wave form)
};

constexpr int SB_ROWS = 144;            // 128 tile rows + the rows a 32-joint fragment of the last frame reaches past them (V = 16: 143)
constexpr int SB_XS = 64;               // bytes per image row and part (32 channels x bf16), 32-byte blocks XOR-swizzled by row bit 2
constexpr int SB_PL = SB_ROWS * SB_XS;  // one plane
constexpr int SB_AHB = 80;              // bytes per [v] row of a split A^ plane (32 joints w x bf16 + 16 pad)
constexpr int SB_IM = 3 * SB_PL;        // staging planes [3] first, then the image [3 subsets][3 parts]
constexpr int SB_AH = SB_IM + 9 * SB_PL;
constexpr int SB_LDS = SB_AH + 9 * 32 * SB_AHB;   // 133632 bytes

__device__ __forceinline__ u32x2 sb_read_tr16(const unsigned char* p) {
    using v4s = __attribute__((ext_vector_type(4))) short;
    const v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
    return __builtin_bit_cast(u32x2, v);
}

// MAXS: mix units per wave and half (the host's table: two for four to six frames per tile, up to four for seven / eight)
// NE: gated addends of dx (0 or 2; 2 only without accumulation)
// NP: bf16 parts per operand -- 3: exact three-way splits (FGCN_MATH_BF16X3), 1: operands rounded to bfloat16 once (FGCN_MATH_BF16; the LDS
// layout keeps room for three parts, the first is used)
// IN16 (NP = 1): dy is a BFLOAT16 tensor (half-precision storage written by fgcn_bn_act_bwd_apply_h; ld_dy in elements): its rows are
// copied into the staging plane instead of fetched as f32 and rounded -- the same staged bytes, half the reads
// IN16 bit 1: x is a BFLOAT16 tensor as well, bit 2: dx and the gated addends are (half-precision activation storage, the `_t` entry point; strides
// in elements; forms 0, 1, 3, 7): x's gram fragment is one 16-byte load of eight bfloat16 and needs no split, the addends / old dx values are
// 8-byte loads converted where they are consumed, dx is rounded once per store.  A per-group first addend (e0_grp) stays float32.
template <bool ACC, int MAXS, int NE = 0, int NP = 3, int IN16 = 0>
__global__ __launch_bounds__(512, 1) void spatial_bwd_tile_x3_kernel(SpBwdP p) {
    constexpr bool X16 = (IN16 & 2) != 0, H16 = (IN16 & 4) != 0;
    static_assert(NE == 0 || (NE == 2 && !ACC), "gated addends: both identity shortcuts, dx not live before");
    static_assert(NP == 1 || NP == 3, "parts");
    static_assert(!IN16 || NP == 1, "bfloat16 dy: the one-part kernel");
    constexpr int LP = 3;                                            // parts the LDS layout has room for
    constexpr unsigned OOB = 0x80000000u;
    auto swz = [](int r) -> unsigned { return (unsigned)(r & 4) << 3; };
    extern __shared__ __attribute__((aligned(16))) unsigned char sb_lds[];
    unsigned char* St = sb_lds;
    unsigned char* Im = sb_lds + SB_IM;
    unsigned char* Ah = sb_lds + SB_AH;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4, q4 = l15 >> 2, c4 = lane & 3;
    const int wm = wave >> 1, wc = wave & 1;
    const int n = blockIdx.x / p.nseg, seg = blockIdx.x - n * p.nseg;
    const int V = p.V, F = p.F, Cin = p.Cin, N3 = 3 * p.Cin;
    const int tile_lo = seg * p.tps, tile_hi = min(tile_lo + p.tps, p.tiles_t);

    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, p.w_plane_bytes * NP, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc((void*)p.dx, 0, p.dx_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rge[2], rgm[2];
    const unsigned e0_row = p.e0_grp ? (unsigned)(n / p.e0_grp) * (unsigned)p.Cin : 0u;      // element offset of this sample's group row in e[0]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        rge[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(NE ? (const void*)p.e[i] : (const void*)p.dx), 0,
                                                   NE ? ((i == 0 && p.e0_grp) ? p.e0_bytes : p.e_bytes) : 0u, 0x00020000);
        rgm[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(NE ? (const void*)p.m[i] : (const void*)p.dx), 0, NE ? p.m_bytes : 0u, 0x00020000);
    }

    // A^_k of this sample, split once per workgroup: planes [subset][part][v][w] bf16 (one ds_read_b128 = 8 joints w of row v)
    const float* asrc = p.a_hat + (p.a_batched ? (long long)n * 3 * V * V : 0);
    for (int i = tid; i < 3 * 32 * 32; i += 512) {
        const int k = i >> 10, v = (i >> 5) & 31, w = i & 31;
        const float a = (v < V && w < V) ? asrc[(k * V + v) * V + w] : 0.f;
        unsigned ph, pm, pl;
        split_bf16_pair(a, 0.f, ph, pm, pl);
        unsigned short* d = reinterpret_cast<unsigned short*>(Ah + ((k * LP) * 32 + v) * SB_AHB) + w;
        d[0] = (unsigned short)ph;
        if constexpr (NP == 3) {
            d[32 * SB_AHB / 2] = (unsigned short)pm;
            d[2 * 32 * SB_AHB / 2] = (unsigned short)pl;
        }
    }
    // rows 128 .. 143 of every staging / image plane are never written again: zero them (a fragment of the last frame reads them)
    for (int i = tid; i < 12 * 256; i += 512) {
        const int pl = i >> 8, o = i & 255;
        *reinterpret_cast<unsigned*>(sb_lds + pl * SB_PL + 128 * SB_XS + o * 4) = 0u;
    }

    // this wave's mix units of a 32-channel half (wave-uniform; the table comes from the host: fgcn_spatial_bwd_tile)
    int sf[MAXS], sct[MAXS];
    bool sok[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        sf[s] = 0;
        sct[s] = 0;
        sok[s] = false;
    }
    {
        int cnt = 0;
        for (int u = 0; u < 2 * F; ++u) {
            const bool mine = p.mix_wave[u] == wave;
#pragma unroll
            for (int s = 0; s < MAXS; ++s)
                if (mine && cnt == s) {
                    sf[s] = u >> 1;
                    sct[s] = u & 1;
                    sok[s] = true;
                }
            cnt += mine ? 1 : 0;
        }
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            sf[s] = __builtin_amdgcn_readfirstlane(sf[s]);
            sct[s] = __builtin_amdgcn_readfirstlane(sct[s]);
            sok[s] = __builtin_amdgcn_readfirstlane(sok[s] ? 1 : 0) != 0;
        }
    }

    f32x4 gacc[3][2];                    // dA^_k (this wave's v tile, w tile), summed over its frames and the workgroup's tiles and channels
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int wt = 0; wt < 2; ++wt) gacc[k][wt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int srow = tid >> 3, sg = tid & 7;                          // staging: rows srow, srow + 64; 16-byte group sg
    const int nks = p.Cout >> 5;                                     // 32-channel steps of the contraction (even: Cout % 64 == 0)

    // The dY rows of a step are requested TWO steps ahead (two register sets: one step of MFMAs, ~1.2 us, is less than an HBM round trip
    // under load -- the FGCN_PROBE_SB bit 4 probe put the exposed wait at 0.18 of 0.90 ms), across group and tile boundaries: the first
    // two steps of the NEXT group are requested before the current group's halves (gram / mix) run.
    f32x4 stg2[2][2];
    // weight ring: three fragments per contraction step.  Three parts: two slots, requested one fragment ahead (slot = step parity + index).  One
    // part (FGCN_MATH_BF16): four MFMAs per fragment instead of 24 leave the L2 latency exposed -- three slots (fragment i of a step always in slot
    // i), requested two ahead (timing probe, weights free: -0.4 of the kernel's 2.8 ms per bf16 step)
#ifndef FGCN_SB_RING_NP1
#define FGCN_SB_RING_NP1 1
#endif
#ifndef FGCN_SB_RING_NP3
#define FGCN_SB_RING_NP3 0              // (A/B builds: the three-part kernel two ahead as well -- 12 more registers at a 251-256 register kernel)
#endif
    constexpr bool WR3 = (NP == 1 && FGCN_SB_RING_NP1 != 0) || (NP == 3 && FGCN_SB_RING_NP3 != 0);
    constexpr int WD = WR3 ? 2 : 1;
    u32x4v wq[WR3 ? 3 : 2][NP];
    bool probe_w_loaded = false;
    auto fetch = [&](f32x4 (&stg)[2], int tile_, int kc) {       // kc >= Cout or no such tile: nothing (branch-free)
        const int t0_ = tile_ * F;
        const int nrows_ = (tile_ < tile_hi && kc < p.Cout) ? min(F, p.T - t0_) * V : 0;
        const unsigned row0_ = (unsigned)((n * p.T + t0_) * V);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = srow + 64 * i;
            const unsigned off = r < nrows_ ? ((row0_ + r) * (unsigned)p.ld_dy + (unsigned)(kc + 4 * sg)) * (IN16 ? 2u : 4u) : OOB;
            if constexpr (IN16) {                                // four bfloat16 = 8 bytes, parked in the first two components
                const u32x2 h = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rdy, off, 0, 0));
                const unsigned b0 = h[0], b1 = h[1];             // (element -> scalar before a bit cast: hipcc 7.2 reads element 0 otherwise)
                stg[i] = f32x4{__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1), 0.f, 0.f};
            } else {
                stg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, off, 0, 0));
            }
        }
    };
    // weight fragment of this wave's tile i (16 dagg channels) of group cg at contraction channel kc: lane (l15, g4) holds k = kc + 8 g4 + j
    // tile m = 3 wm + i of the group's 12: half m / 6, subset (m % 6) / 2, 16-channel tile m % 2 of the half
    auto load_w = [&](u32x4v (&dst)[NP], int i, int kc, int cg_) {
        const int m = 3 * wm + i;
        const int hf = m / 6, mm = m - 6 * hf;
        const int col = (mm >> 1) * Cin + cg_ * 64 + hf * 32 + (mm & 1) * 16 + l15;
        const unsigned off = (unsigned)((((kc >> 3) + g4) * N3 + col) * 16);
        if ((FGCN_PROBE_SB & 128) && probe_w_loaded) return;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) dst[pl] = __builtin_amdgcn_raw_buffer_load_b128(rw, off, pl * p.w_plane_bytes, 0);
    };
    fetch(stg2[0], tile_lo, 0);
    fetch(stg2[1], tile_lo, 32);
    load_w(wq[0], 0, 0, 0);
    if constexpr (WR3) load_w(wq[1], 1, 0, 0);
    if (FGCN_PROBE_SB & 128) {
        load_w(wq[1], 1, 0, 0);
        probe_w_loaded = true;
    }

    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        const int t0 = tile * F;
        const int nf = min(F, p.T - t0);
        const int nrows = nf * V;
        const unsigned row0 = (unsigned)((n * p.T + t0) * V);        // first row of the tile (byte offsets fit 31 bits: checked by the launcher)
        for (int cg = 0; cg < (Cin >> 6); ++cg) {
            const int cg_n = cg + 1 < (Cin >> 6) ? cg + 1 : 0, tile_n = cg_n ? tile : tile + 1;     // the group after this one
            // ---- 1. dagg^T = Wd^T . dY^T ---------------------------------------------------------------------------------------
            f32x4 acc[3][4];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto deposit = [&](const f32x4 (&stg)[2], unsigned char* St) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = srow + 64 * i;
                    u32x2 parts[NP];
                    if constexpr (IN16) {                        // already bfloat16: a copy
                        const float e0 = stg[i][0], e1 = stg[i][1];
                        parts[0] = u32x2{__builtin_bit_cast(unsigned, e0), __builtin_bit_cast(unsigned, e1)};
                    } else if constexpr ((FGCN_PROBE_SB & 256) != 0) {
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl)
                            parts[pl] = u32x2{__builtin_bit_cast(unsigned, stg[i][(pl) & 3]), __builtin_bit_cast(unsigned, stg[i][(pl + 1) & 3])};
                    } else {
                        splitn_x4<NP>(stg[i], parts);
                    }
                    unsigned char* dst = St + r * SB_XS + ((unsigned)(sg * 8) ^ swz(r));
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<u32x2*>(dst + pl * SB_PL) = parts[pl];
                }
            };
            auto load_a = [&](u32x4v (&dst)[NP], int j, const unsigned char* St) {
                const int r = wc * 64 + j * 16 + l15;
                const unsigned char* src = St + r * SB_XS + ((unsigned)(16 * g4) ^ swz(r));
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) dst[pl] = *reinterpret_cast<const u32x4v*>(src + pl * SB_PL);
            };
            u32x4v a[4][NP];
            // one 32-channel step; PB: ring slot of its first weight fragment (three fragments per step: the parity flips every step)
            auto step = [&](int ks, auto pb_tag) {
                constexpr int PB = decltype(pb_tag)::value;
                __syncthreads();                                     // the previous step's (or round's) LDS reads are done
                deposit(stg2[PB], sb_lds);
                __syncthreads();
                if (!(FGCN_PROBE_SB & 16)) fetch(stg2[PB], tile, (ks + 2) * 32);   // lands during this step's and the next one's MFMAs
#pragma unroll
                for (int j = 0; j < 4; ++j) load_a(a[j], j, sb_lds);
                auto slot = [](int t) { return WR3 ? t % 3 : (PB + t) & 1; };    // t: fragment index counted from this step's first
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    // the fragment WD ahead: a later tile of this step, or one of the next step (past the last step: the next group's first step)
                    if (i + WD < 3) load_w(wq[slot(i + WD)], i + WD, ks * 32, cg);
                    else load_w(wq[slot(i + WD)], i + WD - 3, ks + 1 < nks ? (ks + 1) * 32 : 0, ks + 1 < nks ? cg : cg_n);   // (selects, not branches)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr ((FGCN_PROBE_SB & 4) != 0) acc[i][j][0] += __builtin_bit_cast(float, wq[slot(i)][0][0] ^ a[j][0][0]);
                        else acc[i][j] = mfma_np_k32<NP>(wq[slot(i)], a[j], acc[i][j]);
                    }
                }
            };
            for (int ks = 0; ks < nks; ks += 2) {
                step(ks, std::integral_constant<int, 0>{});
                step(ks + 1, std::integral_constant<int, 1>{});
            }
            fetch(stg2[0], tile_n, 0);                               // (the last step left the next group's first weight fragment in wq[0])
            fetch(stg2[1], tile_n, 32);

            // ---- 2. the two 32-channel halves of the group ------------------------------------------------------------------------
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int cbase = cg * 64 + hf * 32;                 // first input channel of the half
                // x rows of this wave's frame (gram A operand: lane = joint v, k = 8 g4 + j channels), split once
                // gram units (frame f, v tile vt): wave 2 (f % 4) + vt takes frames f0 = wave / 2 and f0 + 4 -- its accumulators (one v tile,
                // both w tiles, three subsets: 24 registers) sum over its frames.  x rows of a unit (A operand: lane = joint v, k = 8 g4 + j
                // channels) are requested here, ahead of the image barrier (branch-free: a unit without a frame requests nothing)
                f32x4 xr[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int f = (wave >> 1) + 4 * u, v = 16 * (wave & 1) + l15;
                    const unsigned off = (f < nf && v < V && !(FGCN_PROBE_SB & 8)) ? ((row0 + f * V + v) * (unsigned)p.ld_x + cbase + 8 * g4) * (X16 ? 2u : 4u) : OOB;
                    xr[u][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));      // (X16: the eight bfloat16 of the fragment)
                    if constexpr (!X16) xr[u][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 16, 0));
                }
                // gated addends of this wave's mix units (element index of the lane's four channels; the sign image holds one bit per element:
                // a nibble per lane), requested here as well
                f32x4 ge[NE ? MAXS : 1][2][2];                       // (H16: the raw bfloat16 quads in components 0 / 1, converted at their use)
                f32x4 ge0f[(NE && H16) ? MAXS : 1][2];               // H16: the float32 per-group form of the first addend (requested beside the bfloat16 form, one of the two out of range)
                unsigned gm[NE ? MAXS : 1][2][2];
                if constexpr (NE == 2) {
#pragma unroll
                    for (int s = 0; s < MAXS; ++s)
#pragma unroll
                        for (int vt = 0; vt < 2; ++vt) {
                            const int v = 16 * vt + l15;
                            const unsigned chan = (unsigned)(cbase + sct[s] * 16 + 4 * g4);
                            const unsigned el = (row0 + sf[s] * V + v) * (unsigned)Cin + chan;
                            const bool ok = sok[s] && sf[s] < nf && v < V;
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                // (a per-group addend: the group's row instead of the element's; n is the workgroup's sample)
                                const unsigned ea = (i == 0 && p.e0_grp) ? e0_row + chan : el;
                                if constexpr (H16) {
                                    const bool grp = i == 0 && p.e0_grp;
                                    const u32x2 hq = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rge[i], (ok && !grp) ? ea * 2u : OOB, 0, 0));
                                    const unsigned b0 = hq[0], b1 = hq[1];
                                    ge[s][vt][i] = f32x4{__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1), 0.f, 0.f};
                                    if (i == 0) ge0f[s][vt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rge[0], (ok && grp) ? ea * 4u : OOB, 0, 0));
                                } else {
                                    ge[s][vt][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rge[i], ok ? ea * 4u : OOB, 0, 0));
                                }
                                gm[s][vt][i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rgm[i], ok ? el >> 3 : OOB, 0, 0) >> (el & 4u);
                            }
                        }
                }
                // the owners of this half's tiles write the image: row R = 64 wc + 16 j + l15, channels 16 (mm & 1) + 4 g4 .. + 3 of subset mm >> 1
                if ((wm >> 1) == hf && !(FGCN_PROBE_SB & 32)) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const int mm = 3 * (wm & 1) + i;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int R = wc * 64 + j * 16 + l15;
                            u32x2 parts[NP];
                            splitn_x4<NP>(acc[i][j], parts);
                            unsigned char* dst = Im + ((mm >> 1) * LP) * SB_PL + R * SB_XS + ((unsigned)(((mm & 1) * 16 + 4 * g4) * 2) ^ swz(R));
#pragma unroll
                            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<u32x2*>(dst + pl * SB_PL) = parts[pl];
                        }
                    }
                }
                __syncthreads();
                // dx's old values (or zeros) are the mix accumulators' start: rows (frame, joint v = 16 vt + l15), channels 16 ct + 4 g4 .. + 3
                f32x4 dxa[MAXS][2];
                auto dx_off = [&](int s, int vt) -> unsigned {
                    const int v = 16 * vt + l15;
                    return (sok[s] && sf[s] < nf && v < V) ? ((row0 + sf[s] * V + v) * (unsigned)p.ld_dx + cbase + sct[s] * 16 + 4 * g4) * (H16 ? 2u : 4u) : OOB;
                };
#pragma unroll
                for (int s = 0; s < MAXS; ++s)
#pragma unroll
                    for (int vt = 0; vt < 2; ++vt) {
                        if constexpr (ACC && H16) {                  // (raw bfloat16 quad in components 0 / 1; converted after the gram)
                            const u32x2 hq = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rdx, dx_off(s, vt), 0, 0));
                            const unsigned b0 = hq[0], b1 = hq[1];
                            dxa[s][vt] = f32x4{__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1), 0.f, 0.f};
                        } else if constexpr (ACC && !(FGCN_PROBE_SB & 8)) {
                            dxa[s][vt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdx, dx_off(s, vt), 0, 0));
                        } else {
                            dxa[s][vt] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    }
                // gram: dA^_k (v tile x w) += x_f . dagg_kf^T over the half's 32 channels
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int f = (wave >> 1) + 4 * u;
                    if (f >= nf) continue;                           // wave-uniform
                    u32x4v xs[NP];
                    if constexpr (X16) {                             // already bfloat16: the fragment as loaded
                        xs[0] = __builtin_bit_cast(u32x4v, xr[u][0]);
                    } else if constexpr ((FGCN_PROBE_SB & 512) != 0) {
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl) xs[pl] = __builtin_bit_cast(u32x4v, xr[u][pl & 1]);
                    } else {
                        splitn_x8<NP>(xr[u][0][0], xr[u][0][1], xr[u][0][2], xr[u][0][3], xr[u][1][0], xr[u][1][1], xr[u][1][2], xr[u][1][3], xs);
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int wt = 0; wt < 2; ++wt) {
                            const int R = f * V + 16 * wt + l15;
                            const unsigned char* src = Im + (k * LP) * SB_PL + R * SB_XS + ((unsigned)(16 * g4) ^ swz(R));
                            u32x4v bf[NP];
#pragma unroll
                            for (int pl = 0; pl < NP; ++pl) bf[pl] = *reinterpret_cast<const u32x4v*>(src + pl * SB_PL);
                            if constexpr ((FGCN_PROBE_SB & 1) != 0) gacc[k][wt][0] += __builtin_bit_cast(float, xs[0][0] ^ bf[0][0]);
                            else gacc[k][wt] = mfma_np_k32<NP>(xs, bf, gacc[k][wt]);
                        }
                }
                auto from_bf16 = [](f32x4 raw) -> f32x4 {            // components 0 / 1 hold four bfloat16
                    const float r0 = raw[0], r1 = raw[1];
                    return unpack_bf16x4(u32x2{__builtin_bit_cast(unsigned, r0), __builtin_bit_cast(unsigned, r1)});
                };
                if constexpr (NE == 2) {                             // the gated addends are the mix accumulators' start
#pragma unroll
                    for (int s = 0; s < MAXS; ++s)
#pragma unroll
                        for (int vt = 0; vt < 2; ++vt) {
                            f32x4 a0 = ge[s][vt][0], a1 = ge[s][vt][1];
                            if constexpr (H16) {
                                a0 = p.e0_grp ? ge0f[s][vt] : from_bf16(a0);
                                a1 = from_bf16(a1);
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                dxa[s][vt][e] = (((gm[s][vt][0] >> e) & 1u) ? a0[e] : 0.f) + (((gm[s][vt][1] >> e) & 1u) ? a1[e] : 0.f);
                        }
                } else if constexpr (ACC && H16) {
#pragma unroll
                    for (int s = 0; s < MAXS; ++s)
#pragma unroll
                        for (int vt = 0; vt < 2; ++vt) dxa[s][vt] = from_bf16(dxa[s][vt]);
                }
                // mix: dx^T (16 channels x 32 joints v) += sum_k dagg_kf^T (c x w) . A^_k^T (w x v)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    u32x4v af[2][NP];
#pragma unroll
                    for (int vt = 0; vt < 2; ++vt)
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl)
                            af[vt][pl] = *reinterpret_cast<const u32x4v*>(Ah + ((k * LP + pl) * 32 + 16 * vt + l15) * SB_AHB + 16 * g4);
#pragma unroll
                    for (int s = 0; s < MAXS; ++s) {
                        if (!(sok[s] && sf[s] < nf)) continue;       // wave-uniform
                        // transposed reads: the 16-lane group g4 reads rows 8 g4 + q4 (+ 4) x 16 channels, every lane receives the 8 joints
                        // w = 8 g4 .. 8 g4 + 7 of channel l15 of the tile
                        const int r_lo = sf[s] * V + 8 * g4 + q4, r_hi = r_lo + 4;
                        const unsigned cb = (unsigned)(sct[s] * 32 + 8 * c4);
                        u32x4v df[NP];
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl) {
                            const unsigned char* base = Im + (k * LP + pl) * SB_PL;
                            const u32x2 lo = sb_read_tr16(base + r_lo * SB_XS + (cb ^ swz(r_lo)));
                            const u32x2 hi = sb_read_tr16(base + r_hi * SB_XS + (cb ^ swz(r_hi)));
                            df[pl] = u32x4v{lo[0], lo[1], hi[0], hi[1]};
                        }
#pragma unroll
                        for (int vt = 0; vt < 2; ++vt) {
                            if constexpr ((FGCN_PROBE_SB & 2) != 0) dxa[s][vt][0] += __builtin_bit_cast(float, df[0][0] ^ af[vt][0][0]);
                            else dxa[s][vt] = mfma_np_k32<NP>(df, af[vt], dxa[s][vt]);
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < MAXS; ++s)
#pragma unroll
                    for (int vt = 0; vt < 2; ++vt) {
                        if constexpr (H16) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pack_bf16(dxa[s][vt])), rdx, dx_off(s, vt), 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, dxa[s][vt]), rdx, (FGCN_PROBE_SB & 64) ? OOB : dx_off(s, vt), 0, 0);
                    }
                __syncthreads();                                     // the image is free for the next half / the next group
            }
        }
    }

    // ---- the workgroup's dA^ partial: fixed-order sum of the eight waves' frames --------------------------------------------------
    __syncthreads();
    float* red = reinterpret_cast<float*>(sb_lds);                   // [8 waves][3][32 v][32 w]
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int vt = 0; vt < 2; ++vt)
#pragma unroll
            for (int wt = 0; wt < 2; ++wt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[(wave * 3 + k) * 1024 + (16 * vt + 4 * g4 + r) * 32 + 16 * wt + l15] = vt == (wave & 1) ? gacc[k][wt][r] : 0.f;
    __syncthreads();
    float* dst = p.partial + ((long long)n * p.nseg + seg) * 3 * 1024;
    for (int e = tid; e < 3 * 1024; e += 512) {
        const int k = e >> 10, idx = e & 1023;
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) s += red[(wv * 3 + k) * 1024 + idx];
        dst[e] = ((idx >> 5) < V && (idx & 31) < V) ? s : 0.f;
    }
}

// (A variant as TWO four-wave workgroups per CU -- one 32-channel half per group, one subset in the image at a time, 75 KB of LDS -- was built
// in round 4, parity-tested and measured 8-22 % slower at every shape (DESIGN_HISTORY.md section 3.10); removed in round 6.)

}  // namespace fgcn

using namespace fgcn;

// 1 when fgcn_spatial_bwd_tile runs these sizes in the current math mode (FGCN_MATH_BF16X3 with either product form: the kernel always
// multiplies three-way bf16 splits and takes the fgcn_pack_split3 form of the weights; whole 64-channel input groups, 64-channel steps of the
// contraction; 16..32 joints: at most 8 frames per 128-row tile)
extern "C" int fgcn_spatial_bwd_tile_available(int V, int Cin, int Cout) {
    return ((fgcn::math_mode() == FGCN_MATH_BF16X3 || fgcn::math_mode() == FGCN_MATH_BF16) && V >= 16 && V <= FGCN_MAX_V && Cin % 64 == 0 && Cin > 0 &&
            Cout % 64 == 0 && Cout > 0) ? 1 : 0;
}

// segments per sample = partial matrices per sample: one workgroup per CU (256 in all) when the batch allows it -- every workgroup pays the
// split of the adjacency and the cross-wave sum of its partial matrices once, and a single round of workgroups measured fastest at every
// batch size (8 clips 9.41 -> 9.18 ms a step, 16: 16.23 -> 16.03, 32: 29.41 -> 29.21, 64 within noise; profiles/r04_ab_spatial_bwd_segments.txt);
// tuning key 15 overrides the target
extern "C" int fgcn_spatial_bwd_tile_segments(int B, int T, int V) {
    if (V < 16 || V > FGCN_MAX_V || B <= 0 || T <= 0) return 0;
    const int tiles_t = (int)cdiv(T, 128 / V);
    const int want = (int)std::max<long long>(1, cdiv(fgcn::tuning(15) > 0 ? fgcn::tuning(15) : 256, B));
    const int tps = (int)cdiv(tiles_t, std::min(tiles_t, want));
    return (int)cdiv(tiles_t, tps);
}

static int spatial_bwd_tile_launch(const float* dy, const float* x, const float* a_hat, const void* w3, float* dx, float* partial,
                                   int B, int T, int V, int Cin, int Cout, int ld_dy, int ld_x, int ld_dx, int a_hat_batched,
                                   int accumulate, const float* extra1, const unsigned char* mask1, const float* extra2,
                                   const unsigned char* mask2, int extra1_group, void* stream, int dy16 = 0);

// dy as a BFLOAT16 tensor (math mode bf16 only; ld_dy in elements): fgcn_spatial_bwd_tile (extra1_group = 0) / fgcn_spatial_bwd_tile_g
// otherwise unchanged; bit-identical to the f32-dy call on the tensor fgcn_bn_act_bwd_apply would have written
extern "C" int fgcn_spatial_bwd_tile_h(const unsigned short* dy_h, const float* x, const float* a_hat, const void* w3, float* dx, float* partial,
                                       int B, int T, int V, int Cin, int Cout, int ld_dy, int ld_x, int ld_dx, int a_hat_batched, int accumulate,
                                       const float* extra1, int extra1_group, const unsigned char* mask1, const float* extra2,
                                       const unsigned char* mask2, void* stream) {
    FGCN_REQUIRE(extra1_group >= 0 && (extra1_group == 0 || (extra1 && B % extra1_group == 0 && !accumulate)), FGCN_E_BADARG,
                 "spatial_bwd_tile_h: %d samples are not whole groups of %d", B, extra1_group);
    return spatial_bwd_tile_launch(reinterpret_cast<const float*>(dy_h), x, a_hat, w3, dx, partial, B, T, V, Cin, Cout, ld_dy, ld_x, ld_dx,
                                   a_hat_batched, accumulate, extra1, mask1, extra2, mask2, extra1_group, stream, 1);
}

// typed form (math mode bf16): half_mask bit 0 = dy is a bfloat16 tensor, bit 1 = x is, bit 2 = dx AND the gated addends are (a per-group extra1
// stays float32); masks 0, 1, 3, 7.  Strides in elements.
extern "C" int fgcn_spatial_bwd_tile_t(const void* dy, const void* x, const float* a_hat, const void* w3, void* dx, float* partial,
                                       int B, int T, int V, int Cin, int Cout, int ld_dy, int ld_x, int ld_dx, int a_hat_batched, int accumulate,
                                       const void* extra1, int extra1_group, const unsigned char* mask1, const void* extra2,
                                       const unsigned char* mask2, int half_mask, void* stream) {
    FGCN_REQUIRE(half_mask == 0 || half_mask == 1 || half_mask == 3 || half_mask == 7, FGCN_E_BADARG, "spatial_bwd_tile_t: half_mask=%d (0, 1, 3 or 7)",
                 half_mask);
    FGCN_REQUIRE(extra1_group >= 0 && (extra1_group == 0 || (extra1 && B % extra1_group == 0 && !accumulate)), FGCN_E_BADARG,
                 "spatial_bwd_tile_t: %d samples are not whole groups of %d", B, extra1_group);
    return spatial_bwd_tile_launch(static_cast<const float*>(dy), static_cast<const float*>(x), a_hat, w3, static_cast<float*>(dx), partial, B, T, V,
                                   Cin, Cout, ld_dy, ld_x, ld_dx, a_hat_batched, accumulate, static_cast<const float*>(extra1), mask1,
                                   static_cast<const float*>(extra2), mask2, extra1_group, stream, half_mask);
}

extern "C" int fgcn_spatial_bwd_tile(const float* dy, const float* x, const float* a_hat, const void* w3, float* dx, float* partial,
                                     int B, int T, int V, int Cin, int Cout, int ld_dy, int ld_x, int ld_dx, int a_hat_batched,
                                     int accumulate, const float* extra1, const unsigned char* mask1, const float* extra2,
                                     const unsigned char* mask2, void* stream) {
    return spatial_bwd_tile_launch(dy, x, a_hat, w3, dx, partial, B, T, V, Cin, Cout, ld_dy, ld_x, ld_dx, a_hat_batched, accumulate, extra1, mask1,
                                   extra2, mask2, 0, stream);
}

// the same with the first gated addend given per GROUP of `extra1_group` consecutive samples: extra1 is float[B / extra1_group][Cin]
extern "C" int fgcn_spatial_bwd_tile_g(const float* dy, const float* x, const float* a_hat, const void* w3, float* dx, float* partial,
                                       int B, int T, int V, int Cin, int Cout, int ld_dy, int ld_x, int ld_dx, int a_hat_batched,
                                       const float* extra1, int extra1_group, const unsigned char* mask1, const float* extra2,
                                       const unsigned char* mask2, void* stream) {
    FGCN_REQUIRE(extra1 && extra1_group > 0 && B % extra1_group == 0, FGCN_E_BADARG, "spatial_bwd_tile_g: %d samples are not whole groups of %d", B,
                 extra1_group);
    return spatial_bwd_tile_launch(dy, x, a_hat, w3, dx, partial, B, T, V, Cin, Cout, ld_dy, ld_x, ld_dx, a_hat_batched, 0, extra1, mask1, extra2,
                                   mask2, extra1_group, stream);
}

static int spatial_bwd_tile_launch(const float* dy, const float* x, const float* a_hat, const void* w3, float* dx, float* partial,
                                   int B, int T, int V, int Cin, int Cout, int ld_dy, int ld_x, int ld_dx, int a_hat_batched,
                                   int accumulate, const float* extra1, const unsigned char* mask1, const float* extra2,
                                   const unsigned char* mask2, int extra1_group, void* stream, int dy16) {      // dy16: the typed entry's half_mask (0, 1, 3, 7)
    const bool gated = extra1 != nullptr;
    FGCN_REQUIRE(!dy16 || fgcn::math_mode() == FGCN_MATH_BF16, FGCN_E_BADARG, "spatial_bwd_tile_h: a bfloat16 dy needs math mode bf16");
    FGCN_REQUIRE(!gated || (mask1 && extra2 && mask2 && !accumulate && ld_x == Cin && Cin % 8 == 0), FGCN_E_BADARG,
                 "spatial_bwd_tile: gated addends come in pairs with their sign images, without accumulation, on contiguous (B, T, V, Cin) tensors");
    FGCN_REQUIRE(!gated || (aligned16(extra1) && aligned16(extra2)), FGCN_E_ALIGN, "spatial_bwd_tile: 16-byte aligned addends");
    FGCN_REQUIRE(dy && x && a_hat && w3 && dx && partial, FGCN_E_BADARG, "spatial_bwd_tile: null pointer");
    FGCN_REQUIRE(B > 0 && T > 0, FGCN_E_BADARG, "spatial_bwd_tile: bad sizes B=%d T=%d", B, T);
    FGCN_REQUIRE(fgcn_spatial_bwd_tile_available(V, Cin, Cout), FGCN_E_BADARG,
                 "spatial_bwd_tile: needs math mode bf16x3 or bf16, 16 <= V <= %d, Cin %% 64 == 0, Cout %% 64 == 0 (V=%d Cin=%d Cout=%d)",
                 FGCN_MAX_V, V, Cin, Cout);
    FGCN_REQUIRE(ld_dy % 4 == 0 && ld_x % 4 == 0 && ld_dx % 4 == 0 && ld_dy >= Cout && ld_x >= Cin && ld_dx >= Cin, FGCN_E_ALIGN,
                 "spatial_bwd_tile: row strides");
    FGCN_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(w3) && aligned16(dx), FGCN_E_ALIGN, "spatial_bwd_tile: 16-byte alignment");
    const long long rows = (long long)B * T * V;
    const int abytes = (dy16 & 4) ? 2 : 4;
    const long long dy_bytes = rows * ld_dy * (dy16 ? 2 : 4), x_bytes = rows * ld_x * ((dy16 & 2) ? 2 : 4), dx_bytes = rows * ld_dx * abytes;
    const long long plane = (long long)3 * Cin * Cout * 2;
    FGCN_REQUIRE(dy_bytes < 0x7FFF0000ll && x_bytes < 0x7FFF0000ll && dx_bytes < 0x7FFF0000ll && plane * 3 < 0x7FFF0000ll, FGCN_E_BADARG,
                 "spatial_bwd_tile: tensors must be smaller than 2 GiB (32-bit buffer offsets)");
    SpBwdP p;
    p.dy = dy; p.x = x; p.a_hat = a_hat; p.w3 = w3; p.dx = dx; p.partial = partial;
    p.B = B; p.T = T; p.V = V; p.Cin = Cin; p.Cout = Cout; p.ld_dy = ld_dy; p.ld_x = ld_x; p.ld_dx = ld_dx; p.a_batched = a_hat_batched;
    p.F = 128 / V;
    p.tiles_t = (int)cdiv(T, p.F);
    p.nseg = fgcn_spatial_bwd_tile_segments(B, T, V);
    p.tps = (int)cdiv(p.tiles_t, p.nseg);
    FGCN_REQUIRE((long long)B * p.nseg < (1ll << 30), FGCN_E_BADARG, "spatial_bwd_tile: too many workgroups");
    p.dy_bytes = (unsigned)dy_bytes; p.x_bytes = (unsigned)x_bytes; p.dx_bytes = (unsigned)dx_bytes; p.w_plane_bytes = (unsigned)plane;
    p.e[0] = extra1; p.e[1] = extra2; p.m[0] = mask1; p.m[1] = mask2; p.e_bytes = (unsigned)(rows * Cin * abytes);
    p.m_bytes = (unsigned)(rows * Cin / 8);
    p.e0_grp = extra1_group;
    p.e0_bytes = extra1_group ? (unsigned)((long long)(B / extra1_group) * Cin * 4) : 0u;
    // mix units (frame, 16-channel tile of a 32-channel half) -> waves: wave 2 (f mod 4) + vt already carries the gram units (f, vt) -- one or
    // two per half, 12 MFMA groups each like a mix unit; greedy on the lightest wave, the later wave on ties
    {
        int load[8], cnt[8];
        for (int w = 0; w < 8; ++w) {
            load[w] = (w >> 1) < p.F ? ((w >> 1) + 4 < p.F ? 2 : 1) : 0;
            cnt[w] = 0;
        }
        for (int u = 0; u < 16; ++u) p.mix_wave[u] = -1;
        for (int u = 0; u < 2 * p.F; ++u) {
            int best = -1;
            for (int w = 7; w >= 0; --w)
                if (cnt[w] < 4 && (best < 0 || load[w] < load[best])) best = w;
            FGCN_REQUIRE(best >= 0, FGCN_E_BADARG, "spatial_bwd_tile: no wave left for mix unit %d", u);
            p.mix_wave[u] = best;
            load[best] += 1;
            cnt[best] += 1;
        }
    }
    const dim3 grid((unsigned)(B * p.nseg));
    hipStream_t s = (hipStream_t)stream;
    const bool one_part = fgcn::math_mode() == FGCN_MATH_BF16;     // operands rounded to bfloat16 once
#define FGCN_SB_GO4(ACC_, MS_, NE_, NP_, I16_)                                                                         \
    do {                                                                                                                \
        static bool opted = false;   /* once per instantiation; not a stream operation (stays out of graph captures) */ \
        if (!opted) {                                                                                                   \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_bwd_tile_x3_kernel<ACC_, MS_, NE_, NP_, I16_>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)SB_LDS);                         \
            opted = true;                                                                                               \
        }                                                                                                               \
        hipLaunchKernelGGL((spatial_bwd_tile_x3_kernel<ACC_, MS_, NE_, NP_, I16_>), grid, dim3(512), SB_LDS, s, p);     \
    } while (0)
#define FGCN_SB_GO3(ACC_, MS_, NE_)                                                                                    \
    do {                                                                                                                \
        if (one_part && dy16 == 7) FGCN_SB_GO4(ACC_, MS_, NE_, 1, 7);                                                   \
        else if (one_part && dy16 == 3) FGCN_SB_GO4(ACC_, MS_, NE_, 1, 3);                                              \
        else if (one_part && dy16) FGCN_SB_GO4(ACC_, MS_, NE_, 1, 1);                                                   \
        else if (one_part) FGCN_SB_GO4(ACC_, MS_, NE_, 1, 0);                                                           \
        else FGCN_SB_GO4(ACC_, MS_, NE_, 3, 0);                                                                         \
    } while (0)
#define FGCN_SB_GO(ACC_, MS_)                                                                                          \
    do {                                                                                                                \
        if (gated) FGCN_SB_GO3(false, MS_, 2);                                                                          \
        else FGCN_SB_GO3(ACC_, MS_, 0);                                                                                 \
    } while (0)
    int max_units = 0;
    for (int w = 0; w < 8; ++w) {
        int c = 0;
        for (int u = 0; u < 2 * p.F; ++u) c += p.mix_wave[u] == w ? 1 : 0;
        max_units = std::max(max_units, c);
    }
    if (max_units <= 2) {
        if (accumulate) FGCN_SB_GO(true, 2);
        else FGCN_SB_GO(false, 2);
    } else if (max_units <= 3) {
        if (accumulate) FGCN_SB_GO(true, 3);
        else FGCN_SB_GO(false, 3);
    } else {
        if (accumulate) FGCN_SB_GO(true, 4);
        else FGCN_SB_GO(false, 4);
    }
#undef FGCN_SB_GO
#undef FGCN_SB_GO3
#undef FGCN_SB_GO4
    return launch_status("spatial_bwd_tile");
}
