// Fused spatial graph convolution, forward, TILE form (north-star kernel 1 in the split-bf16 math mode):
//
//     y[(n,t,w), o] = bias[o] + sum_k sum_c Wd_k[o][c] * ( sum_v x[(n,t,v), c] * A^_k[n][v][w] )
//
// reference: SpatialGraphConv.forward, torch_src/models/mmargcn/agcn.py:103-111; still ONE kernel and no aggregation tensor in HBM.
//
// Why a second form.  spatial_fwd_x3_kernel (fgcn_spatial.hip) gives every wave two frames: its feature GEMM runs on 25 of 32 matrix
// columns (the joints of ONE frame per 32-column tile), the aggregation is re-formed for every 64-column block of outputs, and each
// wave streams its own weight fragments from L2 with one unit of MFMAs to cover the load.  Timing probes (tools/build_probe.py,
// profiles/r03_probe_pw_spatial.txt) put that kernel at 0.92 ms for 256 -> 256 channels with the aggregation (a third of its MFMAs)
// costing 0.43 ms of it.  Here the feature GEMM is the halo conv's (fgcn_tconv.hip): a workgroup owns F = 128 / V whole frames of
// one sample -- F V rows, 125 of 128 at V = 25 -- times 64 NT output columns; per pair (32-channel tile ci, subset k) the
// aggregation tile agg_k[(f, w)][c] is formed ONCE per workgroup on the matrix pipe (split-bf16, as before), split into its three
// bf16 parts and written to an LDS image [row f V + w][32 channels]; the feature contraction then runs from that image exactly like
// one tap of the temporal conv: 2 x 2 waves over 128 rows x 64 NT columns, image fragments by ds_read_b128, pre-split weights
// (fgcn_pack_split3, the (3 Cin) x Cout matrix of the three conv_d weights) streamed from L2 through a two-slot ring, each fragment
// feeding four row tiles.  Two pairs are staged per barrier pair (the image of all three subsets of a channel tile would not leave
// room for two workgroups per CU beside the A^ planes).
//   aggregation units of a chunk = (pair, frame): 2 F units, dealt round-robin to the four waves; a unit is 2 x 6 MFMAs
//   (v_mfma_f32_32x32x16_bf16: agg^T (32 c x 32 w) = X_t^T . A^_k), its x rows are requested one chunk ahead and parked in registers.
// Rows of the image beyond the tile's F V rows are never written and never stored (each output row depends on its own image row only).
#include "fgcn_common.hpp"

// Timing probes (wrong results; tools/build_probe.py only): bit 0 = the image is staged for the first chunk only, bit 1 = no feature
// MFMAs, bit 2 = x is fetched for the first chunk only, bit 3 = no aggregation MFMAs / splits (the image receives x's split instead)
#ifndef FGCN_ST_RING
#define FGCN_ST_RING 4
#endif
#ifndef FGCN_PROBE_ST
#define FGCN_PROBE_ST 0
#endif

namespace fgcn {

struct SpTileP {
    const float* x;
    const float* a_hat;
    const void* w3;                     // fgcn_pack_split3 form of the (3 Cin) x Cout matrix: [part][(k Cin + c) / 8][o][8] bf16
    const float* bias;
    float* y;
    float* stats;                       // float[tiles_m][2][Cout] or NULL
    int B, T, V, Cin, Cout, ld_x, ld_y, a_batched;
    int F, tiles_t, tiles_m, tiles_n, per_xcd;
    unsigned x_bytes, y_bytes, w_plane_bytes;
    // Inference epilogue (spatial_tile_x3_kernel<.., FEP = true>; fgcn_spatial_fwd_tile_bn_relu): with eval-mode BatchNorm the statistics are
    // constants, so BatchNorm + shortcut + ReLU of the graph convolution (agcn.py:113-115) are the kernel's epilogue and `y` receives
    // G = relu((acc + bias) * scale + shift + res * rscale + rshift) -- no pre-BatchNorm tensor, no bn_act pass.
    const float* ep_vec;                // float[4][Cout] of fgcn_bn_eval_coeffs (scale at [2 Cout, 3 Cout), shift at [3 Cout, 4 Cout))
    const float* ep_res;                // the shortcut operand, rows of ld_res floats (x for an identity block, the down conv's output), or NULL
    const float* ep_rvec;               // float[4][Cout]: BatchNorm of the shortcut (the down branch), or NULL (identity)
    int ld_res;
    unsigned res_bytes;
};

constexpr int ST_AHB = 80;              // bytes per [w] row of a split A^ plane (32 joints x bf16 + 16 pad: conflict-free b128 reads)
constexpr int ST_XS = 64;               // bytes per image row and part (32 channels x bf16), 32-byte blocks XOR-swizzled by row bit 2
constexpr int ST_PLANE = 256 * ST_XS;   // one part of the image: two pairs x 128 rows

// STR: non-temporal output stores (fgcn_common.hpp, stream_out); NP: bf16 parts per operand -- 3: exact three-way splits (FGCN_MATH_BF16X3),
// 1: operands rounded to bfloat16 once (FGCN_MATH_BF16; the LDS layout keeps room for three parts, the first is used)
// FEP: the inference epilogue (SpTileP::ep_*) instead of bias + BatchNorm partial sums
// IO (NP = 1, training epilogue): bit 0 = x is a BFLOAT16 tensor, bit 1 = y is (half-precision activation storage in math mode bf16, the `_t`
// entry point; ld_x / ld_y in elements).  A bfloat16 x gives the same staged bytes as the float32 tensor of the same values (the one-part
// kernel rounds x to bfloat16 anyway); a bfloat16 y is the rounded float32 result, the BatchNorm partial sums stay those of the accumulators.
template <int NT, int MAXU, bool STR = false, int NP = 3, bool FEP = false, int IO = 0>
__global__ __launch_bounds__(256, 2) void spatial_tile_x3_kernel(SpTileP p) {
    constexpr int LP = 3, MTW = 4, NU = 2 * NT, BN = 64 * NT;
    static_assert(NP == 1 || NP == 3, "parts");
    static_assert(IO == 0 || (NP == 1 && !FEP), "bfloat16 tensors: the one-part kernel's training form");
    constexpr bool X16 = (IO & 1) != 0, O16 = (IO & 2) != 0;
    constexpr unsigned OOB = 0x80000000u;
    auto swz = [](int r) -> unsigned { return (unsigned)(r & 4) << 3; };
    extern __shared__ __attribute__((aligned(16))) float smem_st[];
    unsigned char* Xh = reinterpret_cast<unsigned char*>(smem_st);   // [3 parts][2 pairs x 128 rows][64 B]
    unsigned char* ahs = Xh + LP * ST_PLANE;                         // [3 subsets][3 parts][32 w][ST_AHB]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4, l31 = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware order, column tile fastest: the column tiles of a row tile (same x rows, same A^) run back to back on one XCD
    const int vid = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
    if (vid >= p.tiles_m * p.tiles_n) return;
    const int bm = vid / p.tiles_n, bn = vid - bm * p.tiles_n;
    const int n = bm / p.tiles_t, tf = bm - n * p.tiles_t;
    const int V = p.V, F = p.F;
    const int t0 = tf * F;
    const int nf = min(F, p.T - t0);                                 // frames of this tile
    const int nrows = nf * V;
    const int n0 = bn * BN;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, p.w_plane_bytes * NP, 0x00020000);

    // A^_k of this sample, split once per workgroup: [subset][part][w][v] bf16 (one ds_read_b128 = the 8 joints of a lane's fragment).
    // All twelve requests of a thread are in flight at once (branch-free buffer loads; absent joints carry the out-of-range offset): as a
    // loop of conditional loads they were twelve dependent round trips at the start of EVERY 128-row tile (round 5: the same pattern cost
    // the embedding-backward kernel 22 % of its time, profiles/r05_kbench_emb_bwd_variants.txt).
    {
        const float* asrc = p.a_hat + (p.a_batched ? (long long)n * 3 * V * V : 0);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)asrc, 0, (unsigned)(3 * V * V) * 4u, 0x00020000);
        float av[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            const int i = tid + 256 * e;
            const int k = e >> 2, w = (i >> 5) & 31, v = i & 31;       // (256 threads: subset = e / 4)
            av[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, (v < V && w < V) ? (unsigned)((k * V + v) * V + w) * 4u : OOB, 0, 0));
        }
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            const int i = tid + 256 * e;
            const int k = e >> 2, w = (i >> 5) & 31, v = i & 31;
            unsigned ph, pm, pl;
            split_bf16_pair(av[e], 0.f, ph, pm, pl);
            unsigned short* d = reinterpret_cast<unsigned short*>(ahs + ((k * LP) * 32 + w) * ST_AHB) + v;
            d[0] = (unsigned short)ph;
            if constexpr (NP == 3) {
                d[32 * ST_AHB / 2] = (unsigned short)pm;
                d[2 * 32 * ST_AHB / 2] = (unsigned short)pl;
            }
        }
    }

    // this wave's aggregation units of a chunk: the 2 F units in frame-major order (f0 q0, f0 q1, f1 q0, ...) are dealt to the waves in
    // contiguous blocks, so the two pairs of a frame mostly land on ONE wave -- when they share the channel tile (two chunks of three) that
    // wave loads and splits the frame's x rows once for both
    int uq[MAXU], uf[MAXU];
    bool uok[MAXU], ushare[MAXU];
    const int u_lo = __builtin_amdgcn_readfirstlane((wave * 2 * F) >> 2), u_hi = __builtin_amdgcn_readfirstlane(((wave + 1) * 2 * F) >> 2);
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        const int u = u_lo + i;
        uq[i] = u & 1;
        uf[i] = u >> 1;
        uok[i] = u < u_hi && uf[i] < nf;                             // (wave-uniform)
        ushare[i] = i > 0 && uok[i] && uok[i - 1] && uf[i] == uf[i - 1];   // same frame as the previous unit (then q = 1 after q = 0)
    }
    const int npairs = 3 * (p.Cin >> 5);                             // (channel tile, subset) pairs; even (Cin % 64 == 0)
    const int nchunks = npairs >> 1;
    const unsigned row_b = (unsigned)p.ld_x * (X16 ? 2u : 4u);
    // the x rows of chunk c + 1 are requested at the start of chunk c's feature phase and parked in registers.  (A second register set,
    // requested a phase earlier so that the loads are not queued in front of the phase's weight fragments in the in-order vmcnt queue,
    // measured the same and cost 48 registers: FGCN_PROBE_ST puts the fetches at 0.12-0.17 ms of a 0.8 ms launch either way.)
    float xrA[MAXU][16];
    // x of (frame, channel tile): lane = channel, register 8 s + j = joint 16 s + 8 h + j (the k order of the 32x32x16 fragment); the
    // joint's row offset splits into a per-lane part (8 h rows) and a scalar part (16 s + j rows, the instruction's soffset)
    auto fetch_units = [&](int c, float (&xr)[MAXU][16]) {
        const bool same_ci = (2 * c + 1) % 3 != 0;                   // both pairs of chunk c read the same channel tile
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            if (ushare[i] && same_ci) continue;                      // (wave-uniform) the previous unit's rows serve this one too
            const int pq = 2 * c + uq[i];
            const int ci = pq / 3;
            const unsigned base = (unsigned)((((long long)n * p.T + t0 + (uok[i] ? uf[i] : 0)) * V + 8 * h) * p.ld_x + ci * 32 + l31) * (X16 ? 2u : 4u);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int vs = 16 * (r >> 3) + (r & 7);
                if constexpr (X16)       // one bfloat16 -> the float it stands for (the split below rounds it back to the same 16 bits)
                    xr[i][r] = __builtin_bit_cast(float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(
                                                             rx, (uok[i] && vs + 8 * h < V) ? base : OOB, (unsigned)vs * row_b, 0) << 16);
                else
                    xr[i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, (uok[i] && vs + 8 * h < V) ? base : OOB,
                                                                                              (unsigned)vs * row_b, 0));
            }
        }
    };
    const unsigned char* af_lane = ahs + l31 * ST_AHB + 16 * h;      // + (k * NP + part) * 32 * ST_AHB + 32 * s2
    auto stage_units = [&](int c, float (&xr)[MAXU][16]) {
        const bool same_ci = (2 * c + 1) % 3 != 0;
        u32x4v xs[2][NP];                                            // the split x rows of the current frame (kept across a shared pair)
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            if (!uok[i]) continue;                                   // wave-uniform
            const int pq = 2 * c + uq[i];
            const int k = pq - 3 * (pq / 3);
            f32x16 agg = zero16();
            if (!(ushare[i] && same_ci)) {                           // wave-uniform
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    splitn_x8<NP>(xr[i][8 * s2], xr[i][8 * s2 + 1], xr[i][8 * s2 + 2], xr[i][8 * s2 + 3], xr[i][8 * s2 + 4], xr[i][8 * s2 + 5],
                                  xr[i][8 * s2 + 6], xr[i][8 * s2 + 7], xs[s2]);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4v af[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    af[pl] = *reinterpret_cast<const u32x4v*>(af_lane + (k * LP + pl) * 32 * ST_AHB + 32 * s2);
                agg = mfma_np_k16<NP>(xs[s2], af, agg);                  // agg^T (32 c x 32 w): lane = joint w, register r = channel acc_row(r)
            }
            // this lane's joint w = l31 of frame uf: image row uq * 128 + uf * V + w, channels 8 g + 4 h + (0..3) per register group
            const int R = uq[i] * 128 + uf[i] * V + l31;
            if (l31 < V) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 parts[NP];
                    splitn_x4<NP>(f32x4{agg[4 * g], agg[4 * g + 1], agg[4 * g + 2], agg[4 * g + 3]}, parts);
                    unsigned char* dst = Xh + R * ST_XS + ((unsigned)(16 * g + 8 * h) ^ swz(R));
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<u32x2*>(dst + pl * ST_PLANE) = parts[pl];
                }
            }
        }
    };

    f32x4 acc[MTW][NU];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nu = 0; nu < NU; ++nu) acc[mt][nu] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int col = n0 + wc * NT * 32 + l15;                         // + nu * 16
    unsigned wvoff[NU];                                              // per-lane byte offset into one part: (g4 * N + col) * 8 bf16
#pragma unroll
    for (int nu = 0; nu < NU; ++nu) wvoff[nu] = col + nu * 16 < p.Cout ? (unsigned)(((long long)g4 * p.Cout + col + nu * 16) * 16) : OOB;
    // weight fragment of (column unit nu, pair pq): contraction rows k * Cin + 32 ci + 8 g4 + j; past the last pair: pair 0 (a valid, unused load)
    auto load_w = [&](u32x4v (&dst)[NP], int nu, int pq) {
        if (pq >= npairs) pq = 0;
        const int ci = pq / 3, k = pq - 3 * ci;
        const unsigned so = (unsigned)(((long long)((k * p.Cin + 32 * ci) >> 3) * p.Cout) * 16);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) dst[pl] = __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff[nu], so + pl * p.w_plane_bytes, 0);
    };
    const int xrow = wr * (16 * MTW) + l15;
    auto load_a = [&](u32x4v (&dst)[NP], int mt, int q) {
        const int r = q * 128 + xrow + mt * 16;
        const unsigned char* src = Xh + r * ST_XS + ((unsigned)(16 * g4) ^ swz(r));
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) dst[pl] = *reinterpret_cast<const u32x4v*>(src + pl * ST_PLANE);
    };

    constexpr int RS = (FGCN_ST_RING == 4 && NU == 4) ? 4 : 2;      // weight ring slots: fragments requested RS - 1 units ahead (see fgcn_pw.hip)
    u32x4v a[MTW][NP], wq[RS][NP];
    auto feature_phase = [&](int c) {                                // the two pairs of chunk c from the image
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) load_a(a[mt], mt, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pq = 2 * c + q;
#pragma unroll
            for (int nu = 0; nu < NU; ++nu) {
                const int t = nu + RS - 1;
                if (t < NU) load_w(wq[t % RS], t, pq);
                else load_w(wq[t % RS], t - NU, pq + 1);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    if constexpr ((FGCN_PROBE_ST & 2) != 0) acc[mt][nu][0] += __builtin_bit_cast(float, a[mt][0][0] ^ wq[nu % RS][0][0]);
                    else acc[mt][nu] = mfma_np_k32<NP>(a[mt], wq[nu % RS], acc[mt][nu]);
                    if (nu == NU - 1 && q == 0) load_a(a[mt], mt, 1);    // this fragment's last use: fetch the next step's
                }
            }
        }
    };
    fetch_units(0, xrA);
#pragma unroll
    for (int nu = 0; nu < RS - 1; ++nu) load_w(wq[nu], nu, 0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();                                             // the previous chunk's image reads are done (first pass: the A^ planes are written)
        if (!(FGCN_PROBE_ST & 1) || c == 0) stage_units(c, xrA);
        __syncthreads();
        if (c + 1 < nchunks && !(FGCN_PROBE_ST & 4)) fetch_units(c + 1, xrA);   // lands during the MFMAs below
        feature_phase(c);
    }

    // ---- epilogue: bias, branch-free buffer stores, BatchNorm partial sums (accumulator register r of lane (col l15, g4) = row
    // 4 g4 + r of its 16 x 16 tile); rows beyond the tile's frames carry the out-of-range offset ----------------------------------
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? (const void*)p.bias : p.w3), 0,
                                                                           p.bias ? (unsigned)p.Cout * 4u : 0u, 0x00020000);
    const long long m0 = ((long long)n * p.T + t0) * V;
    float ssum[NU], ssq[NU], bv[NU];
    unsigned coff[NU];
#pragma unroll
    for (int nu = 0; nu < NU; ++nu) {
        ssum[nu] = 0.f;
        ssq[nu] = 0.f;
        coff[nu] = col + nu * 16 < p.Cout ? (unsigned)(col + nu * 16) * 4u : OOB;
        bv[nu] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, coff[nu], 0, 0));
    }
    // Store order: the column units of a row group back to back (a store instruction covers 16 columns = 64 bytes of four rows; units nu, nu + 1
    // are the two halves of a 128-byte line).  With the unit loop outside the row loop the halves of a line were four stores apart and the
    // streamed (non-temporal) form wrote 1.45-1.48x the output's bytes -- WRITE_SIZE 0.366 GB per launch for a 0.246 GB tensor, 0.250 GB with
    // plain stores (profiles/r05_pmc_spatial_tile_writes.txt): half-lines left the L2 one by one.  Each unit's sums still run over (mt, r) in
    // the same order: same bits.
    if constexpr (FEP) {
        // G = relu((acc + bias) * scale + shift + res * rscale + rshift): per-column constants first, the shortcut values of a row tile
        // requested one row tile ahead of its stores (two register sets)
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)p.ep_vec, 0, (unsigned)p.Cout * 16u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rrv = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ep_rvec ? p.ep_rvec : p.ep_vec), 0,
                                                                             p.ep_rvec ? (unsigned)p.Cout * 16u : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ep_res ? (const void*)p.ep_res : (const void*)p.y), 0,
                                                                              p.ep_res ? p.res_bytes : 0u, 0x00020000);
        float esc[NU], esh[NU], rsc[NU], rsh[NU];
        const float res_unit = p.ep_rvec ? 0.f : 1.f;
#pragma unroll
        for (int nu = 0; nu < NU; ++nu) {
            esc[nu] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, coff[nu], (unsigned)p.Cout * 8u, 0));
            esh[nu] = __builtin_fmaf(bv[nu], esc[nu], __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, coff[nu], (unsigned)p.Cout * 12u, 0)));
            // (branch-free: without a shortcut BatchNorm the descriptor is empty, the load returns 0 and the scalar addend makes it 1)
            rsc[nu] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrv, coff[nu], (unsigned)p.Cout * 8u, 0)) + res_unit;
            rsh[nu] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrv, coff[nu], (unsigned)p.Cout * 12u, 0));   // (no BatchNorm: zero bytes -> 0)
        }
        float resv[2][NU][4];
        auto load_res = [&](int mt, float (&dst)[NU][4]) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nu = 0; nu < NU; ++nu) {
                    const int row = wr * (16 * MTW) + mt * 16 + 4 * g4 + r;
                    const unsigned off = (row < nrows && coff[nu] != OOB) ? (unsigned)((m0 + row) * p.ld_res * 4) + coff[nu] : OOB;
                    dst[nu][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, off, 0, 0));
                }
        };
        load_res(0, resv[0]);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            if (mt + 1 < MTW) load_res(mt + 1, resv[(mt + 1) & 1]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int nu = 0; nu < NU; ++nu) {
                    const int row = wr * (16 * MTW) + mt * 16 + 4 * g4 + r;
                    const unsigned off = (row < nrows && coff[nu] != OOB) ? (unsigned)((m0 + row) * p.ld_y * 4) + coff[nu] : OOB;
                    float val = __builtin_fmaf(acc[mt][nu][r], esc[nu], esh[nu]);
                    val += __builtin_fmaf(resv[mt & 1][nu][r], rsc[nu], rsh[nu]);
                    val = fmaxf(val, 0.f);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), ry, off, 0, STR ? FGCN_STORE_AUX : 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        if constexpr (O16) {
            // two rows at a time: the even lane of a pair stores columns (c, c + 1) of row rp as one dword, the odd lane those of row rp + 1
            // (fgcn_tconv.hip's bfloat16 epilogue); each unit's sums run over (mt, r) in the float32 form's order
            const bool odd = lane & 1;
#pragma unroll
            for (int rp = 0; rp < 4; rp += 2) {
#pragma unroll
                for (int nu = 0; nu < NU; ++nu) {
                    const int row = wr * (16 * MTW) + mt * 16 + 4 * g4 + rp;
                    const bool ok0 = row < nrows && coff[nu] != OOB, ok1 = row + 1 < nrows && coff[nu] != OOB;
                    const float v0 = acc[mt][nu][rp] + bv[nu], v1 = acc[mt][nu][rp + 1] + bv[nu];
                    const float other = lane_xor1(odd ? v0 : v1);
                    const unsigned pk = odd ? pack_bf16x2(other, v1) : pack_bf16x2(v0, other);
                    const unsigned off = (odd ? ok1 : ok0) ? (unsigned)((m0 + row + (odd ? 1 : 0)) * p.ld_y * 2) + ((coff[nu] - (odd ? 4u : 0u)) >> 1) : OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(pk, ry, off, 0, STR ? FGCN_STORE_AUX : 0);
                    const float k0 = ok0 ? v0 : 0.f, k1 = ok1 ? v1 : 0.f;
                    ssum[nu] += k0;
                    ssq[nu] = __builtin_fmaf(k0, k0, ssq[nu]);
                    ssum[nu] += k1;
                    ssq[nu] = __builtin_fmaf(k1, k1, ssq[nu]);
                }
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int nu = 0; nu < NU; ++nu) {
                const int row = wr * (16 * MTW) + mt * 16 + 4 * g4 + r;
                const unsigned off = (row < nrows && coff[nu] != OOB) ? (unsigned)((m0 + row) * p.ld_y * 4) + coff[nu] : OOB;
                const float val = acc[mt][nu][r] + bv[nu];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), ry, off, 0, STR ? FGCN_STORE_AUX : 0);
                const float kept = off != OOB ? val : 0.f;
                ssum[nu] += kept;
                ssq[nu] = __builtin_fmaf(kept, kept, ssq[nu]);
            }
        }
    }
    if (p.stats) {                                                   // (kernel-uniform)
        __syncthreads();                                             // every wave has left the last MFMA step: the image is free
        float* red = smem_st;                                        // [which][wr][BN]
#pragma unroll
        for (int nu = 0; nu < NU; ++nu) {
            float sa = ssum[nu] + __shfl_xor(ssum[nu], 16);
            float sb = ssq[nu] + __shfl_xor(ssq[nu], 16);
            sa += __shfl_xor(sa, 32);
            sb += __shfl_xor(sb, 32);
            if (lane < 16) {
                red[(0 * 2 + wr) * BN + wc * NT * 32 + nu * 16 + lane] = sa;
                red[(1 * 2 + wr) * BN + wc * NT * 32 + nu * 16 + lane] = sb;
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, c = tid - which * BN;
            if (n0 + c < p.Cout)
                p.stats[((long long)bm * 2 + which) * p.Cout + n0 + c] = red[(which * 2 + 0) * BN + c] + red[(which * 2 + 1) * BN + c];
        }
    }
}

}  // namespace fgcn

using namespace fgcn;

static int sp_tile_frames(int V) { return 128 / V; }

// 1 when fgcn_spatial_fwd_tile runs these sizes in the current math mode (split-bf16 products, whole 64-channel input groups,
// 16..32 joints: at most 8 frames per 128-row tile)
extern "C" int fgcn_spatial_fwd_tile_available(int V, int Cin, int Cout) {
    return (((fgcn::math_mode() == FGCN_MATH_BF16X3 && !fgcn::f16x2_products()) || fgcn::math_mode() == FGCN_MATH_BF16) && V >= 16 && V <= FGCN_MAX_V &&
            Cin % 64 == 0 && Cout % 4 == 0) ? 1 : 0;
}

extern "C" int fgcn_spatial_fwd_tile_tiles(int B, int T, int V) {
    return V >= 16 && V <= FGCN_MAX_V ? (int)(B * cdiv(T, sp_tile_frames(V))) : 0;
}

static int spatial_fwd_tile_impl(const float* x, const float* a_hat, const void* w3, const float* bias_sum, float* y,
                                 float* stat_partials, int B, int T, int V, int Cin, int Cout, int ld_x, int ld_y,
                                 int a_hat_batched, void* stream, const float* ep_vec, const float* ep_res, int ld_res, const float* ep_rvec, int io = 0);

extern "C" int fgcn_spatial_fwd_tile(const float* x, const float* a_hat, const void* w3, const float* bias_sum, float* y,
                                     float* stat_partials, int B, int T, int V, int Cin, int Cout, int ld_x, int ld_y,
                                     int a_hat_batched, void* stream) {
    return spatial_fwd_tile_impl(x, a_hat, w3, bias_sum, y, stat_partials, B, T, V, Cin, Cout, ld_x, ld_y, a_hat_batched, stream, nullptr, nullptr, 0,
                                 nullptr);
}

// typed form (math mode bf16, half-precision activation storage): half_mask bit 0 = x is a bfloat16 tensor, bit 1 = y is (masks 0, 2, 3);
// ld_x / ld_y in elements; stat_partials: the moments of the float32 accumulators
extern "C" int fgcn_spatial_fwd_tile_t(const void* x, const float* a_hat, const void* w3, const float* bias_sum, void* y,
                                       float* stat_partials, int B, int T, int V, int Cin, int Cout, int ld_x, int ld_y,
                                       int a_hat_batched, int half_mask, void* stream) {
    FGCN_REQUIRE(half_mask == 0 || half_mask == 2 || half_mask == 3, FGCN_E_BADARG, "spatial_fwd_tile_t: half_mask=%d (0, 2 or 3)", half_mask);
    return spatial_fwd_tile_impl(static_cast<const float*>(x), a_hat, w3, bias_sum, static_cast<float*>(y), stat_partials, B, T, V, Cin, Cout, ld_x,
                                 ld_y, a_hat_batched, stream, nullptr, nullptr, 0, nullptr, half_mask);
}

// Inference form of north-star kernel 1: aggregation + 1x1 feature contraction + (eval-mode) BatchNorm + shortcut + ReLU in ONE kernel --
// g = relu(BN(sum_k conv_d[k](x . A^_k)) + shortcut), agcn.py:103-115 with the BatchNorm's running statistics folded into a per-channel
// scale / shift (bn_vec = fgcn_bn_eval_coeffs).  res: the shortcut operand (x for an identity block, the down conv's output with its own
// res_vec) or NULL; rows of ld_res floats.  No pre-BatchNorm tensor is written and nothing is kept for a backward.
extern "C" int fgcn_spatial_fwd_tile_bn_relu(const float* x, const float* a_hat, const void* w3, const float* bias_sum, float* g,
                                             const float* bn_vec, const float* res, int ld_res, const float* res_vec,
                                             int B, int T, int V, int Cin, int Cout, int ld_x, int ld_g, int a_hat_batched, void* stream) {
    FGCN_REQUIRE(bn_vec && aligned16(bn_vec) && (!res || (ld_res >= Cout && ld_res % 4 == 0)) && (!res_vec || res), FGCN_E_BADARG,
                 "spatial_fwd_tile_bn_relu: the BatchNorm vector is required; a shortcut needs ld_res >= Cout, its BatchNorm needs the shortcut");
    return spatial_fwd_tile_impl(x, a_hat, w3, bias_sum, g, nullptr, B, T, V, Cin, Cout, ld_x, ld_g, a_hat_batched, stream, bn_vec, res, ld_res, res_vec);
}

static int spatial_fwd_tile_impl(const float* x, const float* a_hat, const void* w3, const float* bias_sum, float* y,
                                 float* stat_partials, int B, int T, int V, int Cin, int Cout, int ld_x, int ld_y,
                                 int a_hat_batched, void* stream, const float* ep_vec, const float* ep_res, int ld_res, const float* ep_rvec, int io) {
    FGCN_REQUIRE(x && a_hat && w3 && y, FGCN_E_BADARG, "spatial_fwd_tile: null pointer");
    FGCN_REQUIRE(io == 0 || ((io == 2 || io == 3) && fgcn::math_mode() == FGCN_MATH_BF16 && !ep_vec), FGCN_E_BADARG,
                 "spatial_fwd_tile_t: bfloat16 tensors (half_mask 2 or 3) need math mode bf16 and the training form");
    FGCN_REQUIRE(B > 0 && T > 0 && Cin > 0 && Cout > 0, FGCN_E_BADARG, "spatial_fwd_tile: bad sizes B=%d T=%d Cin=%d Cout=%d", B, T, Cin, Cout);
    FGCN_REQUIRE(fgcn_spatial_fwd_tile_available(V, Cin, Cout), FGCN_E_BADARG,
                 "spatial_fwd_tile: needs math mode bf16x3 (bf16x3 products) or bf16, 16 <= V <= %d, Cin %% 64 == 0, Cout %% 4 == 0 (V=%d Cin=%d Cout=%d)",
                 FGCN_MAX_V, V, Cin, Cout);
    FGCN_REQUIRE(ld_x % 4 == 0 && ld_y % 4 == 0 && ld_x >= Cin && ld_y >= Cout, FGCN_E_ALIGN, "spatial_fwd_tile: row strides");
    FGCN_REQUIRE(aligned16(x) && aligned16(w3) && aligned16(y), FGCN_E_ALIGN, "spatial_fwd_tile: 16-byte alignment");
    const long long x_bytes = (long long)B * T * V * ld_x * ((io & 1) ? 2 : 4), y_bytes = (long long)B * T * V * ld_y * ((io & 2) ? 2 : 4);
    const long long plane = (long long)3 * Cin * Cout * 2;
    FGCN_REQUIRE(x_bytes < 0x7FFF0000ll && y_bytes < 0x7FFF0000ll && plane * 3 < 0x7FFF0000ll, FGCN_E_BADARG,
                 "spatial_fwd_tile: tensors must be smaller than 2 GiB (32-bit buffer offsets)");
    SpTileP p;
    p.x = x; p.a_hat = a_hat; p.w3 = w3; p.bias = bias_sum; p.y = y; p.stats = stat_partials;
    p.B = B; p.T = T; p.V = V; p.Cin = Cin; p.Cout = Cout; p.ld_x = ld_x; p.ld_y = ld_y; p.a_batched = a_hat_batched;
    p.F = sp_tile_frames(V);
    p.tiles_t = (int)cdiv(T, p.F);
    p.tiles_m = B * p.tiles_t;
    const bool narrow = Cout <= 64;
    p.tiles_n = (int)cdiv(Cout, narrow ? 64 : 128);
    const long long total = (long long)p.tiles_m * p.tiles_n;
    FGCN_REQUIRE(total < (1ll << 30), FGCN_E_BADARG, "spatial_fwd_tile: too many tiles");
    p.per_xcd = (int)cdiv(total, 8);
    p.x_bytes = (unsigned)x_bytes; p.y_bytes = (unsigned)y_bytes; p.w_plane_bytes = (unsigned)plane;
    const bool fep = ep_vec != nullptr;
    const long long res_bytes = ep_res ? (long long)B * T * V * ld_res * 4 : 0;
    FGCN_REQUIRE(res_bytes < 0x7FFF0000ll, FGCN_E_BADARG, "spatial_fwd_tile: the shortcut tensor must be smaller than 2 GiB");
    p.ep_vec = ep_vec; p.ep_res = ep_res; p.ep_rvec = ep_rvec; p.ld_res = ld_res; p.res_bytes = (unsigned)res_bytes;
    const size_t lds = (size_t)3 * ST_PLANE + 9 * 32 * ST_AHB;
    const dim3 grid((unsigned)(p.per_xcd * 8));
    hipStream_t s = (hipStream_t)stream;
    const bool four = 2 * p.F > 12;                                  // aggregation units per wave and chunk: ceil(2 F / 4)
    const bool str = !(io & 2) ? fgcn::stream_out(y_bytes) : ((fgcn::tuning(25) & 2) && fgcn::stream_out(y_bytes));   // (a bfloat16 y: 32-byte pieces, stored plainly; key 25 bit 1: streamed)
#define FGCN_ST_GO6(NT_, MU_, STR_, NP_, FEP_, IO_)                                                                 \
    do {                                                                                                            \
        static bool opted = false;   /* once per instantiation; not a stream operation (stays out of graph captures) */ \
        if (!opted) {                                                                                               \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_tile_x3_kernel<NT_, MU_, STR_, NP_, FEP_, IO_>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
            opted = true;                                                                                           \
        }                                                                                                           \
        hipLaunchKernelGGL((spatial_tile_x3_kernel<NT_, MU_, STR_, NP_, FEP_, IO_>), grid, dim3(256), lds, s, p);   \
    } while (0)
#define FGCN_ST_GO5(NT_, MU_, STR_, NP_, FEP_)                                                                      \
    do {                                                                                                            \
        if constexpr (NP_ == 1 && !FEP_) {                                                                          \
            if (io == 3) { FGCN_ST_GO6(NT_, MU_, STR_, 1, false, 3); break; }                                       \
            if (io == 2) { FGCN_ST_GO6(NT_, MU_, STR_, 1, false, 2); break; }                                       \
        }                                                                                                           \
        FGCN_ST_GO6(NT_, MU_, STR_, NP_, FEP_, 0);                                                                  \
    } while (0)
#define FGCN_ST_GO4(NT_, MU_, STR_, NP_)                \
    do {                                                \
        if (fep) FGCN_ST_GO5(NT_, MU_, STR_, NP_, true); \
        else FGCN_ST_GO5(NT_, MU_, STR_, NP_, false);   \
    } while (0)
    const bool one_part = fgcn::math_mode() == FGCN_MATH_BF16;     // operands rounded to bfloat16 once
#define FGCN_ST_GO3(NT_, MU_, STR_)                     \
    do {                                                \
        if (one_part) FGCN_ST_GO4(NT_, MU_, STR_, 1);   \
        else FGCN_ST_GO4(NT_, MU_, STR_, 3);            \
    } while (0)
#define FGCN_ST_GO(NT_, MU_)                                                                                        \
    do {                                                                                                            \
        if (str) FGCN_ST_GO3(NT_, MU_, true);                                                                       \
        else FGCN_ST_GO3(NT_, MU_, false);                                                                          \
    } while (0)
    if (narrow) {
        if (four) FGCN_ST_GO(1, 4);
        else FGCN_ST_GO(1, 3);
    } else {
        if (four) FGCN_ST_GO(2, 4);
        else FGCN_ST_GO(2, 3);
    }
#undef FGCN_ST_GO
#undef FGCN_ST_GO3
#undef FGCN_ST_GO4
#undef FGCN_ST_GO5
#undef FGCN_ST_GO6
    return launch_status("spatial_fwd_tile");
}
