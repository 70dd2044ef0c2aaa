// Backward of the attention embeddings with the embedding gradient on chip, tile form (split-bf16 math modes).
//
// reference: the autograd backward of SpatialGraphConv.forward through
//     A1 = conv_a[k](x) (theta_k),  A2 = conv_b[k](x) (phi_k),  S_k = softmax_v(theta_k^T phi_k / (ic T))     torch_src/models/mmargcn/agcn.py:104-106
// (SURVEY.md Appendix A.2).  With dS_k (B, 3, V, V) the gradient in front of the softmax (fgcn_adj_softmax_bwd, scale folded in) and the
// embedding tensor emb (B, T, V, 6 ic) = [th0 ph0 th1 ph1 th2 ph2]:
//
//     demb[(n,t,v), th_k + e] = sum_w dS_k[n][v][w] emb[(n,t,w), ph_k + e]           (d theta_k = dS_k . phi_k)
//     demb[(n,t,w), ph_k + e] = sum_v dS_k[n][v][w] emb[(n,t,v), th_k + e]           (d phi_k  = dS_k^T . theta_k)
//     dx[(n,t,v), c]   (+)= sum_j demb[(n,t,v), j] Wemb[j][c]                         (fgcn_emb_dx_tile)
//     dWemb[j][c]        = sum_{n,t,v} demb[(n,t,v), j] x[(n,t,v), c],  dbemb[j] = sum_{n,t,v} demb[(n,t,v), j]     (fgcn_emb_wgrad_tile)
//
// Until round 5 demb (1.5 activations wide) was written by joint_mix_vec and read back by a 1x1 data-gradient GEMM and a 1x1 weight-gradient
// GEMM: three launches and four HBM passes of the wide tensor per block.  Here demb never exists in HBM: both consumers form it per frame
// on the matrix pipe from emb and dS -- a joint mixing in which every 16-channel tile has ONE matrix (dS_k^T for a theta tile, dS_k for a
// phi tile) and reads the PARTNER group's channels -- the two tile kernels of the spatial stage (fgcn_spatial_tile.hip,
// fgcn_spatial_wgrad_tile.hip) with that change:
//
//   * emb_dx_tile_kernel: a workgroup owns F = 128 / V whole frames of one sample times 64 NT columns of dx.  Per chunk of 64 demb channels
//     the mixing units (frame, 16-channel tile) are dealt to the four waves: demb_f^T (16 c x 32 w) = emb_f^T (A operand: eight strided
//     dwords per lane, requested a chunk ahead, split once) . M (B operand: split planes [w][v] in LDS), so that an accumulator lane holds
//     four consecutive channels of one joint = 8-byte pieces of the row-major image [row f V + w][32 channels] (three bf16 planes); the
//     contraction with the pre-split weights (fgcn_pack_split3 of the (6 ic) x Cin matrix) then runs from that image exactly like the
//     forward tile kernel's.  The accumulators START from dx's old values (accumulating form), requested before the first chunk.
//   * emb_wgrad_tile_kernel: one 8-wave workgroup per CU owns a (16 CT demb channels) x (16 NT input channels) tile of dWemb and walks a
//     contiguous range of (sample, frame tile) pairs.  The x tile goes through registers into LDS as bf16 planes; per frame the wave's
//     demb tile (32 joints x 16 channels) is formed in accumulator registers, which after an in-register split ARE the A fragment of the
//     contraction over rows against transposing LDS reads of the x planes (fgcn_spatial_wgrad_tile.hip's scheme with one accumulator set
//     per channel tile instead of three).  The bias gradient is the sum of the same accumulators.
// NP = 3: exact three-way bf16 splits (FGCN_MATH_BF16X3, either product form); NP = 1: operands rounded to bfloat16 once (FGCN_MATH_BF16).
// Every sum has a fixed order (bitwise reproducible).
#include <algorithm>
#include <type_traits>
#include <utility>

#include "fgcn_common.hpp"

// Timing probes (wrong results; tools/build_probe.py only).  dx kernel: bit 0 = no contraction MFMAs, 1 = no mixing (the image stays unwritten),
// 16 = the emb values' bytes in a quarter of the requests, 17 = no in-register split of the emb values (16 + 17: what pre-split transposed
// planes written by the producer would leave of this kernel's input side -- VERDICT r05 item 1),
// 2 = emb values requested for the first chunk only, 3 = no matrix planes, 4 = accumulators start from zero, 5 = no stores, 6 = no weight
// requests past the prologue, 7 = no image fragment reads past the first.  Weight-gradient kernel: bit 8 = no contraction MFMAs, 9 = no mixing
// MFMAs, 10 = emb values requested for the first slot only, 11 = x rows requested / deposited for the first tile only, 12 = one transposing
// read per slot
#ifndef FGCN_PROBE_EMB
#define FGCN_PROBE_EMB 0
#endif

namespace fgcn {

constexpr unsigned ET_OOB = 0x80000000u;
constexpr int ET_AHB = 80;              // bytes per [w] row of a split matrix plane (32 joints v x bf16 + 16 pad: conflict-free b128 reads)

// split matrix planes of `nm` consecutive (subset, side) groups starting at group g_lo -> LDS [slot][part][w][v]: the plane of group g holds
// M[v_in][w_out] at [w_out][v_in] -- g even (theta_k): M = dS_k^T, g odd (phi_k): M = dS_k
template <int NP, int NTHREADS>
__device__ __forceinline__ void et_stage_planes(unsigned char* Ah, const float* src, int V, int g_lo, int nm, int tid) {
    for (int i = tid; i < nm * 1024; i += NTHREADS) {
        const int m = i >> 10, w = (i >> 5) & 31, v = i & 31;
        const int g = g_lo + m, k = g >> 1;
        float a = 0.f;
        if (g < 6 && v < V && w < V) a = (g & 1) ? src[(k * V + v) * V + w] : src[(k * V + w) * V + v];
        unsigned ph, pm, pl;
        split_bf16_pair(a, 0.f, ph, pm, pl);
        unsigned short* d = reinterpret_cast<unsigned short*>(Ah + ((m * NP) * 32 + w) * ET_AHB) + v;
        d[0] = (unsigned short)ph;
        if constexpr (NP == 3) {
            d[32 * ET_AHB / 2] = (unsigned short)pm;
            d[2 * 32 * ET_AHB / 2] = (unsigned short)pl;
        }
    }
}
// The same for all six groups ONCE per sample, into global memory in the LDS image's own layout [group][part][w][ET_AHB bytes]
// (fgcn_emb_dx_tile's workspace): a workgroup of the dx kernel -- one per 128-row tile -- then copies 16-byte pieces instead of splitting
// 24 values per thread (300 of its 1500 vector instructions per tile and 72 two-byte LDS writes; SQ counters, profiles/r05_pmc_emb_dx.txt).
template <int NP> constexpr int et_planes_bytes() { return 6 * NP * 32 * ET_AHB; }
template <int NP>
__global__ __launch_bounds__(256) void emb_planes_kernel(const float* __restrict__ d_s, unsigned char* __restrict__ planes, int V) {
    const float* src = d_s + (long long)blockIdx.x * 3 * V * V;
    unsigned char* dst = planes + (long long)blockIdx.x * et_planes_bytes<NP>();
    for (int i = threadIdx.x; i < 6 * 1024; i += 256) {
        const int g = i >> 10, k = g >> 1, w = (i >> 5) & 31, v = i & 31;
        float a = 0.f;
        if (v < V && w < V) a = (g & 1) ? src[(k * V + v) * V + w] : src[(k * V + w) * V + v];
        unsigned ph, pm, pl;
        split_bf16_pair(a, 0.f, ph, pm, pl);
        unsigned short* d = reinterpret_cast<unsigned short*>(dst + ((g * NP) * 32 + w) * ET_AHB) + v;
        d[0] = (unsigned short)ph;
        if constexpr (NP == 3) {
            d[32 * ET_AHB / 2] = (unsigned short)pm;
            d[2 * 32 * ET_AHB / 2] = (unsigned short)pl;
        }
        if (v < 8) d[32] = 0;                                        // the 16 pad bytes of the row (copied, never read as operands)
        if constexpr (NP == 3) {
            if (v < 8) d[32 * ET_AHB / 2 + 32] = 0, d[2 * 32 * ET_AHB / 2 + 32] = 0;
        }
    }
}
// one sample's planes, global -> LDS, every request in flight at once (the last pass is whole waves: the piece count is a multiple of 64)
template <int NP, int NTHREADS>
__device__ __forceinline__ void et_copy_planes(unsigned char* Ah, const unsigned char* src, int tid) {
    constexpr int PIECES = et_planes_bytes<NP>() / 16, NE = (PIECES + NTHREADS - 1) / NTHREADS;
    static_assert(PIECES % 64 == 0, "whole waves");
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)et_planes_bytes<NP>(), 0x00020000);
    u32x4v v[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(tid + NTHREADS * e) * 16u, 0, 0);
#pragma unroll
    for (int e = 0; e < NE; ++e)
        if (tid + NTHREADS * e < PIECES) *reinterpret_cast<u32x4v*>(Ah + (tid + NTHREADS * e) * 16) = v[e];     // (wave-uniform)
}

// =====================================================================================================================================
// dx (+)= demb . Wemb
// =====================================================================================================================================
struct EmbDxP {
    const float* emb;
    const unsigned char* planes;        // the split matrix planes of every sample (emb_planes_kernel)
    const void* w3;                     // fgcn_pack_split3 form of the (6 ic) x Cout matrix: [part][j / 8][c][8] bf16
    float* dx;
    const float* dx_old;                // H16 bit 2: the accumulating form reads its old values from THIS float32 tensor (ld_dx floats per row) and
    unsigned old_bytes;                 // writes the sums to the bfloat16 dx -- the block's last writer of dx converts on the way (no torch pass)
    int B, T, V, ic, Ce, Cout, ld_e, ld_dx, s_batched;
    unsigned ic_inv;                    // ceil(2^32 / ic): d / ic = the high word of d * ic_inv for d < 6 ic
    int F, tiles_t, tiles_m, tiles_n, per_xcd;
    unsigned e_bytes, dx_bytes, w_plane_bytes;
};

constexpr int ED_XS = 64;               // bytes per image row and part (32 channels x bf16), 32-byte blocks XOR-swizzled by row bit 2
constexpr int ED_PLANE = 128 * ED_XS;   // one part of the image: one 32-channel pair x 128 rows
constexpr int ED_NMAT = 6;              // matrix slots: all (subset, side) groups stay resident
template <int NP> constexpr int ed_lds() { return NP * ED_PLANE + ED_NMAT * NP * 32 * ET_AHB; }

// One chunk = one 32-channel pair of demb (the image of two pairs beside six resident matrices would not leave room for two workgroups
// per CU; re-staging the matrices per chunk cost 16 prefetch registers and spilled).  MAXU: mixing units (frame, 16-channel half) of a
// wave per chunk = ceil(2 F / 4).
// PD: chunks the emb values are requested ahead (the chunk loop is unrolled PD times; 6 ic / 32 is a multiple of 3); RSN: weight ring slots
// (Measured and not kept, profiles/r05_kbench_emb_bwd_variants.txt: the emb values of all three chunks requested at once -- 215-244 registers,
// 3-10 % slower; eight waves per workgroup, 4 x 2 over the tile, four waves per SIMD at 104-122 registers -- bit-identical, 6-14 % slower.)
// E16 (NP = 1): emb is a BFLOAT16 tensor (fgcn_emb_fwd_tile_h; ld_e in elements): the strided requests fetch two bytes per value and the value
// is widened by a shift -- the fragment then holds the same bfloat16 the f32 form would have rounded to
// H16 bit 0 = that, bit 1 = dx is a BFLOAT16 tensor too (half-precision activation storage, the `_t` entry point; ld_dx in elements): the old
// values are 2-byte loads, the result is rounded once and adjacent lanes pair their columns into dword stores (fgcn_tconv.hip's bfloat16 epilogue)
template <int NP, int NT, int MAXU, bool ACC, int PD, int RSN, int H16 = 0>
__global__ __launch_bounds__(256, 2) void emb_dx_tile_kernel(EmbDxP p) {
    static_assert(!H16 || NP == 1, "bfloat16 tensors: the one-part kernel");
    constexpr bool E16 = (H16 & 1) != 0, DX16 = (H16 & 2) != 0, OLD32 = (H16 & 4) != 0;
    static_assert(!OLD32 || (DX16 && ACC), "float32 old values: the accumulating form with a bfloat16 dx");
    constexpr unsigned ES = E16 ? 2u : 4u;                           // bytes per stored emb value
    constexpr unsigned DS = DX16 ? 2u : 4u;                          // ... per dx value
    constexpr int MTW = 4, NU = 2 * NT, BN = 64 * NT, NW = 4;
    static_assert(PD == 1 || PD == 3, "the chunk count is a multiple of three");
    auto swz = [](int r) -> unsigned { return (unsigned)(r & 4) << 3; };
    extern __shared__ __attribute__((aligned(16))) unsigned char ed_lds_raw[];
    unsigned char* Xh = ed_lds_raw;                                  // [NP parts][128 rows][64 B]
    unsigned char* ahs = Xh + NP * ED_PLANE;                         // [6 groups][NP parts][32 w][ET_AHB]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int vid = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);    // XCD-aware order, column tile fastest
    if (vid >= p.tiles_m * p.tiles_n) return;
    const int bm = vid / p.tiles_n, bn = vid - bm * p.tiles_n;
    const int n = bm / p.tiles_t, tf = bm - n * p.tiles_t;
    const int V = p.V, F = p.F, ic = p.ic;
    const int t0 = tf * F;
    const int nf = min(F, p.T - t0);
    const int nrows = nf * V;
    const int n0 = bn * BN;

    const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)p.emb, 0, p.e_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, p.w_plane_bytes * NP, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdx = __builtin_amdgcn_make_buffer_rsrc((void*)p.dx, 0, p.dx_bytes, 0x00020000);

    // the six matrices of this sample, split once per workgroup
    if (!(FGCN_PROBE_EMB & 8)) et_copy_planes<NP, 64 * NW>(ahs, p.planes + (p.s_batched ? (long long)n * et_planes_bytes<NP>() : 0), tid);

    // ---- accumulators: dx's old values (accumulating form) or zero; register r of lane (col l15, g4) = row 4 g4 + r of its 16 x 16 tile.
    // Address of element (mt, r, nu) = one per-lane offset + a scalar row part + a constant column part.  The LOADS carry no masks: rows past
    // the tile and columns past Cout read other valid memory (or zeros past the tensor) into accumulators that are never stored -- every
    // output element depends on its own image row and weight column only.
    const unsigned m0 = (unsigned)((n * p.T + t0) * V);              // (32-bit row math: the launcher bounds the tensors by 2 GiB)
    const int col = n0 + wc * NT * 32 + l15;                         // + nu * 16
    const unsigned lane_base = ((m0 + (unsigned)(wr * (16 * MTW) + 4 * g4)) * (unsigned)p.ld_dx + (unsigned)col) * DS;
    const unsigned dx_row_b = (unsigned)p.ld_dx * DS;
    f32x4 acc[MTW][NU];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int nu = 0; nu < NU; ++nu) {
                if constexpr (OLD32)
                    acc[mt][nu][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                   __builtin_amdgcn_make_buffer_rsrc((void*)p.dx_old, 0, p.old_bytes, 0x00020000),
                                                                   lane_base * 2u + nu * 64, (unsigned)(mt * 16 + r) * dx_row_b * 2u, 0));
                else if constexpr (ACC && DX16)
                    acc[mt][nu][r] = __builtin_bit_cast(float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(
                                                                   rdx, lane_base + nu * 32, (unsigned)(mt * 16 + r) * dx_row_b, 0) << 16);
                else if constexpr (ACC && !(FGCN_PROBE_EMB & 16))
                    acc[mt][nu][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdx, lane_base + nu * 64, (unsigned)(mt * 16 + r) * dx_row_b, 0));
                else
                    acc[mt][nu][r] = 0.f;
            }

    // ---- mixing units of a chunk: (frame f, 16-channel half hh of the pair); the 2 F units in frame-major order are dealt to the waves in
    // contiguous blocks
    int uf[MAXU], uh[MAXU];
    bool uok[MAXU];
    const int u_lo = __builtin_amdgcn_readfirstlane((wave * 2 * F) / NW), u_hi = __builtin_amdgcn_readfirstlane(((wave + 1) * 2 * F) / NW);
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        const int u = u_lo + i;
        uf[i] = u >> 1;
        uh[i] = u & 1;
        uok[i] = u < u_hi && uf[i] < nf;                             // (wave-uniform)
    }
    const int nchunks = p.Ce >> 5;                                   // 32-channel pairs of demb
    const unsigned row_b = (unsigned)p.ld_e * ES;
    // emb values of a unit: lane (c = l15, g4) <- emb[(f, v = 8 g4 + j)][partner channel + l15], j = 0 .. 7 (the A fragment of the mixing).
    // Per request: per-lane offset = the unit's row base | the joint's out-of-range bit (both fixed for the kernel), scalar offset = the
    // joint's rows + the chunk's partner channels; past the last chunk the requests go to an empty descriptor (zeros, no traffic).
    const __amdgpu_buffer_rsrc_t re_none = __builtin_amdgcn_make_buffer_rsrc((void*)p.emb, 0, 0, 0x00020000);
    unsigned ubase[MAXU], jinv[8];
#pragma unroll
    for (int i = 0; i < MAXU; ++i)
        ubase[i] = uok[i] ? (((unsigned)((n * p.T + t0 + uf[i]) * V) + 8u * g4) * (unsigned)p.ld_e + (unsigned)l15) * ES : ET_OOB;
#pragma unroll
    for (int j = 0; j < 8; ++j) jinv[j] = 8 * g4 + j < V ? 0u : ET_OOB;
    auto group_of = [&](int d0) -> int { return (int)__umulhi((unsigned)d0, p.ic_inv); };
    float xr[PD][MAXU][8];
    auto fetch_units = [&](int c, float (&xr)[MAXU][8]) {
        const bool live = c < nchunks && (!(FGCN_PROBE_EMB & 4) || c == 0);
        const __amdgpu_buffer_rsrc_t rc = live ? re : re_none;
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            const int d0 = 32 * c + 16 * uh[i];                      // the unit's demb channels; their group -> the partner group's channels
            const int csrc = d0 + ((group_of(d0) & 1) ? -ic : ic);
            if constexpr ((FGCN_PROBE_EMB & 65536) != 0) {           // the same bytes in a quarter of the requests (wrong values)
#pragma unroll
                for (int j = 0; j < 8; j += 4) {
                    const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rc, (ubase[i] | jinv[j]) & ~15u, (unsigned)j * row_b + (unsigned)csrc * 4u, 0));
                    xr[i][j] = q[0], xr[i][j + 1] = q[1], xr[i][j + 2] = q[2], xr[i][j + 3] = q[3];
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (E16) {
                    const unsigned h = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rc, ubase[i] | jinv[j], (unsigned)j * row_b + (unsigned)csrc * 2u, 0);
                    xr[i][j] = __builtin_bit_cast(float, h << 16);
                } else {
                    xr[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, ubase[i] | jinv[j], (unsigned)j * row_b + (unsigned)csrc * 4u, 0));
                }
            }
        }
    };
    // Branch-free over the wave's units: a unit that does not exist mixes the zeros its requests returned and writes nothing, so that the
    // 2 MAXU independent MFMA chains of a chunk are ONE basic block (a wave-uniform `continue` per unit made every unit its own block:
    // six dependent MFMAs, then the splits that wait for them, one unit after the other -- half of the kernel's time).
    auto stage_units = [&](int c, float (&xr)[MAXU][8]) {
        if ((FGCN_PROBE_EMB & 2)) return;
        u32x4v xs[MAXU][NP];
        int g[MAXU];
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            g[i] = group_of(32 * c + 16 * uh[i]);                    // the unit's group = its matrix
            if constexpr ((FGCN_PROBE_EMB & 131072) != 0) {         // probe: emb arrives pre-split (its bits stand in for the planes)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    xs[i][pl] = u32x4v{__builtin_bit_cast(unsigned, xr[i][pl]), __builtin_bit_cast(unsigned, xr[i][pl + 1]),
                                       __builtin_bit_cast(unsigned, xr[i][pl + 2]), __builtin_bit_cast(unsigned, xr[i][pl + 3])};
            } else {
                splitn_x8<NP>(xr[i][0], xr[i][1], xr[i][2], xr[i][3], xr[i][4], xr[i][5], xr[i][6], xr[i][7], xs[i]);
            }
        }
        f32x4 m[MAXU][2];
#pragma unroll
        for (int wt = 0; wt < 2; ++wt)
#pragma unroll
            for (int i = 0; i < MAXU; ++i) {
                u32x4v af[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    if constexpr ((FGCN_PROBE_EMB & 32768) != 0) af[pl] = xs[i][pl] + (unsigned)(wt + 1);
                    else af[pl] = *reinterpret_cast<const u32x4v*>(ahs + ((g[i] * NP + pl) * 32 + 16 * wt + l15) * ET_AHB + 16 * g4);
                }
                // demb_f^T (16 c x 16 w): lane (w = 16 wt + l15, g4) holds channels 4 g4 .. + 3 of the half
                if constexpr ((FGCN_PROBE_EMB & 8192) != 0) m[i][wt] = __builtin_bit_cast(f32x4, xs[i][0] ^ af[0] ^ xs[i][NP - 1] ^ af[NP - 1]);
                else m[i][wt] = mfma_np_k32<NP>(xs[i], af, f32x4{0.f, 0.f, 0.f, 0.f});
            }
#pragma unroll
        for (int i = 0; i < MAXU; ++i)
#pragma unroll
            for (int wt = 0; wt < 2; ++wt) {
                const int w = 16 * wt + l15;
                const int R = uf[i] * V + w;
                u32x2 parts[NP];
                splitn_x4<NP>(m[i][wt], parts);
                if (w < V && uok[i] && !((FGCN_PROBE_EMB & 16384) && parts[0][0] != 0x12345u)) {
                    unsigned char* dst = Xh + R * ED_XS + ((unsigned)(32 * uh[i] + 8 * g4) ^ swz(R));
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<u32x2*>(dst + pl * ED_PLANE) = parts[pl];
                }
            }
    };

    unsigned wvoff[NU];                                              // per-lane byte offset into one part: (g4 * Cout + col) * 8 bf16
#pragma unroll
    for (int nu = 0; nu < NU; ++nu) wvoff[nu] = col + nu * 16 < p.Cout ? (unsigned)(((long long)g4 * p.Cout + col + nu * 16) * 16) : ET_OOB;
    // weight fragment of (column unit nu, pair pq): contraction rows 32 pq + 8 g4 + j; past the last pair: pair 0 (a valid, unused load)
    auto load_w = [&](u32x4v (&dst)[NP], int nu, int pq) {
        if ((FGCN_PROBE_EMB & 64) && pq > 0) return;
        if (pq >= nchunks) pq = 0;
        const unsigned so = (unsigned)(((long long)(4 * pq) * p.Cout) * 16);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) dst[pl] = __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff[nu], so + pl * p.w_plane_bytes, 0);
    };
    const int xrow = wr * (16 * MTW) + l15;
    bool first_a = true;
    auto load_a = [&](u32x4v (&dst)[NP], int mt) {
        if ((FGCN_PROBE_EMB & 128) && !first_a) return;
        if (mt == MTW - 1) first_a = false;
        const int r = xrow + mt * 16;
        const unsigned char* src = Xh + r * ED_XS + ((unsigned)(16 * g4) ^ swz(r));
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) dst[pl] = *reinterpret_cast<const u32x4v*>(src + pl * ED_PLANE);
    };
    constexpr int RS = RSN;                                          // weight ring slots (fragments requested RS - 1 units ahead)
    static_assert(RS >= 2 && RS <= NU, "ring slots");
    u32x4v a[MTW][NP], wq[RS][NP];
    auto feature_phase = [&](int c) {                                // pair c from the image
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) load_a(a[mt], mt);
#pragma unroll
        for (int nu = 0; nu < NU; ++nu) {
            const int t = nu + RS - 1;
            if (t < NU) load_w(wq[t % RS], t, c);
            else load_w(wq[t % RS], t - NU, c + 1);
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                if constexpr ((FGCN_PROBE_EMB & 1) != 0) acc[mt][nu][0] += __builtin_bit_cast(float, a[mt][0][0] ^ wq[nu % RS][0][0]);
                else acc[mt][nu] = mfma_np_k32<NP>(a[mt], wq[nu % RS], acc[mt][nu]);
            }
        }
    };

#pragma unroll
    for (int d = 0; d < PD; ++d) fetch_units(d, xr[d]);
#pragma unroll
    for (int nu = 0; nu < RS - 1; ++nu) load_w(wq[nu], nu, 0);
    for (int c0 = 0; c0 < nchunks; c0 += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int c = c0 + d;
            __syncthreads();                                         // the previous chunk's image reads are done (first pass: the matrices are written)
            stage_units(c, xr[d]);
            __syncthreads();
            fetch_units(c + PD, xr[d]);                              // lands during the MFMAs below (past the last chunk: nothing is read)
            feature_phase(c);
        }
    }

    auto val_guard = [](float v) { return v != 1.2345e-30f; };      // (probe bit 5: the stores depend on the values, nothing is written)
    // ---- epilogue: branch-free buffer stores; rows beyond the tile's frames and columns beyond Cout carry the out-of-range offset
    if constexpr (DX16) {
        // two rows at a time: the even lane of a pair stores columns (c, c + 1) of row rp as one dword, the odd lane those of row rp + 1
        const bool odd = lane & 1;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int rp = 0; rp < 4; rp += 2) {
                const int row = wr * (16 * MTW) + mt * 16 + 4 * g4 + rp + (odd ? 1 : 0);
#pragma unroll
                for (int nu = 0; nu < NU; ++nu) {
                    const float v0 = acc[mt][nu][rp], v1 = acc[mt][nu][rp + 1];
                    const float other = lane_xor1(odd ? v0 : v1);
                    const unsigned pk = odd ? pack_bf16x2(other, v1) : pack_bf16x2(v0, other);
                    const unsigned off = (row < nrows && col + nu * 16 < p.Cout) ? lane_base + nu * 32 + (odd ? dx_row_b - 2u : 0u) : ET_OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(pk, rdx, off, (unsigned)(mt * 16 + rp) * dx_row_b, 0);
                }
            }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool rowok = wr * (16 * MTW) + mt * 16 + 4 * g4 + r < nrows;
#pragma unroll
            for (int nu = 0; nu < NU; ++nu) {
                const unsigned off = (rowok && col + nu * 16 < p.Cout && !((FGCN_PROBE_EMB & 32) && val_guard(acc[mt][nu][r]))) ? lane_base + nu * 64 : ET_OOB;
                const float val = acc[mt][nu][r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rdx, off, (unsigned)(mt * 16 + r) * dx_row_b, 0);
            }
        }
}

// =====================================================================================================================================
// dWemb = demb^T . x,  dbemb = column sums of demb
// =====================================================================================================================================
struct EmbWgP {
    const float* emb;
    const float* x;
    const float* d_s;
    float* partial;                     // float[nseg][Ce][Cx]
    float* bias_partial;                // float[nseg][Ce]
    int B, T, V, ic, Ce, Cx, ld_e, ld_x, s_batched;
    int F, tiles_t, gtiles, tps, nseg, n_cg, n_og;
    unsigned e_bytes, x_bytes, p_bytes, b_bytes;
};

constexpr int EW_ROWS = 160;            // rows of an x plane: (F - 1) V + 32 <= 160
// row stride of an x plane: the channels' bytes + 32 -- eight consecutive rows then start 32 bytes apart modulo 256 (a transposing read's
// half wave touches 8 rows x 32 bytes)
template <int NT> constexpr int ew_rs() { return NT * 32 + 32; }
template <int NP, int NT, int NM> constexpr int ew_lds() { return NP * EW_ROWS * ew_rs<NT>() + NM * NP * 32 * ET_AHB; }

__device__ __forceinline__ u32x2 ew_read_tr16(const unsigned char* p) {
    using v4s = __attribute__((ext_vector_type(4))) short;
    const v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
    return __builtin_bit_cast(u32x2, v);
}

template <class Fn, int... S>
__device__ __forceinline__ void ew_for_slots(Fn&& fn, std::integer_sequence<int, S...>) {
    (fn(std::integral_constant<int, S>{}), ...);
}

// CT: 16-channel demb tiles of the workgroup (8: every wave walks all frames; 4: two waves per tile take alternate frames, added at the
// end); NT: 16-channel tiles of x (4 or 8); NSLOT: frame slots of a wave per tile, compile time (straight-line code: exact request counts);
// NM: matrix slots in LDS (the (subset, side) groups the workgroup's channels touch: 2, or 6 for ic = 16)
// PF: frame slots the emb values are requested ahead (a ring of PF register sets, by slot index modulo PF: NSLOT % PF == 0)
// E16 (NP = 1): emb as BFLOAT16, as in the dx kernel
// H16 bit 0 = that, bit 1 = x is a BFLOAT16 tensor too (the `_t` entry point; ld_x in elements): its rows are copied into the image
template <int NP, int CT, int NT, int NSLOT, int NM, int PF, int H16 = 0>
__global__ __launch_bounds__(512, 1) void emb_wgrad_tile_kernel(EmbWgP p) {
    static_assert(!H16 || NP == 1, "bfloat16 tensors: the one-part kernel");
    constexpr bool E16 = (H16 & 1) != 0, X16 = (H16 & 2) != 0;
    constexpr int FP = 8 / CT;
    static_assert(PF >= 1 && NSLOT % PF == 0, "the slot ring must close over a tile");
    constexpr int RS = ew_rs<NT>(), PL = EW_ROWS * RS;
    constexpr int GPR = NT * 4, RPP = 512 / GPR, NPASS = EW_ROWS / RPP;      // 16-byte groups per row, rows per pass, passes
    static_assert(EW_ROWS % RPP == 0 && NSLOT % 2 == 0, "staging passes / slot pairs");
    constexpr int SPREAD = NSLOT > 2 ? NSLOT - 2 : 1;                // the next tile's x rows are requested in pieces over the slots
    extern __shared__ __attribute__((aligned(16))) unsigned char ew_lds_raw[];
    unsigned char* Im = ew_lds_raw;
    unsigned char* Ah = ew_lds_raw + NP * PL;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4, q4 = l15 >> 2, c4 = lane & 3;
    const int ct = wave % CT, fp = wave / CT;
    const int combos = p.n_cg * p.n_og;
    const int seg = blockIdx.x / combos, combo = blockIdx.x - seg * combos;
    const int cgi = combo / p.n_og, ogi = combo - cgi * p.n_og;
    const int cg0 = cgi * 16 * CT;                                   // first demb channel of the workgroup
    const int c0 = cg0 + 16 * ct, o0 = ogi * 16 * NT;
    const bool active = c0 < p.Ce;                                   // (wave-uniform) this wave's channel tile exists
    const int g_lo = cg0 / p.ic, gq = (active ? c0 : cg0) / p.ic;
    const int ms = gq - g_lo;                                        // matrix slot (< NM: checked by the launcher)
    const int csrc = c0 + ((gq & 1) ? -p.ic : p.ic);                 // the partner group's channels
    const int V = p.V, F = p.F;
    const int t_lo = seg * p.tps, t_hi = min(t_lo + p.tps, p.gtiles);

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)p.emb, 0, p.e_bytes, 0x00020000);

    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    const int srow = tid / GPR, sg = tid % GPR;
    f32x4 stg[NPASS];
    // pass i of the x rows of (sample, tile) pair g (nothing past the segment: branch-free)
    auto fetch_pass = [&](int g, int i) {
        const int n_ = g / p.tiles_t, tile_ = g - n_ * p.tiles_t;
        const int t0_ = tile_ * F;
        const int nrows_ = (g < t_hi && (!(FGCN_PROBE_EMB & 2048) || g == t_lo)) ? min(F, p.T - t0_) * V : 0;
        const unsigned row0_ = (unsigned)((n_ * p.T + t0_) * V);
        const int r = srow + RPP * i;
        const unsigned off = r < nrows_ ? ((row0_ + r) * (unsigned)p.ld_x + (unsigned)(o0 + 4 * sg)) * (X16 ? 2u : 4u) : ET_OOB;
        if constexpr (X16) {                                         // four bfloat16 = 8 bytes, parked in the first two components
            const u32x2 h = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, off, 0, 0));
            const unsigned b0 = h[0], b1 = h[1];                     // (element -> scalar before a bit cast: hipcc 7.2 reads element 0 otherwise)
            stg[i] = f32x4{__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1), 0.f, 0.f};
        } else {
            stg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
    };
    // emb values of frame slot s of pair g for this wave: lane (c = l15, g4) <- emb[(f, v = 8 g4 + j)][csrc + l15], j = 0 .. 7
    float xr_ring[PF][8];
    auto xfetch = [&](float (&xr)[8], int g, int s) {
        const int n_ = g / p.tiles_t, tile_ = g - n_ * p.tiles_t;
        const int t0_ = tile_ * F, f = fp + FP * s;
        const int vlim = (active && g < t_hi && f < min(F, p.T - t0_) && (!(FGCN_PROBE_EMB & 1024) || (g == t_lo && s == 0))) ? V : 0;   // (a scalar select: no frame, no joints)
        const unsigned base = (unsigned)((n_ * p.T + t0_ + f) * V + 8 * g4) * (unsigned)p.ld_e + (unsigned)((active ? csrc : 0) + l15);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned off = 8 * g4 + j < vlim ? (base + (unsigned)(j * p.ld_e)) * (E16 ? 2u : 4u) : ET_OOB;
            if constexpr (E16) {
                const unsigned h = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(re, off, 0, 0);
                xr[j] = __builtin_bit_cast(float, h << 16);
            } else {
                xr[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(re, off, 0, 0));
            }
        }
    };
    auto planes = [&](int n) { et_stage_planes<NP, 512>(Ah, p.d_s + (p.s_batched ? (long long)n * 3 * V * V : 0), V, g_lo, NM, tid); };
    auto deposit = [&]() {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int r = srow + RPP * i;
            u32x2 parts[NP];
            if constexpr (X16) {                                     // already bfloat16: a copy
                const float e0 = stg[i][0], e1 = stg[i][1];
                parts[0] = u32x2{__builtin_bit_cast(unsigned, e0), __builtin_bit_cast(unsigned, e1)};
            } else {
                splitn_x4<NP>(stg[i], parts);
            }
            unsigned char* dst = Im + r * RS + sg * 8;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<u32x2*>(dst + pl * PL) = parts[pl];
        }
    };
#pragma unroll
    for (int i = 0; i < NPASS; ++i) fetch_pass(t_lo, i);
#pragma unroll
    for (int d = 0; d < PF; ++d) xfetch(xr_ring[d], t_lo, d);
    if (t_lo < t_hi) planes(t_lo / p.tiles_t);
    deposit();
    __syncthreads();

    for (int g = t_lo; g < t_hi; ++g) {
        const int n = g / p.tiles_t, tile = g - n * p.tiles_t;
        const int nf = min(F, p.T - tile * F);
        // Branch-free per slot: a frame past the tile's last mixes the zeros its requests returned (vlim = 0) against rows of frame 0 --
        // finite values, so the product is an exact zero -- and the slots of a tile are ONE basic block for the waves that own a channel
        // tile: slot s + 1's mixing chains can issue under slot s's contraction chains (with one matrix per channel tile a slot has a third
        // of the independent work of the conv_d kernel's).  Waves without a channel tile (ic = 16: two of eight) only take part in the staging.
        auto slot = [&](auto s_tag, auto act_tag) {
            constexpr int s = decltype(s_tag)::value;
            constexpr bool ACT = decltype(act_tag)::value;
            const int f = fp + FP * s;
            float (&xr)[8] = xr_ring[s % PF];
            u32x4v xs[NP];
            if constexpr (ACT) splitn_x8<NP>(xr[0], xr[1], xr[2], xr[3], xr[4], xr[5], xr[6], xr[7], xs);
            if constexpr (s + PF < NSLOT) xfetch(xr, g, s + PF);     // a later slot of this tile, or of the next tile
            else xfetch(xr, g + 1, s + PF - NSLOT);
#pragma unroll
            for (int i = 0; i < NPASS; ++i)
                if (i * SPREAD / NPASS == s) fetch_pass(g + 1, i);
            if constexpr (ACT) {
                f32x4 m[2];
#pragma unroll
                for (int wt = 0; wt < 2; ++wt) {
                    u32x4v af[NP];
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        af[pl] = *reinterpret_cast<const u32x4v*>(Ah + ((ms * NP + pl) * 32 + 16 * wt + l15) * ET_AHB + 16 * g4);
                    // demb_f (32 joints w x 16 channels): lane (c = l15, g4) holds joints w = 4 g4 + r (wt = 0) and 16 + 4 g4 + r (wt = 1)
                    if constexpr ((FGCN_PROBE_EMB & 512) != 0) m[wt] = __builtin_bit_cast(f32x4, af[0] ^ xs[0]);
                    else m[wt] = mfma_np_k32<NP>(af, xs, f32x4{0.f, 0.f, 0.f, 0.f});
                }
                bsum += ((m[0][0] + m[0][1]) + (m[0][2] + m[0][3])) + ((m[1][0] + m[1][1]) + (m[1][2] + m[1][3]));   // joints >= V are exact zeros
                u32x4v a3[NP];
                splitn_x8<NP>(m[0][0], m[0][1], m[0][2], m[0][3], m[1][0], m[1][1], m[1][2], m[1][3], a3);
                const int r_lo = (f < nf ? f : 0) * V + 4 * g4 + q4, r_hi = r_lo + 16;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    u32x4v df[NP];
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) {
                        if ((FGCN_PROBE_EMB & 4096) && nt > 0) break;
                        const unsigned char* base = Im + pl * PL + nt * 32 + 8 * c4;
                        const u32x2 lo = ew_read_tr16(base + r_lo * RS);
                        const u32x2 hi = ew_read_tr16(base + r_hi * RS);
                        df[pl] = u32x4v{lo[0], lo[1], hi[0], hi[1]};
                    }
                    if constexpr ((FGCN_PROBE_EMB & 256) != 0) acc[nt][0] += __builtin_bit_cast(float, a3[0][0] ^ df[0][0]);
                    else acc[nt] = mfma_np_k32<NP>(a3, df, acc[nt]);
                }
            }
        };
        if (active) ew_for_slots([&](auto s_tag) { slot(s_tag, std::true_type{}); }, std::make_integer_sequence<int, NSLOT>{});
        else ew_for_slots([&](auto s_tag) { slot(s_tag, std::false_type{}); }, std::make_integer_sequence<int, NSLOT>{});
        // the next tile's x rows (and its sample's matrix planes) replace this one's
        __syncthreads();                                             // this tile's fragment reads are done
        const int n1 = (g + 1) / p.tiles_t;
        if (g + 1 < t_hi && n1 != n && p.s_batched) planes(n1);
        if (!(FGCN_PROBE_EMB & 2048)) deposit();
        __syncthreads();
    }

    // ---- the workgroup's slabs: partial[seg][c][o], bias_partial[seg][c] ---------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)p.partial, 0, p.p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias_partial, 0, p.b_bytes, 0x00020000);
    float bs = bsum + __shfl_xor(bsum, 16);                          // the four joint groups of a channel, fixed order
    bs += __shfl_xor(bs, 32);
    if constexpr (FP == 2) {                                         // fixed order: frames of the even slots + frames of the odd slots
        __syncthreads();
        float* red = reinterpret_cast<float*>(ew_lds_raw);           // [ct][nt][r][lane], then [ct][16]
        float* redb = red + CT * NT * 4 * 64;
        if (fp == 1) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[((ct * NT + nt) * 4 + r) * 64 + lane] = acc[nt][r];
            if (lane < 16) redb[ct * 16 + lane] = bs;
        }
        __syncthreads();
        if (fp == 1) return;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[nt][r] += red[((ct * NT + nt) * 4 + r) * 64 + lane];
        bs += redb[ct * 16 + l15];
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned off = active ? (((unsigned)seg * (unsigned)p.Ce + (unsigned)(c0 + 4 * g4 + r)) * (unsigned)p.Cx + (unsigned)(o0 + 16 * nt + l15)) * 4u : ET_OOB;
            const float val = acc[nt][r];                            // (a bit_cast of the vector-element lvalue itself reads element 0: hipcc 7.2)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rp, off, 0, 0);
        }
    {
        const unsigned off = (active && ogi == 0 && lane < 16) ? ((unsigned)seg * (unsigned)p.Ce + (unsigned)(c0 + l15)) * 4u : ET_OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bs), rb, off, 0, 0);
    }
}

// frames per tile of the weight-gradient kernel: as many whole frames as keep a 32-joint fragment of the last one inside EW_ROWS rows, at most 8
static int ew_frames(int V) { return std::max(1, std::min(8, (EW_ROWS - 32) / V + 1)); }

struct EwGeom {
    int CT, NT, NM, F, tiles_t, gtiles, tps, nseg, n_cg, n_og;
    bool ok;
};
static EwGeom ew_geom(int B, int T, int V, int ic, int Cx) {
    EwGeom g;
    const int Ce = 6 * ic;
    if (ic < 32) {                                                   // ic = 16: all six groups in one 128-channel workgroup (two idle waves)
        g.CT = 8, g.NM = 6, g.NT = 4;
    } else {
        g.CT = (Ce % 128 == 0 && ic % 64 == 0) ? 8 : 4;
        g.NM = 2;
        g.NT = Cx % 128 == 0 ? 8 : 4;
    }
    g.F = ew_frames(V);
    g.tiles_t = (int)cdiv(T, g.F);
    g.gtiles = B * g.tiles_t;
    if (ic >= 32 && fgcn::tuning(21) == 2) g.CT = 4, g.NT = 4;        // (tuning key 21 = 2: 64 x 64 tiles; see fgcn_spatial_wgrad_tile.hip, swt_geom)
    g.n_cg = (int)cdiv(Ce, 16 * g.CT);
    g.n_og = Cx / (16 * g.NT);
    // every workgroup's channels must touch at most NM (subset, side) groups
    g.ok = true;
    for (int cg = 0; cg < g.n_cg; ++cg) {
        const int lo = cg * 16 * g.CT, hi = std::min(lo + 16 * g.CT, Ce) - 1;
        if (hi / ic - lo / ic + 1 > g.NM) g.ok = false;
    }
    const int want = std::max(1, (fgcn::tuning(17) > 0 ? fgcn::tuning(17) : 256) / (g.n_cg * g.n_og));   // one workgroup per CU
    g.tps = (int)cdiv(g.gtiles, std::min(g.gtiles, want));
    g.nseg = (int)cdiv(g.gtiles, g.tps);
    return g;
}

static bool emb_tile_mode_ok() { return fgcn::math_mode() == FGCN_MATH_BF16X3 || fgcn::math_mode() == FGCN_MATH_BF16; }
static bool emb_tile_sizes_ok(int V, int ic, int Cx) {
    return V >= 16 && V <= FGCN_MAX_V && ic >= 16 && ic % 16 == 0 && Cx >= 64 && Cx % 64 == 0;
}

}  // namespace fgcn

using namespace fgcn;

// 1 when fgcn_emb_dx_tile / fgcn_emb_wgrad_tile run these sizes in the current math mode (FGCN_MATH_BF16X3 with either product form -- the
// kernels always multiply three-way bf16 splits there -- or FGCN_MATH_BF16; 16 .. 32 joints; ic a multiple of 16; Cx in 64s)
extern "C" int fgcn_emb_tile_available(int V, int ic, int Cx) {
    if (!emb_tile_mode_ok() || !emb_tile_sizes_ok(V, ic, Cx)) return 0;
    return ew_geom(1, 1, V, ic, Cx).ok ? 1 : 0;
}

// bytes of fgcn_emb_dx_tile's workspace (the split matrix planes of every sample) in the current math mode
extern "C" long long fgcn_emb_dx_tile_workspace(int B, int d_s_batched) {
    const long long per = fgcn::math_mode() == FGCN_MATH_BF16 ? et_planes_bytes<1>() : et_planes_bytes<3>();
    return (d_s_batched ? (long long)B : 1ll) * per;
}

// one instantiation of the dx kernel (LDS opt-in once per instantiation; not a stream operation: stays out of graph captures); the bfloat16-emb
// form exists for the one-part kernel
template <int NP, int NT, int MU, bool ACC, int PD, int RS>
static void ed_go(int e16, dim3 grid, hipStream_t s, const EmbDxP& p) {      // e16: 1 = emb bfloat16, 3 = emb and dx
    if constexpr (NP == 1 && ACC) {
        if (e16 == 7) {
            static bool opted167 = false;
            if (!opted167) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&emb_dx_tile_kernel<NP, NT, MU, ACC, PD, RS, 7>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, ed_lds<NP>());
                opted167 = true;
            }
            hipLaunchKernelGGL((emb_dx_tile_kernel<NP, NT, MU, ACC, PD, RS, 7>), grid, dim3(256), ed_lds<NP>(), s, p);
            return;
        }
    }
    if constexpr (NP == 1) {
        if (e16 == 3) {
            static bool opted163 = false;
            if (!opted163) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&emb_dx_tile_kernel<NP, NT, MU, ACC, PD, RS, 3>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, ed_lds<NP>());
                opted163 = true;
            }
            hipLaunchKernelGGL((emb_dx_tile_kernel<NP, NT, MU, ACC, PD, RS, 3>), grid, dim3(256), ed_lds<NP>(), s, p);
            return;
        }
        if (e16) {
            static bool opted16 = false;
            if (!opted16) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&emb_dx_tile_kernel<NP, NT, MU, ACC, PD, RS, 1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, ed_lds<NP>());
                opted16 = true;
            }
            hipLaunchKernelGGL((emb_dx_tile_kernel<NP, NT, MU, ACC, PD, RS, 1>), grid, dim3(256), ed_lds<NP>(), s, p);
            return;
        }
    }
    static bool opted = false;
    if (!opted) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&emb_dx_tile_kernel<NP, NT, MU, ACC, PD, RS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, ed_lds<NP>());
        opted = true;
    }
    hipLaunchKernelGGL((emb_dx_tile_kernel<NP, NT, MU, ACC, PD, RS>), grid, dim3(256), ed_lds<NP>(), s, p);
}

static int emb_dx_tile_impl(const float* emb, const float* d_s, const void* w3, float* dx, void* workspace, int B, int T, int V, int ic, int Cx,
                            int ld_e, int ld_dx, int d_s_batched, int accumulate, void* stream, int e16, const float* dx_old = nullptr);

extern "C" int fgcn_emb_dx_tile(const float* emb, const float* d_s, const void* w3, float* dx, void* workspace, int B, int T, int V, int ic, int Cx,
                                int ld_e, int ld_dx, int d_s_batched, int accumulate, void* stream) {
    return emb_dx_tile_impl(emb, d_s, w3, dx, workspace, B, T, V, ic, Cx, ld_e, ld_dx, d_s_batched, accumulate, stream, 0);
}

// emb as a BFLOAT16 tensor (fgcn_emb_fwd_tile_h; math mode bf16 only; ld_e in elements): bit-identical to the f32-emb call on the same values
extern "C" int fgcn_emb_dx_tile_h(const unsigned short* emb_h, const float* d_s, const void* w3, float* dx, void* workspace, int B, int T, int V,
                                  int ic, int Cx, int ld_e, int ld_dx, int d_s_batched, int accumulate, void* stream) {
    return emb_dx_tile_impl(reinterpret_cast<const float*>(emb_h), d_s, w3, dx, workspace, B, T, V, ic, Cx, ld_e, ld_dx, d_s_batched, accumulate, stream,
                            1);
}

// typed form (math mode bf16): half_mask bit 0 = emb is a bfloat16 tensor, bit 1 = dx is (masks 0, 1, 3); strides in elements.
// dx_old (mask 3, accumulate): a float32 tensor laid out like dx that holds the values to add to -- dx itself is then only written
// (dx = bfloat16(dx_old + term): the last writer of a float32-accumulated gradient hands it over as bfloat16); NULL: dx is read and written.
extern "C" int fgcn_emb_dx_tile_t(const void* emb, const float* d_s, const void* w3, void* dx, void* workspace, int B, int T, int V,
                                  int ic, int Cx, int ld_e, int ld_dx, int d_s_batched, int accumulate, const float* dx_old, int half_mask,
                                  void* stream) {
    FGCN_REQUIRE(half_mask == 0 || half_mask == 1 || half_mask == 3, FGCN_E_BADARG, "emb_dx_tile_t: half_mask=%d (0, 1 or 3)", half_mask);
    FGCN_REQUIRE(!dx_old || (half_mask == 3 && accumulate && aligned16(dx_old)), FGCN_E_BADARG,
                 "emb_dx_tile_t: dx_old comes with a bfloat16 emb and dx and accumulation");
    return emb_dx_tile_impl(static_cast<const float*>(emb), d_s, w3, static_cast<float*>(dx), workspace, B, T, V, ic, Cx, ld_e, ld_dx, d_s_batched,
                            accumulate, stream, dx_old ? 7 : half_mask, dx_old);
}

static int emb_dx_tile_impl(const float* emb, const float* d_s, const void* w3, float* dx, void* workspace, int B, int T, int V, int ic, int Cx,
                            int ld_e, int ld_dx, int d_s_batched, int accumulate, void* stream, int e16, const float* dx_old) {
    FGCN_REQUIRE(emb && d_s && w3 && dx && workspace, FGCN_E_BADARG, "emb_dx_tile: null pointer");
    FGCN_REQUIRE(!e16 || fgcn::math_mode() == FGCN_MATH_BF16, FGCN_E_BADARG, "emb_dx_tile_h: bfloat16 tensors need math mode bf16");
    FGCN_REQUIRE(B > 0 && T > 0, FGCN_E_BADARG, "emb_dx_tile: bad sizes B=%d T=%d", B, T);
    FGCN_REQUIRE(emb_tile_mode_ok() && emb_tile_sizes_ok(V, ic, Cx), FGCN_E_BADARG,
                 "emb_dx_tile: needs math mode bf16x3 or bf16, 16 <= V <= %d, ic %% 16 == 0, Cx %% 64 == 0 (V=%d ic=%d Cx=%d, mode %d)", FGCN_MAX_V, V,
                 ic, Cx, fgcn::math_mode());
    const int Ce = 6 * ic;
    FGCN_REQUIRE(ld_e % 4 == 0 && ld_dx % 4 == 0 && ld_e >= Ce && ld_dx >= Cx, FGCN_E_ALIGN, "emb_dx_tile: row strides");
    FGCN_REQUIRE(aligned16(emb) && aligned16(w3) && aligned16(dx) && aligned16(workspace) && (reinterpret_cast<uintptr_t>(d_s) & 3u) == 0, FGCN_E_ALIGN,
                 "emb_dx_tile: 16-byte alignment");
    const long long e_bytes = (long long)B * T * V * ld_e * (e16 ? 2 : 4), dx_bytes = (long long)B * T * V * ld_dx * ((e16 & 2) ? 2 : 4);
    const long long plane = (long long)Ce * Cx * 2;
    FGCN_REQUIRE(e_bytes < 0x7FFF0000ll && dx_bytes < 0x7FFF0000ll && plane * 3 < 0x7FFF0000ll, FGCN_E_BADARG,
                 "emb_dx_tile: tensors must be smaller than 2 GiB (32-bit buffer offsets)");
    const int np = fgcn::math_mode() == FGCN_MATH_BF16 ? 1 : 3;
    EmbDxP p;
    p.emb = emb; p.planes = static_cast<const unsigned char*>(workspace); p.w3 = w3; p.dx = dx;
    p.dx_old = dx_old; p.old_bytes = dx_old ? (unsigned)(dx_bytes * 2) : 0u;
    p.B = B; p.T = T; p.V = V; p.ic = ic; p.Ce = Ce; p.Cout = Cx; p.ld_e = ld_e; p.ld_dx = ld_dx; p.s_batched = d_s_batched;
    p.ic_inv = (unsigned)(((1ull << 32) + (unsigned)ic - 1) / (unsigned)ic);
    p.F = 128 / V;
    p.tiles_t = (int)cdiv(T, p.F);
    p.tiles_m = B * p.tiles_t;
    const bool narrow = Cx <= 64;
    p.tiles_n = (int)cdiv(Cx, narrow ? 64 : 128);
    const long long total = (long long)p.tiles_m * p.tiles_n;
    FGCN_REQUIRE(total < (1ll << 30), FGCN_E_BADARG, "emb_dx_tile: too many tiles");
    p.per_xcd = (int)cdiv(total, 8);
    p.e_bytes = (unsigned)e_bytes; p.dx_bytes = (unsigned)dx_bytes; p.w_plane_bytes = (unsigned)plane;
    const dim3 grid((unsigned)(p.per_xcd * 8));
    hipStream_t s = (hipStream_t)stream;
    // the split matrix planes, once per sample
    if (np == 3) hipLaunchKernelGGL((emb_planes_kernel<3>), dim3(d_s_batched ? B : 1), dim3(256), 0, s, d_s, static_cast<unsigned char*>(workspace), V);
    else hipLaunchKernelGGL((emb_planes_kernel<1>), dim3(d_s_batched ? B : 1), dim3(256), 0, s, d_s, static_cast<unsigned char*>(workspace), V);
    const bool big = 2 * p.F > 12;                                   // mixing units per wave and chunk: ceil(2 F / 4)
#define FGCN_ED_GO6(NP_, NT_, MU_, ACC_, PD_, RS_) ed_go<NP_, NT_, MU_, ACC_, PD_, RS_>(e16, grid, s, p)
    /* tuning key 18 = 1: 128-column tiles with a two-slot weight ring (default four; 64-column tiles always two) */
#define FGCN_ED_GO4(NP_, NT_, MU_, ACC_)                                   \
    do {                                                                   \
        if (NT_ == 1) FGCN_ED_GO6(NP_, 1, MU_, ACC_, 1, 2);                \
        else if (variant == 1) FGCN_ED_GO6(NP_, 2, MU_, ACC_, 1, 2);       \
        else FGCN_ED_GO6(NP_, 2, MU_, ACC_, 1, 4);                         \
    } while (0)
#define FGCN_ED_GO3(NP_, NT_, MU_)                     \
    do {                                               \
        if (accumulate) FGCN_ED_GO4(NP_, NT_, MU_, true); \
        else FGCN_ED_GO4(NP_, NT_, MU_, false);        \
    } while (0)
#define FGCN_ED_GO2(NP_, NT_)                  \
    do {                                       \
        if (big) FGCN_ED_GO3(NP_, NT_, 4);     \
        else FGCN_ED_GO3(NP_, NT_, 3);         \
    } while (0)
#define FGCN_ED_GO(NP_)                        \
    do {                                       \
        if (narrow) FGCN_ED_GO2(NP_, 1);       \
        else FGCN_ED_GO2(NP_, 2);              \
    } while (0)
    const int variant = fgcn::tuning(18);
    if (np == 3) FGCN_ED_GO(3);
    else FGCN_ED_GO(1);
#undef FGCN_ED_GO
#undef FGCN_ED_GO2
#undef FGCN_ED_GO3
#undef FGCN_ED_GO4
#undef FGCN_ED_GO6
    return launch_status("emb_dx_tile");
}

// slabs of `partial` / `bias_partial` (0: sizes the kernel does not take)
extern "C" int fgcn_emb_wgrad_tile_slabs(int B, int T, int V, int ic, int Cx) {
    if (B <= 0 || T <= 0 || !emb_tile_sizes_ok(V, ic, Cx)) return 0;
    const EwGeom g = ew_geom(B, T, V, ic, Cx);
    return g.ok ? g.nseg : 0;
}

template <int NP, int CT, int NT, int NS, int NM, int PF>
static void ew_go(int e16, dim3 grid, hipStream_t s, const EmbWgP& p) {      // e16: 1 = emb bfloat16, 3 = emb and x
    constexpr int lds_ = ew_lds<NP, NT, NM>();
    if constexpr (NP == 1) {
        if (e16 == 3) {
            static bool attr163 = false;
            if (!attr163) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&emb_wgrad_tile_kernel<NP, CT, NT, NS, NM, PF, 3>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_);
                attr163 = true;
            }
            hipLaunchKernelGGL((emb_wgrad_tile_kernel<NP, CT, NT, NS, NM, PF, 3>), grid, dim3(512), lds_, s, p);
            return;
        }
        if (e16) {
            static bool attr16 = false;
            if (!attr16) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&emb_wgrad_tile_kernel<NP, CT, NT, NS, NM, PF, 1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_);
                attr16 = true;
            }
            hipLaunchKernelGGL((emb_wgrad_tile_kernel<NP, CT, NT, NS, NM, PF, 1>), grid, dim3(512), lds_, s, p);
            return;
        }
    }
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&emb_wgrad_tile_kernel<NP, CT, NT, NS, NM, PF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_);
        attr = true;
    }
    hipLaunchKernelGGL((emb_wgrad_tile_kernel<NP, CT, NT, NS, NM, PF>), grid, dim3(512), lds_, s, p);
}

static int emb_wgrad_tile_impl(const float* emb, const float* x, const float* d_s, float* partial, float* bias_partial, int B, int T,
                               int V, int ic, int Cx, int ld_e, int ld_x, int d_s_batched, void* stream, int e16);

extern "C" int fgcn_emb_wgrad_tile(const float* emb, const float* x, const float* d_s, float* partial, float* bias_partial, int B, int T,
                                   int V, int ic, int Cx, int ld_e, int ld_x, int d_s_batched, void* stream) {
    return emb_wgrad_tile_impl(emb, x, d_s, partial, bias_partial, B, T, V, ic, Cx, ld_e, ld_x, d_s_batched, stream, 0);
}

// emb as a BFLOAT16 tensor (fgcn_emb_fwd_tile_h; math mode bf16 only; ld_e in elements)
extern "C" int fgcn_emb_wgrad_tile_h(const unsigned short* emb_h, const float* x, const float* d_s, float* partial, float* bias_partial, int B,
                                     int T, int V, int ic, int Cx, int ld_e, int ld_x, int d_s_batched, void* stream) {
    return emb_wgrad_tile_impl(reinterpret_cast<const float*>(emb_h), x, d_s, partial, bias_partial, B, T, V, ic, Cx, ld_e, ld_x, d_s_batched, stream,
                               1);
}

// typed form (math mode bf16): half_mask bit 0 = emb is a bfloat16 tensor, bit 1 = x is (masks 0, 1, 3); strides in elements
extern "C" int fgcn_emb_wgrad_tile_t(const void* emb, const void* x, const float* d_s, float* partial, float* bias_partial, int B,
                                     int T, int V, int ic, int Cx, int ld_e, int ld_x, int d_s_batched, int half_mask, void* stream) {
    FGCN_REQUIRE(half_mask == 0 || half_mask == 1 || half_mask == 3, FGCN_E_BADARG, "emb_wgrad_tile_t: half_mask=%d (0, 1 or 3)", half_mask);
    return emb_wgrad_tile_impl(static_cast<const float*>(emb), static_cast<const float*>(x), d_s, partial, bias_partial, B, T, V, ic, Cx, ld_e, ld_x,
                               d_s_batched, stream, half_mask);
}

static int emb_wgrad_tile_impl(const float* emb, const float* x, const float* d_s, float* partial, float* bias_partial, int B, int T,
                               int V, int ic, int Cx, int ld_e, int ld_x, int d_s_batched, void* stream, int e16) {
    FGCN_REQUIRE(emb && x && d_s && partial && bias_partial, FGCN_E_BADARG, "emb_wgrad_tile: null pointer");
    FGCN_REQUIRE(!e16 || fgcn::math_mode() == FGCN_MATH_BF16, FGCN_E_BADARG, "emb_wgrad_tile_h: bfloat16 tensors need math mode bf16");
    FGCN_REQUIRE(B > 0 && T > 0, FGCN_E_BADARG, "emb_wgrad_tile: bad sizes B=%d T=%d", B, T);
    FGCN_REQUIRE(fgcn_emb_tile_available(V, ic, Cx), FGCN_E_BADARG,
                 "emb_wgrad_tile: V=%d ic=%d Cx=%d in math mode %d not supported (bf16x3 or bf16, 16 <= V <= %d, ic %% 16 == 0, Cx in 64s)", V, ic,
                 Cx, fgcn::math_mode(), FGCN_MAX_V);
    const int Ce = 6 * ic;
    FGCN_REQUIRE(ld_e >= Ce && ld_x >= Cx && ld_x % 4 == 0, FGCN_E_BADARG, "emb_wgrad_tile: bad row strides ld_e=%d ld_x=%d", ld_e, ld_x);
    FGCN_REQUIRE(aligned16(x) && (reinterpret_cast<uintptr_t>(emb) & 3u) == 0 && (reinterpret_cast<uintptr_t>(partial) & 3u) == 0 &&
                     (reinterpret_cast<uintptr_t>(bias_partial) & 3u) == 0 && (reinterpret_cast<uintptr_t>(d_s) & 3u) == 0,
                 FGCN_E_ALIGN, "emb_wgrad_tile: x must be 16-byte aligned (emb, d_s, partial, bias_partial: 4)");
    const long long rows = (long long)B * T * V;
    FGCN_REQUIRE(rows * ld_e * 4 < (1ll << 31) && rows * ld_x * 4 < (1ll << 31), FGCN_E_BADARG, "emb_wgrad_tile: tensors must be smaller than 2 GiB");
    const EwGeom g = ew_geom(B, T, V, ic, Cx);
    FGCN_REQUIRE((long long)g.nseg * Ce * Cx * 4 < (1ll << 31), FGCN_E_BADARG, "emb_wgrad_tile: partial slabs must be smaller than 2 GiB");
    const int np = fgcn::math_mode() == FGCN_MATH_BF16 ? 1 : 3;
    EmbWgP p;
    p.emb = emb, p.x = x, p.d_s = d_s, p.partial = partial, p.bias_partial = bias_partial;
    p.B = B, p.T = T, p.V = V, p.ic = ic, p.Ce = Ce, p.Cx = Cx, p.ld_e = ld_e, p.ld_x = ld_x, p.s_batched = d_s_batched;
    p.F = g.F, p.tiles_t = g.tiles_t, p.gtiles = g.gtiles, p.tps = g.tps, p.nseg = g.nseg, p.n_cg = g.n_cg, p.n_og = g.n_og;
    p.e_bytes = (unsigned)(rows * ld_e * (e16 ? 2 : 4)), p.x_bytes = (unsigned)(rows * ld_x * (e16 == 3 ? 2 : 4));
    p.p_bytes = (unsigned)((long long)g.nseg * Ce * Cx * 4), p.b_bytes = (unsigned)((long long)g.nseg * Ce * 4);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(g.nseg * g.n_cg * g.n_og));
#define FGCN_EW6(NP_, CT_, NT_, NS_, NM_, PF_) ew_go<NP_, CT_, NT_, NS_, NM_, PF_>(e16, grid, s, p)
    /* tuning key 19: slots the emb values are requested ahead (0 = two, 1 = one) */
#define FGCN_EW(NP_, CT_, NT_, NS_, NM_)                           \
    do {                                                           \
        if (pf1) FGCN_EW6(NP_, CT_, NT_, NS_, NM_, 1);             \
        else FGCN_EW6(NP_, CT_, NT_, NS_, NM_, 2);                 \
    } while (0)
    const bool pf1 = fgcn::tuning(19) == 1;
    // frame slots of a wave per tile: F frames over 8 / CT waves per channel tile, rounded up to even
    const int nslot = ((g.F + 8 / g.CT - 1) / (8 / g.CT) + 1) & ~1;
    FGCN_REQUIRE(nslot == (g.CT == 8 ? (g.F > 6 ? 8 : 6) : 4), FGCN_E_BADARG, "emb_wgrad_tile: %d frames per tile: no such kernel form", g.F);
#define FGCN_EW_NP(CT_, NT_, NS_, NM_)                     \
    do {                                                   \
        if (np == 3) FGCN_EW(3, CT_, NT_, NS_, NM_);       \
        else FGCN_EW(1, CT_, NT_, NS_, NM_);               \
    } while (0)
    if (g.NM == 6) {                                                 // ic = 16
        if (nslot == 8) FGCN_EW_NP(8, 4, 8, 6);
        else FGCN_EW_NP(8, 4, 6, 6);
    } else if (g.CT == 8) {
        if (g.NT == 8) {
            if (nslot == 8) FGCN_EW_NP(8, 8, 8, 2);
            else FGCN_EW_NP(8, 8, 6, 2);
        } else {
            if (nslot == 8) FGCN_EW_NP(8, 4, 8, 2);
            else FGCN_EW_NP(8, 4, 6, 2);
        }
    } else {
        if (g.NT == 8) FGCN_EW_NP(4, 8, 4, 2);
        else FGCN_EW_NP(4, 4, 4, 2);
    }
#undef FGCN_EW_NP
#undef FGCN_EW
#undef FGCN_EW6
    return launch_status("emb_wgrad_tile");
}
