// conv_d weight gradient with the joint aggregation on chip, tile form (north-star kernel 1, weight-gradient side, split-bf16 math mode):
//
//     agg_k[(n,t,w), c] = sum_v x[(n,t,v), c] A^_k[n][v][w]                      (never written to HBM)
//     dWd_k[c][o]       = sum_{n,t,w} agg_k[(n,t,w), c] dy[(n,t,w), o]           (partial sums per workgroup; the caller adds the slabs)
//
// reference: the autograd backward of SpatialGraphConv.forward with respect to conv_d's weights, torch_src/models/mmargcn/agcn.py:103-111
// (SURVEY.md Appendix A.2).
//
// fgcn_spatial_wgrad (fgcn_joint.hip) gives a wave a frame and a workgroup a 32 x 64 tile of the gradient, so x is read Cout / 64 times and dy
// Cin / 32 times: beyond 128 outputs the step fell back to joint_mix_vec (agg, three activations wide, through HBM) + the 1x1 weight-gradient
// GEMM.  Here a workgroup (8 waves, one per CU) owns a (16 CT input channels) x (16 NT output channels) tile of all three subsets and walks
// a contiguous range of (sample, frame tile) pairs:
//   * the dY tile (F frames, <= 160 rows x 16 NT channels) goes through registers (requested one tile ahead) into LDS as three bf16 planes,
//     split once;
//   * wave (ct, fp) takes input channels 16 ct .. + 15 and the frames fp, fp + FP, .. of the tile (FP = 8 / CT).  Per frame: the x values
//     x[(f, v = 8 g + j)][c = lane % 16] arrive as eight strided dword loads per lane (requested two frames ahead; this IS the B fragment of the
//     joint mixing, no LDS), split once; agg_k^T... agg_k (32 joints w x 16 channels) = A^_k^T . x_f on the matrix pipe (A^ planes in LDS); the
//     accumulator registers of lane (c, g) hold w = 4 g + r and 16 + 4 g + r: after an in-register split they ARE the A fragment of the
//     contraction over rows (k slot j <-> joint w(g, j)), whose B fragments are transposing LDS reads (ds_read_b64_tr_b16) of the dY planes in
//     the same joint order: dWd_k (16 c x 16 o) += agg_kf^T . dY_f.
//   * joints w >= V meet zero rows of the A^ planes (their agg is exactly zero), so the rows of the next frame that a 32-joint fragment
//     reaches cost nothing but must be finite: rows past the tile are staged as zeros.
// Every sum has a fixed order (bitwise reproducible).
#include <algorithm>
#include <type_traits>
#include <utility>

#include "fgcn_common.hpp"

// Timing probes (wrong results; tools/build_probe.py only): bit 0 = no contraction MFMAs, bit 1 = no mixing MFMAs, bit 2 = x values requested
// for the first slot only, bit 3 = dY rows requested for the first tile only, bit 4 = no deposit of the dY tile (first tile only),
// bit 5 = no transposing reads (one fragment per slot), bit 6 = no in-register splits of the aggregation
#ifndef FGCN_PROBE_SW
#define FGCN_PROBE_SW 0
#endif
#ifndef FGCN_SWT_PF
#define FGCN_SWT_PF 1                   // slots the x values are requested ahead (1 or 2; 2 measured equal in the kernel loop, 0.1 ms behind in the step)
#endif

namespace fgcn {

struct SwTileP {
    const float* x;
    const float* dy;
    const float* a_hat;
    float* partial;                     // float[nseg][3 Cin][Cout]
    int B, T, V, Cin, Cout, ld_x, ld_dy, a_batched;
    int F, tiles_t, gtiles, tps, nseg;  // frames per tile, tiles per sample, B * tiles_t, (sample, tile) pairs per segment, segments
    int n_cg, n_og;                     // input-channel / output-channel groups of a workgroup tile
    unsigned x_bytes, dy_bytes, p_bytes;
};

constexpr int SWT_ROWS = 160;           // rows of a dY plane: (F - 1) V + 32 <= 160
constexpr int SWT_AHB = 80;             // bytes per [w] row of a split A^ plane (32 joints v x bf16 + 16 pad)
// row stride of a dY plane: the channels' bytes + 32 -- eight consecutive rows then start 32 bytes apart modulo 256, which is what a
// transposing read's half wave touches (8 rows x 32 bytes)
template <int NT> constexpr int swt_rs() { return NT * 32 + 32; }
template <int NT> constexpr int swt_lds() { return 3 * SWT_ROWS * swt_rs<NT>() + 9 * 32 * SWT_AHB; }

__device__ __forceinline__ u32x2 swt_read_tr16(const unsigned char* p) {
    using v4s = __attribute__((ext_vector_type(4))) short;
    const v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
    return __builtin_bit_cast(u32x2, v);
}

template <class Fn, int... S>
__device__ __forceinline__ void swt_for_slots(Fn&& fn, std::integer_sequence<int, S...>) {
    (fn(std::integral_constant<int, S>{}), ...);
}

// CT: 16-channel input tiles of the workgroup (8: every wave walks all frames; 4: two waves per tile take alternate frames and their
// accumulators are added at the end); NT: 16-channel output tiles (4 or 8); NSLOT: frame slots of a wave per tile (ceil(F / FP) rounded up to
// even -- compile time: the slots of a tile are straight-line code, so that the compiler counts the outstanding requests exactly (a run-time
// slot loop put an s_waitcnt vmcnt(0) and register copies on its back edge: every request made ahead was waited for one slot later))
// NP: bf16 parts per operand -- 3: exact three-way splits (FGCN_MATH_BF16X3), 1: operands rounded to bfloat16 once (FGCN_MATH_BF16; the LDS layout
// keeps room for three parts, the first is used)
// IN16 (NP = 1): dy is a BFLOAT16 tensor (fgcn_bn_act_bwd_apply_h; ld_dy in elements): its rows are copied into the plane, half the reads
// IN16 = 3: x is a BFLOAT16 tensor as well (half-precision activation storage, the `_t` entry point; ld_x in elements): 2-byte loads of the
// values the float32 form would round to the same 16 bits
template <int CT, int NT, int NSLOT, int NP = 3, int IN16 = 0>
__global__ __launch_bounds__(512, 1) void spatial_wgrad_tile_x3_kernel(SwTileP p) {
    constexpr int LP = 3, FP = 8 / CT;
    static_assert(NP == 1 || NP == 3, "parts");
    static_assert(!IN16 || NP == 1, "bfloat16 dy: the one-part kernel");
    constexpr int RS = swt_rs<NT>(), PL = SWT_ROWS * RS;
    constexpr int GPR = NT * 4, RPP = 512 / GPR, NPASS = SWT_ROWS / RPP;    // 16-byte groups per row, rows per pass, passes
    static_assert(SWT_ROWS % RPP == 0 && NSLOT % 2 == 0, "staging passes / slot pairs");
    constexpr int PF = FGCN_SWT_PF;
    // the dY rows of the NEXT tile are requested in pieces, pass i in slot i * SPREAD / NPASS: in one burst at the start of the tile (80 KB per
    // CU, all CUs at once) the x values requested right after it came back behind the whole burst (requests return in order)
    constexpr int SPREAD = NSLOT > 2 ? NSLOT - 2 : 1;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char sw_lds[];
    unsigned char* Im = sw_lds;
    unsigned char* Ah = sw_lds + LP * PL;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4, q4 = l15 >> 2, c4 = lane & 3;
    const int ct = wave % CT, fp = wave / CT;
    const int combos = p.n_cg * p.n_og;
    const int seg = blockIdx.x / combos, combo = blockIdx.x - seg * combos;
    const int cgi = combo / p.n_og, ogi = combo - cgi * p.n_og;
    const int c0 = cgi * 16 * CT + 16 * ct, o0 = ogi * 16 * NT;
    const int V = p.V, F = p.F;
    const int g_lo = seg * p.tps, g_hi = min(g_lo + p.tps, p.gtiles);

    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);

    f32x4 acc[3][NT];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[k][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int srow = tid / GPR, sg = tid % GPR;
    f32x4 stg[NPASS];
    // pass i of the dY rows of (sample, tile) pair g (nothing past the segment: branch-free)
    auto fetch_pass = [&](int g, int i) {
        const int n_ = g / p.tiles_t, tile_ = g - n_ * p.tiles_t;
        const int t0_ = tile_ * F;
        const int nrows_ = g < g_hi ? min(F, p.T - t0_) * V : 0;
        const unsigned row0_ = (unsigned)((n_ * p.T + t0_) * V);
        const int r = srow + RPP * i;
        const unsigned off = r < nrows_ ? ((row0_ + r) * (unsigned)p.ld_dy + (unsigned)(o0 + 4 * sg)) * (IN16 ? 2u : 4u) : OOB;
        if constexpr (IN16) {                                    // four bfloat16 = 8 bytes, parked in the first two components
            const u32x2 h = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rdy, off, 0, 0));
            const unsigned b0 = h[0], b1 = h[1];                 // (element -> scalar before a bit cast: hipcc 7.2 reads element 0 otherwise)
            stg[i] = f32x4{__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1), 0.f, 0.f};
        } else {
            stg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, off, 0, 0));
        }
    };
    // x values of frame slot s of pair g for this wave: lane (c = l15, g4) <- x[(f, v = 8 g4 + j)][c0 + l15], j = 0 .. 7; requested PF slots
    // ahead (PF = 2: two register sets, by slot parity)
    float xr2[2][8];
    auto xfetch = [&](float (&xr)[8], int g, int s) {
        const int n_ = g / p.tiles_t, tile_ = g - n_ * p.tiles_t;
        const int t0_ = tile_ * F, f = fp + FP * s;
        const int vlim = (g < g_hi && f < min(F, p.T - t0_)) ? V : 0;      // (a scalar select: no frame, no joints -- branch-free loads)
        const unsigned base = (unsigned)((n_ * p.T + t0_ + f) * V + 8 * g4) * (unsigned)p.ld_x + (unsigned)(c0 + l15);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned off = 8 * g4 + j < vlim ? (base + (unsigned)(j * p.ld_x)) * ((IN16 & 2) ? 2u : 4u) : OOB;
            if constexpr ((IN16 & 2) != 0) xr[j] = __builtin_bit_cast(float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rx, off, 0, 0) << 16);
            else xr[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
        }
    };
    // the A^ planes of sample n -> LDS (between two barriers): A^_k split once, planes [subset][part][w][v] bf16 (one ds_read_b128 = the 8
    // joints v of a lane's fragment)
    auto planes = [&](int n) {
        const float* asrc = p.a_hat + (p.a_batched ? (long long)n * 3 * V * V : 0);
        for (int i = tid; i < 3 * 32 * 32; i += 512) {
            const int k = i >> 10, w = (i >> 5) & 31, v = i & 31;
            const float a = (v < V && w < V) ? asrc[(k * V + v) * V + w] : 0.f;
            unsigned ph, pm, pl;
            split_bf16_pair(a, 0.f, ph, pm, pl);
            unsigned short* d = reinterpret_cast<unsigned short*>(Ah + ((k * LP) * 32 + w) * SWT_AHB) + v;
            d[0] = (unsigned short)ph;
            if constexpr (NP == 3) {
                d[32 * SWT_AHB / 2] = (unsigned short)pm;
                d[2 * 32 * SWT_AHB / 2] = (unsigned short)pl;
            }
        }
    };
    auto deposit = [&]() {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int r = srow + RPP * i;
            u32x2 parts[NP];
            if constexpr (IN16) {                                // already bfloat16: a copy
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
    for (int i = 0; i < NPASS; ++i) fetch_pass(g_lo, i);
    xfetch(xr2[0], g_lo, 0);
    if (PF == 2) xfetch(xr2[1], g_lo, 1);
    if (g_lo < g_hi) planes(g_lo / p.tiles_t);
    deposit();
    __syncthreads();

    for (int g = g_lo; g < g_hi; ++g) {
        const int n = g / p.tiles_t, tile = g - n * p.tiles_t;
        const int nf = min(F, p.T - tile * F);
        auto slot = [&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            float (&xr)[8] = xr2[PF == 2 ? (s & 1) : 0];
            const int f = fp + FP * s;
            u32x4v xs[NP];
            splitn_x8<NP>(xr[0], xr[1], xr[2], xr[3], xr[4], xr[5], xr[6], xr[7], xs);
            if (!(FGCN_PROBE_SW & 4)) {
                if constexpr (s + PF < NSLOT) xfetch(xr, g, s + PF);  // a later slot of this tile, or of the next tile
                else xfetch(xr, g + 1, s + PF - NSLOT);
            }
            if (!(FGCN_PROBE_SW & 8)) {
#pragma unroll
                for (int i = 0; i < NPASS; ++i)
                    if (i * SPREAD / NPASS == s) fetch_pass(g + 1, i);
            }
            if (f >= nf) return;                                     // wave-uniform: no such frame in this tile
            u32x4v a3[3][NP];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                f32x4 m[2];
#pragma unroll
                for (int wt = 0; wt < 2; ++wt) {
                    u32x4v af[NP];
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        af[pl] = *reinterpret_cast<const u32x4v*>(Ah + ((k * LP + pl) * 32 + 16 * wt + l15) * SWT_AHB + 16 * g4);
                    if constexpr ((FGCN_PROBE_SW & 2) != 0) m[wt] = __builtin_bit_cast(f32x4, af[0] ^ xs[0] ^ af[NP - 1] ^ xs[NP - 1]);
                    else m[wt] = mfma_np_k32<NP>(af, xs, f32x4{0.f, 0.f, 0.f, 0.f});
                }
                if constexpr ((FGCN_PROBE_SW & 64) != 0) {
                    a3[k][0] = __builtin_bit_cast(u32x4v, m[0]);
                    a3[k][NP - 1] = __builtin_bit_cast(u32x4v, m[1]);
                } else
                    splitn_x8<NP>(m[0][0], m[0][1], m[0][2], m[0][3], m[1][0], m[1][1], m[1][2], m[1][3], a3[k]);
            }
            const int r_lo = f * V + 4 * g4 + q4, r_hi = r_lo + 16;
            u32x4v df[NP];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    if ((FGCN_PROBE_SW & 32) && nt > 0) break;
                    const unsigned char* base = Im + pl * PL + nt * 32 + 8 * c4;
                    const u32x2 lo = swt_read_tr16(base + r_lo * RS);
                    const u32x2 hi = swt_read_tr16(base + r_hi * RS);
                    df[pl] = u32x4v{lo[0], lo[1], hi[0], hi[1]};
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if constexpr ((FGCN_PROBE_SW & 1) != 0) acc[k][nt][0] += __builtin_bit_cast(float, a3[k][0][0] ^ a3[k][NP - 1][1] ^ df[0][0] ^ df[NP - 1][1]);
                    else acc[k][nt] = mfma_np_k32<NP>(a3[k], df, acc[k][nt]);
                }
            }
        };
        swt_for_slots(slot, std::make_integer_sequence<int, NSLOT>{});
        // the next tile's dY rows (and its sample's A^ planes) replace this one's
        __syncthreads();                                             // this tile's fragment reads are done
        const int n1 = (g + 1) / p.tiles_t;
        if (g + 1 < g_hi && n1 != n && p.a_batched) planes(n1);
        if (!((FGCN_PROBE_SW & 16))) deposit();
        __syncthreads();
    }

    // ---- the workgroup's slab: partial[seg][k Cin + c][o] -------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)p.partial, 0, p.p_bytes, 0x00020000);
    if constexpr (FP == 2) {                                         // fixed order: frames of the even slots + frames of the odd slots
        __syncthreads();
        float* red = reinterpret_cast<float*>(sw_lds);               // [ct][k][nt][r][lane]
        if (fp == 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[(((ct * 3 + k) * NT + nt) * 4 + r) * 64 + lane] = acc[k][nt][r];
        }
        __syncthreads();
        if (fp == 1) return;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[k][nt][r] += red[(((ct * 3 + k) * NT + nt) * 4 + r) * 64 + lane];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned off = (((unsigned)seg * 3u + k) * (unsigned)p.Cin + (unsigned)(c0 + 4 * g4 + r)) * (unsigned)p.Cout + (unsigned)(o0 + 16 * nt + l15);
                const float val = acc[k][nt][r];                     // (a bit_cast of the vector-element lvalue itself reads element 0: hipcc 7.2)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rp, off * 4u, 0, 0);
            }
}

// frames per tile: as many whole frames as keep a 32-joint fragment of the last one inside SWT_ROWS rows, at most 8
static int swt_frames(int V) { return std::max(1, std::min(8, (SWT_ROWS - 32) / V + 1)); }

struct SwtGeom {
    int CT, NT, F, tiles_t, gtiles, tps, nseg, n_cg, n_og;
};
static SwtGeom swt_geom(int B, int T, int V, int Cin, int Cout) {
    SwtGeom g;
    g.CT = Cin % 128 == 0 ? 8 : 4;
    g.NT = Cout % 128 == 0 ? 8 : 4;
    g.F = swt_frames(V);
    g.tiles_t = (int)cdiv(T, g.F);
    g.gtiles = B * g.tiles_t;
    // Tuning key 21 = 2: 64 x 64 tiles whatever the channels allow.  (Tried as the rule for small batches -- with 128-wide tiles a 256 -> 256
    // layer has 64 row segments, at 8 clips each walks 3 (sample, frame tile) pairs and writes a 196 KB slab, 50 MB of slabs per launch,
    // which the block's reduction reads back; 64 x 64 tiles quarter the slab bytes -- and measured SLOWER in the replayed step: 8 clips
    // 9.16 -> 9.34 ms, 16 clips 16.11 -> 16.18, profiles/r05_ab_small_batch.txt: each operand is then read four times instead of twice.)
    if (fgcn::tuning(21) == 2) g.CT = 4, g.NT = 4;
    g.n_cg = Cin / (16 * g.CT);
    g.n_og = Cout / (16 * g.NT);
    // one workgroup per CU (tuning key 16 overrides the target)
    const int want = std::max(1, (fgcn::tuning(16) > 0 ? fgcn::tuning(16) : 256) / (g.n_cg * g.n_og));
    g.tps = (int)cdiv(g.gtiles, std::min(g.gtiles, want));
    g.nseg = (int)cdiv(g.gtiles, g.tps);
    return g;
}

}  // namespace fgcn

using namespace fgcn;

// 1 when fgcn_spatial_wgrad_tile runs these sizes in the current math mode (FGCN_MATH_BF16X3 with either product form: the kernel always
// multiplies three-way bf16 splits; whole 64-channel groups on both sides; 16..32 joints)
extern "C" int fgcn_spatial_wgrad_tile_available(int V, int Cin, int Cout) {
    return ((fgcn::math_mode() == FGCN_MATH_BF16X3 || fgcn::math_mode() == FGCN_MATH_BF16) && V >= 16 && V <= FGCN_MAX_V && Cin % 64 == 0 && Cin > 0 &&
            Cout % 64 == 0 && Cout > 0) ? 1 : 0;
}

// slabs of `partial` (0: sizes the kernel does not take)
extern "C" int fgcn_spatial_wgrad_tile_slabs(int B, int T, int V, int Cin, int Cout) {
    if (V < 16 || V > FGCN_MAX_V || B <= 0 || T <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64) return 0;
    return swt_geom(B, T, V, Cin, Cout).nseg;
}

static int spatial_wgrad_tile_launch(const float* x, const float* dy, const float* a_hat, float* partial, int B, int T, int V, int Cin,
                                     int Cout, int ld_x, int ld_dy, int a_hat_batched, void* stream, int dy16);

extern "C" int fgcn_spatial_wgrad_tile(const float* x, const float* dy, const float* a_hat, float* partial, int B, int T, int V, int Cin,
                                       int Cout, int ld_x, int ld_dy, int a_hat_batched, void* stream) {
    return spatial_wgrad_tile_launch(x, dy, a_hat, partial, B, T, V, Cin, Cout, ld_x, ld_dy, a_hat_batched, stream, 0);
}

// dy as a BFLOAT16 tensor (math mode bf16 only; ld_dy in elements); otherwise fgcn_spatial_wgrad_tile, bit-identical to its result on the
// f32 tensor fgcn_bn_act_bwd_apply would have written
extern "C" int fgcn_spatial_wgrad_tile_h(const float* x, const unsigned short* dy_h, const float* a_hat, float* partial, int B, int T, int V,
                                         int Cin, int Cout, int ld_x, int ld_dy, int a_hat_batched, void* stream) {
    return spatial_wgrad_tile_launch(x, reinterpret_cast<const float*>(dy_h), a_hat, partial, B, T, V, Cin, Cout, ld_x, ld_dy, a_hat_batched, stream,
                                     1);
}

// typed form (math mode bf16): half_mask bit 0 = x is a bfloat16 tensor, bit 1 = dy is (masks 0, 2, 3: a bfloat16 x comes with a bfloat16 dy)
extern "C" int fgcn_spatial_wgrad_tile_t(const void* x, const void* dy, const float* a_hat, float* partial, int B, int T, int V, int Cin,
                                         int Cout, int ld_x, int ld_dy, int a_hat_batched, int half_mask, void* stream) {
    FGCN_REQUIRE(half_mask == 0 || half_mask == 2 || half_mask == 3, FGCN_E_BADARG, "spatial_wgrad_tile_t: half_mask=%d (0, 2 or 3)", half_mask);
    return spatial_wgrad_tile_launch(static_cast<const float*>(x), static_cast<const float*>(dy), a_hat, partial, B, T, V, Cin, Cout, ld_x, ld_dy,
                                     a_hat_batched, stream, half_mask == 3 ? 3 : (half_mask == 2 ? 1 : 0));
}

static int spatial_wgrad_tile_launch(const float* x, const float* dy, const float* a_hat, float* partial, int B, int T, int V, int Cin,
                                     int Cout, int ld_x, int ld_dy, int a_hat_batched, void* stream, int dy16) {   // dy16: 1 = dy bfloat16, 3 = dy and x
    FGCN_REQUIRE(x && dy && a_hat && partial, FGCN_E_BADARG, "spatial_wgrad_tile: null pointer");
    FGCN_REQUIRE(!dy16 || fgcn::math_mode() == FGCN_MATH_BF16, FGCN_E_BADARG, "spatial_wgrad_tile_h: a bfloat16 dy needs math mode bf16");
    FGCN_REQUIRE(fgcn_spatial_wgrad_tile_available(V, Cin, Cout), FGCN_E_BADARG,
                 "spatial_wgrad_tile: V=%d Cin=%d Cout=%d in math mode %d not supported (split-bf16 mode, 16 <= V <= %d, channels in 64s)", V,
                 Cin, Cout, fgcn::math_mode(), FGCN_MAX_V);
    FGCN_REQUIRE(B > 0 && T > 0 && ld_x >= Cin && ld_dy >= Cout && ld_dy % 4 == 0, FGCN_E_BADARG,
                 "spatial_wgrad_tile: bad sizes B=%d T=%d ld_x=%d ld_dy=%d", B, T, ld_x, ld_dy);
    FGCN_REQUIRE(aligned16(dy) && (reinterpret_cast<uintptr_t>(x) & 3u) == 0 && (reinterpret_cast<uintptr_t>(partial) & 3u) == 0, FGCN_E_ALIGN,
                 "spatial_wgrad_tile: dy must be 16-byte aligned (x, partial: 4)");
    const long long rows = (long long)B * T * V;
    FGCN_REQUIRE(rows * ld_x * 4 < (1ll << 31) && rows * ld_dy * 4 < (1ll << 31), FGCN_E_BADARG,
                 "spatial_wgrad_tile: tensors must be smaller than 2 GiB");
    const SwtGeom g = swt_geom(B, T, V, Cin, Cout);
    FGCN_REQUIRE((long long)g.nseg * 3 * Cin * Cout * 4 < (1ll << 31), FGCN_E_BADARG, "spatial_wgrad_tile: partial slabs must be smaller than 2 GiB");
    SwTileP p;
    p.x = x, p.dy = dy, p.a_hat = a_hat, p.partial = partial;
    p.B = B, p.T = T, p.V = V, p.Cin = Cin, p.Cout = Cout, p.ld_x = ld_x, p.ld_dy = ld_dy, p.a_batched = a_hat_batched;
    p.F = g.F, p.tiles_t = g.tiles_t, p.gtiles = g.gtiles, p.tps = g.tps, p.nseg = g.nseg, p.n_cg = g.n_cg, p.n_og = g.n_og;
    p.x_bytes = (unsigned)(rows * ld_x * ((dy16 & 2) ? 2 : 4)), p.dy_bytes = (unsigned)(rows * ld_dy * (dy16 ? 2 : 4));
    p.p_bytes = (unsigned)((long long)g.nseg * 3 * Cin * Cout * 4);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(g.nseg * g.n_cg * g.n_og));
#define FGCN_SWT4(CT_, NT_, NS_, NP_, I16_)                                                                                      \
    do {                                                                                                                          \
        static bool attr = false;                                                                                                 \
        if (!attr) {                                                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_wgrad_tile_x3_kernel<CT_, NT_, NS_, NP_, I16_>),   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, swt_lds<NT_>());                                \
            attr = true;                                                                                                          \
        }                                                                                                                         \
        hipLaunchKernelGGL((spatial_wgrad_tile_x3_kernel<CT_, NT_, NS_, NP_, I16_>), grid, dim3(512), swt_lds<NT_>(), s, p);     \
    } while (0)
    const bool one_part = fgcn::math_mode() == FGCN_MATH_BF16;     // operands rounded to bfloat16 once
#define FGCN_SWT(CT_, NT_, NS_)                                    \
    do {                                                           \
        if (one_part && dy16 == 3) FGCN_SWT4(CT_, NT_, NS_, 1, 3); \
        else if (one_part && dy16) FGCN_SWT4(CT_, NT_, NS_, 1, 1); \
        else if (one_part) FGCN_SWT4(CT_, NT_, NS_, 1, 0);         \
        else FGCN_SWT4(CT_, NT_, NS_, 3, 0);                       \
    } while (0)
    // frame slots of a wave per tile: F frames over 8 / CT waves per channel tile, rounded up to even
    const int nslot = ((g.F + 8 / g.CT - 1) / (8 / g.CT) + 1) & ~1;
#define FGCN_SWT_NT(CT_, NS_)                  \
    do {                                       \
        if (g.NT == 8) FGCN_SWT(CT_, 8, NS_);  \
        else FGCN_SWT(CT_, 4, NS_);            \
    } while (0)
    // (5 .. 8 frames per tile for 16 .. 32 joints: 6 or 8 slots with one wave per channel tile, 4 with two)
    FGCN_REQUIRE(nslot == (g.CT == 8 ? (g.F > 6 ? 8 : 6) : 4), FGCN_E_BADARG, "spatial_wgrad_tile: %d frames per tile: no such kernel form", g.F);
    if (g.CT == 8) {
        if (nslot == 8) FGCN_SWT_NT(8, 8);
        else FGCN_SWT_NT(8, 6);
    } else {
        FGCN_SWT_NT(4, 4);
    }
#undef FGCN_SWT_NT
#undef FGCN_SWT
#undef FGCN_SWT4
    return launch_status("spatial_wgrad_tile");
}
