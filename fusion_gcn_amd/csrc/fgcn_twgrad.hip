// Weight gradient of the (taps x 1) temporal convolution, all taps in one pass:
//     dW[j][k][n] = sum_{sample, t, v} a[(t + shift_j, v), k] * g[(t, v), n]          (agcn.py:41-51, backward of conv)
// The per-tap kernel (rows_wgrad) re-reads a and g once per tap and tile: 16 FLOP per byte fetched from L2, one
// dependent MFMA chain per wave, two barriers per 32 MFMAs.  Here a workgroup owns a (32 in-channel x TN out-channel)
// tile of ALL taps:
//   * rows are walked in stages of 64 (TN = 128) or 128 (TN = 64) consecutive rows of ONE sample; the stage's rows of g
//     and the window of a they touch over all taps (stage + (taps-1) V rows x 32 channels) are staged once in LDS (LDS-DMA); rows of the window
//     outside the sample are zeros, so no per-(row, tap) masks are needed (a stage never straddles two samples);
//   * every MFMA k-step (2 rows) reads ONE g fragment and one a fragment per tap: taps independent accumulators
//     (9 x 16 registers), ~1.1 LDS dwords per MFMA, 288 MFMAs between barriers at 9 taps;
//   * strided convolutions are two calls over the even / odd frames of a (`a_s`, `a_o`: frame view), each tap lands
//     in its own slab (`tap0`, `tap_step`), so no structurally-zero taps are computed.
// Output: deterministic partial slabs [nslab][taps_total][K][N], summed by reduce_sum.
#include "fgcn_common.hpp"

namespace fgcn {


struct TWgradP {
    const float* a;
    const float* g;
    float* partial;
    int B, T_g, V, K, N, ld_a, ld_g;
    int T_a_full, a_s, a_o, Th_a;
    int shift0, tap0, tap_step, taps_total;
    int stages_per_sample, total_stages, stages_per_split;
    int tiles_n, win_rows, stage_rows;
    int chunk_mode;   // 0: accumulator j = tap j (window rows shifted by j*V); 1: accumulator j = in-channel chunk j (1x1 conv)
    int ring_rows;    // tap mode of the split-bf16 kernel, RING form: rows of the circular window image (256 or 512)
    unsigned a_bytes, g_bytes, p_bytes;
    // FGCN_PRODUCTS_F16X2 (tconv_wgrad_x3_kernel<..., NP = 2>): float bits of max |a| and max |g| over the whole tensors, gathered by
    // the kernels that staged them before (fgcn_tconv_halo / fgcn_pw_gemm `in_amax`); both operands are scaled by the exact power of
    // two that puts that maximum into [2^14, 2^15) as they are split, the slabs leave with the scales multiplied back out
    const unsigned* a_amax;
    const unsigned* g_amax;
    // tconv_wgrad_x3_kernel, per_xcd > 0: 1-D grid in XCD-aware order.  Workgroup ids go round-robin over the 8 XCDs (each with its
    // own L2); with the plain (tile, row split) grid the tiles_xy (channel group, column tile) workgroups of one row split -- which
    // read the SAME rows of a and g -- landed on different XCDs whenever tiles_xy divides 8 (conv_d at 256 channels: 8 tiles, one
    // per XCD) and every one fetched its rows through the fabric: 1.84 GB per launch against 0.98 GB algorithmic
    // (profiles/r03_bf16x3_step_traffic_by_kernel.txt).  Id b takes virtual id (b % 8) * per_xcd + b / 8, tile fastest: the tiles of
    // a row split run side by side on one XCD and share its L2.
    int per_xcd, tiles_xy, n_split;
    int in16;                           // a and g are BFLOAT16 tensors (tconv_wgrad_x3_kernel<.., NP = 1, .., IN16 = true>; ld_a / ld_g in elements)
};

// TN = out-channel tile (64: waves = 2 column tiles x 2 row halves of the stage, 128: 4 column tiles).
// Staging is LDS-DMA (`buffer_load_dwordx4 ... lds`: one wave instruction drops 1 KiB = 8 window rows x 32 channels, or
// 1024/TN g rows, straight into LDS; rows / channels that do not exist are out-of-range buffer reads = zeros): no
// staging registers, which is what lets 9 x 16 accumulator registers and two workgroups per CU coexist.
// MM: FGCN_MATH_BF16 -- 8 rows per bf16 MFMA (lane half h contracts rows 8g + 4h + (0..3)), operands rounded as read;
//     FGCN_MATH_BF16X3 -- 16 rows per step on v_mfma_f32_32x32x16_bf16 (lane half h contracts rows 16g + 8h + (0..7)), both
//     fragments split into three bf16 parts as they are read (the contraction runs along the image rows, so the split
//     cannot be done once per row at staging time without a transposed plane layout): 44 vector instructions per
//     fragment, ten fragments per 54 MFMAs -- about even with the matrix pipe, still ~2x the f32 rate
template <int NTAP, int TN, int MM>
__global__ __launch_bounds__(256, 2) void tconv_wgrad_kernel(TWgradP p) {
    constexpr int TW_BR = 8192 / TN;                  // rows per stage: 64 (TN 128) or 128 (TN 64), 32 KiB of g
    constexpr int NSUB = TN / 32, NPART = 4 / NSUB;   // column tiles, row parts of a stage
    constexpr int STEPS = TW_BR / 2 / NPART;          // MFMA k-steps per wave and stage
    constexpr int GROWS = 256 / TN;                   // g rows per LDS-DMA piece (64 lanes x 4 floats)
    constexpr int GPIECES = TW_BR / GROWS;
    constexpr unsigned OOB = 0x80000000u;
    using lds_ptr = __attribute__((address_space(3))) void*;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ppp = (p.win_rows + 7) >> 3;            // LDS-DMA pieces (8 rows x 32 channels) per plane
    const int apieces = p.chunk_mode ? ppp * NTAP : ppp;
    float* Aw = smem;                                 // tap mode: [ppp * 8][32]; chunk mode: [NTAP][ppp * 8][32]
    float* Gs = smem + apieces * 256;                 // [TW_BR][TN]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int tk = blockIdx.x / p.tiles_n, tn = blockIdx.x - tk * p.tiles_n;
    const int k0 = tk * (p.chunk_mode ? 32 * NTAP : 32), n0 = tn * TN;
    const int nsub = wave % NSUB, part = wave / NSUB;
    const int V = p.V, TVg = p.T_g * V;
    const int sbeg = blockIdx.y * p.stages_per_split;
    const int send = min(sbeg + p.stages_per_split, p.total_stages);

    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, p.g_bytes, 0x00020000);

    // per-lane part of the source addresses of a piece: a: row lane/8, channels (lane%8)*4;  g: row lane/(TN/4)
    const int a_lr = lane >> 3, akc = k0 + (lane & 7) * 4;
    const int g_lr = lane / (TN / 4), gnc = n0 + (lane % (TN / 4)) * 4;
    const bool g_cok = gnc < p.N;
    const bool strided = p.a_s != 1 || p.a_o != 0;

    f32x16 acc[NTAP];
#pragma unroll
    for (int j = 0; j < NTAP; ++j) acc[j] = zero16();

    // fragment bases: window row of g-row r and tap j is r + j*V; this wave's rows start at part * 2 * STEPS
    const float* abase = Aw + (part * STEPS * 2 + h) * 32 + l31;
    const float* gbase = Gs + (part * STEPS * 2 + h) * TN + nsub * 32 + l31;
    const int tapstride = p.chunk_mode ? ppp * 256 : V * 32;

    for (int sid = sbeg; sid < send; ++sid) {
        const int n = sid / p.stages_per_sample;
        const int r0 = (sid - n * p.stages_per_sample) * TW_BR;      // first g row of the stage inside the sample
        __syncthreads();                                             // previous stage's fragment reads are done
        for (int pc = wave; pc < apieces; pc += 4) {
            const int plane = p.chunk_mode ? pc / ppp : 0;           // wave-uniform
            const int wr = (pc - plane * ppp) * 8 + a_lr;
            const int q = r0 + p.shift0 * V + wr;                    // row of the frame view inside the sample
            const int chan = akc + plane * 32;
            const bool ok = chan < p.K && wr < p.win_rows && q >= 0 && q < p.Th_a * V;
            int row = ok ? q : 0;
            if (strided) {                                           // wave-uniform: even / odd frames of a
                const int f = (int)((unsigned)row / (unsigned)V);
                row = (f * p.a_s + p.a_o) * V + (row - f * V);
            }
            const unsigned off = ok ? (unsigned)((n * p.T_a_full * V + row) * p.ld_a + chan) * 4u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(Aw + pc * 256), 16, off, 0, 0, 0);
        }
        for (int pc = wave; pc < GPIECES; pc += 4) {
            const int r = r0 + pc * GROWS + g_lr;
            const unsigned off = (g_cok && r < TVg) ? (unsigned)((n * TVg + r) * p.ld_g + gnc) * 4u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_ptr)(Gs + pc * 256), 16, off, 0, 0, 0);
        }
        __syncthreads();                                             // drains the DMA (vmcnt(0)) and publishes the stage
        // ---- NTAP independent MFMA chains over this wave's rows of the stage ----------------------------------------
        if constexpr (MM == 2) {
            const float* ab = abase + 7 * h * 32;          // rows 8h + e instead of h
            const float* gb = gbase + 7 * h * TN;
#pragma unroll 1
            for (int g16 = 0; g16 < STEPS / 8; ++g16) {
                const float* g = gb + 16 * g16 * TN;
                u32x4v gq[3];
                split3_x8(g[0], g[TN], g[2 * TN], g[3 * TN], g[4 * TN], g[5 * TN], g[6 * TN], g[7 * TN], gq);
#pragma unroll
                for (int j = 0; j < NTAP; ++j) {
                    const float* a = ab + 16 * g16 * 32 + j * tapstride;
                    u32x4v aq[3];
                    split3_x8(a[0], a[32], a[64], a[96], a[128], a[160], a[192], a[224], aq);
                    acc[j] = mfma_x3_k16(aq, gq, acc[j]);
                }
            }
        } else if constexpr (MM == 1) {
            const float* ab = abase + 3 * h * 32;          // rows 4h + e instead of h
            const float* gb = gbase + 3 * h * TN;
#pragma unroll 2
            for (int g8 = 0; g8 < STEPS / 4; ++g8) {
                const float* g = gb + 8 * g8 * TN;
                const s16x4 gp = pack_bf16(g[0], g[TN], g[2 * TN], g[3 * TN]);
#pragma unroll
                for (int j = 0; j < NTAP; ++j) {
                    const float* a = ab + 8 * g8 * 32 + j * tapstride;
                    acc[j] = mfma_bf16(pack_bf16(a[0], a[32], a[64], a[96]), gp, acc[j]);
                }
            }
        } else {
#pragma unroll 4
            for (int s = 0; s < STEPS; ++s) {
                const float gv = gbase[2 * s * TN];
#pragma unroll
                for (int j = 0; j < NTAP; ++j) acc[j] = mfma32(abase[2 * s * 32 + j * tapstride], gv, acc[j]);
            }
        }
    }

    // ---- partial slabs: [slab = split * NPART + part][tap][k][n] ------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)p.partial, 0, p.p_bytes, 0x00020000);
    const int slab = blockIdx.y * NPART + part;
    const int ncol = n0 + nsub * 32 + l31;
#pragma unroll
    for (int j = 0; j < NTAP; ++j) {
        const int tap = p.chunk_mode ? 0 : p.tap0 + j * p.tap_step;
        const int kj = k0 + (p.chunk_mode ? 32 * j : 0);
        const unsigned base = (unsigned)((slab * p.taps_total + tap) * p.K) * (unsigned)p.N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = kj + acc_row(r, lane);
            const unsigned off = (k < p.K && ncol < p.N) ? (base + (unsigned)(k * p.N + ncol)) * 4u : OOB;
            const float val = acc[j][r];   // (bit_cast straight from a vector element stores element 0: go through a scalar)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rp, off, 0, 0);
        }
    }
}

// ---- FGCN_MATH_BF16X3, all taps, >= 128 output channels -------------------------------------------------------------------
// The contraction of a weight gradient runs along the image ROWS, so the split fragments of v_mfma_f32_32x32x16_bf16 (8
// consecutive rows of one channel per lane) cannot be read from a row-major image with plain LDS reads.  Splitting the
// fragments in registers as they are read (kernel above, MM = 2) costs 44 vector instructions per fragment, ten fragments
// per 54 MFMAs: the vector unit, not the matrix pipe, sets the pace (1.2x over f32).  Here the stage is split ONCE, as it
// is written to LDS (three bf16 planes, row-major), and the fragments come from `ds_read_b64_tr_b16`, gfx950's transposing
// LDS read (a 16-lane group reads 4 rows x 16 channels and every lane receives 4 rows of its channel): no vector work in
// the MFMA loop.
//   * workgroup = 512 threads (8 waves, one workgroup per CU): wave = (column tile nsub 0..3, row half `part`); a stage is
//     64 rows of g x 128 columns and the a window over all taps (64 + (taps-1) V rows x 32 channels);
//   * staging goes through registers: the NEXT stage's rows are requested before the MFMAs of the current one, split and
//     written between two barriers afterwards (one LDS image, the global latency stays behind the matrix work);
//   * planes: a [row][32] bf16 (64-byte rows: the 4 x 64 bytes of a transposed read cover all banks once),
//     g [row][128] bf16 with 320-byte rows (4 consecutive rows land 64 bytes apart modulo 256: conflict-free).
constexpr int X3_SA = 64;                // bytes per row of an a plane
constexpr int X3_APASS = 6;              // passes of 64 rows: window <= 128 + 8 * 32 = 384 rows (V <= 32, 9 taps)
// TN = 128: waves = 4 column tiles x 2 row halves of a 64-row stage; TN = 64: 2 column tiles x 4 row quarters of a 128-row
// stage (a wave always owns 32 rows = two 16-row steps).  g rows are padded by 64 bytes: 4 consecutive rows then land 64
// bytes apart modulo 256 = one transposed read touches every bank once.
constexpr int x3_rows(int tn, int wv = 8) { return 32 * (wv / (tn / 32)); }   // 8 waves: 64 (TN 128) / 128 (TN 64); 4 waves: half
constexpr int x3_sg(int tn) { return tn == 64 ? 160 : tn * 2 + 64; }   // (64 columns: 32 bytes of padding -- the circular window fits beside 128 g rows)

__device__ __forceinline__ u32x2 lds_read_tr16(const unsigned char* p) {
    using v4s = __attribute__((ext_vector_type(4))) short;
    const v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
    return __builtin_bit_cast(u32x2, v);
}

// CH (1x1 convolutions, fgcn_pw_wgrad): accumulator j = in-channel chunk j instead of tap j -- the "window" is then NTAP
// images of the stage's rows, one per 32-channel chunk (window row chunk * X3_R + r), and the fragment of accumulator j
// starts X3_R rows further instead of V rows further; everything else is the same kernel.
// NP = bf16 parts per operand: 3 (FGCN_MATH_BF16X3) or 1 (FGCN_MATH_BF16: operands rounded once as the stage is written, one
// MFMA per product group); 2 = two f16 parts, three products (FGCN_PRODUCTS_F16X2; tensor-level scales, see TWgradP).
// WV = waves per workgroup: 8 (one workgroup per CU, the next stage's rows prefetched into registers across the MFMAs) or 4 (half
// the rows per stage, TWO workgroups per CU and no prefetch: one workgroup stages while the other multiplies, as in the halo conv).
// RING (tap mode): consecutive stages of a sample need windows that overlap in all but X3_R rows -- [r0 + sh, r0 + sh + X3_R +
// (NTAP-1) V) slides by X3_R.  Instead of fetching, splitting and writing the whole window per stage (264 rows for 64 new ones at
// 9 taps, V = 25) the a planes are a circular image: row q of the sample's frame view lives at slot q & (ring_rows - 1), a stage
// adds its X3_R new rows (prefetched across the previous stage's MFMAs), and only the first stage of a sample (or of the
// workgroup's share) fills the (NTAP-1) V older rows, synchronously.  Fragment addresses wrap per read.
// IN16 (NP = 1, tap mode): a and g are BFLOAT16 tensors (half-precision storage written by fgcn_bn_act_h / fgcn_bn_act_bwd_apply_h): the
// stage is copied, 8 bytes per four values, instead of fetched as f32 and rounded here -- the same staged bytes, half the reads.
template <int NTAP, int TN, bool CH, int NP, int WV = 8, bool RING = false, bool IN16 = false>
__global__ __launch_bounds__(64 * WV, WV == 8 ? 1 : 2) void tconv_wgrad_x3_kernel(TWgradP p) {
    static_assert(!(RING && CH), "the circular window is the tap mode's");
    static_assert(!IN16 || NP == 1, "bfloat16 inputs: the one-part kernel (tap mode, and the chunk mode of the 1x1 weight gradients)");
    constexpr unsigned ES = IN16 ? 2u : 4u;                       // bytes per stored element
    // four values of a row: 16 bytes of f32, or 8 bytes of bfloat16 parked in the first two components
    auto ld4 = [](__amdgpu_buffer_rsrc_t r, unsigned off) -> f32x4 {
        if constexpr (IN16) {
            const u32x2 h = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
            const unsigned b0 = h[0], b1 = h[1];      // (element -> scalar first: bit casts of ext-vector ELEMENTS read element 0 with hipcc 7.2)
            return f32x4{__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1), 0.f, 0.f};
        } else {
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
        }
    };
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NT = 64 * WV;
    constexpr int X3_R = x3_rows(TN, WV), X3_SG = x3_sg(TN);
    constexpr int NSUBS = TN / 32, NPARTS = WV / NSUBS;
    constexpr int GT = TN / 4, GRP = NT / GT;                     // g staging: threads per row, rows per pass (4 passes)
    constexpr int RA = NT / 8;                                    // a staging: 8 threads per row of 32 channels, rows per pass
    constexpr int APASS = RING ? X3_R / RA : (WV == 8 ? X3_APASS : (X3_R + 8 * 32 + RA - 1) / RA);
    constexpr bool PF = WV == 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nsub = wave % NSUBS, part = wave / NSUBS;
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.per_xcd > 0) {
        const int vid = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
        if (vid >= p.tiles_xy * p.n_split) return;
        by = vid / p.tiles_xy;
        bx = vid - by * p.tiles_xy;
    }
    const int tk = bx / p.tiles_n, tn = bx - tk * p.tiles_n;
    const int k0 = tk * (CH ? 32 * NTAP : 32), n0 = tn * TN;
    const int V = p.V, TVg = p.T_g * V;
    const int win = CH ? NTAP * X3_R : p.win_rows;                // tap mode: X3_R + (NTAP - 1) * V
    const unsigned a_plane = (unsigned)(RING ? p.ring_rows : win) * X3_SA, g_plane = X3_R * X3_SG;
    const int rmask = p.ring_rows - 1;
    unsigned char* Ap = lds_raw;                                  // [3][win][32] bf16
    unsigned char* Gp = lds_raw + NP * a_plane;                   // [NP][X3_R][128 (+32 pad)] bf16
    const int sbeg = by * p.stages_per_split;
    const int send = min(sbeg + p.stages_per_split, p.total_stages);

    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, p.g_bytes, 0x00020000);
    const bool strided = p.a_s != 1 || p.a_o != 0;
    int ea = 0, eg = 0;
    if constexpr (NP == 2) {
        ea = min(scale_exp_for(*p.a_amax), 126);                // (2^-ea must be a normal float too)
        eg = min(scale_exp_for(*p.g_amax), 126);
    }
    const float sc_a = exp2i(ea), sc_g = exp2i(eg);

    // staging roles: a: row tid/8 + RA*i, channels k0 + (tid%8)*4;  g: row tid/GT + GRP*i, columns n0 + (tid%GT)*4
    const int a_row = tid >> 3, a_c4 = tid & 7, g_row = tid / GT, g_c4 = tid % GT;
    const bool a_cok = k0 + a_c4 * 4 < p.K, g_cok = n0 + g_c4 * 4 < p.N;
    f32x4 sa[APASS], sg[4];
    auto fetch = [&](int sid) {
        const int n = sid / p.stages_per_sample;
        const int r0 = (sid - n * p.stages_per_sample) * X3_R;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int wr = (RING ? win - X3_R : 0) + a_row + RA * i;    // RING: only the X3_R newest rows of the window
            const int chunk = CH ? (RA * i) / X3_R : 0;           // compile-time per pass
            const int q = CH ? r0 + wr - chunk * X3_R : r0 + p.shift0 * V + wr;   // row of the frame view inside the sample
            const bool ok = (CH ? k0 + chunk * 32 + a_c4 * 4 < p.K : a_cok) && wr < win && q >= 0 && q < p.Th_a * V;
            int row = ok ? q : 0;
            if (strided) {
                const int f = (int)((unsigned)row / (unsigned)V);
                row = (f * p.a_s + p.a_o) * V + (row - f * V);
            }
            const unsigned off = ok ? (unsigned)((n * p.T_a_full * V + row) * p.ld_a + k0 + chunk * 32 + a_c4 * 4) * ES : OOB;
            if (RING || i * RA < win) sa[i] = ld4(ra, off);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + g_row + GRP * i;
            const unsigned off = (g_cok && r < TVg) ? (unsigned)((n * TVg + r) * p.ld_g + n0 + g_c4 * 4) * ES : OOB;
            sg[i] = ld4(rg, off);
        }
    };
    auto put_a = [&](int slot, f32x4 v) {
        u32x2 ph, pm, pl;
        unsigned char* d = Ap + slot * X3_SA + a_c4 * 8;
        if constexpr (NP == 2) {
            split2h_x4(v * sc_a, ph, pm);
            *reinterpret_cast<u32x2*>(d) = ph;
            *reinterpret_cast<u32x2*>(d + a_plane) = pm;
            return;
        }
        if constexpr (IN16) {                                     // already bfloat16: a copy
            const float e0 = v[0], e1 = v[1];
            *reinterpret_cast<u32x2*>(d) = u32x2{__builtin_bit_cast(unsigned, e0), __builtin_bit_cast(unsigned, e1)};
            return;
        }
        split3_x4(v, ph, pm, pl);
        *reinterpret_cast<u32x2*>(d) = ph;
        if constexpr (NP == 3) {
            *reinterpret_cast<u32x2*>(d + a_plane) = pm;
            *reinterpret_cast<u32x2*>(d + 2 * a_plane) = pl;
        }
    };
    // RING: the (NTAP - 1) V rows in front of the stage's new ones -- once per sample / per workgroup share, not prefetched
    auto fill = [&](int sid) {
        const int n = sid / p.stages_per_sample;
        const int r0 = (sid - n * p.stages_per_sample) * X3_R;
        const int cnt = win - X3_R;
        for (int i = 0; i * RA < cnt; ++i) {
            const int wr = a_row + RA * i;
            const int q = r0 + p.shift0 * V + wr;
            const bool ok = a_cok && wr < cnt && q >= 0 && q < p.Th_a * V;
            int row = ok ? q : 0;
            if (strided) {
                const int f = (int)((unsigned)row / (unsigned)V);
                row = (f * p.a_s + p.a_o) * V + (row - f * V);
            }
            const unsigned off = ok ? (unsigned)((n * p.T_a_full * V + row) * p.ld_a + k0 + a_c4 * 4) * ES : OOB;
            const f32x4 v = ld4(ra, off);
            if (wr < cnt) put_a(q & rmask, v);
        }
    };
    auto deposit = [&](int sid) {
        if constexpr (RING) {
            const int n = sid / p.stages_per_sample;
            const int q0 = (sid - n * p.stages_per_sample) * X3_R + p.shift0 * V + win - X3_R;
#pragma unroll
            for (int i = 0; i < APASS; ++i) put_a((q0 + a_row + RA * i) & rmask, sa[i]);
        }
#pragma unroll
        for (int i = 0; i < (RING ? 0 : APASS); ++i) {
            const int wr = a_row + RA * i;
            if (i * RA < win && wr < win) put_a(wr, sa[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u32x2 ph, pm, pl;
            unsigned char* d = Gp + (g_row + GRP * i) * X3_SG + g_c4 * 8;
            if constexpr (NP == 2) {
                split2h_x4(sg[i] * sc_g, ph, pm);
                *reinterpret_cast<u32x2*>(d) = ph;
                *reinterpret_cast<u32x2*>(d + g_plane) = pm;
                continue;
            }
            if constexpr (IN16) {
                const float e0 = sg[i][0], e1 = sg[i][1];
                *reinterpret_cast<u32x2*>(d) = u32x2{__builtin_bit_cast(unsigned, e0), __builtin_bit_cast(unsigned, e1)};
                continue;
            }
            split3_x4(sg[i], ph, pm, pl);
            *reinterpret_cast<u32x2*>(d) = ph;
            if constexpr (NP == 3) {
                *reinterpret_cast<u32x2*>(d + g_plane) = pm;
                *reinterpret_cast<u32x2*>(d + 2 * g_plane) = pl;
            }
        }
    };

    // 16x16x32 MFMAs: the wave's (32 in-channel x 32 column) region is 2 x 2 tiles of 16 x 16 per accumulator set, and its 32
    // rows of the stage are ONE contraction step
    f32x4 acc[NTAP][2][2];
#pragma unroll
    for (int j = 0; j < NTAP; ++j)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[j][kt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed-read addresses: the 16-lane group g4 = lane >> 4 reads rows 8 g4 + q (+4 for the second half of the fragment)
    // x 16 channels (lane (q = (lane & 15) >> 2, c = lane & 3) supplies row q, channels 4c .. 4c+3) and every lane receives the 8
    // rows 8 g4 .. 8 g4 + 7 of channel lane & 15 of the tile: the k = 8g + j operand order of v_mfma_f32_16x16x32_bf16
    const int g4 = lane >> 4, q4 = (lane & 15) >> 2, c4 = lane & 3, l15 = lane & 15;
    const unsigned char* a_lane = Ap + (part * 32 + 8 * g4 + q4) * X3_SA + (4 * c4) * 2;                 // + 32 bytes per tile
    const unsigned char* g_lane = Gp + (part * 32 + 8 * g4 + q4) * X3_SG + (nsub * 32 + 4 * c4) * 2;
    auto frag = [&](const unsigned char* base, unsigned plane, int row_stride, u32x4v (&f)[NP]) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            const u32x2 lo = lds_read_tr16(base + pl * plane);
            const u32x2 hi = lds_read_tr16(base + pl * plane + 4 * row_stride);
            f[pl] = u32x4v{lo[0], lo[1], hi[0], hi[1]};
        }
    };

    if (PF && sbeg < send) fetch(sbeg);
    for (int sid = sbeg; sid < send; ++sid) {
        __syncthreads();                                          // the previous stage's fragment reads are done
        if constexpr (!PF) fetch(sid);                            // (the CU's other workgroup multiplies meanwhile)
        if constexpr (RING) {
            if (sid == sbeg || sid % p.stages_per_sample == 0) fill(sid);
        }
        deposit(sid);
        __syncthreads();
        if (PF && sid + 1 < send) fetch(sid + 1);                 // lands during the MFMAs below
        u32x4v gq[2][NP];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) frag(g_lane + nt * 32, g_plane, X3_SG, gq[nt]);
        int lane_q = 0;                                           // RING: this lane's first window row, before the tap shift
        if constexpr (RING) {
            const int n = sid / p.stages_per_sample;
            lane_q = (sid - n * p.stages_per_sample) * X3_R + p.shift0 * V + part * 32 + 8 * g4 + q4;
        }
#pragma unroll
        for (int j = 0; j < NTAP; ++j) {
            u32x4v aq[2][NP];
            if constexpr (RING) {
                const unsigned char* lo = Ap + ((lane_q + j * V) & rmask) * X3_SA + (4 * c4) * 2;
                const unsigned char* hi = Ap + ((lane_q + j * V + 4) & rmask) * X3_SA + (4 * c4) * 2;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) {
                        const u32x2 l = lds_read_tr16(lo + kt * 32 + pl * a_plane);
                        const u32x2 h = lds_read_tr16(hi + kt * 32 + pl * a_plane);
                        aq[kt][pl] = u32x4v{l[0], l[1], h[0], h[1]};
                    }
            }
#pragma unroll
            for (int kt = 0; kt < (RING ? 0 : 2); ++kt) frag(a_lane + (j * (CH ? X3_R : V)) * X3_SA + kt * 32, a_plane, X3_SA, aq[kt]);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if constexpr (NP == 3) acc[j][kt][nt] = mfma_x3_k32(aq[kt], gq[nt], acc[j][kt][nt]);
                    else if constexpr (NP == 2) acc[j][kt][nt] = mfma_h2_k32(aq[kt], gq[nt], acc[j][kt][nt]);
                    else acc[j][kt][nt] = mfma_bf16_k32(aq[kt][0], gq[nt][0], acc[j][kt][nt]);
                }
        }
    }

    // ---- partial slabs: [slab = split * NPARTS + part][tap][k][n] -----------------------------------------------------
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)p.partial, 0, p.p_bytes, 0x00020000);
    const int slab = by * NPARTS + part;
    const float un_a = exp2i(-ea), un_g = exp2i(-eg);               // (NP == 2; 1 otherwise)
#pragma unroll
    for (int j = 0; j < NTAP; ++j) {
        const int tap = CH ? 0 : p.tap0 + j * p.tap_step;
        const int kj = k0 + (CH ? 32 * j : 0);
        const unsigned base = (unsigned)((slab * p.taps_total + tap) * p.K) * (unsigned)p.N;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int ncol = n0 + nsub * 32 + 16 * nt + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = kj + 16 * kt + 4 * g4 + r;
                    const unsigned off = (k < p.K && ncol < p.N) ? (base + (unsigned)(k * p.N + ncol)) * 4u : OOB;
                    const float val = NP == 2 ? acc[j][kt][nt][r] * un_a * un_g : acc[j][kt][nt][r];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rp, off, 0, 0);
                }
            }
    }
}

}  // namespace fgcn

using namespace fgcn;

// row parts of a stage = partial slabs per row split.  The split-bf16 all-taps kernel (math mode bf16x3, > 64 output
// channels) always works in two row halves.
// (tuning key 6 bit 0: 1x1 weight gradients of that mode on the 256-thread kernel that splits fragments as it reads them --
// the A/B switch of tools/kbench.py)
static bool twgrad_use_x3(int N, int chunk_mode) {
    return fgcn::math_mode() != FGCN_MATH_F32 && !(chunk_mode && (fgcn::tuning(6) & 1));   // both bf16 modes (3 parts / 1 part)
}
// The split-bf16 kernel as one 8-wave workgroup per CU (next stage prefetched) or two 4-wave ones (half the rows per stage, no
// prefetch), and in the tap mode with the circular window image (RING).  Measured at B = 128 (tools/kbench.py wgrad, same box):
//   1x1 weight gradients: two small workgroups -3 .. -9 % (K = 64: +3 %);
//   all taps, 128 / 256 columns: circular window -10 .. -14 % on 8 waves, another 0-2 % on 4 (without the circular window the
//     4-wave form re-stages the 8 V-row tap window twice as often: +1-2 %);
//   all taps, 64 columns (128-row stages): the circular window does not fit beside the g planes on 8 waves and loses 8 % on 4.
// -> 1x1: 4 waves; all taps: 4 waves + circular window above 64 columns, 8 waves without it at 64.
// (tuning key 6: bit 5 = 1x1 on 8 waves, bit 6 = all taps above 64 columns on 8 waves, bit 8 = all taps at 64 columns on 4 waves,
// bit 7 = no circular window.  In the step the wave counts of the tap kernels are within the run-to-run noise of each other.)
static int twgrad_x3_waves(int N, int chunk_mode) {
    if (chunk_mode) return (fgcn::tuning(6) & 32) ? 8 : 4;
    // (FGCN_MATH_BF16: with a sixth of the matrix work per staged row the tap kernels above 64 columns are faster on 8 waves -- 26.67 -> 26.50 ms per
    // step, profiles/r06_ab_bf16_half_activations.txt; bit 6 flips the choice in either mode)
    if (N > 64) return (((fgcn::tuning(6) & 64) != 0) != (fgcn::math_mode() == FGCN_MATH_BF16)) ? 8 : 4;
    return (fgcn::tuning(6) & 256) ? 4 : 8;
}
static int twgrad_parts(int N, int chunk_mode) {
    if (twgrad_use_x3(N, chunk_mode)) return twgrad_x3_waves(N, chunk_mode) / (N <= 64 ? 2 : 4);
    return N <= 64 ? 2 : 1;
}

extern "C" int fgcn_tconv_wgrad_slabs(int N, int nsplit) { return nsplit * twgrad_parts(N, 0); }
extern "C" int fgcn_pw_wgrad_slabs(int N, int nsplit) { return nsplit * twgrad_parts(N, 1); }
/* workgroups of one launch that are resident at once (the row-split count is chosen so that tiles * nsplit fits) */
extern "C" int fgcn_tconv_wgrad_resident(int N) { return twgrad_use_x3(N, 0) && twgrad_x3_waves(N, 0) == 8 ? 256 : 512; }
extern "C" int fgcn_pw_wgrad_resident(int N) { return twgrad_use_x3(N, 1) && twgrad_x3_waves(N, 1) == 8 ? 256 : 512; }

template <int NTAP>
static void launch_twgrad(const TWgradP& p, int N, dim3 grid, size_t lds, hipStream_t s) {
    static bool opt_in = false;   // once per instantiation; not a stream operation (stays out of graph captures)
    if (!opt_in) {
        const int max_lds = 160 * 1024;
#define FGCN_TW_ATTR(TN_, BF_)                                                                      \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_wgrad_kernel<NTAP, TN_, BF_>),  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)
        FGCN_TW_ATTR(64, 0); FGCN_TW_ATTR(64, 1); FGCN_TW_ATTR(64, 2); FGCN_TW_ATTR(128, 0); FGCN_TW_ATTR(128, 1);
        FGCN_TW_ATTR(128, 2);
#undef FGCN_TW_ATTR
        opt_in = true;
    }
    const int mm = fgcn::math_mode();
#define FGCN_TW_LAUNCH(TN_)                                                                                   \
    do {                                                                                                      \
        if (mm == FGCN_MATH_BF16X3) hipLaunchKernelGGL((tconv_wgrad_kernel<NTAP, TN_, 2>), grid, dim3(256), lds, s, p); \
        else if (mm == FGCN_MATH_BF16) hipLaunchKernelGGL((tconv_wgrad_kernel<NTAP, TN_, 1>), grid, dim3(256), lds, s, p); \
        else hipLaunchKernelGGL((tconv_wgrad_kernel<NTAP, TN_, 0>), grid, dim3(256), lds, s, p);              \
    } while (0)
    if (N <= 64) FGCN_TW_LAUNCH(64);
    else FGCN_TW_LAUNCH(128);
#undef FGCN_TW_LAUNCH
}

// one instantiation of the split kernel; the bfloat16-input form exists for the one-part kernel in tap mode
template <int NTAP, int TN, bool CH, int NP, int WV, bool RING>
static void twx_go(const TWgradP& p, dim3 grid, size_t lds, hipStream_t s) {
    if constexpr (NP == 1) {
        if (p.in16) {
            static bool opted16 = false;
            if (!opted16) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_wgrad_x3_kernel<NTAP, TN, CH, NP, WV, RING, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                opted16 = true;
            }
            hipLaunchKernelGGL((tconv_wgrad_x3_kernel<NTAP, TN, CH, NP, WV, RING, true>), grid, dim3(64 * WV), lds, s, p);
            return;
        }
    }
    hipLaunchKernelGGL((tconv_wgrad_x3_kernel<NTAP, TN, CH, NP, WV, RING>), grid, dim3(64 * WV), lds, s, p);
}

template <int NTAP, bool CH>
static void launch_twgrad_x3(const TWgradP& p, int N, dim3 grid, size_t lds, hipStream_t s) {
    static_assert(!CH || NTAP <= 6, "chunk mode: at most 6 (128 columns) / 3 (64 columns) chunks fit the staging passes");
    constexpr bool RG = !CH;                          // the circular-window form exists for the tap mode only
    static bool opt_in = false;
    if (!opt_in) {
#define FGCN_TWX_ATTR1(TN_, NP_, WV_, RING_)                                                                          \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_wgrad_x3_kernel<NTAP, TN_, CH, NP_, WV_, RING_>), \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
#define FGCN_TWX_ATTR(TN_)                                                                                            \
    FGCN_TWX_ATTR1(TN_, 3, 8, false); FGCN_TWX_ATTR1(TN_, 1, 8, false); FGCN_TWX_ATTR1(TN_, 3, 4, false);            \
    FGCN_TWX_ATTR1(TN_, 1, 4, false); FGCN_TWX_ATTR1(TN_, 3, 8, RG); FGCN_TWX_ATTR1(TN_, 1, 8, RG);                  \
    FGCN_TWX_ATTR1(TN_, 3, 4, RG); FGCN_TWX_ATTR1(TN_, 1, 4, RG);                                                    \
    FGCN_TWX_ATTR1(TN_, 2, 8, false); FGCN_TWX_ATTR1(TN_, 2, 4, false); FGCN_TWX_ATTR1(TN_, 2, 8, RG); FGCN_TWX_ATTR1(TN_, 2, 4, RG)
        FGCN_TWX_ATTR(128);
        if constexpr (!CH || NTAP <= 3) { FGCN_TWX_ATTR(64); }
#undef FGCN_TWX_ATTR
#undef FGCN_TWX_ATTR1
        opt_in = true;
    }
    const bool one = fgcn::math_mode() == FGCN_MATH_BF16;
    const bool two = !one && p.a_amax && p.g_amax;                // f16x2 products (twgrad_launch cleared the pointers otherwise)
    const bool half = twgrad_x3_waves(N, CH ? 1 : 0) == 4;
    const bool ring = RG && p.ring_rows > 0;
#define FGCN_TWX_GO(TN_, NP_, WV_, RING_) twx_go<NTAP, TN_, CH, NP_, WV_, RING_>(p, grid, lds, s)
#define FGCN_TWX_LAUNCH(TN_)                                                                                    \
    do {                                                                                                        \
        if (ring) {                                                                                             \
            if (half) { if (one) FGCN_TWX_GO(TN_, 1, 4, RG); else if (two) FGCN_TWX_GO(TN_, 2, 4, RG); else FGCN_TWX_GO(TN_, 3, 4, RG); } \
            else { if (one) FGCN_TWX_GO(TN_, 1, 8, RG); else if (two) FGCN_TWX_GO(TN_, 2, 8, RG); else FGCN_TWX_GO(TN_, 3, 8, RG); }      \
        } else {                                                                                                \
            if (half) { if (one) FGCN_TWX_GO(TN_, 1, 4, false); else if (two) FGCN_TWX_GO(TN_, 2, 4, false); else FGCN_TWX_GO(TN_, 3, 4, false); } \
            else { if (one) FGCN_TWX_GO(TN_, 1, 8, false); else if (two) FGCN_TWX_GO(TN_, 2, 8, false); else FGCN_TWX_GO(TN_, 3, 8, false); }      \
        }                                                                                                       \
    } while (0)
    if (N <= 64) {
        if constexpr (!CH || NTAP <= 3) FGCN_TWX_LAUNCH(64);
    } else {
        FGCN_TWX_LAUNCH(128);
    }
#undef FGCN_TWX_LAUNCH
#undef FGCN_TWX_GO
}

static int twgrad_launch(const float* a, const float* g, float* partial, int B, int T_g, int V, int K, int N,
                         int ld_a, int ld_g, int T_a_full, int a_s, int a_o, int Th_a, int nacc, int chunk_mode,
                         int shift0, int tap0, int tap_step, int taps_total, int nsplit, const unsigned* a_amax,
                         const unsigned* g_amax, void* stream, const char* what, bool in16 = false) {
    FGCN_REQUIRE(a && g && partial, FGCN_E_BADARG, "%s: null pointer", what);
    FGCN_REQUIRE(!in16 || fgcn::math_mode() == FGCN_MATH_BF16, FGCN_E_BADARG, "%s: bfloat16 inputs need math mode bf16", what);
    FGCN_REQUIRE(B > 0 && T_g > 0 && V > 0 && V <= FGCN_MAX_V && K > 0 && N > 0 && nsplit > 0 && nsplit <= 65535,
                 FGCN_E_BADARG, "%s: bad sizes B=%d T_g=%d V=%d K=%d N=%d nsplit=%d", what, B, T_g, V, K, N, nsplit);
    FGCN_REQUIRE(K % 4 == 0 && N % 4 == 0 && ld_a % 4 == 0 && ld_g % 4 == 0 && ld_a >= K && ld_g >= N, FGCN_E_ALIGN,
                 "%s: K, N and the row strides must be multiples of 4 (K=%d N=%d ld_a=%d ld_g=%d)", what, K, N, ld_a, ld_g);
    FGCN_REQUIRE(aligned16(a) && aligned16(g), FGCN_E_ALIGN, "%s: 16-byte alignment", what);
    FGCN_REQUIRE(a_s >= 1 && a_o >= 0 && Th_a > 0 && (long long)(Th_a - 1) * a_s + a_o < T_a_full, FGCN_E_BADARG,
                 "%s: frame view exceeds the tensor", what);
    const long long a_bytes = (long long)B * T_a_full * V * ld_a * (in16 ? 2 : 4), g_bytes = (long long)B * T_g * V * ld_g * (in16 ? 2 : 4);
    const int parts = twgrad_parts(N, chunk_mode);
    const bool x3 = twgrad_use_x3(N, chunk_mode);
    const long long p_bytes = (long long)nsplit * parts * taps_total * K * N * 4;
    FGCN_REQUIRE(a_bytes < 0x7FFF0000ll && g_bytes < 0x7FFF0000ll && p_bytes < 0x7FFF0000ll, FGCN_E_BADARG,
                 "%s: tensors must be smaller than 2 GiB (32-bit buffer offsets)", what);
    TWgradP p;
    p.a = a; p.g = g; p.partial = partial;
    p.B = B; p.T_g = T_g; p.V = V; p.K = K; p.N = N; p.ld_a = ld_a; p.ld_g = ld_g;
    p.T_a_full = T_a_full; p.a_s = a_s; p.a_o = a_o; p.Th_a = Th_a;
    p.shift0 = shift0; p.tap0 = tap0; p.tap_step = tap_step; p.taps_total = taps_total;
    p.in16 = in16 ? 1 : 0;
    FGCN_REQUIRE(!in16 || x3, FGCN_E_BADARG, "%s: bfloat16 inputs need the split kernel", what);
    p.stage_rows = x3 ? x3_rows(N <= 64 ? 64 : 128, twgrad_x3_waves(N, chunk_mode)) : (N <= 64 ? 128 : 64);
    p.stages_per_sample = (int)cdiv((long long)T_g * V, p.stage_rows);
    p.total_stages = B * p.stages_per_sample;
    p.stages_per_split = (int)cdiv(p.total_stages, nsplit);
    p.tiles_n = (int)cdiv(N, N <= 64 ? 64 : 128);
    p.chunk_mode = chunk_mode;
    p.win_rows = chunk_mode ? p.stage_rows : p.stage_rows + (nacc - 1) * V;
    p.a_bytes = (unsigned)a_bytes; p.g_bytes = (unsigned)g_bytes; p.p_bytes = (unsigned)p_bytes;
    const bool two = x3 && fgcn::f16x2_products() && a_amax && g_amax;     // both maxima known: the f16x2 form
    p.a_amax = two ? a_amax : nullptr;
    p.g_amax = two ? g_amax : nullptr;
    const int planes = chunk_mode ? nacc : 1;
    const int tn_x3 = N <= 64 ? 64 : 128;
    const int wv_x3 = twgrad_x3_waves(N, chunk_mode);
    const int win_x3 = chunk_mode ? nacc * x3_rows(tn_x3, wv_x3) : p.win_rows;
    // tap mode, more than one tap: the circular window image (tuning key 6 bit 7 switches it off)
    p.ring_rows = (x3 && !chunk_mode && nacc > 1 && !(fgcn::tuning(6) & 128)) ? (win_x3 <= 256 ? 256 : 512) : 0;
    const size_t np_x3 = fgcn::math_mode() == FGCN_MATH_BF16 ? 1 : (two ? 2 : 3), g_x3 = (size_t)x3_rows(tn_x3, wv_x3) * x3_sg(tn_x3);
    if (p.ring_rows && np_x3 * ((size_t)p.ring_rows * X3_SA + g_x3) > 160 * 1024) p.ring_rows = 0;   // (64-column tiles, 128-row stages)
    const size_t lds = x3 ? np_x3 * ((size_t)(p.ring_rows ? p.ring_rows : win_x3) * X3_SA + g_x3)
                          : (size_t)(((p.win_rows + 7) / 8) * 256 * planes + 8192) * sizeof(float);
    FGCN_REQUIRE(!x3 || win_x3 <= (p.ring_rows ? p.ring_rows : (wv_x3 == 8 ? 64 * X3_APASS : x3_rows(tn_x3, 4) + 8 * 32)), FGCN_E_BADARG,
                 "%s: window of %d rows too large", what, win_x3);
    FGCN_REQUIRE(lds <= 160 * 1024, FGCN_E_BADARG, "%s: stage needs %zu bytes of LDS", what, lds);
    const int tiles_k = (int)cdiv(K, chunk_mode ? 32 * nacc : 32);
    dim3 grid((unsigned)(tiles_k * p.tiles_n), (unsigned)nsplit);
    p.per_xcd = 0;
    p.tiles_xy = tiles_k * p.tiles_n;
    p.n_split = nsplit;
    if (x3 && p.tiles_xy > 1 && !(fgcn::tuning(5) & 64)) {      // key 5 bit 6: the plain 2-D grid (A/B control)
        p.per_xcd = (int)cdiv((long long)p.tiles_xy * nsplit, 8);
        grid = dim3((unsigned)(p.per_xcd * 8));
    }
    hipStream_t s = (hipStream_t)stream;
    if (x3 && chunk_mode) {
        switch (nacc) {
            case 6: launch_twgrad_x3<6, true>(p, N, grid, lds, s); break;
            case 5: launch_twgrad_x3<5, true>(p, N, grid, lds, s); break;
            case 4: launch_twgrad_x3<4, true>(p, N, grid, lds, s); break;
            case 3: launch_twgrad_x3<3, true>(p, N, grid, lds, s); break;
            case 2: launch_twgrad_x3<2, true>(p, N, grid, lds, s); break;
            case 1: launch_twgrad_x3<1, true>(p, N, grid, lds, s); break;
            default: return fgcn::fail(FGCN_E_BADARG, "%s: %d chunks per pass not instantiated", what, nacc);
        }
        return launch_status(what);
    }
    if (x3) {
        switch (nacc) {
            case 9: launch_twgrad_x3<9, false>(p, N, grid, lds, s); break;
            case 5: launch_twgrad_x3<5, false>(p, N, grid, lds, s); break;
            case 4: launch_twgrad_x3<4, false>(p, N, grid, lds, s); break;
            case 3: launch_twgrad_x3<3, false>(p, N, grid, lds, s); break;
            case 2: launch_twgrad_x3<2, false>(p, N, grid, lds, s); break;
            case 1: launch_twgrad_x3<1, false>(p, N, grid, lds, s); break;
            default: return fgcn::fail(FGCN_E_BADARG, "%s: %d taps per pass not instantiated", what, nacc);
        }
        return launch_status(what);
    }
    switch (nacc) {
        case 9: launch_twgrad<9>(p, N, grid, lds, s); break;
        case 6: launch_twgrad<6>(p, N, grid, lds, s); break;
        case 5: launch_twgrad<5>(p, N, grid, lds, s); break;
        case 4: launch_twgrad<4>(p, N, grid, lds, s); break;
        case 3: launch_twgrad<3>(p, N, grid, lds, s); break;
        case 2: launch_twgrad<2>(p, N, grid, lds, s); break;
        case 1: launch_twgrad<1>(p, N, grid, lds, s); break;
        default: return fgcn::fail(FGCN_E_BADARG, "%s: %d accumulators per wave not instantiated (1-6, 9)", what, nacc);
    }
    return launch_status(what);
}

extern "C" int fgcn_tconv_wgrad(const float* a, const float* g, float* partial, int B, int T_g, int V, int K, int N,
                                int ld_a, int ld_g, int T_a_full, int a_s, int a_o, int Th_a,
                                int ntaps, int shift0, int tap0, int tap_step, int taps_total, int nsplit,
                                const unsigned* a_amax, const unsigned* g_amax, void* stream) {
    FGCN_REQUIRE(ntaps >= 1 && ntaps <= 9 && tap_step >= 1 && tap0 >= 0 && tap0 + (ntaps - 1) * tap_step < taps_total,
                 FGCN_E_BADARG, "tconv_wgrad: taps (%d from %d step %d of %d)", ntaps, tap0, tap_step, taps_total);
    return twgrad_launch(a, g, partial, B, T_g, V, K, N, ld_a, ld_g, T_a_full, a_s, a_o, Th_a, ntaps, 0, shift0, tap0,
                         tap_step, taps_total, nsplit, a_amax, g_amax, stream, "tconv_wgrad");
}

// The same weight gradient from BFLOAT16 tensors a (the conv's input) and g (the gradient of its output), math mode bf16 only; ld_a / ld_g in
// elements.  Bit-identical to fgcn_tconv_wgrad on the f32 tensors the producers would have written (operands are rounded to bfloat16, to
// nearest even, when staged either way).
extern "C" int fgcn_tconv_wgrad_h(const unsigned short* a_h, const unsigned short* g_h, float* partial, int B, int T_g, int V, int K, int N,
                                  int ld_a, int ld_g, int T_a_full, int a_s, int a_o, int Th_a,
                                  int ntaps, int shift0, int tap0, int tap_step, int taps_total, int nsplit, void* stream) {
    FGCN_REQUIRE(ntaps >= 1 && ntaps <= 9 && tap_step >= 1 && tap0 >= 0 && tap0 + (ntaps - 1) * tap_step < taps_total,
                 FGCN_E_BADARG, "tconv_wgrad_h: taps (%d from %d step %d of %d)", ntaps, tap0, tap_step, taps_total);
    return twgrad_launch(reinterpret_cast<const float*>(a_h), reinterpret_cast<const float*>(g_h), partial, B, T_g, V, K, N, ld_a, ld_g, T_a_full,
                         a_s, a_o, Th_a, ntaps, 0, shift0, tap0, tap_step, taps_total, nsplit, nullptr, nullptr, stream, "tconv_wgrad_h", true);
}

// fgcn_pw_wgrad with bfloat16 tensors a and g (math mode bf16; ld_a / ld_g in elements): the shortcut convolutions' weight gradients under
// half-precision activation storage -- bit-identical to the float32 call on the same values
extern "C" int fgcn_pw_wgrad_h(const unsigned short* a_h, const unsigned short* g_h, float* partial, int B, int T_g, int V, int K, int N,
                               int ld_a, int ld_g, int T_a_full, int a_s, int a_o, int nsplit, void* stream) {
    FGCN_REQUIRE(a_s >= 1 && T_g > 0, FGCN_E_BADARG, "pw_wgrad_h: bad frame view");
    return twgrad_launch(reinterpret_cast<const float*>(a_h), reinterpret_cast<const float*>(g_h), partial, B, T_g, V, K, N, ld_a, ld_g, T_a_full, a_s,
                         a_o, T_g, fgcn_pw_wgrad_chunks(K, N), 1, 0, 0, 1, 1, nsplit, nullptr, nullptr, stream, "pw_wgrad_h", true);
}

/* in-channel chunks (32 channels each = one accumulator) per wave for a 1x1 weight gradient: a divisor of the chunk
 * count, at most 6 (128-column tiles, 64-row stages) or 3 (64-column tiles, 128-row stages): two workgroups per CU */
extern "C" int fgcn_pw_wgrad_chunks(int K, int N) {
    const int c = (int)cdiv(K, 32), cap = N <= 64 ? 3 : 6;
    if (c <= cap) return c;
    for (int d = cap; d >= 2; --d)
        if (c % d == 0) return d;
    return cap;
}

extern "C" int fgcn_pw_wgrad(const float* a, const float* g, float* partial, int B, int T_g, int V, int K, int N,
                             int ld_a, int ld_g, int T_a_full, int a_s, int a_o, int nsplit, const unsigned* a_amax,
                             const unsigned* g_amax, void* stream) {
    FGCN_REQUIRE(a_s >= 1 && T_g > 0, FGCN_E_BADARG, "pw_wgrad: bad frame view");
    return twgrad_launch(a, g, partial, B, T_g, V, K, N, ld_a, ld_g, T_a_full, a_s, a_o, T_g, fgcn_pw_wgrad_chunks(K, N), 1,
                         0, 0, 1, 1, nsplit, a_amax, g_amax, stream, "pw_wgrad");
}
