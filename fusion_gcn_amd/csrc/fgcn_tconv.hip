// Temporal (kt x 1) convolution, forward and data gradient, as a halo-tile implicit GEMM (north-star kernel 2).
//
// rows_gemm (fgcn_gemm.hip) re-stages the row-shifted input tile and a weight tile for every tap, with two barriers
// per 32-channel chunk.  Here the 128 output rows of a workgroup plus their temporal halo ((kt-1) frames =
// (kt-1)*V rows) are staged ONCE per 32-channel chunk and all kt taps run from that LDS image: tap j of output row m
// reads image row m + d_j*V (d_j = j*tb + tc).
//   A operand : LDS halo image [row][32 + 4 pad] floats, one ds_read_b128 feeds 4 MFMA k-steps (k = 8q + 4h + e),
//               read one (tap, q) step ahead of its MFMAs.
//   B operand : weights streamed from L2 straight into registers with buffer_load_dwordx4 from a k-interleaved packing
//               w4[tap][k/4][n][4] (lane = output channel -> 512 contiguous bytes per half-wave, 4 consecutive k per
//               lane).  Buffer addressing = per-lane VGPR offset (fixed) + SGPR offset (per step): no per-step vector
//               address arithmetic and no exec-masked branches, so hipcc keeps the next step's loads in flight behind a
//               counted vmcnt instead of draining them (a plain guarded global_load made it emit vmcnt(0) before every
//               MFMA group).  No weight staging: the only barriers are the pair around the halo fill.
//   Halo fill : buffer loads again; rows outside the tensor / the frame view carry an out-of-range offset and the
//               hardware returns zeros, so all fill loads issue back to back without branches.
//   Frames outside [0, T) (and rows of a neighbouring sample that share the image) are masked to exact zeros per
//   (row, tap) with a select on the A fragment.
// Strided convolutions run as two stride-1 passes over the even / odd frames ("virtual frames" with a stride and
// offset on the input or output side), so no tap is ever multiplied with a structurally-zero row.
#include "fgcn_common.hpp"
#include <type_traits>

namespace fgcn {

struct HaloP {
    const float* in;
    float* out;
    const float* w4;
    const float* bias;
    float* stats;
    long long Mv;                       // virtual rows = B * Tv * V
    unsigned in_bytes, w_bytes, out_bytes;
    unsigned w_plane_bytes;             // FGCN_MATH_BF16X3: bytes of one part (high / middle / low) of the split weights
    int tiles_m, tiles_n, per_xcd;   // per_xcd > 0: 1-D grid in XCD-aware order (column tiles of a row tile share an L2)
    int Tv, V, K, N, ld_in, ld_out;
    FastDiv dV, dTvV, dTv;              // row index / V, / (Tv V), frame index / Tv without integer divisions (rows < 2^29)
    int T_in_full, in_s, in_o, Th_in;   // input frame of virtual frame th: th*in_s + in_o (valid while th < Th_in)
    int T_out_full, out_s, out_o, Th_out;  // output frame of virtual frame th: th*out_s + out_o (th < Th_out)
    int taps, tb, tc;                   // d_j = j*tb + tc
    int dmin, halo_rows;
    int accumulate;
    // BatchNorm-backward sums in the epilogue (conv_halo_x3k32_kernel; data-gradient calls): with bn_a != NULL `stats` receives, per
    // row tile and column, sum dp and sum dp * (a - mean) * rstd, dp = value written * [bit of bn_mask] -- what
    // fgcn_bn_act_bwd_reduce computes in a pass of its own over the tensor this kernel has just produced
    const float* bn_a;
    const unsigned char* bn_mask;
    const float* bn_vec;
    // Input stage fused into the image fill (conv_halo_x3k32_kernel<.., FIN = true>; the 9x1 forward of an identity block): `in` is
    // the BatchNorm INPUT y and the image is G = relu(y * scale + shift + fin_res) (agcn.py:113-115), formed as the rows are
    // staged; the workgroups of the first column tile also store their own (non-halo) rows of G to fin_out and its sign image
    // (fgcn_bn_act's layout) to fin_mask -- what a stand-alone fgcn_bn_act pass in front of this kernel writes.
    const float* fin_vec;               // float[4][K] of fgcn_bn_finalize (scale at [2K, 3K), shift at [3K, 4K))
    const float* fin_res;               // the shortcut operand, laid out like `in`
    float* fin_out;                     // G, laid out like `in`
    unsigned char* fin_mask;            // rows * K / 8 bytes
    // Inference epilogue (conv_halo_x3k32_kernel<.., EPI = 4>; fgcn_tconv_halo_bn_relu): north-star kernel 2 as the north star states it --
    // "temporal 9x1 conv + BN + ReLU" in one kernel.  With eval-mode BatchNorm the statistics are constants, so the block's output stage
    // (agcn.py:49-51,134-136) is this kernel's epilogue: out = relu((acc + bias) * scale + shift + res * rscale + rshift).
    const float* ep_vec;                // float[4][N] of fgcn_bn_eval_coeffs (scale at [2N, 3N), shift at [3N, 4N))
    const float* ep_res;                // the block's shortcut operand, laid out like `out` (x, or the residual conv's output), or NULL
    const float* ep_rvec;               // float[4][N]: BatchNorm of the shortcut (the residual conv), or NULL (identity)
    unsigned* in_amax;                  // NP == 2: receives max |in| over everything staged (integer atomic maximum of the float bits) or NULL
    int stats_rows;                     // rows of `stats` the caller allocated (fgcn_tconv_halo_tiles: 128-row tiles); a kernel form with larger
                                        // row tiles fills tiles_m of them and zeroes the rest
};

// Timing probes (wrong results; tools/probes builds only): bit 0 = no output stores, bit 1 = no MFMAs, bit 2 = the image of the first chunk
// only (later chunks skip fetch, split and LDS writes), bit 3 = image fragments read for the first step of a chunk only, bit 4 = weight
// fragments loaded once, bit 5 = no statistics in the epilogue
#ifndef FGCN_PROBE_HALO
#define FGCN_PROBE_HALO 0
#endif
#ifndef FGCN_HALO_RING
#define FGCN_HALO_RING 4                // 2: the two-slot weight ring everywhere (A/B builds)
#endif
#ifndef FGCN_HALO_PF64
#define FGCN_HALO_PF64 1
#endif
constexpr int HAS = 36;                 // LDS row stride of the halo image (32 channels + 4 pad: conflict-free b128)
constexpr int HALO_MAX_STAGE = 13;      // ceil((128 + 8*32) / 32) + 1
constexpr int XSB = 80;                 // FGCN_MATH_BF16X3: LDS row stride in bytes of one bf16 part (32 channels + 8 pad)
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// MINB = workgroups per CU the register allocation aims at (launch-bounds hint; fgcn_set_tuning key 4 picks 2 or 3)
// MM: math mode -- FGCN_MATH_F32 or FGCN_MATH_BF16 (one bf16 MFMA per four f32 MFMAs, operands rounded as the fragments
// are read); FGCN_MATH_BF16X3 is conv_halo_x3k32_kernel below
template <int NT, int MINB, int MM>
__global__ __launch_bounds__(256, MINB) void conv_halo_kernel(HaloP p) {
    extern __shared__ __attribute__((aligned(16))) float Ah[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    // Consecutive workgroup ids go round-robin over the 8 XCDs: id b takes virtual tile (b % 8) * per_xcd + b / 8 with the
    // column tile fastest, so neighbouring row tiles (shared halo rows) and the column tiles of one row tile (same
    // image) meet in one L2.  Speed only.
    int bm, bn;
    if (p.per_xcd > 0) {
        const int vid = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
        if (vid >= p.tiles_m * p.tiles_n) return;
        bm = vid / p.tiles_n;
        bn = vid - bm * p.tiles_n;
    } else {
        bm = blockIdx.x;
        bn = blockIdx.y;
    }
    const long long m0 = (long long)bm * 128;
    const int n0 = bn * 32 * NT;
    const int V = p.V, TvV = p.Tv * p.V;
    const unsigned k4b = (tid & 7) * 16;            // byte offset of this thread's 4 channels inside a 32-chunk

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w4, 0, p.w_bytes, 0x00020000);

    // this lane's output row (for the per-tap frame mask)
    const long long mrow = m0 + wave * 32 + l31;
    const bool row_ok = mrow < p.Mv;
    const int th_lane = row_ok ? (int)(((unsigned)mrow / (unsigned)V) % (unsigned)p.Tv) : 0;   // Mv < 2^31

    // halo staging plan: image row r = (tid >> 3) + 32*i  <->  virtual row m0 + dmin*V + r.
    // Rows that do not exist get an offset beyond the buffer: the load returns zeros.
    unsigned src_off[HALO_MAX_STAGE];
    const int nstage = (p.halo_rows + 31) >> 5;
#pragma unroll
    for (int i = 0; i < HALO_MAX_STAGE; ++i) {
        src_off[i] = 0x80000000u;   // >= num_records: the hardware returns zeros
        const int r = (tid >> 3) + 32 * i;
        const long long hv = m0 + (long long)p.dmin * V + r;
        if (i < nstage && hv >= 0 && hv < p.Mv) {
            const unsigned hu = (unsigned)hv;                      // 32-bit decode: no 64-bit divisions
            const int n = (int)(hu / (unsigned)TvV);
            const int rem = (int)(hu - (unsigned)n * (unsigned)TvV);
            const int th = (int)((unsigned)rem / (unsigned)V);
            const int v = rem - th * V;
            const int fr = th * p.in_s + p.in_o;
            if (th < p.Th_in && fr < p.T_in_full)
                src_off[i] = (unsigned)(((((long long)n * p.T_in_full + fr) * V + v) * p.ld_in) * 4) + k4b;
        }
    }

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = zero16();

    const int IT = p.taps * 4;                      // (tap, q) steps per channel chunk, q = 8-channel group
    const int K4 = p.K >> 2;
    const int col = n0 + l31;                       // + nt*32
    unsigned wvoff[NT];                             // per-lane byte offset into w4: ((h*N + col) * 4 floats)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wvoff[nt] = (unsigned)(((long long)h * p.N + col + nt * 32) * 16);
    const float* arow = &Ah[(wave * 32 + l31) * HAS + 4 * h];

    auto w_soff = [&](int it, int kc) -> unsigned {   // wave-uniform byte offset of step (tap j, group q)
        const int j = it >> 2, q = it & 3;
        return (unsigned)(((long long)(j * K4 + (kc >> 2) + 2 * q) * p.N) * 16);
    };
    auto a_read = [&](int it, bool& ok) -> f32x4 {
        const int j = it >> 2, q = it & 3;
        const int d = j * p.tb + p.tc;
        const int ts = th_lane + d;
        ok = row_ok && ts >= 0 && ts < p.Th_in;
        return *reinterpret_cast<const f32x4*>(arow + (d - p.dmin) * V * HAS + 8 * q);
    };

    for (int kc = 0; kc < p.K; kc += 32) {
        __syncthreads();                             // previous chunk's image reads are done
        {
            f32x4 stage[HALO_MAX_STAGE];
#pragma unroll
            for (int i = 0; i < HALO_MAX_STAGE; ++i)
                if (i < nstage) stage[i] = buf_load4(rin, src_off[i], (unsigned)kc * 4);
#pragma unroll
            for (int i = 0; i < HALO_MAX_STAGE; ++i) {
                const int r = (tid >> 3) + 32 * i;
                if (i < nstage && r < p.halo_rows) *reinterpret_cast<f32x4*>(&Ah[r * HAS + (tid & 7) * 4]) = stage[i];
            }
        }
        __syncthreads();

        f32x4 b0[NT], b1[NT];
        {
            const unsigned so = w_soff(0, kc);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b0[nt] = buf_load4(rw, wvoff[nt], so);
        }
        bool ok0, ok1;
        f32x4 a0 = a_read(0, ok0), a1;
        for (int it = 0; it < IT; it += 2) {         // IT is even (4 steps per tap)
            {
                const unsigned so = w_soff(it + 1, kc);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b1[nt] = buf_load4(rw, wvoff[nt], so);
            }
            a1 = a_read(it + 1, ok1);
            if (!ok0) a0 = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (MM != 0) {
                const Frag<MM> ap = make_frag<MM>(a0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma_frag<MM>(ap, make_frag<MM>(b0[nt]), acc[nt]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32(a0[e], b0[nt][e], acc[nt]);
            }
            if (it + 2 < IT) {
                const unsigned so = w_soff(it + 2, kc);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b0[nt] = buf_load4(rw, wvoff[nt], so);
                a0 = a_read(it + 2, ok0);
            }
            if (!ok1) a1 = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (MM != 0) {
                const Frag<MM> ap = make_frag<MM>(a1);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma_frag<MM>(ap, make_frag<MM>(b1[nt]), acc[nt]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma32(a1[e], b1[nt][e], acc[nt]);
            }
        }
    }


    // ---- epilogue ---------------------------------------------------------------------------------------------------
    const bool plain_out = p.out_s == 1 && p.out_o == 0 && p.T_out_full == p.Tv && p.Th_out == p.Tv;
    // Branch-free buffer stores (rows / channels / frames that do not exist carry the out-of-range offset and are
    // dropped): guarded global stores made hipcc wait for vmcnt(0) after every single store.
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.bias ? p.bias : p.w4), 0, p.bias ? (unsigned)p.N * 4u : 0u, 0x00020000);
    float ssum[NT], ssq[NT], bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        ssum[nt] = 0.f;
        ssq[nt] = 0.f;
        bv[nt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                               rbias, col + nt * 32 < p.N ? (unsigned)(col + nt * 32) * 4u : OOB, 0, 0));
    }
    unsigned rowoff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long m = m0 + wave * 32 + acc_row(r, lane);
        bool ok = m < p.Mv;
        unsigned orow = (unsigned)(ok ? m : 0);
        if (!plain_out) {                              // wave-uniform
            const int n = (int)(orow / (unsigned)TvV);
            const int rem = (int)(orow - (unsigned)n * (unsigned)TvV);
            const int th = (int)((unsigned)rem / (unsigned)V);
            const int v = rem - th * V;
            ok = ok && th < p.Th_out;
            orow = (unsigned)((n * p.T_out_full + th * p.out_s + p.out_o) * V + v);
        }
        rowoff[r] = ok ? orow * (unsigned)p.ld_out * 4u : OOB;
    }
    // (one branch around the whole loop: an `if (p.accumulate)` around the loads inside it made hipcc drain vmcnt(0) at every join,
    // on the plain path too)
    auto epilogue = [&](auto acc_c) {
        constexpr bool ACC = decltype(acc_c)::value;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = col + nt * 32;
            const unsigned coff = c < p.N ? (unsigned)c * 4u : OOB;
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                old[r] = 0.f;
                if constexpr (ACC)
                    old[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rout, (rowoff[r] == OOB || coff == OOB) ? OOB : rowoff[r] + coff, 0, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned off = (rowoff[r] == OOB || coff == OOB) ? OOB : rowoff[r] + coff;
                const float val = acc[nt][r] + bv[nt] + old[r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rout, off, 0, 0);
                const float kept = off != OOB ? val : 0.f;
                ssum[nt] += kept;
                ssq[nt] += kept * kept;
            }
        }
    };
    if (p.accumulate) epilogue(std::true_type{});      // wave-uniform
    else epilogue(std::false_type{});
    if (p.stats) {
        constexpr int BN = 32 * NT;
        __syncthreads();
        float* red = Ah;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float a = ssum[nt] + __shfl_xor(ssum[nt], 32);
            const float b = ssq[nt] + __shfl_xor(ssq[nt], 32);
            if (lane < 32) {
                red[(0 * 4 + wave) * BN + nt * 32 + lane] = a;
                red[(1 * 4 + wave) * BN + nt * 32 + lane] = b;
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, c = tid - which * BN;
            if (n0 + c < p.N)
                p.stats[((long long)bm * 2 + which) * p.N + n0 + c] =
                    red[(which * 4 + 0) * BN + c] + red[(which * 4 + 1) * BN + c] + red[(which * 4 + 2) * BN + c] +
                    red[(which * 4 + 3) * BN + c];
        }
    }
}

// ---- FGCN_MATH_BF16X3 / FGCN_MATH_BF16: the halo-tile scheme on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16) ----------------
// The image is split into its bf16 parts as it is staged (planes of unpadded, XOR-swizzled bf16 rows: one ds_read_b128 per part =
// the 8 k of a lane), the weights come pre-split from HBM (fgcn_pack_split3: [part][tap][k/8][n][8]).  Six bf16 MFMAs move the
// work of eight f32 ones in 3/8 of the cycles, so the weight stream (6 instead of 4 bytes per weight) would need 4x the L2
// bandwidth of the f32 kernel: the waves are arranged 2 x 2 over the (128 rows x 64 NT columns) tile -- a wave owns 64 rows x
// 32 NT columns and every weight fragment feeds all its row tiles (the 4 x 1 arrangement of the f32 kernel measured L2-bound:
// 64 B/clk/CU).  Weights are prefetched one unit ahead in a ring of two fragment sets, across step and chunk boundaries; past the
// last unit the address wraps to the first one (a valid, unused load).  KC = channels per staged chunk: 32 for the temporal convs
// (image = tile + halo rows); 64 for 1x1 convolutions (taps = 1: no halo), where the NEXT chunk's rows are requested before the
// MFMAs of the current one and parked in registers, so the global latency is not paid between two barriers.
// (The first build of this kernel issued v_mfma_f32_32x32x16_bf16: 4-10 % slower at equal FLOPs -- half the accumulator traffic per
// FLOP with the 16x16 shape and the chip holds its clock better; that form was removed in round 3.)
// A step is one tap x 32 channels (the
// whole chunk at KC = 32); lane (i = lane & 15, g = lane >> 4) holds k = 8g + j of row / column i.  The wave's 64 x (32 NT)
// tile is 4 row tiles x 2 NT column tiles of 16 x 16; the four image fragments of a step stay resident and are replaced
// one by one during the step's last unit (right after each one's last MFMA), the weight fragments ride the same two-slot
// ring, one unit (one 16-column tile = 24 MFMAs) ahead.
// NP = bf16 parts per operand: 3 = FGCN_MATH_BF16X3 (six partial products), 1 = FGCN_MATH_BF16 (operands rounded to bf16 once --
// activations as the image is staged, weights by fgcn_pack_split3, whose part 0 is the round-to-nearest-even bf16 -- and ONE
// MFMA per product group: a sixth of the matrix work, a third of the LDS image and of the weight stream).
// EPI = what the epilogue does besides bias + store (compile time: run-time `if (p.accumulate)` / `if (bnb)` guards around the
// epilogue's loads, wave-uniform as they were, made hipcc branch inside the unrolled loops and drain vmcnt(0) in front of every
// group of four stores -- sixteen serialised write round trips per workgroup; found with the timing probes of fgcn_pw.hip):
//   0  store;  2  store + the BatchNorm-backward sums (bn_a / bn_mask / bn_vec);  3  accumulate: load / add / store with the old
//   values of a row tile requested one row tile AHEAD of the stores (the second pass of a strided forward convolution; statistics
//   are those of the final values);  4  the INFERENCE output stage: BatchNorm (eval-mode coefficients) + shortcut + ReLU applied to the
//   accumulators, the shortcut values of a row tile requested one row tile ahead like EPI 3's old values -- the block's output without a
//   pre-BatchNorm tensor and without a bn_act pass (HaloP::ep_*).  (1 was accumulation by one no-return float atomic per element -- deterministic, every element
//   has one contributor -- measured 5-40 % slower in fgcn_pw.hip: the L2 performs about one per clock and channel.)
// (A 96-row tile for images that do not fit LDS twice is not needed: with the swizzled unpadded image the 128-row tile fits up to
// 36 joints, FGCN_MAX_V is 32.)
// WR = wave rows: 2 = waves 2 x 2 over (128 rows x 64 NT columns); 4 = waves 4 x 1 over (192 rows x 32 NT columns), the form for 64
// output columns (NT = 2): every image fragment then feeds FOUR 16-column units instead of two (half the LDS fragment reads per MFMA --
// the 2 x 2 form at 64 columns reads 12 KB of fragments per 48 MFMAs and wave) and the halo image is re-staged 2.04x instead of 2.56x.
// IN16 (NP = 1): `in` is a BFLOAT16 tensor (half-precision storage of the conv's input, written by its producer: fgcn_bn_act_h /
// fgcn_bn_act_bwd_apply_h) -- the image rows are copied, 8 bytes per four channels, instead of fetched as f32 and rounded here.  The staged
// bytes are the same either way (one round-to-nearest-even per value), so the results are bit-identical to the f32-input form; the row
// traffic through L2 -- what bounds this kernel in math mode bf16, where the matrix work is a sixth -- halves.
// IN16 = 2: `out` is a bfloat16 tensor as well (half-precision ACTIVATION storage, the `_t` entry point: the pre-BatchNorm output U of the
// forward, dG of the data gradient; ld_out in elements).  Epilogue form 0 only; the BatchNorm partial sums are those of the float32
// accumulators (before the rounding).  A lane holds ONE column of four rows: adjacent lanes exchange a value (DPP) so that each stores
// two adjacent columns of one row as a dword -- half the store instructions of the float32 form instead of twice as many 2-byte stores.
template <int NT, int KC, int NP, int EPI = 0, bool FIN = false, int WR = 2, bool STR = false, int IN16 = 0>   // STR: non-temporal output stores (stream_out)
__global__ __launch_bounds__(256, (NP == 1 && !FIN && EPI != 4) ? 3 : 2) void conv_halo_x3k32_kernel(HaloP p) {   // (EPI 4 at 168 registers spilled)
    static_assert(!IN16 || (NP == 1 && !FIN), "bfloat16 input: the one-part kernel without the fused input stage");
    static_assert(IN16 != 2 || EPI == 0, "bfloat16 output: the plain store epilogue");
    constexpr bool O16 = IN16 == 2;
    static_assert(!(STR && EPI == 3), "an accumulating epilogue stores plainly");
    static_assert(WR == 2 || (WR == 4 && KC == 32 && !FIN), "wave arrangement: 2 x 2, or 4 x 1 for the tap form");
    static_assert(!FIN || KC == 32, "the fused input stage is built for the tap form (32-channel chunks)");
    static_assert((EPI == 0 || EPI == 2 || EPI == 3 || EPI == 4) && !(FIN && EPI != 0),
                  "epilogue: store / store + BatchNorm-backward sums / accumulate / inference output stage");
    constexpr int MTW = WR == 4 ? 3 : 4;             // 16-row tiles per wave
    constexpr int BMR = WR * 16 * MTW;               // output rows per workgroup: 128 (2 x 2 waves) or 192 (4 x 1)
    static_assert(NP == 1 || NP == 2 || NP == 3, "one or three bf16 parts per operand, or two f16 parts (FGCN_PRODUCTS_F16X2)");
    static_assert(!(FIN && NP == 2), "the fused input stage is not built for the f16x2 products");
    static_assert((NT == 1 || NT == 2 || (NT == 4 && WR == 4)) && (KC == 32 || KC == 64), "wave tile is 64 rows x 32 or 64 columns (48 x 128 as 4 x 1 waves)");
    // Image rows are KC bf16 = 64 / 128 bytes with NO padding; the 32-byte blocks of a row are XOR-swizzled with row bits instead.
    // A fragment read is ds_read_b128 of (row base + l15, 16-byte chunk g4 [+ 4 per 32-channel step]); its four lane groups are the
    // NON-contiguous sets {0-3,12-15,20-27}, ... (MI355X_MICROARCH.md, LDS): each holds 8 rows at chunk c and 8 other rows at
    // chunk c + 1, and with padded rows (80 / 144 bytes) three of its sixteen 16-byte slots always fell on busy banks whatever
    // the padding -- SQ_LDS_BANK_CONFLICT was 0.49 (KC 32) / 0.37 (KC 64) of SQ_LDS_IDX_ACTIVE.  Exhaustive search over strides
    // and row-bit swizzles, all row bases (the tap shift j * V is arbitrary): these two are conflict-free and also the smallest.
    constexpr int XS = KC * 2;
    auto swz = [](int r) -> unsigned { return KC == 32 ? (unsigned)(r & 4) << 3 : (unsigned)(r & 6) << 4; };
    constexpr int TPR = KC / 4;
    constexpr int RPP = 256 / TPR;
    constexpr int SPC = KC / 32;                     // steps per tap and chunk
    // PF: the NEXT chunk's rows are requested before the MFMAs of the current one and parked in registers.  Always for the 1x1 form
    // (KC = 64); for the 9-tap form where the registers allow it: 64 output columns (NT = 1: half the accumulators) -- a 64-channel
    // tile is only two chunks, so the second chunk's exposed global-load latency was a visible share of it
    constexpr bool PF = KC == 64 || (NT == 1 && NP == 3 && !FIN && FGCN_HALO_PF64);
    constexpr int NST = KC == 64 ? 8 : HALO_MAX_STAGE;
    constexpr int NU = 2 * NT;                       // 16-column tiles (units) of a wave
#ifndef FGCN_HALO_RING_NP1
#define FGCN_HALO_RING_NP1 4            // weight ring slots of the one-part (FGCN_MATH_BF16) kernel: with ONE MFMA per fragment and row tile the two-slot ring left the L2 latency of the weight stream exposed (28.40 -> 27.43 ms per bf16 step; A/B builds: 2)
#endif
    constexpr int RS = (FGCN_HALO_RING == 4 && NU == 4 && (NP >= 2 || FGCN_HALO_RING_NP1 == 4) && (EPI != 2 || WR == 4) && !FIN) ? 4 : 2;   // weight ring slots (requested RS - 1 units ahead)
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float Ah[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wr = WR == 4 ? wave : wave >> 1, wc = WR == 4 ? 0 : wave & 1;
    // per_xcd > 0: 1-D grid in XCD-aware order -- consecutive workgroup ids go round-robin over the 8 XCDs, so id b takes virtual
    // tile (b % 8) * per_xcd + b / 8, column tile fastest: the column tiles of a row tile (same image) and neighbouring row tiles
    // (shared halo rows) run back to back on ONE XCD and find each other's rows in its L2 instead of fetching them again
    int bm = blockIdx.x, bn = blockIdx.y;
    if (p.per_xcd > 0) {
        const int vid = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
        if (vid >= p.tiles_m * p.tiles_n) return;
        bm = vid / p.tiles_n;
        bn = vid - bm * p.tiles_n;
    }
    const int m0 = bm * BMR;                         // (32-bit row math: the launcher bounds the rows of the split kernels by 2^29)
    const int Mv = (int)p.Mv;
    constexpr int BN = (4 / WR) * NT * 32;
    const int n0 = bn * BN;
    const int V = p.V, TvV = p.Tv * p.V;
    const unsigned k4b = (tid % TPR) * 16;

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const unsigned char*>(p.w4) + (NP == 2 ? 16 : 0)), 0, p.w_bytes, 0x00020000);
    // NP == 2 (f16x2 products, fgcn_common.hpp): block scaling.  The staged chunk (tile + halo rows x KC channels) is scaled by 2^ea as it
    // is split, ea from its largest magnitude (wave maxima through four LDS words); the accumulators are rescaled (a power of two:
    // exact) whenever the scale moves, and the epilogue multiplies 2^-ea 2^-ew back out (ew: the packed form's scale, header word of
    // FGCN_PACK_SPLIT2H).
    constexpr int EA_NONE = 1000;
    const int ew = NP == 2 ? min(scale_exp_for(*reinterpret_cast<const unsigned*>(p.w4)), 126) : 0;
    int ea = EA_NONE, abound = 0;

    bool row_ok[MTW];
    int th_lane[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int mrow = m0 + wr * (16 * MTW) + mt * 16 + l15;
        row_ok[mt] = mrow < Mv;
        const unsigned fr_all = fastdiv(row_ok[mt] ? (unsigned)mrow : 0u, p.dV);     // frame index over all samples
        th_lane[mt] = (int)(fr_all - fastdiv(fr_all, p.dTv) * (unsigned)p.Tv);
    }

    unsigned src_off[NST];
    const int nstage = (p.halo_rows + RPP - 1) / RPP;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        src_off[i] = OOB;
        const int r = tid / TPR + RPP * i;
        const int hv = m0 + p.dmin * V + r;
        if (i < nstage && hv >= 0 && hv < Mv) {
            const unsigned hu = (unsigned)hv;
            const int n = (int)fastdiv(hu, p.dTvV);
            const int rem = (int)(hu - (unsigned)n * (unsigned)TvV);
            const int th = (int)fastdiv((unsigned)rem, p.dV);
            const int v = rem - th * V;
            const int fr = th * p.in_s + p.in_o;
            if (th < p.Th_in && fr < p.T_in_full)
                src_off[i] = (unsigned)(((n * p.T_in_full + fr) * V + v) * p.ld_in) * 4u + k4b;      // (the launcher bounds the tensors by 2 GiB)
        }
    }

    f32x4 acc[MTW][NU];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nu = 0; nu < NU; ++nu) acc[mt][nu] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int col = n0 + wc * NT * 32 + l15;         // + nu*16
    unsigned wvoff[NU];                              // per-lane byte offset into one part: (g4*N + col) * 8 bf16
#pragma unroll
    for (int nu = 0; nu < NU; ++nu) wvoff[nu] = (unsigned)(((long long)g4 * p.N + col + nu * 16) * 16);

    unsigned char* Xh = reinterpret_cast<unsigned char*>(Ah);
    const unsigned plane = (unsigned)p.halo_rows * XS;
    const int xrow = wr * (16 * MTW) + l15;          // this lane's image row before the tap shift and the tile index
    const int IT2 = p.taps * SPC;                    // (tap, 32-channel group) steps per chunk
    const int K8 = p.K >> 3;
    bool probe_started = false;                      // (FGCN_PROBE_HALO: set after the first chunk)
    auto load_w = [&](u32x4v (&dst)[NP], int nu, int it, int kc) {
        if (it >= IT2) {                             // (at most one step past the chunk)
            it -= IT2;
            kc += KC;
        }
        if (kc >= p.K) {                             // past the last unit: a valid, unused load
            it = 0;
            kc = 0;
        }
        const int j = it / SPC, s2 = it - j * SPC;
        const unsigned so = (unsigned)(((long long)(j * K8 + (kc >> 3) + 4 * s2) * p.N) * 16);
        if ((FGCN_PROBE_HALO & 16) && probe_started) return;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            dst[pl] = __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff[nu], so + pl * p.w_plane_bytes, 0);
    };
    auto load_a = [&](u32x4v (&dst)[NP], int mt, int it) {
        if ((FGCN_PROBE_HALO & 8) && it > 0) return;                     // (probe: the first step's image fragments only)
        const int j = it / SPC, s2 = it - j * SPC;
        const int d = j * p.tb + p.tc;
        const int r = xrow + (d - p.dmin) * V + mt * 16;
        const unsigned char* src = Xh + r * XS + ((unsigned)(16 * g4 + 64 * s2) ^ swz(r));
        const int ts = th_lane[mt] + d;
        const bool ok = row_ok[mt] && ts >= 0 && ts < p.Th_in;          // frame mask of this (row, tap)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            const u32x4v v = *reinterpret_cast<const u32x4v*>(src + pl * plane);
            dst[pl] = ok ? v : u32x4v{0u, 0u, 0u, 0u};
        }
    };

    f32x4 stage[NST], stage2[FIN ? NST : 1];   // (FIN: only HALF of each live at a time)
    // (the descriptors of the fused stage's three tensors are built where they are used, from the kernel arguments: kept live
    // across the MFMA loop they pushed the scalar registers over their limit and the spills landed in vector registers)
    // (the fused input stage fills the image in two halves -- stages [0, HALF) then [HALF, NST) -- so that the two staged tensors
    // take 2 x HALF instead of 2 x NST registers: all at once spilled)
    constexpr int HALF = FIN ? (NST + 1) / 2 : NST;
    auto fetch = [&](int kc, int lo = 0, int hi = 64) {
#pragma unroll
        for (int i = 0; i < NST; ++i)
            if (i >= lo && i < hi && i < nstage) {
                if constexpr (IN16) {     // four bfloat16 = 8 bytes (every byte offset of the f32 form halves; the descriptor is the bf16 tensor's)
                    const u32x2 h2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rin, src_off[i] == OOB ? OOB : src_off[i] >> 1, (unsigned)kc * 2, 0));
                    const unsigned b0 = h2[0], b1 = h2[1];
                    stage[i] = f32x4{__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1), 0.f, 0.f};
                    continue;
                }
                stage[i] = buf_load4(rin, src_off[i], (unsigned)kc * 4);
                if constexpr (FIN)
                    stage2[i] = buf_load4(__builtin_amdgcn_make_buffer_rsrc((void*)p.fin_res, 0, p.in_bytes, 0x00020000), src_off[i],
                                          (unsigned)kc * 4);
            }
    };
    auto deposit = [&](int kc, int lo = 0, int hi = 64) {   // split the staged rows into the three bf16 planes
        const float a_scale = (NP == 2 && ea != EA_NONE) ? exp2i(ea) : 1.f;
        f32x4 fsc = {0.f, 0.f, 0.f, 0.f}, fsh = fsc;
        const __amdgpu_buffer_rsrc_t rgo = __builtin_amdgcn_make_buffer_rsrc((void*)(FIN ? (void*)p.fin_out : (void*)p.out), 0,
                                                                             FIN ? p.in_bytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rgm = __builtin_amdgcn_make_buffer_rsrc((void*)(FIN ? (void*)p.fin_mask : (void*)p.out), 0,
                                                                             FIN ? p.in_bytes >> 5 : 0u, 0x00020000);
        const int own0 = -p.dmin * V;                // image row of the tile's first own (non-halo) row
        if constexpr (FIN) {
            fsc = *reinterpret_cast<const f32x4*>(p.fin_vec + 2 * p.K + kc + (tid % TPR) * 4);
            fsh = *reinterpret_cast<const f32x4*>(p.fin_vec + 3 * p.K + kc + (tid % TPR) * 4);
        }
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            if (i < lo || i >= hi) continue;
            const int r = tid / TPR + RPP * i;
            if constexpr (FIN) {
                if (i < nstage) {                    // (wave-uniform; the shuffles below need whole 8-lane row groups)
                    // G = relu(BatchNorm(y) + shortcut) of this row chunk; rows outside the tensor / frame view stay exact zeros
                    const bool live = src_off[i] != OOB;
                    f32x4 v = stage[i] * fsc + fsh + stage2[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = live ? fmaxf(v[e], 0.f) : 0.f;
                    stage[i] = v;
                    // the first column tile's workgroups keep their own rows of G and its sign image (one dword = this row's 32
                    // channels of the chunk: the eight 4-bit groups of the row's eight lanes, gathered with three shuffles)
                    const bool mine = live && bn == 0 && r >= own0 && r < own0 + BMR;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rgo, mine ? src_off[i] : OOB,
                                                           (unsigned)kc * 4, 0);
                    unsigned w = ((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u))
                                 << (4 * (lane & 7));
                    w |= (unsigned)__shfl_xor((int)w, 1);
                    w |= (unsigned)__shfl_xor((int)w, 2);
                    w |= (unsigned)__shfl_xor((int)w, 4);
                    __builtin_amdgcn_raw_buffer_store_b32(w, rgm, (mine && (lane & 7) == 0) ? (src_off[i] + (unsigned)kc * 4) >> 5 : OOB,
                                                          0, 0);
                }
            }
            if (i < nstage && r < p.halo_rows) {
                u32x2 ph, pm, pl;
                unsigned char* dst = Xh + r * XS + ((unsigned)((tid % TPR) * 8) ^ swz(r));
                if constexpr (NP == 2) {
                    split2h_x4(stage[i] * a_scale, ph, pm);
                    *reinterpret_cast<u32x2*>(dst) = ph;
                    *reinterpret_cast<u32x2*>(dst + plane) = pm;
                    continue;
                }
                if constexpr (IN16) {                // already bfloat16: a copy
                    // (element -> scalar first: __builtin_bit_cast on an ext-vector ELEMENT reads element 0 with hipcc 7.2)
                    const float e0 = stage[i][0], e1 = stage[i][1];
                    *reinterpret_cast<u32x2*>(dst) = u32x2{__builtin_bit_cast(unsigned, e0), __builtin_bit_cast(unsigned, e1)};
                    continue;
                }
                split3_x4(stage[i], ph, pm, pl);
                *reinterpret_cast<u32x2*>(dst) = ph;
                if constexpr (NP == 3) {
                    *reinterpret_cast<u32x2*>(dst + plane) = pm;
                    *reinterpret_cast<u32x2*>(dst + 2 * plane) = pl;
                }
            }
        }
    };

    u32x4v a[MTW][NP], wq[RS][NP];
#pragma unroll
    for (int nu = 0; nu < RS - 1; ++nu) load_w(wq[nu], nu, 0, 0);
    if constexpr (PF) fetch(0);
    float* smax = reinterpret_cast<float*>(Xh + NP * plane);      // NP == 2: the four waves' chunk maxima (16 bytes behind the planes)
    auto chunk_max = [&]() {                         // this wave's largest staged magnitude -> its LDS word
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < NST; ++i)
            if (i < nstage)
                m = fmaxf(fmaxf(m, fmaxf(fabsf(stage[i][0]), fabsf(stage[i][1]))), fmaxf(fabsf(stage[i][2]), fabsf(stage[i][3])));
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
        if (lane == 0) smax[wave] = m;
    };
    for (int kc = 0; kc < p.K; kc += KC) {
        if constexpr (NP == 2 && PF) chunk_max();    // (the chunk is already parked in registers)
        __syncthreads();                             // previous chunk's image reads are done
        if constexpr (FIN) {
            fetch(kc, 0, HALF);
            deposit(kc, 0, HALF);
            fetch(kc, HALF, NST);
            deposit(kc, HALF, NST);
        } else if ((FGCN_PROBE_HALO & 4) && probe_started) {
            // (probe: the first chunk's image stays)
        } else {
            if constexpr (!PF) fetch(kc);
            if constexpr (NP == 2) {
                if constexpr (!PF) {
                    chunk_max();
                    __syncthreads();
                }
                const unsigned mb = __builtin_bit_cast(unsigned, fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
                const int ec = __builtin_amdgcn_readfirstlane((mb >> 23) == 0u ? EA_NONE : min(scale_exp_for(mb), 126));
                if (p.in_amax && tid == 0 && bn == 0) atomicMax(p.in_amax, mb);   // (the column tiles of a row tile stage the same rows)
                // the chunk's scale: its own (largest magnitude into [2^14, 2^15)) whenever the accumulators can follow -- down always
                // (exact), up while their magnitude bound stays below 2^120 (abound: log2 bound of |acc| in units of the scale in
                // force; a chunk adds at most 2^44 per accumulator) -- so every chunk is split at full f16 resolution unless the
                // chunks of one tile span more than ~2^75
                if (ec != EA_NONE && ec != ea) {
                    int d = ea == EA_NONE ? 0 : ec - ea;
                    if (d > 120 - abound) d = 120 - abound;
                    if (ea == EA_NONE) ea = ec;
                    else if (d != 0) {
                        const float f = exp2i(d);
#pragma unroll
                        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                            for (int nu = 0; nu < NU; ++nu) acc[mt][nu] *= f;
                        ea += d;
                        abound += d;
                    }
                }
                abound = (abound > 44 ? abound : 44) + 1;
            }
            deposit(kc);
        }
        __syncthreads();
        if constexpr (PF) {
            if (kc + KC < p.K) fetch(kc + KC);       // lands during the MFMAs below
        }
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) load_a(a[mt], mt, 0);
#pragma unroll 1
        for (int it = 0; it < IT2; ++it) {
            const int itn = it + 1 < IT2 ? it + 1 : it;                  // the chunk's last step re-reads itself (unused)
#pragma unroll
            for (int nu = 0; nu < NU; ++nu) {
                const int t = nu + RS - 1;                               // the ring: the unit RS - 1 ahead (this step's, or the next's)
                if (t < NU) load_w(wq[t % RS], t, it, kc);
                else load_w(wq[t % RS], t - NU, it + 1, kc);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    if constexpr ((FGCN_PROBE_HALO & 2) != 0) acc[mt][nu][0] += __builtin_bit_cast(float, a[mt][0][0] ^ wq[nu % RS][0][0] ^ a[mt][NP - 1][3] ^ wq[nu % RS][NP - 1][3]);
                    else if constexpr (NP == 3) acc[mt][nu] = mfma_x3_k32(a[mt], wq[nu % RS], acc[mt][nu]);
                    else if constexpr (NP == 2) acc[mt][nu] = mfma_h2_k32(a[mt], wq[nu % RS], acc[mt][nu]);
                    else acc[mt][nu] = mfma_bf16_k32(a[mt][0], wq[nu % RS][0], acc[mt][nu]);
                    if (nu == NU - 1) load_a(a[mt], mt, itn);            // this fragment's last use: fetch the next step's
                }
            }
        }
        probe_started = true;
    }

    // ---- epilogue: bias, (accumulate), branch-free buffer stores, BatchNorm partial sums ---------------------------------
    // accumulator register r of lane (col l15, g4) = row 4 g4 + r of the 16 x 16 tile
    __builtin_amdgcn_sched_barrier(0);                    // (epilogue loads hoisted into the last MFMA step spilled registers)
    const bool plain_out = p.out_s == 1 && p.out_o == 0 && p.T_out_full == p.Tv && p.Th_out == p.Tv;
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.bias ? p.bias : p.w4), 0, p.bias ? (unsigned)p.N * 4u : 0u, 0x00020000);
    float ssum[NU], ssq[NU], bv[NU];
    unsigned coff[NU];
    const float un_a = (NP == 2 && ea != EA_NONE) ? exp2i(-ea) : 1.f, un_w = NP == 2 ? exp2i(-ew) : 1.f;
    constexpr bool bnb = EPI == 2;                        // BatchNorm-backward sums instead of the forward moments
    const __amdgpu_buffer_rsrc_t rba = __builtin_amdgcn_make_buffer_rsrc((void*)(bnb ? p.bn_a : p.out), 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbm = __builtin_amdgcn_make_buffer_rsrc((void*)(bnb ? (const void*)p.bn_mask : (const void*)p.out), 0,
                                                                         p.out_bytes >> 5, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbv = __builtin_amdgcn_make_buffer_rsrc((void*)(bnb ? p.bn_vec : p.w4), 0,
                                                                         bnb ? (unsigned)p.N * 8u : 0u, 0x00020000);
    float bmean[NU], brstd[NU];
#pragma unroll
    for (int nu = 0; nu < NU; ++nu) {
        ssum[nu] = 0.f;
        ssq[nu] = 0.f;
        coff[nu] = col + nu * 16 < p.N ? (unsigned)(col + nu * 16) * 4u : OOB;
        bv[nu] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, coff[nu], 0, 0));
        bmean[nu] = bnb ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbv, coff[nu], 0, 0)) : 0.f;
        brstd[nu] = bnb ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbv, coff[nu], (unsigned)p.N * 4u, 0)) : 0.f;
    }
    // all row offsets first (the frame-view mapping of the strided passes is the only branch of the epilogue, and no memory
    // operation lies behind it)
    unsigned rowoff[MTW][4];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wr * (16 * MTW) + mt * 16 + 4 * g4 + r;
            bool ok = m < Mv;
            unsigned orow = (unsigned)(ok ? m : 0);
            if (!plain_out) {                              // wave-uniform
                const int n = (int)fastdiv(orow, p.dTvV);
                const int rem = (int)(orow - (unsigned)n * (unsigned)TvV);
                const int th = (int)fastdiv((unsigned)rem, p.dV);
                const int v = rem - th * V;
                ok = ok && th < p.Th_out;
                orow = (unsigned)((n * p.T_out_full + th * p.out_s + p.out_o) * V + v);
            }
            rowoff[mt][r] = ok ? orow * (unsigned)p.ld_out * 4u : OOB;
        }
    // EPI 2: the BatchNorm input and the ReLU sign bits of a row tile's elements are requested one row tile AHEAD of its stores, so a
    // wait for them never includes the write acknowledgement of stores issued before them (vmcnt counts in issue order)
    float av[bnb ? 2 : 1][NU][4];
    unsigned mbits[bnb ? 2 : 1][NU][4];
    auto load_bn = [&](int mt, float (&a_)[NU][4], unsigned (&m_)[NU][4]) {
#pragma unroll
        for (int nu = 0; nu < NU; ++nu)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned off = (rowoff[mt][r] == OOB || coff[nu] == OOB) ? OOB : rowoff[mt][r] + coff[nu];
                a_[nu][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rba, off, 0, 0));
                m_[nu][r] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rbm, off == OOB ? OOB : off >> 5, 0, 0);
            }
    };
    constexpr bool fep = EPI == 4;                        // inference output stage (HaloP::ep_*)
    constexpr bool ldacc = EPI == 3 || fep;               // per-element operand of the epilogue: out's old values / the shortcut values
    const __amdgpu_buffer_rsrc_t rold = fep ? __builtin_amdgcn_make_buffer_rsrc((void*)(p.ep_res ? (const void*)p.ep_res : (const void*)p.out), 0,
                                                                                p.ep_res ? p.out_bytes : 0u, 0x00020000)
                                            : rout;
    float esc[NU], esh[NU], rsc[NU], rsh[NU];
    if constexpr (fep) {
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)p.ep_vec, 0, (unsigned)p.N * 16u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rrv = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ep_rvec ? p.ep_rvec : p.ep_vec), 0,
                                                                             p.ep_rvec ? (unsigned)p.N * 16u : 0u, 0x00020000);
        const float res_unit = p.ep_rvec ? 0.f : 1.f;     // (branch-free: an empty descriptor returns 0, the scalar addend makes the scale 1)
#pragma unroll
        for (int nu = 0; nu < NU; ++nu) {
            esc[nu] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, coff[nu], (unsigned)p.N * 8u, 0));
            esh[nu] = __builtin_fmaf(bv[nu], esc[nu], __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, coff[nu], (unsigned)p.N * 12u, 0)));
            rsc[nu] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrv, coff[nu], (unsigned)p.N * 8u, 0)) + res_unit;
            rsh[nu] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrv, coff[nu], (unsigned)p.N * 12u, 0));
        }
    }
    float oldv[ldacc ? 2 : 1][NU][4];
    auto load_old = [&](int mt, float (&o_)[NU][4]) {
#pragma unroll
        for (int nu = 0; nu < NU; ++nu)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                o_[nu][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rold, (rowoff[mt][r] == OOB || coff[nu] == OOB) ? OOB : rowoff[mt][r] + coff[nu], 0, 0));
    };
    if constexpr (bnb) load_bn(0, av[0], mbits[0]);
    if constexpr (ldacc) load_old(0, oldv[0]);
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        if constexpr (bnb) {
            if (mt + 1 < MTW) load_bn(mt + 1, av[(mt + 1) & 1], mbits[(mt + 1) & 1]);
        }
        if constexpr (ldacc) {
            if (mt + 1 < MTW) load_old(mt + 1, oldv[(mt + 1) & 1]);
        }
        if constexpr (O16) {
            // two rows at a time: the even lane of a pair stores columns (c, c + 1) of row rp, the odd lane those of row rp + 1
            const bool odd = lane & 1;
#pragma unroll
            for (int rp = 0; rp < 4; rp += 2) {
#pragma unroll
                for (int nu = 0; nu < NU; ++nu) {
                    const unsigned off0 = (rowoff[mt][rp] == OOB || coff[nu] == OOB) ? OOB : rowoff[mt][rp] + coff[nu];
                    const unsigned off1 = (rowoff[mt][rp + 1] == OOB || coff[nu] == OOB) ? OOB : rowoff[mt][rp + 1] + coff[nu];
                    const float v0 = acc[mt][nu][rp] + bv[nu], v1 = acc[mt][nu][rp + 1] + bv[nu];
                    const float other = lane_xor1(odd ? v0 : v1);
                    const unsigned pk = odd ? pack_bf16x2(other, v1) : pack_bf16x2(v0, other);
                    const unsigned mine = odd ? off1 : off0;       // (float32-form byte offset of this lane's row and column: halves, minus the odd lane's column)
                    __builtin_amdgcn_raw_buffer_store_b32(pk, rout, mine == OOB ? OOB : (mine - (odd ? 4u : 0u)) >> 1, 0, STR ? FGCN_STORE_AUX : 0);
                    const float k0 = off0 != OOB ? v0 : 0.f, k1 = off1 != OOB ? v1 : 0.f;
                    ssum[nu] += k0;
                    ssq[nu] = __builtin_fmaf(k0, k0, ssq[nu]);
                    ssum[nu] += k1;
                    ssq[nu] = __builtin_fmaf(k1, k1, ssq[nu]);
                }
            }
            continue;
        }
        // (row loop outside the unit loop: the 64-byte halves of a 128-byte line leave back to back -- fgcn_spatial_tile.hip's epilogue has
        // the measurement; each unit's sums keep their order)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int nu = 0; nu < NU; ++nu) {
                const unsigned off = (rowoff[mt][r] == OOB || coff[nu] == OOB) ? OOB : rowoff[mt][r] + coff[nu];
                float val = (NP == 2 ? acc[mt][nu][r] * un_a * un_w : acc[mt][nu][r]) + bv[nu];
                if constexpr (fep) {
                    val = __builtin_fmaf(acc[mt][nu][r], esc[nu], esh[nu]) + __builtin_fmaf(oldv[mt & 1][nu][r], rsc[nu], rsh[nu]);
                    val = fmaxf(val, 0.f);
                } else if constexpr (ldacc) {
                    val += oldv[mt & 1][nu][r];
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rout, ((FGCN_PROBE_HALO & 1) && val != 123.456f) ? OOB : off, 0, STR ? FGCN_STORE_AUX : 0);
                const float kept = off != OOB ? val : 0.f;
                if constexpr (bnb) {
                    const float dp = (mbits[mt & 1][nu][r] >> ((off >> 2) & 7u)) & 1u ? kept : 0.f;
                    ssum[nu] += dp;
                    ssq[nu] = __builtin_fmaf(dp, (av[mt & 1][nu][r] - bmean[nu]) * brstd[nu], ssq[nu]);
                } else {
                    ssum[nu] += kept;
                    ssq[nu] = __builtin_fmaf(kept, kept, ssq[nu]);   // (explicit: every instantiation sums the same way, bit for bit)
                }
            }
        }
        // (left to itself the scheduler requests all four row tiles' operands first and sums after the last store: 128 live values)
        if constexpr (bnb || ldacc) __builtin_amdgcn_sched_barrier(0);
    }
    if (p.stats && !((FGCN_PROBE_HALO & 32) && ssum[0] != 123.456f)) {
        __syncthreads();
        float* red = Ah;                                   // [which][wr][BN]
#pragma unroll
        for (int nu = 0; nu < NU; ++nu) {
            float sa = ssum[nu] + __shfl_xor(ssum[nu], 16);
            float sb = ssq[nu] + __shfl_xor(ssq[nu], 16);
            sa += __shfl_xor(sa, 32);
            sb += __shfl_xor(sb, 32);
            if (lane < 16) {
                red[(0 * WR + wr) * BN + wc * NT * 32 + nu * 16 + lane] = sa;
                red[(1 * WR + wr) * BN + wc * NT * 32 + nu * 16 + lane] = sb;
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, c = tid - which * BN;
            if (n0 + c < p.N) {
                float t = red[(which * WR + 0) * BN + c] + red[(which * WR + 1) * BN + c];
                if constexpr (WR == 4) t = (t + red[(which * WR + 2) * BN + c]) + red[(which * WR + 3) * BN + c];
                p.stats[((long long)bm * 2 + which) * p.N + n0 + c] = t;
                if constexpr (WR == 4) {                   // the partial rows of the 128-row tiling that this tiling does not fill
                    const long long extra = (long long)bm + p.tiles_m;
                    if (extra < p.stats_rows) p.stats[(extra * 2 + which) * p.N + n0 + c] = 0.f;
                }
            }
        }
    }
}

}  // namespace fgcn

using namespace fgcn;

// output rows per workgroup: 128, or 96 in the bf16 math modes when the 128-row tile's halo image (9 taps: 128 + 8 V rows, three
// planes of unpadded 64-byte rows) would not fit twice into LDS -- with the swizzled image that is V > 36, i.e. never (V <= 32):
// the 27- and 22-joint shapes of BASELINE configs 3 / 4 run the 128-row tile too (they took the 96-row one with 80-byte rows)
static int halo_tile_rows(int V) {
    const bool k32 = fgcn::math_mode() == FGCN_MATH_BF16 || fgcn::math_mode() == FGCN_MATH_BF16X3;
    return (k32 && (128 + 8 * V) * 64 * 3 > 80 * 1024) ? 96 : 128;
}

// 1 when fgcn_tconv_halo can emit the BatchNorm-backward sums in the current math mode / tuning (the 16x16x32 split-bf16 kernel)
extern "C" int fgcn_tconv_halo_bn_sums(void) {
    const int mm = fgcn::math_mode();
    return (mm == FGCN_MATH_BF16 || mm == FGCN_MATH_BF16X3) ? 1 : 0;
}

extern "C" int fgcn_tconv_halo_tiles(int B, int Th_out, int Th_in, int V) {
    const int Tv = Th_out > Th_in ? Th_out : Th_in;
    return (int)cdiv((long long)B * Tv * V, halo_tile_rows(V));
}

// one instantiation of the split kernel (LDS opt-in once per instantiation: not a stream operation, stays out of graph captures); the
// bfloat16-input form exists for the one-part tap kernel only
template <int NT, int KC, int NP, int EPI, bool FIN, int WR, bool STR>
static void halo_k32_launch(int in16, dim3 grid, size_t lds, hipStream_t s, const HaloP& p) {
    constexpr int max_lds = 32 * HALO_MAX_STAGE * XSB * 3;
    if constexpr (NP == 1 && !FIN && KC == 32 && EPI == 0) {
        if (in16 == 2) {             // bfloat16 in and out (the launcher has checked that this is the form it dispatches to)
            static bool opted162 = false;
            if (!opted162) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_x3k32_kernel<NT, KC, NP, EPI, FIN, WR, STR, 2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
                opted162 = true;
            }
            hipLaunchKernelGGL((conv_halo_x3k32_kernel<NT, KC, NP, EPI, FIN, WR, STR, 2>), grid, dim3(256), lds, s, p);
            return;
        }
    }
    if constexpr (NP == 1 && !FIN && KC == 32 && EPI != 4) {
        if (in16) {
            static bool opted16 = false;
            if (!opted16) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_x3k32_kernel<NT, KC, NP, EPI, FIN, WR, STR, 1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
                opted16 = true;
            }
            hipLaunchKernelGGL((conv_halo_x3k32_kernel<NT, KC, NP, EPI, FIN, WR, STR, 1>), grid, dim3(256), lds, s, p);
            return;
        }
    }
    static bool opted = false;
    if (!opted) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_x3k32_kernel<NT, KC, NP, EPI, FIN, WR, STR>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        opted = true;
    }
    hipLaunchKernelGGL((conv_halo_x3k32_kernel<NT, KC, NP, EPI, FIN, WR, STR>), grid, dim3(256), lds, s, p);
}

static int tconv_halo_impl(const float* in, float* out, const float* w4, const float* bias, float* stat_partials,
                           int B, int Th, int V, int K, int N, int ld_in, int ld_out,
                           int T_in_full, int in_s, int in_o, int Th_in,
                           int T_out_full, int out_s, int out_o,
                           int taps, int tb, int tc, int accumulate, const float* bn_a, const unsigned char* bn_mask,
                           const float* bn_vec, const float* fin_vec, const float* fin_res, float* fin_out,
                           unsigned char* fin_mask, unsigned* in_amax, void* stream, int in16, const float* ep_vec = nullptr,
                           const float* ep_res = nullptr, const float* ep_rvec = nullptr) {
    FGCN_REQUIRE(in && out && w4, FGCN_E_BADARG, "tconv_halo: null pointer");
    const bool fep = ep_vec != nullptr;
    FGCN_REQUIRE(!fep || ((fgcn::math_mode() == FGCN_MATH_BF16X3 || fgcn::math_mode() == FGCN_MATH_BF16) && !fgcn::f16x2_products() && !in16 &&
                          !accumulate && !bn_a && !stat_partials && !(fin_vec || fin_res || fin_out || fin_mask) && taps > 1 && out_s == 1 && out_o == 0 &&
                          T_out_full == Th && aligned16(ep_vec) && (!ep_res || aligned16(ep_res)) && (!ep_rvec || ep_res)),
                 FGCN_E_BADARG, "tconv_halo_bn_relu: the inference output stage needs the split kernel (bf16x3 products or bf16), the tap form, a "
                 "plain output view, no statistics / accumulation / fused input stage");
    FGCN_REQUIRE(!in16 || (fgcn::math_mode() == FGCN_MATH_BF16 && !(fin_vec || fin_res || fin_out || fin_mask) && !(taps == 1 && K % 64 == 0)),
                 FGCN_E_BADARG, "tconv_halo_h: a bfloat16 input needs math mode bf16, the tap form and no fused input stage");
    // (in16 == 2: the output is bfloat16 too -- the plain store epilogue, with or without the forward moments)
    FGCN_REQUIRE(in16 != 2 || (!accumulate && !bn_a && !fep && ld_out % 2 == 0), FGCN_E_BADARG,
                 "tconv_halo_t: a bfloat16 output is written by the plain store epilogue (no accumulation, no BatchNorm-backward sums)");
    const bool fin = fin_vec || fin_res || fin_out || fin_mask;
    FGCN_REQUIRE(!fin || !fgcn::f16x2_products(), FGCN_E_BADARG, "tconv_halo: the fused input stage is not built for the f16x2 products");
    FGCN_REQUIRE(!fin || (fin_vec && fin_res && fin_out && fin_mask && fgcn_tconv_halo_bn_sums() && !bn_a && !accumulate && taps > 1 &&
                          ld_in == K && in_s == 1 && in_o == 0 && Th_in == T_in_full && Th_in == Th && aligned16(fin_res) &&
                          aligned16(fin_out) && aligned16(fin_vec) && ((uintptr_t)fin_mask & 3u) == 0),
                 FGCN_E_BADARG, "tconv_halo: the fused input stage (BatchNorm + shortcut + ReLU while the image is staged) needs the "
                 "split-bf16 kernel, all four of (vec, shortcut, G, sign image), a plain contiguous input view (ld_in == K) and taps > 1");
    FGCN_REQUIRE(!bn_a || (bn_mask && bn_vec && stat_partials && fgcn_tconv_halo_bn_sums() && ld_out == N && N % 8 == 0 &&
                           out_s == 1 && out_o == 0 && T_out_full == Th),
                 FGCN_E_BADARG, "tconv_halo: BatchNorm-backward sums need the split-bf16 kernel, a contiguous plain output (ld_out == N, "
                 "N %% 8 == 0), the sign image, the coefficient vector and a partials buffer");
    FGCN_REQUIRE(B > 0 && Th > 0 && V > 0 && V <= FGCN_MAX_V && K > 0 && N > 0, FGCN_E_BADARG,
                 "tconv_halo: bad sizes B=%d Th=%d V=%d K=%d N=%d", B, Th, V, K, N);
    FGCN_REQUIRE(K % 32 == 0 && N % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0 && ld_in >= K && ld_out >= N, FGCN_E_ALIGN,
                 "tconv_halo: K must be a multiple of 32, N and strides multiples of 4 (K=%d N=%d ld_in=%d ld_out=%d)", K,
                 N, ld_in, ld_out);
    FGCN_REQUIRE(aligned16(in) && aligned16(w4), FGCN_E_ALIGN, "tconv_halo: 16-byte alignment");
    FGCN_REQUIRE(taps >= 1 && taps <= 16 && (tb == 1 || tb == -1), FGCN_E_BADARG, "tconv_halo: taps=%d tb=%d", taps, tb);
    FGCN_REQUIRE(in_s >= 1 && out_s >= 1 && in_o >= 0 && out_o >= 0 && Th_in > 0, FGCN_E_BADARG,
                 "tconv_halo: bad frame views");
    FGCN_REQUIRE((long long)(Th_in - 1) * in_s + in_o < T_in_full && (long long)(Th - 1) * out_s + out_o < T_out_full,
                 FGCN_E_BADARG, "tconv_halo: frame view exceeds the tensor (Th=%d Th_in=%d)", Th, Th_in);
    const int mm = fgcn::math_mode();
    // FGCN_MATH_BF16X3 / FGCN_MATH_BF16: w4 is the split form (fgcn_pack_split3: three bf16 parts of [tap][K/8][N][8]) = 6 bytes
    // per weight; the bf16 mode reads part 0 only (the round-to-nearest-even bf16 of the weight)
    const long long in_bytes = (long long)B * T_in_full * V * ld_in * (in16 ? 2 : 4);
    const bool two = fgcn::f16x2_products();                                                // FGCN_PACK_SPLIT2H weights: two f16 parts
    const long long w_bytes = (long long)taps * K * N * (mm != FGCN_MATH_F32 ? (two ? 4 : 6) : 4);   // split form in both bf16 modes
    const long long out_bytes = (long long)B * T_out_full * V * ld_out * (in16 == 2 ? 2 : 4);
    FGCN_REQUIRE(in_bytes < 0x7FFF0000ll && w_bytes < 0x7FFF0000ll && out_bytes < 0x7FFF0000ll, FGCN_E_BADARG,
                 "tconv_halo: tensors must be smaller than 2 GiB (32-bit buffer offsets)");
    HaloP p;
    p.in = in; p.out = out; p.w4 = w4; p.bias = bias; p.stats = stat_partials;
    p.in_bytes = (unsigned)in_bytes; p.w_bytes = (unsigned)w_bytes; p.out_bytes = (unsigned)out_bytes;
    p.w_plane_bytes = (unsigned)((long long)taps * K * N * 2);
    p.Tv = Th > Th_in ? Th : Th_in;
    p.Mv = (long long)B * p.Tv * V;
    p.V = V; p.K = K; p.N = N; p.ld_in = ld_in; p.ld_out = ld_out;
    p.dV = make_fastdiv((unsigned)V); p.dTvV = make_fastdiv((unsigned)(p.Tv * V)); p.dTv = make_fastdiv((unsigned)p.Tv);
    FGCN_REQUIRE(mm == FGCN_MATH_F32 || p.Mv < (1ll << 29), FGCN_E_BADARG, "tconv_halo: too many rows for the split kernels' index arithmetic (< 2^29)");
    p.T_in_full = T_in_full; p.in_s = in_s; p.in_o = in_o; p.Th_in = Th_in;
    p.T_out_full = T_out_full; p.out_s = out_s; p.out_o = out_o; p.Th_out = Th;
    p.taps = taps; p.tb = tb; p.tc = tc; p.accumulate = accumulate;
    p.bn_a = bn_a; p.bn_mask = bn_mask; p.bn_vec = bn_vec;
    p.fin_vec = fin_vec; p.fin_res = fin_res; p.fin_out = fin_out; p.fin_mask = fin_mask;
    p.ep_vec = ep_vec; p.ep_res = ep_res; p.ep_rvec = ep_rvec;
    p.in_amax = fgcn::f16x2_products() ? in_amax : nullptr;
    const int d0 = tc, d1 = (taps - 1) * tb + tc;
    p.dmin = d0 < d1 ? d0 : d1;
    const int dmax = d0 < d1 ? d1 : d0;
    int bmr = halo_tile_rows(V);
    p.stats_rows = (int)cdiv(p.Mv, bmr);
    // split kernels, <= 64 output columns, tap form: waves 4 x 1 over 192-row tiles (see the kernel; key 7 bit 3 keeps the 2 x 2 form)
    // the same arrangement for ONE 128-column tile (64 < N <= 128): a wave owns 48 rows x 128 columns, every image fragment feeds eight
    // units: -5 % (bf16x3) / -9 % (f16x2) at 128 channels, nothing at 256 (two column tiles; key 7 bit 4 forces it there, bit 5 switches it
    // off); epilogue forms 0 and 3 only -- the BatchNorm-sums epilogue would spill at 255 registers
    // (FGCN_MATH_BF16: the one-part instantiation of this form runs with three workgroups per CU and spills -- 136 to 424 bytes of scratch -- and
    // the matrix work it saves LDS reads for is a sixth: the 2 x 2 form measured 29.65 -> 28.47 ms per bf16 step, profiles/r06_ab_bf16_half_activations.txt)
    const bool wide128 = ((fgcn::tuning(7) & 16) || (N <= 128 && !(fgcn::tuning(7) & 32) && mm != FGCN_MATH_BF16)) && N > 64 && !bn_a && !fep;   // (EPI 4 spills in that form)
    // (FGCN_MATH_BF16: the 4 x 1 arrangements lose to the 2 x 2 form at every width -- 64 columns: 28.75 -> 28.55 ms per step -- key 7 bit 4 brings them back)
    const bool wide_rows = (mm == FGCN_MATH_BF16X3 || (mm == FGCN_MATH_BF16 && (fgcn::tuning(7) & 16))) && ((N <= 64 && N > 32) || wide128) && !(taps == 1 && K % 64 == 0) &&
                           !(fin_vec || fin_res || fin_out || fin_mask) && !(fgcn::tuning(7) & 8) &&
                           192 + (dmax - p.dmin) * V <= 32 * HALO_MAX_STAGE && (size_t)(192 + (dmax - p.dmin) * V) * 64 * 3 + 16 <= 80 * 1024 &&
                           (cdiv(p.Mv, 192) >= 1536 || (fgcn::tuning(7) & 64));   // (three rounds of 512 workgroups: below, the coarser tiling quantises worse -- 8-clip step +0.07 ms)
    if (wide_rows) bmr = 192;
    p.halo_rows = bmr + (dmax - p.dmin) * V;
    FGCN_REQUIRE(p.halo_rows <= 32 * HALO_MAX_STAGE, FGCN_E_BADARG, "tconv_halo: halo of %d rows too large", p.halo_rows);
    const size_t lds = mm == FGCN_MATH_BF16X3 ? (size_t)p.halo_rows * XSB * 3 : (size_t)p.halo_rows * HAS * sizeof(float);
    const long long tiles = cdiv(p.Mv, bmr);
    FGCN_REQUIRE(p.Mv < (1ll << 31) - 4096, FGCN_E_BADARG, "tconv_halo: too many rows (32-bit row indices)");
    hipStream_t s = (hipStream_t)stream;
    static bool lds_opt_in = false;  // once per process (not a stream operation: keep it out of graph captures)
    if (!lds_opt_in) {               // V > 25 needs more than the default dynamic-LDS limit (gfx950: 160 KiB per CU)
        const int max_lds = 32 * HALO_MAX_STAGE * XSB * 3;   // the larger of the two image forms
#define FGCN_HALO_ATTR(NT_, MB_)                                                                            \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<NT_, MB_, 0>),              \
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);                       \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<NT_, MB_, 1>),              \
                              hipFuncAttributeMaxDynamicSharedMemorySize, max_lds)
        FGCN_HALO_ATTR(2, 2); FGCN_HALO_ATTR(2, 3); FGCN_HALO_ATTR(4, 2); FGCN_HALO_ATTR(4, 3);
#undef FGCN_HALO_ATTR
        lds_opt_in = true;
    }
    const bool three = fgcn::tuning(4) == 0;   // 3 workgroups per CU measured faster (64 channels: 0.73 -> 0.63 ms)
    p.tiles_m = (int)tiles;
    p.tiles_n = (int)cdiv(N, N <= 64 ? 64 : 128);
    dim3 grid((unsigned)tiles, (unsigned)p.tiles_n);
    p.per_xcd = 0;
    if (mm == FGCN_MATH_BF16X3 || mm == FGCN_MATH_BF16) {
        // split-bf16 kernel (16x16x32 MFMAs); FGCN_MATH_BF16: the same kernel with one bf16 part
        const bool one = mm == FGCN_MATH_BF16;
        const bool pw = taps == 1 && K % 64 == 0;
        // XCD-aware workgroup order (key 5 bit 5 switches it off): measured on MI355X at 256 channels, B = 128: FETCH_SIZE per launch
        // 765 -> 332 MiB (HBM-side traffic 3.7x -> 1.9x the algorithmic bytes), 1.335 -> 1.310 ms, the step 65.3 -> 64.6 ms
        if (!(fgcn::tuning(5) & 32) && tiles * p.tiles_n < (1ll << 30)) {
            p.per_xcd = (int)cdiv(tiles * p.tiles_n, 8);
            grid = dim3((unsigned)(p.per_xcd * 8));
        }
        // image rows: 128 (1x1, 64-channel chunks) / 64 bytes per bf16 part, unpadded (swizzled); the epilogue's 2 KB of partial sums fit
        const int np = one ? 1 : (two ? 2 : 3);
        const size_t lds_k = (pw ? (size_t)bmr * 128 * np : (size_t)p.halo_rows * 64 * np) + 16;
        FGCN_REQUIRE(bmr == 128 || wide_rows, FGCN_E_BADARG, "tconv_halo: the split kernels run the 128-row tile (V <= %d)", FGCN_MAX_V);
        FGCN_REQUIRE(!(bn_a && accumulate), FGCN_E_BADARG, "tconv_halo: BatchNorm-backward sums of an accumulating call are not built");
        const int epi = fep ? 4 : (bn_a ? 2 : (accumulate ? 3 : 0));     // epilogue form (compile time, see the kernel)
        // the bytes this call writes (a bfloat16 output: 32-byte pieces, stored plainly unless tuning key 25 bit 0 asks for streamed stores)
        const bool stream_k = in16 != 2 ? fgcn::stream_out((long long)B * Th * V * N * 4) : ((fgcn::tuning(25) & 1) && fgcn::stream_out((long long)B * Th * V * N * 2));
#define FGCN_K32_GO7(NT_, KC_, NP_, EPI_, FIN_, WR_, STR_) halo_k32_launch<NT_, KC_, NP_, EPI_, FIN_, WR_, STR_>(in16, grid, lds_k, s, p)
#define FGCN_K32_GO6(NT_, KC_, NP_, EPI_, FIN_, WR_)                                                                     \
    do {                                                                                                                 \
        if (EPI_ != 3 && stream_k) FGCN_K32_GO7(NT_, KC_, NP_, EPI_, FIN_, WR_, (EPI_ != 3));                            \
        else FGCN_K32_GO7(NT_, KC_, NP_, EPI_, FIN_, WR_, false);                                                        \
    } while (0)
#define FGCN_K32_GO(NT_, KC_, NP_, EPI_, FIN_) FGCN_K32_GO6(NT_, KC_, NP_, EPI_, FIN_, 2)
#define FGCN_K32_WIDE_NP(EPI_)                                                                                           \
    do {                                                                                                                 \
        if (one) FGCN_K32_GO6(2, 32, 1, EPI_, false, 4);                                                                 \
        else if (two) FGCN_K32_GO6(2, 32, 2, EPI_, false, 4);                                                            \
        else FGCN_K32_GO6(2, 32, 3, EPI_, false, 4);                                                                     \
    } while (0)
#define FGCN_K32_NP(NT_, KC_, EPI_)                                                                                      \
    do {                                                                                                                 \
        if (one) FGCN_K32_GO(NT_, KC_, 1, EPI_, false);                                                                  \
        else if (two) FGCN_K32_GO(NT_, KC_, 2, EPI_, false);                                                             \
        else FGCN_K32_GO(NT_, KC_, 3, EPI_, false);                                                                      \
    } while (0)
#define FGCN_K32_NP13(NT_, KC_, EPI_)      /* (the inference stage: three bf16 parts or one -- not the f16x2 products) */ \
    do {                                                                                                                 \
        if (one) FGCN_K32_GO(NT_, KC_, 1, EPI_, false);                                                                  \
        else FGCN_K32_GO(NT_, KC_, 3, EPI_, false);                                                                      \
    } while (0)
#define FGCN_K32_LAUNCH(NT_, KC_)                                                                                        \
    do {                                                                                                                 \
        if (epi == 0) FGCN_K32_NP(NT_, KC_, 0);                                                                          \
        else if (epi == 3) FGCN_K32_NP(NT_, KC_, 3);                                                                     \
        else if (epi == 4 && KC_ == 32) FGCN_K32_NP13(NT_, 32, 4);                                                       \
        else if (epi == 2 && KC_ == 32) FGCN_K32_NP(NT_, 32, 2);                                                         \
        else return fgcn::fail(FGCN_E_BADARG, "tconv_halo: BatchNorm-backward sums are built for the tap kernel only"); \
    } while (0)
        if (fin) {
            FGCN_REQUIRE(!pw, FGCN_E_BADARG, "tconv_halo: the fused input stage runs on the 128-row tap tile (V <= 32)");
            if (N <= 64) {
                if (one) FGCN_K32_GO(1, 32, 1, 0, true);
                else FGCN_K32_GO(1, 32, 3, 0, true);
            } else {
                if (one) FGCN_K32_GO(2, 32, 1, 0, true);
                else FGCN_K32_GO(2, 32, 3, 0, true);
            }
        } else if (pw) {
            if (N <= 64) FGCN_K32_LAUNCH(1, 64);
            else FGCN_K32_LAUNCH(2, 64);
        } else {
            if (wide_rows && epi == 4) {                 // (N <= 64: the 128-column 4 x 1 form is excluded for this epilogue above)
                if (one) FGCN_K32_GO6(2, 32, 1, 4, false, 4); else FGCN_K32_GO6(2, 32, 3, 4, false, 4);
            } else if (wide_rows && N > 64) {
                if (epi == 0) { if (one) FGCN_K32_GO6(4, 32, 1, 0, false, 4); else if (two) FGCN_K32_GO6(4, 32, 2, 0, false, 4); else FGCN_K32_GO6(4, 32, 3, 0, false, 4); }
                else { if (one) FGCN_K32_GO6(4, 32, 1, 3, false, 4); else if (two) FGCN_K32_GO6(4, 32, 2, 3, false, 4); else FGCN_K32_GO6(4, 32, 3, 3, false, 4); }
            } else if (wide_rows) {
                if (epi == 0) FGCN_K32_WIDE_NP(0);
                else if (epi == 3) FGCN_K32_WIDE_NP(3);
                else FGCN_K32_WIDE_NP(2);
            } else if (N <= 64) FGCN_K32_LAUNCH(1, 32);
            else FGCN_K32_LAUNCH(2, 32);
        }
#undef FGCN_K32_LAUNCH
#undef FGCN_K32_NP13
#undef FGCN_K32_NP
#undef FGCN_K32_GO
#undef FGCN_K32_GO6
#undef FGCN_K32_GO7
#undef FGCN_K32_WIDE_NP
        return launch_status("tconv_halo");
    }
    if ((fgcn::tuning(5) & 2) && tiles * p.tiles_n < (1ll << 30)) {   // measured neutral: off
        p.per_xcd = (int)cdiv(tiles * p.tiles_n, 8);
        grid = dim3((unsigned)(p.per_xcd * 8));
    }
#define FGCN_HALO_LAUNCH(NT_, MB_)                                                                           \
    do {                                                                                                     \
        if (mm == FGCN_MATH_BF16) hipLaunchKernelGGL((conv_halo_kernel<NT_, MB_, 1>), grid, dim3(256), lds, s, p); \
        else hipLaunchKernelGGL((conv_halo_kernel<NT_, MB_, 0>), grid, dim3(256), lds, s, p);                \
    } while (0)
    if (N <= 64) {
        if (three) FGCN_HALO_LAUNCH(2, 3);
        else FGCN_HALO_LAUNCH(2, 2);
    } else {
        if (three) FGCN_HALO_LAUNCH(4, 3);
        else FGCN_HALO_LAUNCH(4, 2);
    }
#undef FGCN_HALO_LAUNCH
    return launch_status("tconv_halo");
}

extern "C" int fgcn_tconv_halo(const float* in, float* out, const float* w4, const float* bias, float* stat_partials,
                               int B, int Th, int V, int K, int N, int ld_in, int ld_out,
                               int T_in_full, int in_s, int in_o, int Th_in,
                               int T_out_full, int out_s, int out_o,
                               int taps, int tb, int tc, int accumulate, const float* bn_a, const unsigned char* bn_mask,
                               const float* bn_vec, const float* fin_vec, const float* fin_res, float* fin_out,
                               unsigned char* fin_mask, unsigned* in_amax, void* stream) {
    return tconv_halo_impl(in, out, w4, bias, stat_partials, B, Th, V, K, N, ld_in, ld_out, T_in_full, in_s, in_o, Th_in, T_out_full, out_s, out_o,
                           taps, tb, tc, accumulate, bn_a, bn_mask, bn_vec, fin_vec, fin_res, fin_out, fin_mask, in_amax, stream, false);
}

// The same convolution with a BFLOAT16 input tensor (math mode bf16 only; ld_in in elements): half-precision storage of the conv's input,
// written by fgcn_bn_act_h (G) / fgcn_bn_act_bwd_apply_h (dU).  Bit-identical to fgcn_tconv_halo on the f32 tensor those kernels would
// have written (the bf16 kernel rounds its input to bfloat16, to nearest even, as it stages it).
extern "C" int fgcn_tconv_halo_h(const unsigned short* in_h, float* out, const float* w4, const float* bias, float* stat_partials,
                                 int B, int Th, int V, int K, int N, int ld_in, int ld_out,
                                 int T_in_full, int in_s, int in_o, int Th_in,
                                 int T_out_full, int out_s, int out_o,
                                 int taps, int tb, int tc, int accumulate, const float* bn_a, const unsigned char* bn_mask,
                                 const float* bn_vec, void* stream) {
    return tconv_halo_impl(reinterpret_cast<const float*>(in_h), out, w4, bias, stat_partials, B, Th, V, K, N, ld_in, ld_out, T_in_full, in_s, in_o,
                           Th_in, T_out_full, out_s, out_o, taps, tb, tc, accumulate, bn_a, bn_mask, bn_vec, nullptr, nullptr, nullptr, nullptr,
                           nullptr, stream, 1);
}

// typed form: half_mask bit 0 = `in` is bfloat16, bit 1 = `out` is (math mode bf16; a bfloat16 output needs a bfloat16 input: masks 0, 1, 3).
// The plain store epilogue (no accumulation, no BatchNorm-backward sums); stat_partials: the forward moments of the float32 accumulators.
extern "C" int fgcn_tconv_halo_t(const void* in, void* out, const float* w4, const float* bias, float* stat_partials,
                                 int B, int Th, int V, int K, int N, int ld_in, int ld_out,
                                 int T_in_full, int in_s, int in_o, int Th_in,
                                 int T_out_full, int out_s, int out_o,
                                 int taps, int tb, int tc, int half_mask, void* stream) {
    FGCN_REQUIRE(half_mask == 0 || half_mask == 1 || half_mask == 3, FGCN_E_BADARG, "tconv_halo_t: half_mask=%d (0, 1 or 3)", half_mask);
    return tconv_halo_impl(static_cast<const float*>(in), static_cast<float*>(out), w4, bias, stat_partials, B, Th, V, K, N, ld_in, ld_out, T_in_full,
                           in_s, in_o, Th_in, T_out_full, out_s, out_o, taps, tb, tc, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, stream, half_mask == 3 ? 2 : half_mask);
}

// North-star kernel 2 as the north star states it, for INFERENCE: the (taps x 1) temporal convolution (stride 1) with the block's output
// stage in its epilogue -- out = relu(BN(conv(in) + bias) + shortcut), agcn.py:49-51,134-136 with the BatchNorm's running statistics folded
// into a per-channel scale / shift (bn_vec = fgcn_bn_eval_coeffs).  res: the shortcut operand laid out like out (x of an identity block, the
// residual conv's output with its own res_vec) or NULL (the first block).  Split kernel only (FGCN_MATH_BF16X3 with the bf16x3 products, or
// FGCN_MATH_BF16).  Nothing is kept for a backward: a training step runs fgcn_tconv_halo (statistics in the epilogue) + fgcn_bn_act, because
// train-mode BatchNorm needs the statistics of the WHOLE batch before it can be applied.
extern "C" int fgcn_tconv_halo_bn_relu(const float* in, float* out, const float* w4, const float* bias, const float* bn_vec,
                                       const float* res, const float* res_vec, int B, int T, int V, int K, int N, int ld_in, int ld_out,
                                       int taps, int tb, int tc, void* stream) {
    FGCN_REQUIRE(bn_vec, FGCN_E_BADARG, "tconv_halo_bn_relu: the BatchNorm vector is required");
    return tconv_halo_impl(in, out, w4, bias, nullptr, B, T, V, K, N, ld_in, ld_out, T, 1, 0, T, T, 1, 0, taps, tb, tc, 0, nullptr, nullptr, nullptr,
                           nullptr, nullptr, nullptr, nullptr, nullptr, stream, false, bn_vec, res, res_vec);
}
