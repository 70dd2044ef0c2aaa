// Row GEMMs on the f32 MFMA (v_mfma_f32_32x32x2_f32): every 1x1 / (kt x 1) convolution of the AGCN block,
// forward, data gradient and weight gradient, as an implicit GEMM over the flattened (n, t, v) rows of a
// channels-last activation.  See include/fgcn.h for the contracts and the reference lines replaced.
//
// rows_gemm : M = rows (B*T_out*V), N = out channels, K = taps * in channels.
//   workgroup = 4 waves, tile 128 rows x (32*NT) channels; wave w owns rows [32w, 32w+32) x all tile channels
//   (NT accumulators of 32x32).  K is walked in 32-wide chunks staged through LDS; the next chunk's global
//   loads are issued before the current chunk's MFMAs (register double buffer).  f32 MFMA issues one
//   32x32x2 per 64 cycles per SIMD, so LDS/L2 bandwidth needs are tiny; what matters is keeping the four
//   SIMDs issuing: several workgroups per CU (<= 36 KiB LDS, ~100 VGPRs) overlap each other's staging.
//   The temporal taps shift the source row by (tap offset)*V rows inside the same sample, zero outside [0,T).
//
// rows_wgrad: dW[tap][k][n] = sum_rows a[src(row,tap)][k] * g[row][n]: both operands are K(=row)-major, so
//   A/B fragments are single LDS dwords with the channel on the lane.  64x64 output tile per workgroup
//   (one 32x32 accumulator per wave), rows split across blockIdx.z into deterministic partial slabs.
#include "fgcn_common.hpp"
#include <type_traits>

namespace fgcn {

struct RowsGemmP {
    const float* in;
    float* out;
    const float* w;
    const float* bias;
    float* stats;
    long long M;
    int T_in, T_out, V, K, N, ld_in, ld_out;
    int taps, ta, tb, tc, td;
    int accumulate;
    long long in_elems;  // B * T_in * V * ld_in
    unsigned w_bytes;
    int tiles_m, tiles_n, per_xcd;  // per_xcd > 0: 1-D grid in XCD-aware order (column tiles of a row tile share an L2)
    long long in_bs, out_bs, w_bs;  // fgcn_rows_gemm_batched: element strides of blockIdx.z's problem (0 otherwise)
    int stream;                     // non-temporal output stores (fgcn_common.hpp, stream_out)
    FastDiv dTV, dV;                // row -> (sample, frame, joint) by multiply-shift (rows < 2^29: the launcher checks)
    int inner;                      // fgcn_rows_gemm_batched2: blockIdx.z = outer * inner + i; problem i of an outer group adds the *_bs2 strides
    long long in_bs2, out_bs2, w_bs2;
};

// MT x NT 32x32 accumulators per wave; the four waves stack along the rows: tile = (128*MT) rows x (32*NT) channels.
// DB = double-buffered LDS (one barrier per K chunk instead of two, at twice the LDS footprint).
// BF: FGCN_MATH_BF16 (one bf16 MFMA per four f32 MFMAs, operands rounded as the fragments are read)
// IO (BF only; the typed entry point fgcn_rows_gemm_t, half-precision activation storage): bit 0 = `in` is a BFLOAT16 tensor (its values are
// widened into the same float32 LDS image: the fragments round them back to the same 16 bits), bit 1 = `out` is (the float32 result rounded
// once; BatchNorm sums of the float32 values; not with accumulation; adjacent lanes pair their columns into dword stores)
template <int MT, int NT, bool DB, bool BF, int IO = 0>
__global__ __launch_bounds__(256, (MT == 1 && !DB) ? 3 : 2) void rows_gemm_kernel(RowsGemmP p) {
    constexpr int BM = 128 * MT, BK = 32, BN = 32 * NT, AS = BK + 4, NBUF = DB ? 2 : 1;
    static_assert(IO == 0 || BF, "bfloat16 tensors: math mode bf16");
    constexpr bool IN16 = (IO & 1) != 0, OUT16 = (IO & 2) != 0;
    constexpr unsigned IS = IN16 ? 2u : 4u;            // bytes per input element
    {                                                  // batched forms: one independent problem per blockIdx.z
        const int zo = p.inner > 1 ? (int)blockIdx.z / p.inner : (int)blockIdx.z;
        const int zi = (int)blockIdx.z - zo * (p.inner > 1 ? p.inner : 1);
        p.in += (long long)zo * p.in_bs + (long long)zi * p.in_bs2;
        p.out += (long long)zo * p.out_bs + (long long)zi * p.out_bs2;
        p.w += (long long)zo * p.w_bs + (long long)zi * p.w_bs2;
    }
    constexpr int AR = 4 * MT;                         // A-tile rows staged per thread
    __shared__ __attribute__((aligned(16))) float As[NBUF * BM * AS];
    __shared__ __attribute__((aligned(16))) float Bs[NBUF * BK * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Consecutive workgroup ids go round-robin over the 8 XCDs: id b takes virtual tile (b % 8) * per_xcd + b / 8, column
    // tile fastest, so the workgroups that re-read one A row tile run on one XCD (one L2).  Speed only.
    int bm, bn;
    if (p.per_xcd > 0) {
        const int vid = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
        if (vid >= p.tiles_m * p.tiles_n) return;
        bm = vid / p.tiles_n;
        bn = vid - bm * p.tiles_n;
    } else if (p.per_xcd < 0) {   // column tile fastest (2-D grid transposed): the column tiles of a row tile run together
        bm = blockIdx.y;
        bn = blockIdx.x;
    } else {
        bm = blockIdx.x;
        bn = blockIdx.y;
    }
    const long long m0 = (long long)bm * BM;
    const int n0 = bn * BN;
    const int k4 = (tid & 7) * 4;

    // All global accesses are buffer instructions with 32-bit offsets relative to a per-workgroup base (no tensor-size
    // limit) and the out-of-range sentinel for absent rows / taps / channels: no exec-mask branches around memory
    // instructions, so hipcc keeps counted vmcnt waits (guarded global loads / stores made it drain to vmcnt(0) after
    // every store of the epilogue: 64 serialised store round trips per tile).
    constexpr unsigned OOB = 0x80000000u;
    const int TV = p.T_out * p.V;
    const int n_first = (int)fastdiv((unsigned)m0, p.dTV);    // 32-bit decode: host guarantees M < 2^29
    const long long in_base = (long long)n_first * p.T_in * p.V * p.ld_in;
    const long long in_left = (p.in_elems - in_base) * IS;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const char*>(p.in) + in_base * IS), 0, (unsigned)(in_left < 0x7FFFFFFFll ? in_left : 0x7FFFFFFFll), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    // the A-tile rows this thread stages: r = (tid >> 3) + 32*i
    unsigned roff[AR];
    int rto[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const long long m = m0 + (tid >> 3) + 32 * i;
        const unsigned mu = (unsigned)(m < p.M ? m : m0);
        const int n = (int)fastdiv(mu, p.dTV);
        const int rem = (int)(mu - (unsigned)n * (unsigned)TV);
        const int to = (int)fastdiv((unsigned)rem, p.dV);
        const int v = rem - to * p.V;
        roff[i] = (unsigned)(((n - n_first) * p.T_in * p.V + v) * p.ld_in) * IS;
        rto[i] = m < p.M ? to : -1;
    }

    const int KC = (p.K + BK - 1) / BK;
    const int S = p.taps * KC;
    f32x4 areg[AR], breg[NT];
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero16();

    auto load_stage = [&](int s) {
        const int tap = s / KC;
        const int kc = (s - tap * KC) * BK;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int ti = rto[i] >= 0 ? tmap_src(rto[i], tap, p.ta, p.tb, p.tc, p.td, p.T_in) : -1;
            const int k = kc + k4;   // K % 4 == 0: a 16-byte group is either whole or absent
            const unsigned off = (ti >= 0 && k < p.K) ? roff[i] + (unsigned)(ti * p.V * p.ld_in + k) * IS : OOB;
            if constexpr (IN16) areg[i] = unpack_bf16x4(__builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rin, off, 0, 0)));
            else areg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int idx = tid + 256 * i;
            const int kk = idx / (BN / 4), n4 = idx - kk * (BN / 4);
            const int k = kc + kk, n = n0 + 4 * n4;
            const unsigned off = (k < p.K && n < p.N) ? (unsigned)((tap * p.K + k) * p.N + n) * 4u : OOB;
            breg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0));
        }
    };

    load_stage(0);
    for (int s = 0; s < S; ++s) {
        float* Ab = As + (DB ? (s & 1) * BM * AS : 0);
        float* Bb = Bs + (DB ? (s & 1) * BK * BN : 0);
        if (!DB) __syncthreads();  // previous chunk's LDS reads are done (DB: the other buffer is being read)
#pragma unroll
        for (int i = 0; i < AR; ++i)
            *reinterpret_cast<f32x4*>(&Ab[((tid >> 3) + 32 * i) * AS + k4]) = areg[i];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int idx = tid + 256 * i;
            const int kk = idx / (BN / 4), n4 = idx - kk * (BN / 4);
            *reinterpret_cast<f32x4*>(&Bb[kk * BN + 4 * n4]) = breg[i];
        }
        __syncthreads();
        const int kc = (s % KC) * BK;
        if (s + 1 < S) load_stage(s + 1);  // in flight while this chunk's MFMAs run
        const int kleft = p.K - kc;
        const int nq = kleft >= BK ? BK / 8 : (kleft + 7) / 8;
        const float* arow = &Ab[(wave * 32 * MT + (lane & 31)) * AS + 4 * (lane >> 5)];
        const float* bcol = &Bb[(4 * (lane >> 5)) * BN + (lane & 31)];
        for (int q = 0; q < nq; ++q) {
            // lane half h holds k = 8q + 4h + e (e = 0..3): any k permutation is fine as long as A and B agree
            f32x4 av[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const f32x4*>(arow + mt * 32 * AS + 8 * q);
            if constexpr (BF) {
                s16x4 ap[MT], bp[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) ap[mt] = pack_bf16(av[mt]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bp[nt] = pack_bf16(bcol[(8 * q + 0) * BN + nt * 32], bcol[(8 * q + 1) * BN + nt * 32],
                                       bcol[(8 * q + 2) * BN + nt * 32], bcol[(8 * q + 3) * BN + nt * 32]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_bf16(ap[mt], bp[nt], acc[mt][nt]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float bv[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv[nt] = bcol[(8 * q + e) * BN + nt * 32];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma32(av[mt][e], bv[nt], acc[mt][nt]);
                }
            }
        }
    }

    // ---- epilogue: bias, optional accumulate, store, optional BatchNorm partial statistics -----------------
    // The output buffer covers exactly this tile's rows that exist (rows >= M fall outside and are dropped).
    const long long rows_left = p.M - m0;
    const unsigned tile_rows = (unsigned)(rows_left < BM ? rows_left : BM);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<char*>(p.out) + m0 * p.ld_out * (OUT16 ? 2 : 4)), 0, tile_rows * (unsigned)p.ld_out * (OUT16 ? 2u : 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.bias ? p.bias : p.w), 0, p.bias ? (unsigned)p.N * 4u : 0u, 0x00020000);
    float ssum[NT], ssq[NT];
    // One branch around the whole epilogue (an `if (p.accumulate)` around the loads inside the unrolled loops made hipcc drain vmcnt(0)
    // -- every earlier store's write acknowledgement -- at each join, on the plain path too).  Accumulating form: the old values of
    // group g + 1 are requested BEFORE the stores of group g, so a wait for them never includes those stores (vmcnt counts in issue order).
    const unsigned rstep = (unsigned)p.ld_out * 4u;
    auto epilogue = [&](auto mode_c) {                     // 0 store, 1 accumulate, 2 store non-temporally (fgcn_common.hpp, stream_out)
        constexpr bool ACC = decltype(mode_c)::value == 1;
        constexpr int AUX = decltype(mode_c)::value == 2 ? FGCN_STORE_AUX : 0;
        constexpr int NG = NT * MT;                        // groups of 16 values: g = nt * MT + mt
        float bvs[NT];
        unsigned off0[NG];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            ssum[nt] = 0.f;
            ssq[nt] = 0.f;
            const int col = n0 + nt * 32 + (lane & 31);
            const bool cok = col < p.N;
            bvs[nt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, cok ? (unsigned)col * 4u : OOB, 0, 0));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                // register r holds row rel0 + (r & 3) + 8 (r >> 2); rows past the tile's last row are outside the buffer
                const unsigned rel0 = (unsigned)(wave * 32 * MT + mt * 32 + 4 * (lane >> 5));
                off0[nt * MT + mt] = cok ? (rel0 * (unsigned)p.ld_out + (unsigned)col) * 4u : OOB;
            }
        }
        float old[ACC ? 2 : 1][16];
        auto load_old = [&](int g, float (&o)[16]) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                o[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                     rout, off0[g] + (unsigned)((r & 3) + 8 * (r >> 2)) * rstep, 0, 0));
        };
        if constexpr (ACC) load_old(0, old[0]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int nt = g / MT, mt = g - nt * MT;
            if constexpr (ACC) {
                if (g + 1 < NG) load_old(g + 1, old[(g + 1) & 1]);
            }
            const unsigned rel0 = (unsigned)(wave * 32 * MT + mt * 32 + 4 * (lane >> 5));
            if constexpr (OUT16 && !ACC) {
                // two rows at a time: the even lane of a pair stores columns (c, c + 1) of row dr(r) as one dword, the odd lane those of row dr(r + 1)
                const bool odd = lane & 1;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const unsigned dr0 = (unsigned)((r & 3) + 8 * (r >> 2));
                    const float v0 = acc[mt][nt][r] + bvs[nt], v1 = acc[mt][nt][r + 1] + bvs[nt];
                    const float other = lane_xor1(odd ? v0 : v1);
                    const unsigned pk = odd ? pack_bf16x2(other, v1) : pack_bf16x2(v0, other);
                    // (off0: the float32-form byte offset of (row rel0, this lane's column): halves, minus the odd lane's column, plus its row)
                    const unsigned off = off0[g] == OOB ? OOB : ((off0[g] - (odd ? 4u : 0u)) >> 1) + (dr0 + (odd ? 1u : 0u)) * (rstep >> 1);
                    __builtin_amdgcn_raw_buffer_store_b32(pk, rout, off, 0, AUX);
                    const float k0 = (off0[g] != OOB && rel0 + dr0 < tile_rows) ? v0 : 0.f, k1 = (off0[g] != OOB && rel0 + dr0 + 1 < tile_rows) ? v1 : 0.f;
                    ssum[nt] += k0;
                    ssq[nt] += k0 * k0;
                    ssum[nt] += k1;
                    ssq[nt] += k1 * k1;
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned dr = (unsigned)((r & 3) + 8 * (r >> 2));
                float val = acc[mt][nt][r] + bvs[nt];
                if constexpr (ACC) val += old[g & 1][r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rout, off0[g] + dr * rstep, 0, AUX);
                const float kept = (off0[g] != OOB && rel0 + dr < tile_rows) ? val : 0.f;
                ssum[nt] += kept;
                ssq[nt] += kept * kept;
            }
            if constexpr (ACC) __builtin_amdgcn_sched_barrier(0);   // (keep the request / store order as written)
        }
    };
    if (p.accumulate) epilogue(std::integral_constant<int, 1>{});          // (kernel-uniform)
    else if (p.stream) epilogue(std::integral_constant<int, 2>{});
    else epilogue(std::integral_constant<int, 0>{});
    if (p.stats) {
        __syncthreads();  // As is free now: reuse as [2][4 waves][BN]
        float* red = As;
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
            const int col = n0 + c;
            if (col < p.N) {
                const float t = red[(which * 4 + 0) * BN + c] + red[(which * 4 + 1) * BN + c] +
                                red[(which * 4 + 2) * BN + c] + red[(which * 4 + 3) * BN + c];
                // the partials buffer has one row per 128 output rows: a 256-row tile fills the first of its two
                // rows and zeroes the second
                const long long prow = (long long)bm * MT;
                p.stats[(prow * 2 + which) * p.N + col] = t;
                if (MT == 2 && (prow + 1) * 128 < p.M) p.stats[((prow + 1) * 2 + which) * p.N + col] = 0.f;
            }
        }
    }
}

struct WgradP {
    const float* a;
    const float* g;
    float* partial;
    long long M, rows_per_split;
    unsigned a_bytes, g_bytes;
    int T_a, T_g, V, K, N, ld_a, ld_g;
    int taps, ta, tb, tc, td;
    int tilesN;
    int tiles, nsplit, per_xcd;   // per_xcd > 0: 1-D grid, XCD-aware order (see the kernel)
};

template <bool BF>
__global__ __launch_bounds__(256) void rows_wgrad_kernel(WgradP p) {
    constexpr int BR = 64, TK = 64, TN = 64;
    __shared__ __attribute__((aligned(16))) float As[BR * TK];
    __shared__ __attribute__((aligned(16))) float Gs[BR * TN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wn = wave & 1;
    // Workgroups that read the same rows (the taps and K/N tiles of one row split) should share an L2: consecutive
    // workgroup ids go round-robin over the 8 XCDs, so id b works on virtual id (b % 8) * per_xcd + b / 8 and each XCD
    // walks a contiguous range of (split, tap, tile) with the split slowest.  Speed only: any placement is correct.
    int tile, tap, split;
    if (p.per_xcd > 0) {
        const int b = blockIdx.x;
        const int vid = (b & 7) * p.per_xcd + (b >> 3);
        if (vid >= p.tiles * p.taps * p.nsplit) return;
        split = vid / (p.tiles * p.taps);
        const int inner = vid - split * (p.tiles * p.taps);
        tap = inner / p.tiles;
        tile = inner - tap * p.tiles;
    } else {
        tile = blockIdx.x;
        tap = blockIdx.y;
        split = blockIdx.z;
    }
    const int tk = tile / p.tilesN, tn = tile - tk * p.tilesN;
    const long long mbeg = (long long)split * p.rows_per_split;
    long long mend = mbeg + p.rows_per_split;
    if (mend > p.M) mend = p.M;
    const int k0 = tk * TK, n0 = tn * TN;
    const int c4 = (tid & 15) * 4;
    const int TVg = p.T_g * p.V;

    // Row decode without divisions in the loop: each thread stages rows (tid >> 4) + 16*j of every 64-row stage;
    // (n, tg, v) of those rows are decoded once (32-bit) and advanced by 64 rows per stage with carries.
    // Loads are buffer loads: rows / channels that do not exist carry an out-of-range offset and read as zeros, so the
    // eight loads of a stage issue back to back with no branches.
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, p.g_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const bool kcol_ok = k0 + c4 < p.K, ncol_ok = n0 + c4 < p.N;
    const int dt = BR / p.V, dv = BR - dt * p.V;
    unsigned rm[4];
    int rn[4], rt[4], rv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        rm[j] = (unsigned)mbeg + (tid >> 4) + 16 * j;
        rn[j] = (int)(rm[j] / (unsigned)TVg);
        const unsigned rem = rm[j] - (unsigned)rn[j] * (unsigned)TVg;
        rt[j] = (int)(rem / (unsigned)p.V);
        rv[j] = (int)rem - rt[j] * p.V;
    }
    const unsigned mend_u = (unsigned)mend;

    f32x4 areg[4], greg[4];
    f32x16 acc = zero16();

    auto load_stage = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned ao = OOB, go = OOB;
            if (rm[j] < mend_u) {
                const int ti = tmap_src(rt[j], tap, p.ta, p.tb, p.tc, p.td, p.T_a);
                if (ti >= 0) {
                    if (kcol_ok) ao = ((unsigned)((rn[j] * p.T_a + ti) * p.V + rv[j]) * (unsigned)p.ld_a + k0 + c4) * 4u;
                    if (ncol_ok) go = (rm[j] * (unsigned)p.ld_g + n0 + c4) * 4u;
                }
            }
            areg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, ao, 0, 0));
            greg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, go, 0, 0));
        }
    };
    auto advance = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rm[j] += BR;
            rv[j] += dv;
            rt[j] += dt;
            if (rv[j] >= p.V) {
                rv[j] -= p.V;
                rt[j] += 1;
            }
            while (rt[j] >= p.T_g) {
                rt[j] -= p.T_g;
                rn[j] += 1;
            }
        }
    };

    if (mbeg < mend) load_stage();
    for (long long mb = mbeg; mb < mend; mb += BR) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (tid >> 4) + 16 * j;
            *reinterpret_cast<f32x4*>(&As[row * TK + c4]) = areg[j];
            *reinterpret_cast<f32x4*>(&Gs[row * TN + c4]) = greg[j];
        }
        __syncthreads();
        if (mb + BR < mend) {
            advance();
            load_stage();
        }
        if constexpr (BF) {   // 8 rows per MFMA: lane half h contracts rows 8g + 4h + (0..3)
            const float* ap = &As[4 * (lane >> 5) * TK + wk * 32 + (lane & 31)];
            const float* gp = &Gs[4 * (lane >> 5) * TN + wn * 32 + (lane & 31)];
#pragma unroll 4
            for (int g8 = 0; g8 < BR / 8; ++g8) {
                const float* a = ap + 8 * g8 * TK;
                const float* g = gp + 8 * g8 * TN;
                acc = mfma_bf16(pack_bf16(a[0], a[TK], a[2 * TK], a[3 * TK]), pack_bf16(g[0], g[TN], g[2 * TN], g[3 * TN]), acc);
            }
        } else {
            const float* ap = &As[(lane >> 5) * TK + wk * 32 + (lane & 31)];
            const float* gp = &Gs[(lane >> 5) * TN + wn * 32 + (lane & 31)];
#pragma unroll 8
            for (int s = 0; s < BR / 2; ++s) acc = mfma32(ap[2 * s * TK], gp[2 * s * TN], acc);
        }
    }

    const int n = n0 + wn * 32 + (lane & 31);
    if (n < p.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k0 + wk * 32 + acc_row(r, lane);
            if (k < p.K)
                p.partial[(((long long)split * p.taps + tap) * p.K + k) * p.N + n] = acc[r];
        }
    }
}

// dst[i] (+)= sum_s src[s][i].  block = 64 outputs x 16 slice-lanes: lane y adds slices y, y+16, ... in order, then
// the 16 partial sums are added in a fixed order -> deterministic, and S-way parallel (S reaches thousands for the
// BatchNorm / bias partials, where one thread per output would serialise thousands of dependent loads).
__global__ __launch_bounds__(1024) void reduce_sum_kernel(float* dst, const float* src, int S, long long count,
                                                          int accumulate) {
    __shared__ float red[16][65];
    const int x = threadIdx.x, y = threadIdx.y;
    const long long i = (long long)blockIdx.x * 64 + x;
    // eight independent loads in flight per thread (the slab walk is latency-bound: S reaches 2048 for the BatchNorm partials
    // while only a few workgroups cover `count`); the eight sub-sums are added in a fixed order
    float t = 0.f;
    if (i < count) {
        float u[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int s = y;
        for (; s + 7 * 16 < S; s += 8 * 16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) u[k] += src[(long long)(s + 16 * k) * count + i];
        }
        for (int k = 0; s < S; s += 16, ++k) u[k] += src[(long long)s * count + i];
        t = ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]));
    }
    red[y][x] = t;
    __syncthreads();
    if (y == 0 && i < count) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += red[k][x];
        dst[i] = accumulate ? dst[i] + a : a;
    }
}

// Same sum, but dst is written through strides: src slab element i = (tap, k, n) of a [taps][K][N] weight gradient goes
// to dst[tap*st_tap + k*st_k + n*st_n] (k < K_dst), i.e. straight into the parameter's own (out, in, taps, 1) layout,
// so autograd takes the tensor as it is instead of launching a transposing copy per parameter.
__global__ __launch_bounds__(1024) void reduce_sum_strided_kernel(float* dst, const float* src, int S, int taps, int K, int N,
                                                                  int K_dst, long long st_tap, long long st_k,
                                                                  long long st_n, int accumulate) {
    __shared__ float red[16][65];
    const int x = threadIdx.x, y = threadIdx.y;
    const long long count = (long long)taps * K * N;
    const long long i = (long long)blockIdx.x * 64 + x;
    // eight independent loads in flight per thread (the slab walk is latency-bound: S reaches 2048 for the BatchNorm partials
    // while only a few workgroups cover `count`); the eight sub-sums are added in a fixed order
    float t = 0.f;
    if (i < count) {
        float u[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int s = y;
        for (; s + 7 * 16 < S; s += 8 * 16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) u[k] += src[(long long)(s + 16 * k) * count + i];
        }
        for (int k = 0; s < S; s += 16, ++k) u[k] += src[(long long)s * count + i];
        t = ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]));
    }
    red[y][x] = t;
    __syncthreads();
    if (y == 0 && i < count) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += red[k][x];
        const int n = (int)(i % N);
        const long long tk = i / N;
        const int k = (int)(tk % K), tap = (int)(tk / K);
        if (k < K_dst) {
            float* d = dst + tap * st_tap + k * st_k + n * st_n;
            *d = accumulate ? *d + a : a;
        }
    }
}

// Several slab sums in one launch (fgcn_reduce_multi): the weight-gradient slabs, adj_b and bias-gradient partials of a block's
// backward are leaves -- nothing in the block reads them -- so their reductions are collected and issued together at the end
// (6-8 launches less per block; at 8 clips per GPU a launch costs as much as the sum it carries).  Item i covers workgroups
// [first[i], first[i+1]); the arithmetic and its order are exactly reduce_sum_strided_kernel's.
struct ReduceMultiP {
    fgcn_reduce_item it[FGCN_REDUCE_MAX_ITEMS];
    int first[FGCN_REDUCE_MAX_ITEMS + 1];
    int n;
    unsigned vec_mask;      // bit i: item i takes the few-slabs form below
};

// Few slabs, many elements (weight-gradient slabs: S = 16..64 slabs of up to 590k elements): a thread owns four consecutive
// elements and walks the slabs itself with eight 16-byte loads in flight -- fully coalesced, no LDS, no barrier.  (The 16-way
// slab-parallel form above is for the opposite shape, thousands of BatchNorm / bias partials of a few hundred elements; on
// the weight-gradient slabs it ran 7x below the HBM rate: 0.45 ms per step.)  Fixed order: bitwise reproducible.
__device__ __forceinline__ void reduce_item_vec(const fgcn_reduce_item& it, int block_local) {
    const long long count = (long long)it.taps * it.K * it.N;
    const long long i0 = ((long long)block_local * 1024 + threadIdx.y * 64 + threadIdx.x) * 4;
    if (i0 >= count) return;
    const float* src = it.src + i0;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    int s = 0;
    for (; s + 8 <= it.S; s += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (long long)(s + u) * count);
        a0 += v[0] + v[4];
        a1 += v[1] + v[5];
        a2 += v[2] + v[6];
        a3 += v[3] + v[7];
    }
    for (; s < it.S; ++s) a0 += *reinterpret_cast<const f32x4*>(src + (long long)s * count);
    const f32x4 a = (a0 + a1) + (a2 + a3);
    const int n = (int)(i0 % it.N);                     // N % 4 == 0: the four elements share (tap, k)
    const long long tk = i0 / it.N;
    const int k = (int)(tk % it.K), tap = (int)(tk / it.K);
    if (k >= it.K_dst) return;
    float* d = it.dst + tap * it.st_tap + k * it.st_k + n * it.st_n;
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e * it.st_n] = it.accumulate ? d[e * it.st_n] + a[e] : a[e];
}

__global__ __launch_bounds__(1024) void reduce_multi_kernel(ReduceMultiP p) {
    __shared__ float red[16][65];
    int which = 0;
#pragma unroll
    for (int i = 1; i < FGCN_REDUCE_MAX_ITEMS; ++i)
        if (i < p.n && (int)blockIdx.x >= p.first[i]) which = i;
    const fgcn_reduce_item& it = p.it[which];
    if ((p.vec_mask >> which) & 1u) {                  // block-uniform
        reduce_item_vec(it, (int)blockIdx.x - p.first[which]);
        return;
    }
    const int x = threadIdx.x, y = threadIdx.y;
    const long long count = (long long)it.taps * it.K * it.N;
    const long long i = (long long)((int)blockIdx.x - p.first[which]) * 64 + x;
    float t = 0.f;
    if (i < count) {
        float u[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int s = y;
        for (; s + 7 * 16 < it.S; s += 8 * 16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) u[k] += it.src[(long long)(s + 16 * k) * count + i];
        }
        for (int k = 0; s < it.S; s += 16, ++k) u[k] += it.src[(long long)s * count + i];
        t = ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]));
    }
    red[y][x] = t;
    __syncthreads();
    if (y == 0 && i < count) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += red[k][x];
        const int n = (int)(i % it.N);
        const long long tk = i / it.N;
        const int k = (int)(tk % it.K), tap = (int)(tk / it.K);
        if (k < it.K_dst) {
            float* d = it.dst + tap * it.st_tap + k * it.st_k + n * it.st_n;
            *d = it.accumulate ? *d + a : a;
        }
    }
}

// FGCN_MATH_BF16X3 weights: (taps, K, N) f32 -> [part][tap][ceil(K/8)][N][8] bf16, part 0/1/2 = high / middle / low term of
// the exact three-way split w = w_h + w_m + w_l; 8 consecutive k per (n) = one lane's B fragment of
// v_mfma_f32_32x32x16_bf16 (16 bytes, lanes = consecutive n).  Channels beyond K are zeros.
// acc_order: group k8 = 2*k16 + h holds k = 16*k16 + 4h + (j & 3) + 8*(j >> 2) instead of 8*k8 + j -- the order in which
// the registers of a 32x32 accumulator tile (rows (r&3) + 8(r>>2) + 4h) enumerate its rows, for kernels that feed an
// accumulator straight back as the other operand (spatial_fwd step 2).
__global__ void pack_split3_kernel(unsigned short* dst, const float* src, int taps, int K, int N, int K8, int acc_order) {
    const long long total = (long long)taps * K8 * N;
    const long long plane = total * 8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);
        const long long tk = i / N;
        const int k8 = (int)(tk % K8), tap = (int)(tk / K8);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = acc_order ? 16 * (k8 >> 1) + 4 * (k8 & 1) + (j & 3) + 8 * (j >> 2) : 8 * k8 + j;
            v[j] = k < K ? src[((long long)tap * K + k) * N + n] : 0.f;
        }
        u32x2 h0, m0, l0, h1, m1, l1;
        split3_x4(f32x4{v[0], v[1], v[2], v[3]}, h0, m0, l0);
        split3_x4(f32x4{v[4], v[5], v[6], v[7]}, h1, m1, l1);
        u32x4v* out = reinterpret_cast<u32x4v*>(dst + i * 8);
        out[0] = u32x4v{h0[0], h0[1], h1[0], h1[1]};
        *reinterpret_cast<u32x4v*>(dst + plane + i * 8) = u32x4v{m0[0], m0[1], m1[0], m1[1]};
        *reinterpret_cast<u32x4v*>(dst + 2 * plane + i * 8) = u32x4v{l0[0], l0[1], l1[0], l1[1]};
    }
}

__global__ void pack_weight_kernel(float* dst, const float* src, int taps, int K, int N_src, int N_dst,
                                   long long st_tap, long long st_k, long long st_n, int flip) {
    const long long total = (long long)taps * K * N_dst;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N_dst);
        const long long jk = i / N_dst;
        const int k = (int)(jk % K);
        const int j = (int)(jk / K);
        const int jj = flip ? taps - 1 - j : j;
        dst[i] = n < N_src ? src[(long long)n * st_n + (long long)k * st_k + (long long)jj * st_tap] : 0.f;
    }
}

}  // namespace fgcn

using namespace fgcn;

extern "C" int fgcn_rows_gemm_tiles(long long M) { return (int)cdiv(M, 128); }

static int check_tmap(const fgcn_tmap& m) {
    FGCN_REQUIRE(m.taps >= 1 && m.taps <= 16 && m.td >= 1 && m.ta >= 0, FGCN_E_BADARG,
                 "bad temporal map (taps=%d ta=%d td=%d)", m.taps, m.ta, m.td);
    return FGCN_OK;
}

static int rows_gemm_launch(const float* in, float* out, const float* w, const float* bias, float* stat_partials,
                            int B, int T_in, int T_out, int V, int K, int N, int ld_in, int ld_out,
                            fgcn_tmap map, int accumulate, int batch, long long in_bs, long long out_bs, long long w_bs,
                            void* stream, int inner = 1, long long in_bs2 = 0, long long out_bs2 = 0, long long w_bs2 = 0, int io = 0) {
    FGCN_REQUIRE(in && out && w, FGCN_E_BADARG, "rows_gemm: null pointer");
    FGCN_REQUIRE(io == 0 || (fgcn::math_mode() == FGCN_MATH_BF16 && batch == 1 && inner == 1 && !((io & 2) && accumulate)), FGCN_E_BADARG,
                 "rows_gemm_t: bfloat16 tensors need math mode bf16, a single problem and (for a bfloat16 output) no accumulation");
    FGCN_REQUIRE(batch >= 1 && inner >= 1 && (long long)batch * inner <= 65535 && in_bs % 4 == 0 && out_bs % 4 == 0 && w_bs % 4 == 0 &&
                     in_bs2 % 4 == 0 && out_bs2 % 4 == 0 && w_bs2 % 4 == 0,
                 FGCN_E_BADARG, "rows_gemm: batch=%d x %d / batch strides must be multiples of 4 floats", batch, inner);
    FGCN_REQUIRE(B > 0 && T_in > 0 && T_out > 0 && V > 0 && K > 0 && N > 0, FGCN_E_BADARG,
                 "rows_gemm: non-positive size B=%d T_in=%d T_out=%d V=%d K=%d N=%d", B, T_in, T_out, V, K, N);
    FGCN_REQUIRE(K % 4 == 0 && N % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0, FGCN_E_ALIGN,
                 "rows_gemm: K, N, ld_in, ld_out must be multiples of 4 (K=%d N=%d ld_in=%d ld_out=%d)", K, N, ld_in, ld_out);
    FGCN_REQUIRE((long long)T_in * V * ld_in < (1ll << 27) && (long long)map.taps * K * N < (1ll << 28) && ld_out < (1 << 20),
                 FGCN_E_BADARG, "rows_gemm: one sample / the weights exceed the 32-bit offset range");
    FGCN_REQUIRE(ld_in >= ((K + 3) & ~3) && ld_out >= N, FGCN_E_BADARG,
                 "rows_gemm: row strides too small (K=%d ld_in=%d N=%d ld_out=%d)", K, ld_in, N, ld_out);
    FGCN_REQUIRE(aligned16(in) && aligned16(out) && aligned16(w), FGCN_E_ALIGN, "rows_gemm: 16-byte alignment");
    if (int e = check_tmap(map)) return e;
    RowsGemmP p{in, out, w, bias, stat_partials, (long long)B * T_out * V, T_in, T_out, V, K, N, ld_in, ld_out,
                map.taps, map.ta, map.tb, map.tc, map.td, accumulate,
                (long long)B * T_in * V * ld_in, (unsigned)((long long)map.taps * K * N * 4), 0, 0, 0, in_bs, out_bs, w_bs,
                (!accumulate && fgcn::stream_out((long long)batch * inner * B * T_out * V * N * 4)) ? 1 : 0};   // (all batch levels: `batch *= inner` below)
    hipStream_t s = (hipStream_t)stream;
    // tile width (32*nt channels) with the fewest padded columns; ties go to the wider tile
    int nt = 4;
    long long best = -1;
    for (int c = 4; c >= 1; --c) {
        const long long padded = cdiv(N, 32 * c) * 32 * c;
        if (best < 0 || padded < best) {
            best = padded;
            nt = c;
        }
    }
    // narrow outputs (<= 64 channels per tile) take two row tiles per wave so every A/B fragment feeds 2 MFMAs
    const int tune_small = fgcn::tuning(0), tune_wide = fgcn::tuning(1);
    int mt = (nt <= 2 && tune_small != 0) ? 2 : 1;
    // small problems (the per-sample V x V products of the IMU graph convolutions: 8 x 652 rows per launch): 256-row x 64-column tiles
    // left 24 workgroups for 256 CUs -- the smallest tiles that still pad no extra column, until the grid covers the chip
    // (tuning key 24 = 1: the large-problem tiles everywhere)
    batch *= inner;
    if (fgcn::tuning(24) != 1) {
        auto wgs = [&](int mt_, int nt_) { return cdiv(p.M, 128 * mt_) * cdiv(N, 32 * nt_) * batch; };
        if (mt == 2 && wgs(mt, nt) < 256) mt = 1;
        while (nt % 2 == 0 && wgs(mt, nt) < 256) nt /= 2;       // (32 nt/2 divides 32 nt: never more padded columns)
        if (nt == 3 && wgs(mt, nt) < 256) nt = 1;
    }
    p.inner = inner; p.in_bs2 = in_bs2; p.out_bs2 = out_bs2; p.w_bs2 = w_bs2;
    const bool db = nt <= 2 ? tune_small == 2 : tune_wide == 1;
    const long long tiles_m = cdiv(p.M, 128 * mt);
    FGCN_REQUIRE(p.M < (1ll << 29) - 4096, FGCN_E_BADARG, "rows_gemm: too many rows (2^29: multiply-shift row decode)");
    p.dTV = make_fastdiv((unsigned)(T_out * V));
    p.dV = make_fastdiv((unsigned)V);
    p.tiles_m = (int)tiles_m;
    p.tiles_n = (int)cdiv(N, 32 * nt);
    const long long total = tiles_m * p.tiles_n;
    dim3 grid((unsigned)tiles_m, (unsigned)p.tiles_n, (unsigned)batch);
    if ((fgcn::tuning(5) & 1) && p.tiles_n > 1 && total < (1ll << 30)) {   // measured slower than the plain 2-D grid: off
        p.per_xcd = (int)cdiv(total, 8);
        grid = dim3((unsigned)(p.per_xcd * 8), 1, (unsigned)batch);
    }
    else if (!(fgcn::tuning(5) & 8) && p.tiles_n > 1 && tiles_m < 65536) {   // measured 1-3 % faster than row tile fastest
        p.per_xcd = -1;
        grid = dim3((unsigned)p.tiles_n, (unsigned)tiles_m, (unsigned)batch);
    }
    const bool bf = fgcn::math_mode() == FGCN_MATH_BF16;
#define FGCN_LAUNCH(MT_, NT_, DB_)                                                                         \
    do {                                                                                                   \
        if (bf && io == 3) hipLaunchKernelGGL((rows_gemm_kernel<MT_, NT_, DB_, true, 3>), grid, dim3(256), 0, s, p);      \
        else if (bf && io == 2) hipLaunchKernelGGL((rows_gemm_kernel<MT_, NT_, DB_, true, 2>), grid, dim3(256), 0, s, p); \
        else if (bf && io == 1) hipLaunchKernelGGL((rows_gemm_kernel<MT_, NT_, DB_, true, 1>), grid, dim3(256), 0, s, p); \
        else if (bf) hipLaunchKernelGGL((rows_gemm_kernel<MT_, NT_, DB_, true>), grid, dim3(256), 0, s, p); \
        else hipLaunchKernelGGL((rows_gemm_kernel<MT_, NT_, DB_, false>), grid, dim3(256), 0, s, p);       \
    } while (0)
    if (mt == 2) {
        if (nt == 1) { if (db) FGCN_LAUNCH(2, 1, true); else FGCN_LAUNCH(2, 1, false); }
        else { if (db) FGCN_LAUNCH(2, 2, true); else FGCN_LAUNCH(2, 2, false); }
    } else {
        switch (nt) {
            case 1: FGCN_LAUNCH(1, 1, false); break;
            case 2: FGCN_LAUNCH(1, 2, false); break;
            case 3: if (db) FGCN_LAUNCH(1, 3, true); else FGCN_LAUNCH(1, 3, false); break;
            default: if (db) FGCN_LAUNCH(1, 4, true); else FGCN_LAUNCH(1, 4, false); break;
        }
    }
#undef FGCN_LAUNCH
    return launch_status("rows_gemm");
}

extern "C" int fgcn_rows_gemm(const float* in, float* out, const float* w, const float* bias, float* stat_partials,
                              int B, int T_in, int T_out, int V, int K, int N, int ld_in, int ld_out,
                              fgcn_tmap map, int accumulate, void* stream) {
    return rows_gemm_launch(in, out, w, bias, stat_partials, B, T_in, T_out, V, K, N, ld_in, ld_out, map, accumulate, 1, 0, 0, 0,
                            stream);
}

// typed form (math mode bf16, half-precision activation storage): half_mask bit 0 = `in` is a bfloat16 tensor, bit 1 = `out` is (not with accumulation);
// strides in elements, stat_partials: the moments of the float32 results
extern "C" int fgcn_rows_gemm_t(const void* in, void* out, const float* w, const float* bias, float* stat_partials,
                                int B, int T_in, int T_out, int V, int K, int N, int ld_in, int ld_out,
                                fgcn_tmap map, int accumulate, int half_mask, void* stream) {
    FGCN_REQUIRE((half_mask & ~3) == 0, FGCN_E_BADARG, "rows_gemm_t: half_mask=%d", half_mask);
    return rows_gemm_launch(static_cast<const float*>(in), static_cast<float*>(out), w, bias, stat_partials, B, T_in, T_out, V, K, N, ld_in, ld_out,
                            map, accumulate, 1, 0, 0, 0, stream, 1, 0, 0, 0, half_mask);
}

extern "C" int fgcn_rows_gemm_batched(const float* in, float* out, const float* w, int batch, long long in_bstride,
                                      long long out_bstride, long long w_bstride, int rows, int K, int N, int ld_in, int ld_out,
                                      int accumulate, void* stream) {
    FGCN_REQUIRE(rows > 0, FGCN_E_BADARG, "rows_gemm_batched: rows=%d", rows);
    const fgcn_tmap pointwise{1, 1, 0, 0, 1};
    return rows_gemm_launch(in, out, w, nullptr, nullptr, 1, rows, rows, 1, K, N, ld_in, ld_out, pointwise, accumulate, batch,
                            in_bstride, out_bstride, w_bstride, stream);
}

extern "C" int fgcn_rows_gemm_batched2(const float* in, float* out, const float* w, int batch, long long in_bstride,
                                       long long out_bstride, long long w_bstride, int inner, long long in_bstride2,
                                       long long out_bstride2, long long w_bstride2, int rows, int K, int N, int ld_in, int ld_out,
                                       int accumulate, void* stream) {
    FGCN_REQUIRE(rows > 0, FGCN_E_BADARG, "rows_gemm_batched2: rows=%d", rows);
    const fgcn_tmap pointwise{1, 1, 0, 0, 1};
    return rows_gemm_launch(in, out, w, nullptr, nullptr, 1, rows, rows, 1, K, N, ld_in, ld_out, pointwise, accumulate, batch,
                            in_bstride, out_bstride, w_bstride, stream, inner, in_bstride2, out_bstride2, w_bstride2);
}

extern "C" int fgcn_rows_wgrad(const float* a, const float* g, float* partial,
                               int B, int T_a, int T_g, int V, int K, int N, int ld_a, int ld_g,
                               fgcn_tmap map, int nsplit, void* stream) {
    FGCN_REQUIRE(a && g && partial, FGCN_E_BADARG, "rows_wgrad: null pointer");
    FGCN_REQUIRE(B > 0 && T_a > 0 && T_g > 0 && V > 0 && K > 0 && N > 0 && nsplit > 0, FGCN_E_BADARG,
                 "rows_wgrad: non-positive size");
    FGCN_REQUIRE(ld_a % 4 == 0 && ld_g % 4 == 0 && ld_a >= ((K + 3) & ~3) && ld_g >= ((N + 3) & ~3), FGCN_E_ALIGN,
                 "rows_wgrad: row strides must be multiples of 4 and cover the channels (K=%d ld_a=%d N=%d ld_g=%d)",
                 K, ld_a, N, ld_g);
    FGCN_REQUIRE(aligned16(a) && aligned16(g), FGCN_E_ALIGN, "rows_wgrad: 16-byte alignment");
    if (int e = check_tmap(map)) return e;
    FGCN_REQUIRE(nsplit <= 65535, FGCN_E_BADARG, "rows_wgrad: nsplit too large");
    WgradP p;
    p.a = a; p.g = g; p.partial = partial;
    p.M = (long long)B * T_g * V;
    FGCN_REQUIRE(p.M < (1ll << 31) - 4096, FGCN_E_BADARG, "rows_wgrad: too many rows (32-bit row indices)");
    const long long a_bytes = (long long)B * T_a * V * ld_a * 4, g_bytes = p.M * ld_g * 4;
    FGCN_REQUIRE(a_bytes < 0x7FFF0000ll && g_bytes < 0x7FFF0000ll, FGCN_E_BADARG,
                 "rows_wgrad: operands must be smaller than 2 GiB (32-bit buffer offsets)");
    p.a_bytes = (unsigned)a_bytes; p.g_bytes = (unsigned)g_bytes;
    p.rows_per_split = cdiv(cdiv(p.M, nsplit), 64) * 64;
    p.T_a = T_a; p.T_g = T_g; p.V = V; p.K = K; p.N = N; p.ld_a = ld_a; p.ld_g = ld_g;
    p.taps = map.taps; p.ta = map.ta; p.tb = map.tb; p.tc = map.tc; p.td = map.td;
    p.tilesN = (int)cdiv(N, 64);
    p.tiles = (int)cdiv(K, 64) * p.tilesN;
    p.nsplit = nsplit;
    const long long total = (long long)p.tiles * map.taps * nsplit;
    const bool bf = fgcn::math_mode() == FGCN_MATH_BF16;
    if (!(fgcn::tuning(5) & 4) && total < (1ll << 30)) {   // measured: -7 % at 64 channels, neutral above
        p.per_xcd = (int)cdiv(total, 8);
        const dim3 grid((unsigned)(p.per_xcd * 8));
        if (bf) hipLaunchKernelGGL(rows_wgrad_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(rows_wgrad_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, p);
    } else {
        p.per_xcd = 0;
        const dim3 grid((unsigned)p.tiles, (unsigned)map.taps, (unsigned)nsplit);
        if (bf) hipLaunchKernelGGL(rows_wgrad_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(rows_wgrad_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, p);
    }
    return launch_status("rows_wgrad");
}

extern "C" int fgcn_reduce_sum(float* dst, const float* src, int S, long long count, int accumulate, void* stream) {
    FGCN_REQUIRE(dst && src && S > 0 && count > 0, FGCN_E_BADARG, "reduce_sum: bad argument");
    FGCN_REQUIRE(cdiv(count, 64) < (1ll << 31), FGCN_E_BADARG, "reduce_sum: count too large");
    hipLaunchKernelGGL(reduce_sum_kernel, dim3((unsigned)cdiv(count, 64)), dim3(64, 16), 0, (hipStream_t)stream, dst, src,
                       S, count, accumulate);
    return launch_status("reduce_sum");
}

extern "C" int fgcn_reduce_sum_strided(float* dst, const float* src, int S, int taps, int K, int N, int K_dst,
                                       long long st_tap, long long st_k, long long st_n, int accumulate, void* stream) {
    FGCN_REQUIRE(dst && src && S > 0 && taps > 0 && K > 0 && N > 0 && K_dst > 0 && K_dst <= K, FGCN_E_BADARG,
                 "reduce_sum_strided: bad argument");
    const long long count = (long long)taps * K * N;
    FGCN_REQUIRE(cdiv(count, 64) < (1ll << 31), FGCN_E_BADARG, "reduce_sum_strided: count too large");
    hipLaunchKernelGGL(reduce_sum_strided_kernel, dim3((unsigned)cdiv(count, 64)), dim3(64, 16), 0, (hipStream_t)stream, dst,
                       src, S, taps, K, N, K_dst, st_tap, st_k, st_n, accumulate);
    return launch_status("reduce_sum_strided");
}

extern "C" int fgcn_reduce_multi(const fgcn_reduce_item* items, int n_items, void* stream) {
    FGCN_REQUIRE(items && n_items >= 1 && n_items <= FGCN_REDUCE_MAX_ITEMS, FGCN_E_BADARG, "reduce_multi: %d items (1..%d)",
                 n_items, FGCN_REDUCE_MAX_ITEMS);
    ReduceMultiP p;
    p.vec_mask = 0;
    long long blocks = 0;
    for (int i = 0; i < n_items; ++i) {
        const fgcn_reduce_item& it = items[i];
        FGCN_REQUIRE(it.dst && it.src && it.S > 0 && it.taps > 0 && it.K > 0 && it.N > 0 && it.K_dst > 0 && it.K_dst <= it.K,
                     FGCN_E_BADARG, "reduce_multi: item %d malformed", i);
        p.it[i] = it;
        p.first[i] = (int)blocks;
        const long long count = (long long)it.taps * it.K * it.N;
        const bool vec = it.S <= 128 && it.N % 4 == 0 && count >= 4096 && aligned16(it.src) && count % 4 == 0;
        if (vec) p.vec_mask |= 1u << i;
        blocks += vec ? cdiv(count, 4096) : cdiv(count, 64);
        FGCN_REQUIRE(blocks < (1ll << 31), FGCN_E_BADARG, "reduce_multi: too much work for one grid");
    }
    for (int i = n_items; i <= FGCN_REDUCE_MAX_ITEMS; ++i) p.first[i] = (int)blocks;
    p.n = n_items;
    hipLaunchKernelGGL(reduce_multi_kernel, dim3((unsigned)blocks), dim3(64, 16), 0, (hipStream_t)stream, p);
    return launch_status("reduce_multi");
}

extern "C" int fgcn_pack_split3(unsigned short* dst, const float* src, int taps, int K, int N, int acc_order,
                                void* stream) {
    FGCN_REQUIRE(dst && src && taps > 0 && K > 0 && N > 0, FGCN_E_BADARG, "pack_split3: bad argument (taps=%d K=%d N=%d)",
                 taps, K, N);
    FGCN_REQUIRE(aligned16(dst), FGCN_E_ALIGN, "pack_split3: 16-byte alignment");
    const int K8 = acc_order ? (K + 15) / 16 * 2 : (K + 7) / 8;
    const long long total = (long long)taps * K8 * N;
    const unsigned blocks = (unsigned)(cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048);
    hipLaunchKernelGGL(pack_split3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dst, src, taps, K, N, K8,
                       acc_order);
    return launch_status("pack_split3");
}

extern "C" int fgcn_pack_weight(float* dst, const float* src, int taps, int K, int N_src, int N_dst,
                                long long st_tap, long long st_k, long long st_n, int flip, void* stream) {
    FGCN_REQUIRE(dst && src && taps > 0 && K > 0 && N_src > 0 && N_dst >= N_src && N_dst % 4 == 0, FGCN_E_BADARG,
                 "pack_weight: bad argument (taps=%d K=%d N_src=%d N_dst=%d)", taps, K, N_src, N_dst);
    const long long total = (long long)taps * K * N_dst;
    const unsigned blocks = (unsigned)(cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dst, src, taps, K, N_src,
                       N_dst, st_tap, st_k, st_n, flip);
    return launch_status("pack_weight");
}
