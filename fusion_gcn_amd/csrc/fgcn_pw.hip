// 1x1 convolutions (theta|phi embedding, dY.Wd, down, dEmb.W; reference torch_src/models/mmargcn/agcn.py:71-73,77,104-111 and their
// data gradients) in the split-bf16 math modes: a PERSISTENT row GEMM  out[m][n] (+)= sum_k in[m][k] W[k][n] + bias[n].
//
// Why a kernel of its own.  A 1x1 convolution has K = 64..384: one to six 64-channel chunks per (128 rows x 128 columns) tile.  Run on
// the halo-tile kernel (fgcn_tconv.hip, one workgroup per tile) such a tile is a staging latency, <= 6 short MFMA phases and a store
// tail, with nothing of its own to overlap them: 104-140 TFLOP/s where the nine-tap form of the same kernel reaches 210-237, and
// below K = 192 the exact-f32 row GEMM was the faster choice (3.2-4.3 TB/s, neither matrix- nor HBM-bound).  Here a workgroup walks a
// list of tiles: the rows of the NEXT chunk -- or of the next tile's first chunk -- are requested before the MFMAs of the current one
// and parked in registers, so the global latency and the previous tile's store tail lie under matrix work; the weight ring crosses
// tile boundaries the same way.  MFMA core, LDS image (unpadded 128-byte bf16 rows, 32-byte blocks XOR-swizzled by row bits:
// conflict-free ds_read_b128), 2 x 2 wave arrangement, weight form (fgcn_pack_split3) and XCD-aware tile order (column tiles of a row
// tile back to back on one XCD) are the halo kernel's.
//   NP = parts per operand: 3 (FGCN_MATH_BF16X3: six bf16 partial products, f32 accuracy), 1 (FGCN_MATH_BF16), or 2 = two f16 parts and
//        three products (FGCN_PRODUCTS_F16X2, fgcn_common.hpp): the rows of a chunk are scaled by 2^ea as they are split, ea from the
//        chunk's largest magnitude (wave maxima through four LDS words ahead of the barrier that exists anyway); the accumulators are
//        rescaled (a power of two: exact) whenever the scale moves, and the epilogue multiplies 2^-ea 2^-ew back out.
//   NT = 1 / 2: 64 / 128 output columns per tile.
#include "fgcn_common.hpp"

// Timing probes (wrong results; tools/probes builds only): bit 0 = no output stores, bit 1 = no MFMAs, bit 2 = the image is deposited
// once per workgroup (later chunks skip the split + LDS writes), bit 3 = no input fetches after the first, bit 4 = weight fragments
// loaded for the first chunk only
#ifndef FGCN_PW_RING
#define FGCN_PW_RING 4                  // 2: the two-slot ring everywhere (A/B builds)
#endif
#ifndef FGCN_PROBE_PW
#define FGCN_PROBE_PW 0
#endif

namespace fgcn {

struct PwP {
    const float* in;
    float* out;
    const void* w3;                     // [part][K/8][N][8] bf16; NP == 2: FGCN_PACK_SPLIT2H (16-byte header, then two f16 parts)
    const float* bias;
    float* stats;                       // float[tiles_m][2][N] or NULL: per row tile, sum and sum of squares of the values written
    long long M;
    unsigned in_bytes, w_plane_bytes, out_bytes;
    int K, N, ld_in, ld_out, accumulate;
    int tiles_m, tiles_n, per_xcd, wg_per_xcd;
    unsigned* in_amax;                  // NP == 2: receives max |in| over everything staged (integer atomic maximum of the float bits) or NULL
};

using u32x4p = __attribute__((ext_vector_type(4))) unsigned int;

// ACC (compile time): out += ... (see the epilogue).  A run-time `if (p.accumulate)` around loads of the old values,
// wave-uniform as it was, made hipcc branch inside the unrolled epilogue and lose count of the outstanding memory operations: it
// drained vmcnt(0) in front of EVERY group of four stores, i.e. sixteen full write round trips per tile and wave (found with the
// FGCN_PROBE_PW timing probes: the stores cost 22 % of the kernel, the time the written bytes take at the HBM rate, with nothing
// overlapping them).
// IO (NP = 1; the typed entry point fgcn_pw_gemm_t, half-precision activation storage): bit 0 = `in` is a BFLOAT16 tensor (its rows are copied into
// the image: the staged bytes of the float32 tensor of the same values), bit 1 = `out` is (not with ACC; the float32 result rounded once, BatchNorm
// sums of the float32 values, adjacent lanes pair their columns into dword stores)
template <int NT, int NP, bool ACC, bool STR = false, int IO = 0>          // STR: non-temporal output stores (fgcn_common.hpp, stream_out)
__global__ __launch_bounds__(256, 2) void pw_x3_kernel(PwP p) {
    static_assert(!(ACC && STR), "an accumulating epilogue stores plainly");
    static_assert(IO == 0 || (NP == 1 && !((IO & 2) && ACC)), "bfloat16 tensors: the one-part kernel; a bfloat16 output without accumulation");
    constexpr bool IN16 = (IO & 1) != 0, OUT16 = (IO & 2) != 0;
    static_assert((NT == 1 || NT == 2) && (NP == 1 || NP == 2 || NP == 3), "64 / 128 columns; one or three bf16 parts, or two f16 parts");
    constexpr int KC = 64, XS = 2 * KC, BMR = 128, MTW = 4, NU = 2 * NT, BN = 64 * NT;
    // Weight ring: fragments are requested RS - 1 units (of 24 MFMAs) ahead.  One unit is 384 matrix cycles, less than an L2 round trip
    // under load: the FGCN_PROBE_PW bit 4 probe (weights loaded once) ran 16-29 % faster than the kernel with a two-slot ring.  Four
    // slots (three units of cover) where the registers are there -- 128-column tiles without the accumulating epilogue (256 VGPRs, no
    // scratch): -3 .. -18 % per launch (profiles/r03_ab_pw_ring.txt); the accumulating form would spill (and its old-value loads shrunk
    // to half row tiles to make room cost more than the ring wins), the 64-column form has two units per step.
    constexpr int RS = (FGCN_PW_RING == 4 && NU == 4 && !ACC) ? 4 : 2;
    constexpr int TPR = KC / 4, RPP = 256 / TPR, NST = BMR / RPP;      // 16 threads per row, 16 rows per pass, 8 passes
    constexpr unsigned OOB = 0x80000000u;
    auto swz = [](int r) -> unsigned { return (unsigned)(r & 6) << 4; };
    extern __shared__ __attribute__((aligned(16))) float Ah[];
    unsigned char* Xh = reinterpret_cast<unsigned char*>(Ah);
    constexpr unsigned plane = BMR * XS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const unsigned char*>(p.w3) + (NP == 2 ? 16 : 0)), 0, p.w_plane_bytes * NP, 0x00020000);
    constexpr int EA_NONE = 1000;                                    // "no scale yet" (every chunk so far was all zeros)
    const int ew = NP == 2 ? min(scale_exp_for(*reinterpret_cast<const unsigned*>(p.w3)), 126) : 0;
    float* smax = reinterpret_cast<float*>(Xh + NP * plane);         // NP == 2: the four waves' chunk maxima
    int ea = EA_NONE, abound = 0;
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? (const void*)p.bias : p.w3), 0,
                                                                           p.bias ? (unsigned)p.N * 4u : 0u, 0x00020000);

    // this workgroup's tiles: ids go round-robin over the 8 XCDs, so workgroup b works inside XCD (b & 7)'s share of the virtual tile
    // list [x * per_xcd, (x + 1) * per_xcd) (column tile fastest), taking every wg_per_xcd-th tile from its local index on
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int v_end = min((xcd + 1) * p.per_xcd, p.tiles_m * p.tiles_n);
    int vid = xcd * p.per_xcd + local;
    if (vid >= v_end) return;

    const unsigned k4b = (unsigned)(tid % TPR) * 16u;
    const int k4 = (tid % TPR) * 4;
    const int K8 = p.K >> 3;
    const int xrow = wr * (16 * MTW) + l15;

    unsigned src_off[NST];
    f32x4 stage[NST];
    auto set_rows = [&](int bm) {                                    // image row r = tid / 16 + 16 i of tile bm
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const long long m = (long long)bm * BMR + tid / TPR + RPP * i;
            src_off[i] = m < p.M ? (unsigned)(m * p.ld_in * (IN16 ? 2 : 4)) + (IN16 ? k4b >> 1 : k4b) : OOB;
        }
    };
    bool probe_first = true;
    auto fetch = [&](int kc) {
        if ((FGCN_PROBE_PW & 8) && !probe_first) return;
        const bool kok = kc + k4 < p.K;                              // K % 4 == 0: a 16-byte group is whole or absent
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            if constexpr (IN16) {                                    // four bfloat16 = 8 bytes, parked in the first two components
                const u32x2 h = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rin, kok ? src_off[i] : OOB, (unsigned)kc * 2u, 0));
                const unsigned b0 = h[0], b1 = h[1];                 // (element -> scalar before a bit cast: hipcc 7.2 reads element 0 otherwise)
                stage[i] = f32x4{__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1), 0.f, 0.f};
            } else {
                stage[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, kok ? src_off[i] : OOB, (unsigned)kc * 4u, 0));
            }
        }
    };
    auto deposit = [&]() {
        const float a_scale = ea == EA_NONE ? 1.f : exp2i(ea);
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int r = tid / TPR + RPP * i;
            u32x2 ph, pm, pl;
            unsigned char* dst = Xh + r * XS + ((unsigned)((tid % TPR) * 8) ^ swz(r));
            if constexpr (NP == 2) {
                split2h_x4(stage[i] * a_scale, ph, pm);
                *reinterpret_cast<u32x2*>(dst) = ph;
                *reinterpret_cast<u32x2*>(dst + plane) = pm;
                continue;
            }
            if constexpr (IN16) {                                    // already bfloat16: a copy
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
    };
    // weight fragment of (column unit nu of column tile bn, 32-channel step at channel k): lane (l15, g4) holds k + 8 g4 + j
    auto load_w = [&](u32x4v (&dst)[NP], int bn, int nu, int k) {
        const int col = bn * BN + wc * NT * 32 + nu * 16 + l15;
        const int kg = (k >> 3) + g4;
        const unsigned off = (col < p.N && kg < K8) ? (unsigned)(((long long)kg * p.N + col) * 16) : OOB;
        if ((FGCN_PROBE_PW & 16) && !probe_first) return;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) dst[pl] = __builtin_amdgcn_raw_buffer_load_b128(rw, off, pl * p.w_plane_bytes, 0);
    };
    auto load_a = [&](u32x4v (&dst)[NP], int mt, int s2) {
        const int r = xrow + mt * 16;
        const unsigned char* src = Xh + r * XS + ((unsigned)(16 * g4 + 64 * s2) ^ swz(r));
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) dst[pl] = *reinterpret_cast<const u32x4v*>(src + pl * plane);
    };

    int bm = vid / p.tiles_n, bn = vid - bm * p.tiles_n;
    u32x4v a[MTW][NP], wq[RS][NP];
    set_rows(bm);
    fetch(0);
#pragma unroll
    for (int nu = 0; nu < RS - 1; ++nu) load_w(wq[nu], bn, nu, 0);
    while (true) {
        const int vnext = vid + p.wg_per_xcd;
        const bool more = vnext < v_end;
        const int bm_n = more ? vnext / p.tiles_n : bm, bn_n = more ? vnext - bm_n * p.tiles_n : bn;
        f32x4 acc[MTW][NU];
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nu = 0; nu < NU; ++nu) acc[mt][nu] = f32x4{0.f, 0.f, 0.f, 0.f};

        if constexpr (NP == 2) {
            ea = EA_NONE;
            abound = 0;
        }
        for (int kc = 0; kc < p.K; kc += KC) {
            if constexpr (NP == 2) {                                 // this wave's largest magnitude of the chunk now parked in registers
                float m = 0.f;
#pragma unroll
                for (int i = 0; i < NST; ++i)
                    m = fmaxf(fmaxf(m, fmaxf(fabsf(stage[i][0]), fabsf(stage[i][1]))), fmaxf(fabsf(stage[i][2]), fabsf(stage[i][3])));
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
                if (lane == 0) smax[wave] = m;
            }
            __syncthreads();                                         // the previous chunk's (or tile's) LDS reads are done
            if constexpr (NP == 2) {
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
            if (!(FGCN_PROBE_PW & 4) || probe_first) deposit();
            __syncthreads();
            const bool last_chunk = kc + KC >= p.K;
            if (!last_chunk) {
                fetch(kc + KC);                                      // lands during the MFMAs below
            } else if (more) {                                       // ... or the next tile's first chunk, across this tile's store tail
                set_rows(bm_n);
                fetch(0);
            }
            const int nsteps = p.K - kc > 32 ? 2 : 1;                // a 32-channel tail runs one step
#pragma unroll 1
            for (int s2 = 0; s2 < nsteps; ++s2) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) load_a(a[mt], mt, s2);
                const bool last_step = s2 + 1 == nsteps;
#pragma unroll
                for (int nu = 0; nu < NU; ++nu) {
                    // the ring: the weights of the unit RS - 1 ahead -- a later column unit of this step, or a unit of the next step, the
                    // next chunk, or the next tile's first step
                    const int t = nu + RS - 1;
                    if (t < NU) load_w(wq[t % RS], bn, t, kc + 32 * s2);
                    else if (!last_step) load_w(wq[t % RS], bn, t - NU, kc + 32);
                    else if (!last_chunk) load_w(wq[t % RS], bn, t - NU, kc + KC);
                    else load_w(wq[t % RS], bn_n, t - NU, 0);
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) {
                        if constexpr ((FGCN_PROBE_PW & 2) != 0) {
                            acc[mt][nu][0] += __builtin_bit_cast(float, a[mt][0][0] ^ wq[nu % RS][0][0]);
                            continue;
                        }
                        if constexpr (NP == 3) acc[mt][nu] = mfma_x3_k32(a[mt], wq[nu % RS], acc[mt][nu]);
                        else if constexpr (NP == 2) acc[mt][nu] = mfma_h2_k32(a[mt], wq[nu % RS], acc[mt][nu]);
                        else acc[mt][nu] = mfma_bf16_k32(a[mt][0], wq[nu % RS][0], acc[mt][nu]);
                    }
                }
            }
            probe_first = false;
        }

        // ---- epilogue: bias, accumulate, branch-free buffer stores, BatchNorm partial sums (accumulator register r of lane
        // (col l15, g4) = row 4 g4 + r of its 16 x 16 tile) ------------------------------------------------------------------
        const long long m0 = (long long)bm * BMR;
        const int col = bn * BN + wc * NT * 32 + l15;
        const float un_a = (NP == 2 && ea != EA_NONE) ? exp2i(-ea) : 1.f, un_w = NP == 2 ? exp2i(-ew) : 1.f;   // (two factors: the sum of
        float ssum[NU], ssq[NU], bv[NU];                                                                        // the exponents may exceed 126)
        unsigned coff[NU];
#pragma unroll
        for (int nu = 0; nu < NU; ++nu) {
            ssum[nu] = 0.f;
            ssq[nu] = 0.f;
            coff[nu] = col + nu * 16 < p.N ? (unsigned)(col + nu * 16) * 4u : OOB;
            bv[nu] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbias, coff[nu], 0, 0));
        }
        // ACC: out += value by load / add / store, the old values of row tile mt + 1 requested BEFORE the stores of row tile mt: vmcnt
        // counts in issue order, so a load behind a store could only be waited for together with that store's write acknowledgement.
        // (One no-return float atomic per element -- every element has a single contributor, so it would be deterministic -- was
        // measured too: the L2 performs them at about one element per clock and channel, 5-40 % slower than this form.)
        // Addresses: one per-lane register per column unit (row 4 g4 of the wave's first tile + column); the row inside the wave's 64
        // travels in the instruction's scalar offset, which the hardware does not range-check: rows >= M (last row tile only) are masked
        // per lane.
        const unsigned ld_b = (unsigned)p.ld_out * 4u;
        const long long mrow0 = m0 + wr * (16 * MTW) + 4 * g4;
        unsigned rmask = 0;                                          // bit 4 mt + r: row mrow0 + 16 mt + r exists
#pragma unroll
        for (int i = 0; i < 4 * MTW; ++i) rmask |= (mrow0 + (i >> 2) * 16 + (i & 3) < p.M ? 1u : 0u) << i;
        unsigned cbase[NU];
#pragma unroll
        for (int nu = 0; nu < NU; ++nu) cbase[nu] = (coff[nu] == OOB || rmask == 0u) ? OOB : (unsigned)mrow0 * ld_b + coff[nu];
        const bool whole = __builtin_amdgcn_readfirstlane(m0 + BMR <= p.M ? 1 : 0) != 0;   // (uniform: every row of the tile exists)
        float old[ACC ? 2 : 1][NU][4];
        auto load_old = [&](int mt, float (&o)[NU][4]) {
#pragma unroll
            for (int nu = 0; nu < NU; ++nu)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    o[nu][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rout, (whole || ((rmask >> (4 * mt + r)) & 1u)) ? cbase[nu] : OOB, (unsigned)(mt * 16 + r) * ld_b, 0));
        };
        if constexpr (ACC) load_old(0, old[0]);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            if constexpr (ACC) {
                if (mt + 1 < MTW) load_old(mt + 1, old[(mt + 1) & 1]);
            }
            if constexpr (OUT16) {
                // two rows at a time: the even lane of a pair stores columns (c, c + 1) of row rp as one dword, the odd lane those of row rp + 1
                const bool odd = lane & 1;
#pragma unroll
                for (int rp = 0; rp < 4; rp += 2) {
#pragma unroll
                    for (int nu = 0; nu < NU; ++nu) {
                        const float v0 = acc[mt][nu][rp] + bv[nu], v1 = acc[mt][nu][rp + 1] + bv[nu];
                        const bool ok0 = cbase[nu] != OOB && (whole || ((rmask >> (4 * mt + rp)) & 1u));
                        const bool ok1 = cbase[nu] != OOB && (whole || ((rmask >> (4 * mt + rp + 1)) & 1u));
                        const float other = lane_xor1(odd ? v0 : v1);
                        const unsigned pk = odd ? pack_bf16x2(other, v1) : pack_bf16x2(v0, other);
                        const unsigned vo = (odd ? ok1 : ok0) ? ((cbase[nu] - (odd ? 4u : 0u)) >> 1) + (odd ? ld_b >> 1 : 0u) : OOB;
                        __builtin_amdgcn_raw_buffer_store_b32(pk, rout, vo, (unsigned)(mt * 16 + rp) * (ld_b >> 1), STR ? FGCN_STORE_AUX : 0);
                        const float k0 = ok0 ? v0 : 0.f, k1 = ok1 ? v1 : 0.f;
                        ssum[nu] += k0;
                        ssq[nu] += k0 * k0;
                        ssum[nu] += k1;
                        ssq[nu] += k1 * k1;
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
                    float val = (NP == 2 ? acc[mt][nu][r] * un_a * un_w : acc[mt][nu][r]) + bv[nu];
                    if constexpr (ACC) val += old[mt & 1][nu][r];
                    if ((FGCN_PROBE_PW & 1) && val != 123.456f) continue;
                    const unsigned vo = (whole || ((rmask >> (4 * mt + r)) & 1u)) ? cbase[nu] : OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rout, vo, (unsigned)(mt * 16 + r) * ld_b, STR ? FGCN_STORE_AUX : 0);
                    const float kept = vo != OOB ? val : 0.f;
                    ssum[nu] += kept;
                    ssq[nu] += kept * kept;
                }
            }
            if constexpr (ACC) __builtin_amdgcn_sched_barrier(0);    // (keep the request / store order as written)
        }
        if (p.stats) {                                               // (kernel-uniform)
            __syncthreads();                                         // every wave has left the tile's last MFMA step: the image is free
            float* red = Ah;                                         // [which][wr][BN]
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
                if (bn * BN + c < p.N)
                    p.stats[((long long)bm * 2 + which) * p.N + bn * BN + c] = red[(which * 2 + 0) * BN + c] + red[(which * 2 + 1) * BN + c];
            }
        }
        if (!more) break;
        vid = vnext;
        bm = bm_n;
        bn = bn_n;
    }
}

}  // namespace fgcn

using namespace fgcn;

extern "C" int fgcn_pw_gemm_tiles(long long rows) { return (int)cdiv(rows, 128); }

// 1 when fgcn_pw_gemm runs in the current math mode (the split-bf16 modes)
extern "C" int fgcn_pw_gemm_available(void) {
    const int mm = fgcn::math_mode();
    return (mm == FGCN_MATH_BF16X3 || mm == FGCN_MATH_BF16) ? 1 : 0;
}

static int pw_gemm_impl(const float* in, float* out, const void* w3, const float* bias, float* stat_partials, long long rows,
                        int K, int N, int ld_in, int ld_out, int accumulate, unsigned* in_amax, void* stream_, int io);

extern "C" int fgcn_pw_gemm(const float* in, float* out, const void* w3, const float* bias, float* stat_partials, long long rows,
                            int K, int N, int ld_in, int ld_out, int accumulate, unsigned* in_amax, void* stream_) {
    return pw_gemm_impl(in, out, w3, bias, stat_partials, rows, K, N, ld_in, ld_out, accumulate, in_amax, stream_, 0);
}

// typed form (math mode bf16, half-precision activation storage): half_mask bit 0 = `in` is a bfloat16 tensor, bit 1 = `out` is (not with
// accumulation); strides in elements; stat_partials: the moments of the float32 results
extern "C" int fgcn_pw_gemm_t(const void* in, void* out, const void* w3, const float* bias, float* stat_partials, long long rows,
                              int K, int N, int ld_in, int ld_out, int accumulate, int half_mask, void* stream_) {
    FGCN_REQUIRE((half_mask & ~3) == 0 && !((half_mask & 2) && accumulate), FGCN_E_BADARG, "pw_gemm_t: half_mask=%d (a bfloat16 output: no accumulation)",
                 half_mask);
    return pw_gemm_impl(static_cast<const float*>(in), static_cast<float*>(out), w3, bias, stat_partials, rows, K, N, ld_in, ld_out, accumulate,
                        nullptr, stream_, half_mask);
}

static int pw_gemm_impl(const float* in, float* out, const void* w3, const float* bias, float* stat_partials, long long rows,
                        int K, int N, int ld_in, int ld_out, int accumulate, unsigned* in_amax, void* stream_, int io) {
    FGCN_REQUIRE(in && out && w3 && rows > 0, FGCN_E_BADARG, "pw_gemm: null pointer or no rows");
    FGCN_REQUIRE(io == 0 || fgcn::math_mode() == FGCN_MATH_BF16, FGCN_E_BADARG, "pw_gemm_t: bfloat16 tensors need math mode bf16");
    FGCN_REQUIRE(fgcn_pw_gemm_available(), FGCN_E_BADARG, "pw_gemm: a split-bf16 math mode (bf16x3 / bf16) only");
    FGCN_REQUIRE(K > 0 && K % 32 == 0 && N > 0 && N % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0 && ld_in >= K && ld_out >= N,
                 FGCN_E_ALIGN, "pw_gemm: K must be a multiple of 32, N and the row strides multiples of 4 (K=%d N=%d ld_in=%d ld_out=%d)", K,
                 N, ld_in, ld_out);
    FGCN_REQUIRE(aligned16(in) && aligned16(w3) && aligned16(out), FGCN_E_ALIGN, "pw_gemm: 16-byte alignment");
    const long long in_bytes = rows * ld_in * ((io & 1) ? 2 : 4), out_bytes = rows * ld_out * ((io & 2) ? 2 : 4), plane = (long long)K * N * 2;
    FGCN_REQUIRE(in_bytes < 0x7FFF0000ll && out_bytes < 0x7FFF0000ll && plane * 3 < 0x7FFF0000ll, FGCN_E_BADARG,
                 "pw_gemm: tensors must be smaller than 2 GiB (32-bit buffer offsets)");
    PwP p;
    p.in = in; p.out = out; p.w3 = w3; p.bias = bias; p.stats = stat_partials;
    p.M = rows;
    p.in_bytes = (unsigned)in_bytes; p.out_bytes = (unsigned)out_bytes; p.w_plane_bytes = (unsigned)plane;
    p.K = K; p.N = N; p.ld_in = ld_in; p.ld_out = ld_out; p.accumulate = accumulate;
    p.in_amax = fgcn::f16x2_products() ? in_amax : nullptr;
    const bool narrow = N <= 64;
    p.tiles_m = (int)cdiv(rows, 128);
    p.tiles_n = (int)cdiv(N, narrow ? 64 : 128);
    const long long total = (long long)p.tiles_m * p.tiles_n;
    FGCN_REQUIRE(total < (1ll << 30), FGCN_E_BADARG, "pw_gemm: too many tiles");
    p.per_xcd = (int)cdiv(total, 8);
    // two workgroups per CU (48 KB of LDS, <= 256 registers): 64 per XCD fill the chip; fewer when there are fewer tiles.  key 8: tiles
    // per workgroup cap (0 = persistent; 1 = one tile per workgroup, the non-persistent control of the A/B)
    int per = 64;
    if (fgcn::tuning(8) == 1) per = p.per_xcd;
    p.wg_per_xcd = p.per_xcd < per ? p.per_xcd : per;
    const dim3 grid((unsigned)(p.wg_per_xcd * 8));
    const bool one = fgcn::math_mode() == FGCN_MATH_BF16, two = fgcn::f16x2_products();
    const size_t lds = (size_t)128 * 128 * (one ? 1 : (two ? 2 : 3)) + 16;
    hipStream_t s = (hipStream_t)stream_;
    const bool stream = !(io & 2) && fgcn::stream_out(rows * (long long)N * 4);
#define FGCN_PW_LAUNCH(NT_, NP_)                                                                                         \
    do {                                                                                                                 \
        if (accumulate) hipLaunchKernelGGL((pw_x3_kernel<NT_, NP_, true>), grid, dim3(256), lds, s, p);                  \
        else if (stream) hipLaunchKernelGGL((pw_x3_kernel<NT_, NP_, false, true>), grid, dim3(256), lds, s, p);         \
        else hipLaunchKernelGGL((pw_x3_kernel<NT_, NP_, false>), grid, dim3(256), lds, s, p);                            \
    } while (0)
#define FGCN_PW_LAUNCH_T(NT_)       /* one part, typed tensors */                                                        \
    do {                                                                                                                 \
        if (accumulate) hipLaunchKernelGGL((pw_x3_kernel<NT_, 1, true, false, 1>), grid, dim3(256), lds, s, p);          \
        else if (io == 1 && stream) hipLaunchKernelGGL((pw_x3_kernel<NT_, 1, false, true, 1>), grid, dim3(256), lds, s, p); \
        else if (io == 1) hipLaunchKernelGGL((pw_x3_kernel<NT_, 1, false, false, 1>), grid, dim3(256), lds, s, p);       \
        else if (io == 2) hipLaunchKernelGGL((pw_x3_kernel<NT_, 1, false, false, 2>), grid, dim3(256), lds, s, p);       \
        else hipLaunchKernelGGL((pw_x3_kernel<NT_, 1, false, false, 3>), grid, dim3(256), lds, s, p);                    \
    } while (0)
    if (io) {
        FGCN_REQUIRE(one && !(accumulate && io != 1), FGCN_E_BADARG, "pw_gemm_t: an accumulating call takes a bfloat16 input only");
        if (narrow) FGCN_PW_LAUNCH_T(1);
        else FGCN_PW_LAUNCH_T(2);
        return launch_status("pw_gemm");
    }
    if (narrow) {
        if (one) FGCN_PW_LAUNCH(1, 1);
        else if (two) FGCN_PW_LAUNCH(1, 2);
        else FGCN_PW_LAUNCH(1, 3);
    } else {
        if (one) FGCN_PW_LAUNCH(2, 1);
        else if (two) FGCN_PW_LAUNCH(2, 2);
        else FGCN_PW_LAUNCH(2, 3);
    }
#undef FGCN_PW_LAUNCH
#undef FGCN_PW_LAUNCH_T
    return launch_status("pw_gemm");
}
