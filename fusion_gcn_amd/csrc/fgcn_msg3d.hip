// Data-movement kernels of the MS-G3D blocks (reference torch_src/models/msg3d/ms_tcn.py:72-78,
// ms_gtcn.py:24-45): the (3 x 1) temporal max pooling of MultiScale_TemporalConv's pooling branch and the temporal-window unfold
// in front of the spatial-temporal graph convolution.  Channels-last (B, T, V, C) like every other kernel; pure HBM streams of
// 16-byte loads; the backward passes are GATHERS (every input element sums the few outputs it fed), so no atomics and bitwise
// reproducible results.  The arithmetic of those blocks (1x1 / dilated convolutions, aggregation, BatchNorm) runs on the
// existing GEMM / BatchNorm kernels.
#include "fgcn_common.hpp"

namespace fgcn {

struct PoolP {
    const float* in;       // (B, T_in, V, ld_in), channel window [0, C) of the pointer handed in
    float* out;            // (B, T_out, V, C) contiguous
    unsigned char* idx;    // (B, T_out, V, C): which of the three taps won (0, 1, 2)
    long long n4;          // B * T_out * V * C / 4
    int T_in, T_out, V, C, ld_in, stride;
};

// out[b, to, v, c] = max_{j in 0..2, 0 <= ti < T_in} in[b, ti = to*stride + j - 1, v, c]; the FIRST maximal tap wins (torch's
// max_pool2d keeps the first maximum it meets), NaN propagates like torch's (a NaN tap wins).
__global__ __launch_bounds__(256) void tmaxpool3_fwd_kernel(PoolP p) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n4; i += (long long)gridDim.x * blockDim.x) {
        const unsigned e = (unsigned)i * 4u;                       // n4 < 2^29 (host check)
        const unsigned c = e % (unsigned)p.C, row = e / (unsigned)p.C;
        const unsigned v = row % (unsigned)p.V, bt = row / (unsigned)p.V;
        const unsigned to = bt % (unsigned)p.T_out, b = bt / (unsigned)p.T_out;
        f32x4 best = {0.f, 0.f, 0.f, 0.f};
        int which[4] = {0, 0, 0, 0};
        bool have = false;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ti = (int)to * p.stride + j - 1;
            if (ti < 0 || ti >= p.T_in) continue;
            const f32x4 x = *reinterpret_cast<const f32x4*>(p.in + (((long long)b * p.T_in + ti) * p.V + v) * p.ld_in + c);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (!have || x[q] > best[q] || x[q] != x[q]) {
                    best[q] = x[q];
                    which[q] = j;
                }
            have = true;
        }
        *reinterpret_cast<f32x4*>(p.out + i * 4) = best;
        *reinterpret_cast<unsigned*>(p.idx + i * 4) = (unsigned)which[0] | ((unsigned)which[1] << 8) | ((unsigned)which[2] << 16) |
                                                      ((unsigned)which[3] << 24);
    }
}

struct PoolBP {
    const float* dout;     // (B, T_out, V, C)
    const unsigned char* idx;
    float* din;            // (B, T_in, V, ld_in) window [0, C)
    long long n4;          // B * T_in * V * C / 4
    int T_in, T_out, V, C, ld_in, stride, accumulate;
};

// din[b, ti, v, c] (+)= sum over the outputs `to` whose window holds ti (to*stride - 1 <= ti <= to*stride + 1) and whose winning tap
// is ti - (to*stride - 1)
__global__ __launch_bounds__(256) void tmaxpool3_bwd_kernel(PoolBP p) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n4; i += (long long)gridDim.x * blockDim.x) {
        const unsigned e = (unsigned)i * 4u;
        const unsigned c = e % (unsigned)p.C, row = e / (unsigned)p.C;
        const unsigned v = row % (unsigned)p.V, bt = row / (unsigned)p.V;
        const int ti = (int)(bt % (unsigned)p.T_in);
        const unsigned b = bt / (unsigned)p.T_in;
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        // candidates: to in [ceil((ti - 1) / stride), floor((ti + 1) / stride)]
        const int lo = ti - 1 <= 0 ? 0 : (ti - 1 + p.stride - 1) / p.stride;
        const int hi = min((ti + 1) / p.stride, p.T_out - 1);
        for (int to = lo; to <= hi; ++to) {
            const int j = ti - (to * p.stride - 1);
            const long long o = (((long long)b * p.T_out + to) * p.V + v) * p.C + c;
            const f32x4 d = *reinterpret_cast<const f32x4*>(p.dout + o);
            const unsigned w = *reinterpret_cast<const unsigned*>(p.idx + o);
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] += (int)((w >> (8 * q)) & 255u) == j ? d[q] : 0.f;
        }
        float* dst = p.din + (((long long)b * p.T_in + ti) * p.V + v) * p.ld_in + c;
        if (p.accumulate) g += *reinterpret_cast<const f32x4*>(dst);
        *reinterpret_cast<f32x4*>(dst) = g;
    }
}

struct UnfoldP {
    const float* in;       // forward: x (B, T, V, C); backward: d(out) (B, T_out, W*V, C)
    float* out;            // forward: (B, T_out, W*V, C); backward: dx (B, T, V, C)
    long long n4;
    int T, T_out, V, C, window, stride, dilation, pad;
};

// out[b, to, j*V + v, c] = x[b, to*stride + j*dilation - pad, v, c]  (zero outside [0, T))
__global__ __launch_bounds__(256) void unfold_fwd_kernel(UnfoldP p) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n4; i += (long long)gridDim.x * blockDim.x) {
        const unsigned long long e = (unsigned long long)i * 4ull;
        const unsigned c = (unsigned)(e % (unsigned)p.C);
        const unsigned long long row = e / (unsigned)p.C;           // (b, to, j, v)
        const unsigned v = (unsigned)(row % (unsigned)p.V);
        const unsigned long long r2 = row / (unsigned)p.V;
        const int j = (int)(r2 % (unsigned)p.window);
        const unsigned long long bto = r2 / (unsigned)p.window;
        const int to = (int)(bto % (unsigned)p.T_out);
        const long long b = (long long)(bto / (unsigned)p.T_out);
        const int ti = to * p.stride + j * p.dilation - p.pad;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (ti >= 0 && ti < p.T) x = *reinterpret_cast<const f32x4*>(p.in + ((b * p.T + ti) * p.V + v) * p.C + c);
        *reinterpret_cast<f32x4*>(p.out + i * 4) = x;
    }
}

// dx[b, t, v, c] = sum over (to, j) with to*stride + j*dilation - pad == t of d(out)[b, to, j*V + v, c]
__global__ __launch_bounds__(256) void unfold_bwd_kernel(UnfoldP p) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n4; i += (long long)gridDim.x * blockDim.x) {
        const unsigned long long e = (unsigned long long)i * 4ull;
        const unsigned c = (unsigned)(e % (unsigned)p.C);
        const unsigned long long row = e / (unsigned)p.C;           // (b, t, v)
        const unsigned v = (unsigned)(row % (unsigned)p.V);
        const unsigned long long bt = row / (unsigned)p.V;
        const int t = (int)(bt % (unsigned)p.T);
        const long long b = (long long)(bt / (unsigned)p.T);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < p.window; ++j) {
            const int num = t + p.pad - j * p.dilation;
            if (num < 0 || num % p.stride) continue;
            const int to = num / p.stride;
            if (to >= p.T_out) continue;
            g += *reinterpret_cast<const f32x4*>(p.in + (((b * p.T_out + to) * p.window + j) * p.V + v) * p.C + c);
        }
        *reinterpret_cast<f32x4*>(p.out + i * 4) = g;
    }
}

static unsigned msg3d_blocks(long long n4) {
    const long long b = cdiv(n4, 256);
    return (unsigned)(b < (1 << 22) ? b : (1 << 22));   // one 16-byte group per thread (fgcn_elem.hip stream_blocks: a grid-stride walk waits, per trip, behind its own previous store)
}

}  // namespace fgcn

using namespace fgcn;

extern "C" int fgcn_tmaxpool3_fwd(const float* in, float* out, unsigned char* idx, int B, int T_in, int T_out, int V, int C,
                                  int ld_in, int stride, void* stream) {
    FGCN_REQUIRE(in && out && idx && B > 0 && T_in > 0 && V > 0 && C > 0 && C % 4 == 0 && ld_in % 4 == 0 && ld_in >= C &&
                     stride >= 1, FGCN_E_BADARG, "tmaxpool3_fwd: bad argument (B=%d T=%d V=%d C=%d ld=%d stride=%d)", B, T_in, V, C,
                 ld_in, stride);
    FGCN_REQUIRE(T_out == (T_in - 1) / stride + 1, FGCN_E_BADARG, "tmaxpool3_fwd: T_out must be %d", (T_in - 1) / stride + 1);
    FGCN_REQUIRE(aligned16(in) && aligned16(out) && ((uintptr_t)idx & 3u) == 0, FGCN_E_ALIGN, "tmaxpool3_fwd: alignment");
    PoolP p{in, out, idx, (long long)B * T_out * V * C / 4, T_in, T_out, V, C, ld_in, stride};
    FGCN_REQUIRE(p.n4 < (1ll << 29), FGCN_E_BADARG, "tmaxpool3_fwd: tensor too large");
    hipLaunchKernelGGL(tmaxpool3_fwd_kernel, dim3(msg3d_blocks(p.n4)), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status("tmaxpool3_fwd");
}

extern "C" int fgcn_tmaxpool3_bwd(const float* dout, const unsigned char* idx, float* din, int B, int T_in, int T_out, int V,
                                  int C, int ld_in, int stride, int accumulate, void* stream) {
    FGCN_REQUIRE(dout && idx && din && B > 0 && T_in > 0 && V > 0 && C > 0 && C % 4 == 0 && ld_in % 4 == 0 && ld_in >= C &&
                     stride >= 1 && T_out == (T_in - 1) / stride + 1, FGCN_E_BADARG, "tmaxpool3_bwd: bad argument");
    FGCN_REQUIRE(aligned16(dout) && aligned16(din) && ((uintptr_t)idx & 3u) == 0, FGCN_E_ALIGN, "tmaxpool3_bwd: alignment");
    PoolBP p{dout, idx, din, (long long)B * T_in * V * C / 4, T_in, T_out, V, C, ld_in, stride, accumulate};
    FGCN_REQUIRE(p.n4 < (1ll << 29), FGCN_E_BADARG, "tmaxpool3_bwd: tensor too large");
    hipLaunchKernelGGL(tmaxpool3_bwd_kernel, dim3(msg3d_blocks(p.n4)), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status("tmaxpool3_bwd");
}

extern "C" int fgcn_unfold_windows(const float* in, float* out, int B, int T, int T_out, int V, int C, int window, int stride,
                                   int dilation, int backward, void* stream) {
    FGCN_REQUIRE(in && out && B > 0 && T > 0 && V > 0 && C > 0 && C % 4 == 0 && window >= 1 && stride >= 1 && dilation >= 1,
                 FGCN_E_BADARG, "unfold_windows: bad argument");
    const int pad = (window + (window - 1) * (dilation - 1) - 1) / 2;
    FGCN_REQUIRE(T_out == (T + 2 * pad - dilation * (window - 1) - 1) / stride + 1, FGCN_E_BADARG,
                 "unfold_windows: T_out must be %d", (T + 2 * pad - dilation * (window - 1) - 1) / stride + 1);
    FGCN_REQUIRE(aligned16(in) && aligned16(out), FGCN_E_ALIGN, "unfold_windows: 16-byte alignment");
    UnfoldP p{in, out, 0, T, T_out, V, C, window, stride, dilation, pad};
    p.n4 = backward ? (long long)B * T * V * C / 4 : (long long)B * T_out * window * V * C / 4;
    if (backward) hipLaunchKernelGGL(unfold_bwd_kernel, dim3(msg3d_blocks(p.n4)), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(unfold_fwd_kernel, dim3(msg3d_blocks(p.n4)), dim3(256), 0, (hipStream_t)stream, p);
    return launch_status("unfold_windows");
}
