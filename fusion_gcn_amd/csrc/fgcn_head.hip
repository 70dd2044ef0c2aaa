// The two ends of the step around the ten blocks: the BatchNorm of the network input (`data_bn`, reference
// torch_src/models/mmargcn/agcn.py:150,186-188) and the CrossEntropy loss (session/session.py:53, step.py:38-46).
// Both are tiny (11.5 MB of input, a 64 x 60 logit matrix) -- they are here so that EVERY gradient of the step is a fixed-order
// libfgcn sum (MIOpen's BatchNorm backward was the one gradient of 274 that was not bitwise reproducible).
//
// data_bn = nn.BatchNorm1d(M*V*C) over x (N, M, T, V, C) viewed as (N, M*V*C, T): channel ch = (m*V + v)*C + c, statistics over
// (n, t).  x is already the blocks' (B = N*M, T, V, C) layout, so the apply pass writes the block input (B, T, V, Cp) directly,
// zero pad channels included; no permute / view / pad launches.
#include "fgcn_common.hpp"

namespace fgcn {

constexpr int DBN_T_CHUNK = 32;     // frames per tile: N * ceil(T / 32) tiles (640 at the headline shape)

// partials[tile][0][ch] = sum x, [tile][1][ch] = sum x^2 over the tile's frames of clip n; thread <-> channel, so consecutive
// threads of one body m read consecutive floats of a frame row
__global__ __launch_bounds__(256) void data_bn_stats_kernel(const float* x, float* partials, int M, int T, int VC, int chunks) {
    const int n = blockIdx.x / chunks, t0 = (blockIdx.x % chunks) * DBN_T_CHUNK;
    const int t1 = min(t0 + DBN_T_CHUNK, T), MVC = M * VC;
    for (int ch = threadIdx.x; ch < MVC; ch += blockDim.x) {
        const int m = ch / VC, vc = ch - m * VC;
        const float* p = x + ((long long)(n * M + m) * T + t0) * VC + vc;
        float s1 = 0.f, s2 = 0.f;
        for (int t = t0; t < t1; ++t, p += VC) {
            const float v = *p;
            s1 += v;
            s2 += v * v;
        }
        partials[((long long)blockIdx.x * 2 + 0) * MVC + ch] = s1;
        partials[((long long)blockIdx.x * 2 + 1) * MVC + ch] = s2;
    }
}

// one thread per (b, t, v) row: out[row][0..C) = x[row][c] * scale[ch] + shift[ch], out[row][C..Cp) = 0
__global__ __launch_bounds__(256) void data_bn_apply_kernel(const float* x, const float* vec, float* out, long long rows, int M,
                                                            int T, int V, int C, int Cp) {
    const int MVC = M * V * C;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(r % V);
        const int m = (int)((r / ((long long)T * V)) % M);
        const int ch0 = (m * V + v) * C;
        for (int c = 0; c < Cp; ++c)
            out[r * Cp + c] = c < C ? x[r * C + c] * vec[2 * MVC + ch0 + c] + vec[3 * MVC + ch0 + c] : 0.f;
    }
}

// partials[tile][0][ch] = sum dout, [tile][1][ch] = sum dout * (x - mean) * rstd  (d beta, d gamma)
__global__ __launch_bounds__(256) void data_bn_bwd_reduce_kernel(const float* dout, const float* x, const float* vec, float* partials,
                                                                 int M, int T, int V, int C, int Cp, int chunks) {
    const int VC = V * C, MVC = M * VC;
    const int n = blockIdx.x / chunks, t0 = (blockIdx.x % chunks) * DBN_T_CHUNK;
    const int t1 = min(t0 + DBN_T_CHUNK, T);
    for (int ch = threadIdx.x; ch < MVC; ch += blockDim.x) {
        const int m = ch / VC, vc = ch - m * VC, v = vc / C, c = vc - v * C;
        const float mean = vec[ch], rstd = vec[MVC + ch];
        float s1 = 0.f, s2 = 0.f;
        for (int t = t0; t < t1; ++t) {
            const long long row = ((long long)(n * M + m) * T + t) * V + v;
            const float d = dout[row * Cp + c];
            s1 += d;
            s2 += d * ((x[row * C + c] - mean) * rstd);
        }
        partials[((long long)blockIdx.x * 2 + 0) * MVC + ch] = s1;
        partials[((long long)blockIdx.x * 2 + 1) * MVC + ch] = s2;
    }
}

// dx = scale * (dout - sum_d / m - xhat * sum_dxhat / m)  (train)  |  scale * dout  (eval)
__global__ __launch_bounds__(256) void data_bn_bwd_apply_kernel(const float* dout, const float* x, const float* vec, const float* sums,
                                                                float* dx, long long rows, int M, int T, int V, int C, int Cp,
                                                                int train, float inv_m) {
    const int MVC = M * V * C;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(r % V);
        const int m = (int)((r / ((long long)T * V)) % M);
        const int ch0 = (m * V + v) * C;
        for (int c = 0; c < C; ++c) {
            const int ch = ch0 + c;
            float g = dout[r * Cp + c];
            if (train) {
                const float xh = (x[r * C + c] - vec[ch]) * vec[MVC + ch];
                g = g - sums[ch] * inv_m - xh * (sums[MVC + ch] * inv_m);
            }
            dx[r * C + c] = g * vec[2 * MVC + ch];
        }
    }
}

// ---- CrossEntropyLoss, mean reduction -------------------------------------------------------------------------------------------
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// ONE workgroup (the row count is the clip batch): wave w takes rows w, w + 4, ...; probs = softmax(logits) is kept for the
// backward, row_loss[i] = logsumexp - logit[label]; rows labelled -100 (torch's ignore_index) do not count; any OTHER label outside
// [0, classes) -- torch raises a device assert there -- makes the row's loss, hence the batch loss and every gradient, NaN: a corrupt
// label fails loudly instead of being dropped from the mean;
// loss[0] = sum row_loss / n_valid, loss[1] = n_valid -- summed by wave 0 in a fixed order (lane l: rows l, l + 64, ...; then the
// butterfly): bitwise reproducible.
__global__ __launch_bounds__(256) void cross_entropy_fwd_kernel(const float* logits, const long long* labels, float* probs,
                                                                float* row_loss, float* loss, int rows, int classes, int ld) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = wave; i < rows; i += 4) {
        const float* z = logits + (long long)i * ld;
        float mx = -INFINITY;
        for (int c = lane; c < classes; c += 64) mx = fmaxf(mx, z[c]);
        mx = wmax(mx);
        float s = 0.f;
        for (int c = lane; c < classes; c += 64) s += expf(z[c] - mx);
        s = wsum(s);
        const float inv = 1.f / s;
        for (int c = lane; c < classes; c += 64) probs[(long long)i * classes + c] = expf(z[c] - mx) * inv;
        if (lane == 0) {
            const long long y = labels[i];
            row_loss[i] = (y >= 0 && y < classes) ? (mx + logf(s)) - z[y] : (y == -100 ? 0.f : NAN);
        }
    }
    __syncthreads();
    if (wave == 0) {
        float a = 0.f, cnt = 0.f;
        for (int i = lane; i < rows; i += 64) {
            const long long y = labels[i];
            const bool ok = y != -100;
            a += ok ? row_loss[i] : 0.f;
            cnt += ok ? 1.f : 0.f;
        }
        a = wsum(a);
        cnt = wsum(cnt);
        if (lane == 0) {
            loss[0] = cnt > 0.f ? a / cnt : NAN;      // torch: mean over no rows is NaN
            loss[1] = cnt;
        }
    }
}

// dlogits[i][c] = (probs[i][c] - [c == label_i]) * dloss / n_valid   (0 for ignored rows); columns [classes, ld_out) zero
__global__ __launch_bounds__(256) void cross_entropy_bwd_kernel(const float* probs, const long long* labels, const float* loss,
                                                                const float* dloss, float* dlogits, int rows, int classes,
                                                                int ld_out) {
    const float g = dloss[0] / loss[1];
    const long long n = (long long)rows * ld_out;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(e / ld_out), c = (int)(e - (long long)i * ld_out);
        const long long y = labels[i];
        float d = 0.f;
        if (c < classes && y >= 0 && y < classes) d = (probs[(long long)i * classes + c] - (c == y ? 1.f : 0.f)) * g;
        else if (c < classes && y != -100) d = NAN;             // a label outside [0, classes) that is not ignore_index
        dlogits[e] = d;
    }
}

}  // namespace fgcn

using namespace fgcn;

static unsigned stream_blocks(long long n) {
    const long long b = cdiv(n, 256);
    return (unsigned)(b < 4096 ? b : 4096);
}

extern "C" int fgcn_data_bn_tiles(int N, int T) { return N * (int)cdiv(T, DBN_T_CHUNK); }

static int check_dbn(const char* what, int N, int M, int T, int V, int C, int Cp) {
    FGCN_REQUIRE(N > 0 && M > 0 && T > 0 && V > 0 && C > 0 && Cp >= C, FGCN_E_BADARG, "%s: bad shape N=%d M=%d T=%d V=%d C=%d Cp=%d", what,
                 N, M, T, V, C, Cp);
    FGCN_REQUIRE((long long)N * M * T * V * Cp < (1ll << 31) && (long long)N * cdiv(T, DBN_T_CHUNK) < (1ll << 31), FGCN_E_BADARG,
                 "%s: tensor too large", what);
    return FGCN_OK;
}

extern "C" int fgcn_data_bn_stats(const float* x, float* partials, int N, int M, int T, int V, int C, void* stream) {
    FGCN_REQUIRE(x && partials, FGCN_E_BADARG, "data_bn_stats: null pointer");
    if (int e = check_dbn("data_bn_stats", N, M, T, V, C, C)) return e;
    const int chunks = (int)cdiv(T, DBN_T_CHUNK);
    hipLaunchKernelGGL(data_bn_stats_kernel, dim3((unsigned)(N * chunks)), dim3(256), 0, (hipStream_t)stream, x, partials, M, T, V * C,
                       chunks);
    return launch_status("data_bn_stats");
}

extern "C" int fgcn_data_bn_apply(const float* x, const float* vec, float* out, int N, int M, int T, int V, int C, int Cp,
                                  void* stream) {
    FGCN_REQUIRE(x && vec && out, FGCN_E_BADARG, "data_bn_apply: null pointer");
    if (int e = check_dbn("data_bn_apply", N, M, T, V, C, Cp)) return e;
    const long long rows = (long long)N * M * T * V;
    hipLaunchKernelGGL(data_bn_apply_kernel, dim3(stream_blocks(rows)), dim3(256), 0, (hipStream_t)stream, x, vec, out, rows, M, T, V, C,
                       Cp);
    return launch_status("data_bn_apply");
}

extern "C" int fgcn_data_bn_bwd_reduce(const float* dout, const float* x, const float* vec, float* partials, int N, int M, int T,
                                       int V, int C, int Cp, void* stream) {
    FGCN_REQUIRE(dout && x && vec && partials, FGCN_E_BADARG, "data_bn_bwd_reduce: null pointer");
    if (int e = check_dbn("data_bn_bwd_reduce", N, M, T, V, C, Cp)) return e;
    const int chunks = (int)cdiv(T, DBN_T_CHUNK);
    hipLaunchKernelGGL(data_bn_bwd_reduce_kernel, dim3((unsigned)(N * chunks)), dim3(256), 0, (hipStream_t)stream, dout, x, vec, partials,
                       M, T, V, C, Cp, chunks);
    return launch_status("data_bn_bwd_reduce");
}

extern "C" int fgcn_data_bn_bwd_apply(const float* dout, const float* x, const float* vec, const float* sums, float* dx, int N, int M,
                                      int T, int V, int C, int Cp, int train, void* stream) {
    FGCN_REQUIRE(dout && vec && dx && (!train || (x && sums)), FGCN_E_BADARG, "data_bn_bwd_apply: null pointer");
    if (int e = check_dbn("data_bn_bwd_apply", N, M, T, V, C, Cp)) return e;
    const long long rows = (long long)N * M * T * V;
    hipLaunchKernelGGL(data_bn_bwd_apply_kernel, dim3(stream_blocks(rows)), dim3(256), 0, (hipStream_t)stream, dout, x, vec, sums, dx,
                       rows, M, T, V, C, Cp, train, 1.f / (float)((long long)N * T));
    return launch_status("data_bn_bwd_apply");
}

extern "C" int fgcn_cross_entropy_fwd(const float* logits, const long long* labels, float* probs, float* row_loss, float* loss,
                                      int rows, int classes, int ld, void* stream) {
    FGCN_REQUIRE(logits && labels && probs && row_loss && loss && rows > 0 && classes > 0 && ld >= classes, FGCN_E_BADARG,
                 "cross_entropy_fwd: bad argument (rows=%d classes=%d ld=%d)", rows, classes, ld);
    hipLaunchKernelGGL(cross_entropy_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, probs, row_loss, loss, rows,
                       classes, ld);
    return launch_status("cross_entropy_fwd");
}

extern "C" int fgcn_cross_entropy_bwd(const float* probs, const long long* labels, const float* loss, const float* dloss,
                                      float* dlogits, int rows, int classes, int ld_out, void* stream) {
    FGCN_REQUIRE(probs && labels && loss && dloss && dlogits && rows > 0 && classes > 0 && ld_out >= classes, FGCN_E_BADARG,
                 "cross_entropy_bwd: bad argument (rows=%d classes=%d ld_out=%d)", rows, classes, ld_out);
    hipLaunchKernelGGL(cross_entropy_bwd_kernel, dim3(stream_blocks((long long)rows * ld_out)), dim3(256), 0, (hipStream_t)stream, probs,
                       labels, loss, dloss, dlogits, rows, classes, ld_out);
    return launch_status("cross_entropy_bwd");
}
