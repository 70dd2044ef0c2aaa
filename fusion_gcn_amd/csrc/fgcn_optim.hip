// Parameter update over the flat parameter / gradient buffers: the step after the hot path.
// The reference builds torch.optim.{SGD, Adam, AdamW} over model.parameters() (torch_src/session_helper.py:48-53,80-84; ADAM
// with weight_decay 0.01 in config/utd-mhad/skeleton/agcn.yaml:16-18) and calls optimizer.step() once per batch
// (session/session.py:176-183): 274 tensors, i.e. ~1000 small launches per step.  Here every trainable value of the model
// lives in one contiguous float32 buffer (fusion_gcn_amd/optim.py; the gradients already do, dp.FlatGradients), and one
// launch applies torch's update formulas element by element -- the data-parallel 1/world average rides along as
// `grad_scale`.  Pure HBM stream: 16-byte loads / stores, 28 B per parameter (Adam).
#include <cmath>

#include "fgcn_common.hpp"

namespace fgcn {

struct OptimP {
    float* p;
    const float* g;
    float* m;     // SGD: momentum buffer; Adam: exp_avg
    float* v;     // Adam: exp_avg_sq
    long long n4;
    float lr, wd, grad_scale;
    float beta1, beta2, eps, step_size, bc2_sqrt;      // Adam / AdamW
    float momentum, dampening;                         // SGD
    int nesterov, first_step;
};

// kind 0: SGD (torch/optim/sgd.py), 1: Adam (L2 weight decay folded into the gradient), 2: AdamW (decoupled decay)
template <int KIND>
__global__ __launch_bounds__(256) void optim_step_kernel(OptimP q) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < q.n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 p = *reinterpret_cast<const f32x4*>(q.p + i * 4);
        f32x4 g = *reinterpret_cast<const f32x4*>(q.g + i * 4) * q.grad_scale;
        if (KIND == 0) {
            if (q.wd != 0.f) g += p * q.wd;
            if (q.momentum != 0.f) {
                f32x4 buf = g;                                           // first step: buf = clone(d_p)
                if (!q.first_step) buf = *reinterpret_cast<const f32x4*>(q.m + i * 4) * q.momentum + g * (1.f - q.dampening);
                *reinterpret_cast<f32x4*>(q.m + i * 4) = buf;
                g = q.nesterov ? g + buf * q.momentum : buf;
            }
            p -= g * q.lr;
        } else {
            if (KIND == 1 && q.wd != 0.f) g += p * q.wd;
            if (KIND == 2) p *= 1.f - q.lr * q.wd;
            f32x4 m = *reinterpret_cast<const f32x4*>(q.m + i * 4);
            f32x4 v = *reinterpret_cast<const f32x4*>(q.v + i * 4);
            m += (g - m) * (1.f - q.beta1);                              // exp_avg.lerp_(grad, 1 - beta1)
            v = v * q.beta2 + g * g * (1.f - q.beta2);                   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
            *reinterpret_cast<f32x4*>(q.m + i * 4) = m;
            *reinterpret_cast<f32x4*>(q.v + i * 4) = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float denom = __fsqrt_rn(v[e]) / q.bc2_sqrt + q.eps;
                p[e] -= q.step_size * (m[e] / denom);
            }
        }
        *reinterpret_cast<f32x4*>(q.p + i * 4) = p;
    }
}

}  // namespace fgcn

using namespace fgcn;

extern "C" int fgcn_optim_step(float* params, const float* grads, float* state1, float* state2, long long n, int kind,
                               float lr, float weight_decay, float grad_scale, float beta1, float beta2, float eps,
                               float momentum, float dampening, int nesterov, long long step, void* stream) {
    FGCN_REQUIRE(params && grads && n > 0, FGCN_E_BADARG, "optim_step: null pointer or empty buffer");
    FGCN_REQUIRE(n % 4 == 0 && aligned16(params) && aligned16(grads), FGCN_E_ALIGN,
                 "optim_step: buffers must be 16-byte aligned and a multiple of 4 floats long (n=%lld)", n);
    FGCN_REQUIRE(kind >= FGCN_OPT_SGD && kind <= FGCN_OPT_ADAMW, FGCN_E_BADARG, "optim_step: kind %d", kind);
    FGCN_REQUIRE(step >= 1, FGCN_E_BADARG, "optim_step: step counts from 1 (got %lld)", step);
    FGCN_REQUIRE(lr >= 0.f && weight_decay >= 0.f, FGCN_E_BADARG, "optim_step: negative lr / weight_decay");
    OptimP q{};
    q.p = params; q.g = grads; q.m = state1; q.v = state2; q.n4 = n / 4;
    q.lr = lr; q.wd = weight_decay; q.grad_scale = grad_scale;
    if (kind == FGCN_OPT_SGD) {
        FGCN_REQUIRE(momentum >= 0.f && (momentum == 0.f || (state1 && aligned16(state1))), FGCN_E_BADARG,
                     "optim_step: SGD with momentum needs the momentum buffer");
        FGCN_REQUIRE(!nesterov || (momentum > 0.f && dampening == 0.f), FGCN_E_BADARG,
                     "optim_step: Nesterov momentum requires a momentum and zero dampening");
        q.momentum = momentum; q.dampening = dampening; q.nesterov = nesterov; q.first_step = step == 1;
    } else {
        FGCN_REQUIRE(state1 && state2 && aligned16(state1) && aligned16(state2), FGCN_E_BADARG,
                     "optim_step: Adam needs exp_avg and exp_avg_sq");
        FGCN_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, FGCN_E_BADARG,
                     "optim_step: betas / eps out of range");
        // the bias corrections in double, as torch's Python scalars
        const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
        q.beta1 = beta1; q.beta2 = beta2; q.eps = eps;
        q.step_size = (float)((double)lr / bc1);
        q.bc2_sqrt = (float)std::sqrt(bc2);
    }
    const long long blocks = cdiv(q.n4, 256);
    dim3 grid((unsigned)(blocks < 4096 ? blocks : 4096));
    hipStream_t s = (hipStream_t)stream;
    if (kind == FGCN_OPT_SGD) hipLaunchKernelGGL((optim_step_kernel<0>), grid, dim3(256), 0, s, q);
    else if (kind == FGCN_OPT_ADAM) hipLaunchKernelGGL((optim_step_kernel<1>), grid, dim3(256), 0, s, q);
    else hipLaunchKernelGGL((optim_step_kernel<2>), grid, dim3(256), 0, s, q);
    return launch_status("optim_step");
}
