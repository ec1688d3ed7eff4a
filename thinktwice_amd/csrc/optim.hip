// Optimizer step of the training loop over the FLAT parameter / gradient buffers
// (thinktwice_amd/grad_sync.py): global-norm gradient clip and AdamW, the reference's
// `optimizer = AdamW(lr=1e-4, weight_decay=1e-7)` + `grad_clip(max_norm=100)` (configs/thinktwice.py:282-287,
// applied by mmcv's OptimizerHook through torch.nn.utils.clip_grad_norm_ and torch.optim.AdamW).
// Both are pure HBM streams: the norm reads 4 B per parameter, the fused update reads 16 B and writes 12 B per
// parameter (p, g, m, v -> p, m, v), one launch each, no host synchronisation (the clip factor stays on the device).
#include "tt_common.h"

namespace tt {

constexpr int kNormBlocks = 1024;

// phase 1: per-block partial sums of squares (fixed summation order => deterministic)
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long long n,
                                                            float* __restrict__ partial) {
    __shared__ float red[256];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = g[i];
        s += v * v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// phase 2: total norm and clip coefficient min(1, max_norm / (norm + 1e-6))  (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(256) void norm_finalize_kernel(const float* __restrict__ partial, int nblocks,
                                                            float max_norm, float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        const float coef = max_norm / (norm + 1e-6f);
        out[0] = norm;
        // a non-finite gradient norm (overflow, NaN anywhere in the buffer): the factor is NaN = "skip this update"; adamw_kernel
        // then leaves p / m / v untouched, so a bad iteration cannot poison the weights before the host has looked at the norm
        out[1] = (norm == norm && norm <= 3.0e38f) ? (coef < 1.f ? coef : 1.f) : __builtin_nanf("");
    }
}

// torch.optim.AdamW (decoupled weight decay, bias-corrected), gradient pre-multiplied by *grad_scale (clip factor)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, long long n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, float bias_c1, float bias_c2_sqrt,
                             const float* __restrict__ grad_scale) {
    const float gs = grad_scale ? *grad_scale : 1.f;
    if (gs != gs) return;          // NaN clip factor: non-finite gradient norm, the update is skipped (block-uniform)
    const float step_size = lr / bias_c1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gs;
        float pi = p[i] * (1.f - lr * weight_decay);
        const float mi = m[i] + (gi - m[i]) * (1.f - beta1);              // lerp_(grad, 1 - beta1)
        const float vi = v[i] * beta2 + gi * gi * (1.f - beta2);          // mul_(beta2).addcmul_(g, g, 1 - beta2)
        const float denom = sqrtf(vi) / bias_c2_sqrt + eps;
        pi = pi - step_size * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

// The same update with the STEP COUNT on the device: `*good_steps` counts the updates that were actually applied, so a skipped
// iteration (NaN clip factor) leaves the bias corrections, like p / m / v, at the last good step -- whatever the host does
// meanwhile.  Every launch of one optimizer step (one per live parameter range) reads the same count; adamw_advance_kernel,
// issued once after them, increments it when the factor was finite.
__global__ void adamw_dev_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                      float* __restrict__ v, long long n, float lr, float beta1, float beta2, float eps,
                                      float weight_decay, const int* __restrict__ good_steps,
                                      const float* __restrict__ grad_scale) {
    const float gs = grad_scale ? *grad_scale : 1.f;
    if (gs != gs) return;
    const double step = (double)(*good_steps + 1);
    const float bias_c1 = (float)(1.0 - pow((double)beta1, step));
    const float bias_c2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
    const float step_size = lr / bias_c1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gs;
        float pi = p[i] * (1.f - lr * weight_decay);
        const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
        const float vi = v[i] * beta2 + gi * gi * (1.f - beta2);
        const float denom = sqrtf(vi) / bias_c2_sqrt + eps;
        pi = pi - step_size * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

__global__ void adamw_advance_kernel(int* good_steps, const float* __restrict__ grad_scale) {
    const float gs = grad_scale ? *grad_scale : 1.f;
    if (gs == gs) *good_steps += 1;
}

}  // namespace tt

using namespace tt;

extern "C" int tt_grad_norm_clip(const float* grad, long long n, float max_norm, float* workspace_1024,
                                 float* out_norm_scale, void* stream) {
    TT_REQUIRE(grad && workspace_1024 && out_norm_scale && n > 0 && max_norm > 0.f, "tt_grad_norm_clip: bad args");
    hipStream_t st = (hipStream_t)stream;
    long long want = (n + 255) / 256;
    const int blocks = (int)(want < kNormBlocks ? want : kNormBlocks);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, st, grad, n, workspace_1024);
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(256), 0, st, workspace_1024, blocks, max_norm,
                       out_norm_scale);
    return check_launch("tt_grad_norm_clip");
}

extern "C" int tt_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                             float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                             const float* grad_scale_or_null, void* stream) {
    TT_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "tt_adamw_step: bad args");
    TT_REQUIRE(lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f,
               "tt_adamw_step: bad hyper-parameters");
    const double b1 = 1.0 - pow((double)beta1, (double)step);
    const double b2 = 1.0 - pow((double)beta2, (double)step);
    long long blocks = (n + 255) / 256;
    if (blocks > 256LL * 32) blocks = 256LL * 32;        // grid-stride beyond 8,192 workgroups (32 per CU)
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, (float)b1, (float)sqrt(b2),
                       grad_scale_or_null);
    return check_launch("tt_adamw_step");
}

extern "C" int tt_adamw_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, const int* good_steps_dev,
                                 const float* grad_scale_or_null, void* stream) {
    TT_REQUIRE(param && grad && exp_avg && exp_avg_sq && good_steps_dev && n > 0, "tt_adamw_step_dev: bad args");
    TT_REQUIRE(lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f,
               "tt_adamw_step_dev: bad hyper-parameters");
    long long blocks = (n + 255) / 256;
    if (blocks > 256LL * 32) blocks = 256LL * 32;
    hipLaunchKernelGGL(adamw_dev_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, good_steps_dev, grad_scale_or_null);
    return check_launch("tt_adamw_step_dev");
}

extern "C" int tt_adamw_advance(int* good_steps_dev, const float* grad_scale_or_null, void* stream) {
    TT_REQUIRE(good_steps_dev, "tt_adamw_advance: null");
    hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, good_steps_dev, grad_scale_or_null);
    return check_launch("tt_adamw_advance");
}
