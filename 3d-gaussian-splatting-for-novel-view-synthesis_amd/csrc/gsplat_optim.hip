// gsplat_optim.hip -- fused optimiser step of the reference's training loop (SURVEY.md §8f "next" row 2).
//
// Reference call sites (scripts/train.py): Adam with six parameter groups and eps = 1e-15 (:394-401), the position
// learning-rate schedule (:446-457, host-side arithmetic, see optim.py), clip_grad_norm_(model.pos, 1.0) (:536) and
// optimizer.step() (:538).  The arithmetic restated here is torch.optim.Adam's (torch 2.x, amsgrad = False,
// weight_decay = 0, maximize = False):
//     m = b1 m + (1 - b1) g;   v = b2 v + (1 - b2) g^2;   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// and torch.nn.utils.clip_grad_norm_'s: coef = min(1, max_norm / (||g||_2 + 1e-6)), g *= coef.
// The clip coefficient stays on the device (no host synchronisation); the Adam kernel multiplies it into the gradient
// it reads and also writes the clipped gradient back, as clip_grad_norm_ does in place.
#include <cstdio>

#include <hip/hip_runtime.h>

#include "../../include/gsplat_mi355x.h"

extern thread_local char gsplat_err_buf[512];

namespace {

constexpr int NSHARD = 64;

__global__ void sqnorm_zero_kernel(float* shards) {
    if (threadIdx.x < NSHARD) shards[threadIdx.x] = 0.f;
}

__global__ __launch_bounds__(256) void sqnorm_kernel(int64_t n, const float* __restrict__ g, float* __restrict__ shards) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const float v = g[i]; acc += v * v; }
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&shards[blockIdx.x % NSHARD], red[0] + red[1] + red[2] + red[3]);
}

__global__ void clip_coef_kernel(const float* __restrict__ shards, float max_norm, float* __restrict__ out /* coef, norm */) {
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < NSHARD; ++k) s += shards[k];
        const float norm = (float)sqrt(s);
        const float coef = max_norm / (norm + 1e-6f);
        out[0] = coef < 1.0f ? coef : 1.0f;
        out[1] = norm;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(int64_t n, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, float step_size, float b1, float b2, float inv_sqrt_bc2,
                                                   float eps, const float* __restrict__ grad_scale) {
    const float gs = grad_scale ? *grad_scale : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float gi = g[i];
        if (grad_scale) { gi *= gs; g[i] = gi; }                 // clip_grad_norm_ scales the gradient in place
        const float mi = m[i] + (1.0f - b1) * (gi - m[i]);       // lerp, as torch does
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    }
}

inline unsigned grid_for(int64_t n) {
    const int64_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

int launch_err(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return GSPLAT_OK;
    snprintf(gsplat_err_buf, sizeof(gsplat_err_buf), "%s: %s", what, hipGetErrorString(e));
    return GSPLAT_ERR_HIP;
}

}  // namespace

extern "C" {

int64_t gsplat_clip_scratch_bytes(void) { return NSHARD * sizeof(float); }

int gsplat_clip_grad_norm(int64_t n, const float* grad, float max_norm, float* coef_and_norm, void* scratch, void* stream_) {
    if (n < 0 || (n > 0 && !grad) || !coef_and_norm || !scratch) {
        snprintf(gsplat_err_buf, sizeof(gsplat_err_buf), "gsplat_clip_grad_norm: bad argument");
        return GSPLAT_ERR_BAD_ARG;
    }
    hipStream_t st = (hipStream_t)stream_;
    float* shards = (float*)scratch;
    hipLaunchKernelGGL(sqnorm_zero_kernel, dim3(1), dim3(64), 0, st, shards);
    if (n > 0) hipLaunchKernelGGL(sqnorm_kernel, dim3(grid_for(n) > 1024 ? 1024 : grid_for(n)), dim3(256), 0, st, n, grad, shards);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, st, shards, max_norm, coef_and_norm);
    return launch_err("gsplat_clip_grad_norm");
}

int gsplat_adam_step(int64_t n, float* param, float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                     float eps, int32_t step, const float* grad_scale, void* stream_) {
    if (n < 0 || step < 1 || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) {
        snprintf(gsplat_err_buf, sizeof(gsplat_err_buf), "gsplat_adam_step: bad argument");
        return GSPLAT_ERR_BAD_ARG;
    }
    if (n == 0) return GSPLAT_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream_, n, param, grad, exp_avg, exp_avg_sq,
                       (float)((double)lr / bc1), beta1, beta2, (float)(1.0 / sqrt(bc2)), eps, grad_scale);
    return launch_err("gsplat_adam_step");
}

}  // extern "C"
