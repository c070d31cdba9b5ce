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
#include <cmath>
#include <cstdint>
#include <cstdio>

#include <hip/hip_runtime.h>

#include "../../include/gsplat_mi355x.h"
#include "gs_adam.h"

extern thread_local char gsplat_err_buf[512];

namespace {

constexpr int NPART = 1024;         // workgroups of the norm kernel = partial sums (no atomics, no kernel to zero them)

__global__ __launch_bounds__(256) void sqnorm_kernel(int64_t n, const float* __restrict__ g, float* __restrict__ part) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const float v = g[i]; acc += v * v; }
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// one wave: sum of the partial sums in a fixed order (double) -> (coef, norm)
__global__ __launch_bounds__(64) void clip_coef_kernel(const float* __restrict__ part, int n_part, float max_norm, float* __restrict__ out /* coef, norm */) {
    double s = 0.0;
    for (int k = threadIdx.x; k < n_part; k += 64) s += (double)part[k];
    for (int sft = 32; sft > 0; sft >>= 1) s += __shfl_xor(s, sft);
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(s);
        const float coef = max_norm / (norm + 1e-6f);
        out[0] = coef < 1.0f ? coef : 1.0f;
        out[1] = norm;
    }
}

// One launch updates up to MAX_GROUPS tensors (the reference has six parameter groups: six launches of a streaming kernel left
// the chip draining and refilling five times).  A block belongs to the group whose block range holds it; inside a group the blocks
// stride over 16-byte pieces when the four arrays are 16-byte aligned (vec), else over single values.
constexpr int MAX_GROUPS = 8;
struct AdamGroup {
    float* p; float* g; float* m; float* v;
    const float* grad_scale;
    int64_t n;
    float step_size, inv_sqrt_bc2;
    uint32_t first_block, blocks;
    int32_t vec;
};
struct AdamGroups { AdamGroup g[MAX_GROUPS]; int32_t count; float b1, b2, eps; };

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void adam_kernel(AdamGroups a) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < MAX_GROUPS; ++j)
        if (j < a.count && blockIdx.x >= a.g[j].first_block) k = j;
    // (uniform selects instead of a dynamic index into the kernel-argument struct)
    AdamGroup G = a.g[0];
#pragma unroll
    for (int j = 1; j < MAX_GROUPS; ++j)
        if (k == j) G = a.g[j];
    const float b1 = a.b1, b2 = a.b2, eps = a.eps;
    const bool scaled = G.grad_scale != nullptr;
    const float gs = scaled ? *G.grad_scale : 1.0f;
    const int64_t b = blockIdx.x - G.first_block, stride = (int64_t)G.blocks * 256;
    if (G.vec) {
        const int64_t n4 = G.n >> 2;
        f4* __restrict__ P = reinterpret_cast<f4*>(G.p); f4* __restrict__ Gr = reinterpret_cast<f4*>(G.g);
        f4* __restrict__ M = reinterpret_cast<f4*>(G.m); f4* __restrict__ V = reinterpret_cast<f4*>(G.v);
        for (int64_t i = b * 256 + threadIdx.x; i < n4; i += stride) {
            f4 p = P[i], g = Gr[i], m = __builtin_nontemporal_load(&M[i]), v = __builtin_nontemporal_load(&V[i]);   // (the moments are touched once per step: streaming)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float pc = p[c], gc = g[c], mc = m[c], vc = v[c];
                adam_one(pc, gc, mc, vc, gs, scaled, G.step_size, b1, b2, G.inv_sqrt_bc2, eps);
                p[c] = pc; g[c] = gc; m[c] = mc; v[c] = vc;
            }
            if (scaled) Gr[i] = g;
            __builtin_nontemporal_store(m, &M[i]); __builtin_nontemporal_store(v, &V[i]); P[i] = p;
        }
        for (int64_t i = (n4 << 2) + b * 256 + threadIdx.x; i < G.n; i += stride)       // up to three values
            adam_one(G.p[i], G.g[i], G.m[i], G.v[i], gs, scaled, G.step_size, b1, b2, G.inv_sqrt_bc2, eps);
    } else {
        for (int64_t i = b * 256 + threadIdx.x; i < G.n; i += stride)
            adam_one(G.p[i], G.g[i], G.m[i], G.v[i], gs, scaled, G.step_size, b1, b2, G.inv_sqrt_bc2, eps);
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

int64_t gsplat_clip_scratch_bytes(void) { return NPART * sizeof(float); }

int gsplat_clip_grad_norm(int64_t n, const float* grad, float max_norm, float* coef_and_norm, void* scratch, void* stream_) {
    if (n < 0 || (n > 0 && !grad) || !coef_and_norm || !scratch) {
        snprintf(gsplat_err_buf, sizeof(gsplat_err_buf), "gsplat_clip_grad_norm: bad argument");
        return GSPLAT_ERR_BAD_ARG;
    }
    hipStream_t st = (hipStream_t)stream_;
    float* part = (float*)scratch;
    const int n_part = n > 0 ? (int)(grid_for(n) > (unsigned)NPART ? (unsigned)NPART : grid_for(n)) : 0;
    if (n > 0) hipLaunchKernelGGL(sqnorm_kernel, dim3(n_part), dim3(256), 0, st, n, grad, part);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, st, (const float*)part, n_part, max_norm, coef_and_norm);
    return launch_err("gsplat_clip_grad_norm");
}

int gsplat_adam_step_multi(int32_t n_groups, const gsplat_adam_group* groups, float beta1, float beta2, float eps, void* stream_) {
    if (n_groups < 0 || n_groups > MAX_GROUPS || (n_groups > 0 && !groups)) {
        snprintf(gsplat_err_buf, sizeof(gsplat_err_buf), "gsplat_adam_step_multi: 0..%d groups", MAX_GROUPS);
        return GSPLAT_ERR_BAD_ARG;
    }
    AdamGroups a;
    a.count = 0; a.b1 = beta1; a.b2 = beta2; a.eps = eps;
    uint32_t blocks = 0;
    for (int k = 0; k < n_groups; ++k) {
        const gsplat_adam_group& q = groups[k];
        if (q.n < 0 || q.step < 1 || (q.n > 0 && (!q.param || !q.grad || !q.exp_avg || !q.exp_avg_sq))) {
            snprintf(gsplat_err_buf, sizeof(gsplat_err_buf), "gsplat_adam_step_multi: bad group %d", k);
            return GSPLAT_ERR_BAD_ARG;
        }
        if (q.n == 0) continue;
        const double bc1 = 1.0 - pow((double)beta1, (double)q.step), bc2 = 1.0 - pow((double)beta2, (double)q.step);
        AdamGroup& G = a.g[a.count++];
        G.p = q.param; G.g = q.grad; G.m = q.exp_avg; G.v = q.exp_avg_sq; G.grad_scale = q.grad_scale; G.n = q.n;
        G.step_size = (float)((double)q.lr / bc1);
        G.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
        G.vec = ((((uintptr_t)q.param | (uintptr_t)q.grad | (uintptr_t)q.exp_avg | (uintptr_t)q.exp_avg_sq) & 15u) == 0) ? 1 : 0;
        // blocks in proportion to the bytes: 16 values per thread and pass at least, 4096 blocks for the largest tensors
        const int64_t per_block = 256 * 16;
        int64_t nb = (q.n + per_block - 1) / per_block;
        if (nb > 4096) nb = 4096;
        G.first_block = blocks; G.blocks = (uint32_t)nb;
        blocks += (uint32_t)nb;
    }
    for (int k = a.count; k < MAX_GROUPS; ++k) a.g[k] = AdamGroup{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0.f, 0xFFFFFFFFu, 1u, 0};
    if (a.count == 0) return GSPLAT_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, a);
    return launch_err("gsplat_adam_step_multi");
}

int gsplat_adam_step(int64_t n, float* param, float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                     float eps, int32_t step, const float* grad_scale, void* stream_) {
    if (n < 0 || step < 1 || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) {
        snprintf(gsplat_err_buf, sizeof(gsplat_err_buf), "gsplat_adam_step: bad argument");
        return GSPLAT_ERR_BAD_ARG;
    }
    const gsplat_adam_group one = {n, param, grad, exp_avg, exp_avg_sq, lr, step, grad_scale};
    return gsplat_adam_step_multi(1, &one, beta1, beta2, eps, stream_);
}

}  // extern "C"
