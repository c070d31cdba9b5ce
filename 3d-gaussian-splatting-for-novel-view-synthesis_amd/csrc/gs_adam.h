// gs_adam.h -- one Adam update of one value, shared by the optimiser kernel (gsplat_optim.hip) and the projection backward that
// applies it to the SH coefficients as their gradient is formed (gsplat_kernels.hip, gsplat_backward_adam_rest): the same
// instructions in both places, so the two ways to step a parameter give the same bits.
// torch.optim.Adam (amsgrad = False, weight_decay = 0, maximize = False):
//     m = b1 m + (1 - b1) g;   v = b2 v + (1 - b2) g^2;   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#pragma once
#include <hip/hip_runtime.h>

// (contraction off inside: the compiler fuses a * b + c into an fma where it sees fit, and it sees fit differently in the two kernels
//  this is inlined into -- HIP's __fmul_rn and friends are plain operators, no barrier; the fused operations wanted are written out)
__device__ __forceinline__ float adam_one(float& p, float& g, float& m, float& v, float gs, bool scaled, float step_size, float b1, float b2,
                                          float inv_sqrt_bc2, float eps) {
#pragma clang fp contract(off)
    float gi = g;
    if (scaled) { gi = gi * gs; g = gi; }                      // clip_grad_norm_ scales the gradient in place
    const float mi = __builtin_fmaf(1.0f - b1, gi - m, m);     // lerp, as torch does
    const float g2 = gi * gi, bv = b2 * v;
    const float vi = __builtin_fmaf(1.0f - b2, g2, bv);
    m = mi; v = vi;
    const float den = __builtin_fmaf(sqrtf(vi), inv_sqrt_bc2, eps);
    const float q = mi / den;
    const float dp = step_size * q;
    p = p - dp;
    return gi;
}

// the per-tensor constants of one step (host side: bias corrections in double)
struct AdamStep { float step_size, inv_sqrt_bc2, b1, b2, eps; };
