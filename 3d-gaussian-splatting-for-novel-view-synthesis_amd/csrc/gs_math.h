// gs_math.h -- per-Gaussian math of the projection stage (forward and analytic backward), fp32.
//
// Product code.  The functions are `__host__ __device__` so the same source that the HIP kernels
// inline can be compiled by g++ into a host test library (csrc/host_math_check.cpp) and checked
// against the oracle on a machine without a GPU.  That host build is a TEST of this file; it is
// never used as a fallback for the product path.
//
// Stage ids (F*/B*) refer to SURVEY.md §8(a); reference lines are cited per function.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GS_HD __host__ __device__ __forceinline__
#else
#define GS_HD inline
#endif

// Everything in gs_math.h / gs_body.h is compiled with fp contraction "on": a * b + c written in ONE expression becomes an fma,
// decided by the front end per source expression.  (hipcc's default, "fast", lets the back end fuse across statements depending
// on the surrounding code, so the same function gave last-bit different results in different kernels -- a render under no_grad
// differed from the same render with gradients, the frame-by-frame path from render_frames.)  gs_body.h restores the default.
#if defined(__clang__)
#pragma clang fp contract(on)
#endif

namespace gsm {

// Camera block derived on the device from c2w (utils.py:25-29); read uniformly by every thread.
struct Camera {
    float w[9];     // Rwc = c2w[:3,:3]^T, row-major
    float t[3];     // -Rwc * c2w[:3,3]
    float eye[3];   // c2w[:3,3] (camera centre; spherical_harmonics.py:132)
    float pad;
};

// Scalars of one view, prepared on the host from gsplat_view.
struct ViewK {
    float fx, fy, cx, cy;
    float near_z, far_z;
    float gl, gr, gt, gb;       // guard-band bounds: -g-cx, W+g-cx, -g-cy, H+g-cy (utils.py:82-91)
    float opacity_min;          // alpha_cutoff * 0.5 (render.py:107)
    float min_conis, chi_clip, alpha_max, alpha_cutoff;
    float chi_pad;              // chi_clip * 1.001 + 1e-4, rounded ONCE on the host: every kernel that enumerates the row spans of a large
                                // Gaussian (projection, binning, deterministic backward) must see the same float
    int32_t H, W, tiles_x, tiles_y, tile;     // tile = the reference's T (render.py:62): only F10/F11's rectangle and pair count use it
    int32_t lists_x, lists_y;                 // grid of LIST_W x LIST_H-pixel lists: what is actually binned and rasterised
};

// A "list" is the depth-ordered set of Gaussians of one 16 x 8-pixel region: the unit one wave64 rasterises.  Fixed:
// the image does not depend on the binning granularity, so every T of the reference maps onto it.
constexpr int LIST_W = 16, LIST_H = 8;

enum : int { VIS_OK = 0, VIS_CULLED = 1, VIS_OFFSCREEN = 2 };

// 1 / x and sqrt(x) to 1 ulp in ONE instruction on the device (v_rcp_f32, v_sqrt_f32), used where nothing the image or the reference's
// integer outputs depend on is computed: padded extents and masks, gradient-only terms.  (An IEEE division is ~10 instructions, and
// the projection kernel had 35 of them.)
#if defined(__HIP_DEVICE_COMPILE__)
#define GS_RCP_FAST(x) __builtin_amdgcn_rcpf(x)
#define GS_SQRT_FAST(x) __builtin_amdgcn_sqrtf(x)
#else
#define GS_RCP_FAST(x) (1.0f / (x))
#define GS_SQRT_FAST(x) sqrtf(x)
#endif

GS_HD float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
GS_HD float clampf_(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

GS_HD void build_camera(const float* c2w, Camera& c) {
    // w2c = [R^T | -R^T t]  (utils.py:25-29)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c.w[i * 3 + j] = c2w[j * 4 + i];
    for (int i = 0; i < 3; ++i) {
        c.eye[i] = c2w[i * 4 + 3];
    }
    for (int i = 0; i < 3; ++i) c.t[i] = -(c.w[i * 3 + 0] * c.eye[0] + c.w[i * 3 + 1] * c.eye[1] + c.w[i * 3 + 2] * c.eye[2]);
    c.pad = 0.f;
}

// ---------------------------------------------------------------------------------------------
// F1 + F2: world covariance from (scale_raw, q_raw).  gaussian.py:24-68 (x,y,z,w quaternion),
// gaussian.py:115-127.  Output: symmetric S as (xx, xy, xz, yy, yz, zz).
// ---------------------------------------------------------------------------------------------
struct CovMid {
    float s[3];        // clamped scales
    float e[3];        // exp(scale_raw)
    float r[3];        // scale_raw
    float q[4];        // normalised quaternion
    float qn;          // |q_raw|
    float R[9];
};

GS_HD void quat_to_rot(const float q[4], float R[9]) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, xw = x * w, yw = y * w, zw = z * w;
    R[0] = 1.f - 2.f * (yy + zz); R[1] = 2.f * (xy - zw);       R[2] = 2.f * (xz + yw);
    R[3] = 2.f * (xy + zw);       R[4] = 1.f - 2.f * (xx + zz); R[5] = 2.f * (yz - xw);
    R[6] = 2.f * (xz - yw);       R[7] = 2.f * (yz + xw);       R[8] = 1.f - 2.f * (xx + yy);
}

GS_HD void cov_from_params(const float scale_raw[3], const float q_raw[4], float S[6], CovMid& m) {
    for (int k = 0; k < 3; ++k) { m.r[k] = scale_raw[k]; m.e[k] = expf(scale_raw[k]); m.s[k] = fmaxf(m.e[k], 1e-6f); }
    m.qn = sqrtf(q_raw[0] * q_raw[0] + q_raw[1] * q_raw[1] + q_raw[2] * q_raw[2] + q_raw[3] * q_raw[3]);
    const float inv = 1.0f / (m.qn + 1e-9f);
    for (int k = 0; k < 4; ++k) m.q[k] = q_raw[k] * inv;
    quat_to_rot(m.q, m.R);
    const float d0 = m.s[0] * m.s[0], d1 = m.s[1] * m.s[1], d2 = m.s[2] * m.s[2];
    const float* R = m.R;
    S[0] = R[0] * R[0] * d0 + R[1] * R[1] * d1 + R[2] * R[2] * d2;
    S[1] = R[0] * R[3] * d0 + R[1] * R[4] * d1 + R[2] * R[5] * d2;
    S[2] = R[0] * R[6] * d0 + R[1] * R[7] * d1 + R[2] * R[8] * d2;
    S[3] = R[3] * R[3] * d0 + R[4] * R[4] * d1 + R[5] * R[5] * d2;
    S[4] = R[3] * R[6] * d0 + R[4] * R[7] * d1 + R[5] * R[8] * d2;
    S[5] = R[6] * R[6] * d0 + R[7] * R[7] * d1 + R[8] * R[8] * d2;
}

// s_i^2 - s_j^2 of the clamped scales
GS_HD float cov_d_diff(const CovMid& m, int i, int j) {
    if (m.e[i] >= 1e-6f && m.e[j] >= 1e-6f) return m.s[j] * m.s[j] * expm1f(2.f * (m.r[i] - m.r[j]));
    return (m.s[i] - m.s[j]) * (m.s[i] + m.s[j]);
}

// B3 (covariance part).  G = dL/dS as a full symmetric 3x3 (row-major 9), i.e. dL = sum_ij G_ij dS_ij.
GS_HD void cov_from_params_backward(const float q_raw[4], const CovMid& m, const float G[9], float g_scale_raw[3],
                                    float g_q_raw[4]) {
    const float* R = m.R;
    const float d[3] = {m.s[0] * m.s[0], m.s[1] * m.s[1], m.s[2] * m.s[2]};
    // GR = G * R ; dL/dR = 2 * GR * D ; dL/dD_k = (R^T G R)_kk
    float GR[9];
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) GR[i * 3 + k] = G[i * 3 + 0] * R[0 + k] + G[i * 3 + 1] * R[3 + k] + G[i * 3 + 2] * R[6 + k];
    float dR[9];
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) dR[i * 3 + k] = 2.f * GR[i * 3 + k] * d[k];
    for (int k = 0; k < 3; ++k) {
        const float dD = R[0 + k] * GR[0 + k] + R[3 + k] * GR[3 + k] + R[6 + k] * GR[6 + k];
        const float ds = 2.f * m.s[k] * dD;
        g_scale_raw[k] = (m.e[k] >= 1e-6f) ? ds * m.e[k] : 0.f;
    }
    const float x = m.q[0], y = m.q[1], z = m.q[2], w = m.q[3];
    // dL/dq by the chain rule through quat_to_rot.  Its TANGENTIAL part (the one that survives the normalisation below) is a
    // difference of large terms -- sums like dR[7] - dR[5] extract the antisymmetric part of dR R^T = 2 G Sigma, which vanishes
    // as the scales approach each other: for a near-isotropic Gaussian the rounding of G R (1e-7 of |G| d) is then larger than
    // the result (~ |G| (d_i - d_j)), and the rotation gradient of that Gaussian came out 20x less accurate than autograd's.
    float dq[4];
    dq[0] = 2.f * (y * (dR[1] + dR[3]) + z * (dR[2] + dR[6]) - 2.f * x * (dR[4] + dR[8]) + w * (dR[7] - dR[5]));
    dq[1] = 2.f * (x * (dR[1] + dR[3]) + z * (dR[5] + dR[7]) - 2.f * y * (dR[0] + dR[8]) + w * (dR[2] - dR[6]));
    dq[2] = 2.f * (x * (dR[2] + dR[6]) + y * (dR[5] + dR[7]) - 2.f * z * (dR[0] + dR[4]) + w * (dR[3] - dR[1]));
    dq[3] = 2.f * (x * (dR[7] - dR[5]) + y * (dR[2] - dR[6]) + z * (dR[3] - dR[1]));
    // q = q_raw / (n + eps)
    const float ne = m.qn + 1e-9f;
    const float dot = dq[0] * q_raw[0] + dq[1] * q_raw[1] + dq[2] * q_raw[2] + dq[3] * q_raw[3];
    if (m.qn > 1e-4f) {
        // The same gradient without the cancellation.  A rotation of the Gaussian's own axes by d phi changes
        // Sigma = R D R^T by R [[d phi]x, D] R^T, so dL = tau . d phi with the torque (body frame, G' = R^T G R)
        //     tau_x = 2 G'_12 (d_1 - d_2),  tau_y = 2 G'_02 (d_2 - d_0),  tau_z = 2 G'_01 (d_0 - d_1):
        // proportional to the scale differences by construction (exactly zero for equal scales).  A unit quaternion moves by
        // dq = 1/2 q (x) (d phi, 0), whose 4 x 3 matrix M(q) has orthonormal columns: the tangential gradient is 2 M(q) tau.
        // q is a unit quaternion up to eps / |q_raw| = 1e-9 / n here (n > 1e-4: below fp32 resolution).  What the
        // normalisation does to the RADIAL part of dL/dq -- it survives with the factor eps / (n + eps)^2 -- is kept from the
        // chain-rule form, where it is a plain sum.
        const float gp01 = R[0] * GR[1] + R[3] * GR[4] + R[6] * GR[7];
        const float gp02 = R[0] * GR[2] + R[3] * GR[5] + R[6] * GR[8];
        const float gp12 = R[1] * GR[2] + R[4] * GR[5] + R[7] * GR[8];
        // d_i - d_j = d_j expm1(2 (r_i - r_j)) from the raw log-scales: the difference of two nearly equal r is exact, so the
        // result keeps its 1e-7 where s_i - s_j (two rounded exponentials) would be down to 1e-7 / |r_i - r_j|
        const float tx = 2.f * gp12 * cov_d_diff(m, 1, 2);
        const float ty = 2.f * gp02 * cov_d_diff(m, 2, 0);
        const float tz = 2.f * gp01 * cov_d_diff(m, 0, 1);
        const float gt[4] = {2.f * (w * tx - z * ty + y * tz), 2.f * (z * tx + w * ty - x * tz), 2.f * (-y * tx + x * ty + w * tz),
                             -2.f * (x * tx + y * ty + z * tz)};
        const float radial = dot * 1e-9f / (m.qn * m.qn * ne * ne);          // (g . q_hat) eps / ne^2, times q_hat = q_raw / n
        for (int k = 0; k < 4; ++k) g_q_raw[k] = gt[k] / ne + q_raw[k] * radial;
        return;
    }
    // tiny |q_raw| (comparable with the reference's eps): R(q) is not a rotation there; the plain chain rule
    const float c = (m.qn > 0.f) ? dot / (m.qn * ne * ne) : 0.f;
    for (int k = 0; k < 4; ++k) g_q_raw[k] = dq[k] / ne - q_raw[k] * c;
}

// ---------------------------------------------------------------------------------------------
// F3: view-dependent colour.  spherical_harmonics.py:118-166.  f_rest is channel-major.
// ---------------------------------------------------------------------------------------------
#define GS_K0 0.28209479177387814f
#define GS_K1 0.4886025119029199f
#define GS_K2A 1.0925484305920792f
#define GS_K2B 0.31539156525252005f
#define GS_K2C 0.5462742152960396f
#define GS_K3A 0.5900435899266435f
#define GS_K3B 2.890611442640554f
#define GS_K3C 0.4570457994644658f
#define GS_K3D 0.3731763325901154f
#define GS_K3E 1.445305721320277f

struct ShMid {
    float d[3];     // unit view direction
    float v[3];     // p - eye
    float n;        // |v|
    float Y[16];
};

// The colour of a Gaussian must not depend on WHICH kernel evaluates it (projection kernel, colour pass, with or without the
// saved Jacobian: the compiler fuses a * b + c into an fma or not depending on the surrounding code), or a render under
// no_grad would differ from the same render with gradients in the last bit.  So contraction is switched off in the functions
// the colour goes through, and the one fma that matters for speed is written out.
#if defined(__clang__)
#define GS_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define GS_NO_CONTRACT
#endif

GS_HD void sh_basis(const float p[3], const float eye[3], ShMid& m) {
    GS_NO_CONTRACT
    m.v[0] = p[0] - eye[0]; m.v[1] = p[1] - eye[1]; m.v[2] = p[2] - eye[2];
    m.n = sqrtf(m.v[0] * m.v[0] + m.v[1] * m.v[1] + m.v[2] * m.v[2]);
    const float inv = 1.0f / (m.n + 1e-8f);
    const float x = m.v[0] * inv, y = m.v[1] * inv, z = m.v[2] * inv;
    m.d[0] = x; m.d[1] = y; m.d[2] = z;
    const float xx = x * x, yy = y * y, zz = z * z;
    float* Y = m.Y;
    Y[0] = GS_K0;
    Y[1] = -GS_K1 * y; Y[2] = GS_K1 * z; Y[3] = -GS_K1 * x;
    Y[4] = GS_K2A * x * y; Y[5] = GS_K2A * y * z; Y[6] = GS_K2B * (3.f * zz - 1.f); Y[7] = GS_K2A * x * z;
    Y[8] = GS_K2C * (xx - yy);
    Y[9] = GS_K3A * y * (3.f * xx - yy); Y[10] = GS_K3B * x * y * z; Y[11] = GS_K3C * y * (4.f * zz - xx - yy);
    Y[12] = GS_K3D * z * (2.f * zz - 3.f * xx - 3.f * yy); Y[13] = GS_K3C * x * (4.f * zz - xx - yy);
    Y[14] = GS_K3E * z * (xx - yy); Y[15] = GS_K3A * x * (xx - 3.f * yy);
}

// coef(k, ch) returns the SH coefficient of basis k, channel ch.
template <class Coef>
GS_HD void sh_colour(const ShMid& m, Coef coef, float rgb[3]) {
    GS_NO_CONTRACT
    for (int ch = 0; ch < 3; ++ch) {
        float acc = 0.f;
        for (int k = 0; k < 16; ++k) acc = fmaf(coef(k, ch), m.Y[k], acc);
        rgb[ch] = sigmoidf_(acc);
    }
}

// dd[m] = sum_k dY[k] * d Y_k / d d_m: the basis gradient w.r.t. the unit direction, contracted with dY.
GS_HD void sh_basis_grad(const ShMid& m, const float dY[16], float dd[3]) {
    const float x = m.d[0], y = m.d[1], z = m.d[2];
    const float xx = x * x, yy = y * y, zz = z * z;
    dd[0] = -GS_K1 * dY[3] + GS_K2A * y * dY[4] + GS_K2A * z * dY[7] + 2.f * GS_K2C * x * dY[8] + 6.f * GS_K3A * x * y * dY[9] +
            GS_K3B * y * z * dY[10] - 2.f * GS_K3C * x * y * dY[11] - 6.f * GS_K3D * x * z * dY[12] +
            GS_K3C * (4.f * zz - 3.f * xx - yy) * dY[13] + 2.f * GS_K3E * x * z * dY[14] + GS_K3A * (3.f * xx - 3.f * yy) * dY[15];
    dd[1] = -GS_K1 * dY[1] + GS_K2A * x * dY[4] + GS_K2A * z * dY[5] - 2.f * GS_K2C * y * dY[8] +
            GS_K3A * (3.f * xx - 3.f * yy) * dY[9] + GS_K3B * x * z * dY[10] + GS_K3C * (4.f * zz - xx - 3.f * yy) * dY[11] -
            6.f * GS_K3D * y * z * dY[12] - 2.f * GS_K3C * x * y * dY[13] - 2.f * GS_K3E * y * z * dY[14] -
            6.f * GS_K3A * x * y * dY[15];
    dd[2] = GS_K1 * dY[2] + GS_K2A * y * dY[5] + 6.f * GS_K2B * z * dY[6] + GS_K2A * x * dY[7] + GS_K3B * x * y * dY[10] +
            8.f * GS_K3C * y * z * dY[11] + GS_K3D * (6.f * zz - 3.f * xx - 3.f * yy) * dY[12] + 8.f * GS_K3C * x * z * dY[13] +
            GS_K3E * (xx - yy) * dY[14];
}

// d = v / (n + eps), v = p - eye: gradient w.r.t. the direction -> gradient w.r.t. the point.
GS_HD void sh_dir_to_point(const ShMid& m, const float dd[3], float g_p[3]) {
    const float ne = m.n + 1e-8f, ine = GS_RCP_FAST(ne);
    const float dot = dd[0] * m.v[0] + dd[1] * m.v[1] + dd[2] * m.v[2];
    const float c = (m.n > 0.f) ? dot * GS_RCP_FAST(m.n * ne * ne) : 0.f;
    for (int k = 0; k < 3; ++k) g_p[k] = dd[k] * ine - m.v[k] * c;
}

// B3 (colour part).  g_rgb = dL/d colour.  Emits dL/dcoef through `emit(k, ch, value)`; returns dL/dp in g_p.
template <class Coef, class Emit>
GS_HD void sh_colour_backward(const ShMid& m, Coef coef, const float rgb[3], const float g_rgb[3], Emit emit, float g_p[3]) {
    float dY[16];
    for (int k = 0; k < 16; ++k) dY[k] = 0.f;
    for (int ch = 0; ch < 3; ++ch) {
        const float dpre = g_rgb[ch] * rgb[ch] * (1.f - rgb[ch]);
        for (int k = 0; k < 16; ++k) {
            const float cf = coef(k, ch);       // read before emit(): callers may alias the gradient onto the coefficient
            emit(k, ch, dpre * m.Y[k]);
            dY[k] += dpre * cf;
        }
    }
    float dd[3];
    sh_basis_grad(m, dY, dd);
    sh_dir_to_point(m, dd, g_p);
}

// F3 together with what its backward needs, so that the backward does not have to read the 48 coefficients again (192 of
// the 236 input bytes of a Gaussian):  KJ[ch] = d rgb_ch / d logit_ch,  KJ[3 + 3 ch + m] = d logit_ch / d p_m.
template <class Coef>
GS_HD void sh_colour_jac(const ShMid& m, Coef coef, float rgb[3], float KJ[12]) {
    sh_colour(m, coef, rgb);                                 // (the same instructions as without the Jacobian: see sh_basis)
    for (int ch = 0; ch < 3; ++ch) {
        float cf[16];
        for (int k = 0; k < 16; ++k) cf[k] = coef(k, ch);
        KJ[ch] = rgb[ch] * (1.f - rgb[ch]);
        float dd[3];
        sh_basis_grad(m, cf, dd);
        sh_dir_to_point(m, dd, KJ + 3 + 3 * ch);
    }
}

// B3 from the saved KJ: dL/dcoef(k, ch) = dpre_ch Y_k, dL/dp = sum_ch dpre_ch (d logit_ch / d p).
template <class Emit>
GS_HD void sh_colour_backward_jac(const ShMid& m, const float KJ[12], const float g_rgb[3], Emit emit, float g_p[3]) {
    g_p[0] = g_p[1] = g_p[2] = 0.f;
    for (int ch = 0; ch < 3; ++ch) {
        const float dpre = g_rgb[ch] * KJ[ch];
        for (int k = 0; k < 16; ++k) emit(k, ch, dpre * m.Y[k]);
        for (int mm = 0; mm < 3; ++mm) g_p[mm] += dpre * KJ[3 + 3 * ch + mm];
    }
}

// ---------------------------------------------------------------------------------------------
// F4-F8, F10, F13: projection of one Gaussian.  render.py:106-258, 307-315; utils.py:10-96,152-191.
// ---------------------------------------------------------------------------------------------
struct ProjMid {
    float xc, yc, zc;             // camera-space centre
    float sg;                     // sigmoid(opacity_raw)
    float iz;                     // 1 / max(zc, 1e-6)
    float j00, j11, j02, j12;     // Jacobian entries (render.py:165-171)
    float C[6];                   // camera-space covariance (xx,xy,xz,yy,yz,zz)
    float a, b, d;                // 2D covariance before the eigen clamp
    float l1, l2, f1, f2, rad, diff;
    bool clamped;
    float a2, b2, d2;             // after the eigen clamp (render.py:177-179)
    float det, sdet;              // utils.py:184-185
    float i00, i11;               // inverse diagonal before the min_conis clamp
};

struct Proj {
    float u, v, z;
    float A11, A12, A22;          // conic
    float ex, ey;                 // half-extents of {q <= chi_square_clip} along u and v (+inf if the conic is not PD)
    float opacity;
    int tx0, ty0, tx1, ty1;       // inclusive tile rectangle of the reference (F10/F11: square 2.5-sigma AABB, T x T tiles)
    int bx0, by0, bx1, by1;       // inclusive rectangle actually binned: tight box, in 16 x 8-pixel lists (empty: bx1 < bx0)
    uint32_t bmask;               // which lists of that rectangle the ellipse can touch (binned_mask)
    int btiles;                   // rectangles of more than 32 lists: lists the ellipse can touch, row by row (big_row_span)
    float bk4[4];                 // ... and the constants of those spans (big_span_constants), kept in the record
    int vis;                      // VIS_*
};

// Does the region {q <= chi} of a projected Gaussian touch the pixel rectangle [x0, x1] x [y0, y1] (pixel centres,
// inclusive)?  q is convex: over the rectangle its minimum is 0 if the centre is inside, else it lies on an edge that faces the
// centre.  With X = the centre's x clamped to the rectangle (in centre-relative coordinates: 0 if inside the x-range, else the
// nearer vertical edge) the line x = X is that edge -- or a line through the rectangle, whose points are harmless extra
// candidates -- and q along it is a clamped 1-D quadratic; the same with Y: min(qx, qy) is the exact minimum in every case.
// Conservative (a relative 1e-3 margin on chi dwarfs the fp32 differences with the rasterizer's own evaluation of q), so a
// "no" means alpha = 0 on every pixel of the rectangle.
GS_HD bool ellipse_touches_rect(const Proj& o, float chi_pad, float r12_22, float r12_11, float x0, float y0, float x1, float y1) {
    const float dx0 = x0 - o.u, dx1 = x1 - o.u, dy0 = y0 - o.v, dy1 = y1 - o.v;
    const float X = clampf_(0.f, dx0, dx1), t = clampf_(r12_22 * X, dy0, dy1);               // minimiser of q along the line x = X
    const float qx = o.A11 * X * X + (2.f * o.A12 * X + o.A22 * t) * t;
    const float Y = clampf_(0.f, dy0, dy1), s_ = clampf_(r12_11 * Y, dx0, dx1);              // ... along the line y = Y
    const float qy = o.A22 * Y * Y + (2.f * o.A12 * Y + o.A11 * s_) * s_;
    return !(fminf(qx, qy) > chi_pad);          // NaN -> true
}

// Rectangles of more than 32 lists (large Gaussians) have no mask; their lists are enumerated ROW BY ROW instead: which lists of
// row y (pixel rows y LIST_H .. y LIST_H + LIST_H - 1) does {q <= chi_pad} reach?  The region is convex, so inside a horizontal band
// it projects onto ONE x-interval [lo, hi]: with c = -A12 / A11 the curve's right side is x_hi(t) = c t + sqrt(chi / A11 - D t^2 / A11^2)
// (concave in t = dv, maximal = +ex at t = -(A12 / A22) ex), the left side x_lo(t) = c t - sqrt(..) (convex, minimal = -ex at the
// mirrored t): over the band's part [t0, t1] of the ellipse's dv-range the extreme is that stationary value if its t lies inside,
// else the larger (smaller) endpoint value.  Conservative like the list test (same padded chi, + 0.02 px); lists hold pixel CENTRES
// x LIST_W .. x LIST_W + LIST_W - 1.  Returns an empty span (xa > xb) for a row the ellipse misses.  Non-PD conic: the whole row.
struct RowSpan { int xa, xb; };
struct BigSpanK {               // per-Gaussian constants of big_row_span
    float u, v, ex, ey;         // centre; half-extents of {q <= chi} (the record's: Proj::ex, ey)
    float c, k0, k1, ts;        // c = -A12 / A11, k0 = chi_pad / A11, k1 = D / A11^2, ts = (A12 / A22) ex;  k0 < 0: the conic is not PD
    int bx0, bx1;
};
// (the projection kernel computes the four derived constants once and leaves them in the record's spare 16 bytes: the binning
//  kernels enumerate a large Gaussian's rows three times and would otherwise repeat four divisions each time)
GS_HD void big_span_constants(float A11, float A12, float A22, float ex, float chi_pad, float out[4]) {
    const float D = A11 * A22 - A12 * A12;
    const bool pd = (D > 0.f) && (A11 > 0.f) && (A22 > 0.f) && (ex < 1e30f);
    const float i11 = GS_RCP_FAST(A11), i22 = GS_RCP_FAST(A22);
    out[0] = -A12 * i11; out[1] = pd ? chi_pad * i11 : -1.f; out[2] = D * i11 * i11; out[3] = A12 * i22 * ex;
}
GS_HD BigSpanK big_span_setup(float u, float v, float ex, float ey, const float k4[4], int bx0, int bx1) {
    BigSpanK k;
    k.u = u; k.v = v; k.ex = ex; k.ey = ey; k.c = k4[0]; k.k0 = k4[1]; k.k1 = k4[2]; k.ts = k4[3]; k.bx0 = bx0; k.bx1 = bx1;
    return k;
}
GS_HD RowSpan big_row_span(const BigSpanK& k, int y) {
    if (!(k.k0 >= 0.f)) return RowSpan{k.bx0, k.bx1};
    const float d0 = (float)(y * LIST_H) - k.v - 0.02f, d1 = (float)(y * LIST_H + LIST_H - 1) - k.v + 0.02f;
    if (d0 > k.ey || d1 < -k.ey) return RowSpan{1, 0};
    const float t0 = fmaxf(d0, -k.ey), t1 = fminf(d1, k.ey);
    const float r0 = GS_SQRT_FAST(fmaxf(k.k0 - k.k1 * t0 * t0, 0.f)), r1 = GS_SQRT_FAST(fmaxf(k.k0 - k.k1 * t1 * t1, 0.f));
    float hi = fmaxf(k.c * t0 + r0, k.c * t1 + r1), lo = fminf(k.c * t0 - r0, k.c * t1 - r1);
    if (-k.ts >= t0 && -k.ts <= t1) hi = k.ex;
    if (k.ts >= t0 && k.ts <= t1) lo = -k.ex;
    hi = fminf(hi, k.ex); lo = fmaxf(lo, -k.ex);
    const float pad = 0.02f + 1e-4f * k.ex;
    // smallest list whose last pixel centre is >= u + lo - pad, largest whose first is <= u + hi + pad
    const float fa = ceilf((k.u + lo - pad - (float)(LIST_W - 1)) * (1.0f / LIST_W)), fb = floorf((k.u + hi + pad) * (1.0f / LIST_W));
    const int xa = (int)fmaxf(fa, (float)k.bx0), xb = (int)fminf(fb, (float)k.bx1);
    return RowSpan{xa, xb};
}

// Bit k (row-major inside the binned rectangle) = the Gaussian can touch list k.  Rectangles of more than 32 lists are
// not refined (all ones): their lists are counted and enumerated row by row (big_row_span).
GS_HD uint32_t binned_mask(const Proj& o, const ViewK& vk) {
    if (o.bx1 < o.bx0 || o.by1 < o.by0) return 0u;
    const int w = o.bx1 - o.bx0 + 1, h = o.by1 - o.by0 + 1;
    if (w * h > 32 || !(o.A11 > 0.f && o.A22 > 0.f)) return 0xFFFFFFFFu >> (w * h > 32 ? 0 : 32 - w * h);
    const float chi_pad = vk.chi_clip * 1.001f + 1e-4f, r12_22 = -o.A12 * GS_RCP_FAST(o.A22), r12_11 = -o.A12 * GS_RCP_FAST(o.A11);
    uint32_t m = 0u;
    int k = 0;
    for (int y = o.by0; y <= o.by1; ++y)
        for (int x = o.bx0; x <= o.bx1; ++x, ++k)
            if (ellipse_touches_rect(o, chi_pad, r12_22, r12_11, (float)(x * LIST_W), (float)(y * LIST_H), (float)(x * LIST_W + LIST_W - 1),
                                     (float)(y * LIST_H + LIST_H - 1)))
                m |= 1u << k;
    return m;
}

// cmid (nullable): the scales and the rotation S was built from (fused inputs).  With them the determinant of the 2-D covariance is
// formed WITHOUT the cancellation of a d - b^2 (a needle seen along its length: eigenvalues 0.17 and 1200 px^2 lose 4 digits there,
// and the conic -- its entries are entries of the covariance over the determinant -- with them):
//   det(J C J^T) = n^T adj(C) n,  n = j0 x j1,  C = Q diag(s^2) Q^T (Q = W R)  =>  det = sum_k (s_i s_j (Q^T n)_k)^2,  a sum of squares.
GS_HD void project_gaussian(const float p[3], const float S[6], float o_raw, const Camera& cam, const ViewK& vk, Proj& o,
                            ProjMid& m, const CovMid* cmid = nullptr) {
    o.vis = VIS_CULLED;
    o.tx0 = o.ty0 = 0; o.tx1 = o.ty1 = -1;
    o.bx0 = o.by0 = 0; o.bx1 = o.by1 = -1; o.bmask = 0u; o.btiles = 0;
    o.bk4[0] = o.bk4[1] = o.bk4[2] = o.bk4[3] = 0.f;
    // F4 opacity prefilter
    m.sg = sigmoidf_(o_raw);
    o.opacity = clampf_(m.sg, 0.f, 0.999f);
    if (!(o.opacity >= vk.opacity_min)) return;
    // F5 camera transform
    const float* w = cam.w;
    m.xc = w[0] * p[0] + w[1] * p[1] + w[2] * p[2] + cam.t[0];
    m.yc = w[3] * p[0] + w[4] * p[1] + w[5] * p[2] + cam.t[1];
    m.zc = w[6] * p[0] + w[7] * p[1] + w[8] * p[2] + cam.t[2];
    o.z = m.zc;
    // F6 frustum + guard band, strict inequalities
    const float fxx = vk.fx * m.xc, fyy = vk.fy * m.yc;
    const bool in = (m.zc > 0.f) && (m.zc > vk.near_z) && (m.zc < vk.far_z) && (fxx > m.zc * vk.gl) && (fxx < m.zc * vk.gr) &&
                    (fyy > m.zc * vk.gt) && (fyy < m.zc * vk.gb);
    if (!in) return;
    // F7 projection and EWA covariance
    o.u = fxx / m.zc + vk.cx;
    o.v = fyy / m.zc + vk.cy;
    // C = W S W^T
    float M[9];
    for (int i = 0; i < 3; ++i) {
        const float w0 = w[i * 3 + 0], w1 = w[i * 3 + 1], w2 = w[i * 3 + 2];
        M[i * 3 + 0] = w0 * S[0] + w1 * S[1] + w2 * S[2];
        M[i * 3 + 1] = w0 * S[1] + w1 * S[3] + w2 * S[4];
        M[i * 3 + 2] = w0 * S[2] + w1 * S[4] + w2 * S[5];
    }
    m.C[0] = M[0] * w[0] + M[1] * w[1] + M[2] * w[2];
    m.C[1] = M[0] * w[3] + M[1] * w[4] + M[2] * w[5];
    m.C[2] = M[0] * w[6] + M[1] * w[7] + M[2] * w[8];
    m.C[3] = M[3] * w[3] + M[4] * w[4] + M[5] * w[5];
    m.C[4] = M[3] * w[6] + M[4] * w[7] + M[5] * w[8];
    m.C[5] = M[6] * w[6] + M[7] * w[7] + M[8] * w[8];
    m.iz = 1.0f / fmaxf(m.zc, 1e-6f);
    const float iz2 = m.iz * m.iz;
    m.j00 = vk.fx * m.iz; m.j11 = vk.fy * m.iz;
    m.j02 = -vk.fx * m.xc * iz2; m.j12 = -vk.fy * m.yc * iz2;
    const float* C = m.C;
    // rows of J*C
    const float r0x = m.j00 * C[0] + m.j02 * C[2], r0y = m.j00 * C[1] + m.j02 * C[4], r0z = m.j00 * C[2] + m.j02 * C[5];
    const float r1y = m.j11 * C[3] + m.j12 * C[4], r1z = m.j11 * C[4] + m.j12 * C[5];
    m.a = r0x * m.j00 + r0z * m.j02;
    m.b = r0y * m.j11 + r0z * m.j12;
    m.d = r1y * m.j11 + r1z * m.j12;
    // F8 eigenvalues of the symmetric 2x2, clamp to [1e-6, 1e4], recomposition
    const float mid = 0.5f * (m.a + m.d);
    m.diff = 0.5f * (m.a - m.d);
    m.rad = sqrtf(m.diff * m.diff + m.b * m.b);
    m.l1 = mid + m.rad; m.l2 = mid - m.rad;
    float det_exact = -1.f;                                                          // (< 0: not available)
    if (cmid) {
        const float n0 = -m.j02 * m.j11, n1 = -m.j00 * m.j12, n2 = m.j00 * m.j11;    // j0 x j1
        const float t0 = w[0] * n0 + w[3] * n1 + w[6] * n2, t1 = w[1] * n0 + w[4] * n1 + w[7] * n2, t2 = w[2] * n0 + w[5] * n1 + w[8] * n2;
        const float* R = cmid->R;
        const float e0 = cmid->s[1] * cmid->s[2] * (R[0] * t0 + R[3] * t1 + R[6] * t2);
        const float e1 = cmid->s[0] * cmid->s[2] * (R[1] * t0 + R[4] * t1 + R[7] * t2);
        const float e2 = cmid->s[0] * cmid->s[1] * (R[2] * t0 + R[5] * t1 + R[8] * t2);
        det_exact = e0 * e0 + e1 * e1 + e2 * e2;
        if (m.l1 > 1e-30f && det_exact < 3e38f) m.l2 = det_exact / m.l1;            // the small eigenvalue without mid - rad
        else det_exact = -1.f;
    }
    m.f1 = clampf_(m.l1, 1e-6f, 1e4f); m.f2 = clampf_(m.l2, 1e-6f, 1e4f);
    m.clamped = (m.f1 != m.l1) || (m.f2 != m.l2);
    float a2, b2, d2;        // (locals, stored once: stores to m inside the branches were merged into a store through a pointer phi,
                             //  which kept a field of m in scratch memory)
    if (!m.clamped) {
        a2 = m.a; b2 = m.b; d2 = m.d;
    } else if (m.rad > 0.f) {
        // f(S) = f2 I + k (S - l2 I), k = (f1 - f2) / (l1 - l2); (S - l2 I) computed without cancellation
        const float k = (m.f1 - m.f2) / (2.f * m.rad);
        float am, dm;   // a - l2, d - l2
        if (m.diff > 0.f) { am = m.rad + m.diff; dm = (m.b * m.b) / am; }
        else { dm = m.rad - m.diff; am = (m.b * m.b) / dm; }
        a2 = m.f2 + k * am; d2 = m.f2 + k * dm; b2 = k * m.b;
    } else {
        a2 = m.f1; d2 = m.f1; b2 = 0.f;
    }
    m.a2 = a2; m.b2 = b2; m.d2 = d2;
    if (!(isfinite(m.a2) && isfinite(m.b2) && isfinite(m.d2))) return;       // render.py:187-201
    // F10 radius, AABB, on-screen test
    const float lam = fminf(fmaxf(m.f1, 1e-12f), 1e4f);
    const float r = ceilf(2.5f * sqrtf(lam));
    const float umin = floorf(o.u - r), umax = floorf(o.u + r), vmin = floorf(o.v - r), vmax = floorf(o.v + r);
    // F13 conic
    // (after the clamp the recomposed matrix has the eigenvalues f1, f2 by construction: their product, not a2 d2 - b2^2)
    m.det = m.clamped ? (m.rad > 0.f ? m.f1 * m.f2 : m.f1 * m.f1) : (det_exact >= 0.f ? det_exact : m.a2 * m.d2 - m.b2 * m.b2);
    m.sdet = fmaxf(m.det, 1e-12f);
    m.i00 = m.d2 / m.sdet; m.i11 = m.a2 / m.sdet;
    o.A11 = fmaxf(m.i00, vk.min_conis);
    o.A12 = -m.b2 / m.sdet;
    o.A22 = fmaxf(m.i11, vk.min_conis);
    // Tight bounding box of the region the rasterizer can touch: q(du,dv) <= chi  =>  |du| <= sqrt(chi A22 / D),
    // |dv| <= sqrt(chi A11 / D), D = A11 A22 - A12^2, derived from the conic actually used (after its clamps).
    // Padded (1e-4 relative + 1e-2 px) so that fp32 rounding of q can never put a covered pixel outside the box.
    {
        const float D = o.A11 * o.A22 - o.A12 * o.A12;
        if (D > 0.f && o.A11 > 0.f && o.A22 > 0.f) {
            const float cd = vk.chi_clip * GS_RCP_FAST(D);
            o.ex = GS_SQRT_FAST(cd * o.A22) * 1.0001f + 0.01f;
            o.ey = GS_SQRT_FAST(cd * o.A11) * 1.0001f + 0.01f;
            if (!(o.ex < 1e30f)) o.ex = 1e30f;
            if (!(o.ey < 1e30f)) o.ey = 1e30f;
        } else {
            o.ex = 1e30f; o.ey = 1e30f;
        }
    }
    if (!(umax >= 0.f && umin < (float)vk.W && vmax >= 0.f && vmin < (float)vk.H)) { o.vis = VIS_OFFSCREEN; return; }
    const float wm = (float)(vk.W - 1), hm = (float)(vk.H - 1);
    // (integer division by a run-time T costs ~35 instructions, four of them 10 % of this function: the default T = 16 is a shift)
    const int ux0 = (int)clampf_(umin, 0.f, wm), ux1 = (int)clampf_(umax, 0.f, wm);
    const int vy0 = (int)clampf_(vmin, 0.f, hm), vy1 = (int)clampf_(vmax, 0.f, hm);
    if (vk.tile == 16) { o.tx0 = ux0 >> 4; o.tx1 = ux1 >> 4; o.ty0 = vy0 >> 4; o.ty1 = vy1 >> 4; }
    else { o.tx0 = ux0 / vk.tile; o.tx1 = ux1 / vk.tile; o.ty0 = vy0 / vk.tile; o.ty1 = vy1 / vk.tile; }
    o.vis = VIS_OK;
    // What is binned: the tight box (the only pixels with q <= chi, i.e. alpha != 0) cut to the reference's AABB and
    // to the image, in lists of LIST_W x LIST_H pixels (one wave64, two pixels per lane).  The image does
    // not depend on the binning: a Gaussian missing from a list has alpha = 0 on all of its pixels.
    {
        const float lo_u = fmaxf(floorf(o.u - o.ex), umin), hi_u = fminf(floorf(o.u + o.ex), umax);
        const float lo_v = fmaxf(floorf(o.v - o.ey), vmin), hi_v = fminf(floorf(o.v + o.ey), vmax);
        if (hi_u >= 0.f && lo_u <= wm && hi_v >= 0.f && lo_v <= hm && lo_u <= hi_u && lo_v <= hi_v) {
            o.bx0 = (int)clampf_(lo_u, 0.f, wm) / LIST_W;
            o.bx1 = (int)clampf_(hi_u, 0.f, wm) / LIST_W;
            o.by0 = (int)clampf_(lo_v, 0.f, hm) / LIST_H;
            o.by1 = (int)clampf_(hi_v, 0.f, hm) / LIST_H;
            o.bmask = binned_mask(o, vk);
            const int w = o.bx1 - o.bx0 + 1, h = o.by1 - o.by0 + 1;
            o.btiles = 0;
            if (w * h > 32) {            // a large Gaussian: its lists row by row (what bin_count / bin_scatter enumerate)
                big_span_constants(o.A11, o.A12, o.A22, o.ex, vk.chi_pad, o.bk4);
                const BigSpanK bk = big_span_setup(o.u, o.v, o.ex, o.ey, o.bk4, o.bx0, o.bx1);
                for (int y = o.by0; y <= o.by1; ++y) {
                    const RowSpan sp = big_row_span(bk, y);
                    if (sp.xb >= sp.xa) o.btiles += sp.xb - sp.xa + 1;
                }
            }
        }
    }
}

// B2.  Inputs: gradients w.r.t. (u, v, A11, A12, A22, opacity) of a VISIBLE Gaussian (its ProjMid recomputed by
// project_gaussian).  Outputs: g_p[3] (position), G_S[9] (full symmetric world-covariance gradient), g_o_raw.
GS_HD void project_gaussian_backward(const ProjMid& m, const Proj& o, const Camera& cam, const ViewK& vk, float g_u, float g_v,
                                     float g_A11, float g_A12, float g_A22, float g_opacity, float g_p[3], float G_S[9],
                                     float& g_o_raw) {
    // opacity = clamp(sigmoid, 0, 0.999)
    g_o_raw = (m.sg <= 0.999f) ? g_opacity * m.sg * (1.f - m.sg) : 0.f;
    // min_conis clamp (render.py:310-311): gradient passes where the value is >= the bound
    const float g00 = (m.i00 >= vk.min_conis) ? g_A11 : 0.f;
    const float g11 = (m.i11 >= vk.min_conis) ? g_A22 : 0.f;
    const float g01 = g_A12;
    // inv2x2 with clamped determinant (utils.py:180-191); b and c of the reference are both b2 here
    const float is = 1.0f / m.sdet;
    float ga = g11 * is, gd = g00 * is, gb = -g01 * is, gc = 0.f;
    const float gsdet = -(g00 * m.d2 - g01 * m.b2 + g11 * m.a2) * is * is;
    const float gdet = (m.det >= 1e-12f) ? gsdet : 0.f;
    ga += m.d2 * gdet; gd += m.a2 * gdet; gb += -m.b2 * gdet; gc += -m.b2 * gdet;
    float Ga = ga, Gb = 0.5f * (gb + gc), Gd = gd;            // symmetrised 2x2 gradient w.r.t. the clamped covariance
    if (m.clamped) {
        // Daleckii-Krein: G = V [(V^T Gs V) o K] V^T, K_ii = f'(l_i), K_12 = (f1-f2)/(l1-l2)
        float cx_, sx_;
        if (m.rad > 0.f) {
            float ex, ey;
            if (m.diff > 0.f) { ex = m.rad + m.diff; ey = m.b; } else { ex = m.b; ey = m.rad - m.diff; }
            const float inv = 1.0f / sqrtf(ex * ex + ey * ey);
            cx_ = ex * inv; sx_ = ey * inv;
        } else { cx_ = 1.f; sx_ = 0.f; }
        const float k11 = (m.l1 >= 1e-6f && m.l1 <= 1e4f) ? 1.f : 0.f;
        const float k22 = (m.l2 >= 1e-6f && m.l2 <= 1e4f) ? 1.f : 0.f;
        const float k12 = (m.rad > 0.f) ? (m.f1 - m.f2) / (2.f * m.rad) : k11;
        // v1 = (c, s), v2 = (-s, c)
        const float t11 = cx_ * (Ga * cx_ + Gb * sx_) + sx_ * (Gb * cx_ + Gd * sx_);
        const float t12 = cx_ * (-Ga * sx_ + Gb * cx_) + sx_ * (-Gb * sx_ + Gd * cx_);
        const float t22 = -sx_ * (-Ga * sx_ + Gb * cx_) + cx_ * (-Gb * sx_ + Gd * cx_);
        const float h11 = k11 * t11, h12 = k12 * t12, h22 = k22 * t22;
        Ga = h11 * cx_ * cx_ - 2.f * h12 * cx_ * sx_ + h22 * sx_ * sx_;
        Gb = h11 * cx_ * sx_ + h12 * (cx_ * cx_ - sx_ * sx_) - h22 * cx_ * sx_;
        Gd = h11 * sx_ * sx_ + 2.f * h12 * cx_ * sx_ + h22 * cx_ * cx_;
    }
    // S2 = J C J^T :  dC = J^T G J,  dJ = 2 G J C
    const float* C = m.C;
    const float jc0x = m.j00 * C[0] + m.j02 * C[2], jc0y = m.j00 * C[1] + m.j02 * C[4], jc0z = m.j00 * C[2] + m.j02 * C[5];
    const float jc1x = m.j11 * C[1] + m.j12 * C[2], jc1y = m.j11 * C[3] + m.j12 * C[4], jc1z = m.j11 * C[4] + m.j12 * C[5];
    const float dJ00 = 2.f * (Ga * jc0x + Gb * jc1x);
    const float dJ02 = 2.f * (Ga * jc0z + Gb * jc1z);
    const float dJ11 = 2.f * (Gb * jc0y + Gd * jc1y);
    const float dJ12 = 2.f * (Gb * jc0z + Gd * jc1z);
    // J^T G J with J = [[j00,0,j02],[0,j11,j12]]
    float dC[9];
    const float gj0x = Ga * m.j00, gj0y = Gb * m.j11, gj0z = Ga * m.j02 + Gb * m.j12;   // row 0 of G J
    const float gj1x = Gb * m.j00, gj1y = Gd * m.j11, gj1z = Gb * m.j02 + Gd * m.j12;   // row 1 of G J
    dC[0] = m.j00 * gj0x; dC[1] = m.j00 * gj0y; dC[2] = m.j00 * gj0z;
    dC[3] = m.j11 * gj1x; dC[4] = m.j11 * gj1y; dC[5] = m.j11 * gj1z;
    dC[6] = m.j02 * gj0x + m.j12 * gj1x; dC[7] = m.j02 * gj0y + m.j12 * gj1y; dC[8] = m.j02 * gj0z + m.j12 * gj1z;
    // dS = W^T dC W
    const float* w = cam.w;
    float T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) T[i * 3 + j] = dC[i * 3 + 0] * w[0 + j] + dC[i * 3 + 1] * w[3 + j] + dC[i * 3 + 2] * w[6 + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) G_S[i * 3 + j] = w[0 + i] * T[0 + j] + w[3 + i] * T[3 + j] + w[6 + i] * T[6 + j];
    // camera-space point
    const float iz2 = m.iz * m.iz;
    float dxc = -vk.fx * iz2 * dJ02;
    float dyc = -vk.fy * iz2 * dJ12;
    const float diz = vk.fx * dJ00 + vk.fy * dJ11 - 2.f * vk.fx * m.xc * m.iz * dJ02 - 2.f * vk.fy * m.yc * m.iz * dJ12;
    float dzc = (m.zc >= 1e-6f) ? -iz2 * diz : 0.f;
    const float rz = 1.0f / m.zc;
    dxc += vk.fx * rz * g_u; dzc += -vk.fx * m.xc * rz * rz * g_u;
    dyc += vk.fy * rz * g_v; dzc += -vk.fy * m.yc * rz * rz * g_v;
    g_p[0] = w[0] * dxc + w[3] * dyc + w[6] * dzc;
    g_p[1] = w[1] * dxc + w[4] * dyc + w[7] * dzc;
    g_p[2] = w[2] * dxc + w[5] * dyc + w[8] * dzc;
    (void)o;
}

}  // namespace gsm
