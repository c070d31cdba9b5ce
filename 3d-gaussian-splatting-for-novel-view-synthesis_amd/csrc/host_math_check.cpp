// host_math_check.cpp -- host build of the per-Gaussian bodies in gs_body.h, for CPU unit tests only.
//
// NOT part of the product path: the product library (libgsplat_mi355x.so) contains only HIP kernels and fails
// loudly without a GPU.  This file lets `pytest -m "not gpu"` check the projection math (forward and analytic
// backward) against the oracle on a machine with no GPU, before any GPU time is spent.
#include "gs_body.h"

using namespace gsm;

extern "C" {

// rect / tiles: the reference's own tile rectangle and pair count (F10/F11); brect / btiles: what the kernels bin
// (tight box, 16 x 8 half-tile lists)
void hm_project(const gsplat_gaussians* g, const float* c2w, const gsplat_view* v, float* rec64, uint32_t* rect, float* depth,
                uint32_t* tiles, int32_t* vis, uint32_t* brect, uint32_t* btiles, uint32_t* bmask) {
    Camera cam; build_camera(c2w, cam);
    const ViewK vk = make_viewk(*v);
    const bool fused = g->scale_raw != nullptr;
    Records out{(Rec64*)rec64, (u2*)brect, depth, btiles, bmask, (u2*)rect, tiles};
    for (int64_t i = 0; i < g->n; ++i) {
        ShCoefGlobal coef{fused ? g->f_dc + i * 3 : nullptr, fused ? g->f_rest + i * 45 : nullptr};
        vis[i] = project_one(i, *g, fused, coef, cam, vk, out);
    }
}

void hm_project_backward(const gsplat_gaussians* g, const float* c2w, const gsplat_view* v, const uint32_t* tiles,
                         const float* grad2d, const gsplat_gaussian_grads* out) {
    Camera cam; build_camera(c2w, cam);
    const ViewK vk = make_viewk(*v);
    const bool fused = g->scale_raw != nullptr;
    for (int64_t i = 0; i < g->n; ++i) {
        ShCoefGlobal coef{fused ? g->f_dc + i * 3 : nullptr, fused ? g->f_rest + i * 45 : nullptr};
        ShEmitGlobal emit{fused ? out->f_dc + i * 3 : nullptr, fused ? out->f_rest + i * 45 : nullptr};
        project_backward_one(i, *g, fused, coef, emit, cam, vk, tiles, grad2d, *out);
    }
}

// the row spans of a large Gaussian's rectangle (gs_math.h big_row_span), as the binning kernels enumerate them: xa[r], xb[r] for the
// rows by0 .. by1 of the rectangle (rect_lo = bx0 | by0 << 16, rect_hi = bx1 | by1 << 16); empty rows have xa > xb
void hm_row_spans(const float* rec16, uint32_t rect_lo, uint32_t rect_hi, const gsplat_view* v, int32_t* xa, int32_t* xb) {
    const ViewK vk = make_viewk(*v);
    const int bx0 = rect_lo & 0xFFFF, by0 = rect_lo >> 16, bx1 = rect_hi & 0xFFFF, by1 = rect_hi >> 16;
    float k4[4];
    big_span_constants(rec16[2], rec16[3], rec16[4], rec16[6], vk.chi_pad, k4);
    const BigSpanK bk = big_span_setup(rec16[0], rec16[1], rec16[6], rec16[7], k4, bx0, bx1);
    for (int y = by0; y <= by1; ++y) {
        const RowSpan sp = big_row_span(bk, y);
        xa[y - by0] = sp.xa; xb[y - by0] = sp.xb;
    }
}

void hm_build_sigma(int64_t n, const float* scale_raw, const float* q_raw, float* sigma) {
    for (int64_t i = 0; i < n; ++i) build_sigma_one(i, scale_raw, q_raw, sigma);
}
void hm_build_sigma_backward(int64_t n, const float* scale_raw, const float* q_raw, const float* grad_sigma, float* gs, float* gq) {
    for (int64_t i = 0; i < n; ++i) build_sigma_backward_one(i, scale_raw, q_raw, grad_sigma, gs, gq);
}
void hm_evaluate_sh(int64_t n, const float* f_dc, const float* f_rest, const float* pts, const float* c2w, float* color) {
    Camera cam; build_camera(c2w, cam);
    for (int64_t i = 0; i < n; ++i) evaluate_sh_one(i, f_dc, f_rest, pts, cam, color);
}
void hm_evaluate_sh_backward(int64_t n, const float* f_dc, const float* f_rest, const float* pts, const float* c2w,
                             const float* grad_color, float* g_dc, float* g_rest, float* g_pts) {
    Camera cam; build_camera(c2w, cam);
    for (int64_t i = 0; i < n; ++i) evaluate_sh_backward_one(i, f_dc, f_rest, pts, cam, grad_color, g_dc, g_rest, g_pts);
}
}
