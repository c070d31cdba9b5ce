// gs_body.h -- per-Gaussian thread bodies of the projection kernels (forward K1, backward K8) and of the
// two stand-alone ops.  Shared by gsplat_kernels.hip (device) and host_math_check.cpp (host unit test).
#pragma once
#include "../../include/gsplat_mi355x.h"
#include "gs_math.h"

namespace gsm {

struct alignas(16) f4 { float x, y, z, w; };

GS_HD uint32_t f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
GS_HD float u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

struct u2 { uint32_t x, y; };

// Projected record: ONE 64-byte line per Gaussian (the rasterizer gathers records by id; three separate 16-byte
// streams cost three cache lines per gather), plus small per-Gaussian streams for the binning kernels:
//   rec[i] = { (u, v, A11, A12), (A22, opacity, ex, ey), (r, g, b, depth z), (row-span constants of a large Gaussian) }
//   rect[i] = (bx0 | by0 << 16, bx1 | by1 << 16)   inclusive rectangle of lists (16 x 8 pixels each) binned
//   depth[i] = z                                   tiles[i] = lists touched (0 = contributes to no pixel)
//   mask[i]: bit k = the ellipse {q <= chi} touches list k of the rectangle (row-major; all ones for rectangles > 32 lists, whose
//            lists are the row spans of gs_math.h big_row_span instead: tiles[i] counts those)
//   ref_rect[i] (host check only) = the reference's own tile rectangle (F10), T x T tiles
// (ex, ey) are the half-extents of {q <= chi_square_clip}: the rasterizer culls with them at staging time.
struct alignas(64) Rec64 { f4 r0, r1, r2, pad; };

struct Records {
    Rec64* rec;
    u2* rect;
    float* depth;
    uint32_t* tiles;
    uint32_t* mask;
    u2* ref_rect;          // nullable
    uint32_t* ref_tiles;   // nullable
};

GS_HD ViewK make_viewk(const gsplat_view& v) {
    ViewK k;
    k.fx = v.fx; k.fy = v.fy; k.cx = v.cx; k.cy = v.cy;
    k.near_z = v.near_z; k.far_z = v.far_z;
    // computed in double like the Python scalars of utils.py:82-91, then rounded once
    k.gl = (float)(-(double)v.pix_guard - (double)v.cx);
    k.gr = (float)((double)v.W + (double)v.pix_guard - (double)v.cx);
    k.gt = (float)(-(double)v.pix_guard - (double)v.cy);
    k.gb = (float)((double)v.H + (double)v.pix_guard - (double)v.cy);
    k.opacity_min = (float)((double)v.alpha_cutoff * 0.5);
    k.min_conis = v.min_conis; k.chi_clip = v.chi_square_clip; k.alpha_max = v.alpha_max; k.alpha_cutoff = v.alpha_cutoff;
    k.chi_pad = (float)((double)v.chi_square_clip * 1.001 + 1e-4);
    k.H = v.H; k.W = v.W; k.tile = v.tile;
    k.tiles_x = (v.W + v.tile - 1) / v.tile; k.tiles_y = (v.H + v.tile - 1) / v.tile;
    k.lists_x = (v.W + LIST_W - 1) / LIST_W; k.lists_y = (v.H + LIST_H - 1) / LIST_H;
    return k;
}

// Coefficient access in the reference layout: basis 0 from f_dc[N,3], bases 1..15 from f_rest[N,45] channel-major.
struct ShCoefGlobal {
    const float* dc;      // &f_dc[i*3]
    const float* rest;    // &f_rest[i*45]
    GS_HD float operator()(int k, int ch) const { return k == 0 ? dc[ch] : rest[ch * 15 + (k - 1)]; }
};

GS_HD void load_cov6(const float* sigma9, float S[6]) {
    // symmetrised: the projected 2x2 is symmetrised by the reference (render.py:175), which equals projecting sym(Sigma)
    S[0] = sigma9[0]; S[1] = 0.5f * (sigma9[1] + sigma9[3]); S[2] = 0.5f * (sigma9[2] + sigma9[6]);
    S[3] = sigma9[4]; S[4] = 0.5f * (sigma9[5] + sigma9[7]); S[5] = sigma9[8];
}

// One Gaussian's inputs as plain values (fed from global arrays by the host check, from LDS by the kernels).
struct GaussIn {
    float p[3];
    float o_raw;
    float sr[3], qr[4];     // fused inputs
    float S9[9], col[3];    // un-fused inputs (sigma row-major, colour)
};

struct RecOut {             // what K1 stores for one Gaussian
    f4 r0, r1, r2, r3;      // r3: the row-span constants of a large Gaussian (gs_math.h big_span_constants), else zeros
    u2 rect;                // binned lists
    uint32_t tiles;         // number of lists
    uint32_t mask;          // which lists of the rectangle (see Records)
    u2 ref_rect;            // the reference's tile rectangle (F10)
    uint32_t ref_tiles;     // its tile count: the reference's (tile, Gaussian) pairs (F11)
    int vis;
};

struct GradOut {            // what K8 stores for one Gaussian
    float p[3], o_raw, sr[3], qr[4], S9[9], col[3];
};

// K1 core, part 1: everything except the colour.
GS_HD Proj project_geometry(const GaussIn& in, bool fused, const Camera& cam, const ViewK& vk) {
    float S[6];
    CovMid cm;
    if (fused) cov_from_params(in.sr, in.qr, S, cm);
    else load_cov6(in.S9, S);
    Proj o; ProjMid m;
    project_gaussian(in.p, S, in.o_raw, cam, vk, o, m, fused ? &cm : nullptr);
    return o;
}

// K1 core, part 2: colour of a visible Gaussian (SH when fused) and the record.
// with_colour = false (fused inputs only): the SH colour is filled in later by colour_kernel; rgb = 0 here.
// kj (nullable, fused inputs with colour): receives the 12 values sh_colour_jac leaves for the backward.
template <class Coef>
GS_HD RecOut project_finish(const GaussIn& in, const Proj& o, bool fused, Coef coef, const Camera& cam, bool with_colour = true,
                            float* kj = nullptr) {
    RecOut r;
    r.vis = o.vis;
    r.tiles = 0;
    r.mask = 0u;
    r.r0 = r.r1 = r.r2 = r.r3 = f4{0.f, 0.f, 0.f, 0.f};
    r.rect = u2{0u, 0u};
    r.ref_rect = u2{0u, 0u};
    r.ref_tiles = 0;
    if (o.vis == VIS_OK) {
        float rgb[3] = {0.f, 0.f, 0.f};
        if (fused) {
            if (with_colour) {
                ShMid sm;
                sh_basis(in.p, cam.eye, sm);
                if (kj) sh_colour_jac(sm, coef, rgb, kj);
                else sh_colour(sm, coef, rgb);
            }
        } else {
            rgb[0] = in.col[0]; rgb[1] = in.col[1]; rgb[2] = in.col[2];
        }
        r.ref_tiles = (uint32_t)((o.tx1 - o.tx0 + 1) * (o.ty1 - o.ty0 + 1));
        r.ref_rect = u2{(uint32_t)o.tx0 | ((uint32_t)o.ty0 << 16), (uint32_t)o.tx1 | ((uint32_t)o.ty1 << 16)};
        if (o.bx1 >= o.bx0 && o.by1 >= o.by0) {
            const uint32_t area = (uint32_t)((o.bx1 - o.bx0 + 1) * (o.by1 - o.by0 + 1));
            r.mask = o.bmask;
            r.tiles = area > 32u ? (uint32_t)o.btiles : (uint32_t)__builtin_popcount(r.mask);
        }
        r.r0 = f4{o.u, o.v, o.A11, o.A12};
        r.r1 = f4{o.A22, o.opacity, o.ex, o.ey};
        r.r2 = f4{rgb[0], rgb[1], rgb[2], o.z};
        r.r3 = f4{o.bk4[0], o.bk4[1], o.bk4[2], o.bk4[3]};
        if (r.tiles) r.rect = u2{(uint32_t)o.bx0 | ((uint32_t)o.by0 << 16), (uint32_t)o.bx1 | ((uint32_t)o.by1 << 16)};
    }
    return r;
}

template <class Coef>
GS_HD RecOut project_core(const GaussIn& in, bool fused, Coef coef, const Camera& cam, const ViewK& vk) {
    return project_finish(in, project_geometry(in, fused, cam, vk), fused, coef, cam);
}

// K8 core.  r9 = (g_u, g_v, g_A11, g_A12, g_A22, g_opacity, g_r, g_g, g_b) of a visible Gaussian.
// emit_sh(k, ch, val) receives dL/d(SH coefficient) for k = 0..15 (k = 0 -> f_dc); it is called for every (k, ch),
// with zeros for a Gaussian that is not visible, so that every output row is written.
// moments = true (what raster_backward_kernel accumulates): r9[0..5] = (Mx, My, Mxx, Mxy, Myy, M0), the moments sum du^a dv^b a
// over the pixels of a = dL/d alpha * g.  dL/d opacity = M0, and dL/dq = -0.5 o a (alpha = o exp(-q/2)): with
// q = A11 du^2 + 2 A12 du dv + A22 dv^2 and du = px - u:  d u = o (A11 Mx + A12 My), d v = o (A12 Mx + A22 My),
// d A11 = -0.5 o Mxx, d A12 = -o Mxy, d A22 = -0.5 o Myy.  moments = false: r9[0..5] are those gradients themselves.
// kj (nullable): the 12 values the forward saved with sh_colour_jac; then `coef` is not read.
template <class Coef, class Emit>
GS_HD GradOut project_backward_core(const GaussIn& in, bool fused, Coef coef, Emit emit_sh, const Camera& cam, const ViewK& vk,
                                    bool vis, const float r9[9], bool moments = false, const float* kj = nullptr) {
    GradOut g;
    for (int k = 0; k < 3; ++k) { g.p[k] = 0.f; g.sr[k] = 0.f; g.col[k] = 0.f; }
    for (int k = 0; k < 4; ++k) g.qr[k] = 0.f;
    for (int k = 0; k < 9; ++k) g.S9[k] = 0.f;
    g.o_raw = 0.f;
    if (vis) {
        float S[6];
        CovMid cm;
        if (fused) cov_from_params(in.sr, in.qr, S, cm);
        else load_cov6(in.S9, S);
        Proj o; ProjMid m;
        project_gaussian(in.p, S, in.o_raw, cam, vk, o, m, fused ? &cm : nullptr);
        float gu = r9[0], gv = r9[1], ga = r9[2], gb = r9[3], gc = r9[4];
        if (moments) {
            gu = o.opacity * (o.A11 * r9[0] + o.A12 * r9[1]);
            gv = o.opacity * (o.A12 * r9[0] + o.A22 * r9[1]);
            ga = -0.5f * o.opacity * r9[2];
            gb = -o.opacity * r9[3];
            gc = -0.5f * o.opacity * r9[4];
        }
        project_gaussian_backward(m, o, cam, vk, gu, gv, ga, gb, gc, r9[5], g.p, g.S9, g.o_raw);
        g.col[0] = r9[6]; g.col[1] = r9[7]; g.col[2] = r9[8];
        if (fused) {
            cov_from_params_backward(in.qr, cm, g.S9, g.sr, g.qr);
            ShMid sm;
            sh_basis(in.p, cam.eye, sm);
            float gps[3];
            if (kj) {
                sh_colour_backward_jac(sm, kj, g.col, emit_sh, gps);
            } else {
                float rgb[3];
                sh_colour(sm, coef, rgb);
                sh_colour_backward(sm, coef, rgb, g.col, emit_sh, gps);
            }
            g.p[0] += gps[0]; g.p[1] += gps[1]; g.p[2] += gps[2];
        }
    } else if (fused) {
        for (int k = 0; k < 16; ++k)
            for (int ch = 0; ch < 3; ++ch) emit_sh(k, ch, 0.f);
    }
    return g;
}

GS_HD GaussIn load_gauss_global(int64_t i, const gsplat_gaussians& g, bool fused) {
    GaussIn in;
    for (int k = 0; k < 3; ++k) in.p[k] = g.pos[i * 3 + k];
    in.o_raw = g.opacity_raw[i];
    if (fused) {
        for (int k = 0; k < 3; ++k) in.sr[k] = g.scale_raw[i * 3 + k];
        for (int k = 0; k < 4; ++k) in.qr[k] = g.q_raw[i * 4 + k];
    } else {
        for (int k = 0; k < 9; ++k) in.S9[k] = g.sigma[i * 9 + k];
        for (int k = 0; k < 3; ++k) in.col[k] = g.color[i * 3 + k];
    }
    return in;
}

// Reference-layout wrappers (host unit test; the kernels stage through LDS instead, see gsplat_kernels.hip).
template <class Coef>
GS_HD int project_one(int64_t i, const gsplat_gaussians& g, bool fused, Coef coef, const Camera& cam, const ViewK& vk,
                      const Records& out) {
    const RecOut r = project_core(load_gauss_global(i, g, fused), fused, coef, cam, vk);
    if (r.vis == VIS_OK) {
        out.rec[i].r0 = r.r0; out.rec[i].r1 = r.r1; out.rec[i].r2 = r.r2; out.rec[i].pad = r.r3;
        out.rect[i] = r.rect; out.depth[i] = r.r2.w; out.mask[i] = r.mask;
        if (out.ref_rect) out.ref_rect[i] = r.ref_rect;
    }
    out.tiles[i] = r.tiles;
    if (out.ref_tiles) out.ref_tiles[i] = r.ref_tiles;
    return r.vis;
}

template <class Coef, class Emit>
GS_HD void project_backward_one(int64_t i, const gsplat_gaussians& g, bool fused, Coef coef, Emit emit_sh, const Camera& cam,
                                const ViewK& vk, const uint32_t* tiles, const float* grad2d,
                                const gsplat_gaussian_grads& out) {
    const GradOut o = project_backward_core(load_gauss_global(i, g, fused), fused, coef, emit_sh, cam, vk, tiles[i] != 0,
                                            grad2d + i * 16);
    out.pos[i * 3 + 0] = o.p[0]; out.pos[i * 3 + 1] = o.p[1]; out.pos[i * 3 + 2] = o.p[2];
    out.opacity_raw[i] = o.o_raw;
    if (fused) {
        for (int k = 0; k < 3; ++k) out.scale_raw[i * 3 + k] = o.sr[k];
        for (int k = 0; k < 4; ++k) out.q_raw[i * 4 + k] = o.qr[k];
    } else {
        for (int k = 0; k < 9; ++k) out.sigma[i * 9 + k] = o.S9[k];
        for (int k = 0; k < 3; ++k) out.color[i * 3 + k] = o.col[k];
    }
}

// Stand-alone build_sigma_from_params (gaussian.py:71-127) and its backward.
GS_HD void build_sigma_one(int64_t i, const float* scale_raw, const float* q_raw, float* sigma) {
    const float sr[3] = {scale_raw[i * 3 + 0], scale_raw[i * 3 + 1], scale_raw[i * 3 + 2]};
    const float qr[4] = {q_raw[i * 4 + 0], q_raw[i * 4 + 1], q_raw[i * 4 + 2], q_raw[i * 4 + 3]};
    float S[6]; CovMid cm;
    cov_from_params(sr, qr, S, cm);
    float* o = sigma + i * 9;
    o[0] = S[0]; o[1] = S[1]; o[2] = S[2]; o[3] = S[1]; o[4] = S[3]; o[5] = S[4]; o[6] = S[2]; o[7] = S[4]; o[8] = S[5];
}

GS_HD void build_sigma_backward_one(int64_t i, const float* scale_raw, const float* q_raw, const float* grad_sigma,
                                    float* g_scale_raw, float* g_q_raw) {
    const float sr[3] = {scale_raw[i * 3 + 0], scale_raw[i * 3 + 1], scale_raw[i * 3 + 2]};
    const float qr[4] = {q_raw[i * 4 + 0], q_raw[i * 4 + 1], q_raw[i * 4 + 2], q_raw[i * 4 + 3]};
    float S[6]; CovMid cm;
    cov_from_params(sr, qr, S, cm);
    // Sigma = R D R^T is symmetric in its own formula: only the symmetric part of the incoming gradient matters
    const float* gi = grad_sigma + i * 9;
    float G[9];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) G[a * 3 + b] = 0.5f * (gi[a * 3 + b] + gi[b * 3 + a]);
    float gs[3], gq[4];
    cov_from_params_backward(qr, cm, G, gs, gq);
    for (int k = 0; k < 3; ++k) g_scale_raw[i * 3 + k] = gs[k];
    for (int k = 0; k < 4; ++k) g_q_raw[i * 4 + k] = gq[k];
}

// Stand-alone evaluate_sh (spherical_harmonics.py:70-166) and its backward.
GS_HD void evaluate_sh_one(int64_t i, const float* f_dc, const float* f_rest, const float* points, const Camera& cam,
                           float* color) {
    const float p[3] = {points[i * 3 + 0], points[i * 3 + 1], points[i * 3 + 2]};
    ShMid sm;
    sh_basis(p, cam.eye, sm);
    float rgb[3];
    sh_colour(sm, ShCoefGlobal{f_dc + i * 3, f_rest + i * 45}, rgb);
    color[i * 3 + 0] = rgb[0]; color[i * 3 + 1] = rgb[1]; color[i * 3 + 2] = rgb[2];
}

struct ShEmitGlobal {
    float* dc;      // &grad_f_dc[i*3]
    float* rest;    // &grad_f_rest[i*45]
    GS_HD void operator()(int k, int ch, float v) const {
        if (k == 0) dc[ch] = v; else rest[ch * 15 + (k - 1)] = v;
    }
};

GS_HD void evaluate_sh_backward_one(int64_t i, const float* f_dc, const float* f_rest, const float* points, const Camera& cam,
                                    const float* grad_color, float* g_f_dc, float* g_f_rest, float* g_points) {
    const float p[3] = {points[i * 3 + 0], points[i * 3 + 1], points[i * 3 + 2]};
    ShMid sm;
    sh_basis(p, cam.eye, sm);
    ShCoefGlobal coef{f_dc + i * 3, f_rest + i * 45};
    float rgb[3];
    sh_colour(sm, coef, rgb);
    const float gc[3] = {grad_color[i * 3 + 0], grad_color[i * 3 + 1], grad_color[i * 3 + 2]};
    float gp[3];
    sh_colour_backward(sm, coef, rgb, gc, ShEmitGlobal{g_f_dc + i * 3, g_f_rest + i * 45}, gp);
    g_points[i * 3 + 0] = gp[0]; g_points[i * 3 + 1] = gp[1]; g_points[i * 3 + 2] = gp[2];
}

}  // namespace gsm

#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
