/*
 * gs_oracle.c -- plain-C, double-precision restatement of the reference hot path, forward AND analytic backward.
 *
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Built by oracle/Makefile into oracle/_build/libgs_oracle.so and loaded only by
 * tests/ (through oracle/c_oracle.py).  It shares no source with the product (csrc/), and differs from it on purpose:
 * explicit eigenvectors, per-pixel evaluation in double, and a TRUE back-to-front backward (suffix sums accumulated from
 * the last contributor to the first) where the product runs front-to-back with "total - prefix".
 *
 * Pinned by tests/test_c_oracle_golden.py against tests/golden/*.npz (outputs of the real reference in float64).
 *
 * Restates (reference file:line):
 *   gaussian_splatting/gaussian.py:24-68, 115-127      quaternion (x,y,z,w) -> R, Sigma = R diag(s)^2 R^T
 *   gaussian_splatting/spherical_harmonics.py:118-166  degree-3 SH colour, channel-major f_rest, sigmoid
 *   gaussian_splatting/utils.py:25-34, 72-96, 180-191  camera transform, frustum + guard band, inv2x2
 *   gaussian_splatting/render.py:106-410               prefilter, EWA projection, eigen clamp, depth sort, AABB tiles,
 *                                                      per-tile compositing, output clamp
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t H, W;
    double fx, fy, cx, cy, near_z, far_z, pix_guard;
    int32_t T;
    double min_conis, chi, alpha_max, alpha_cutoff;
} ora_view;

typedef struct {           /* per visible Gaussian */
    int64_t id;
    double u, v, z, A11, A12, A22, op, rgb[3];
    int tx0, ty0, tx1, ty1;
    /* kept for the backward chain */
    double xc, yc, zc, sg, iz, j00, j11, j02, j12, C[9], a, b, d, l1, l2, f1, f2, ct, st, a2, b2, d2, det, i00, i11;
    double s[3], e[3], q[4], qn, R[9], dir[3], vv[3], vn, Y[16];
    /* 2D gradients accumulated by the rasterizer backward */
    double g_u, g_v, g_A11, g_A12, g_A22, g_op, g_rgb[3];
} ora_rec;

static const double SHK[16] = {0.28209479177387814, 0.4886025119029199, 0.4886025119029199, 0.4886025119029199,
                               1.0925484305920792, 1.0925484305920792, 0.31539156525252005, 1.0925484305920792,
                               0.5462742152960396, 0.5900435899266435, 2.890611442640554, 0.4570457994644658,
                               0.3731763325901154, 0.4570457994644658, 1.445305721320277, 0.5900435899266435};

static double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
static double sigm(double x) { return 1.0 / (1.0 + exp(-x)); }

static void sh_basis(const double d[3], double Y[16]) {
    const double x = d[0], y = d[1], z = d[2], xx = x * x, yy = y * y, zz = z * z;
    Y[0] = SHK[0]; Y[1] = -SHK[1] * y; Y[2] = SHK[2] * z; Y[3] = -SHK[3] * x;
    Y[4] = SHK[4] * x * y; Y[5] = SHK[5] * y * z; Y[6] = SHK[6] * (3 * zz - 1); Y[7] = SHK[7] * x * z; Y[8] = SHK[8] * (xx - yy);
    Y[9] = SHK[9] * y * (3 * xx - yy); Y[10] = SHK[10] * x * y * z; Y[11] = SHK[11] * y * (4 * zz - xx - yy);
    Y[12] = SHK[12] * z * (2 * zz - 3 * xx - 3 * yy); Y[13] = SHK[13] * x * (4 * zz - xx - yy); Y[14] = SHK[14] * z * (xx - yy);
    Y[15] = SHK[15] * x * (xx - 3 * yy);
}

/* dY_k/d(x,y,z) contracted with dY[k] */
static void sh_basis_grad(const double d[3], const double dY[16], double g[3]) {
    const double x = d[0], y = d[1], z = d[2], xx = x * x, yy = y * y, zz = z * z;
    g[0] = -SHK[3] * dY[3] + SHK[4] * y * dY[4] + SHK[7] * z * dY[7] + 2 * SHK[8] * x * dY[8] + 6 * SHK[9] * x * y * dY[9] +
           SHK[10] * y * z * dY[10] - 2 * SHK[11] * x * y * dY[11] - 6 * SHK[12] * x * z * dY[12] +
           SHK[13] * (4 * zz - 3 * xx - yy) * dY[13] + 2 * SHK[14] * x * z * dY[14] + SHK[15] * (3 * xx - 3 * yy) * dY[15];
    g[1] = -SHK[1] * dY[1] + SHK[4] * x * dY[4] + SHK[5] * z * dY[5] - 2 * SHK[8] * y * dY[8] + SHK[9] * (3 * xx - 3 * yy) * dY[9] +
           SHK[10] * x * z * dY[10] + SHK[11] * (4 * zz - xx - 3 * yy) * dY[11] - 6 * SHK[12] * y * z * dY[12] -
           2 * SHK[13] * x * y * dY[13] - 2 * SHK[14] * y * z * dY[14] - 6 * SHK[15] * x * y * dY[15];
    g[2] = SHK[2] * dY[2] + SHK[5] * y * dY[5] + 6 * SHK[6] * z * dY[6] + SHK[7] * x * dY[7] + SHK[10] * x * y * dY[10] +
           8 * SHK[11] * y * z * dY[11] + SHK[12] * (6 * zz - 3 * xx - 3 * yy) * dY[12] + 8 * SHK[13] * x * z * dY[13] +
           SHK[14] * (xx - yy) * dY[14];
}

typedef struct { double z; int64_t id; int32_t k; } ora_key;
static int cmp_key(const void* a, const void* b) {
    const ora_key *p = (const ora_key*)a, *q = (const ora_key*)b;
    if (p->z != q->z) return p->z < q->z ? -1 : 1;
    return p->id < q->id ? -1 : (p->id > q->id ? 1 : 0);
}

/* Returns 0 (image rendered), 10 (nothing survives the culls: zero image, zero gradients), 11 (survivors, none on screen:
 * the reference raises), -1 (allocation failure).  Gradient outputs may be NULL together with grad_image. */
int ora_render(int64_t n, const double* pos, const double* f_dc, const double* f_rest, const double* opacity_raw,
               const double* scale_raw, const double* q_raw, const double* c2w, const ora_view* vw, double* image,
               const double* grad_image, double* g_pos, double* g_f_dc, double* g_f_rest, double* g_opacity_raw,
               double* g_scale_raw, double* g_q_raw, int64_t* counts /* [2]: V, P */) {
    const int H = vw->H, W = vw->W, T = vw->T;
    const int tiles_x = (W + T - 1) / T, tiles_y = (H + T - 1) / T, ntile = tiles_x * tiles_y;
    memset(image, 0, sizeof(double) * (size_t)H * W * 3);
    if (grad_image) {
        memset(g_pos, 0, sizeof(double) * n * 3); memset(g_f_dc, 0, sizeof(double) * n * 3); memset(g_f_rest, 0, sizeof(double) * n * 45);
        memset(g_opacity_raw, 0, sizeof(double) * n); memset(g_scale_raw, 0, sizeof(double) * n * 3); memset(g_q_raw, 0, sizeof(double) * n * 4);
    }
    counts[0] = counts[1] = 0;
    /* w2c = [R^T | -R^T t] */
    double Wm[9], tr[3], eye[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Wm[i * 3 + j] = c2w[j * 4 + i];
    for (int i = 0; i < 3; ++i) eye[i] = c2w[i * 4 + 3];
    for (int i = 0; i < 3; ++i) tr[i] = -(Wm[i * 3] * eye[0] + Wm[i * 3 + 1] * eye[1] + Wm[i * 3 + 2] * eye[2]);
    const double gl = -vw->pix_guard - vw->cx, gr = W + vw->pix_guard - vw->cx, gt = -vw->pix_guard - vw->cy, gb = H + vw->pix_guard - vw->cy;

    ora_rec* rec = (ora_rec*)malloc(sizeof(ora_rec) * (size_t)(n > 0 ? n : 1));
    if (!rec) return -1;
    int64_t nsurv = 0, nvis = 0, npairs = 0;
    for (int64_t i = 0; i < n; ++i) {
        ora_rec r; memset(&r, 0, sizeof(r)); r.id = i;
        r.sg = sigm(opacity_raw[i]); r.op = clampd(r.sg, 0.0, 0.999);
        if (!(r.op >= vw->alpha_cutoff * 0.5)) continue;                                         /* render.py:106-117 */
        const double* p = pos + i * 3;
        r.xc = Wm[0] * p[0] + Wm[1] * p[1] + Wm[2] * p[2] + tr[0];
        r.yc = Wm[3] * p[0] + Wm[4] * p[1] + Wm[5] * p[2] + tr[1];
        r.zc = Wm[6] * p[0] + Wm[7] * p[1] + Wm[8] * p[2] + tr[2];
        if (!(r.zc > 0 && r.zc > vw->near_z && r.zc < vw->far_z && vw->fx * r.xc > r.zc * gl && vw->fx * r.xc < r.zc * gr &&
              vw->fy * r.yc > r.zc * gt && vw->fy * r.yc < r.zc * gb)) continue;                  /* utils.py:72-96 */
        r.z = r.zc; r.u = vw->fx * r.xc / r.zc + vw->cx; r.v = vw->fy * r.yc / r.zc + vw->cy;
        /* Sigma (gaussian.py:115-127) */
        for (int k = 0; k < 3; ++k) { r.e[k] = exp(scale_raw[i * 3 + k]); r.s[k] = r.e[k] < 1e-6 ? 1e-6 : r.e[k]; }
        const double* qr = q_raw + i * 4;
        r.qn = sqrt(qr[0] * qr[0] + qr[1] * qr[1] + qr[2] * qr[2] + qr[3] * qr[3]);
        for (int k = 0; k < 4; ++k) r.q[k] = qr[k] / (r.qn + 1e-9);
        {
            const double x = r.q[0], y = r.q[1], z = r.q[2], w = r.q[3];
            double* R = r.R;
            R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
            R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
            R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
        }
        double S[9];
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
            double acc = 0; for (int k = 0; k < 3; ++k) acc += r.R[a * 3 + k] * r.s[k] * r.s[k] * r.R[b * 3 + k];
            S[a * 3 + b] = acc;
        }
        /* Sigma_c = W S W^T, J, Sigma_2D (render.py:156-175) */
        double M[9];
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { double acc = 0; for (int k = 0; k < 3; ++k) acc += Wm[a * 3 + k] * S[k * 3 + b]; M[a * 3 + b] = acc; }
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { double acc = 0; for (int k = 0; k < 3; ++k) acc += M[a * 3 + k] * Wm[b * 3 + k]; r.C[a * 3 + b] = acc; }
        r.iz = 1.0 / (r.zc < 1e-6 ? 1e-6 : r.zc);
        r.j00 = vw->fx * r.iz; r.j11 = vw->fy * r.iz; r.j02 = -vw->fx * r.xc * r.iz * r.iz; r.j12 = -vw->fy * r.yc * r.iz * r.iz;
        {
            const double* C = r.C;
            const double J[2][3] = {{r.j00, 0, r.j02}, {0, r.j11, r.j12}};
            double JC[2][3];
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b) { double acc = 0; for (int k = 0; k < 3; ++k) acc += J[a][k] * C[k * 3 + b]; JC[a][b] = acc; }
            double S2[2][2];
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) { double acc = 0; for (int k = 0; k < 3; ++k) acc += JC[a][k] * J[b][k]; S2[a][b] = acc; }
            r.a = S2[0][0]; r.d = S2[1][1]; r.b = 0.5 * (S2[0][1] + S2[1][0]);
        }
        /* eigen decomposition of the symmetric 2x2 (explicit angle), clamp, recomposition (render.py:177-179) */
        {
            const double th = 0.5 * atan2(2 * r.b, r.a - r.d);
            r.ct = cos(th); r.st = sin(th);
            r.l1 = r.a * r.ct * r.ct + 2 * r.b * r.ct * r.st + r.d * r.st * r.st;       /* along (c, s): the larger one */
            r.l2 = r.a * r.st * r.st - 2 * r.b * r.ct * r.st + r.d * r.ct * r.ct;
            r.f1 = clampd(r.l1, 1e-6, 1e4); r.f2 = clampd(r.l2, 1e-6, 1e4);
            r.a2 = r.f1 * r.ct * r.ct + r.f2 * r.st * r.st;
            r.b2 = (r.f1 - r.f2) * r.ct * r.st;
            r.d2 = r.f1 * r.st * r.st + r.f2 * r.ct * r.ct;
        }
        if (!(isfinite(r.a2) && isfinite(r.b2) && isfinite(r.d2))) continue;                      /* render.py:187-201 */
        ++nsurv;
        /* conic (utils.py:180-191, render.py:307-315) */
        r.det = r.a2 * r.d2 - r.b2 * r.b2;
        const double sdet = r.det < 1e-12 ? 1e-12 : r.det;
        r.i00 = r.d2 / sdet; r.i11 = r.a2 / sdet;
        r.A11 = r.i00 < vw->min_conis ? vw->min_conis : r.i00;
        r.A22 = r.i11 < vw->min_conis ? vw->min_conis : r.i11;
        r.A12 = -r.b2 / sdet;
        /* radius, AABB, on-screen (render.py:227-258) */
        const double lam = clampd(r.f1 > r.f2 ? r.f1 : r.f2, 1e-12, 1e4);
        const double rad = ceil(2.5 * sqrt(lam));
        const double umin = floor(r.u - rad), umax = floor(r.u + rad), vmin = floor(r.v - rad), vmax = floor(r.v + rad);
        if (!(umax >= 0 && umin < W && vmax >= 0 && vmin < H)) continue;
        r.tx0 = (int)clampd(umin, 0, W - 1) / T; r.tx1 = (int)clampd(umax, 0, W - 1) / T;
        r.ty0 = (int)clampd(vmin, 0, H - 1) / T; r.ty1 = (int)clampd(vmax, 0, H - 1) / T;
        /* colour (spherical_harmonics.py:118-166) */
        for (int k = 0; k < 3; ++k) r.vv[k] = p[k] - eye[k];
        r.vn = sqrt(r.vv[0] * r.vv[0] + r.vv[1] * r.vv[1] + r.vv[2] * r.vv[2]);
        for (int k = 0; k < 3; ++k) r.dir[k] = r.vv[k] / (r.vn + 1e-8);
        sh_basis(r.dir, r.Y);
        for (int ch = 0; ch < 3; ++ch) {
            double acc = f_dc[i * 3 + ch] * r.Y[0];
            for (int k = 1; k < 16; ++k) acc += f_rest[i * 45 + ch * 15 + (k - 1)] * r.Y[k];
            r.rgb[ch] = sigm(acc);
        }
        npairs += (int64_t)(r.tx1 - r.tx0 + 1) * (r.ty1 - r.ty0 + 1);
        rec[nvis++] = r;
    }
    counts[0] = nvis; counts[1] = npairs;
    if (nsurv == 0) { free(rec); return 10; }
    if (nvis == 0) { free(rec); return 11; }

    /* per-tile lists, sorted by (depth, index) (render.py:211-303) */
    int64_t* start = (int64_t*)calloc((size_t)ntile + 1, sizeof(int64_t));
    ora_key* keys = (ora_key*)malloc(sizeof(ora_key) * (size_t)npairs);
    int64_t* fill = (int64_t*)calloc((size_t)ntile, sizeof(int64_t));
    if (!start || !keys || !fill) { free(rec); free(start); free(keys); free(fill); return -1; }
    for (int64_t k = 0; k < nvis; ++k)
        for (int ty = rec[k].ty0; ty <= rec[k].ty1; ++ty) for (int tx = rec[k].tx0; tx <= rec[k].tx1; ++tx) start[ty * tiles_x + tx + 1]++;
    for (int t = 0; t < ntile; ++t) start[t + 1] += start[t];
    for (int64_t k = 0; k < nvis; ++k)
        for (int ty = rec[k].ty0; ty <= rec[k].ty1; ++ty) for (int tx = rec[k].tx0; tx <= rec[k].tx1; ++tx) {
            const int t = ty * tiles_x + tx;
            ora_key* e = keys + start[t] + fill[t]++;
            e->z = rec[k].z; e->id = rec[k].id; e->k = (int32_t)k;
        }
    for (int t = 0; t < ntile; ++t) qsort(keys + start[t], (size_t)(start[t + 1] - start[t]), sizeof(ora_key), cmp_key);

    /* compositing (render.py:325-410) and its backward, pixel by pixel */
    int64_t maxlen = 1;
    for (int t = 0; t < ntile; ++t) if (start[t + 1] - start[t] > maxlen) maxlen = start[t + 1] - start[t];
#pragma omp parallel
    {
        double *al = (double*)malloc(sizeof(double) * maxlen), *Tt = (double*)malloc(sizeof(double) * maxlen),
               *gg = (double*)malloc(sizeof(double) * maxlen), *og = (double*)malloc(sizeof(double) * maxlen);
#pragma omp for schedule(dynamic, 4)
        for (int t = 0; t < ntile; ++t) {
            const int64_t s0 = start[t], cnt = start[t + 1] - s0;
            const int txi = t % tiles_x, tyi = t / tiles_x;
            for (int py = tyi * T; py < (tyi + 1) * T && py < H; ++py)
                for (int px = txi * T; px < (txi + 1) * T && px < W; ++px) {
                    double Tr = 1.0, C[3] = {0, 0, 0};
                    int64_t used = cnt;        /* entries up to the one that kills the pixel (T <= 5e-5): every later one has w = 0 and, the
                                                  suffix sum behind it being 0, d alpha = 0 -- skipping them changes no bit of any result */
                    for (int64_t j = 0; j < cnt; ++j) {
                        const ora_rec* r = rec + keys[s0 + j].k;
                        const double du = px - r->u, dv = py - r->v;
                        const double q = r->A11 * du * du + 2 * r->A12 * du * dv + r->A22 * dv * dv;
                        const double g = q <= vw->chi ? exp(-0.5 * (q < vw->chi ? q : vw->chi)) : 0.0;
                        double a = r->op * g; og[j] = a;
                        if (a > vw->alpha_max) a = vw->alpha_max;
                        if (!(a >= vw->alpha_cutoff)) a = 0.0;
                        al[j] = a; Tt[j] = Tr; gg[j] = g;
                        const double w = (Tr > 5e-5) ? a * Tr : 0.0;
                        C[0] += w * r->rgb[0]; C[1] += w * r->rgb[1]; C[2] += w * r->rgb[2];
                        Tr *= (1.0 - a);
                        if (!(Tr > 5e-5)) { used = j + 1; break; }
                    }
                    double* o = image + ((int64_t)py * W + px) * 3;
                    for (int c = 0; c < 3; ++c) o[c] = clampd(C[c], 0.0, 1.0);
                    if (!grad_image) continue;
                    double G[3];
                    for (int c = 0; c < 3; ++c) G[c] = (C[c] >= 0.0 && C[c] <= 1.0) ? grad_image[((int64_t)py * W + px) * 3 + c] : 0.0;
                    double suffix = 0.0;                               /* sum over k > i of w_k (c_k . G) */
                    for (int64_t j = used - 1; j >= 0; --j) {
                        ora_rec* r = rec + keys[s0 + j].k;
                        const int alive = Tt[j] > 5e-5;
                        const double sdot = r->rgb[0] * G[0] + r->rgb[1] * G[1] + r->rgb[2] * G[2];
                        const double w = alive ? al[j] * Tt[j] : 0.0;
                        double dal = (alive ? Tt[j] * sdot : 0.0) - suffix / (1.0 - al[j]);
                        suffix += w * sdot;
                        if (w != 0.0) {
                            for (int c = 0; c < 3; ++c) {
#pragma omp atomic
                                r->g_rgb[c] += w * G[c];
                            }
                        }
                        if (!(og[j] <= vw->alpha_max) || !(al[j] >= vw->alpha_cutoff) || al[j] == 0.0) continue;
                        if (dal == 0.0) continue;
                        const double du = px - r->u, dv = py - r->v;
                        const double dq = -0.5 * gg[j] * r->op * dal;     /* q <= chi here */
#pragma omp atomic
                        r->g_op += dal * gg[j];
#pragma omp atomic
                        r->g_A11 += du * du * dq;
#pragma omp atomic
                        r->g_A12 += 2 * du * dv * dq;
#pragma omp atomic
                        r->g_A22 += dv * dv * dq;
#pragma omp atomic
                        r->g_u -= (2 * r->A11 * du + 2 * r->A12 * dv) * dq;
#pragma omp atomic
                        r->g_v -= (2 * r->A12 * du + 2 * r->A22 * dv) * dq;
                    }
                }
        }
        free(al); free(Tt); free(gg); free(og);
    }

    if (grad_image) {
        for (int64_t k = 0; k < nvis; ++k) {
            const ora_rec* r = rec + k;
            const int64_t i = r->id;
            /* opacity */
            g_opacity_raw[i] = (r->sg <= 0.999) ? r->g_op * r->sg * (1 - r->sg) : 0.0;
            /* conic -> clamped covariance M = [[a2,b2],[b2,d2]] (full-matrix convention, b and c separate) */
            const double g00 = (r->i00 >= vw->min_conis) ? r->g_A11 : 0.0, g11 = (r->i11 >= vw->min_conis) ? r->g_A22 : 0.0, g01 = r->g_A12;
            const double sdet = r->det < 1e-12 ? 1e-12 : r->det;
            double Ga = g11 / sdet, Gd = g00 / sdet, Gb = -g01 / sdet, Gc = 0.0;
            const double gsd = -(g00 * r->d2 - g01 * r->b2 + g11 * r->a2) / (sdet * sdet);
            const double gdet = (r->det >= 1e-12) ? gsd : 0.0;
            Ga += r->d2 * gdet; Gd += r->a2 * gdet; Gb += -r->b2 * gdet; Gc += -r->b2 * gdet;
            double Gs[2][2] = {{Ga, 0.5 * (Gb + Gc)}, {0.5 * (Gb + Gc), Gd}};
            /* eigen clamp: G = V [(V^T Gs V) o K] V^T */
            {
                const double c = r->ct, s = r->st;
                const double V[2][2] = {{c, -s}, {s, c}};
                double Tm[2][2], Gt[2][2];
                for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) Tm[a][b] = Gs[a][0] * V[0][b] + Gs[a][1] * V[1][b];
                for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) Gt[a][b] = V[0][a] * Tm[0][b] + V[1][a] * Tm[1][b];
                const double k11 = (r->l1 >= 1e-6 && r->l1 <= 1e4) ? 1.0 : 0.0, k22 = (r->l2 >= 1e-6 && r->l2 <= 1e4) ? 1.0 : 0.0;
                const double k12 = (r->l1 != r->l2) ? (r->f1 - r->f2) / (r->l1 - r->l2) : k11;
                Gt[0][0] *= k11; Gt[1][1] *= k22; Gt[0][1] *= k12; Gt[1][0] *= k12;
                for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) Tm[a][b] = V[a][0] * Gt[0][b] + V[a][1] * Gt[1][b];
                for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) Gs[a][b] = Tm[a][0] * V[b][0] + Tm[a][1] * V[b][1];
            }
            /* S2 = J C J^T */
            const double J[2][3] = {{r->j00, 0, r->j02}, {0, r->j11, r->j12}};
            double GJ[2][3], dC[9], JC[2][3], dJ[2][3];
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b) GJ[a][b] = Gs[a][0] * J[0][b] + Gs[a][1] * J[1][b];
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) dC[a * 3 + b] = J[0][a] * GJ[0][b] + J[1][a] * GJ[1][b];
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b) { double acc = 0; for (int k2 = 0; k2 < 3; ++k2) acc += J[a][k2] * r->C[k2 * 3 + b]; JC[a][b] = acc; }
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 3; ++b) dJ[a][b] = 2 * (Gs[a][0] * JC[0][b] + Gs[a][1] * JC[1][b]);
            double dxc = -vw->fx * r->iz * r->iz * dJ[0][2], dyc = -vw->fy * r->iz * r->iz * dJ[1][2];
            const double diz = vw->fx * dJ[0][0] + vw->fy * dJ[1][1] - 2 * vw->fx * r->xc * r->iz * dJ[0][2] - 2 * vw->fy * r->yc * r->iz * dJ[1][2];
            double dzc = (r->zc >= 1e-6) ? -r->iz * r->iz * diz : 0.0;
            dxc += vw->fx / r->zc * r->g_u; dzc += -vw->fx * r->xc / (r->zc * r->zc) * r->g_u;
            dyc += vw->fy / r->zc * r->g_v; dzc += -vw->fy * r->yc / (r->zc * r->zc) * r->g_v;
            double gp[3];
            for (int a = 0; a < 3; ++a) gp[a] = Wm[0 * 3 + a] * dxc + Wm[1 * 3 + a] * dyc + Wm[2 * 3 + a] * dzc;
            /* dSigma = W^T dC W */
            double Tm3[9], GS[9];
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { double acc = 0; for (int k2 = 0; k2 < 3; ++k2) acc += dC[a * 3 + k2] * Wm[k2 * 3 + b]; Tm3[a * 3 + b] = acc; }
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { double acc = 0; for (int k2 = 0; k2 < 3; ++k2) acc += Wm[k2 * 3 + a] * Tm3[k2 * 3 + b]; GS[a * 3 + b] = acc; }
            /* Sigma = R D R^T */
            double GR[9], dR[9];
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) { double acc = 0; for (int k2 = 0; k2 < 3; ++k2) acc += GS[a * 3 + k2] * r->R[k2 * 3 + b]; GR[a * 3 + b] = acc; }
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) dR[a * 3 + b] = 2 * GR[a * 3 + b] * r->s[b] * r->s[b];
            for (int k2 = 0; k2 < 3; ++k2) {
                double dD = 0; for (int a = 0; a < 3; ++a) dD += r->R[a * 3 + k2] * GR[a * 3 + k2];
                g_scale_raw[i * 3 + k2] = (r->e[k2] >= 1e-6) ? 2 * r->s[k2] * dD * r->e[k2] : 0.0;
            }
            {
                const double x = r->q[0], y = r->q[1], z = r->q[2], w = r->q[3];
                double dq[4];
                dq[0] = 2 * (y * (dR[1] + dR[3]) + z * (dR[2] + dR[6]) - 2 * x * (dR[4] + dR[8]) + w * (dR[7] - dR[5]));
                dq[1] = 2 * (x * (dR[1] + dR[3]) + z * (dR[5] + dR[7]) - 2 * y * (dR[0] + dR[8]) + w * (dR[2] - dR[6]));
                dq[2] = 2 * (x * (dR[2] + dR[6]) + y * (dR[5] + dR[7]) - 2 * z * (dR[0] + dR[4]) + w * (dR[3] - dR[1]));
                dq[3] = 2 * (x * (dR[7] - dR[5]) + y * (dR[2] - dR[6]) + z * (dR[3] - dR[1]));
                const double* qr = q_raw + i * 4;
                const double ne = r->qn + 1e-9, dot = dq[0] * qr[0] + dq[1] * qr[1] + dq[2] * qr[2] + dq[3] * qr[3];
                for (int k2 = 0; k2 < 4; ++k2) g_q_raw[i * 4 + k2] = dq[k2] / ne - (r->qn > 0 ? qr[k2] * dot / (r->qn * ne * ne) : 0.0);
            }
            /* colour */
            double dY[16]; memset(dY, 0, sizeof(dY));
            for (int ch = 0; ch < 3; ++ch) {
                const double dpre = r->g_rgb[ch] * r->rgb[ch] * (1 - r->rgb[ch]);
                g_f_dc[i * 3 + ch] = dpre * r->Y[0];
                dY[0] += dpre * f_dc[i * 3 + ch];
                for (int k2 = 1; k2 < 16; ++k2) {
                    g_f_rest[i * 45 + ch * 15 + (k2 - 1)] = dpre * r->Y[k2];
                    dY[k2] += dpre * f_rest[i * 45 + ch * 15 + (k2 - 1)];
                }
            }
            double dd[3];
            sh_basis_grad(r->dir, dY, dd);
            const double ne = r->vn + 1e-8, dot = dd[0] * r->vv[0] + dd[1] * r->vv[1] + dd[2] * r->vv[2];
            for (int a = 0; a < 3; ++a) g_pos[i * 3 + a] = gp[a] + dd[a] / ne - (r->vn > 0 ? r->vv[a] * dot / (r->vn * ne * ne) : 0.0);
        }
    }
    free(rec); free(start); free(keys); free(fill);
    return 0;
}
