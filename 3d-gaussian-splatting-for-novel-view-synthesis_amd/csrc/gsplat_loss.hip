// gsplat_loss.hip -- fused L1 + SSIM training loss (value AND gradient w.r.t. the prediction in one pass).
//
// SURVEY.md §8(f) "next" row 1: reference gaussian_splatting/losses.py:27-185 (l1_loss, ssim_loss, _ssim_single_channel,
// _create_gaussian_window, compute_loss).  The reference runs 15 conv2d calls on permuted copies per view and autograd
// replays them; here two kernels (value + partial-derivative maps, then the gradient) read pred/target twice and write the gradient once.
//
//   L = l1w * mean|x - y| + sw * (1 - mean SSIM),      SSIM per channel with an 11 x 11 Gaussian window (sigma 1.5),
//   zero padding, C1 = 0.01^2, C2 = 0.03^2; the 2-D window is the outer product of a normalised 1-D Gaussian, so every
//   convolution is done separably.
//
// Gradient: with mu1 = w*x, mu2 = w*y, E11 = w*x^2, E22 = w*y^2, E12 = w*xy (w* = windowed sum) and
//   S = A1 A2 / (B1 B2),  A1 = 2 mu1 mu2 + C1, A2 = 2 (E12 - mu1 mu2) + C2, B1 = mu1^2 + mu2^2 + C1,
//   B2 = E11 - mu1^2 + E22 - mu2^2 + C2:
//   dS/dmu1 = 2 mu2 (A2 - A1) / (B1 B2) - 2 mu1 S / B1 + 2 mu1 S / B2,   dS/dE11 = -S / B2,   dS/dE12 = 2 A1 / (B1 B2),
//   d(sum S)/dx(q) = (w * dS/dmu1)(q) + 2 x(q) (w * dS/dE11)(q) + y(q) (w * dS/dE12)(q)     (w is symmetric).
//
// TWO kernels per call, 32 x 16 output tiles, the three channels one after the other through the same LDS buffers:
//   loss_stats_kernel   x, y with a 5-pixel halo -> five horizontally filtered planes -> the SSIM value (summed) and the three
//                       partial-derivative maps dS/dmu1, dS/dE11, dS/dE12 of the tile, written to scratch (planar, 36 B per pixel);
//   loss_grad_kernel    the three maps with a 5-pixel halo -> horizontal, vertical filter -> the gradient (+ the L1 term), written as
//                       whole 12-byte pixels.
// One kernel per 16 x 16 tile with BOTH halos (a 36 x 36 input region, round 2) filtered 3.7 / 2.6 / 1.6 times as many values as
// the tile holds in its three first passes: 432 FMAs per value, 0.24 ms at 1080p, VALU-bound.  Cut in two at the maps, each half
// has one 5-pixel halo: 230 FMAs per value for 72 B per pixel of extra traffic that stays in the 256 MB cache.
// Sums: one pair per workgroup, added up in a fixed order by block 0 of the second kernel (no atomics: the value is reproducible).
#include <cstdio>

#include <hip/hip_runtime.h>

#include "../../include/gsplat_mi355x.h"

namespace {

constexpr int TW = 32, TH = 16, R = 5, TAPS = 11;
constexpr int W1 = TW + 2 * R, H1 = TH + 2 * R;      // region with one halo   42 x 26
constexpr int THREADS = 256;
constexpr int G = 4;                                 // outputs per thread along a row in the horizontal passes
constexpr int GV = 2;                                // rows per thread in the vertical passes
static_assert(TW % G == 0 && TH % GV == 0 && (TH / GV) * TW == THREADS, "tile shape vs register blocking");

// LDS access patterns.  Horizontal passes: lanes run over ROWS (row index fastest), every lane reads G + 10 = 14 consecutive words of
// its own row as 3 x 16 + 8 bytes and writes G = 4 words as 16 bytes; the row pitches are multiples of 4 words (alignment) chosen so
// that the 8 lanes a 16-byte access serves per cycle fall on 8 disjoint groups of 4 banks (44 r mod 32 = 0, 12, 24, 4, 16, 28, 8, 20;
// 36 r mod 32 = 0, 4, ... 28).  (Odd pitches with 4-byte accesses, the first version: 14 + 14 instead of 4 + 4 LDS instructions per
// thread and pass, 30-37 % of the LDS cycles bank conflicts.)  Vertical passes: lanes run over the 32 columns of a row group, two
// row groups per wave: consecutive words.
constexpr int XP = 44;           // pitch of the input planes / the map planes (W1 = 42 columns)
constexpr int HP = 36;           // pitch of the horizontally filtered planes (TW = 32 columns)
static_assert(XP >= W1 && XP % 4 == 0 && HP >= TW && HP % 4 == 0, "row pitches: 16-byte rows");

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
// 14 consecutive floats of an LDS row from a 16-byte aligned position
__device__ __forceinline__ void load14(const float* p, float (&a)[G + 10]) {
    const f4v v0 = *reinterpret_cast<const f4v*>(p), v1 = *reinterpret_cast<const f4v*>(p + 4), v2 = *reinterpret_cast<const f4v*>(p + 8);
    const f2v v3 = *reinterpret_cast<const f2v*>(p + 12);
    a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w; a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
    a[8] = v2.x; a[9] = v2.y; a[10] = v2.z; a[11] = v2.w; a[12] = v3.x; a[13] = v3.y;
}
static_assert(G == 4, "load14 / the 16-byte stores assume four outputs per thread");

__device__ __forceinline__ void gauss_taps(float g[TAPS]) {
    // losses.py:131-155: exp(-(i - 5)^2 / (2 * 1.5^2)), normalised
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TAPS; ++i) { const float d = (float)(i - R); g[i] = expf(-(d * d) / 4.5f); s += g[i]; }
    const float inv = 1.0f / s;
    // (the same eleven values in every lane: kept in SGPRs, they cost no vector register)
#pragma unroll
    for (int i = 0; i < TAPS; ++i) g[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(g[i] * inv)));
}

// region element i (row-major over H1 x W1) of channel ch of a [H][W][3] image, zero outside the image
struct alignas(16) StatsLds {
    float x[3][H1][XP], y[3][H1][XP];     // the three channels of the region, de-interleaved
    float h[5][H1][HP];                   // horizontally filtered x, y, xx, yy, xy of the channel being processed
};
static_assert(sizeof(StatsLds) <= 52 * 1024, "three workgroups per CU");

// Sum of the blocks' partial sums in a fixed order (double), by one workgroup: -> values[3] = (l1, 1 - ssim, total)
__device__ __forceinline__ void finish_sums(const float* __restrict__ partial, int n_blocks, double inv_n, float l1w, float sw,
                                            float* __restrict__ values, double (*red)[THREADS / 64], double scale = 1.0,
                                            float* __restrict__ total_out = nullptr) {
    const int tid = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int k = tid; k < n_blocks; k += THREADS) { a += (double)partial[2 * k]; b += (double)partial[2 * k + 1]; }
    for (int sft = 32; sft > 0; sft >>= 1) { a += __shfl_xor(a, sft); b += __shfl_xor(b, sft); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = a; red[1][tid >> 6] = b; }
    __syncthreads();
    if (tid == 0) {
        a = 0.0; b = 0.0;
        for (int k = 0; k < THREADS / 64; ++k) { a += red[0][k]; b += red[1][k]; }
        const double l1 = a * inv_n, sl = 1.0 - b * inv_n;
        values[0] = (float)(scale * l1); values[1] = (float)(scale * sl); values[2] = (float)(scale * (l1w * l1 + sw * sl));
        if (total_out) *total_out = values[2];
    }
}

// maps: [image][channel][3][H][W] (planar: the halo reads of loss_grad_kernel are rows of consecutive floats)
// partial: [block][2] = the block's sums of |x - y| and of the SSIM values (no atomics: the total is formed in a fixed order)
__global__ __launch_bounds__(THREADS) void loss_stats_kernel(const float* __restrict__ pred, const float* __restrict__ target, int H, int W,
                                                             float* __restrict__ partial, float* __restrict__ maps) {
    __shared__ StatsLds s;
    __shared__ float red[2][THREADS / 64];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int64_t img = (int64_t)blockIdx.z * H * W * 3;
    const int64_t plane = (int64_t)H * W;
    float g[TAPS];
    gauss_taps(g);
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float l1_acc = 0.f, ssim_acc = 0.f;
    {   // the region of all three channels at once: rows of 3 * 42 consecutive floats of the interleaved image, every load of the
        // workgroup in flight together (one exposed round trip per workgroup; fetched channel by channel, with the next channel's
        // prefetch in registers, a workgroup waited for memory three times: 87 -> 63 us was the LDS side, this is the other half)
        // (pixel e = tid + 256 k of the region as (row, column): stepped, not divided -- 256 = 6 * 42 + 4; the three channels of a
        //  pixel are 12 consecutive bytes of the interleaved image: one address per pixel and image)
        constexpr int N = H1 * W1, SLOTS = (N + THREADS - 1) / THREADS;
        static_assert(THREADS == 6 * W1 + 4, "the stepping below");
        float ra[SLOTS][3], rb[SLOTS][3];
        const int r_first = tid / W1, c_first = tid - r_first * W1;
        {
            int r = r_first, c = c_first;
#pragma unroll
            for (int k = 0; k < SLOTS; ++k) {
                const int gy = y0 - R + r, gx = x0 - R + c;
                const bool in = r < H1 && gy >= 0 && gy < H && gx >= 0 && gx < W;      // zero outside the image = the reference's zero padding
                const int64_t o = img + ((int64_t)gy * W + gx) * 3;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) { ra[k][ch] = in ? pred[o + ch] : 0.f; rb[k][ch] = in ? target[o + ch] : 0.f; }
                r += 6; c += 4;
                if (c >= W1) { c -= W1; ++r; }
            }
        }
        {
            int r = r_first, c = c_first;
#pragma unroll
            for (int k = 0; k < SLOTS; ++k) {
                if (r < H1) {
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) { s.x[ch][r][c] = ra[k][ch]; s.y[ch][r][c] = rb[k][ch]; }
                }
                r += 6; c += 4;
                if (c >= W1) { c -= W1; ++r; }
            }
        }
    }
    for (int ch = 0; ch < 3; ++ch) {
        __syncthreads();                                                   // the region is staged / the previous channel's h is read
        // horizontal pass of the five products: G outputs per thread from G + 10 inputs (the products formed once per input)
        for (int i = tid; i < H1 * (TW / G); i += THREADS) {
            const int r = i % H1, c = (i / H1) * G;                        // rows fastest across lanes
            float a[G + 10], b[G + 10], aa[G + 10], bb[G + 10], ab[G + 10];
            load14(&s.x[ch][r][c], a);
            load14(&s.y[ch][r][c], b);
#pragma unroll
            for (int t = 0; t < G + 10; ++t) { aa[t] = a[t] * a[t]; bb[t] = b[t] * b[t]; ab[t] = a[t] * b[t]; }
            f4v o0, o1, o2, o3, o4;
#pragma unroll
            for (int o = 0; o < G; ++o) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const float w = g[t];
                    a0 += w * a[o + t]; a1 += w * b[o + t]; a2 += w * aa[o + t]; a3 += w * bb[o + t]; a4 += w * ab[o + t];
                }
                o0[o] = a0; o1[o] = a1; o2[o] = a2; o3[o] = a3; o4[o] = a4;
            }
            *reinterpret_cast<f4v*>(&s.h[0][r][c]) = o0; *reinterpret_cast<f4v*>(&s.h[1][r][c]) = o1; *reinterpret_cast<f4v*>(&s.h[2][r][c]) = o2;
            *reinterpret_cast<f4v*>(&s.h[3][r][c]) = o3; *reinterpret_cast<f4v*>(&s.h[4][r][c]) = o4;
        }
        __syncthreads();
        // vertical pass -> SSIM value and its partial derivatives on the tile: GV rows per thread; the L1 term rides along
        {
            const int rg = tid / TW, c = tid - rg * TW, r0 = rg * GV;
            float v[5][GV + 10];
#pragma unroll
            for (int k = 0; k < 5; ++k)
#pragma unroll
                for (int t = 0; t < GV + 10; ++t) v[k][t] = s.h[k][r0 + t][c];
#pragma unroll
            for (int o = 0; o < GV; ++o) {
                const int gy = y0 + r0 + o, gx = x0 + c;
                if (gy < H && gx < W) {
                    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
                    for (int t = 0; t < TAPS; ++t) {
                        const float w = g[t];
                        mu1 += w * v[0][o + t]; mu2 += w * v[1][o + t]; e11 += w * v[2][o + t]; e22 += w * v[3][o + t];
                        e12 += w * v[4][o + t];
                    }
                    const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * (e12 - mu1 * mu2) + C2;
                    const float B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = (e11 - mu1 * mu1) + (e22 - mu2 * mu2) + C2;
                    // (B1 >= C1, B2 >= C2 up to rounding: reciprocals by v_rcp_f32, 1 ulp, instead of three IEEE divisions)
                    const float rb1 = __builtin_amdgcn_rcpf(B1), rb2 = __builtin_amdgcn_rcpf(B2);
                    const float ib = rb1 * rb2;
                    const float S = A1 * A2 * ib;
                    ssim_acc += S;
                    l1_acc += fabsf(s.x[ch][r0 + o + R][c + R] - s.y[ch][r0 + o + R][c + R]);
                    if (maps) {
                        float* m = maps + ((int64_t)blockIdx.z * 3 + ch) * 3 * plane + (int64_t)gy * W + gx;
                        const float s2 = 2.f * mu1 * S;
                        m[0] = 2.f * mu2 * (A2 - A1) * ib + s2 * (rb2 - rb1);
                        m[plane] = -S * rb2;
                        m[2 * plane] = 2.f * A1 * ib;
                    }
                }
            }
        }
    }
    // the block's two sums (fixed order inside the block)
    for (int sft = 32; sft > 0; sft >>= 1) { l1_acc += __shfl_xor(l1_acc, sft); ssim_acc += __shfl_xor(ssim_acc, sft); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = l1_acc; red[1][tid >> 6] = ssim_acc; }
    __syncthreads();
    if (tid == 0) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < THREADS / 64; ++k) { a += red[0][k]; b += red[1][k]; }
        const int64_t blk = blockIdx.x + (int64_t)gridDim.x * (blockIdx.y + (int64_t)gridDim.y * blockIdx.z);
        partial[2 * blk] = a;
        partial[2 * blk + 1] = b;
    }
}

struct alignas(16) GradLds {
    float p[3][3][H1][XP];       // per channel: dS/dmu1, dS/dE11, dS/dE12 with the halo (zero outside the image)
    float q[3][H1][HP];          // horizontally filtered maps of the channel being processed
};
static_assert(sizeof(GradLds) <= 52 * 1024 + 256, "three workgroups per CU");

// d(sum S)/dx(q) = (w * dS/dmu1)(q) + 2 x(q) (w * dS/dE11)(q) + y(q) (w * dS/dE12)(q); with the L1 term and the weights -> grad.
// Block (0, 0, 0) also adds up the partial sums loss_stats_kernel left (values[3]).
__global__ __launch_bounds__(THREADS) void loss_grad_kernel(const float* __restrict__ pred, const float* __restrict__ target, int H, int W,
                                                            float l1w, float sw, float inv_n_, double inv_n_d, const float* __restrict__ maps,
                                                            const float* __restrict__ partial, float* __restrict__ values,
                                                            float* __restrict__ grad, const float* __restrict__ upstream) {
    // upstream (nullable device scalar): d L / d total of the caller's graph -- multiplied in here, not by a pass over the gradient
    const float inv_n = upstream ? inv_n_ * *upstream : inv_n_;
    __shared__ GradLds s;
    __shared__ double red[2][THREADS / 64];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int64_t img = (int64_t)blockIdx.z * H * W * 3;
    const int64_t plane = (int64_t)H * W;
    float g[TAPS];
    gauss_taps(g);
    // this thread's GV pixels (rows r0, r0 + 1 of column c): x, y of the three channels, and the gradient as it comes
    const int rg = tid / TW, c = tid - rg * TW, r0 = rg * GV;
    float px[GV][3], py[GV][3], out[GV][3];
    {   // the nine map planes of the region at once (see loss_stats_kernel), then this thread's pixels
        // (pixel e = tid + 256 k of the region as (row, column): stepped, not divided -- 256 = 6 * 42 + 4; one address per pixel, the
        //  nine planes at constant distances)
        constexpr int N = H1 * W1, SLOTS = (N + THREADS - 1) / THREADS;
        static_assert(THREADS == 6 * W1 + 4, "the stepping below");
        const float* m = maps + (int64_t)blockIdx.z * 9 * plane;
        float rp[SLOTS][9];
        const int r_first = tid / W1, c_first = tid - r_first * W1;
        {
            int r = r_first, cc = c_first;
#pragma unroll
            for (int k = 0; k < SLOTS; ++k) {
                const int gy = y0 - R + r, gx = x0 - R + cc;
                const bool in = r < H1 && gy >= 0 && gy < H && gx >= 0 && gx < W;
                const float* mp = m + (int64_t)gy * W + gx;
#pragma unroll
                for (int pl = 0; pl < 9; ++pl) rp[k][pl] = in ? mp[pl * plane] : 0.f;
                r += 6; cc += 4;
                if (cc >= W1) { cc -= W1; ++r; }
            }
        }
#pragma unroll
        for (int o = 0; o < GV; ++o) {
            const int gy = y0 + r0 + o, gx = x0 + c;
            const bool in = gy < H && gx < W;
            const int64_t a = img + ((int64_t)gy * W + gx) * 3;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) { px[o][ch] = in ? pred[a + ch] : 0.f; py[o][ch] = in ? target[a + ch] : 0.f; out[o][ch] = 0.f; }
        }
        {
            int r = r_first, cc = c_first;
#pragma unroll
            for (int k = 0; k < SLOTS; ++k) {
                if (r < H1) {
#pragma unroll
                    for (int pl = 0; pl < 9; ++pl) s.p[pl / 3][pl % 3][r][cc] = rp[k][pl];
                }
                r += 6; cc += 4;
                if (cc >= W1) { cc -= W1; ++r; }
            }
        }
    }
    if (values && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
        finish_sums(partial, (int)(gridDim.x * gridDim.y * gridDim.z), inv_n_d, l1w, sw, values, red);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        __syncthreads();
        for (int i = tid; i < H1 * (TW / G); i += THREADS) {
            const int r = i % H1, cc = (i / H1) * G;                       // rows fastest across lanes
            float v[3][G + 10];
#pragma unroll
            for (int k = 0; k < 3; ++k) load14(&s.p[ch][k][r][cc], v[k]);
            f4v o0, o1, o2;
#pragma unroll
            for (int o = 0; o < G; ++o) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) { const float w = g[t]; a0 += w * v[0][o + t]; a1 += w * v[1][o + t]; a2 += w * v[2][o + t]; }
                o0[o] = a0; o1[o] = a1; o2[o] = a2;
            }
            *reinterpret_cast<f4v*>(&s.q[0][r][cc]) = o0; *reinterpret_cast<f4v*>(&s.q[1][r][cc]) = o1; *reinterpret_cast<f4v*>(&s.q[2][r][cc]) = o2;
        }
        __syncthreads();
        {
            float v[3][GV + 10];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int t = 0; t < GV + 10; ++t) v[k][t] = s.q[k][r0 + t][c];
#pragma unroll
            for (int o = 0; o < GV; ++o) {
                float cm = 0.f, c11 = 0.f, c12 = 0.f;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) { const float w = g[t]; cm += w * v[0][o + t]; c11 += w * v[1][o + t]; c12 += w * v[2][o + t]; }
                const float a = px[o][ch], b = py[o][ch];
                const float d = a - b;
                const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                out[o][ch] = inv_n * (l1w * sgn - sw * (cm + 2.f * a * c11 + b * c12));
            }
        }
    }
#pragma unroll
    for (int o = 0; o < GV; ++o) {
        const int gy = y0 + r0 + o, gx = x0 + c;
        if (gy < H && gx < W) {
            float* gp = grad + img + ((int64_t)gy * W + gx) * 3;
            gp[0] = out[o][0]; gp[1] = out[o][1]; gp[2] = out[o][2];
        }
    }
}

// value only (no gradient asked for): the sums by a kernel of their own
__global__ __launch_bounds__(THREADS) void loss_finish_kernel(const float* __restrict__ partial, int n_blocks, double inv_n, float l1w, float sw,
                                                              float* __restrict__ values, double scale, float* __restrict__ total_out) {
    __shared__ double red[2][THREADS / 64];
    finish_sums(partial, n_blocks, inv_n, l1w, sw, values, red, scale, total_out);
}

}  // namespace

extern thread_local char gsplat_err_buf[512];
#define g_loss_err gsplat_err_buf

extern "C" {

// scratch = [per block of loss_stats_kernel: 2 partial sums | the three partial-derivative maps of every channel: batch x 3 x 3 x H x W floats]
static int64_t loss_blocks(int64_t batch, int32_t H, int32_t W) { return batch * ((W + TW - 1) / TW) * ((H + TH - 1) / TH); }
static int64_t loss_sums_bytes(int64_t batch, int32_t H, int32_t W) { return 256 * ((loss_blocks(batch, H, W) * 2 * (int64_t)sizeof(float) + 255) / 256); }
int64_t gsplat_loss_scratch_bytes(int64_t batch, int32_t H, int32_t W, int32_t with_grad) {
    if (batch <= 0 || H <= 0 || W <= 0) return -1;
    return loss_sums_bytes(batch, H, W) + (with_grad ? batch * 9 * (int64_t)H * W * (int64_t)sizeof(float) : 0);
}

int gsplat_loss(const float* pred, const float* target, int64_t batch, int32_t H, int32_t W, float lambda_l1, float lambda_ssim,
                float* values, float* grad_pred, void* scratch, void* stream_) {
    if (!pred || !target || !values || !scratch || batch <= 0 || H <= 0 || W <= 0 || batch > 65535 || loss_blocks(batch, H, W) > 0x3FFFFFFF) {
        snprintf(g_loss_err, sizeof(g_loss_err), "gsplat_loss: bad argument");
        return GSPLAT_ERR_BAD_ARG;
    }
    hipStream_t st = (hipStream_t)stream_;
    float* partial = (float*)scratch;
    const double n = (double)batch * H * W * 3;
    const dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, (unsigned)batch);
    float* maps = grad_pred ? (float*)((char*)scratch + loss_sums_bytes(batch, H, W)) : nullptr;
    hipLaunchKernelGGL(loss_stats_kernel, grid, dim3(THREADS), 0, st, pred, target, (int)H, (int)W, partial, maps);
    if (grad_pred)
        hipLaunchKernelGGL(loss_grad_kernel, grid, dim3(THREADS), 0, st, pred, target, (int)H, (int)W, lambda_l1, lambda_ssim, (float)(1.0 / n),
                           1.0 / n, (const float*)maps, (const float*)partial, values, grad_pred, (const float*)nullptr);
    else
        hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(THREADS), 0, st, (const float*)partial, (int)loss_blocks(batch, H, W), 1.0 / n,
                           lambda_l1, lambda_ssim, values, 1.0, (float*)nullptr);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_loss_err, sizeof(g_loss_err), "gsplat_loss launch: %s", hipGetErrorString(e));
        return GSPLAT_ERR_HIP;
    }
    return GSPLAT_OK;
}

/* The same loss in two calls, for a caller whose graph supplies d L / d total only later (an autograd node): the forward leaves the
 * partial-derivative maps in scratch, the backward multiplies the upstream scalar and `scale` into the gradient it writes -- no pass
 * over the gradient afterwards, no gradient computed for a value nobody differentiates.                                          */
int gsplat_loss_forward(const float* pred, const float* target, int64_t batch, int32_t H, int32_t W, float lambda_l1, float lambda_ssim,
                        float scale, float* values, float* total, void* scratch, int32_t keep_maps, void* stream_) {
    if (!pred || !target || !values || !scratch || batch <= 0 || H <= 0 || W <= 0 || batch > 65535 || loss_blocks(batch, H, W) > 0x3FFFFFFF) {
        snprintf(g_loss_err, sizeof(g_loss_err), "gsplat_loss_forward: bad argument");
        return GSPLAT_ERR_BAD_ARG;
    }
    hipStream_t st = (hipStream_t)stream_;
    float* partial = (float*)scratch;
    const double n = (double)batch * H * W * 3;
    const dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, (unsigned)batch);
    float* maps = keep_maps ? (float*)((char*)scratch + loss_sums_bytes(batch, H, W)) : nullptr;
    hipLaunchKernelGGL(loss_stats_kernel, grid, dim3(THREADS), 0, st, pred, target, (int)H, (int)W, partial, maps);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(THREADS), 0, st, (const float*)partial, (int)loss_blocks(batch, H, W), 1.0 / n,
                       lambda_l1, lambda_ssim, values, (double)scale, total);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_loss_err, sizeof(g_loss_err), "gsplat_loss_forward launch: %s", hipGetErrorString(e));
        return GSPLAT_ERR_HIP;
    }
    return GSPLAT_OK;
}

int gsplat_loss_backward(const float* pred, const float* target, int64_t batch, int32_t H, int32_t W, float lambda_l1, float lambda_ssim,
                         float scale, const float* upstream, float* grad_pred, void* scratch, void* stream_) {
    if (!pred || !target || !grad_pred || !scratch || batch <= 0 || H <= 0 || W <= 0 || batch > 65535 || loss_blocks(batch, H, W) > 0x3FFFFFFF) {
        snprintf(g_loss_err, sizeof(g_loss_err), "gsplat_loss_backward: bad argument");
        return GSPLAT_ERR_BAD_ARG;
    }
    const double n = (double)batch * H * W * 3;
    const dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, (unsigned)batch);
    const float* maps = (const float*)((char*)scratch + loss_sums_bytes(batch, H, W));
    hipLaunchKernelGGL(loss_grad_kernel, grid, dim3(THREADS), 0, (hipStream_t)stream_, pred, target, (int)H, (int)W, lambda_l1, lambda_ssim,
                       (float)((double)scale / n), 1.0 / n, maps, (const float*)scratch, (float*)nullptr, grad_pred, upstream);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_loss_err, sizeof(g_loss_err), "gsplat_loss_backward launch: %s", hipGetErrorString(e));
        return GSPLAT_ERR_HIP;
    }
    return GSPLAT_OK;
}

}  // extern "C"
