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
// Sums go to 64 shards (same-address atomics serialise).
#include <cstdio>

#include <hip/hip_runtime.h>

#include "../../include/gsplat_mi355x.h"

namespace {

constexpr int TW = 32, TH = 16, R = 5, TAPS = 11;
constexpr int W1 = TW + 2 * R, H1 = TH + 2 * R;      // region with one halo   42 x 26
constexpr int THREADS = 256, SHARDS = 64;
constexpr int G = 4;                                 // outputs per thread along a row in the horizontal passes
constexpr int GV = 2;                                // rows per thread in the vertical passes
static_assert(TW % G == 0 && TH % GV == 0 && (TH / GV) * TW == THREADS, "tile shape vs register blocking");

// LDS access patterns.  Horizontal passes: lanes run over ROWS (row index fastest), every lane reads G + 10 consecutive words of
// its own row and writes G words; with odd row pitches consecutive rows start on different banks.  Vertical passes: lanes run
// over the 32 columns of a row group, two row groups per wave.
constexpr int XP = W1 + 1;       // 43: pitch of the input planes / the map planes
constexpr int HP = TW + 1;       // 33: pitch of the horizontally filtered planes

__device__ __forceinline__ void gauss_taps(float g[TAPS]) {
    // losses.py:131-155: exp(-(i - 5)^2 / (2 * 1.5^2)), normalised
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TAPS; ++i) { const float d = (float)(i - R); g[i] = expf(-(d * d) / 4.5f); s += g[i]; }
    const float inv = 1.0f / s;
#pragma unroll
    for (int i = 0; i < TAPS; ++i) g[i] *= inv;
}

// region element i (row-major over H1 x W1) of channel ch of a [H][W][3] image, zero outside the image
struct RegionFetch {
    static constexpr int SLOTS = (H1 * W1 + THREADS - 1) / THREADS;
};

struct StatsLds {
    float x[H1][XP], y[H1][XP];
    float h[5][H1][HP];          // horizontally filtered x, y, xx, yy, xy
};

// maps: [image][channel][3][H][W] (planar: the halo reads of loss_grad_kernel are rows of consecutive floats)
__global__ __launch_bounds__(THREADS) void loss_stats_kernel(const float* __restrict__ pred, const float* __restrict__ target, int H, int W,
                                                             float* __restrict__ sums, float* __restrict__ maps) {
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
    constexpr int SLOTS = RegionFetch::SLOTS;
    float ra[SLOTS], rb[SLOTS];
    // channel ch + 1 is fetched into registers while channel ch is being processed, so only the first fetch is exposed
    auto fetch = [&](int ch) {
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            const int i = tid + k * THREADS;
            float a = 0.f, b = 0.f;
            if (i < H1 * W1) {
                const int r = i / W1, c = i - r * W1;
                const int gy = y0 - R + r, gx = x0 - R + c;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) {      // zero outside the image = the reference's zero padding
                    const int64_t o = img + ((int64_t)gy * W + gx) * 3 + ch;
                    a = pred[o]; b = target[o];
                }
            }
            ra[k] = a; rb[k] = b;
        }
    };
    fetch(0);
    for (int ch = 0; ch < 3; ++ch) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            const int i = tid + k * THREADS;
            if (i < H1 * W1) { const int r = i / W1, c = i - r * W1; s.x[r][c] = ra[k]; s.y[r][c] = rb[k]; }
        }
        __syncthreads();
        if (ch < 2) fetch(ch + 1);
        // horizontal pass of the five products: G outputs per thread
        for (int i = tid; i < H1 * (TW / G); i += THREADS) {
            const int r = i % H1, c = (i / H1) * G;                        // rows fastest across lanes
            float a[G + 10], b[G + 10];
#pragma unroll
            for (int t = 0; t < G + 10; ++t) { a[t] = s.x[r][c + t]; b[t] = s.y[r][c + t]; }
#pragma unroll
            for (int o = 0; o < G; ++o) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const float w = g[t], u = a[o + t], v = b[o + t];
                    a0 += w * u; a1 += w * v; a2 += w * u * u; a3 += w * v * v; a4 += w * u * v;
                }
                s.h[0][r][c + o] = a0; s.h[1][r][c + o] = a1; s.h[2][r][c + o] = a2; s.h[3][r][c + o] = a3; s.h[4][r][c + o] = a4;
            }
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
                    const float ib = 1.0f / (B1 * B2);
                    const float S = A1 * A2 * ib;
                    ssim_acc += S;
                    l1_acc += fabsf(s.x[r0 + o + R][c + R] - s.y[r0 + o + R][c + R]);
                    if (maps) {
                        float* m = maps + ((int64_t)blockIdx.z * 3 + ch) * 3 * plane + (int64_t)gy * W + gx;
                        m[0] = 2.f * mu2 * (A2 - A1) * ib - 2.f * mu1 * S / B1 + 2.f * mu1 * S / B2;
                        m[plane] = -S / B2;
                        m[2 * plane] = 2.f * A1 * ib;
                    }
                }
            }
        }
    }
    // block reduction of the two sums, one atomic pair per block into a shard
    for (int sft = 32; sft > 0; sft >>= 1) { l1_acc += __shfl_xor(l1_acc, sft); ssim_acc += __shfl_xor(ssim_acc, sft); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = l1_acc; red[1][tid >> 6] = ssim_acc; }
    __syncthreads();
    if (tid == 0) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < THREADS / 64; ++k) { a += red[0][k]; b += red[1][k]; }
        const int shard = (blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y) % SHARDS;
        atomicAdd(&sums[shard * 2 + 0], a);
        atomicAdd(&sums[shard * 2 + 1], b);
    }
}

struct GradLds {
    float p[3][H1][XP];          // dS/dmu1, dS/dE11, dS/dE12 with the halo (zero outside the image)
    float q[3][H1][HP];          // horizontally filtered
};

// d(sum S)/dx(q) = (w * dS/dmu1)(q) + 2 x(q) (w * dS/dE11)(q) + y(q) (w * dS/dE12)(q); with the L1 term and the weights -> grad
__global__ __launch_bounds__(THREADS) void loss_grad_kernel(const float* __restrict__ pred, const float* __restrict__ target, int H, int W,
                                                            float l1w, float sw, float inv_n, const float* __restrict__ maps,
                                                            float* __restrict__ grad) {
    __shared__ GradLds s;
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int64_t img = (int64_t)blockIdx.z * H * W * 3;
    const int64_t plane = (int64_t)H * W;
    float g[TAPS];
    gauss_taps(g);
    constexpr int SLOTS = RegionFetch::SLOTS;
    float rp[3][SLOTS];
    auto fetch = [&](int ch) {
        const float* m = maps + ((int64_t)blockIdx.z * 3 + ch) * 3 * plane;
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            const int i = tid + k * THREADS;
            float a = 0.f, b = 0.f, c_ = 0.f;
            if (i < H1 * W1) {
                const int r = i / W1, c = i - r * W1;
                const int gy = y0 - R + r, gx = x0 - R + c;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                    const int64_t o = (int64_t)gy * W + gx;
                    a = m[o]; b = m[plane + o]; c_ = m[2 * plane + o];
                }
            }
            rp[0][k] = a; rp[1][k] = b; rp[2][k] = c_;
        }
    };
    fetch(0);
    // this thread's GV pixels (rows r0, r0 + 1 of column c): x, y of the three channels, and the gradient as it comes
    const int rg = tid / TW, c = tid - rg * TW, r0 = rg * GV;
    float px[GV][3], py[GV][3], out[GV][3];
#pragma unroll
    for (int o = 0; o < GV; ++o) {
        const int gy = y0 + r0 + o, gx = x0 + c;
        const bool in = gy < H && gx < W;
        const int64_t a = img + ((int64_t)gy * W + gx) * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) { px[o][ch] = in ? pred[a + ch] : 0.f; py[o][ch] = in ? target[a + ch] : 0.f; out[o][ch] = 0.f; }
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            const int i = tid + k * THREADS;
            if (i < H1 * W1) { const int r = i / W1, cc = i - r * W1; s.p[0][r][cc] = rp[0][k]; s.p[1][r][cc] = rp[1][k]; s.p[2][r][cc] = rp[2][k]; }
        }
        __syncthreads();
        if (ch < 2) fetch(ch + 1);
        for (int i = tid; i < H1 * (TW / G); i += THREADS) {
            const int r = i % H1, cc = (i / H1) * G;                       // rows fastest across lanes
            float v[3][G + 10];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int t = 0; t < G + 10; ++t) v[k][t] = s.p[k][r][cc + t];
#pragma unroll
            for (int o = 0; o < G; ++o) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) { const float w = g[t]; a0 += w * v[0][o + t]; a1 += w * v[1][o + t]; a2 += w * v[2][o + t]; }
                s.q[0][r][cc + o] = a0; s.q[1][r][cc + o] = a1; s.q[2][r][cc + o] = a2;
            }
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

__global__ void loss_zero_kernel(float* sums) {
    if (threadIdx.x < SHARDS * 2) sums[threadIdx.x] = 0.f;
}

__global__ void loss_finish_kernel(const float* __restrict__ sums, double inv_n, float l1w, float sw, float* __restrict__ values) {
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < SHARDS; ++k) { a += sums[2 * k]; b += sums[2 * k + 1]; }
        const double l1 = a * inv_n, sl = 1.0 - b * inv_n;
        values[0] = (float)l1; values[1] = (float)sl; values[2] = (float)(l1w * l1 + sw * sl);
    }
}

}  // namespace

extern thread_local char gsplat_err_buf[512];
#define g_loss_err gsplat_err_buf

extern "C" {

// scratch = [64 shards x 2 sums | the three partial-derivative maps of every channel: batch x 3 x 3 x H x W floats]
static int64_t loss_sums_bytes() { return 256 * ((SHARDS * 2 * (int64_t)sizeof(float) + 255) / 256); }
int64_t gsplat_loss_scratch_bytes(int64_t batch, int32_t H, int32_t W, int32_t with_grad) {
    if (batch <= 0 || H <= 0 || W <= 0) return -1;
    return loss_sums_bytes() + (with_grad ? batch * 9 * (int64_t)H * W * (int64_t)sizeof(float) : 0);
}

int gsplat_loss(const float* pred, const float* target, int64_t batch, int32_t H, int32_t W, float lambda_l1, float lambda_ssim,
                float* values, float* grad_pred, void* scratch, void* stream_) {
    if (!pred || !target || !values || !scratch || batch <= 0 || H <= 0 || W <= 0 || batch > 65535) {
        snprintf(g_loss_err, sizeof(g_loss_err), "gsplat_loss: bad argument");
        return GSPLAT_ERR_BAD_ARG;
    }
    hipStream_t st = (hipStream_t)stream_;
    float* sums = (float*)scratch;
    const double n = (double)batch * H * W * 3;
    hipLaunchKernelGGL(loss_zero_kernel, dim3(1), dim3(SHARDS * 2), 0, st, sums);
    const dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, (unsigned)batch);
    float* maps = grad_pred ? (float*)((char*)scratch + loss_sums_bytes()) : nullptr;
    hipLaunchKernelGGL(loss_stats_kernel, grid, dim3(THREADS), 0, st, pred, target, (int)H, (int)W, sums, maps);
    if (grad_pred)
        hipLaunchKernelGGL(loss_grad_kernel, grid, dim3(THREADS), 0, st, pred, target, (int)H, (int)W, lambda_l1, lambda_ssim, (float)(1.0 / n),
                           (const float*)maps, grad_pred);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, st, sums, 1.0 / n, lambda_l1, lambda_ssim, values);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_loss_err, sizeof(g_loss_err), "gsplat_loss launch: %s", hipGetErrorString(e));
        return GSPLAT_ERR_HIP;
    }
    return GSPLAT_OK;
}

}  // extern "C"
