// gsplat_loss.hip -- fused L1 + SSIM training loss (value AND gradient w.r.t. the prediction in one pass).
//
// SURVEY.md §8(f) "next" row 1: reference gaussian_splatting/losses.py:27-185 (l1_loss, ssim_loss, _ssim_single_channel,
// _create_gaussian_window, compute_loss).  The reference runs 15 conv2d calls on permuted copies per view and autograd
// replays them; here one kernel reads pred/target once and writes the gradient once.
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
// One workgroup per 16 x 16 output tile and image (40 KB of LDS: four workgroups per CU; 32 x 16 tiles needed 62 KB and
// ran at two per CU, latency-bound); the three channels are processed one after the other through the
// same LDS buffers: inputs with a 10-pixel halo, five horizontally filtered planes, the three partial-derivative maps
// with a 5-pixel halo, their horizontally filtered planes.  Sums go to 64 shards (same-address atomics serialise).
#include <cstdio>

#include <hip/hip_runtime.h>

#include "../../include/gsplat_mi355x.h"

namespace {

constexpr int TW = 16, TH = 16, R = 5, TAPS = 11;
constexpr int W0 = TW + 4 * R, H0 = TH + 4 * R;      // input region   36 x 36
constexpr int W1 = TW + 2 * R, H1 = TH + 2 * R;      // map region     26 x 26
constexpr int G2 = (W1 % 3 == 0) ? 3 : 2;            // outputs per thread in the first horizontal pass
static_assert(W1 % G2 == 0 && TW % 4 == 0 && TH % 2 == 0, "tile shape vs register blocking");
constexpr int THREADS = 256, SHARDS = 64;

constexpr int W0P = W0 + 5;      // odd row pitch of the input planes (41): rows map to distinct banks; also read slack

// LDS access patterns.  Horizontal passes: lanes run over ROWS (row index fastest), every lane reads G + 10 consecutive
// words of its own row and writes G words; with odd row pitches (41, 29) consecutive rows start on different banks, so both
// are conflict-free.  Vertical passes: lanes run over columns (26 or 16 per row group); the pitches (30, 25) put the next
// row group on the following banks (3 * 30 = 26 mod 32: exact for the 26-column pass; 2 * 25 = 18: 2 lanes overlap).  With the
// natural mapping and pitches (26, 16, column-fastest everywhere) 60 % of all LDS cycles were bank conflicts.
constexpr int HP = 30, QP = 25, PP = 29;
static_assert(HP >= W1 && QP >= TW && PP >= W1 + 2 && 3 * H1 * QP <= 5 * (H0 + 2) * HP, "plane pitches");

struct LossLds {
    float x[H0][W0P], y[H0][W0P];
    float h[5][H0 + 2][HP];      // horizontally filtered x, y, xx, yy, xy (+2 slack rows); later: q[3][H1][QP]
    float p[3][H1][PP];          // dS/dmu1, dS/dE11, dS/dE12 (zero outside the image; slack columns)
};
// Every pass is register-blocked: a thread produces G consecutive outputs along the filter direction from G + 10 inputs
// held in registers, so an output costs (G + 10) / G LDS reads per plane instead of 11.

__device__ __forceinline__ void gauss_taps(float g[TAPS]) {
    // losses.py:131-155: exp(-(i - 5)^2 / (2 * 1.5^2)), normalised
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TAPS; ++i) { const float d = (float)(i - R); g[i] = expf(-(d * d) / 4.5f); s += g[i]; }
    const float inv = 1.0f / s;
#pragma unroll
    for (int i = 0; i < TAPS; ++i) g[i] *= inv;
}

__global__ __launch_bounds__(THREADS) void loss_kernel(const float* __restrict__ pred, const float* __restrict__ target, int H, int W,
                                                       float l1w, float sw, float inv_n, float* __restrict__ sums,
                                                       float* __restrict__ grad) {
    __shared__ LossLds s;
    __shared__ float red[2][THREADS / 64];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int64_t img = (int64_t)blockIdx.z * H * W * 3;
    float g[TAPS];
    gauss_taps(g);
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float l1_acc = 0.f, ssim_acc = 0.f;
    // inputs with a 2R halo, zero outside the image (= the reference's zero padding); channel ch + 1 is fetched into
    // registers while channel ch is being processed, so only the first fetch is exposed
    constexpr int SLOTS = (H0 * W0 + THREADS - 1) / THREADS;
    float ra[SLOTS], rb[SLOTS];
    auto fetch = [&](int ch) {
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            const int i = tid + k * THREADS;
            float a = 0.f, b = 0.f;
            if (i < H0 * W0) {
                const int r = i / W0, c = i - r * W0;
                const int gy = y0 - 2 * R + r, gx = x0 - 2 * R + c;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
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
            if (i < H0 * W0) { const int r = i / W0, c = i - r * W0; s.x[r][c] = ra[k]; s.y[r][c] = rb[k]; }
        }
        __syncthreads();
        if (ch < 2) fetch(ch + 1);
        // 2. horizontal pass of the five products: G2 outputs per thread
        for (int i = tid; i < H0 * (W1 / G2); i += THREADS) {
            const int r = i % H0, c = (i / H0) * G2;                       // rows fastest across lanes
            float a[G2 + 10], b[G2 + 10];
#pragma unroll
            for (int t = 0; t < G2 + 10; ++t) { a[t] = s.x[r][c + t]; b[t] = s.y[r][c + t]; }
#pragma unroll
            for (int o = 0; o < G2; ++o) {
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
        // 3. vertical pass -> SSIM map and its partial derivatives on the tile + R halo: 3 rows per thread
        constexpr int G3 = 3, NG3 = (H1 + G3 - 1) / G3;
        for (int i = tid; i < NG3 * W1; i += THREADS) {
            const int rg = i / W1, c = i - rg * W1, r0 = rg * G3;
            float v[5][G3 + 10];
#pragma unroll
            for (int k = 0; k < 5; ++k)
#pragma unroll
                for (int t = 0; t < G3 + 10; ++t) v[k][t] = s.h[k][r0 + t][c];
#pragma unroll
            for (int o = 0; o < G3; ++o) {
                const int r = r0 + o;
                if (r >= H1) break;
                const int gy = y0 - R + r, gx = x0 - R + c;
                float dmu = 0.f, d11 = 0.f, d12 = 0.f;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
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
                    dmu = 2.f * mu2 * (A2 - A1) * ib - 2.f * mu1 * S / B1 + 2.f * mu1 * S / B2;
                    d11 = -S / B2;
                    d12 = 2.f * A1 * ib;
                    if (r >= R && r < R + TH && c >= R && c < R + TW) ssim_acc += S;
                }
                s.p[0][r][c] = dmu; s.p[1][r][c] = d11; s.p[2][r][c] = d12;
            }
        }
        __syncthreads();
        // 4. horizontal pass of the three derivative maps (into the h buffer, which is free now): 4 outputs per thread
        float(*q)[H1][QP] = reinterpret_cast<float(*)[H1][QP]>(&s.h[0][0][0]);
        for (int i = tid; i < H1 * (TW / 4); i += THREADS) {
            const int r = i % H1, c = (i / H1) * 4;                        // rows fastest across lanes
            float v[3][14];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int t = 0; t < 14; ++t) v[k][t] = s.p[k][r][c + t];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) { const float w = g[t]; a0 += w * v[0][o + t]; a1 += w * v[1][o + t]; a2 += w * v[2][o + t]; }
                q[0][r][c + o] = a0; q[1][r][c + o] = a1; q[2][r][c + o] = a2;
            }
        }
        __syncthreads();
        // 5. vertical pass + L1 term -> gradient of this channel: 2 rows per thread
        for (int i = tid; i < (TH / 2) * TW; i += THREADS) {
            const int rg = i / TW, c = i - rg * TW, r0 = rg * 2;
            float v[3][12];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int t = 0; t < 12; ++t) v[k][t] = q[k][r0 + t][c];
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int r = r0 + o, gy = y0 + r, gx = x0 + c;
                if (gy < H && gx < W) {
                    float cm = 0.f, c11 = 0.f, c12 = 0.f;
#pragma unroll
                    for (int t = 0; t < TAPS; ++t) { const float w = g[t]; cm += w * v[0][o + t]; c11 += w * v[1][o + t]; c12 += w * v[2][o + t]; }
                    const float a = s.x[r + 2 * R][c + 2 * R], b = s.y[r + 2 * R][c + 2 * R];
                    const float d = a - b;
                    l1_acc += fabsf(d);
                    if (grad) {
                        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                        grad[img + ((int64_t)gy * W + gx) * 3 + ch] = inv_n * (l1w * sgn - sw * (cm + 2.f * a * c11 + b * c12));
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

int64_t gsplat_loss_scratch_bytes(void) { return SHARDS * 2 * sizeof(float); }

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
    hipLaunchKernelGGL(loss_kernel, grid, dim3(THREADS), 0, st, pred, target, (int)H, (int)W, lambda_l1, lambda_ssim, (float)(1.0 / n), sums,
                       grad_pred);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, st, sums, 1.0 / n, lambda_l1, lambda_ssim, values);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_loss_err, sizeof(g_loss_err), "gsplat_loss launch: %s", hipGetErrorString(e));
        return GSPLAT_ERR_HIP;
    }
    return GSPLAT_OK;
}

}  // extern "C"
