// Microbenchmark: do v_mfma_f32_16x16x4_f32 instructions ride along with a VALU-bound loop on gfx950?  The raster backward
// loop is ~100 VALU instructions per iteration at 3 waves per SIMD; the question is what 2..5 f32 MFMAs per iteration cost
// there (their pipe is separate), and what the gfx950 lane swaps (v_permlane16/32_swap) cost next to a DPP add.
// hipcc -O3 --offload-arch=gfx950 -o mfma_coexec mfma_coexec.hip && ./mfma_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 2000;

template <int NV, int NM, int NS>
__global__ __launch_bounds__(64) void k(float* out, float seed) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x;
    const float m = seed * 0.999f, c = seed * 0.001f;
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    float keep = 0.f;
    for (int it = 0; it < ITERS; ++it) {
        // the previous iteration's matrix result is consumed here (software pipelined, like the kernel would)
        keep += d[0] + d[1] * c;
        if (NM > 0) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], m, z, 0, 0, 0);
#pragma unroll
            for (int i = 1; i < NM; ++i) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], c, d, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i % 16]) : "v"(m), "v"(c));
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (i & 1) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[2 * (i % 8)]), "+v"(a[2 * (i % 8) + 1]));
            else asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[2 * (i % 8)]), "+v"(a[2 * (i % 8) + 1]));
        }
    }
    float s = keep;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NV, int NM, int NS>
void run(const char* name) {
    float* out;
    hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 3, 4}) {
        const int blocks = 256 * 4 * wps;
        hipLaunchKernelGGL((k<NV, NM, NS>), dim3(blocks), dim3(64), 0, 0, out, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, NM, NS>), dim3(blocks), dim3(64), 0, 0, out, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s waves/SIMD %d: %.3f ms -> %.1f ns per iteration per SIMD\n", name, wps, ms, ms * 1e6 / ((double)wps * ITERS));
    }
    hipFree(out);
}

int main() {
    run<100, 0, 0>("100 fma");
    run<76, 0, 0>("76 fma");
    run<76, 2, 0>("76 fma + 2 mfma");
    run<76, 5, 0>("76 fma + 5 mfma");
    run<68, 2, 8>("68 fma + 2 mfma + 8 lane swaps");
    run<68, 0, 8>("68 fma + 8 lane swaps");
    run<0, 2, 0>("2 mfma");
    run<0, 0, 16>("16 lane swaps");
    return 0;
}
