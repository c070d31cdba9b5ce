// Microbenchmark: cycles per wave64 VALU instruction per SIMD on gfx950, at 1/2/4/8 waves per SIMD.
// hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2000, UNROLL = 16;

template <int OP>
__global__ __launch_bounds__(64) void k(float* out, float seed) {
    float a[UNROLL];
    v2f p[UNROLL / 2];
    for (int i = 0; i < UNROLL; ++i) a[i] = seed + i + threadIdx.x;
    for (int i = 0; i < UNROLL / 2; ++i) p[i] = v2f{a[2 * i], a[2 * i + 1]};
    const float m = seed * 0.999f, c = seed * 0.001f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 3) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(m), "v"(c) : "vcc");
            if (OP == 4) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 7) asm volatile("v_add_f32 %0, %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) % UNROLL]));
        }
        if (OP == 5) {
#pragma unroll
            for (int i = 0; i < UNROLL / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(v2f{m, m}), "v"(v2f{c, c}));
        }
    }
    float s = 0;
    for (int i = 0; i < UNROLL; ++i) s += a[i];
    for (int i = 0; i < UNROLL / 2; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int per_iter) {
    float* out;
    hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * 4 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_simd = (double)wps * ITERS * per_iter;
        printf("%-22s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, wps, ms,
               ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    }
    hipFree(out);
}

int main() {
    run<0>("v_fma_f32", UNROLL);
    run<1>("v_mul_f32", UNROLL);
    run<5>("v_pk_fma_f32", UNROLL / 2);
    run<2>("v_exp_f32", UNROLL);
    run<6>("v_rcp_f32", UNROLL);
    run<3>("v_cmp+v_cndmask (2)", 2 * UNROLL);
    run<4>("v_min_f32", UNROLL);
    run<7>("v_add_f32 dpp", UNROLL);
    return 0;
}
