// Microbenchmark: LDS cycles per wave64 DS instruction on gfx950 for the access shapes the binning and raster kernels use:
// plain stores, integer atomics without / with return, float atomics; addresses all distinct, 8-way duplicated (8 lanes per
// address: what a per-group accumulate does) or random.
// hipcc -O3 --offload-arch=gfx950 -o lds_atomic_rate lds_atomic_rate.hip && ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 4000;

template <int OP, int PATTERN>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed) {
    __shared__ unsigned s[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0u;
    __syncthreads();
    const int lane = threadIdx.x;
    unsigned acc = 0u, r = seed * 2654435761u + lane * 40503u;
    for (int it = 0; it < ITERS; ++it) {
        r = r * 1664525u + 1013904223u;
        unsigned idx = PATTERN == 0 ? (unsigned)(lane * 9 + (it & 7)) & 4095u                       // distinct, stride 9 words
                     : PATTERN == 1 ? (unsigned)((lane >> 3) * 9 + (it & 63)) & 4095u             // 8 lanes per address
                                    : (r >> 20) & 4095u;                                            // random
        if (OP == 0) s[idx] = r;
        if (OP == 1) atomicAdd(&s[idx], 1u);
        if (OP == 2) acc += atomicAdd(&s[idx], 1u);
        if (OP == 3) atomicAdd(reinterpret_cast<float*>(&s[idx]), 1.0f);
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = acc + s[lane];
}

template <int OP, int PATTERN>
void run(const char* name) {
    unsigned* out;
    hipMalloc(&out, 1 << 24);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8;                      // 8 blocks of 4 waves per CU: 8 waves per SIMD
    hipLaunchKernelGGL((k<OP, PATTERN>), dim3(blocks), dim3(256), 0, 0, out, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<OP, PATTERN>), dim3(blocks), dim3(256), 0, 0, out, 2u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double wave_instr_per_cu = (double)blocks / 256 * 4 * ITERS;      // DS wave-instructions issued per CU
    printf("%-34s %8.3f ms   %7.1f ns per wave-instruction per CU  (~%.0f cycles at 2.1 GHz)\n", name, ms, ms * 1e6 / wave_instr_per_cu,
           ms * 1e6 / wave_instr_per_cu * 2.1);
    hipFree(out);
}

int main() {
    run<0, 0>("ds_write_b32        distinct");
    run<1, 0>("ds_add_u32          distinct");
    run<2, 0>("ds_add_rtn_u32      distinct");
    run<3, 0>("ds_add_f32          distinct");
    run<0, 1>("ds_write_b32        8 lanes/address");
    run<1, 1>("ds_add_u32          8 lanes/address");
    run<2, 1>("ds_add_rtn_u32      8 lanes/address");
    run<3, 1>("ds_add_f32          8 lanes/address");
    run<0, 2>("ds_write_b32        random");
    run<1, 2>("ds_add_u32          random");
    run<2, 2>("ds_add_rtn_u32      random");
    run<3, 2>("ds_add_f32          random");
    return 0;
}
