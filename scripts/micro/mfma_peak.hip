// Sustained fp32 MFMA rate on this GPU: the practical ceiling for mlp_split_kernel's roofline.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += acc[c][q];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC>
static void run(int waves_per_simd, int iters) {
    float *out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;       // 256-thread blocks: 4 waves = one per SIMD
    for (int rep = 0; rep < 2; ++rep) mfma_loop<NACC><<<blocks, 256>>>(out, iters, 1.0f, 1e-6f);
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.f; const int R = 5;
    for (int rep = 0; rep < R; ++rep) {
        hipEventRecord(e0);
        mfma_loop<NACC><<<blocks, 256>>>(out, iters, 1.0f, 1e-6f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    const double flop = (double)blocks * 4 * iters * 16 * NACC * (2.0 * 32 * 32 * 2);
    printf("NACC=%d waves/SIMD=%d iters=%d  best %.3f ms  %.1f TFLOP/s (avg %.1f)  implied clock %.3f GHz at 256 flop/clk/CU\n",
           NACC, waves_per_simd, iters, best, flop / best * 1e-9, flop / (sum / R) * 1e-9, flop / best * 1e-9 / (65536e-3));
    hipFree(out);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    run<1>(1, iters); run<1>(2, iters); run<1>(4, iters); run<1>(8, iters / 2);
    run<2>(1, iters); run<2>(4, iters / 2);
    run<4>(1, iters / 2); run<4>(2, iters / 2);
    // long sustained run (about 0.5 s) to see the clock settle
    run<1>(4, iters * 20);
    return 0;
}
