// How many independent vector-ALU instructions ("fillers") hide behind one MFMA, by MFMA shape, filler kind and waves per SIMD?
// Each wave runs REP x 4 x { 1 MFMA (4 rotating accumulators), K fillers on K independent registers }, instruction order pinned with
// inline assembly.  Reported: wall-clock ns per MFMA per SIMD (hipEvents) and s_memtime ticks per MFMA per wave.
// Peak (2.4 GHz): 32x32x16 = 32 cycles = 13.3 ns per SIMD; 16x16x32 = 16 cycles = 6.7 ns.
// Build + run (GPU box): hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_fillers scripts/micro/mfma_fillers.hip && /tmp/mfma_fillers
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// FILL: 0 v_fma_f32, 1 v_exp_f32, 2 v_pk_fma_f32, 3 ds_read_b128 (LDS), 4 v_cvt_pk_f16_f32
template <int SHAPE, int K, int FILL>
__global__ __launch_bounds__(512) void bench(float *out, unsigned long long *cyc, int rep) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x4 c4[4] = {};
    f32x16 c16[4] = {};
    float f[12];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 p2[12];
    f32x4 l4[12];
    for (int i = 0; i < 12; ++i) { f[i] = threadIdx.x * 1e-3f + i; p2[i] = f32x2{f[i], f[i]}; l4[i] = f32x4{0, 0, 0, 0}; }
    const float *lp = lds + (threadIdx.x & 63) * 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (SHAPE == 16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c4[u]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c16[u]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (FILL == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[k]) : "v"(f[11]));
                else if (FILL == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f[k]));
                else if (FILL == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p2[k]) : "v"(p2[11]));
                else if (FILL == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(l4[k]) : "v"((unsigned)(size_t)lp));
                else asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(f[k]) : "v"(f[11]));
            }
        }
        if (FILL == 3) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += f[i] + p2[i][0] + l4[i][0];
    for (int u = 0; u < 4; ++u) { for (int i = 0; i < 4; ++i) s += c4[u][i]; for (int i = 0; i < 16; ++i) s += c16[u][i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int K, int FILL>
void run(int threads, float *out, unsigned long long *cyc) {
    const int rep = 4000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    bench<SHAPE, K, FILL><<<blocks, threads>>>(out, cyc, rep);
    hipEventRecord(e0);
    bench<SHAPE, K, FILL><<<blocks, threads>>>(out, cyc, rep);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks; ++i) m += (double)h[i];
    m /= blocks;
    const int wps = threads / 256;
    const char *fn[] = {"v_fma_f32", "v_exp_f32", "v_pk_fma_f32", "ds_read_b128", "v_cvt_pk_f16"};
    printf("%s  waves/SIMD %d  %2d x %-13s: %6.2f ns per MFMA per SIMD   (%6.1f ticks per MFMA per wave; tick rate %.2f GHz)\n",
           SHAPE == 16 ? "16x16x32" : "32x32x16", wps, K, fn[FILL], ms * 1e6 / (rep * 4.0 * wps), m / (rep * 4.0), m / (ms * 1e6));
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    for (int threads : {256, 512}) {
        run<16, 0, 0>(threads, out, cyc); run<16, 1, 0>(threads, out, cyc); run<16, 2, 0>(threads, out, cyc); run<16, 3, 0>(threads, out, cyc);
        run<16, 4, 0>(threads, out, cyc); run<16, 6, 0>(threads, out, cyc);
        run<32, 0, 0>(threads, out, cyc); run<32, 2, 0>(threads, out, cyc); run<32, 4, 0>(threads, out, cyc); run<32, 6, 0>(threads, out, cyc);
        run<32, 8, 0>(threads, out, cyc); run<32, 12, 0>(threads, out, cyc);
        run<32, 2, 1>(threads, out, cyc); run<32, 4, 1>(threads, out, cyc);
        run<32, 2, 2>(threads, out, cyc); run<32, 4, 2>(threads, out, cyc);
        run<32, 2, 3>(threads, out, cyc); run<32, 4, 3>(threads, out, cyc);
        run<32, 4, 4>(threads, out, cyc);
        run<16, 2, 1>(threads, out, cyc); run<16, 2, 3>(threads, out, cyc);
    }
    return 0;
}
