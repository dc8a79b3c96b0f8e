// How many independent vector-ALU instructions ("fillers") hide behind one MFMA, by MFMA shape and by waves per SIMD?
// Each wave runs REP x { 1 MFMA (4 rotating accumulators), K v_fma_f32 fillers on K independent registers }, timed with s_memtime.
// Build + run (GPU box): hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_fillers scripts/micro/mfma_fillers.hip && /tmp/mfma_fillers
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int K>
__global__ __launch_bounds__(512) void bench(float *out, unsigned long long *cyc, int rep) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x4 c4[4] = {};
    f32x16 c16[4] = {};
    float f[12];
    for (int i = 0; i < 12; ++i) f[i] = threadIdx.x + i;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (SHAPE == 16) c4[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4[u], 0, 0, 0);
            else c16[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c16[u], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) f[k] = __builtin_fmaf(f[k], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += f[i];
    for (int u = 0; u < 4; ++u) { for (int i = 0; i < 4; ++i) s += c4[u][i]; for (int i = 0; i < 16; ++i) s += c16[u][i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int K>
void run(int threads, float *out, unsigned long long *cyc) {
    const int rep = 2000, blocks = 256;
    bench<SHAPE, K><<<blocks, threads>>>(out, cyc, rep);
    bench<SHAPE, K><<<blocks, threads>>>(out, cyc, rep);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks; ++i) m += (double)h[i];
    m /= blocks;
    printf("%s  waves/SIMD %d  fillers/MFMA %2d : %6.1f cycles per MFMA per wave, %6.1f per MFMA per SIMD\n", SHAPE == 16 ? "16x16x32" : "32x32x16",
           threads / 256, K, m / (rep * 4.0), m / (rep * 4.0) / (threads / 256));
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    for (int threads : {256, 512}) {
        run<16, 0>(threads, out, cyc); run<16, 1>(threads, out, cyc); run<16, 2>(threads, out, cyc); run<16, 3>(threads, out, cyc);
        run<16, 4>(threads, out, cyc); run<16, 6>(threads, out, cyc); run<16, 8>(threads, out, cyc);
        run<32, 0>(threads, out, cyc); run<32, 2>(threads, out, cyc); run<32, 4>(threads, out, cyc); run<32, 6>(threads, out, cyc);
        run<32, 8>(threads, out, cyc); run<32, 10>(threads, out, cyc); run<32, 12>(threads, out, cyc);
    }
    return 0;
}
