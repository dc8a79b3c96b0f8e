// One wave per SIMD, v_mfma_f32_32x32x16_f16: what does an MFMA cost as a function of (a) how many accumulators rotate (the distance
// between two MFMAs on the same accumulator), (b) whether the A operand lives in the accumulator half of the register file ("a"),
// (c) how many independent v_fma_f32 sit behind each MFMA, (d) whether the accumulators are arch VGPRs or AGPRs.
// Build + run (GPU box): hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_chain_probe scripts/micro/mfma_chain_probe.hip && /tmp/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool A_AGPR, int K, bool C_AGPR>
__global__ __launch_bounds__(256, 1) void bench(float *out, unsigned long long *cyc, int rep) {
    f16x8 a[4], b;
    for (int u = 0; u < 4; ++u)
        for (int i = 0; i < 8; ++i) a[u][i] = (_Float16)(threadIdx.x * 0.001f + i + u);
    for (int i = 0; i < 8; ++i) b[i] = (_Float16)(i * 0.5f);
    f32x16 c[6] = {};
    float f[12];
    for (int i = 0; i < 12; ++i) f[i] = threadIdx.x * 1e-3f + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            if (C_AGPR) {
                if (A_AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[u % NACC]) : "a"(a[u & 3]), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[u % NACC]) : "v"(a[u & 3]), "v"(b));
            } else {
                if (A_AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[u % NACC]) : "a"(a[u & 3]), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[u % NACC]) : "v"(a[u & 3]), "v"(b));
            }
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[k]) : "v"(f[11]));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += f[i];
    for (int u = 0; u < 6; ++u) for (int i = 0; i < 16; ++i) s += c[u][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, bool A_AGPR, int K, bool C_AGPR>
void run(float *out, unsigned long long *cyc) {
    const int rep = 2000, blocks = 256;
    bench<NACC, A_AGPR, K, C_AGPR><<<blocks, 256>>>(out, cyc, rep);
    bench<NACC, A_AGPR, K, C_AGPR><<<blocks, 256>>>(out, cyc, rep);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks; ++i) m += (double)h[i];
    m /= blocks;
    printf("accumulators %d (%s)  A in %s  %2d x v_fma_f32 per MFMA: %6.1f ticks per MFMA\n", NACC, C_AGPR ? "AGPR" : "VGPR", A_AGPR ? "AGPR" : "VGPR", K, m / (rep * 12.0));
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<1, false, 0, false>(out, cyc); run<2, false, 0, false>(out, cyc); run<3, false, 0, false>(out, cyc); run<4, false, 0, false>(out, cyc); run<6, false, 0, false>(out, cyc);
    run<1, false, 4, false>(out, cyc); run<2, false, 4, false>(out, cyc); run<3, false, 4, false>(out, cyc); run<4, false, 4, false>(out, cyc); run<6, false, 4, false>(out, cyc);
    run<2, false, 6, false>(out, cyc); run<3, false, 6, false>(out, cyc); run<4, false, 6, false>(out, cyc); run<6, false, 6, false>(out, cyc);
    run<3, true, 0, false>(out, cyc); run<3, true, 4, false>(out, cyc); run<3, true, 6, false>(out, cyc); run<6, true, 6, false>(out, cyc);
    run<3, false, 4, true>(out, cyc); run<3, false, 6, true>(out, cyc); run<3, true, 6, true>(out, cyc); run<6, false, 6, true>(out, cyc);
    return 0;
}
