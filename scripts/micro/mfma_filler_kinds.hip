// One wave per SIMD, v_mfma_f32_32x32x16_f16 with SIX vector instructions of one kind behind each MFMA: which kinds hide in the MFMA's
// shadow and which do not (literal operands, VOP2 / VOP3 / VOP3P encodings, transcendental, abs modifiers, f16 mix)?
// Build + run (GPU box): hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_filler_kinds scripts/micro/mfma_filler_kinds.hip && /tmp/mfma_filler_kinds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FILL, int K>
__global__ __launch_bounds__(256, 1) void bench(float *out, unsigned long long *cyc, int rep) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 c[3] = {};
    float f[12], g[12];
    for (int i = 0; i < 12; ++i) { f[i] = threadIdx.x * 1e-3f + i; g[i] = 0.5f * i; }
    const float sc = 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[u % 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (FILL == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[k]) : "v"(g[k]));
                if (FILL == 1) asm volatile("v_fmamk_f32 %0, %0, 0x3a000000, %1" : "+v"(f[k]) : "v"(g[k]));
                if (FILL == 2) asm volatile("v_mul_f32_e32 %0, 0x3fb8aa3b, %1" : "=v"(f[k]) : "v"(g[k]));
                if (FILL == 3) asm volatile("v_exp_f32_e64 %0, %1 clamp" : "=v"(f[k]) : "v"(g[k]));
                if (FILL == 4) asm volatile("v_max_f32_e32 %0, 0, %1" : "=v"(f[k]) : "v"(g[k]));
                if (FILL == 5) asm volatile("v_fmac_f32_e32 %0, 0x45067d5f, %1" : "+v"(f[k]) : "v"(g[k]));
                if (FILL == 6) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(f[k]) : "v"(g[k]), "v"(g[(k + 1) % 12]));
                if (FILL == 7) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(f[k]) : "v"(g[k]), "s"(sc));
                if (FILL == 8) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "+v"(f[k]) : "v"(g[k]), "s"(sc), "v"(g[(k + 1) % 12]));
                if (FILL == 9) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(f[k]) : "v"(g[k]), "v"(g[(k + 1) % 12]));
                if (FILL == 10) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(g[k]), "s"(sc));
                if (FILL == 11) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(f[k]) : "v"(g[k]));
                if (FILL == 12) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(f[k]) : "v"(g[k]));
                if (FILL == 13) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(f[k]));
                if (FILL == 14) asm volatile("v_fma_f32 %0, %1, %2, 0" : "=v"(f[k]) : "v"(g[k]), "s"(sc));
                if (FILL == 15) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(f[k]) : "v"(g[k]), "v"(g[(k + 1) % 12]));
                if (FILL == 16) asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(f[k]) : "v"(g[k]), "v"(g[(k + 1) % 12]));
                if (FILL == 17) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(f[k]) : "v"(g[k]), "v"(g[(k + 1) % 12]));
                if (FILL == 18) asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(f[k]) : "v"(g[k]));
                if (FILL == 19) asm volatile("v_max3_f32 %0, %1, 0, 0" : "=v"(f[k]) : "v"(g[k]));
                if (FILL == 20) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f[k]) : "v"(g[k]), "v"(g[(k + 1) % 12]), "v"(g[(k + 2) % 12]));
                if (FILL == 21) asm volatile("v_mov_b32 %0, %1" : "=v"(f[k]) : "v"(g[k]));
                if (FILL == 22) asm volatile("v_fma_mix_f32 %0, %1, %2, %3" : "=v"(f[k]) : "v"(g[k]), "v"(g[(k + 1) % 12]), "v"(g[(k + 2) % 12]));
                if (FILL == 23) asm volatile("v_med3_f32 %0, %1, 0, 1.0" : "=v"(f[k]) : "v"(g[k]));
                if (FILL == 24) asm volatile("v_and_b32 %0, %1, %2" : "=v"(f[k]) : "v"(g[k]), "v"(g[(k + 1) % 12]));
                if (FILL == 25) asm volatile("v_ldexp_f32 %0, %1, 11" : "=v"(f[k]) : "v"(g[k]));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += f[i] + g[i];
    for (int u = 0; u < 3; ++u) for (int i = 0; i < 16; ++i) s += c[u][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int FILL, int K>
void run(const char *name, float *out, unsigned long long *cyc) {
    const int rep = 2000, blocks = 256;
    bench<FILL, K><<<blocks, 256>>>(out, cyc, rep);
    bench<FILL, K><<<blocks, 256>>>(out, cyc, rep);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks; ++i) m += (double)h[i];
    m /= blocks;
    printf("%2d x %-52s: %6.1f ticks per MFMA\n", K, name, m / (rep * 12.0));
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<0, 6>("v_fma_f32 (VOP3)", out, cyc);
    run<1, 6>("v_fmamk_f32 (VOP2 + 32-bit literal)", out, cyc);
    run<2, 6>("v_mul_f32_e32 literal, v", out, cyc);
    run<3, 6>("v_exp_f32_e64 clamp", out, cyc);
    run<3, 4>("v_exp_f32_e64 clamp", out, cyc);
    run<3, 2>("v_exp_f32_e64 clamp", out, cyc);
    run<4, 6>("v_max_f32_e32 0, v", out, cyc);
    run<5, 6>("v_fmac_f32_e32 literal", out, cyc);
    run<6, 6>("v_max3_f32 v, |v|, |v|", out, cyc);
    run<7, 6>("v_fma_mixlo_f16 v, s, 0", out, cyc);
    run<8, 6>("v_fma_mixlo_f16 v(f16), s, v op_sel_hi", out, cyc);
    run<9, 6>("v_fmac_f32_e32 v, v", out, cyc);
    run<10, 6>("v_fma_f32 v, v, s", out, cyc);
    run<11, 6>("v_mul_f32_e32 d, d, v  (read-modify-write)", out, cyc);
    run<12, 6>("v_max_f32_e32 d, d, v  (read-modify-write)", out, cyc);
    run<13, 4>("v_exp_f32_e32 d, d", out, cyc);
    run<14, 6>("v_fma_f32 d, v, s, 0  (write-only dest)", out, cyc);
    run<15, 6>("v_mul_f32_e32 d, v, v  (write-only dest)", out, cyc);
    run<16, 6>("v_add_f32_e32 d, v, v  (write-only dest)", out, cyc);
    run<17, 6>("v_cvt_pk_f16_f32 d, v, v", out, cyc);
    run<18, 6>("v_cvt_f32_f16_e32 d, v", out, cyc);
    run<19, 6>("v_max3_f32 d, v, 0, 0", out, cyc);
    run<20, 6>("v_fma_f32 d, v, v, v (write-only dest)", out, cyc);
    run<21, 6>("v_mov_b32 d, v", out, cyc);
    run<22, 6>("v_fma_mix_f32 d, v, v, v", out, cyc);
    run<23, 6>("v_med3_f32 d, v, 0, 1.0", out, cyc);
    run<24, 6>("v_and_b32 d, v, v", out, cyc);
    run<25, 6>("v_ldexp_f32 d, v, 11", out, cyc);
    run<0, 8>("v_fma_f32 (VOP3)", out, cyc);
    run<1, 8>("v_fmamk_f32 (VOP2 + 32-bit literal)", out, cyc);
    run<9, 8>("v_fmac_f32_e32 v, v", out, cyc);
    return 0;
}
