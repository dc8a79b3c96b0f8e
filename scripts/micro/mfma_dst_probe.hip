// One wave per SIMD, v_mfma_f32_32x32x16_f16 + six vector instructions behind each: what makes "d = op(x, y)" (destination not among
// the sources) cost ~3 ticks more per instruction than "d = op(d, y)"?  Variants: accumulators in VGPRs / AGPRs, destination register
// recently read or not, operands spread over banks, mixtures.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, bool C_AGPR, bool AB_AGPR = false>
__global__ __launch_bounds__(256, 1) void bench(float *out, unsigned long long *cyc, int rep) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 c[3] = {};
    float f[12], g[12], h[12];
    for (int i = 0; i < 12; ++i) { f[i] = threadIdx.x * 1e-3f + i; g[i] = 0.5f * i; h[i] = 0.25f * i; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            if (AB_AGPR && C_AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[u % 3]) : "a"(a), "a"(b));
            else if (AB_AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[u % 3]) : "a"(a), "a"(b));
            else if (C_AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[u % 3]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[u % 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (MODE == 0) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(f[k]) : "v"(g[k]));                       // RMW
                if (MODE == 1) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(f[k]) : "v"(g[k]), "v"(h[k]));             // write-only
                if (MODE == 2) {                                                                                           // ping-pong: f <- g, g <- f (dst read by the next one)
                    if (k & 1) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(g[k]) : "v"(f[k - 1]), "v"(h[k]));
                    else asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(f[k]) : "v"(g[k + 1]), "v"(h[k]));
                }
                if (MODE == 3) {                                                                                           // 3 RMW + 3 write-only
                    if (k < 3) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(f[k]) : "v"(g[k]));
                    else asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(f[k]) : "v"(g[k]), "v"(h[k]));
                }
                if (MODE == 4) asm volatile("v_mul_f32_e32 %0, %1, %1" : "=v"(f[k]) : "v"(g[k]));                       // write-only, ONE source register
                if (MODE == 5) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(f[k]) : "v"(g[k]));                       // RMW via src1
                if (MODE == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[k]) : "v"(g[k]), "v"(h[k]));             // RMW via src2 (fmac form)
                if (MODE == 7) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(f[k]) : "v"(f[(k + 6)]), "v"(h[k]));       // write-only, source = another f (never written)
                if (MODE == 9) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(f[k]) : "a"(g[k]));                          // AGPR -> VGPR copy
                if (MODE == 8) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(f[k]) : "v"(f[(k + 1) % 6]), "v"(h[k]));   // write-only, source = the NEXT one's destination (written an MFMA ago)
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 12; ++i) s += f[i] + g[i] + h[i];
    for (int u = 0; u < 3; ++u) for (int i = 0; i < 16; ++i) s += c[u][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, bool C_AGPR, bool AB_AGPR = false>
void run(const char *name, float *out, unsigned long long *cyc) {
    const int rep = 2000, blocks = 256;
    bench<MODE, C_AGPR, AB_AGPR><<<blocks, 256>>>(out, cyc, rep);
    bench<MODE, C_AGPR, AB_AGPR><<<blocks, 256>>>(out, cyc, rep);
    hipDeviceSynchronize();
    unsigned long long hh[256];
    hipMemcpy(hh, cyc, sizeof(hh), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks; ++i) m += (double)hh[i];
    m /= blocks;
    printf("A,B in %s  acc in %s  %-78s: %6.1f ticks per MFMA\n", AB_AGPR ? "AGPR" : "VGPR", C_AGPR ? "AGPR" : "VGPR", name, m / (rep * 12.0));
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<0, false>("6 x d = d * x   (read-modify-write)", out, cyc);
    run<1, false>("6 x d = x * y   (destination not a source)", out, cyc);
    run<2, false>("6 x d = x * y, each destination is the next instruction's source (ping-pong)", out, cyc);
    run<3, false>("3 x d = d * x, 3 x d = x * y", out, cyc);
    run<4, false>("6 x d = x * x   (one source register)", out, cyc);
    run<5, false>("6 x d = x * d   (read-modify-write through src1)", out, cyc);
    run<6, false>("6 x d = x * y + d (fmac form)", out, cyc);
    run<7, false>("6 x d = e * y   (e: a register no instruction writes)", out, cyc);
    run<8, false>("6 x d[k] = d[k+1] * y (source written one MFMA ago by a neighbour)", out, cyc);
    run<0, true>("6 x d = d * x   (read-modify-write)", out, cyc);
    run<1, true>("6 x d = x * y   (destination not a source)", out, cyc);
    run<2, true>("6 x d = x * y ping-pong", out, cyc);
    run<8, true>("6 x d[k] = d[k+1] * y", out, cyc);
    run<1, false, true>("6 x d = x * y   (destination not a source)", out, cyc);
    run<1, true, true>("6 x d = x * y   (destination not a source)", out, cyc);
    run<0, true, true>("6 x d = d * x", out, cyc);
    run<9, false, false>("6 x v_accvgpr_read d, a", out, cyc);
    run<9, true, true>("6 x v_accvgpr_read d, a", out, cyc);
    return 0;
}
