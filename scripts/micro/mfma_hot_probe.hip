// One wave per SIMD: v_mfma_f32_32x32x16_f16 + six v_mul_f32 behind each.  Does a vector instruction hide behind the MFMA only when one
// of its sources was written recently?  PERIOD = number of MFMA slots between two touches of the same register (1: every slot).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: d = d * x (RMW)   1: d = x * y (write-only, cold sources)   2: d[k] = d[k+1] * y
// MODE 3: write-only with s_nop 1 between the fillers   4: write-only, fillers BEFORE the MFMA's s_nop-free slot (MFMA last)
template <int MODE, int PERIOD>
__global__ __launch_bounds__(256, 1) void bench(float *out, unsigned long long *cyc, int rep) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 c[3] = {};
    float f[48], g[8];
    for (int i = 0; i < 48; ++i) f[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 8; ++i) g[i] = 0.5f + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[u % 3]) : "v"(a), "v"(b));
            const int base = 6 * (u % PERIOD);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (MODE == 0) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(f[base + k]) : "v"(g[k]));
                if (MODE == 1) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(f[base + k]) : "v"(g[k]), "v"(g[k + 1]));
                if (MODE == 2) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(f[base + k]) : "v"(f[base + (k + 1) % 6]), "v"(g[k]));
                if (MODE == 3) asm volatile("v_mul_f32_e32 %0, %1, %2\n\ts_nop 1" : "=v"(f[base + k]) : "v"(g[k]), "v"(g[k + 1]));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 48; ++i) s += f[i];
    for (int i = 0; i < 8; ++i) s += g[i];
    for (int u = 0; u < 3; ++u) for (int i = 0; i < 16; ++i) s += c[u][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int PERIOD>
void run(const char *name, float *out, unsigned long long *cyc) {
    const int rep = 1000, blocks = 256;
    bench<MODE, PERIOD><<<blocks, 256>>>(out, cyc, rep);
    bench<MODE, PERIOD><<<blocks, 256>>>(out, cyc, rep);
    hipDeviceSynchronize();
    unsigned long long hh[256];
    hipMemcpy(hh, cyc, sizeof(hh), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks; ++i) m += (double)hh[i];
    m /= blocks;
    printf("%-64s each register touched every %d MFMA slot(s): %6.1f ticks per MFMA\n", name, PERIOD, m / (rep * 24.0));
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<0, 1>("6 x d = d * x", out, cyc); run<0, 2>("6 x d = d * x", out, cyc); run<0, 4>("6 x d = d * x", out, cyc); run<0, 8>("6 x d = d * x", out, cyc);
    run<1, 1>("6 x d = x * y (cold sources)", out, cyc); run<1, 4>("6 x d = x * y (cold sources)", out, cyc);
    run<2, 1>("6 x d[k] = d[k+1] * y", out, cyc); run<2, 2>("6 x d[k] = d[k+1] * y", out, cyc); run<2, 4>("6 x d[k] = d[k+1] * y", out, cyc); run<2, 8>("6 x d[k] = d[k+1] * y", out, cyc);
    run<3, 1>("6 x (d = x * y; s_nop 1)", out, cyc);
    return 0;
}
