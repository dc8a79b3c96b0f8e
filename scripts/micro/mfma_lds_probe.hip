// One wave per SIMD: v_mfma_f32_32x32x16_f16 fed by B fragments that ds_read_b128 loads — into arch VGPRs or straight into AGPRs — and
// other LDS traffic beside it: what does an LDS instruction cost a lone wave that is issuing MFMAs?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// MODE 0: no LDS; 1: ONE ds_read_b128 per MFMA into VGPRs (used as B two MFMAs later); 2: the same into AGPRs; 3: TWO per MFMA into VGPRs;
// 4: two into AGPRs; 5: one ds_write2st64_b64 per 3 MFMAs; 6: mode 1 + mode 5; 7: mode 2 + mode 5
template <int MODE>
__global__ __launch_bounds__(256, 1) void bench(float *out, unsigned long long *cyc, int rep) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 0.001f * i;
    f16x8 a;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(threadIdx.x * 0.001f + i);
    f32x16 c[3] = {};
    const unsigned rd = (threadIdx.x & 63) * 16, wr = (threadIdx.x & 63) * 8 + 32768;
    u32x4 bv[4] = {}, ba[4] = {};
    u32x2 wv = {1u, 2u};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int cur = u & 3, nxt = (u + 2) & 3;
            if (MODE == 1 || MODE == 3 || MODE == 6) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv[nxt]) : "v"(rd), "i"(1024 * (u % 8)));
                if (MODE == 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv[(u + 3) & 3]) : "v"(rd), "i"(1024 * (u % 8) + 8192));
                asm volatile("s_waitcnt lgkmcnt(%1)\n\tv_mfma_f32_32x32x16_f16 %0, %2, %3, %0" : "+v"(c[u % 3]) : "i"(MODE == 3 ? 4 : 2), "v"(a), "v"(bv[cur]));
            } else if (MODE == 2 || MODE == 4 || MODE == 7) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(ba[nxt]) : "v"(rd), "i"(1024 * (u % 8)));
                if (MODE == 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(ba[(u + 3) & 3]) : "v"(rd), "i"(1024 * (u % 8) + 8192));
                asm volatile("s_waitcnt lgkmcnt(%1)\n\tv_mfma_f32_32x32x16_f16 %0, %2, %3, %0" : "+v"(c[u % 3]) : "i"(MODE == 4 ? 4 : 2), "v"(a), "a"(ba[cur]));
            } else {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[u % 3]) : "v"(a), "v"(bv[cur]));
            }
            if ((MODE >= 5) && (u % 3) == 2) asm volatile("ds_write2st64_b64 %0, %1, %1 offset1:16" :: "v"(wr), "v"(wv) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int u = 0; u < 3; ++u) for (int i = 0; i < 16; ++i) s += c[u][i];
    for (int u = 0; u < 4; ++u) s += bv[u][0] + ba[u][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, float *out, unsigned long long *cyc) {
    const int rep = 2000, blocks = 256;
    bench<MODE><<<blocks, 256>>>(out, cyc, rep);
    bench<MODE><<<blocks, 256>>>(out, cyc, rep);
    hipDeviceSynchronize();
    unsigned long long hh[256];
    hipMemcpy(hh, cyc, sizeof(hh), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks; ++i) m += (double)hh[i];
    m /= blocks;
    printf("%-70s: %6.1f ticks per MFMA\n", name, m / (rep * 12.0));
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<0>("MFMA only", out, cyc);
    run<1>("+ 1 ds_read_b128 per MFMA -> VGPR (B operand, two MFMAs ahead)", out, cyc);
    run<2>("+ 1 ds_read_b128 per MFMA -> AGPR (B operand read from the AGPR)", out, cyc);
    run<3>("+ 2 ds_read_b128 per MFMA -> VGPR", out, cyc);
    run<4>("+ 2 ds_read_b128 per MFMA -> AGPR", out, cyc);
    run<5>("+ 1 ds_write2st64_b64 per 3 MFMAs", out, cyc);
    run<6>("+ 1 read -> VGPR per MFMA + 1 ds_write2st64_b64 per 3 MFMAs", out, cyc);
    run<7>("+ 1 read -> AGPR per MFMA + 1 ds_write2st64_b64 per 3 MFMAs", out, cyc);
    return 0;
}
