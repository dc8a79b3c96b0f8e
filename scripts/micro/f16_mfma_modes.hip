// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs, and does MODE.FP16_OVFL saturate v_cvt_f16_f32 / v_cvt_pk_f16_f32?
// (questions behind the two-way fp16 operand split of the fused MLP kernels; prints the answers)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__global__ void k(float *out, float a_val, float b_val, int ovfl) {
    if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)0.f; b[e] = (_Float16)0.f; }
    a[0] = (_Float16)a_val; b[0] = (_Float16)b_val;       // one non-zero k per lane half -> C[i][j] = 2 * a*b (two lane halves)
    f32x16 c;
    for (int e = 0; e < 16; ++e) c[e] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    float big = out[8];
    f16x2 pk; pk[0] = (_Float16)big; pk[1] = (_Float16)(-big);
    _Float16 one = (_Float16)(big * 0.5f);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; out[2] = (float)b[0]; out[3] = (float)pk[0]; out[4] = (float)pk[1]; out[5] = (float)one; }
}
int main() {
    float *d; hipMalloc(&d, 64); float h[16] = {0};
    const float cases[4][2] = {{1.f, 1.f}, {3.0e-6f, 1024.f}, {3.0e-6f, 3.0e-6f * 0 + 16384.f}, {5.0e-5f, 2.f}};
    for (int ov = 0; ov < 2; ++ov)
        for (int c = 0; c < 4; ++c) {
            h[8] = 1.0e6f; hipMemcpy(d, h, 64, hipMemcpyHostToDevice);
            k<<<1, 64>>>(d, cases[c][0], cases[c][1], ov); hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
            printf("ovfl=%d a=%g b=%g: a_f16=%g b_f16=%g mfma c[0]=%g (expected %g)  cvt(1e6)=%g cvt(-1e6)=%g cvt(5e5)=%g\n", ov, cases[c][0], cases[c][1],
                   h[1], h[2], h[0], 2.0 * (double)h[1] * (double)h[2], h[3], h[4], h[5]);
        }
    return 0;
}
