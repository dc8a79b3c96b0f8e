// Are the two operands of v_mfma_f32_32x32x16_{bf16,f16} interchangeable bit for bit?  C = A B with A: 32 x 16 (lane (i, h): row i, k = 8 h .. 8 h + 7),
// B: 16 x 32 (lane (j, h): column j, k = 8 h ..); swapping the operands computes C^T from the same fragments.  Prints the number of
// (row, column) pairs whose two results differ.   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_swap scripts/micro/mfma_swap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <bool F16>
__global__ void k(const float *a, const float *b, float *c1, float *c2, int steps) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f32x16 x = {0}, y = {0};
    for (int s = 0; s < steps; ++s) {
        if (F16) {
            f16x8 fa, fb;
            for (int e = 0; e < 8; ++e) { fa[e] = (_Float16)a[(s * 32 + i) * 16 + 8 * h + e]; fb[e] = (_Float16)b[(s * 32 + i) * 16 + 8 * h + e]; }
            x = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, x, 0, 0, 0);      // x[q] = C[row 8 (q / 4) + 4 h + q % 4 of a][column i of b]
            y = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, y, 0, 0, 0);      // y[q] = C[row i of a... transposed]
        } else {
            bf16x8 fa, fb;
            for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)a[(s * 32 + i) * 16 + 8 * h + e]; fb[e] = (__bf16)b[(s * 32 + i) * 16 + 8 * h + e]; }
            x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, x, 0, 0, 0);
            y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, y, 0, 0, 0);
        }
    }
    for (int q = 0; q < 16; ++q) {
        const int r = 8 * (q / 4) + 4 * h + q % 4;
        c1[r * 32 + i] = x[q];          // x: rows of a (M) x columns of b (N): M index r, N index i
        c2[i * 32 + r] = y[q];          // y: rows of b (M) x columns of a (N): M index r is b's, N index i is a's -> store transposed
    }
}
int main() {
    const int steps = 8;
    float *a, *b, *c1, *c2;
    hipMallocManaged(&a, steps * 32 * 16 * 4); hipMallocManaged(&b, steps * 32 * 16 * 4); hipMallocManaged(&c1, 4096); hipMallocManaged(&c2, 4096);
    srand(1);
    for (int t = 0; t < steps * 32 * 16; ++t) { a[t] = (rand() / (float)RAND_MAX - 0.5f) * 4.f; b[t] = (rand() / (float)RAND_MAX - 0.5f) * 0.3f; }
    for (int f = 0; f < 2; ++f) {
        if (f) k<true><<<1, 64>>>(a, b, c1, c2, steps); else k<false><<<1, 64>>>(a, b, c1, c2, steps);
        hipDeviceSynchronize();
        int diff = 0; float mx = 0.f;
        for (int t = 0; t < 1024; ++t) { if (c1[t] != c2[t]) ++diff; mx = fmaxf(mx, fabsf(c1[t] - c2[t])); }
        printf("%s: %d of 1024 results differ between mfma(A, B) and mfma(B, A)^T, max |diff| %.3e\n", f ? "f16 " : "bf16", diff, mx);
    }
    return 0;
}
