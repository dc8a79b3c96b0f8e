// Does TRAPSTS.EXCP (sticky IEEE exception bits, accumulated regardless of EXCP_EN) record the overflow of an fp32 -> fp16 conversion,
// with and without MODE.FP16_OVFL (clamp instead of inf)?  One wave per case; prints TRAPSTS[8:0] before / after.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const float *in, unsigned *out, int ovfl_mode, int do_mix) {
    if (ovfl_mode) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);
    const unsigned before = __builtin_amdgcn_s_getreg(3 | (0 << 6) | (8 << 11));
    const float x = in[threadIdx.x], y = in[threadIdx.x + 64];
    f16x2 b; b[0] = (_Float16)x; b[1] = (_Float16)y;
    unsigned hu = __builtin_bit_cast(unsigned, b), lu = 0;
    if (do_mix) {
        const float c = -2048.f;
        asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lu) : "v"(hu), "s"(c), "v"(x * 2048.f));
    }
    asm volatile("s_nop 4");
    const unsigned after = __builtin_amdgcn_s_getreg(3 | (0 << 6) | (8 << 11));
    out[threadIdx.x] = hu; out[64 + threadIdx.x] = lu;
    if (threadIdx.x == 0) { out[128] = before; out[129] = after; }
}
int main() {
    float h[128]; float *d; unsigned *o, r[130];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    const float cases[][2] = {{1.f, 2.f}, {65504.f, 1.f}, {65519.f, 1.f}, {65520.f, 1.f}, {1e5f, 1.f}, {-3e6f, 1.f}, {1e-9f, 1.f}};
    for (int ov = 0; ov < 2; ++ov)
        for (int mix = 0; mix < 2; ++mix)
            for (auto &c : cases) {
                for (int i = 0; i < 64; ++i) { h[i] = i == 5 ? c[0] : 1.f; h[64 + i] = c[1]; }
                hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
                probe<<<1, 64>>>(d, o, ov, mix);
                hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
                printf("FP16_OVFL=%d mix=%d  x=%-10g -> h=0x%04x  TRAPSTS.EXCP before 0x%03x after 0x%03x  (overflow bit 3: %d, inexact bit 5: %d)\n", ov, mix, c[0],
                       r[5] & 0xffff, r[128] & 0x1ff, r[129] & 0x1ff, (r[129] >> 3) & 1, (r[129] >> 5) & 1);
            }
    return 0;
}
