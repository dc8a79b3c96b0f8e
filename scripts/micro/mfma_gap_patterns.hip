// One wave per SIMD: a "double step" of mlp_w4_kernel = six v_mfma_f32_32x32x16_f16 on three rotating accumulators with the vector work
// of one group of four values in the six gaps (SELU carried as S = y * 2^11, two-way fp16 split), registers only (no LDS, no memory).
// PATTERN selects how the ~30 vector instructions are spread over the gaps.  Ticks per double step; 192 = the six MFMAs alone.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(c) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(b))
#define MUL4 asm volatile("v_mul_f32 %0, 0x3fb8aa3b, %4\n\tv_mul_f32 %1, 0x3fb8aa3b, %5\n\tv_mul_f32 %2, 0x3fb8aa3b, %6\n\tv_mul_f32 %3, 0x3fb8aa3b, %7" : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3))
#define EXP(e) asm volatile("v_exp_f32 %0, %0 clamp" : "+v"(e))
#define MAX4 asm volatile("v_max_f32 %0, 0, %4\n\tv_max_f32 %1, 0, %5\n\tv_max_f32 %2, 0, %6\n\tv_max_f32 %3, 0, %7" : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3))
#define FMK(e) asm volatile("v_fmamk_f32 %0, %0, 0x45610966, %1" : "+v"(e) : "v"(nsa))
#define FMC(e, m) asm volatile("v_fmac_f32 %0, 0x45067d5f, %1" : "+v"(e) : "v"(m))
#define MAX3 asm volatile("v_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%3|, |%4|" : "+v"(rng) : "v"(e0), "v"(e1), "v"(e2), "v"(e3))
#define MIXH(h, x, y) asm volatile("v_fma_mixlo_f16 %0, %1, %3, 0\n\tv_fma_mixhi_f16 %0, %2, %3, 0" : "=&v"(h) : "v"(x), "v"(y), "s"(up))
#define MIXL(l, h, x, y) asm volatile("v_fma_mixlo_f16 %0, %1, %4, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %4, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(l) : "v"(h), "v"(x), "v"(y), "s"(dn))
// the split without v_fma_mix: y = S * 2^-11; h = cvt_pk(y0, y1); hf = f32(h) (low half: plain cvt, high half: SDWA WORD_1);
// d = fma(hf, -2^11, S) (exact); l = cvt_pk(d0, d1)
#define YMUL(y, s) asm volatile("v_mul_f32 %0, 0x3a000000, %1" : "=v"(y) : "v"(s))
#define CVTPK(h, x, y) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x), "v"(y))
#define CVTLO(f, h) asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(f) : "v"(h))
#define CVTHI(f, h) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f) : "v"(h))
#define DFMA(d, f, s) asm volatile("v_fma_f32 %0, %1, %3, %2" : "=v"(d) : "v"(f), "v"(s), "s"(dn))
#define NEXTU asm volatile("v_add_f32 %0, %4, %0\n\tv_add_f32 %1, %4, %1\n\tv_add_f32 %2, %4, %2\n\tv_add_f32 %3, %4, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(eps))

template <int PATTERN, int THREADS = 256>
__global__ __launch_bounds__(THREADS, 1) void bench(float *out, unsigned long long *cyc, int rep) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 c0 = {}, c1 = {}, c2 = {};
    float u0 = 0.1f * threadIdx.x, u1 = -0.3f, u2 = 0.7f, u3 = -1.1f, e0 = 0, e1 = 0, e2 = 0, e3 = 0, m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    float s0 = 1.f, s1 = 2.f, s2 = 3.f, s3 = 4.f;          // S values of the previous group (late pieces)
    float rng = 0.f, nsa = -3600.5875f, eps = 1e-3f;
    const float up = 1.f / 2048.f, dn = -2048.f;
    unsigned h0 = 0, h1 = 0, l0 = 0, l1 = 0, acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rep; ++r) {
        if (PATTERN == 0) {            // MFMAs only
            MFMA(c0); MFMA(c1); MFMA(c2); MFMA(c0); MFMA(c1); MFMA(c2);
        }
        if (PATTERN == 1) {            // the pinned schedule of mlp_w4.hip (park: no fold): late pieces in gaps 0-3, early stages 2-5
            MFMA(c0); MIXH(h0, s0, s1);
            MFMA(c1); MIXH(h1, s2, s3);
            MFMA(c2); MIXL(l0, h0, s0, s1); MUL4; EXP(e0);
            MFMA(c0); MIXL(l1, h1, s2, s3); EXP(e1); EXP(e2);
            MFMA(c1); EXP(e3); MAX4; FMK(e0); FMK(e1);
            MFMA(c2); FMK(e2); FMK(e3); FMC(e0, m0); FMC(e1, m1); FMC(e2, m2); FMC(e3, m3); MAX3;
        }
        if (PATTERN == 2) {            // no v_fma_mix at all
            MFMA(c0);
            MFMA(c1);
            MFMA(c2); MUL4; EXP(e0);
            MFMA(c0); EXP(e1); EXP(e2);
            MFMA(c1); EXP(e3); MAX4; FMK(e0); FMK(e1);
            MFMA(c2); FMK(e2); FMK(e3); FMC(e0, m0); FMC(e1, m1); FMC(e2, m2); FMC(e3, m3); MAX3;
        }
        if (PATTERN == 3) {            // no v_exp
            MFMA(c0); MIXH(h0, s0, s1);
            MFMA(c1); MIXH(h1, s2, s3);
            MFMA(c2); MIXL(l0, h0, s0, s1); MUL4;
            MFMA(c0); MIXL(l1, h1, s2, s3);
            MFMA(c1); MAX4; FMK(e0); FMK(e1);
            MFMA(c2); FMK(e2); FMK(e3); FMC(e0, m0); FMC(e1, m1); FMC(e2, m2); FMC(e3, m3); MAX3;
        }
        if (PATTERN == 4) {            // everything bunched: stage by stage, one stage per gap (the first w4 schedule)
            MFMA(c0); MUL4;
            MFMA(c1); EXP(e0); EXP(e1); EXP(e2); EXP(e3);
            MFMA(c2); MAX4; FMK(e0); FMK(e1); FMK(e2); FMK(e3);
            MFMA(c0); FMC(e0, m0); FMC(e1, m1); FMC(e2, m2); FMC(e3, m3); MAX3;
            MFMA(c1); MIXH(h0, s0, s1); MIXH(h1, s2, s3);
            MFMA(c2); MIXL(l0, h0, s0, s1); MIXL(l1, h1, s2, s3);
        }
        if (PATTERN == 5) {            // evenly: five instructions per gap
            MFMA(c0); MIXH(h0, s0, s1); MUL4;                                      // 6
            MFMA(c1); MIXH(h1, s2, s3); EXP(e0); EXP(e1);                          // 4
            MFMA(c2); MIXL(l0, h0, s0, s1); EXP(e2); EXP(e3);                      // 4
            MFMA(c0); MIXL(l1, h1, s2, s3); MAX4;                                  // 6
            MFMA(c1); FMK(e0); FMK(e1); FMK(e2); FMK(e3); FMC(e0, m0);             // 5
            MFMA(c2); FMC(e1, m1); FMC(e2, m2); FMC(e3, m3); MAX3;                 // 5
        }
        if (PATTERN == 6) {            // only the 16 plain ones (MUL4, MAX4, FMK x4, FMC x4, MAX3 x2 = 18), evenly
            MFMA(c0); MUL4;
            MFMA(c1); MAX4;
            MFMA(c2); FMK(e0); FMK(e1); FMK(e2); FMK(e3);
            MFMA(c0); FMC(e0, m0); FMC(e1, m1);
            MFMA(c1); FMC(e2, m2); FMC(e3, m3);
            MFMA(c2); MAX3;
        }
        if (PATTERN == 7) {            // the split on plain instructions: 16 instead of 8 v_fma_mix; 34 vector instructions in six gaps
            float y0, y1, y2, y3, f0, f1, f2, f3, d0, d1, d2, d3;
            MFMA(c0); YMUL(y0, s0); YMUL(y1, s1); YMUL(y2, s2); YMUL(y3, s3); CVTPK(h0, y0, y1); CVTPK(h1, y2, y3);     // 6
            MFMA(c1); CVTLO(f0, h0); CVTHI(f1, h0); CVTLO(f2, h1); CVTHI(f3, h1); MUL4;                                    // 5 (MUL4 = 4) -> 8
            MFMA(c2); DFMA(d0, f0, s0); DFMA(d1, f1, s1); DFMA(d2, f2, s2); DFMA(d3, f3, s3); EXP(e0);                     // 5
            MFMA(c0); CVTPK(l0, d0, d1); CVTPK(l1, d2, d3); EXP(e1); EXP(e2);                                               // 4
            MFMA(c1); EXP(e3); MAX4; FMK(e0); FMK(e1);                                                                      // 7
            MFMA(c2); FMK(e2); FMK(e3); FMC(e0, m0); FMC(e1, m1); FMC(e2, m2); FMC(e3, m3); MAX3;                           // 8
        }
        if (PATTERN == 9) {            // Veltkamp split: Sp = S * 8193, Sq = S - Sp, Sh = Sq + Sp (S rounded to 11 bits), d = S - Sh, h = cvt_pk(Sh * 2^-11), l = cvt_pk(d)
            float p0, p1, p2, p3, q0, q1, q2, q3;
#define VK1(p, s) asm volatile("v_mul_f32 %0, 0x46000400, %1" : "=v"(p) : "v"(s))
#define VK2(q, s, p) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(q) : "v"(s), "v"(p))
#define VK3(p, q) asm volatile("v_add_f32 %0, %1, %0" : "+v"(p) : "v"(q))
#define VK4(q, s, p) asm volatile("v_sub_f32 %0, %1, %2\n\tv_mul_f32 %2, 0x3a000000, %2" : "=&v"(q), "+v"(s), "+v"(p))
            MFMA(c0); VK1(p0, s0); VK1(p1, s1); VK1(p2, s2); VK1(p3, s3); MUL4;                                               // 8
            MFMA(c1); VK2(q0, s0, p0); VK2(q1, s1, p1); VK2(q2, s2, p2); VK2(q3, s3, p3); EXP(e0);                            // 5
            MFMA(c2); VK3(p0, q0); VK3(p1, q1); VK3(p2, q2); VK3(p3, q3); EXP(e1);                                            // 5
            MFMA(c0); VK4(q0, s0, p0); VK4(q1, s1, p1); VK4(q2, s2, p2); VK4(q3, s3, p3); EXP(e2);                            // 9
            MFMA(c1); CVTPK(h0, p0, p1); CVTPK(h1, p2, p3); CVTPK(l0, q0, q1); CVTPK(l1, q2, q3); EXP(e3); MAX4;              // 9
            MFMA(c2); FMK(e0); FMK(e1); FMK(e2); FMK(e3); FMC(e0, m0); FMC(e1, m1); FMC(e2, m2); FMC(e3, m3); MAX3;           // 10
        }
        if (PATTERN == 8) {            // pattern 7 without the exps (the plain part alone: 30 instructions)
            float y0, y1, y2, y3, f0, f1, f2, f3, d0, d1, d2, d3;
            MFMA(c0); YMUL(y0, s0); YMUL(y1, s1); YMUL(y2, s2); YMUL(y3, s3); CVTPK(h0, y0, y1); CVTPK(h1, y2, y3);
            MFMA(c1); CVTLO(f0, h0); CVTHI(f1, h0); CVTLO(f2, h1); CVTHI(f3, h1); MUL4;
            MFMA(c2); DFMA(d0, f0, s0); DFMA(d1, f1, s1); DFMA(d2, f2, s2); DFMA(d3, f3, s3);
            MFMA(c0); CVTPK(l0, d0, d1); CVTPK(l1, d2, d3);
            MFMA(c1); MAX4; FMK(e0); FMK(e1);
            MFMA(c2); FMK(e2); FMK(e3); FMC(e0, m0); FMC(e1, m1); FMC(e2, m2); FMC(e3, m3); MAX3;
        }
        if (PATTERN >= 10 && PATTERN < 30) {      // eight instructions of ONE kind per double step (two in each of four gaps), on distinct live registers
            float y0 = s0, y1 = s1, y2 = s2, y3 = s3;
#define K2(a_, b_) do { \
            if (PATTERN == 10) { asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h0) : "v"(a_), "v"(b_)); asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h1) : "v"(b_), "v"(a_)); } \
            if (PATTERN == 11) { asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(y0) : "v"(h0)); asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(y1) : "v"(h1)); } \
            if (PATTERN == 12) { asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(y0) : "v"(h0)); asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(y1) : "v"(h1)); } \
            if (PATTERN == 13) { asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(h0) : "v"(a_), "s"(up)); asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(h1) : "v"(b_), "s"(up)); } \
            if (PATTERN == 14) { asm volatile("v_lshl_add_u32 %0, %1, 13, %2" : "=v"(h0) : "v"(a_), "v"(b_)); asm volatile("v_bfi_b32 %0, %1, %2, %1" : "=v"(h1) : "v"(b_), "v"(a_)); } \
            if (PATTERN == 15) { asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(y0) : "v"(h0), "s"(dn), "v"(a_)); asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(y1) : "v"(h0), "s"(dn), "v"(b_)); } \
            if (PATTERN == 16) { asm volatile("v_exp_f32 %0, %1" : "=v"(y0) : "v"(a_)); asm volatile("v_exp_f32 %0, %1" : "=v"(y1) : "v"(b_)); } \
            if (PATTERN == 17) { asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h0) : "v"(a_), "v"(b_)); asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h1) : "v"(b_), "v"(a_)); } \
            if (PATTERN == 18) { asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(h0) : "v"(h1), "v"(l0)); asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(l1) : "v"(h1), "v"(l0)); } \
            if (PATTERN == 19) { asm volatile("v_rcp_f32 %0, %1" : "=v"(y0) : "v"(a_)); asm volatile("v_rsq_f32 %0, %1" : "=v"(y1) : "v"(b_)); } \
            if (PATTERN == 20) { asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(h0) : "v"(a_)); asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(h1) : "v"(b_)); } \
            if (PATTERN == 21) { asm volatile("v_ldexp_f32 %0, %1, 11" : "=v"(y0) : "v"(a_)); asm volatile("v_frexp_mant_f32 %0, %1" : "=v"(y1) : "v"(b_)); } \
            acc ^= h0 ^ h1; rng += y0 + y1; } while (0)
            MFMA(c0); K2(s0, s1);
            MFMA(c1); K2(s2, s3);
            MFMA(c2); K2(s1, s2);
            MFMA(c0); K2(s3, s0);
            MFMA(c1);
            MFMA(c2);
#undef K2
        }
        NEXTU;
        s0 = e0; s1 = e1; s2 = e2; s3 = e3;
        acc ^= h0 ^ h1 ^ l0 ^ l1;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = rng + u0 + u1 + u2 + u3 + (float)acc;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int PATTERN, int THREADS = 256>
void run(const char *name, float *out, unsigned long long *cyc) {
    const int rep = 4000, blocks = 256;
    bench<PATTERN, THREADS><<<blocks, THREADS>>>(out, cyc, rep);
    bench<PATTERN, THREADS><<<blocks, THREADS>>>(out, cyc, rep);
    hipDeviceSynchronize();
    unsigned long long hh[256];
    hipMemcpy(hh, cyc, sizeof(hh), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks; ++i) m += (double)hh[i];
    m /= blocks;
    printf("[%d waves/SIMD] %-78s: %6.1f ticks per double step and wave (%5.1f per MFMA and SIMD)\n", THREADS / 256, name, m / rep, m / rep / 6.0 / (THREADS / 256));
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    run<0>("six MFMAs only (+ the loop's 4 adds, 3 xors)", out, cyc);
    run<1>("pinned w4 schedule (late pieces in gaps 0-3, early stages in 2-5)", out, cyc);
    run<2>("the same without the 8 v_fma_mix", out, cyc);
    run<3>("the same without the 4 v_exp_f32", out, cyc);
    run<4>("stage by stage, one stage per gap", out, cyc);
    run<5>("4 - 6 instructions in every gap", out, cyc);
    run<6>("only the 18 plain instructions", out, cyc);
    run<7>("split on plain instructions (mul, cvt_pk, cvt_f32_f16, fma, cvt_pk) + 4 v_exp_f32", out, cyc);
    run<8>("the same without the 4 v_exp_f32 (30 plain instructions)", out, cyc);
    run<9>("Veltkamp split on plain instructions + 4 v_exp_f32 (46 instructions)", out, cyc);
    run<0, 512>("six MFMAs only", out, cyc);
    run<1, 512>("pinned w4 schedule (8 v_fma_mix, 4 v_exp_f32, 18 plain)", out, cyc);
    run<2, 512>("the same without the 8 v_fma_mix", out, cyc);
    run<3, 512>("the same without the 4 v_exp_f32", out, cyc);
    run<6, 512>("only the 18 plain instructions", out, cyc);
    run<9, 512>("Veltkamp split on plain instructions + 4 v_exp_f32 (46 instructions)", out, cyc);
    run<16, 512>("8 x v_exp_f32", out, cyc);
    run<15, 512>("8 x v_fma_mix_f32 (f16 source)", out, cyc);
    run<10>("8 x v_cvt_pk_f16_f32 (+ 8 plain: 4 xor, 4 add)", out, cyc);
    run<11>("8 x v_cvt_f32_f16_e32", out, cyc);
    run<12>("8 x v_cvt_f32_f16_sdwa WORD_1", out, cyc);
    run<13>("8 x v_fma_mixlo_f16", out, cyc);
    run<14>("4 x v_lshl_add_u32 + 4 x v_bfi_b32", out, cyc);
    run<15>("8 x v_fma_mix_f32 (f16 source)", out, cyc);
    run<16>("8 x v_exp_f32", out, cyc);
    run<17>("8 x v_cvt_pkrtz_f16_f32", out, cyc);
    run<18>("4 x v_pk_mul_f16 + 4 x v_pk_add_f16", out, cyc);
    run<19>("4 x v_rcp_f32 + 4 x v_rsq_f32", out, cyc);
    run<20>("8 x v_cvt_f16_f32", out, cyc);
    run<21>("4 x v_ldexp_f32 + 4 x v_frexp_mant_f32", out, cyc);
    return 0;
}
