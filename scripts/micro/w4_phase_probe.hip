// One matrix phase of mlp_w4_kernel (m_block<EK>: 24 MFMAs + the other tile's vector work) in isolation: 4 waves, one per SIMD, the
// phase called REP times on LDS-resident planes, no barriers, no gathers.  Ticks per phase tell what the MFMA / vector / LDS stream of
// a phase costs by itself (scripts/ws_stamps.py ... w4 gives the phases inside the kernel).
// Build + run (GPU box): hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -Igraphs4cfd_amd/csrc [-DG4C_W4_ABLATE=..] -o /tmp/w4_phase_probe
//                        scripts/micro/w4_phase_probe.hip && /tmp/w4_phase_probe
#include "../../graphs4cfd_amd/csrc/mlp_w4.hip"
#include <cstdio>

template <int EK, bool PACT>
__global__ __launch_bounds__(256, 1) void probe(const float *w, float *out, unsigned long long *cyc, int rep) {
    __shared__ __attribute__((aligned(16))) __bf16 sP[2 * TILE_H];
    __shared__ __attribute__((aligned(16))) float sF[2 * FIN];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kb = lane >> 5;
    const int fbase = 32 * wave + 4 * kb;
    const int pr = tid >> 3, c8 = tid & 7;
    for (int i = tid; i < 2 * TILE_H; i += 256) sP[i] = (__bf16)0.0f;
    for (int i = tid; i < 2 * FIN; i += 256) sF[i] = 0.f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w), 0, 0x7fffffff, 0x00020000);
    u32x4v W[8][2];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) W[s][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(lane * 16 + 1024 * pl), (unsigned)s * 4096u, 0);
    Other oA;
    __bf16 *const sA = sP, *const sB = sP + TILE_H;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        oA.plane_acc[q] = sA + j * PS + 8 * ((4 * wave + q) ^ (j & 15)) + 4 * kb;
        oA.plane_park[q] = sA + pr * PS + 8 * (((c8 >> 1) + 4 * q) ^ (pr & 15)) + 4 * (c8 & 1);
    }
    oA.fin = sF + j * HS + fbase;
    const __bf16 *paB[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) paB[s] = sB + j * PS + 8 * ((2 * s + kb) ^ (j & 15));
    f32x16 accA, accA1, accA2, accB, accB1, accB2;
    for (int i = 0; i < 16; ++i) { accA[i] = 0.01f * i + lane * 1e-3f; accA1[i] = 0.5f; accA2[i] = 0.25f; accB[i] = 0.f; }
    f32x4 xe[4];
    for (int i = 0; i < 4; ++i) xe[i] = f32x4{0.1f * i, 0.2f, -0.3f, 0.4f};
    RangeV rng;
    f16_range_mode();
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rep; ++r) {
        m_block<EK, PACT>(paB, W, accB, accB1, accB2, accA, accA1, accA2, xe, oA, rng);
        asm volatile("" : "+v"(accA), "+v"(accA1), "+v"(accA2));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = rng.m;
    for (int i = 0; i < 16; ++i) s += accB[i] + accB1[i] + accB2[i];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int EK, bool PACT>
void run(const char *name, const float *w, float *out, unsigned long long *cyc) {
    const int rep = 2000, blocks = 256;
    probe<EK, PACT><<<blocks, 256>>>(w, out, cyc, rep);
    probe<EK, PACT><<<blocks, 256>>>(w, out, cyc, rep);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks; ++i) m += (double)h[i];
    m /= blocks;
    printf("%-46s %7.0f ticks per phase (24 MFMAs: %5.1f per MFMA)\n", name, m / rep, m / rep / 24.0);
}

int main() {
    float *w, *out; unsigned long long *cyc;
    hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20);
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    printf("G4C_W4_ABLATE = %d\n", G4C_W4_ABLATE);
    run<0, false>("M only (no other-tile work)", w, out, cyc);
    run<1, false>("M + hidden epilogue (EK 1)", w, out, cyc);
    run<2, true>("M + park with SELU on load (EK 2)", w, out, cyc);
    run<3, false>("M + fp32 rows (EK 3)", w, out, cyc);
    run<4, true>("M + fp32 rows + park with SELU (EK 4)", w, out, cyc);
    return 0;
}
