// "w4": weight-stationary persistent form of the fused split-operand MLP on FOUR waves, one per SIMD, each with the SIMD's whole
// 512-register file — the MP layers' message launch in the f16x3 stream: gather -> [SELU on load] -> Linear/SELU chain as two-way
// fp16 split products on v_mfma_f32_32x32x16_f16 -> LayerNorm -> activation -> store (-> per-target aggregation).  Replaces
// MLP.forward (graphs4cfd/nn/blocks.py:117-144) with the torch.cat / index ops in front of it (nn/blocks.py:181,328) and, with AGG,
// the scatter(e', col, reduce) behind it (nn/blocks.py:183,330).  Same arithmetic per element as mlp_ws_kernel / mlp_bx6_kernel<.., 2>
// (sums over k in another association: equal within fp32 rounding; the fused aggregation bit-identical to g4c_segment_reduce of
// the stored rows).
//
// Why another form (round 4): mlp_ws_kernel (8 waves, two per SIMD, v_mfma_f32_16x16x32) is bound by vector-instruction
// ISSUE — 6 vector instructions per 16-cycle MFMA, of which ~2 hide (MfmaUtil 27 %, VALUBusy 49 %, not overlapping).  A
// v_mfma_f32_32x32x16 occupies the matrix pipe for 32 cycles and hides ~5 single-issue instructions behind it
// (MI355X_MICROARCH.md "one wave per SIMD"), and one wave per SIMD may keep 512 registers:
//   * wave w owns output features [32 w, 32 w + 32) of EVERY layer: its slice of all three layers' weights — 3 layers x 8 k-steps x
//     2 planes x 16 bytes per lane = 192 registers — is loaded once per launch (A operand: 32 features x 16 k);
//   * the B operand (16 k x 32 rows) is a whole 32-row tile: half as many MFMA issues and half the LDS fragment traffic per flop
//     (4 waves read each tile instead of 8), accumulators of a tile = 16 registers per product stream;
//   * inside a pair of tiles the two tiles alternate layer by layer as before (the epilogue of one under the MFMAs of the other),
//     but in ONE wave's instruction stream: hipcc's scheduler places the 5 - 6 vector instructions behind each MFMA;
//   * the next pair's rows (8 loads), additive rows (16 loads) and tables are in flight in registers across the whole pair.
// Envelope: the f16x3 stream (SP = 2), ONE weighted 128-wide aligned fp32 input block (rows direct or through an index, optional SELU
// on load), 0 or 2 additive 128-wide blocks, two or three layers, 128-wide fp32 output rows without residual / heads; an output index
// only without the fused aggregation.  Everything else keeps mlp_ws_kernel / mlp_bx6_kernel.
#include "mlp_common.h"
#include <cstdlib>
using namespace g4cm;

#ifdef G4C_W4_TIMING
__device__ unsigned long long g4c_w4_stamps[256 * 32];
extern "C" int g4c_w4_read_stamps(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g4c_w4_stamps), sizeof(unsigned long long) * n);
}
#define W4_STAMP(k) do { if (it == 1 && tid == 0 && blockIdx.x < 256) g4c_w4_stamps[blockIdx.x * 32 + (k)] = __builtin_readcyclecounter(); } while (0)
#define W4_STAMP_ONCE(k, v) do { if (tid == 0 && blockIdx.x < 256) g4c_w4_stamps[blockIdx.x * 32 + (k)] = (v); } while (0)      // 12 kernel start, 13 end, 14 pairs
#else
#define W4_STAMP(k) do {} while (0)
#define W4_STAMP_ONCE(k, v) do {} while (0)
#endif

// timing-only ablations (wrong results): 2 no MFMAs, 4 no B-fragment reads after the first double step, 8 no plane / fp32-row writes,
// 16 no SELU, 32 no fp16 split, 64 no gathers of the next pair's rows
#ifndef G4C_W4_ABLATE
#define G4C_W4_ABLATE 0
#endif

namespace {

// Operand planes: [32 rows][128 k] fp16, no padding; the 16-byte granule c of row r lives at granule c ^ (r & 15) (mlp_ws.hip): the
// B-fragment reads of v_mfma_f32_32x32x16 (lane (j, kb): row j, granule 2 s + kb; a ds_read_b128 is served in 16-lane groups whose
// rows are distinct mod 16) are conflict-free.
constexpr int PS = 128;                 // row stride of a plane (elements)
constexpr int PLN = 32 * PS;            // elements of one operand plane of a 32-row tile
constexpr int TILE_H = 2 * PLN;         // two planes (h, l * 2^11)
constexpr int FIN = 32 * HS;            // floats of a tile's fp32 final rows [32][132]
constexpr int SEGCAP = 64;              // segment offsets of a tile staged in LDS (more segments: read from global memory)
constexpr int NT = 256;

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

// The MFMAs are inline assembly: the stationary weights are "a" operands — they live in the accumulator half of the register file for
// the whole launch and are read from there (hipcc's builtin keeps MFMA A / B operands in arch VGPRs and shuttles 192 loop-invariant
// registers through v_accvgpr_read / _write every pair: 280 vector-issue slots per pair) — and the accumulators are "v" operands (the
// epilogue reads them with vector instructions).  What hipcc does not do for an asm statement (cdna_hip_programming.md 5.7): it pads
// no hazard — an MFMA's result may be read or overwritten by a non-MFMA instruction 12 wait states after an 8-pass MFMA at the
// earliest: every accumulator written here is next touched (by the other tile's epilogue / bias load) a whole matrix phase and a
// barrier later, and m_block ends with an explicit pad; back-to-back MFMAs that chain through the accumulator need none.
#define G4C_MFMA_ACC(acc, w, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b))
#define G4C_MFMA_NEW(acc, w, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(b))
#ifdef G4C_W4_NOFENCE
#define G4C_FENCE() do {} while (0)
#else
#define G4C_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

struct Meta { int r0[2], n[2], s0[2], s1[2]; };      // rows [r0, r0 + n) and segments [s0, s1) of the two tiles of a pair (wave-uniform)

// ---- The other tile's vector work, pinned instruction by instruction -------------------------------------------------------------
// What the vector ALUs do for the OTHER tile while this tile's MFMAs issue.  A matrix phase is four double steps (two 16-k steps = six
// MFMAs, i.e. six gaps); double step q forms the four values 4 q .. 4 q + 3 of the other tile ("early" stages E0..E5: fold / load,
// SELU) and splits + stores the four values of double step q - 1 ("late" pieces L0..L3: eight v_fma_mix + one LDS store).  Every gap is
// ONE asm statement, so the stream is exactly this one: hipcc moves pure arithmetic across sched_barrier fences before its machine
// scheduler ever sees them, and with a single wave per SIMD the placement decides what hides behind an MFMA — ~6 plain vector
// instructions do, a v_fma_mix (VOP3P) costs ~4.6 and a v_exp_f32 ~2.4 cycles on top of its slot (scripts/micro/mfma_filler_kinds.hip):
// no gap carries more than 8 instructions, two v_fma_mix or two v_exp_f32.  A value is carried as S = y * 2^11 (mlp_ws.hip
// G4C_WS_SCALED): h = fp16(S * 2^-11) (exactly y: one rounding, v_cvt_pk_f16_f32's bits, MODE.FP16_OVFL honoured),
// l = fp16(fma(h, -2^11, S)).
//   EK 1 hidden-layer epilogue: fold the two 2^-11 accumulators, SELU, split, one 8-byte write per plane;
//   EK 2 park four values of the gathered input rows xe[q] (PACT: SELU pending on the stored rows);
//   EK 3 last layer: fp32 values into the tile's final buffer;  EK 4 = 3, then 2 with the NEXT pair's rows;  EK 0 nothing.
struct Other {
    __bf16 *plane_acc[4];     // EK 1: (row j, features fbase + 8 q .. + 3) of the other tile's h plane (swizzled), q = 0..3
    __bf16 *plane_park[4];    // EK 2: (row pr, columns 4 c8 + 32 i .. + 3), i = 0..3
    float *fin;               // EK 3: (row j, feature fbase) of the other tile's fp32 rows (+ 8 q)
};
// registers of one group of four values: u pre-activation, e exp argument -> exp -> S, m max(u, 0); h / l packed fp16 pairs
struct Grp { float u[4], e[4], m[4]; unsigned h0, h1, l0, l1; };
struct Q4 { float v[4]; };

#define G4C_LIT_2M11 "0x3a000000"      /* 2^-11 */
#define G4C_LIT_2P11 "0x45000000"      /* 2^11 */
#define G4C_LIT_LOG2E "0x3fb8aa3b"     /* log2(e) */
#define G4C_LIT_SA "0x45610966"        /* scale * alpha * 2^11 */
#define G4C_LIT_SC "0x45067d5f"        /* scale * 2^11 */
constexpr float W4_NSA = -(1.6732632423543772848170429916717f * 1.0507009873554804934193349852946f * F16_LO_SCALE);

// late pieces of a group whose S values are in g.e
template <int PIECE>
__device__ __forceinline__ void late_piece(Grp &g) {
    const float up = F16_LO_UNSCALE, dn = -F16_LO_SCALE;
    if (PIECE == 0) asm volatile("v_fma_mixlo_f16 %0, %1, %3, 0\n\tv_fma_mixhi_f16 %0, %2, %3, 0" : "=&v"(g.h0) : "v"(g.e[0]), "v"(g.e[1]), "s"(up));
    if (PIECE == 1) asm volatile("v_fma_mixlo_f16 %0, %1, %3, 0\n\tv_fma_mixhi_f16 %0, %2, %3, 0" : "=&v"(g.h1) : "v"(g.e[2]), "v"(g.e[3]), "s"(up));
    if (PIECE == 2) asm volatile("v_fma_mixlo_f16 %0, %1, %4, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %4, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                                 : "=&v"(g.l0) : "v"(g.h0), "v"(g.e[0]), "v"(g.e[1]), "s"(dn));
    if (PIECE == 3) asm volatile("v_fma_mixlo_f16 %0, %1, %4, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %4, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                                 : "=&v"(g.l1) : "v"(g.h1), "v"(g.e[2]), "v"(g.e[3]), "s"(dn));
}
__device__ __forceinline__ void late_store(const Grp &g, __bf16 *dst) {
    u32x2 h2, l2;
    h2[0] = g.h0; h2[1] = g.h1; l2[0] = (G4C_W4_ABLATE & 32) ? g.h0 : g.l0; l2[1] = (G4C_W4_ABLATE & 32) ? g.h1 : g.l1;
    if (G4C_W4_ABLATE & 8) asm volatile("" :: "v"(h2), "v"(l2));
    else {
        *reinterpret_cast<u32x2 *>(dst) = h2;
        *reinterpret_cast<u32x2 *>(dst + PLN) = l2;
    }
}

// early stages.  SRC 0: fold of the accumulators (a0 + 2^-11 a1 + 2^-11 a2), 1: four loaded values x (SELU pending), 2: four loaded
// values, no activation (S = x * 2^11)
template <int SRC, int STAGE>
__device__ __forceinline__ void early_stage(Grp &g, const Q4 &a0, const Q4 &a1, const Q4 &a2, RangeV &rng) {
    const float nsa = W4_NSA;
    if (SRC == 0 && STAGE == 0)
        asm volatile("v_fmamk_f32 %0, %4, " G4C_LIT_2M11 ", %8\n\tv_fmamk_f32 %1, %5, " G4C_LIT_2M11 ", %9\n\t"
                     "v_fmamk_f32 %2, %6, " G4C_LIT_2M11 ", %10\n\tv_fmamk_f32 %3, %7, " G4C_LIT_2M11 ", %11"
                     : "=&v"(g.u[0]), "=&v"(g.u[1]), "=&v"(g.u[2]), "=&v"(g.u[3])
                     : "v"(a1.v[0]), "v"(a1.v[1]), "v"(a1.v[2]), "v"(a1.v[3]), "v"(a0.v[0]), "v"(a0.v[1]), "v"(a0.v[2]), "v"(a0.v[3]));
    if (SRC == 0 && STAGE == 1)
        asm volatile("v_fmac_f32 %0, " G4C_LIT_2M11 ", %4\n\tv_fmac_f32 %1, " G4C_LIT_2M11 ", %5\n\t"
                     "v_fmac_f32 %2, " G4C_LIT_2M11 ", %6\n\tv_fmac_f32 %3, " G4C_LIT_2M11 ", %7"
                     : "+v"(g.u[0]), "+v"(g.u[1]), "+v"(g.u[2]), "+v"(g.u[3]) : "v"(a2.v[0]), "v"(a2.v[1]), "v"(a2.v[2]), "v"(a2.v[3]));
    if (SRC == 1 && STAGE == 1) { g.u[0] = a0.v[0]; g.u[1] = a0.v[1]; g.u[2] = a0.v[2]; g.u[3] = a0.v[3]; }          // (a rename)
    if (SRC == 2) {
        if (STAGE == 1)
            asm volatile("v_mul_f32 %0, " G4C_LIT_2P11 ", %4\n\tv_mul_f32 %1, " G4C_LIT_2P11 ", %5\n\tv_mul_f32 %2, " G4C_LIT_2P11 ", %6\n\tv_mul_f32 %3, " G4C_LIT_2P11 ", %7"
                         : "=&v"(g.e[0]), "=&v"(g.e[1]), "=&v"(g.e[2]), "=&v"(g.e[3]) : "v"(a0.v[0]), "v"(a0.v[1]), "v"(a0.v[2]), "v"(a0.v[3]));
        if (STAGE == 5)
            asm volatile("v_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%3|, |%4|" : "+v"(rng.m) : "v"(g.e[0]), "v"(g.e[1]), "v"(g.e[2]), "v"(g.e[3]));
        return;
    }
    if (G4C_W4_ABLATE & 16) {          // no SELU: S = u * 2^11
        if (STAGE == 5) { for (int i = 0; i < 4; ++i) g.e[i] = g.u[i] * F16_LO_SCALE; }
        return;
    }
    if (STAGE == 2)
        asm volatile("v_mul_f32 %0, " G4C_LIT_LOG2E ", %4\n\tv_mul_f32 %1, " G4C_LIT_LOG2E ", %5\n\tv_mul_f32 %2, " G4C_LIT_LOG2E ", %6\n\t"
                     "v_mul_f32 %3, " G4C_LIT_LOG2E ", %7\n\tv_exp_f32 %0, %0 clamp"
                     : "=&v"(g.e[0]), "=&v"(g.e[1]), "=&v"(g.e[2]), "=&v"(g.e[3]) : "v"(g.u[0]), "v"(g.u[1]), "v"(g.u[2]), "v"(g.u[3]));
    if (STAGE == 3)
        asm volatile("v_exp_f32 %0, %0 clamp\n\tv_exp_f32 %1, %1 clamp" : "+v"(g.e[1]), "+v"(g.e[2]));
    if (STAGE == 4)
        asm volatile("v_exp_f32 %3, %3 clamp\n\tv_max_f32 %4, 0, %8\n\tv_max_f32 %5, 0, %9\n\tv_max_f32 %6, 0, %10\n\tv_max_f32 %7, 0, %11\n\t"
                     "v_fmamk_f32 %0, %0, " G4C_LIT_SA ", %12\n\tv_fmamk_f32 %1, %1, " G4C_LIT_SA ", %12"
                     : "+v"(g.e[0]), "+v"(g.e[1]), "+v"(g.e[2]), "+v"(g.e[3]), "=&v"(g.m[0]), "=&v"(g.m[1]), "=&v"(g.m[2]), "=&v"(g.m[3])
                     : "v"(g.u[0]), "v"(g.u[1]), "v"(g.u[2]), "v"(g.u[3]), "v"(nsa));
    if (STAGE == 5)
        asm volatile("v_fmamk_f32 %2, %2, " G4C_LIT_SA ", %9\n\tv_fmamk_f32 %3, %3, " G4C_LIT_SA ", %9\n\t"
                     "v_fmac_f32 %0, " G4C_LIT_SC ", %5\n\tv_fmac_f32 %1, " G4C_LIT_SC ", %6\n\tv_fmac_f32 %2, " G4C_LIT_SC ", %7\n\tv_fmac_f32 %3, " G4C_LIT_SC ", %8\n\t"
                     "v_max3_f32 %4, %4, |%0|, |%1|\n\tv_max3_f32 %4, %4, |%2|, |%3|"
                     : "+v"(g.e[0]), "+v"(g.e[1]), "+v"(g.e[2]), "+v"(g.e[3]), "+v"(rng.m)
                     : "v"(g.m[0]), "v"(g.m[1]), "v"(g.m[2]), "v"(g.m[3]), "v"(nsa));
}
// gap G = 0..5 of double step q: early stage G of this double step's group (gc), late piece G of the previous one (gp, q > 0)
template <int EK, bool PACT, int G>
__device__ __forceinline__ void other_gap(int q, const f32x16 &accE, const f32x16 &accE1, const f32x16 &accE2, const f32x4 (&xe)[4],
                                          const Other &o, Grp &gc, Grp &gp, Q4 &fin, RangeV &rng) {
    constexpr bool EPI = EK == 1, PARK = EK == 2 || EK == 4, FIN = EK == 3 || EK == 4;
    Q4 a0, a1, a2;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a0.v[i] = accE[4 * q + i]; a1.v[i] = accE1[4 * q + i]; a2.v[i] = accE2[4 * q + i]; }
    if ((EPI || PARK) && q > 0) {
        if (G == 0) late_piece<0>(gp);
        if (G == 1) late_piece<1>(gp);
        if (G == 2) late_piece<2>(gp);
        if (G == 3) { late_piece<3>(gp); late_store(gp, EPI ? o.plane_acc[q - 1] : o.plane_park[q - 1]); }
    }
    if (FIN) {          // last layer's fp32 values: fold in gaps 0 / 1, one 16-byte store
        if (G == 0)
            asm volatile("v_fmamk_f32 %0, %4, " G4C_LIT_2M11 ", %8\n\tv_fmamk_f32 %1, %5, " G4C_LIT_2M11 ", %9\n\t"
                         "v_fmamk_f32 %2, %6, " G4C_LIT_2M11 ", %10\n\tv_fmamk_f32 %3, %7, " G4C_LIT_2M11 ", %11"
                         : "=&v"(fin.v[0]), "=&v"(fin.v[1]), "=&v"(fin.v[2]), "=&v"(fin.v[3])
                         : "v"(a1.v[0]), "v"(a1.v[1]), "v"(a1.v[2]), "v"(a1.v[3]), "v"(a0.v[0]), "v"(a0.v[1]), "v"(a0.v[2]), "v"(a0.v[3]));
        if (G == 1) {
            asm volatile("v_fmac_f32 %0, " G4C_LIT_2M11 ", %4\n\tv_fmac_f32 %1, " G4C_LIT_2M11 ", %5\n\t"
                         "v_fmac_f32 %2, " G4C_LIT_2M11 ", %6\n\tv_fmac_f32 %3, " G4C_LIT_2M11 ", %7"
                         : "+v"(fin.v[0]), "+v"(fin.v[1]), "+v"(fin.v[2]), "+v"(fin.v[3]) : "v"(a2.v[0]), "v"(a2.v[1]), "v"(a2.v[2]), "v"(a2.v[3]));
            f32x4 x;
            x[0] = fin.v[0]; x[1] = fin.v[1]; x[2] = fin.v[2]; x[3] = fin.v[3];
            if (G4C_W4_ABLATE & 8) asm volatile("" :: "v"(x));
            else *reinterpret_cast<f32x4 *>(o.fin + 8 * q) = x;
        }
    }
    if (EPI) {
        if (G == 0) early_stage<0, 0>(gc, a0, a1, a2, rng);
        if (G == 1) early_stage<0, 1>(gc, a0, a1, a2, rng);
        if (G == 2) early_stage<0, 2>(gc, a0, a1, a2, rng);
        if (G == 3) early_stage<0, 3>(gc, a0, a1, a2, rng);
        if (G == 4) early_stage<0, 4>(gc, a0, a1, a2, rng);
        if (G == 5) early_stage<0, 5>(gc, a0, a1, a2, rng);
    }
    if (PARK) {
        Q4 x;
#pragma unroll
        for (int i = 0; i < 4; ++i) x.v[i] = xe[q][i];
        constexpr int SRC = PACT ? 1 : 2;
        if (G == 1) early_stage<SRC, 1>(gc, x, x, x, rng);
        if (G == 2) early_stage<SRC, 2>(gc, x, x, x, rng);
        if (G == 3) early_stage<SRC, 3>(gc, x, x, x, rng);
        if (G == 4) early_stage<SRC, 4>(gc, x, x, x, rng);
        if (G == 5) early_stage<SRC, 5>(gc, x, x, x, rng);
    }
}

// the late pieces of a phase's last group, behind its last MFMA
template <int EK>
__device__ __forceinline__ void other_drain(const Other &o, Grp &g) {
    if (EK == 1 || EK == 2 || EK == 4) {
        late_piece<0>(g); late_piece<1>(g); late_piece<2>(g); late_piece<3>(g);
        late_store(g, EK == 1 ? o.plane_acc[3] : o.plane_park[3]);
    }
}

// One 128-k block for one tile: 8 k-steps of three products each — (Wh, xh) into acc, (Wh, xl) into acc1, (Wl, xh) into acc2 (the two
// 2^-11 streams keep separate accumulators: three streams leave two independent MFMAs between dependent ones).
// pa[s]: this lane's B-operand address (row j, granule (2 s + kb) ^ (j & 15)) in the tile's h plane.  acc1 / acc2 start at 0
// (the first MFMA of each takes the inline constant), acc holds the bias / start values.
template <int EK, bool PACT = false>
__device__ __forceinline__ void m_block(const __bf16 *const (&pa)[8], const u32x4v (&W)[8][2], f32x16 &acc, f32x16 &acc1, f32x16 &acc2,
                                        const f32x16 &accE, const f32x16 &accE1, const f32x16 &accE2, const f32x4 (&xe)[4], const Other &o, RangeV &rng) {
    bf16x8 fh[4], fl[4];          // fragments of two double steps: [2 * (q & 1) + step]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        fh[s] = *reinterpret_cast<const bf16x8 *>(pa[s]);
        fl[s] = *reinterpret_cast<const bf16x8 *>(pa[s] + PLN);
    }
    Grp g[2];
    Q4 fin;
    G4C_FENCE();
    // (VALU write -> MFMA operand: two wait states; the start values may have been formed by vector adds just in front)
    asm volatile("s_nop 1" ::: "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int cur = 2 * (q & 1), nxt = 2 * ((q + 1) & 1);
        if (q + 1 < 4 && !(G4C_W4_ABLATE & 4)) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                fh[nxt + s] = *reinterpret_cast<const bf16x8 *>(pa[2 * q + 2 + s]);
                fl[nxt + s] = *reinterpret_cast<const bf16x8 *>(pa[2 * q + 2 + s] + PLN);
            }
        }
        G4C_FENCE();
#define G4C_W4_OTHER(G) do { if constexpr (EK != 0) other_gap<EK, PACT, G>(q, accE, accE1, accE2, xe, o, g[q & 1], g[(q + 1) & 1], fin, rng); G4C_FENCE(); } while (0)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int fi = (G4C_W4_ABLATE & 4) ? st : cur + st;
            const u32x4v ch = __builtin_bit_cast(u32x4v, fh[fi]), cl = __builtin_bit_cast(u32x4v, fl[fi]);
            const int s = 2 * q + st;
            if (!(G4C_W4_ABLATE & 2)) G4C_MFMA_ACC(acc, W[s][0], ch);
            else asm volatile("" :: "v"(ch), "v"(cl));
            if (st == 0) G4C_W4_OTHER(0); else G4C_W4_OTHER(3);
            if (!(G4C_W4_ABLATE & 2)) { if (s == 0) G4C_MFMA_NEW(acc1, W[s][0], cl); else G4C_MFMA_ACC(acc1, W[s][0], cl); }
            if (st == 0) G4C_W4_OTHER(1); else G4C_W4_OTHER(4);
            if (!(G4C_W4_ABLATE & 2)) { if (s == 0) G4C_MFMA_NEW(acc2, W[s][1], ch); else G4C_MFMA_ACC(acc2, W[s][1], ch); }
            if (st == 0) G4C_W4_OTHER(2); else G4C_W4_OTHER(5);
        }
#undef G4C_W4_OTHER
    }
    other_drain<EK>(o, g[1]);          // (group 3 lives in g[3 & 1])
    // (hazard pad: the accumulators' next reader — the other tile's matrix phase, behind a barrier — or writer must not issue within 12
    // wait states of the last MFMA; the drain, the bias loads and the barrier are more than that, this makes it independent of them)
    asm volatile("s_nop 7\n\ts_nop 3" ::: "memory");
}

// the other tile's work with nothing to overlap with (the first pair's park; tile B's last rows)
template <int EK, bool PACT>
__device__ __forceinline__ void other_all(const f32x16 &accE, const f32x16 &accE1, const f32x16 &accE2, const f32x4 (&xe)[4], const Other &o, RangeV &rng) {
    Grp g[2];
    Q4 fin;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        other_gap<EK, PACT, 0>(q, accE, accE1, accE2, xe, o, g[q & 1], g[(q + 1) & 1], fin, rng);
        other_gap<EK, PACT, 1>(q, accE, accE1, accE2, xe, o, g[q & 1], g[(q + 1) & 1], fin, rng);
        other_gap<EK, PACT, 2>(q, accE, accE1, accE2, xe, o, g[q & 1], g[(q + 1) & 1], fin, rng);
        other_gap<EK, PACT, 3>(q, accE, accE1, accE2, xe, o, g[q & 1], g[(q + 1) & 1], fin, rng);
        other_gap<EK, PACT, 4>(q, accE, accE1, accE2, xe, o, g[q & 1], g[(q + 1) & 1], fin, rng);
        other_gap<EK, PACT, 5>(q, accE, accE1, accE2, xe, o, g[q & 1], g[(q + 1) & 1], fin, rng);
    }
    other_drain<EK>(o, g[1]);
}

// sum over the 16 lanes of a DPP row, four independent sums at once
__device__ __forceinline__ void row16_sum4(float (&v)[4]) {
#define G4C_DPP_ADD(x, ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, false))
#pragma unroll
    for (int i = 0; i < 4; ++i) G4C_DPP_ADD(v[i], 0xB1);          // quad_perm [1,0,3,2]
#pragma unroll
    for (int i = 0; i < 4; ++i) G4C_DPP_ADD(v[i], 0x4E);          // quad_perm [2,3,0,1]
#pragma unroll
    for (int i = 0; i < 4; ++i) G4C_DPP_ADD(v[i], 0x141);         // row_half_mirror
#pragma unroll
    for (int i = 0; i < 4; ++i) G4C_DPP_ADD(v[i], 0x140);         // row_mirror
#undef G4C_DPP_ADD
}

template <bool AGG, bool DIRECT, bool ADDS, int NL>
__global__ __launch_bounds__(NT, 1) void mlp_w4_kernel(const Params p, const int n_pairs) {
    static_assert(NL == 2 || NL == 3, "mlp_w4_kernel: two or three layers");
    __shared__ __attribute__((aligned(16))) __bf16 sP[2 * TILE_H];         // operand planes of tiles A, B (32 768 B)
    __shared__ __attribute__((aligned(16))) float sF[2 * FIN];             // fp32 final rows of tiles A, B (33 792 B)
    __shared__ int sIdx[2][2 * 3 * 32];          // ring of 2: [tile][weighted block, additive 0, additive 1][row]
    __shared__ int sSeg[4][2 * (SEGCAP + 1)];    // ring of 4: [tile][segment offsets seg_off[s0 .. s0 + SEGCAP]]
    __shared__ __attribute__((aligned(16))) float sBias[3 * NP];
    __shared__ __attribute__((aligned(16))) float sGB[2 * NP];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kb = lane >> 5;
    const int fbase = 32 * wave + 4 * kb;                    // accumulator value r: feature fbase + 8 (r >> 2) + (r & 3) of sample row j
    const int pr = tid >> 3, c8 = tid & 7;                   // park layout: row pr, columns 4 c8 + 32 i .. + 3 (i = 0..3): whole 128-byte lines per 8 lanes

    // contiguous range of pairs of this workgroup (XCD-aware: each XCD gets a contiguous share when the grid is a multiple of 8)
    int p_begin, p_end;
    {
        const int G = gridDim.x, b = blockIdx.x;
        const int slot = (G & 7) ? b : (b & 7) * (G >> 3) + (b >> 3);
        p_begin = __builtin_amdgcn_readfirstlane((int)(((long long)slot * n_pairs) / G));
        p_end = __builtin_amdgcn_readfirstlane((int)(((long long)(slot + 1) * n_pairs) / G));
    }
    if (p_begin >= p_end) return;
    W4_STAMP_ONCE(12, __builtin_readcyclecounter());
    W4_STAMP_ONCE(14, (unsigned long long)(p_end - p_begin));

    auto load_meta = [&](int pair) __attribute__((always_inline)) {
        Meta m;
        if (pair > n_pairs - 1) pair = n_pairs - 1;          // (prefetch past the end: a valid pair again, never used)
        const int t0 = 2 * pair;
        if (AGG) {
            const int t1 = t0 + 1 < p.n_tiles ? t0 + 1 : p.n_tiles, t2 = t0 + 2 < p.n_tiles ? t0 + 2 : p.n_tiles;
            const int r0 = p.tile_rows[t0], r1 = p.tile_rows[t1], r2 = p.tile_rows[t2];
            const int q0 = p.tile_seg[t0], q1 = p.tile_seg[t1], q2 = p.tile_seg[t2];
            m.r0[0] = r0; m.n[0] = r1 - r0; m.r0[1] = r1; m.n[1] = r2 - r1;
            m.s0[0] = q0; m.s1[0] = q1; m.s0[1] = q1; m.s1[1] = q2;
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int tile = t0 + t;
                m.s0[t] = 0; m.s1[t] = 0;
                if (tile >= p.n_tiles) { m.r0[t] = 0; m.n[t] = 0; }
                else { m.r0[t] = (int)p.row_base + tile * 32; const int lim = (int)p.M - m.r0[t]; m.n[t] = lim < 32 ? lim : 32; }
            }
        }
        if (m.n[1] == 0) m.r0[1] = m.r0[0];       // (odd tile count: the second tile recomputes the first tile's rows and stores nothing)
        return m;
    };
    auto fix_meta = [&](const Meta &r) __attribute__((always_inline)) {          // wave-uniform: into scalar registers
        Meta m;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            m.r0[t] = __builtin_amdgcn_readfirstlane(r.r0[t]); m.n[t] = __builtin_amdgcn_readfirstlane(r.n[t]);
            m.s0[t] = __builtin_amdgcn_readfirstlane(r.s0[t]); m.s1[t] = __builtin_amdgcn_readfirstlane(r.s1[t]);
        }
        return m;
    };
    // tables of a pair: threads [0, 192) one gather index each [tile][kind][row] (rows past a tile's end are clamped copies of its
    // last row: never stored), threads [0, 2 (SEGCAP + 1)) one segment offset each
    const int *const dummy = reinterpret_cast<const int *>(p.b);
    struct Tab { int idx, seg; };
    auto load_tables = [&](const Meta &m) __attribute__((always_inline)) {
        Tab tb;
        const int *ix0 = p.src[0].idx, *ix1 = ADDS ? p.add[0].idx : nullptr, *ix2 = ADDS ? p.add[1].idx : nullptr;
        const int tt = tid < 192 ? tid : 0;
        const int t = tt / 96, k = (tt % 96) >> 5, r = tt & 31;
        const int nn = m.n[t] > 0 ? m.n[t] : m.n[0];
        const int gr = m.r0[t] + (r < nn ? r : nn - 1);
        const int *ix = (k == 0) ? ix0 : (k == 1 ? ix1 : ix2);
        const int v = *(ix ? ix + gr : dummy);
        tb.idx = ix ? v : gr;
        tb.seg = 0;
        if (AGG) {
            const bool is_seg = tid < 2 * (SEGCAP + 1);
            const int ts = is_seg && tid >= SEGCAP + 1 ? 1 : 0;
            const int jj = is_seg ? tid - ts * (SEGCAP + 1) : 0;
            int sg = m.s0[ts] + jj;
            if (sg > m.s1[ts]) sg = m.s1[ts];
            tb.seg = p.seg_off[sg];
        }
        return tb;
    };
    auto store_tables = [&](const Tab &tb, int it) __attribute__((always_inline)) {
        if (tid < 192) sIdx[it & 1][tid] = tb.idx;
        if (AGG && tid < 2 * (SEGCAP + 1)) sSeg[it & 3][tid] = tb.seg;
    };
    // input rows of the weighted block (park layout: four 16-byte loads per lane, 8 lanes = one 128-byte line) and additive rows
    // (accumulator layout: row j, four runs of four features) of a tile whose indices are in sIdx[ring]
    auto gather_x = [&](const Meta &m, int ring, int t, f32x4 (&xt)[4]) __attribute__((always_inline)) {
        if ((G4C_W4_ABLATE & 64) && ring >= 0) { for (int i = 0; i < 4; ++i) xt[i] = f32x4{1.f, 2.f, 3.f, 4.f}; return; }
        const int nn = m.n[t] > 0 ? m.n[t] : m.n[0];
        const int gr = DIRECT ? m.r0[t] + (pr < nn ? pr : nn - 1) : sIdx[ring][t * 96 + pr];
        const float *rp = p.src[0].ptr + (long long)gr * p.src[0].ld + p.src[0].col0 + 4 * c8;
#pragma unroll
        for (int i = 0; i < 4; ++i) xt[i] = *reinterpret_cast<const f32x4 *>(rp + 32 * i);
    };
    auto gather_add = [&](int t, int ring, int a, f32x4 (&ad)[4]) __attribute__((always_inline)) {
        if ((G4C_W4_ABLATE & 64) && ADDS) { for (int q = 0; q < 4; ++q) ad[q] = f32x4{1.f, 2.f, 3.f, 4.f}; return; }
        if (ADDS) {
            const float *rp = p.add[a].ptr + (long long)sIdx[ring][t * 96 + 32 * (a + 1) + j] * p.add[a].ld + fbase;
#pragma unroll
            for (int q = 0; q < 4; ++q) ad[q] = *reinterpret_cast<const f32x4 *>(rp + 8 * q);
        }
    };

    // ---- this wave's slice of all layers' weights: 32 output features x 128 k x 2 planes per layer, stationary for the launch.
    // A operand of v_mfma_f32_32x32x16: lane (j, kb) holds W[feature 32 wave + j][k = 16 s + 8 kb .. + 7] — in the packed stream
    // (pack_layer_bx6_kernel: [column tile][16-k step][plane][(k / 8 % 2) * 32 + feature % 32][k % 8]) one 16-byte piece per (s, plane)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, 0x7fffffff, 0x00020000);
    const unsigned lo_b = 2u * (unsigned)(wave * 8 * STEP6 + lane * 8);
    u32x4v W[NL][8][2];
    f16_range_mode();
    RangeV rng;

    Meta m0 = fix_meta(load_meta(p_begin)), m1 = fix_meta(load_meta(p_begin + 1)), m2 = fix_meta(load_meta(p_begin + 2));
    {
        const Tab v0 = load_tables(m0), v1 = load_tables(m1);
        store_tables(v0, 0);
        store_tables(v1, 1);
    }
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                W[l][s][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, lo_b + 1024u * pl, (unsigned)l * 2u * BLOCK6 + (unsigned)s * 2u * STEP6, 0);
    for (int i = tid; i < NL * NP; i += NT) sBias[i] = p.b[i];
    sGB[tid] = p.gamma ? (tid < NP ? p.gamma[tid] : p.beta[tid - NP]) : 0.f;
    __syncthreads();

    const bool pact = p.src[0].pre_act != 0;
    __bf16 *const sA = sP, *const sB = sP + TILE_H;
    float *const fA = sF, *const fB = sF + FIN;
    Other oA, oB;       // what to do FOR tile A / FOR tile B while the other multiplies
    {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int acc_off = j * PS + 8 * ((4 * wave + q) ^ (j & 15)) + 4 * kb;                  // features fbase + 8 q .. + 3 of row j
            const int park_off = pr * PS + 8 * (((c8 >> 1) + 4 * q) ^ (pr & 15)) + 4 * (c8 & 1);      // columns 4 c8 + 32 q .. + 3 of row pr
            oA.plane_acc[q] = sA + acc_off; oA.plane_park[q] = sA + park_off;
            oB.plane_acc[q] = sB + acc_off; oB.plane_park[q] = sB + park_off;
        }
        oA.fin = fA + j * HS + fbase; oB.fin = fB + j * HS + fbase;
    }
    const __bf16 *paA[8], *paB[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) { paA[s] = sA + j * PS + 8 * ((2 * s + kb) ^ (j & 15)); paB[s] = paA[s] + TILE_H; }

    f32x16 accA, accB, accA1, accB1, accA2, accB2;
    // a layer's accumulator starts at its bias (the two 2^-11 streams start at the MFMA's inline zero)
    auto bias_init = [&](f32x16 &acc, int l) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(sBias + l * NP + fbase + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * q + e] = b4[e];
        }
    };
    // start values of a tile = bias + additive rows (in this order: what the tile kernels add)
    auto start_values = [&](f32x16 &acc, const f32x4 (&a0)[4], const f32x4 (&a1)[4]) __attribute__((always_inline)) {
        bias_init(acc, 0);
        if (ADDS) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * q + e] = (acc[4 * q + e] + a0[q][e]) + a1[q][e];
        }
    };

    // Loop-carried: xB / adB0, adB1 = input rows / additive rows of the CURRENT pair's tile B (gathered during the previous pair),
    // accA = tile A's start values, tile A's rows in its planes.
    f32x4 xB[4], adB0[4], adB1[4];
    {
        f32x4 xA[4], adA0[4], adA1[4];
        gather_x(m0, 0, 0, xA);
        gather_add(0, 0, 0, adA0); gather_add(0, 0, 1, adA1);
        gather_x(m0, 0, 1, xB);
        gather_add(1, 0, 0, adB0); gather_add(1, 0, 1, adB1);
        if (pact) other_all<2, true>(accA, accA1, accA2, xA, oA, rng);
        else other_all<2, false>(accA, accA1, accA2, xA, oA, rng);
        start_values(accA, adA0, adA1);
    }
    __syncthreads();                                       // tile A's planes of the first pair visible

    for (int it = 0, pair = p_begin; pair < p_end; ++pair, ++it) {
        W4_STAMP(0);
        // ---- tables two pairs ahead (their meta was loaded an iteration ago), meta three pairs ahead
        const Meta m3raw = load_meta(pair + 3);
        const Tab tv = load_tables(m2);
        const int nring = (it + 1) & 1;
        // ---- the next pair's rows: issued a few at a time in front of the matrix phases (a CU's share of the HBM bandwidth is ~13
        // bytes per clock: a burst of 24 loads per lane blocks the memory pipeline's queue, mlp_ws.hip), consumed in issue order.
        // (the index ring slot of the next pair was written in the previous tail: the first barrier of this iteration publishes it)
        f32x4 nxA[4], nadA0[4], nadA1[4], nxB[4], nadB0[4], nadB1[4];
        W4_STAMP(1);
        if (pact) m_block<2, true>(paA, W[0], accA, accA1, accA2, accA, accA1, accA2, xB, oB, rng);                 // for B: park
        else m_block<2, false>(paA, W[0], accA, accA1, accA2, accA, accA1, accA2, xB, oB, rng);
        start_values(accB, adB0, adB1);
        __syncthreads();
        W4_STAMP(2);
        gather_x(m1, nring, 0, nxA);
        m_block<1>(paB, W[0], accB, accB1, accB2, accA, accA1, accA2, xB, oA, rng);                                 // for A: epilogue of layer 0
        bias_init(accA, 1);
        __syncthreads();
        W4_STAMP(3);
        gather_add(0, nring, 0, nadA0);
        m_block<1>(paA, W[1], accA, accA1, accA2, accB, accB1, accB2, xB, oB, rng);                                 // for B: epilogue of layer 0
        bias_init(accB, 1);
        __syncthreads();
        W4_STAMP(4);
        gather_add(0, nring, 1, nadA1);
        if constexpr (NL == 3) {
            m_block<1>(paB, W[1], accB, accB1, accB2, accA, accA1, accA2, xB, oA, rng);                             // for A: epilogue of layer 1
            bias_init(accA, 2);
            __syncthreads();
            W4_STAMP(5);
            gather_x(m1, nring, 1, nxB);
            m_block<1>(paA, W[2], accA, accA1, accA2, accB, accB1, accB2, xB, oB, rng);                             // for B: epilogue of layer 1
            bias_init(accB, 2);
            __syncthreads();
            W4_STAMP(6);
        } else {
            gather_x(m1, nring, 1, nxB);
        }
        gather_add(1, nring, 0, nadB0);
        // for A: last layer's fp32 rows — and the NEXT pair's tile A parked into A's planes (their last readers, M(A, NL - 1), are
        // behind the previous barrier)
        if (pact) m_block<4, true>(paB, W[NL - 1], accB, accB1, accB2, accA, accA1, accA2, nxA, oA, rng);
        else m_block<4, false>(paB, W[NL - 1], accB, accB1, accB2, accA, accA1, accA2, nxA, oA, rng);
        other_all<3, false>(accB, accB1, accB2, nxA, oB, rng);                                       // B's last layer -> fp32 rows
        __syncthreads();
        W4_STAMP(7);
        gather_add(1, nring, 1, nadB1);
        start_values(accA, nadA0, nadA1);             // the next pair's tile A
        // the tables fetched at the top of this iteration go to the ring slot of the pair whose rows were gathered in the previous
        // iteration; the next iteration's first barrier publishes them before anybody reads that slot
        store_tables(tv, it + 2);
        W4_STAMP(8);

        // ---- tail of this pair: LayerNorm / activation of both tiles.  16 lanes per row (8 columns each: [4 n, 4 n + 4) and
        // [64 + 4 n, 64 + 4 n + 4), so that each of the two 16-byte stores of a row's 16 lanes writes 256 contiguous bytes), the row
        // sums reduced inside a 16-lane DPP row; a wave takes 4 rows per pass, the workgroup 16: a lane owns FOUR rows — two halves of
        // two tiles — and works on them stage by stage (four independent dependency chains: with one wave per SIMD nothing else hides
        // the latency of a DPP step or of the rsqrt).  The finished rows are stored straight from the registers; only the aggregation
        // needs them back in LDS.
        {
            const int n16 = lane & 15, g4 = lane >> 4;
            const int cq[2] = {n16 * 4, 64 + n16 * 4};
            // row set r = 2 hf + t: rows hf * 16 + wave * 4 + g4 of tile t
            const int rowi[2] = {wave * 4 + g4, 16 + wave * 4 + g4};
            float x[4][8];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 8; c += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(((r & 1) ? fB : fA) + rowi[r >> 1] * HS + cq[c >> 2]);
                    x[r][c] = v[0]; x[r][c + 1] = v[1]; x[r][c + 2] = v[2]; x[r][c + 3] = v[3];
                }
            if (p.gamma) {
                float sum[4], mean[4], var[4], rstd[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sum[r] = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) sum[r] += x[r][c];
                }
                row16_sum4(sum);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    mean[r] = sum[r] * (1.0f / NP);
                    var[r] = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) { const float dl = x[r][c] - mean[r]; var[r] += dl * dl; }
                }
                row16_sum4(var);
#pragma unroll
                for (int r = 0; r < 4; ++r) rstd[r] = rsqrtf(var[r] * (1.0f / NP) + p.eps);
#pragma unroll
                for (int c = 0; c < 8; c += 4) {
                    const f32x4 g4v = *reinterpret_cast<const f32x4 *>(sGB + cq[c >> 2]), b4 = *reinterpret_cast<const f32x4 *>(sGB + NP + cq[c >> 2]);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int u = 0; u < 4; ++u) x[r][c + u] = fmaf((x[r][c + u] - mean[r]) * rstd[r], g4v[u], b4[u]);
                }
            }
            if (p.act == G4C_ACT_SELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 8; ++c) x[r][c] = g4c::selu_f(x[r][c]);
            } else if (p.act == G4C_ACT_TANH) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 8; ++c) x[r][c] = g4c::tanh_f(x[r][c]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = r & 1, row = rowi[r >> 1];
                float *const rowp = (t ? fB : fA) + row * HS;
                f32x4 v0, v1;
                v0[0] = x[r][0]; v0[1] = x[r][1]; v0[2] = x[r][2]; v0[3] = x[r][3]; v1[0] = x[r][4]; v1[1] = x[r][5]; v1[2] = x[r][6]; v1[3] = x[r][7];
                if (AGG) { *reinterpret_cast<f32x4 *>(rowp + cq[0]) = v0; *reinterpret_cast<f32x4 *>(rowp + cq[1]) = v1; }
                if (p.out && row < m0.n[t]) {
                    const long long orow = (!AGG && p.out_idx) ? p.out_idx[m0.r0[t] + row] : m0.r0[t] + row;
                    float *op = p.out + orow * p.out_ld;
                    *reinterpret_cast<f32x4 *>(op + cq[0]) = v0; *reinterpret_cast<f32x4 *>(op + cq[1]) = v1;
                }
            }
        }
        W4_STAMP(9);
        if (AGG) {
            __syncthreads();
            W4_STAMP(15);
            // aggregation of the targets whose messages the tiles hold (rows in CSR order): the rows of a segment are added in order
            // (clamped loads, predicated adds) and divided by max(count, 1) like segment_reduce_kernel does, so the result is
            // bit-identical to the separate launch.  16 lanes per target (two 16-byte pieces each: columns [4 n, 4 n + 4) and
            // [64 + 4 n, 64 + 4 n + 4) — each store instruction writes 256 contiguous bytes of a target's row), 16 targets per pass
            // over both tiles: the ~11 targets of a pair of in-degree-6 tiles take ONE pass, spread over all four waves.
            const int n16 = lane & 15;
            const int ca = n16 * 4, cb = 64 + n16 * 4;
            const int *sg_tab = sSeg[it & 3];
            const int nsA = m0.s1[0] - m0.s0[0], nsB = m0.s1[1] - m0.s0[1];
            auto reduce_rows = [&](const float *sH, int b, int e, int sg) __attribute__((always_inline)) {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                for (int r0 = b; r0 < e; r0 += 8) {
                    f32x4 v[8], w[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float *rp = sH + (r0 + u < e ? r0 + u : e - 1) * HS;
                        v[u] = *reinterpret_cast<const f32x4 *>(rp + ca);
                        w[u] = *reinterpret_cast<const f32x4 *>(rp + cb);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const bool on = r0 + u < e;
#pragma unroll
                        for (int el = 0; el < 4; ++el) { a0[el] += on ? v[u][el] : 0.f; a1[el] += on ? w[u][el] : 0.f; }
                    }
                }
                if (p.agg_mean) {
                    const float cnt = (float)((e - b) > 1 ? (e - b) : 1);
#pragma unroll
                    for (int el = 0; el < 4; ++el) { a0[el] /= cnt; a1[el] /= cnt; }
                }
                float *op = p.agg + (long long)sg * p.agg_ld;
                *reinterpret_cast<f32x4 *>(op + ca) = a0;
                *reinterpret_cast<f32x4 *>(op + cb) = a1;
            };
            for (int q = tid >> 4; q < nsA + nsB; q += NT / 16) {
                const int t = q >= nsA ? 1 : 0, jj = q - (t ? nsA : 0);
                if (jj < SEGCAP) {
                    const int b = sg_tab[t * (SEGCAP + 1) + jj] - m0.r0[t], e = sg_tab[t * (SEGCAP + 1) + jj + 1] - m0.r0[t];
                    reduce_rows(t ? fB : fA, b, e, m0.s0[t] + jj);
                }
            }
            if (nsA > SEGCAP || nsB > SEGCAP) {          // (a tile with a long run of empty segments: their offsets from global memory)
                for (int q = tid >> 4; q < nsA + nsB; q += NT / 16) {
                    const int t = q >= nsA ? 1 : 0, jj = q - (t ? nsA : 0);
                    if (jj >= SEGCAP) {
                        const int sg = m0.s0[t] + jj;
                        reduce_rows(t ? fB : fA, p.seg_off[sg] - m0.r0[t], p.seg_off[sg + 1] - m0.r0[t], sg);
                    }
                }
            }
        }
        W4_STAMP(10);
        // (the next pair's first matrix phase writes tile B's planes and reads tile A's — neither is touched by this tail —, its
        // second one overwrites nothing the stragglers of this tail still read: the fp32 rows are rewritten in its LAST phase, behind
        // five barriers)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xB[i] = nxB[i];
            if (ADDS) { adB0[i] = nadB0[i]; adB1[i] = nadB1[i]; }
        }
        m0 = m1; m1 = m2; m2 = fix_meta(m3raw);
    }
    W4_STAMP_ONCE(13, __builtin_readcyclecounter());
    rng.m *= F16_LO_UNSCALE;          // (tracked in units of 2^-11)
    range_report(p, rng);
}

}  // namespace

namespace g4cm {

// 0 off, 1 (default; environment G4C_W4) launches of at least G4C_W4_MIN_ROWS rows, 2 every launch it can take (tests)
static int g_w4 = -1;
int w4_enable(int on) {
    if (g_w4 < 0) g_w4 = getenv("G4C_W4") ? atoi(getenv("G4C_W4")) : 1;
    const int old = g_w4;
    if (on >= 0) g_w4 = on > 2 ? 2 : on;
    return old;
}

bool w4_eligible(const Params &p, bool round1, bool agg, bool save, bool f16x2, long long row_count) {
    static const long long min_env = getenv("G4C_W4_MIN_ROWS") ? atoll(getenv("G4C_W4_MIN_ROWS")) : -1;
    const long long min_rows = min_env >= 0 ? min_env : 200000;
    const int mode = w4_enable(-1);
    if (!mode || save || round1 || !f16x2) return false;
    if (mode == 1 && row_count < min_rows) return false;
    if (p.n_src != 1 || p.n_nar != 0 || (p.n_add != 0 && p.n_add != 2) || p.n_heads) return false;
    if ((p.n_layers != 3 && p.n_layers != 2) || p.n_out != NP || p.resid || p.out_bf16) return false;
    if (p.out_idx && (agg || !p.out)) return false;          // (scattered output rows: the plain launch only)
    const Src &s = p.src[0];
    if (s.width != NP || !s.vec || s.seg_off || s.bf16) return false;
    for (int a = 0; a < p.n_add; ++a)
        if (p.add[a].width != NP || (p.add[a].ld & 3) || ((uintptr_t)p.add[a].ptr & 15)) return false;
    if (p.out && ((p.out_ld & 3) || ((uintptr_t)p.out & 15))) return false;
    if (p.gamma && (((uintptr_t)p.gamma & 15) || ((uintptr_t)p.beta & 15))) return false;
    if (((uintptr_t)p.b & 15)) return false;
    if (p.M >= (1LL << 31)) return false;
    return true;
}

int w4_launch(const Params &p, bool agg, hipStream_t st) {
    const int n_pairs = (p.n_tiles + 1) / 2;
    if (n_pairs == 0) return G4C_OK;
    const int n_cu = g4c::cu_count();
    const dim3 grid(n_pairs < n_cu ? n_pairs : n_cu), blk(NT);
    const bool direct = p.src[0].idx == nullptr, adds = p.n_add == 2, two = p.n_layers == 2;
#define G4C_W4_GO(AGG, DIRECT, ADDS, NL) mlp_w4_kernel<AGG, DIRECT, ADDS, NL><<<grid, blk, 0, st>>>(p, n_pairs)
#define G4C_W4_SHAPE(AGG, DIRECT)                                                                    \
    do {                                                                                             \
        if (two) { if (adds) G4C_W4_GO(AGG, DIRECT, true, 2); else G4C_W4_GO(AGG, DIRECT, false, 2); }      \
        else { if (adds) G4C_W4_GO(AGG, DIRECT, true, 3); else G4C_W4_GO(AGG, DIRECT, false, 3); }          \
    } while (0)
    if (agg) { if (direct) G4C_W4_SHAPE(true, true); else G4C_W4_SHAPE(true, false); }
    else { if (direct) G4C_W4_SHAPE(false, true); else G4C_W4_SHAPE(false, false); }
#undef G4C_W4_SHAPE
#undef G4C_W4_GO
    return g4c::check_launch("g4c_mlp_forward (w4)");
}

}  // namespace g4cm

extern "C" int g4c_mlp_w4_enable(int on) { return g4cm::w4_enable(on); }
