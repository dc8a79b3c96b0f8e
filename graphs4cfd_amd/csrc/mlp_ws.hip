// Weight-stationary persistent form of the fused split-operand MLP ("ws") for the MP layers' message launch — the same arithmetic as
// mlp_bx6_kernel<.., SP = 2> (mlp_fused.hip): gather -> [SELU on load] -> Linear/SELU chain as two-way fp16 split products on
// v_mfma_f32_16x16x32_f16 -> LayerNorm -> activation -> store (-> per-target aggregation).  Replaces MLP.forward
// (graphs4cfd/nn/blocks.py:117-144) with the torch.cat / index ops in front of it (nn/blocks.py:181,328) and, with AGG, the
// scatter(e', col, reduce) behind it (nn/blocks.py:183,330).
//
// What is different from the tile kernels:
//   * ONE 8-wave workgroup per CU (two waves per SIMD, 256 VGPRs each), persistent over a contiguous range of tile pairs;
//   * wave w owns output features [16 w, 16 w + 16) of EVERY layer and keeps its slice of ALL THREE layers' weights in registers
//     (3 layers x 4 k-steps x 2 planes x 16 bytes per lane = 96 VGPRs) for the whole launch: no weight stream at all after the
//     first tile, and — because the weights are loop-invariant — the pair loop is an ordinary loop (the single-layer stationary form
//     of mlp_bx6i_kernel redefines its weight registers per layer, which hipcc duplicates across a back edge);
//   * the loop is software-pipelined over pairs: gather indices and segment offsets are fetched two pairs ahead (into LDS), input rows
//     and additive rows one pair ahead (into registers: tile A's in the matrix phases, tile B's — needed a phase later — in the
//     tail, so that they are not live across the phases), the next pair's first tile is parked inside the current pair's last
//     matrix phase, so no dependent memory round trip is left on a pair's critical path;
//   * inside a pair the two tiles alternate layer by layer as in mlp_bx6i_kernel: while a wave issues the MFMAs of one tile its
//     vector ALUs run the other tile's epilogue (bias is the accumulator start value; SELU, fp16 split, planes).
// Envelope (everything else keeps mlp_bx6_kernel / mlp_bx6i_kernel): f16x3 stream (SP = 2) or the rounded-bf16 mode (SP = 1: one
// v_mfma_f32_16x16x32_bf16 per multiply-add on the leading plane of the bf16x6 stream, operands rounded to bf16 exactly where
// mlp_bx6_kernel<.., SP = 1> rounds them; there the rows of the weighted block may be bf16 and the output rows may be stored as bf16 /
// bf16(SELU) — g4c_mlp_forward_bf16_agg), ONE weighted 128-wide aligned input block (rows direct or through an index, optional SELU on
// load), 0 or 2 additive 128-wide blocks (SP = 1: 2), two or three layers, 128-wide output rows without residual / heads; an output
// index only without the fused aggregation.
#include "mlp_common.h"
#include <cstdlib>
using namespace g4cm;

#ifdef G4C_WS_TIMING
__device__ unsigned long long g4c_ws_stamps[256 * 32];
extern "C" int g4c_ws_read_stamps(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g4c_ws_stamps), sizeof(unsigned long long) * n);
}
#define WS_STAMP(k) do { if (it == 1 && tid == 0 && blockIdx.x < 256) g4c_ws_stamps[blockIdx.x * 32 + (k)] = __builtin_readcyclecounter(); } while (0)
// inside the tail lambdas (`its` = the iteration they were called from); the stamp's own s_waitcnt lgkmcnt(0) makes it a point where
// every LDS read issued before it has landed
#define WS_STAMP_T(k) do { if (its == 1 && tid == 0 && blockIdx.x < 256) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g4c_ws_stamps[blockIdx.x * 32 + (k)] = __builtin_readcyclecounter(); } } while (0)
#define WS_STAMP_ONCE(k, v) do { if (tid == 0 && blockIdx.x < 256) g4c_ws_stamps[blockIdx.x * 32 + (k)] = (v); } while (0)      // 12 kernel start, 13 end, 14 pairs
#else
#define WS_STAMP(k) do {} while (0)
#define WS_STAMP_T(k) do {} while (0)
#define WS_STAMP_ONCE(k, v) do {} while (0)
#endif

// timing-only ablations (wrong results; scripts/build_ws_timing.sh <suffix> -DG4C_WS_ABLATE=<bits>, scripts/ws_stamps.py): 2 no MFMAs,
// 4 no B-fragment reads beyond the first two slices, 16 no SELU (identity), 32 no fp16 split (l = h).  (Removing the plane WRITES is
// not a valid ablation: hipcc then treats the never-written planes as undefined and drops half of the MFMAs with them.)
#ifndef G4C_WS_ABLATE
#define G4C_WS_ABLATE 0
#endif

namespace {

// Operand planes: [32 rows][128 k] fp16, NO padding; the 16-byte granule c of row r is stored at granule c ^ (r & 15).  The B
// fragment reads of v_mfma_f32_16x16x32 (lane (n, g): row n, granule 4 ks + g; a ds_read_b128 is served in groups of 16 lanes that
// mix two values of g) are then conflict-free: within a group the granules (4 ks + g) ^ n are all different.  With the padded
// [32][136] layout of the tile kernels 45 % of this kernel's LDS cycles were bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
constexpr int PS = 128;                 // row stride of a plane (elements)
constexpr int PLN = 32 * PS;            // elements of one operand plane of a 32-row tile
constexpr int TILE_BF16 = 2 * PLN;      // two planes (h, l * 2^11)
constexpr int FIN = 32 * HS;            // floats of a tile's fp32 final rows [32][132]
constexpr int SEGCAP = 64;              // segment offsets of a tile staged in LDS (more segments: read from global memory)

template <int SP>
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (SP == 1) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// rows [r0, r0 + n) and segments [s0, s1) of the two tiles of a pair (wave-uniform)
struct Meta { int r0[2], n[2], s0[2], s1[2]; };

// What the vector ALUs do for the OTHER tile while this tile's MFMAs issue, one piece per slice s = 0..7 (unit u = s / 4 = sample
// row block / row half, piece s % 4):
//   EK 0 nothing;  1 hidden-layer epilogue of accE[u] (sample rows 16 u + n);  2 park rows prow + 16 u of the gathered input
//   (PACT: SELU pending on the stored rows);  3 last layer: fp32 rows of accE[u] into the tile's final buffer;  4 (m_block) = 3, then 2
//   with the next pair's rows.
// EK 1 / 2 work on one PAIR of values at a time: pieces 0 / 2 = [fold,] SELU of pair 0 / 1, pieces 1 / 3 = fp16 split + plane writes.
struct Other {
    __bf16 *plane_acc;        // EK 1: this lane's element (row n, feature fcol) of the other tile's planes (swizzled address)
    __bf16 *plane_park;       // EK 2: (row prow, column pc) (swizzled address)
    float *fin;               // EK 3: (row n, feature fcol) of the other tile's fp32 rows
};

// LOADED: x comes straight from memory — fmaxf would first canonicalise it (v_max_f32 x, x: one more instruction per value), the
// instruction itself does not need that (same value: g4c::selu_f's formula)
template <bool LOADED = false>
__device__ __forceinline__ f32x2 selu2w(f32x2 x) {
    if (G4C_WS_ABLATE & 16) return x;
    if constexpr (!LOADED) return selu2(x);
    const float sa = 1.6732632423543772848170429916717f * 1.0507009873554804934193349852946f;
    const float scale = 1.0507009873554804934193349852946f;
    f32x2 r;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float ex = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(x[e] * 1.4426950408889634f), 0.f, 1.f);
        float m;
        asm("v_max_f32 %0, 0, %1" : "=v"(m) : "v"(x[e]));
        r[e] = fmaf(m, scale, fmaf(sa, ex, -sa));
    }
    return r;
}

// G4C_WS_SCALED (f16x3 stream): the values that are about to be split are carried as S = y * 2^11 — the SELU forms them with its two
// FMAs' constants scaled (the same bits as y * 2^11: a power of two commutes with both roundings), a row that is parked without an
// activation is multiplied once, as before — and the split reads S twice: h = fp16(S * 2^-11) by v_fma_mixlo/hi_f16 (the product is
// exactly y, one rounding: the bits of v_cvt_pk_f16_f32), l = fp16(fma(h, -2^11, S)) as before.  One vector instruction less per pair
// (no y * 2^11), the same planes bit for bit; the range tracker compares S with 65504 * 2^11.
#ifndef G4C_WS_SCALED
#define G4C_WS_SCALED 1
#endif
// the fused aggregation's mean by g4c::mean_div4 (g4c_common.h: shared reciprocal + one correction per value, the IEEE quotient bit
// for bit) instead of four divisions: 23 vector instructions per target and lane instead of 47
#ifndef G4C_WS_MEAN_DIV
#define G4C_WS_MEAN_DIV 1
#endif


template <bool LOADED = false>
__device__ __forceinline__ f32x2 selu2w_scaled(f32x2 x) {
    const float sa = 1.6732632423543772848170429916717f * 1.0507009873554804934193349852946f * F16_LO_SCALE;
    const float scale = 1.0507009873554804934193349852946f * F16_LO_SCALE;
    f32x2 r;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float ex = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(x[e] * 1.4426950408889634f), 0.f, 1.f);
        float m;
        if constexpr (LOADED) asm("v_max_f32 %0, 0, %1" : "=v"(m) : "v"(x[e]));
        else m = fmaxf(x[e], 0.f);
        r[e] = fmaf(m, scale, fmaf(sa, ex, -sa));
    }
    return r;
}
__device__ __forceinline__ void put_pair_scaled(__bf16 *d, f32x2 S, RangeV &rng) {
    rng.m = fmaxf(fmaxf(rng.m, fabsf(S[0])), fabsf(S[1]));          // (v_max3_f32; in units of 2^-11: range_report_scaled)
    unsigned hu, lu;
    const float up = F16_LO_UNSCALE, dn = -F16_LO_SCALE;
    // (one statement: hipcc pads an s_nop behind every asm statement whose output the next instruction reads — three per pair before)
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %0, %5, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %0, %5, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(hu), "=&v"(lu) : "v"(S[0]), "v"(S[1]), "s"(up), "s"(dn));
    *reinterpret_cast<unsigned *>(d) = hu;
    *reinterpret_cast<unsigned *>(d + PLN) = lu;
}

// two-way fp16 split of a pair -> one packed pair per plane (split_pair_f16, mlp_common.h: four vector instructions + the range tracker)
// (SP = 1: the pair rounded to bf16, one plane)
template <int SP>
__device__ __forceinline__ void put_pair(__bf16 *d, f32x2 y, RangeV &rng) {
    if constexpr (SP == 1) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        bf16x2 b;
        b[0] = (__bf16)y[0]; b[1] = (__bf16)y[1];
        *reinterpret_cast<bf16x2 *>(d) = b;
    } else {
        unsigned hu, lu;
        split_pair_f16(y, hu, lu, rng);
        if (G4C_WS_ABLATE & 32) lu = hu;
        *reinterpret_cast<unsigned *>(d) = hu;
        *reinterpret_cast<unsigned *>(d + PLN) = lu;
    }
}

// sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));    // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));    // row_mirror
    return v;
}

// (two independent sums at once: each step's DPP adds alternate, so neither waits for its own previous step)
__device__ __forceinline__ void row16_sum2(float &a, float &b) {
#define G4C_DPP_ADD(v, ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false))
    G4C_DPP_ADD(a, 0xB1); G4C_DPP_ADD(b, 0xB1);
    G4C_DPP_ADD(a, 0x4E); G4C_DPP_ADD(b, 0x4E);
    G4C_DPP_ADD(a, 0x141); G4C_DPP_ADD(b, 0x141);
    G4C_DPP_ADD(a, 0x140); G4C_DPP_ADD(b, 0x140);
#undef G4C_DPP_ADD
}

template <int SP, int EK, bool PACT>
__device__ __forceinline__ void other_piece(int s, const f32x4 (&accE)[2], const f32x4 (&accE1)[2], const f32x4 (&xe)[2], const Other &o, f32x2 &hold, RangeV &rng) {
    const int u = s >> 2, pc4 = s & 3, pr = pc4 >> 1;
    if (EK == 1) {
        if ((pc4 & 1) == 0) {
            f32x2 x, x1;
            x[0] = accE[u][2 * pr]; x[1] = accE[u][2 * pr + 1];
            x1[0] = accE1[u][2 * pr]; x1[1] = accE1[u][2 * pr + 1];
            if (SP == 2 && G4C_WS_SCALED && !(G4C_WS_ABLATE & 48)) hold = selu2w_scaled(x1 * F16_LO_UNSCALE + x);
            else hold = SP == 1 ? selu2w(x) : selu2w(x1 * F16_LO_UNSCALE + x);         // (the fold is one v_pk_fma_f32)
        } else {
            if (SP == 2 && G4C_WS_SCALED && !(G4C_WS_ABLATE & 48)) put_pair_scaled(o.plane_acc + u * 16 * PS + 2 * pr, hold, rng);
            else put_pair<SP>(o.plane_acc + u * 16 * PS + 2 * pr, hold, rng);
        }
    } else if (EK == 2) {
        if ((pc4 & 1) == 0) {
            f32x2 x;
            x[0] = xe[u][2 * pr]; x[1] = xe[u][2 * pr + 1];
            if (SP == 2 && G4C_WS_SCALED && !(G4C_WS_ABLATE & 48)) hold = PACT ? selu2w_scaled<true>(x) : x * F16_LO_SCALE;
            else hold = PACT ? selu2w<true>(x) : x;
        } else {
            if (SP == 2 && G4C_WS_SCALED && !(G4C_WS_ABLATE & 48)) put_pair_scaled(o.plane_park + u * 16 * PS + 2 * pr, hold, rng);
            else put_pair<SP>(o.plane_park + u * 16 * PS + 2 * pr, hold, rng);
        }
    } else if (EK == 3) {
        if (pc4 == 0) {
            f32x4 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = SP == 1 ? accE[u][e] : fmaf(accE1[u][e], F16_LO_UNSCALE, accE[u][e]);
            *reinterpret_cast<f32x4 *>(o.fin + u * 16 * HS) = x;
        }
    }
}

// vector instructions scheduled behind each MFMA of a slice (measured: 3, 4, 6 and MFMAs in bursts of 3 / 6 with the vector work behind
// them all within 3 % of each other — DESIGN.md 4.1 "What bounds it")
constexpr int WS_VALU_PER_MFMA = 4;
// slices the B fragments are fetched ahead of their MFMAs (3 / 4: + 8 / 16 registers, scratch, slower: HISTORY.md 4.1)
constexpr int WS_FRAG_AHEAD = 2;
// One 128-k block for one tile: acc += W(layer) x planes, 8 slices (k-step ks = s / 2, sample row block rb = s % 2) of three
// products each: (Wh, xl) and (Wl, xh) into acc1 (the 2^-11 terms), (Wh, xh) into acc.  pa[ks]: this lane's B-operand address
// (row n, granule (4 ks + g) ^ n) in the tile's h plane.
// (SP = 1: the one product (W, x) of the leading planes, no acc1.)
template <int SP, int EK, bool PACT = false>
__device__ __forceinline__ void m_block(const __bf16 *const (&pa)[4], const bf16x8 (&W)[4][SP], f32x4 (&acc)[2], f32x4 (&acc1)[2],
                                        const f32x4 (&accE)[2], const f32x4 (&accE1)[2], const f32x4 (&xe)[2], const Other &o, RangeV &rng) {
    // B fragments (h, l planes) of slice s: row block s % 2, k-step s / 2; fetched WS_FRAG_AHEAD slices ahead of their MFMAs
    constexpr int AH = WS_FRAG_AHEAD, RING = AH + 1;
    bf16x8 fh[RING], fl[RING];
#pragma unroll
    for (int s = 0; s < AH; ++s) {
        const __bf16 *pn = pa[s >> 1] + (s & 1) * 16 * PS;
        fh[s] = *reinterpret_cast<const bf16x8 *>(pn);
        if (SP == 2) fl[s] = *reinterpret_cast<const bf16x8 *>(pn + PLN);
    }
    f32x2 hold = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int ks = s >> 1, rb = s & 1;
        if (s + AH < 8 && !(G4C_WS_ABLATE & 4)) {
            const __bf16 *pn = pa[(s + AH) >> 1] + ((s + AH) & 1) * 16 * PS;
            fh[(s + AH) % RING] = *reinterpret_cast<const bf16x8 *>(pn);
            if (SP == 2) fl[(s + AH) % RING] = *reinterpret_cast<const bf16x8 *>(pn + PLN);
        }
        if (!EK) __builtin_amdgcn_sched_barrier(0);
        if constexpr (EK == 4) {          // last layer's fp32 rows of accE, then the NEXT pair's first tile parked into the same tile's planes
            other_piece<SP, 3, false>(s, accE, accE1, xe, o, hold, rng);
            other_piece<SP, 2, PACT>(s, accE, accE1, xe, o, hold, rng);
        } else {
            other_piece<SP, EK, PACT>(s, accE, accE1, xe, o, hold, rng);
        }
        const bf16x8 ch = fh[s % RING];
        if constexpr (SP == 1) {
            acc[rb] = mfma16<1>(W[ks][0], ch, acc[rb]);
        } else {
            const bf16x8 cl = fl[s % RING];
            if (G4C_WS_ABLATE & 2) {
                asm volatile("" :: "v"(ch), "v"(cl));
            } else {
                acc1[rb] = mfma16<2>(W[ks][0], cl, acc1[rb]);
                acc[rb] = mfma16<2>(W[ks][0], ch, acc[rb]);
                acc1[rb] = mfma16<2>(W[ks][1], ch, acc1[rb]);
            }
        }
        if (EK) {
            __builtin_amdgcn_sched_group_barrier(0x100, SP, 0);                 // DS read (fragments two slices ahead)
#pragma unroll
            for (int m = 0; m < (SP == 2 ? 3 : 1); ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, SP == 2 ? WS_VALU_PER_MFMA : 3 * WS_VALU_PER_MFMA, 0);      // VALU
            }
            __builtin_amdgcn_sched_group_barrier(0x200, SP, 0);                 // DS write
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// a whole unit outside a matrix phase (the first tile of a pair is parked with nothing to overlap with; B's last rows)
template <int SP, int EK, bool PACT>
__device__ __forceinline__ void other_all(const f32x4 (&accE)[2], const f32x4 (&accE1)[2], const f32x4 (&xe)[2], const Other &o, RangeV &rng) {
    f32x2 hold = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) other_piece<SP, EK, PACT>(s, accE, accE1, xe, o, hold, rng);
}

// ---- the node MLP on a contiguous range of rows [S0, S1) by ONE 8-wave workgroup: [x0 | q.v] -> Linear / SELU chain (weights streamed
// block by block from q.w: 32 registers per wave and block, the next block fetched while this one multiplies) -> LayerNorm ->
// activation -> q.out, + the heads.  Rows are taken 64 at a time as two 32-row tiles whose matrix phases carry each other's vector work
// (park / epilogue / fp32 rows), as in the message loop of mlp_ws_kernel; a remainder of <= 32 rows runs as one tile.  Used behind the
// message phase (mlp_ws_kernel<.., NODE>: x0 = the aggregates the workgroup has just written).  (Round 5 also ran it as a launch of
// its own, mlp_node_kernel: 131 against 108 us of the tile kernel at 100k rows — removed in round 6, HISTORY.md 4.1.)
struct NodeCtx {
    int tid, wave, n, g, fcol, prow, pc;
    unsigned lo_b;
    const float *sBiasN, *sGBN;
    float *fA, *fB;
    Other oA, oB;
    const __bf16 *paA[4], *paB[4];
};

template <int SP, int NL>
__device__ __forceinline__ void node_phase(const NodeCtx &c, const float *x0, const int x0_ld, const NodeParams &q, const int S0, const int S1) {
    const int wave = c.wave, n = c.n, g = c.g, fcol = c.fcol, prow = c.prow, pc = c.pc;
    const unsigned lo_b = c.lo_b;
    const float *const sBiasN = c.sBiasN, *const sGBN = c.sGBN;
    float *const fA = c.fA, *const fB = c.fB;
    const Other &oA = c.oA, &oB = c.oB;
    const __bf16 *const (&paA)[4] = c.paA;
    const __bf16 *const (&paB)[4] = c.paB;
    f32x4 accA[2], accB[2], accA1[2], accB1[2];
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(q.w), 0, 0x7fffffff, 0x00020000);
    RangeV rngN;
    const int n_blk = NL + 1 + q.n_heads;          // blocks of the node MLP's stream (a head block that does not exist is not fetched)
    auto ld_block = [&](bf16x8 (&Wb)[4][SP], int blk) __attribute__((always_inline)) {
        if (blk >= n_blk) return;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int pl = 0; pl < SP; ++pl) Wb[ks][pl] = ldw(rq, lo_b + 1024u * pl, (unsigned)blk * 2u * BLOCK6 + (unsigned)ks * 4u * STEP6);
    };
    auto bias_n = [&](f32x4 (&acc)[2], f32x4 (&acc1)[2], int l) __attribute__((always_inline)) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (l >= 0) b4 = *reinterpret_cast<const f32x4 *>(sBiasN + l * NP + fcol);
        acc[0] = b4; acc[1] = b4;
        acc1[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    const int rowL = wave * 4 + g;                                   // LayerNorm / row-store layout: 16 lanes per row
    const int cqn[2] = {n * 4, 64 + n * 4};
    // LayerNorm / activation of the fp32 rows of one tile (rows [rb, rb + nrows) of the launch), stored from the registers; with
    // heads the finished rows go back to the buffer (the heads' operand)
    auto finish_rows = [&](float *frows, int rb, int nrows) __attribute__((always_inline)) {
        float x[8];
        float *const rowp = frows + rowL * HS;
#pragma unroll
        for (int c = 0; c < 8; c += 4) {
            const f32x4 v4 = *reinterpret_cast<const f32x4 *>(rowp + cqn[c >> 2]);
            x[c] = v4[0]; x[c + 1] = v4[1]; x[c + 2] = v4[2]; x[c + 3] = v4[3];
        }
        if (q.gamma) {
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) sum += x[c];
            sum = row16_sum(sum);
            const float mean = sum * (1.0f / NP);
            float var = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) { const float dl = x[c] - mean; var += dl * dl; }
            var = row16_sum(var);
            const float rstd = rsqrtf(var * (1.0f / NP) + q.eps);
#pragma unroll
            for (int c = 0; c < 8; c += 4) {
                const f32x4 g4 = *reinterpret_cast<const f32x4 *>(sGBN + cqn[c >> 2]), b4 = *reinterpret_cast<const f32x4 *>(sGBN + NP + cqn[c >> 2]);
#pragma unroll
                for (int u = 0; u < 4; ++u) x[c + u] = fmaf((x[c + u] - mean) * rstd, g4[u], b4[u]);
            }
        }
        if (q.act == G4C_ACT_SELU) {
#pragma unroll
            for (int c = 0; c < 8; ++c) x[c] = g4c::selu_f(x[c]);
        } else if (q.act == G4C_ACT_TANH) {
#pragma unroll
            for (int c = 0; c < 8; ++c) x[c] = g4c::tanh_f(x[c]);
        }
        f32x4 v0, v1;
        v0[0] = x[0]; v0[1] = x[1]; v0[2] = x[2]; v0[3] = x[3]; v1[0] = x[4]; v1[1] = x[5]; v1[2] = x[6]; v1[3] = x[7];
        if (q.n_heads) { *reinterpret_cast<f32x4 *>(rowp + cqn[0]) = v0; *reinterpret_cast<f32x4 *>(rowp + cqn[1]) = v1; }
        if (rowL < nrows) {
            float *op = q.out + (long long)(rb + rowL) * q.out_ld;
            *reinterpret_cast<f32x4 *>(op + cqn[0]) = v0; *reinterpret_cast<f32x4 *>(op + cqn[1]) = v1;
        }
    };
    auto store_head = [&](int hd, const float *frows, int rb, int nrows) __attribute__((always_inline)) {
        if (rowL < nrows) {
            const float *rowp = frows + rowL * HS;
            float *op = q.head_out[hd] + (long long)(rb + rowL) * q.head_ld;
            *reinterpret_cast<f32x4 *>(op + cqn[0]) = *reinterpret_cast<const f32x4 *>(rowp + cqn[0]);
            *reinterpret_cast<f32x4 *>(op + cqn[1]) = *reinterpret_cast<const f32x4 *>(rowp + cqn[1]);
        }
    };
    // rows [rb, rb + nrows) of a [., ld] tensor in the park layout (rows past the range: clamped copies)
    auto load_rows = [&](const float *base, int ld, int rb, int nrows, f32x4 (&x)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int r = prow + 16 * hh;
            x[hh] = *reinterpret_cast<const f32x4 *>(base + (long long)(rb + (r < nrows ? r : nrows - 1)) * ld + pc);
        }
    };
    // (Requesting the next step's rows and first weight blocks during this step — software pipelining over the 64-row steps — was
    // measured SLOWER: it pushes the kernel past 256 registers, 11 - 14 dwords per lane go to scratch, the level-1 node launch took
    // 137 us instead of 131 and the fused MP layer at 60k edges 62.6 us instead of 55.8: profiles/r05_node_kernel.log.)
    for (int r0 = S0; r0 < S1; r0 += 64) {
        const int nr = (S1 - r0) < 64 ? (S1 - r0) : 64;
        bf16x8 Wa[4][SP], Wb[4][SP];
        ld_block(Wa, 0); ld_block(Wb, 1);
        auto &H0 = (NL == 3) ? Wa : Wb;          // (NL == 3: head 0 = block 4 in Wa, head 1 = block 5 in Wb;  NL == 2: head 0 = block 3 in Wb, head 1 = block 4 in Wa)
        auto &H1 = (NL == 3) ? Wb : Wa;
        f32x4 ga[2], va[2], gb[2], vb[2];
        {
            const int nA = nr < 32 ? nr : 32;
            load_rows(x0, x0_ld, r0, nA, ga); load_rows(q.v, q.v_ld, r0, nA, va);
            if (nr > 32) { load_rows(x0, x0_ld, r0 + 32, nr - 32, gb); load_rows(q.v, q.v_ld, r0 + 32, nr - 32, vb); }
        }
        if (nr > 32) {
            // ================================ two tiles (A: 32 rows, B: nr - 32): every matrix phase of one tile carries the other tile's
            // vector work (park / epilogue / fp32 rows), as in the message loop; a block of weights serves both tiles
            const int nB = nr - 32;
            other_all<SP, 2, false>(accA, accA1, ga, oA, rngN);
            other_all<SP, 2, false>(accA, accA1, gb, oB, rngN);
            bias_n(accA, accA1, 0); bias_n(accB, accB1, 0);
            __syncthreads();
            m_block<SP, 0>(paA, Wa, accA, accA1, accA, accA1, ga, oA, rngN);                    // M(A, aggregate block)
            __syncthreads();
            m_block<SP, 2, false>(paB, Wa, accB, accB1, accB, accB1, va, oA, rngN);             // M(B, aggregate block); A's v rows -> A's planes
            ld_block(Wa, 2);
            __syncthreads();
            m_block<SP, 2, false>(paA, Wb, accA, accA1, accA, accA1, vb, oB, rngN);             // M(A, v block); B's v rows -> B's planes
            __syncthreads();
            m_block<SP, 1>(paB, Wb, accB, accB1, accA, accA1, va, oA, rngN);                    // M(B, v block); A: epilogue of layer 0
            ld_block(Wb, 3);
            bias_n(accA, accA1, 1);
            __syncthreads();
            m_block<SP, 1>(paA, Wa, accA, accA1, accB, accB1, va, oB, rngN);                    // M(A, layer 1); B: epilogue of layer 0
            bias_n(accB, accB1, 1);
            __syncthreads();
            if constexpr (NL == 3) {
                m_block<SP, 1>(paB, Wa, accB, accB1, accA, accA1, va, oA, rngN);                // M(B, layer 1); A: epilogue of layer 1
                ld_block(Wa, 4);
                bias_n(accA, accA1, 2);
                __syncthreads();
                m_block<SP, 1>(paA, Wb, accA, accA1, accB, accB1, va, oB, rngN);                // M(A, layer 2); B: epilogue of layer 1
                bias_n(accB, accB1, 2);
                __syncthreads();
                m_block<SP, 3>(paB, Wb, accB, accB1, accA, accA1, va, oA, rngN);                // M(B, layer 2); A: fp32 rows
                ld_block(Wb, 5);
            } else {
                m_block<SP, 3>(paB, Wa, accB, accB1, accA, accA1, va, oA, rngN);                // M(B, layer 1); A: fp32 rows
                ld_block(Wa, 4);
            }
            other_all<SP, 3, false>(accB, accB1, va, oB, rngN);                                 // B: fp32 rows
            __syncthreads();
            finish_rows(fA, r0, 32);
            finish_rows(fB, r0 + 32, nB);
            if (q.n_heads) {
                __syncthreads();
                f32x4 ha[2], hb[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    ha[hh] = *reinterpret_cast<const f32x4 *>(fA + (prow + 16 * hh) * HS + pc);
                    hb[hh] = *reinterpret_cast<const f32x4 *>(fB + (prow + 16 * hh) * HS + pc);
                }
                other_all<SP, 2, false>(accA, accA1, ha, oA, rngN);
                other_all<SP, 2, false>(accA, accA1, hb, oB, rngN);
                bias_n(accA, accA1, -1); bias_n(accB, accB1, -1);
                __syncthreads();
                m_block<SP, 0>(paA, H0, accA, accA1, accA, accA1, va, oA, rngN);                // head 0 of A
                m_block<SP, 3>(paB, H0, accB, accB1, accA, accA1, va, oA, rngN);                // head 0 of B; A's head rows -> fA
                bias_n(accA, accA1, -1);
                __syncthreads();
                store_head(0, fA, r0, 32);
                if (q.n_heads > 1) {
                    m_block<SP, 3>(paA, H1, accA, accA1, accB, accB1, va, oB, rngN);            // head 1 of A; B's head-0 rows -> fB
                    bias_n(accB, accB1, -1);
                    __syncthreads();
                    store_head(0, fB, r0 + 32, nB);
                    m_block<SP, 3>(paB, H1, accB, accB1, accA, accA1, va, oA, rngN);            // head 1 of B; A's head-1 rows -> fA
                    __syncthreads();
                    store_head(1, fA, r0, 32);
                    other_all<SP, 3, false>(accB, accB1, va, oB, rngN);
                    __syncthreads();
                    store_head(1, fB, r0 + 32, nB);
                } else {
                    other_all<SP, 3, false>(accB, accB1, va, oB, rngN);
                    __syncthreads();
                    store_head(0, fB, r0 + 32, nB);
                }
            }
            __syncthreads();          // the next tiles overwrite the planes and the fp32 rows
            continue;
        }
        // ================================ one tile (nr <= 32 rows): always the last step of the range
        other_all<SP, 2, false>(accA, accA1, ga, oA, rngN);          // the aggregate rows -> A's planes
        other_all<SP, 2, false>(accA, accA1, va, oB, rngN);          // the node rows -> B's planes
        bias_n(accA, accA1, 0);
        __syncthreads();
        m_block<SP, 0>(paA, Wa, accA, accA1, accA, accA1, ga, oA, rngN);
        ld_block(Wa, 2);
        m_block<SP, 0>(paB, Wb, accA, accA1, accA, accA1, ga, oA, rngN);
        ld_block(Wb, 3);
        __syncthreads();                                              // everybody has read both sets of planes
        other_all<SP, 1, false>(accA, accA1, ga, oA, rngN);          // layer 0's epilogue -> A's planes
        bias_n(accA, accA1, 1);
        __syncthreads();
        m_block<SP, 0>(paA, Wa, accA, accA1, accA, accA1, ga, oA, rngN);
        if constexpr (NL == 3) {
            ld_block(Wa, 4);
            other_all<SP, 1, false>(accA, accA1, ga, oB, rngN);      // layer 1's epilogue -> B's planes (nobody reads them in this phase)
            bias_n(accA, accA1, 2);
            __syncthreads();
            m_block<SP, 0>(paB, Wb, accA, accA1, accA, accA1, ga, oA, rngN);
            ld_block(Wb, 5);
        } else {
            ld_block(Wa, 4);          // (NL == 2: blocks 3, 4 are the heads — Wb holds block 3 already)
        }
        other_all<SP, 3, false>(accA, accA1, ga, oA, rngN);          // the last layer's fp32 rows -> fA
        __syncthreads();
        finish_rows(fA, r0, nr);
        if (q.n_heads) {
            // ---- heads: v' rows -> A's planes, one 128-k block per head, fp32 rows through fB / fA, whole-row stores
            __syncthreads();
            f32x4 xh[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) xh[hh] = *reinterpret_cast<const f32x4 *>(fA + (prow + 16 * hh) * HS + pc);
            other_all<SP, 2, false>(accA, accA1, xh, oA, rngN);
            bias_n(accA, accA1, -1);
            bias_n(accB, accB1, -1);
            __syncthreads();
            m_block<SP, 0>(paA, H0, accA, accA1, accA, accA1, ga, oA, rngN);
            if (q.n_heads > 1) m_block<SP, 0>(paA, H1, accB, accB1, accA, accA1, ga, oA, rngN);
            other_all<SP, 3, false>(accA, accA1, ga, oB, rngN);      // head 0 -> fB
            other_all<SP, 3, false>(accB, accB1, ga, oA, rngN);      // head 1 -> fA (its rows are in the planes)
            __syncthreads();
            store_head(0, fB, r0, nr);
            if (q.n_heads > 1) store_head(1, fA, r0, nr);
        }
        __syncthreads();          // (a step of <= 32 rows is the last one)
    }
    if (G4C_WS_SCALED && !(G4C_WS_ABLATE & 48)) rngN.m *= F16_LO_UNSCALE;
    if (q.range_flag && range_hit(rngN)) {
        if ((threadIdx.x & 63) == 0) q.range_flag[q.range_slot] = 1;
    }
}

// SP: 2 the f16x3 stream, 1 the rounded-bf16 mode;  NL: layers (2 or 3);  XB16 (SP = 1): the weighted block's rows are bf16;
// AB16 (SP = 1): the additive rows are bf16 (the first-layer products a g4c_mlp_forward_heads_bf16_out / _bf16_out launch stored:
// half the bytes of the launch's largest gather stream — REMuS-GNN's level-1 angle launch reads 2 x 2.5 M of them)
// NODE (SP = 2, with AGG): the node update of the MP layer fused behind the message launch (NodeParams, mlp_common.h): when its tile
// pairs are done a workgroup holds, in L2, the aggregates of a contiguous range of targets that no other workgroup touches — it runs
// the node MLP (same depth, weights streamed block by block: the message MLP's stationary registers are dead by then) on those
// targets, 32 at a time, and stores v' and the heads.  One launch per MP layer instead of two (three with a separate aggregation):
// small and medium levels are bound by the dependent chain of each launch, not by throughput.
// Rounded-bf16 mode (SP = 1): two waves per SIMD as well (four — 128 registers, scratch — measured slower: HISTORY.md 4.1)
constexpr int G4C_WS_SP1_MINW = 2;
// DENSE (with AGG): the launch's segments all have the same number of rows, 4 .. 8 (G4C_AGG_UNIFORM) — dense pairs, below
template <bool AGG, bool DIRECT, bool ADDS, int SP, int NL, bool XB16, bool AB16 = false, bool NODE = false, bool DENSE = false>
__global__ __launch_bounds__(512, SP == 1 ? G4C_WS_SP1_MINW : 2) void mlp_ws_kernel(const Params p, const int n_pairs, const NodeParams q) {
    static_assert((SP == 1 || SP == 2) && (NL == 2 || NL == 3) && (SP == 1 || !XB16) && (SP == 1 || !AB16) && (ADDS || !AB16) &&
                  (!NODE || (AGG && SP == 2)), "mlp_ws_kernel: unsupported instantiation");
    // an additive row piece as loaded: four fp32 values, or four bf16 values in two dwords (widened where they are added)
    typedef typename std::conditional<AB16, u32x2, f32x4>::type AddV;
    __shared__ __attribute__((aligned(16))) __bf16 sP[2 * TILE_BF16];      // operand planes of tiles A, B (34 816 B)
    __shared__ __attribute__((aligned(16))) float sF[2 * FIN];             // fp32 final rows of tiles A, B (33 792 B)
    __shared__ int sIdx[2][2 * 3 * 32];          // ring of 2: [tile][weighted block, additive 0, additive 1][row]
    __shared__ int sSeg[4][2 * (SEGCAP + 1)];    // ring of 4: [tile][segment offsets seg_off[s0 .. s0 + SEGCAP]]
    // biases and LayerNorm parameters are read from LDS: a global load inside a phase would make the wave wait for every older
    // load of its queue (memory returns in order per wave) — the prefetched rows of the next pair among them
    __shared__ __attribute__((aligned(16))) float sBias[3 * NP];
    __shared__ __attribute__((aligned(16))) float sGB[2 * NP];
    __shared__ __attribute__((aligned(16))) float sZero[NP];               // a row of zeros (the aggregation's padding rows)
    __shared__ __attribute__((aligned(16))) float sCarry[NP];              // dense mode: partial sum of the segment cut by the end of a pair
    __shared__ __attribute__((aligned(16))) float sBiasN[NODE ? 3 * NP : 4];    // NODE: the node MLP's biases and LayerNorm parameters
    __shared__ __attribute__((aligned(16))) float sGBN[NODE ? 2 * NP : 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int fcol = 16 * wave + 4 * g;                     // this lane's four output features (accumulator layout), sample rows n, n + 16
    const int prow = tid >> 5, pc = (tid & 31) * 4;         // park layout: rows prow, prow + 16, four columns from pc

    // contiguous range of pairs of this workgroup (XCD-aware: each XCD gets a contiguous share when the grid is a multiple of 8)
    // Dense mode (AGG with G4C_AGG_UNIFORM(K), 4 <= K <= 8: the level-1 launches of a kNN mesh): every segment has K rows, so nothing
    // about the rows needs a table.  The rows are split over the workgroups at segment boundaries, evenly in segments, and a workgroup
    // cuts ITS range [R0, R1) into pairs of 64 consecutive rows — full 32-row tiles (tiles of whole segments hold 30 of 32 rows at
    // K = 6 or 5: 6.7 % more tiles) whose segments may straddle tile and pair boundaries: the aggregation works on the pair's 64 fp32
    // rows (fA | fB contiguous) and carries the partial sum of a segment cut by the pair's end to the next pair (agg_tail).
    static_assert((AGG || !DENSE) && !(NODE && DENSE), "dense pairs belong to the fused aggregation of a plain message launch");
    const int KU = DENSE ? p.agg_deg : 0;
    int p_begin, p_end, R0 = 0, R1 = 0;
    {
        const int G = gridDim.x, b = blockIdx.x;
        const int slot = (G & 7) ? b : (b & 7) * (G >> 3) + (b >> 3);
        if (KU) {
            const long long n_seg = p.M / KU;
            R0 = __builtin_amdgcn_readfirstlane((int)(((long long)slot * n_seg) / G) * KU);
            R1 = __builtin_amdgcn_readfirstlane((int)(((long long)(slot + 1) * n_seg) / G) * KU);
            p_begin = 0; p_end = (R1 - R0 + 63) >> 6;          // (local pair numbers)
        } else {
            p_begin = __builtin_amdgcn_readfirstlane((int)(((long long)slot * n_pairs) / G));
            p_end = __builtin_amdgcn_readfirstlane((int)(((long long)(slot + 1) * n_pairs) / G));
        }
    }
    if (p_begin >= p_end) return;
    WS_STAMP_ONCE(12, __builtin_readcyclecounter());
    WS_STAMP_ONCE(14, (unsigned long long)(p_end - p_begin));

    // (the loads are issued where load_meta is called; fix_meta — the v_readfirstlanes that wait for them — an iteration later)
    auto load_meta = [&](int pair) __attribute__((always_inline)) {
        Meta m;
        if (AGG && KU) {
            if (pair > p_end - 1) pair = p_end - 1;
            const int r0 = R0 + 64 * pair, nr = (R1 - r0) < 64 ? (R1 - r0) : 64;
            m.r0[0] = r0; m.n[0] = nr < 32 ? nr : 32; m.r0[1] = r0 + 32; m.n[1] = nr - m.n[0];
            m.s0[0] = r0 / KU;          // the segment the pair's first row belongs to (it started before the pair unless KU s0 == r0)
            m.s1[0] = m.s0[1] = m.s1[1] = 0;
            if (m.n[1] == 0) m.r0[1] = m.r0[0];
            return m;
        }
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
    // one int per thread of a pair's tables: threads [0, 192) the gather indices [tile][kind][row] (rows past a tile's end are
    // clamped copies of its last row: never stored), threads [256, 256 + 2 (SEGCAP + 1)) the segment offsets of the tiles' targets
    const int *const dummy = reinterpret_cast<const int *>(p.b);
    auto load_tables = [&](const Meta &m) __attribute__((always_inline)) {
        const int *ix0 = p.src[0].idx, *ix1 = ADDS ? p.add[0].idx : nullptr, *ix2 = ADDS ? p.add[1].idx : nullptr;
        const int tt = tid < 192 ? tid : 0;
        const int t = tt / 96, k = (tt % 96) >> 5, r = tt & 31;
        const int nn = m.n[t] > 0 ? m.n[t] : m.n[0];
        const int gr = m.r0[t] + (r < nn ? r : nn - 1);
        const int *ix = (k == 0) ? ix0 : (k == 1 ? ix1 : ix2);
        const int *addr = ix ? ix + gr : dummy;
        int j = 0, ts = 0;
        if (AGG && !KU) {
            const int q = tid - 256;
            const bool is_seg = q >= 0 && q < 2 * (SEGCAP + 1);
            ts = is_seg && q >= SEGCAP + 1 ? 1 : 0;
            j = is_seg ? q - ts * (SEGCAP + 1) : 0;
            int sg = m.s0[ts] + j;
            if (sg > m.s1[ts]) sg = m.s1[ts];
            if (is_seg) addr = p.seg_off + sg;
        }
        const int v = *addr;
        return (tid < 192 && !ix) ? gr : v;
    };
    auto store_tables = [&](int v, int it) __attribute__((always_inline)) {
        if (tid < 192) sIdx[it & 1][tid] = v;
        if (AGG && !KU && tid >= 256 && tid < 256 + 2 * (SEGCAP + 1)) sSeg[it & 3][tid - 256] = v;
    };
    // input rows of the weighted block (park layout) and additive rows (accumulator layout) of a pair whose indices are in sIdx[ring].
    // Three batches of four 16-byte loads per lane, issued in three different phases: a CU's share of the HBM bandwidth is ~13 bytes
    // per clock, and all eight waves firing twelve loads at once fill the memory pipeline's queue and block the waves' instruction
    // issue for ~4 k cycles (measured: the phase behind the burst took 5.8 k cycles instead of 1.7 k)
    auto gather_x = [&](const Meta &m, int ring, int t, f32x4 (&xt)[2], unsigned (&raw)[2][2]) __attribute__((always_inline)) {
        {
            const int nn = m.n[t] > 0 ? m.n[t] : m.n[0];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int r = prow + 16 * hh;
                const int gr = DIRECT ? m.r0[t] + (r < nn ? r : nn - 1) : sIdx[ring][t * 96 + r];
                if constexpr (XB16) {       // bf16 rows (8-byte aligned: the launcher checks): widened by a shift / a mask — exact
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 w = *reinterpret_cast<const u32x2 *>(reinterpret_cast<const __bf16 *>(p.src[0].ptr) + (long long)gr * p.src[0].ld + p.src[0].col0 + pc);
                    raw[hh][0] = w[0]; raw[hh][1] = w[1];      // (widen_x before use)
                } else
                xt[hh] = *reinterpret_cast<const f32x4 *>(p.src[0].ptr + (long long)gr * p.src[0].ld + p.src[0].col0 + pc);
            }
        }
    };
    // (bf16 rows stay packed in two registers until they are used: the shift / mask would wait for the load where it was issued)
    auto widen_x = [&](f32x4 (&xt)[2], const unsigned (&raw)[2][2]) __attribute__((always_inline)) {
        if constexpr (XB16) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                xt[hh][0] = __builtin_bit_cast(float, raw[hh][0] << 16); xt[hh][1] = __builtin_bit_cast(float, raw[hh][0] & 0xffff0000u);
                xt[hh][2] = __builtin_bit_cast(float, raw[hh][1] << 16); xt[hh][3] = __builtin_bit_cast(float, raw[hh][1] & 0xffff0000u);
            }
        }
    };
    auto gather_adds = [&](int t, int ring, AddV (&ad)[2][2]) __attribute__((always_inline)) {
        if (ADDS) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int r = n + 16 * rb;
                if constexpr (AB16) {
                    ad[rb][0] = *reinterpret_cast<const u32x2 *>(reinterpret_cast<const __bf16 *>(p.add[0].ptr) + (long long)sIdx[ring][t * 96 + 32 + r] * p.add[0].ld + fcol);
                    ad[rb][1] = *reinterpret_cast<const u32x2 *>(reinterpret_cast<const __bf16 *>(p.add[1].ptr) + (long long)sIdx[ring][t * 96 + 64 + r] * p.add[1].ld + fcol);
                } else {
                    ad[rb][0] = *reinterpret_cast<const f32x4 *>(p.add[0].ptr + (long long)sIdx[ring][t * 96 + 32 + r] * p.add[0].ld + fcol);
                    ad[rb][1] = *reinterpret_cast<const f32x4 *>(p.add[1].ptr + (long long)sIdx[ring][t * 96 + 64 + r] * p.add[1].ld + fcol);
                }
            }
        }
    };

    // ---- this wave's slice of all three layers' weights: 16 output features x 128 k x 2 planes per layer, stationary for the launch.
    // A operand of v_mfma_f32_16x16x32: lane (n, g) holds W[feature 16 wave + n][k = 32 ks + 8 g .. + 7] — in the packed stream
    // (pack_layer_bx6_kernel: [column tile][16-k step][plane][(k / 8 % 2) * 32 + feature % 32][k % 8]) one 16-byte piece per (ks, plane)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, 0x7fffffff, 0x00020000);
    const unsigned lo_b = 2u * (unsigned)((wave >> 1) * 8 * STEP6 + (g >> 1) * STEP6 + ((g & 1) * 32 + 16 * (wave & 1) + n) * 8);
    bf16x8 W[NL][4][SP];
    if (SP == 2) f16_range_mode();
    RangeV rng;                       // running max |value converted to fp16| (mlp_common.h range_track)

    Meta m0 = fix_meta(load_meta(p_begin)), m1 = fix_meta(load_meta(p_begin + 1)), m2 = fix_meta(load_meta(p_begin + 2));
    {
        const int v0 = load_tables(m0), v1 = load_tables(m1);
        store_tables(v0, 0);
        store_tables(v1, 1);
    }
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int pl = 0; pl < SP; ++pl) W[l][ks][pl] = ldw(rs, lo_b + 1024u * pl, (unsigned)l * 2u * BLOCK6 + (unsigned)ks * 4u * STEP6);
    if (tid < NL * NP) sBias[tid] = p.b[tid];
    if (tid < NP) sZero[tid] = 0.f;
    if (tid < 2 * NP) sGB[tid] = p.gamma ? (tid < NP ? p.gamma[tid] : p.beta[tid - NP]) : 0.f;
    int S0 = 0, S1 = 0;                   // NODE: this workgroup's targets = the segments of its tiles
    if constexpr (NODE) {
        if (tid < NL * NP) sBiasN[tid] = q.b[tid];
        if (tid < 2 * NP) sGBN[tid] = q.gamma ? (tid < NP ? q.gamma[tid] : q.beta[tid - NP]) : 0.f;
        const int t1 = 2 * p_end < p.n_tiles ? 2 * p_end : p.n_tiles;
        S0 = __builtin_amdgcn_readfirstlane(p.tile_seg[2 * p_begin]); S1 = __builtin_amdgcn_readfirstlane(p.tile_seg[t1]);
    }
    __syncthreads();

    const bool pact = p.src[0].pre_act != 0;
    __bf16 *const sA = sP, *const sB = sP + TILE_BF16;
    float *const fA = sF, *const fB = sF + FIN;
    Other oA, oB;       // what to do FOR tile A / FOR tile B while the other multiplies
    {
        const int l32 = tid & 31;
        const int acc_off = n * PS + 8 * ((2 * wave + (g >> 1)) ^ n) + 4 * (g & 1);            // features fcol .. fcol + 3 of row n
        const int park_off = prow * PS + 8 * ((l32 >> 1) ^ prow) + 4 * (l32 & 1);               // columns pc .. pc + 3 of row prow
        oA.plane_acc = sA + acc_off; oA.plane_park = sA + park_off; oA.fin = fA + n * HS + fcol;
        oB.plane_acc = sB + acc_off; oB.plane_park = sB + park_off; oB.fin = fB + n * HS + fcol;
    }
    const __bf16 *paA[4], *paB[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { paA[ks] = sA + n * PS + 8 * ((4 * ks + g) ^ n); paB[ks] = paA[ks] + TILE_BF16; }

    f32x4 accA[2], accB[2], accA1[2], accB1[2];
    f32x4 xr[2][2];
    auto bias_init = [&](f32x4 (&acc)[2], f32x4 (&acc1)[2], int l) __attribute__((always_inline)) {
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(sBias + l * NP + fcol);
        acc[0] = b4; acc[1] = b4;
        acc1[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // start of the FIRST pair: tile A's input rows -> planes (nothing to overlap with yet), both tiles' start values = bias + additive
    // rows (the later pairs: inside / after the previous pair's last matrix phase)
    // start values of a tile = bias + additive rows (in this order: what the tile kernels add)
    auto start_values = [&](f32x4 (&acc)[2], f32x4 (&acc1)[2], const AddV (&a)[2][2]) __attribute__((always_inline)) {
        bias_init(acc, acc1, 0);
        if (ADDS) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                f32x4 a0, a1;
                if constexpr (AB16) { a0 = widen_bf16x4(a[rb][0]); a1 = widen_bf16x4(a[rb][1]); }
                else { a0 = a[rb][0]; a1 = a[rb][1]; }
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[rb][e] = (acc[rb][e] + a0[e]) + a1[e];
            }
        }
    };
    // LayerNorm / activation / row stores of a finished pair (its fp32 rows in fA / fB; `mm` = that pair's rows)
    auto ln_tail = [&](const Meta &mm, const int its) __attribute__((always_inline)) {
        // ---- tail of this pair: LayerNorm / activation of both tiles.  16 lanes per row (8 columns each), so the row sums are
        // reduced inside a 16-lane DPP row (quad_perm, row_half_mirror, row_mirror: no LDS round trips); a wave takes 4 rows per
        // pass, the workgroup a whole tile per pass.  The finished rows are stored straight from the registers (16 lanes = one
        // 512-byte row); only the aggregation needs them back in LDS.
        // Stage by stage over BOTH tiles (one basic block per stage: the uniform branches on gamma / the activation are outside the
        // per-tile work), so that one tile's LDS reads, DPP reductions and rsqrt hide under the other tile's arithmetic.
        {
            const int row = wave * 4 + g;
            // this lane's eight columns of a row: fp32 rows out — [4 n, 4 n + 4) and [64 + 4 n, 64 + 4 n + 4), so that each of the two
            // 16-byte stores of the 16 lanes of a row writes 256 contiguous bytes (whole 128-byte lines; with eight consecutive
            // columns per lane each store instruction wrote every other 16 bytes of all four lines of the row); rounded-bf16 mode —
            // [8 n, 8 n + 8): its rows usually go out as bf16, one 16-byte store per lane, 256 contiguous bytes per row (and the
            // LayerNorm's sums are formed in the same order whichever way its rows are stored: the compact message rows must give
            // bit for bit the forward of fp32 rows, test_remus_bf16_compact_messages_are_bit_identical)
            constexpr bool rows16 = SP == 1;
            const int cq[2] = {rows16 ? n * 8 : n * 4, rows16 ? n * 8 + 4 : 64 + n * 4};
            float *const rowp[2] = {fA + row * HS, fB + row * HS};
            float x[2][8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 8; c += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(rowp[t] + cq[c >> 2]);
                    x[t][c] = v[0]; x[t][c + 1] = v[1]; x[t][c + 2] = v[2]; x[t][c + 3] = v[3];
                }
            WS_STAMP_T(17);
            if (p.gamma) {
                float sum[2], mean[2], var[2], rstd[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    sum[t] = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) sum[t] += x[t][c];
                }
                row16_sum2(sum[0], sum[1]);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    mean[t] = sum[t] * (1.0f / NP);
                    var[t] = 0.f;
#pragma unroll
                    for (int c = 0; c < 8; ++c) { const float dl = x[t][c] - mean[t]; var[t] += dl * dl; }
                }
                row16_sum2(var[0], var[1]);
#pragma unroll
                for (int t = 0; t < 2; ++t) rstd[t] = rsqrtf(var[t] * (1.0f / NP) + p.eps);
#pragma unroll
                for (int c = 0; c < 8; c += 4) {
                    const f32x4 g4 = *reinterpret_cast<const f32x4 *>(sGB + cq[c >> 2]), b4 = *reinterpret_cast<const f32x4 *>(sGB + NP + cq[c >> 2]);
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int u = 0; u < 4; ++u) x[t][c + u] = fmaf((x[t][c + u] - mean[t]) * rstd[t], g4[u], b4[u]);
                }
            }
            if (p.act == G4C_ACT_SELU) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int c = 0; c < 8; ++c) x[t][c] = g4c::selu_f(x[t][c]);
            } else if (p.act == G4C_ACT_TANH) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int c = 0; c < 8; ++c) x[t][c] = g4c::tanh_f(x[t][c]);
            }
            WS_STAMP_T(18);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 v0, v1;
                v0[0] = x[t][0]; v0[1] = x[t][1]; v0[2] = x[t][2]; v0[3] = x[t][3]; v1[0] = x[t][4]; v1[1] = x[t][5]; v1[2] = x[t][6]; v1[3] = x[t][7];
                if (AGG) { *reinterpret_cast<f32x4 *>(rowp[t] + cq[0]) = v0; *reinterpret_cast<f32x4 *>(rowp[t] + cq[1]) = v1; }
                if (G4C_WS_ABLATE & 64) { asm volatile("" :: "v"(v0), "v"(v1)); continue; }
                if (p.out && row < mm.n[t]) {
                    const long long orow = (!AGG && p.out_idx) ? p.out_idx[mm.r0[t] + row] : mm.r0[t] + row;
                    if (SP == 1 && p.out_bf16) {
                        // rows kept in bf16 (g4c_mlp_forward_bf16_agg out_dtype; == 2: the reader's pending SELU applied before the one
                        // rounding — the aggregation below still sees the fp32 rows without it), 16 bytes per lane
                        f32x4 w0 = v0, w1 = v1;
                        if (p.out_bf16 == 2) { w0 = selu4(v0); w1 = selu4(v1); }         // (the formula the reader's SELU on load uses)
                        bf16x8 b;
#pragma unroll
                        for (int c = 0; c < 4; ++c) { b[c] = (__bf16)w0[c]; b[4 + c] = (__bf16)w1[c]; }
                        *reinterpret_cast<bf16x8 *>(reinterpret_cast<__bf16 *>(p.out) + orow * p.out_ld + n * 8) = b;
                    } else {
                        float *op = p.out + orow * p.out_ld;
                        *reinterpret_cast<f32x4 *>(op + cq[0]) = v0; *reinterpret_cast<f32x4 *>(op + cq[1]) = v1;
                    }
                }
            }
        }
    };
    // aggregation of a finished pair's targets from the LayerNorm'd rows in fA / fB (`itp` = that pair's iteration: its slot of sSeg)
    auto agg_tail = [&](const Meta &mm, const int itp) __attribute__((always_inline)) {
            const int its = itp; (void)its;
            // aggregation of the targets whose messages the tiles hold (rows in CSR order): the rows of a segment are added in order
            // (clamped loads, predicated adds) and divided by max(count, 1) like segment_reduce_kernel does, so the result is
            // bit-identical to the separate launch.  32 lanes per target (16 bytes each), 16 targets per pass over both tiles.
            const int c4 = (tid & 31) * 4;
            const int *sg_tab = sSeg[itp & 3];
            const int nsA = mm.s1[0] - mm.s0[0], nsB = mm.s1[1] - mm.s0[1];
            auto reduce_rows = [&](const float *sH, int b, int e, int sg) __attribute__((always_inline)) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                WS_STAMP_T(20);
                for (int r0 = b; r0 < e; r0 += 8) {
                    f32x4 v[8];
                    if constexpr (SP == 1) {
                        // rounded-bf16 mode: rows past the segment's end are read from a row of zeros — adding 0.f is what the
                        // predicated form adds for them, without a select per value (pair period -2.7 % there; on the f16x3
                        // stream the same change measures +1.3 %: it keeps the selects)
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>((r0 + u < e ? sH + __umul24((unsigned)(r0 + u), (unsigned)HS) : sZero) + c4);
#pragma unroll
                        for (int u = 0; u < 8; ++u)
#pragma unroll
                            for (int el = 0; el < 4; ++el) a[el] += v[u][el];
                    } else {
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4 *>(sH + (r0 + u < e ? r0 + u : e - 1) * HS + c4);
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const bool on = r0 + u < e;
#pragma unroll
                            for (int el = 0; el < 4; ++el) a[el] += on ? v[u][el] : 0.f;
                        }
                    }
                }
                WS_STAMP_T(21);
                if (p.agg_mean) {
                    if (G4C_WS_MEAN_DIV) {
                        a = g4c::mean_div4(a, (e - b) > 1 ? (e - b) : 1);          // (the IEEE quotient, bit for bit: g4c_common.h)
                    } else {
                        const float cnt = (float)((e - b) > 1 ? (e - b) : 1);
#pragma unroll
                        for (int el = 0; el < 4; ++el) a[el] /= cnt;
                    }
                }
                WS_STAMP_T(22);
                *reinterpret_cast<f32x4 *>(p.agg + (long long)sg * p.agg_ld + c4) = a;
            };
            // Dense mode: every segment has K rows, the pair holds rows [r0, r0 + nr) of the workgroup's range as 64 contiguous fp32 rows
            // (fA | fB).  `lead` rows at the top finish the segment the previous pair cut (its partial sum waits in sCarry), then come
            // nfull whole segments, then `rest` rows of a segment the next pair finishes.  Whole segments: 32 lanes per target, K loads
            // at immediate offsets, the adds in segment order, the mean as the correctly rounded quotient (Markstein's correction with
            // the exact reciprocal of the constant) — no offsets to read, a third of the generic path's instructions
            // (profiles/r06_tail_stamps.log: offsets + addresses 316, rows 724, mean 296 cycles of a tail).  The two partial segments are
            // wave 7's (lanes 0 - 31 the leading one, lanes 32 - 63 the trailing one: the same wave reads the carry before it writes
            // the next, in program order, and no other wave touches it).  The sums are those of g4c_segment_reduce, bit for bit.
            auto reduce_uniform = [&](auto KC) __attribute__((always_inline)) {
                constexpr int K = decltype(KC)::value;
                constexpr float cK = (float)K, yK = 1.0f / (float)K;          // (yK: the correctly rounded reciprocal)
                const int r0 = mm.r0[0], nr = mm.n[0] + mm.n[1], j0 = mm.s0[0];
                const int lead = (j0 * K < r0) ? (j0 * K + K - r0) : 0;
                const int nfull = (nr - lead) / K, rest = nr - lead - nfull * K;
                const int jf = j0 + (lead ? 1 : 0);          // target of the first whole segment
                auto mean4 = [&](f32x4 a) __attribute__((always_inline)) {
                    if (p.agg_mean) {
#pragma unroll
                        for (int el = 0; el < 4; ++el) {
                            const float q0 = a[el] * yK;
                            a[el] = fmaf(fmaf(-cK, q0, a[el]), yK, q0);
                        }
                    }
                    return a;
                };
                for (int q = tid >> 5; q < nfull; q += 16) {
                    const float *base = fA + __umul24((unsigned)(lead + q * K), (unsigned)HS) + c4;
                    f32x4 v[K];
                    WS_STAMP_T(20);
#pragma unroll
                    for (int u = 0; u < K; ++u) v[u] = *reinterpret_cast<const f32x4 *>(base + u * HS);
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < K; ++u) a += v[u];
                    WS_STAMP_T(21);
                    a = mean4(a);
                    WS_STAMP_T(22);
                    *reinterpret_cast<f32x4 *>(p.agg + (long long)(jf + q) * p.agg_ld + c4) = a;
                }
                if (wave == 7 && (lead | rest)) {
                    const bool hi = lane >= 32;
                    const int b = hi ? nr - rest : 0, cnt = hi ? rest : lead;          // this half-wave's rows [b, b + cnt), cnt < K
                    f32x4 v[K - 1];
#pragma unroll
                    for (int u = 0; u < K - 1; ++u) v[u] = *reinterpret_cast<const f32x4 *>((u < cnt ? fA + __umul24((unsigned)(b + u), (unsigned)HS) : sZero) + c4);
                    f32x4 a = *reinterpret_cast<const f32x4 *>((hi || !lead ? sZero : sCarry) + c4);
#pragma unroll
                    for (int u = 0; u < K - 1; ++u) a += v[u];          // (rows past cnt: + 0.f, as the predicated form adds)
                    if (!hi && lead) *reinterpret_cast<f32x4 *>(p.agg + (long long)j0 * p.agg_ld + c4) = mean4(a);
                    if (hi && rest) *reinterpret_cast<f32x4 *>(sCarry + c4) = a;
                }
            };
            if (KU) {
                typedef std::integral_constant<int, 4> K4; typedef std::integral_constant<int, 5> K5; typedef std::integral_constant<int, 6> K6;
                typedef std::integral_constant<int, 7> K7; typedef std::integral_constant<int, 8> K8;
                if (KU == 6) reduce_uniform(K6{});
                else if (KU == 5) reduce_uniform(K5{});
                else if (KU == 4) reduce_uniform(K4{});
                else if (KU == 7) reduce_uniform(K7{});
                else reduce_uniform(K8{});
                return;
            }
            for (int q = tid >> 5; q < nsA + nsB; q += 16) {
                const int t = q >= nsA ? 1 : 0, j = q - (t ? nsA : 0);
                if (j < SEGCAP) {
                    const int b = sg_tab[t * (SEGCAP + 1) + j] - mm.r0[t], e = sg_tab[t * (SEGCAP + 1) + j + 1] - mm.r0[t];
                    reduce_rows(t ? fB : fA, b, e, mm.s0[t] + j);
                }
            }
            if (nsA > SEGCAP || nsB > SEGCAP) {          // (a tile with a long run of empty segments: their offsets from global memory)
                for (int q = tid >> 5; q < nsA + nsB; q += 16) {
                    const int t = q >= nsA ? 1 : 0, j = q - (t ? nsA : 0);
                    if (j >= SEGCAP) {
                        const int sg = mm.s0[t] + j;
                        reduce_rows(t ? fB : fA, p.seg_off[sg] - mm.r0[t], p.seg_off[sg + 1] - mm.r0[t], sg);
                    }
                }
            }
    };
    // Loop-carried: xr[1] / adB = input rows / additive rows of the CURRENT pair's tile B (gathered in the previous tail: parked /
    // added in and after the first matrix phase), accA = tile A's start values, tile A's rows in its planes.
    AddV adB[2][2];
    unsigned rawB[2][2] = {{0u, 0u}, {0u, 0u}};        // (XB16: tile B's bf16 rows as loaded)
    {
        AddV adA[2][2];
        unsigned rawA[2][2] = {{0u, 0u}, {0u, 0u}};
        gather_x(m0, 0, 0, xr[0], rawA);
        gather_x(m0, 0, 1, xr[1], rawB);
        gather_adds(0, 0, adA);
        gather_adds(1, 0, adB);
        widen_x(xr[0], rawA);
        if (pact) other_all<SP, 2, true>(accA, accA1, xr[0], oA, rng);
        else other_all<SP, 2, false>(accA, accA1, xr[0], oA, rng);
        start_values(accA, accA1, adA);
    }
    __syncthreads();                                       // tile A's planes of the first pair visible

    for (int it = 0, pair = p_begin; pair < p_end; ++pair, ++it) {
        WS_STAMP(0);
        // ---- tables two pairs ahead (their meta was loaded an iteration ago), meta three pairs ahead
        const Meta m3raw = load_meta(pair + 3);
        const int tv = load_tables(m2);
        // (tile A's planes were written in the previous iteration's last matrix phase — before the loop for the first pair — and
        // that phase's barrier lies between; nothing the stragglers of the previous tail still read is written in this phase)
        WS_STAMP(1);
        widen_x(xr[1], rawB);
        if (pact) m_block<SP, 2, true>(paA, W[0], accA, accA1, accA, accA1, xr[1], oB, rng);                 // for B: park
        else m_block<SP, 2, false>(paA, W[0], accA, accA1, accA, accA1, xr[1], oB, rng);
        start_values(accB, accB1, adB);
        __syncthreads();
        WS_STAMP(2);
        // ---- tile A's rows one pair ahead (indices in LDS since the previous iteration); tile B's: in the tail
        f32x4 nxa[2];
        AddV nadA[2][2];
        unsigned nraw[2][2] = {{0u, 0u}, {0u, 0u}};
        gather_x(m1, (it + 1) & 1, 0, nxa, nraw);
        m_block<SP, 1>(paB, W[0], accB, accB1, accA, accA1, xr[1], oA, rng);                 // for A: epilogue of layer 0
        bias_init(accA, accA1, 1);
        __syncthreads();
        WS_STAMP(3);
        gather_adds(0, (it + 1) & 1, nadA);
        m_block<SP, 1>(paA, W[1], accA, accA1, accB, accB1, xr[1], oB, rng);                 // for B: epilogue of layer 0
        bias_init(accB, accB1, 1);
        __syncthreads();
        WS_STAMP(4);
        if constexpr (NL == 3) {
            m_block<SP, 1>(paB, W[1], accB, accB1, accA, accA1, xr[1], oA, rng);             // for A: epilogue of layer 1
            bias_init(accA, accA1, 2);
            __syncthreads();
            WS_STAMP(5);
            m_block<SP, 1>(paA, W[2], accA, accA1, accB, accB1, xr[1], oB, rng);             // for B: epilogue of layer 1
            bias_init(accB, accB1, 2);
            __syncthreads();
            WS_STAMP(6);
        }
        // for A: last layer's fp32 rows — and the NEXT pair's tile A parked into A's planes (their last readers, M(A, NL - 1), are
        // behind the previous barrier; the rows were gathered four phases ago): this phase has next to no vector work of its own
        widen_x(nxa, nraw);
        if (pact) m_block<SP, 4, true>(paB, W[NL - 1], accB, accB1, accA, accA1, nxa, oA, rng);
        else m_block<SP, 4, false>(paB, W[NL - 1], accB, accB1, accA, accA1, nxa, oA, rng);
        other_all<SP, 3, false>(accB, accB1, xr[1], oB, rng);                                // B's last layer -> fp32 rows
        __syncthreads();
        WS_STAMP(7);
        // ---- the next pair's tile A start values; then its tile B rows (parked under its M(A', 0)) are gathered: in front of this
        // tail's stores (memory returns in order per wave), a tail and a phase ahead of their use, so that they are not live across
        // this pair's matrix phases (tile B's additive rows: at the end of the tail)
        start_values(accA, accA1, nadA);
        // (pinned here: hipcc otherwise sinks these adds — and their wait for the additive rows — to the end of the iteration, behind
        // the tail's stores, where s_waitcnt vmcnt(0) also waits for every store to be acknowledged)
        asm volatile("" : "+v"(accA[0]), "+v"(accA[1]) :: "memory");
        // (the meta three pairs ahead, requested at the top: taken here, in front of the tail's stores — at the end of the iteration
        // the wait for these loads would also wait for every store of the tail, which memory acknowledges in order)
        const Meta m3 = fix_meta(m3raw);
        gather_x(m1, (it + 1) & 1, 1, xr[1], rawB);
        // the tables fetched at the top of this iteration (older than every other load in flight) go to the ring slot of the pair
        // whose rows were gathered in the previous iteration; the next iteration's top barrier publishes them
        store_tables(tv, it + 2);
        WS_STAMP(8);

        if (!(G4C_WS_ABLATE & 256)) ln_tail(m0, it);
        // (rounded-bf16 mode: the row stores are half as many bytes, and the gathers in front of the aggregation's barrier measure
        // 1.8 % faster per pair than behind the aggregation; f16x3 stream: 3 % slower — they queue behind the fp32 row stores)
        if constexpr (SP == 1) gather_adds(1, (it + 1) & 1, adB);
        WS_STAMP(9);
        // (no barrier at the end of a tail without the aggregation: the next pair's parked rows — what a wave that runs ahead into
        // M(A', 0) reads — were written before the barrier that closed the last matrix phase)
        if (AGG) {
            __syncthreads();
            WS_STAMP(15);
            if (!(G4C_WS_ABLATE & 128)) agg_tail(m0, it);
        }
        // the next pair's tile B additive rows (added after its M(A', 0)): behind this tail's stores — issued together with tile B's
        // rows at the top of the tail, the six loads per lane held up the LayerNorm's stores (tail 3.4 k -> 6.6 k ticks)
        WS_STAMP(16);
        if constexpr (SP != 1) gather_adds(1, (it + 1) & 1, adB);
        WS_STAMP(10);
        m0 = m1; m1 = m2; m2 = m3;
    }

    if constexpr (NODE) {
        // ================================================================ node update of this workgroup's targets [S0, S1)
        // The aggregates were written to p.agg by this workgroup's OWN waves (same CU, same write-through L1, lines nobody read before):
        // workgroup scope is enough — the stores have left the waves (vmcnt) before the barrier.  (Agent scope costs a write-back of
        // the XCD's L2 per workgroup: measured +23 us per launch.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        NodeCtx nc;
        nc.tid = tid; nc.wave = wave; nc.n = n; nc.g = g; nc.fcol = fcol; nc.prow = prow; nc.pc = pc; nc.lo_b = lo_b;
        nc.sBiasN = sBiasN; nc.sGBN = sGBN; nc.fA = fA; nc.fB = fB; nc.oA = oA; nc.oB = oB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { nc.paA[ks] = paA[ks]; nc.paB[ks] = paB[ks]; }
        node_phase<SP, NL>(nc, p.agg, p.agg_ld, q, S0, S1);
    }
    WS_STAMP_ONCE(13, __builtin_readcyclecounter());
    if (SP == 2) {
        if (G4C_WS_SCALED && !(G4C_WS_ABLATE & 48)) rng.m *= F16_LO_UNSCALE;       // (tracked in units of 2^-11)
        range_report(p, rng);
    }
}


}  // namespace

namespace g4cm {

// 0 off, 1 (default) launches of at least 20 000 rows (measured on the level-1 message launch against
// the two-way instantiation of mlp_bx6i_kernel, which it replaces: 322 us against 339 us with the fused aggregation, 288 against 292
// without), 2 every launch it can take (tests)
static int g_ws = -1;
int ws_enable(int on) {
    if (g_ws < 0) g_ws = 1;
    const int old = g_ws;
    if (on >= 0) g_ws = on > 2 ? 2 : on;
    return old;
}

bool ws_eligible(const Params &p, bool round1, bool agg, bool save, bool f16x2, long long row_count, bool any_size) {
    constexpr long long min_rows = 20000;          // (same-box sweeps of round 3: ahead of the tile kernel from ~20 k rows)
    const int mode = any_size ? 2 : ws_enable(-1);          // (any_size: the fused MP layer asks whether the SHAPE fits, whatever the mode)
    if (!mode || save || !(f16x2 || round1)) return false;          // (the bf16x6 stream keeps mlp_bx6i_kernel / mlp_bx6_kernel)
    if (mode == 1 && row_count < min_rows) return false;
    if (p.n_src != 1 || p.n_nar != 0 || (p.n_add != 0 && p.n_add != 2) || p.n_heads) return false;
    if (round1 && p.n_add != 2) return false;
    if ((p.n_layers != 3 && p.n_layers != 2) || p.n_out != NP || p.resid) return false;
    if (p.out_bf16 && (!round1 || (p.out_ld & 7) || ((uintptr_t)p.out & 15))) return false;
    if (p.out_idx && (agg || !p.out)) return false;          // (scattered output rows: the plain launch only)
    const Src &s = p.src[0];
    if (s.width != NP || !s.vec || s.seg_off || (s.bf16 && !round1)) return false;
    for (int a = 0; a < p.n_add; ++a)
        if (p.add[a].width != NP || (p.add[a].ld & 3) || ((uintptr_t)p.add[a].ptr & (p.add[a].bf16 ? 7 : 15)) || p.add[a].bf16 != p.add[0].bf16 ||
            (p.add[a].bf16 && !round1)) return false;
    if (p.out && !p.out_bf16 && ((p.out_ld & 3) || ((uintptr_t)p.out & 15))) return false;
    if (p.gamma && (((uintptr_t)p.gamma & 15) || ((uintptr_t)p.beta & 15))) return false;
    if (((uintptr_t)p.b & 15)) return false;
    if (p.M >= (1LL << 31)) return false;
    return true;
}

int ws_launch(const Params &p, bool agg, bool round1, hipStream_t st, const NodeParams *node) {
    // (dense mode — uniform segments of 4 .. 8 rows — cuts the rows into pairs of 64 itself: n_pairs only sizes the grid there)
    // Not for the fused MP layer: its launches are a few pairs per workgroup (nothing to win from denser pairs), and its three-layer
    // instantiation sits at 256 registers — with the dense bookkeeping it spills (config 2: 1 786 -> 1 734 steps/s, same box).
    const bool dense = agg && !node && p.agg_deg >= 4 && p.agg_deg <= 8;
    const int n_pairs = dense ? (int)((p.M + 63) / 64) : (p.n_tiles + 1) / 2;
    if (n_pairs == 0) return G4C_OK;
    const int n_wg = g4c::cu_count() * (round1 ? G4C_WS_SP1_MINW / 2 : 1);          // persistent workgroups: one (SP = 1: G4C_WS_SP1_MINW / 2) per CU
    const dim3 grid(n_pairs < n_wg ? n_pairs : n_wg), blk(512);
    const bool direct = p.src[0].idx == nullptr, adds = p.n_add == 2, two = p.n_layers == 2, xb16 = p.src[0].bf16 != 0;
    const bool ab16 = adds && p.add[0].bf16 != 0;
    NodeParams q{};
    if (node) {          // the fused MP layer (g4c_mp_layer_forward_bx6): f16x3 stream, hoisted message MLP, fused aggregation
        G4C_REQUIRE(agg && !round1 && adds, G4C_EUNSUPPORTED, "g4c_mp_layer_forward_bx6: needs the hoisted f16x3 message launch with the fused aggregation");
        q = *node;
#define G4C_WS_NODE(DIRECT, NL) mlp_ws_kernel<true, DIRECT, true, 2, NL, false, false, true><<<grid, blk, 0, st>>>(p, n_pairs, q)
        if (direct) { if (two) G4C_WS_NODE(true, 2); else G4C_WS_NODE(true, 3); }
        else { if (two) G4C_WS_NODE(false, 2); else G4C_WS_NODE(false, 3); }
#undef G4C_WS_NODE
        return g4c::check_launch("g4c_mp_layer_forward_bx6");
    }
#define G4C_WS_GO(AGG, DIRECT, ADDS, SP, NL, XB16)                                                                                      \
    do { if (AGG && dense) mlp_ws_kernel<AGG, DIRECT, ADDS, SP, NL, XB16, false, false, AGG><<<grid, blk, 0, st>>>(p, n_pairs, q);       \
         else mlp_ws_kernel<AGG, DIRECT, ADDS, SP, NL, XB16><<<grid, blk, 0, st>>>(p, n_pairs, q); } while (0)
#define G4C_WS_GO1(AGG, DIRECT, NL, XB16)                                                                                               \
    do { if (AGG && dense) { if (ab16) mlp_ws_kernel<AGG, DIRECT, true, 1, NL, XB16, true, false, AGG><<<grid, blk, 0, st>>>(p, n_pairs, q);       \
                             else mlp_ws_kernel<AGG, DIRECT, true, 1, NL, XB16, false, false, AGG><<<grid, blk, 0, st>>>(p, n_pairs, q); }        \
         else if (ab16) mlp_ws_kernel<AGG, DIRECT, true, 1, NL, XB16, true><<<grid, blk, 0, st>>>(p, n_pairs, q);                        \
         else mlp_ws_kernel<AGG, DIRECT, true, 1, NL, XB16, false><<<grid, blk, 0, st>>>(p, n_pairs, q); } while (0)
#define G4C_WS_SHAPE(AGG, DIRECT)                                                                    \
    do {                                                                                             \
        if (round1) {                                                                                \
            if (two) { if (xb16) G4C_WS_GO1(AGG, DIRECT, 2, true); else G4C_WS_GO1(AGG, DIRECT, 2, false); }      \
            else { if (xb16) G4C_WS_GO1(AGG, DIRECT, 3, true); else G4C_WS_GO1(AGG, DIRECT, 3, false); }          \
        } else if (two) {                                                                            \
            if (adds) G4C_WS_GO(AGG, DIRECT, true, 2, 2, false); else G4C_WS_GO(AGG, DIRECT, false, 2, 2, false);                 \
        } else {                                                                                     \
            if (adds) G4C_WS_GO(AGG, DIRECT, true, 2, 3, false); else G4C_WS_GO(AGG, DIRECT, false, 2, 3, false);                 \
        }                                                                                            \
    } while (0)
    if (agg) { if (direct) G4C_WS_SHAPE(true, true); else G4C_WS_SHAPE(true, false); }
    else { if (direct) G4C_WS_SHAPE(false, true); else G4C_WS_SHAPE(false, false); }
#undef G4C_WS_SHAPE
#undef G4C_WS_GO1
#undef G4C_WS_GO
    return g4c::check_launch("g4c_mlp_forward (ws)");
}

}  // namespace g4cm

extern "C" int g4c_mlp_ws_enable(int on) { return g4cm::ws_enable(on); }
