// Dual-tile, software-pipelined form of the fused split-operand MLP ("bx6i") for the MP layers' message launch: the same arithmetic
// as mlp_bx6_kernel (mlp_fused.hip) — gather -> [SELU on load] -> Linear/SELU chain on the 16-bit matrix pipe (SP = 2: two-way fp16
// split, three products, v_mfma_f32_32x32x16_f16, the default; SP = 3: exact three-way bf16 split, six products,
// v_mfma_f32_32x32x16_bf16) -> LayerNorm -> activation -> store (-> per-target aggregation) — replacing MLP.forward
// (graphs4cfd/nn/blocks.py:117-144) with the torch.cat / index ops in front of it (nn/blocks.py:181,328) and, with AGG, the
// scatter(e', col, reduce) behind it (nn/blocks.py:183,330).
//
// What is different: a workgroup (4 waves, one 32-column tile each, as in mlp_bx6_kernel) owns TWO 32-row tiles A and B and
// alternates between them layer by layer:  M(A,0) M(B,0) M(A,1) M(B,1) ...  While a wave issues the MFMAs of one tile's
// layer, the vector ALUs of the SAME wave work on the other tile: parking B's input rows under M(A,0), the hidden-layer
// epilogue (bias is the accumulator's start value; SELU, operand split, planes) of B's layer l-1 under M(A,l), of A's layer l
// under M(B,l), the last layer's fp32 rows under the other tile's last M phase.  In mlp_bx6_kernel a wave is either in an
// MFMA phase or in a vector phase and relies on the other waves of its SIMD to fill the pipe it leaves idle; here every wave
// keeps both busy.  The wave's slice of the layer's weights is stationary in registers for the pair (one fetch serves 64 rows).
// The phases are straight-line code on purpose: in a loop hipcc duplicates the stationary weight registers across the back edge.
// Prologue order: gather indices, directly addressed input rows, [weights]; tile A is parked while the additive
// rows of both tiles are in flight.  With AGG the finished rows are stored from their LDS copy, whole rows per store instruction.
//
// Envelope (everything else runs mlp_bx6_kernel): split-operand modes, ONE weighted 128-wide 16-byte aligned input block (rows direct
// or through an index, optional SELU on load), 0 or 2 additive 128-wide blocks, no narrow blocks, three layers, 128-wide output
// rows without residual / output index / heads.
#include "mlp_common.h"
#include <cstdlib>
using namespace g4cm;

#ifdef G4C_BX6I_TIMING
__device__ unsigned long long g4c_bx6i_stamps[256 * 32];
extern "C" int g4c_bx6i_read_stamps(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g4c_bx6i_stamps), sizeof(unsigned long long) * n);
}
#define BI_STAMP(k) do { if (pair < 256 && tid == 0) g4c_bx6i_stamps[pair * 32 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define BI_STAMP(k) do {} while (0)
#endif

namespace {

constexpr int PLN = 32 * HB;                 // bf16 elements of one operand plane of a tile [32][136]
// one tile: SP planes (the fp32 final rows [32][132] alias them: 16 896 B <= 2 planes = 17 408 B)

__device__ __forceinline__ f32x2 selu2i(f32x2 x) { return selu2(x); }

// exact three-way split of a pair -> one packed pair per plane (planes PLN elements apart); SP == 2: the two-way fp16 split
template <int SP>
__device__ __forceinline__ void put_pair(__bf16 *d, f32x2 y) {
    f32x2 hf, mf, lf;
    if (SP == 2) {
        unsigned hu, lu;
        split_pair_f16(y, hu, lu);
        *reinterpret_cast<unsigned *>(d) = hu;
        *reinterpret_cast<unsigned *>(d + PLN) = lu;
        return;
    }
    const unsigned hu = pack_bf16(y, hf);
    const f32x2 r1 = y - hf;
    const unsigned mu = pack_bf16(r1, mf);
    const f32x2 r2 = r1 - mf;
    const unsigned lu = pack_bf16(r2, lf);
    *reinterpret_cast<unsigned *>(d) = hu;
    *reinterpret_cast<unsigned *>(d + PLN) = mu;
    *reinterpret_cast<unsigned *>(d + 2 * PLN) = lu;
}

// What the vector ALUs do for the OTHER tile while this tile's MFMAs issue, one slice per 16-k step (s = 0..7):
//   EK 0 nothing;  1 hidden-layer epilogue of accE (pair s);  2 park chunk s/2 of the gathered input rows xe (odd s);
//   3 last layer: fp32 rows of accE into the tile's final buffer (quad s/2, odd s)
struct Other {
    __bf16 *plane_acc;        // EK 1: this lane's element (row i, feature fbase) of the other tile's planes
    __bf16 *plane_park;       // EK 2: (row grow_l, column c4)
    float *fin;               // EK 3: (row i, feature fbase) of the other tile's fp32 rows
    bool park_act;            // EK 2: SELU pending on the stored rows
};

template <int EK, int SP>
__device__ __forceinline__ void other_slice(int s, const f32x16 &accE, const f32x16 &accE1, const f32x4 (&xe)[4], const Other &o) {
    if (EK == 1) {
        const int gq = s >> 1, pr = s & 1;
        f32x2 x;
        x[0] = accE[4 * gq + 2 * pr]; x[1] = accE[4 * gq + 2 * pr + 1];
        if (SP == 2) { f32x2 x1; x1[0] = accE1[4 * gq + 2 * pr]; x1[1] = accE1[4 * gq + 2 * pr + 1]; x = x1 * F16_LO_UNSCALE + x; }      // (one v_pk_fma_f32)
        put_pair<SP>(o.plane_acc + 8 * gq + 2 * pr, selu2i(x));
    } else if (EK == 2) {
        if (s & 1) {
            const int q = s >> 1;
            f32x4 v = xe[q];
            if (o.park_act) v = selu4(v);
            bf16x4 vh, vm, vl;
            split3x4<SP>(v, vh, vm, vl);
            __bf16 *d = o.plane_park + q * KC;
            *reinterpret_cast<bf16x4 *>(d) = vh;
            *reinterpret_cast<bf16x4 *>(d + PLN) = vm;
            if (SP == 3) *reinterpret_cast<bf16x4 *>(d + 2 * PLN) = vl;
        }
    } else if (EK == 3) {
        if (s & 1) {
            const int gq = s >> 1;
            f32x4 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = SP == 2 ? fmaf(accE1[4 * gq + e], F16_LO_UNSCALE, accE[4 * gq + e]) : accE[4 * gq + e];
            *reinterpret_cast<f32x4 *>(o.fin + 8 * gq) = x;
        }
    }
}

// One 128-k block for one tile: acc += W(block) x planes.  The wave's 32-column slice of the layer's weights is STATIONARY in
// registers for the two tiles of the pair (W[step][plane], 96 VGPRs: one fetch serves 64 rows, half the L1 -> register traffic of
// mlp_bx6_kernel); REFILL (the second tile's phase): step s's fragments are replaced by the next layer's right after their last
// use.  The phases of a pair are straight-line code (three layers, unrolled), so every refill is a plain redefinition.
// SP == 2 (two-way fp16 split): two planes, W[step][2] (64 VGPRs), three products per step — the 2^-11 terms in acc1.
constexpr int BX6I_VALU_PER_MFMA = 8;      // (SP == 2 form only: vector instructions interleaved after each of the step's three MFMAs)
template <int EK, bool REFILL, int SP>
__device__ __forceinline__ void m_block(const __bf16 *pa, bf16x8 (&W)[8][SP], __amdgpu_buffer_rsrc_t rs, unsigned lo_b, unsigned wnext,
                                        f32x16 &acc, f32x16 &acc1, const f32x16 &accE, const f32x16 &accE1, const f32x4 (&xe)[4], const Other &o) {
    bf16x8 cur[SP], nx[SP];
#pragma unroll
    for (int pl = 0; pl < SP; ++pl) cur[pl] = *reinterpret_cast<const bf16x8 *>(pa + pl * PLN);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s < 7) {
#pragma unroll
            for (int pl = 0; pl < SP; ++pl) nx[pl] = *reinterpret_cast<const bf16x8 *>(pa + pl * PLN + 16 * (s + 1));
        }
        if (!EK) __builtin_amdgcn_sched_barrier(0);
        other_slice<EK, SP>(s, accE, accE1, xe, o);
        if (SP == 2) {
            acc1 = mfma_f16(W[s][0], cur[1], acc1);
            acc1 = mfma_f16(W[s][1], cur[0], acc1);
            acc = mfma_f16(W[s][0], cur[0], acc);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[s][0], cur[SP - 1], acc, 0, 0, 0);     // small terms first (as mlp_bx6_kernel)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[s][SP - 1], cur[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[s][1], cur[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[s][0], cur[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[s][1], cur[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[s][0], cur[0], acc, 0, 0, 0);
        }
        if (EK) {
            __builtin_amdgcn_sched_group_barrier(0x100, SP, 0);                 // DS read
#pragma unroll
            for (int m = 0; m < (SP == 2 ? 3 : 6); ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, SP == 2 ? BX6I_VALU_PER_MFMA : 4, 0);              // VALU
            }
            __builtin_amdgcn_sched_group_barrier(0x200, SP, 0);                 // DS write
        }
        __builtin_amdgcn_sched_barrier(0);
        if (REFILL) {
            const unsigned so = wnext + 2u * s * STEP6;
#pragma unroll
            for (int pl = 0; pl < SP; ++pl) W[s][pl] = ldw(rs, lo_b + 1024u * pl, so);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (s < 7) {
#pragma unroll
            for (int pl = 0; pl < SP; ++pl) cur[pl] = nx[pl];
        }
    }
}

// (Round 3: this kernel takes the bf16x6 stream only — the two-way instantiation's knobs (workgroups per CU, late first-layer weight
// fetch, deferred additive rows of tile B: 24 - 28 spilled registers, 469 us against 382) went with it.  Finished rows are stored
// from their LDS copy, whole rows per store instruction, with and without the fused aggregation: -20 us on the level-1 launch.)
constexpr bool BX6I_ROW_STORES = true;
// DIRECT: the weighted block's rows are the tile's own rows (no gather index: the MP layers' edge latents) — their loads do not wait
// for the index round trip.
template <bool AGG, int SP, bool DIRECT>
__global__ __launch_bounds__(256, 2) void mlp_bx6i_kernel(const Params p) {
    // two tiles' operand planes + gather indices: 52 992 B; 96 stationary weight registers -> two workgroups per CU
    // (SP == 2: 35 584 B, 64 weight registers, 168 VGPRs -> three workgroups per CU: 385 us against 412 us at two, level-1 launch)
    constexpr int TILE_BF16 = SP * PLN;
    __shared__ __attribute__((aligned(16))) __bf16 sB[2 * TILE_BF16];
    if (SP == 2) f16_range_mode();
    __shared__ int sIdx[2][3][32];          // [tile][weighted block, additive 0, additive 1][row]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int fbase = wave * 32 + 4 * h;
    const int grow_l = (lane >> 3) + 8 * wave, c4 = (lane & 7) * 4;

    // pair of tiles of this workgroup (XCD-aware order: each XCD gets a contiguous range of pairs)
    const int n_pairs = (p.n_tiles + 1) >> 1;
    int pair;
    {
        const int b = blockIdx.x, q = n_pairs >> 3, r = n_pairs & 7, x = b & 7, j = b >> 3;
        pair = __builtin_amdgcn_readfirstlane((x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j);
    }
    int row0[2], nrow[2];
    if (AGG) {
        // three independent scalar loads (tile 2 pair always exists; the entries of a missing second tile are clamped to the end of
        // the table, which gives it zero rows) instead of two dependent vector round trips
        const int t0 = 2 * pair, t1 = t0 + 1 < p.n_tiles ? t0 + 1 : p.n_tiles, t2 = t0 + 2 < p.n_tiles ? t0 + 2 : p.n_tiles;
        const int r0 = p.tile_rows[t0], r1 = p.tile_rows[t1], r2 = p.tile_rows[t2];
        row0[0] = r0; nrow[0] = r1 - r0; row0[1] = t0 + 1 < p.n_tiles ? r1 : 0; nrow[1] = r2 - r1;
    } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int tile = 2 * pair + t;
            if (tile >= p.n_tiles) { row0[t] = 0; nrow[t] = 0; }
            else { row0[t] = (int)p.row_base + tile * 32; const int lim = (int)p.M - row0[t]; nrow[t] = lim < 32 ? lim : 32; }
        }
    }
    BI_STAMP(0);
    if (nrow[1] == 0) row0[1] = row0[0];        // (odd tile count: the second tile recomputes the first tile's rows and stores nothing)

    // ---- Memory returns in order per wave, so the loads that head a dependent chain go first: the gather indices of both tiles
    // (rows past a tile's end are clamped copies of its last row: never stored), then the input rows of both tiles (park layout; with
    // DIRECT they do not depend on an index and are back when the index round trip ends: tile A is parked while the additive rows,
    // which do, are in flight), then the weights.
    int idx_val;
    {
        const int tt = tid < 192 ? tid : 0;
        const int t = tt / 96, k = (tt % 96) >> 5, r = tt & 31;
        const int nn = nrow[t] > 0 ? nrow[t] : nrow[0];
        const int gr = row0[t] + (r < nn ? r : nn - 1);
        // (selects among three uniform pointers: indexing p.add[] with the per-lane k is a load from the kernel argument segment,
        // one more dependent round trip in front of the index load)
        const int *ix0 = p.src[0].idx, *ix1 = p.n_add > 0 ? p.add[0].idx : nullptr, *ix2 = p.n_add > 1 ? p.add[1].idx : nullptr;
        const int *ix = (k == 0) ? ix0 : (k == 1 ? ix1 : ix2);
        idx_val = gr;
        if (ix) idx_val = ix[gr];
    }
    f32x4 xA[4], xB[4];
    if (DIRECT) {
        const int nA = nrow[0], nB = nrow[1] > 0 ? nrow[1] : nrow[0];
        const float *ra = p.src[0].ptr + (long long)(row0[0] + (grow_l < nA ? grow_l : nA - 1)) * p.src[0].ld + p.src[0].col0 + c4;
        const float *rb = p.src[0].ptr + (long long)(row0[1] + (grow_l < nB ? grow_l : nB - 1)) * p.src[0].ld + p.src[0].col0 + c4;
#pragma unroll
        for (int q = 0; q < 4; ++q) { xA[q] = *reinterpret_cast<const f32x4 *>(ra + q * KC); xB[q] = *reinterpret_cast<const f32x4 *>(rb + q * KC); }
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, 0x7fffffff, 0x00020000);
    const unsigned lo_b = 2u * (unsigned)(wave * 8 * STEP6 + lane * 8);
    bf16x8 W[8][SP];
    auto load_w = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int pl = 0; pl < SP; ++pl) W[s][pl] = ldw(rs, lo_b + 1024u * pl, 2u * s * STEP6);
    };
    // LATE_W: the first layer's weights (L2 hits) are fetched after tile A's additive rows are consumed, which leaves the registers
    // for BOTH tiles' additive gathers to be in flight together — one round trip instead of two in front of the first matrix phase
    constexpr bool LATE_W = false;          // (the two-way instantiation fetched its first layer's weights after tile A's additive rows)
    if (!LATE_W) load_w();
    if (tid < 192) sIdx[0][0][tid] = idx_val;          // [t][k][r] = [tid / 96][(tid % 96) / 32][tid % 32]
    __syncthreads();
    BI_STAMP(1);

    // ---- additive rows of both tiles (accumulator layout) + first bias: the start values
    if (!DIRECT) {
        const float *ra = p.src[0].ptr + (long long)sIdx[0][0][grow_l] * p.src[0].ld + p.src[0].col0 + c4;
        const float *rb = p.src[0].ptr + (long long)sIdx[1][0][grow_l] * p.src[0].ld + p.src[0].col0 + c4;
#pragma unroll
        for (int q = 0; q < 4; ++q) { xA[q] = *reinterpret_cast<const f32x4 *>(ra + q * KC); xB[q] = *reinterpret_cast<const f32x4 *>(rb + q * KC); }
    }
    f32x16 accA, accB, accA1, accB1;         // (acc?1: the 2^-11 terms, SP == 2 only)
#pragma unroll
    for (int q = 0; q < 16; ++q) { accA1[q] = 0.f; accB1[q] = 0.f; }
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(p.b + fbase + 8 * gq);
#pragma unroll
        for (int e = 0; e < 4; ++e) { accA[4 * gq + e] = b4[e]; accB[4 * gq + e] = b4[e]; }
    }
    const bool adds = p.n_add == 2;
    f32x4 a0[4], a1[4], b0[4], b1[4];
    auto issue_adds = [&](int t, f32x4 (&u0)[4], f32x4 (&u1)[4]) __attribute__((always_inline)) {
        const float *p0 = p.add[0].ptr + (long long)sIdx[t][1][i] * p.add[0].ld + fbase;
        const float *p1 = p.add[1].ptr + (long long)sIdx[t][2][i] * p.add[1].ld + fbase;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) { u0[gq] = *reinterpret_cast<const f32x4 *>(p0 + 8 * gq); u1[gq] = *reinterpret_cast<const f32x4 *>(p1 + 8 * gq); }
    };
    auto take_adds = [&](f32x16 &acc, const f32x4 (&u0)[4], const f32x4 (&u1)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * gq + e] = (acc[4 * gq + e] + u0[gq][e]) + u1[gq][e];
    };
    if (adds) { issue_adds(0, a0, a1); if (LATE_W) issue_adds(1, b0, b1); }
    const bool pact = p.src[0].pre_act != 0;
    __bf16 *const sA = sB, *const sBt = sB + TILE_BF16;
    // park tile A (not overlapped with MFMAs: nothing to multiply yet — but under tile A's additive gathers)
    {
        __bf16 *d = sA + grow_l * HB + c4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = xA[q];
            if (pact) v = selu4(v);
            bf16x4 vh, vm, vl;
            split3x4<SP>(v, vh, vm, vl);
            *reinterpret_cast<bf16x4 *>(d + q * KC) = vh;
            *reinterpret_cast<bf16x4 *>(d + PLN + q * KC) = vm;
            if (SP == 3) *reinterpret_cast<bf16x4 *>(d + 2 * PLN + q * KC) = vl;
        }
    }
    BI_STAMP(2);
    if (LATE_W) {
        if (adds) take_adds(accA, a0, a1);
        __builtin_amdgcn_sched_barrier(0);
        load_w();
        __builtin_amdgcn_sched_barrier(0);
        if (adds) take_adds(accB, b0, b1);
    } else {
        if (adds) { take_adds(accA, a0, a1); issue_adds(1, a0, a1); }
        if (adds) take_adds(accB, a0, a1);
    }
    __syncthreads();
    BI_STAMP(3);

    Other oA, oB;       // what to do FOR tile A / FOR tile B while the other multiplies
    oA.plane_acc = sA + i * HB + fbase;   oA.plane_park = sA + grow_l * HB + c4;   oA.fin = reinterpret_cast<float *>(sA) + i * HS + fbase;   oA.park_act = pact;
    oB.plane_acc = sBt + i * HB + fbase;  oB.plane_park = sBt + grow_l * HB + c4;  oB.fin = reinterpret_cast<float *>(sBt) + i * HS + fbase;  oB.park_act = pact;
    const __bf16 *paA = sA + i * HB + 8 * h, *paB = sBt + i * HB + 8 * h;
    const int L = p.n_layers;
    auto bias_init = [&](f32x16 &acc, f32x16 &acc1, int l) __attribute__((always_inline)) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(p.b + l * NP + fbase + 8 * gq);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[4 * gq + e] = b4[e]; if (SP == 2) acc1[4 * gq + e] = 0.f; }
        }
    };

    (void)L;                                    // (the launcher admits three-layer MLPs only: the phases below are unrolled)
    constexpr unsigned WB = 2u * BLOCK6;        // bytes of one layer's block of the stream
    // layer 0
    m_block<2, false, SP>(paA, W, rs, lo_b, 0u, accA, accA1, accB, accB1, xB, oB);             // for B: park
    BI_STAMP(4);
    __syncthreads();
    BI_STAMP(5);
    m_block<1, true, SP>(paB, W, rs, lo_b, WB, accB, accB1, accA, accA1, xB, oA);               // for A: epilogue of layer 0; refill with layer 1
    bias_init(accA, accA1, 1);
    BI_STAMP(6);
    __syncthreads();
    BI_STAMP(7);
    // layer 1
    m_block<1, false, SP>(paA, W, rs, lo_b, 0u, accA, accA1, accB, accB1, xB, oB);             // for B: epilogue of layer 0
    bias_init(accB, accB1, 1);
    BI_STAMP(8);
    __syncthreads();
    BI_STAMP(9);
    m_block<1, true, SP>(paB, W, rs, lo_b, 2u * WB, accB, accB1, accA, accA1, xB, oA);          // for A: epilogue of layer 1; refill with layer 2
    bias_init(accA, accA1, 2);
    BI_STAMP(10);
    __syncthreads();
    BI_STAMP(11);
    // layer 2
    m_block<1, false, SP>(paA, W, rs, lo_b, 0u, accA, accA1, accB, accB1, xB, oB);             // for B: epilogue of layer 1
    bias_init(accB, accB1, 2);
    BI_STAMP(12);
    __syncthreads();
    BI_STAMP(13);
    m_block<3, false, SP>(paB, W, rs, lo_b, 0u, accB, accB1, accA, accA1, xB, oA);             // for A: last layer's fp32 rows
    BI_STAMP(14);
    __syncthreads();
    BI_STAMP(15);
    // ---- B's last layer -> fp32 rows; then LayerNorm / activation / stores of both tiles, rows split over the waves
    {
        float *fb = oB.fin;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            f32x4 x;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = SP == 2 ? fmaf(accB1[4 * gq + e], F16_LO_UNSCALE, accB[4 * gq + e]) : accB[4 * gq + e];
            *reinterpret_cast<f32x4 *>(fb + 8 * gq) = x;
        }
    }
    __syncthreads();
    BI_STAMP(20);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float *sH = reinterpret_cast<float *>(t == 0 ? sA : sBt);
        // wave w owns rows [8 w, 8 w + 8): lane = part * 8 + row_local, each part = 16 consecutive columns
        const int rloc = lane & 7, part = lane >> 3, myrow = wave * 8 + rloc, cb = part * 16;
        float *rowp = sH + myrow * HS + cb;
        float x[16];
#pragma unroll
        for (int c = 0; c < 16; c += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(rowp + c);
            x[c] = v[0]; x[c + 1] = v[1]; x[c + 2] = v[2]; x[c + 3] = v[3];
        }
        if (p.gamma) {
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) sum += x[c];
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) sum += __shfl_xor(sum, o);
            const float mean = sum * (1.0f / NP);
            float var = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) { const float dl = x[c] - mean; var += dl * dl; }
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) var += __shfl_xor(var, o);
            const float rstd = rsqrtf(var * (1.0f / NP) + p.eps);
#pragma unroll
            for (int c = 0; c < 16; c += 4) {
                const f32x4 g4 = *reinterpret_cast<const f32x4 *>(p.gamma + cb + c), b4 = *reinterpret_cast<const f32x4 *>(p.beta + cb + c);
#pragma unroll
                for (int u = 0; u < 4; ++u) x[c + u] = fmaf((x[c + u] - mean) * rstd, g4[u], b4[u]);
            }
        }
        if (p.act == G4C_ACT_SELU) {
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = g4c::selu_f(x[c]);
        } else if (p.act == G4C_ACT_TANH) {
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = g4c::tanh_f(x[c]);
        }
        if (AGG || BX6I_ROW_STORES) {
#pragma unroll
            for (int c = 0; c < 16; c += 4) {
                f32x4 v;
                v[0] = x[c]; v[1] = x[c + 1]; v[2] = x[c + 2]; v[3] = x[c + 3];
                *reinterpret_cast<f32x4 *>(rowp + c) = v;
            }
        }
        if (!BX6I_ROW_STORES && p.out && myrow < nrow[t]) {
            const long long orow = (!AGG && p.out_idx) ? p.out_idx[row0[t] + myrow] : row0[t] + myrow;      // (g4c_mlp_forward's out_idx)
            float *op = p.out + orow * p.out_ld + cb;
#pragma unroll
            for (int c = 0; c < 16; c += 4) {
                f32x4 v;
                v[0] = x[c]; v[1] = x[c + 1]; v[2] = x[c + 2]; v[3] = x[c + 3];
                *reinterpret_cast<f32x4 *>(op + c) = v;
            }
        }
    }
    BI_STAMP(21);
    if (!AGG && BX6I_ROW_STORES && p.out) {
        // whole rows per store instruction from the LDS copy (as with AGG below); out_idx scatters them (g4c_mlp_forward's out_idx)
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float *sH = reinterpret_cast<const float *>(t == 0 ? sA : sBt);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = it * 8 + (tid >> 5), c = (tid & 31) * 4;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(sH + r * HS + c);
                if (r < nrow[t]) {
                    const long long orow = p.out_idx ? p.out_idx[row0[t] + r] : row0[t] + r;
                    *reinterpret_cast<f32x4 *>(p.out + orow * p.out_ld + c) = v;
                }
            }
        }
    }
    if (AGG) {
        // aggregation of the targets whose messages the tiles hold (rows in CSR order): same summation order and the same mean
        // formula as segment_reduce_kernel, so the result is bit-identical to the separate launch
        __syncthreads();
        if (BX6I_ROW_STORES && p.out) {
            // the finished rows are in LDS for the reduction anyway: store them from there, 32 lanes along a row (every store
            // instruction of a wave writes two complete 512-byte rows instead of a 16-byte piece of each 64-byte chunk of 8 rows)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float *sH = reinterpret_cast<const float *>(t == 0 ? sA : sBt);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int r = it * 8 + (tid >> 5), c = (tid & 31) * 4;
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(sH + r * HS + c);
                    if (r < nrow[t]) *reinterpret_cast<f32x4 *>(p.out + (long long)(row0[t] + r) * p.out_ld + c) = v;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (nrow[t] == 0) continue;
            const float *sH = reinterpret_cast<const float *>(t == 0 ? sA : sBt);
            const int tile = 2 * pair + t;
            const int s0 = p.tile_seg[tile], s1 = p.tile_seg[tile + 1];
            const int col = tid & (NP - 1);
            for (int sg = s0 + (tid >> 7); sg < s1; sg += 2) {
                const int b = p.seg_off[sg] - row0[t], e = p.seg_off[sg + 1] - row0[t];
                float a = 0.f;
                for (int r = b; r < e; ++r) a += sH[r * HS + col];
                if (p.agg_mean) a /= (float)((e - b) > 1 ? (e - b) : 1);
                p.agg[(long long)sg * p.agg_ld + col] = a;
            }
        }
    }
}

}  // namespace

namespace g4cm {

// 0 off, 1 (default) launches of at least BX6I_MIN_ROWS rows (the kernel needs a full machine of its two workgroups per CU: same-box
// crossover against the tile kernel ~300 k rows), 2 every launch it can take (tests)
static int g_bx6i = -1;
int bx6i_enable(int on) {
    if (g_bx6i < 0) g_bx6i = 1;
    const int old = g_bx6i;
    if (on >= 0) g_bx6i = on > 2 ? 2 : on;
    return old;
}

bool bx6i_eligible(const Params &p, bool round1, bool agg, bool save, bool f16x2, long long row_count) {
    // f16x3 mode (three workgroups per CU, shorter pairs): the kernel is ahead from ~20 k rows (6k-node mesh +1 %, 12.5k-node mesh and 2-scale 10k-node mesh +5 %, 25k / 50k-node meshes +6 / +10 %, level-2 launches of the 100k mesh +0.5 %
    // of the step; the interior launches of a 2- / 4-way partition); bf16x6 mode (two workgroups per CU): from ~300 k
    constexpr long long BX6I_MIN_ROWS = 400000;
    const long long min_rows = BX6I_MIN_ROWS;
    const int mode = bx6i_enable(-1);
    // (the f16x3 stream goes to the weight-stationary kernel, mlp_ws.hip, which also tracks the fp16 range; this kernel's two-way
    // instantiation sits at its register limit — one more live register and it spills a hundred — and is no longer launched)
    if (!mode || round1 || save || f16x2) return false;
    if (mode == 1 && row_count < min_rows) return false;
    if (p.n_src != 1 || p.n_nar != 0 || (p.n_add != 0 && p.n_add != 2) || p.n_heads) return false;
    if (p.n_layers != 3 || p.n_out != NP || p.resid || p.out_bf16) return false;
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

int bx6i_launch(const Params &p, bool agg, bool f16x2, hipStream_t st) {
    const int n_pairs = (p.n_tiles + 1) / 2;
    if (n_pairs == 0) return G4C_OK;
    const dim3 grid(n_pairs), blk(256);
#define G4C_BX6I_LAUNCH(AGG, SP)                                                                     \
    do {                                                                                             \
        if (p.src[0].idx) mlp_bx6i_kernel<AGG, SP, false><<<grid, blk, 0, st>>>(p);                  \
        else mlp_bx6i_kernel<AGG, SP, true><<<grid, blk, 0, st>>>(p);                                \
    } while (0)
    if (f16x2) return G4C_EUNSUPPORTED;
    if (agg) G4C_BX6I_LAUNCH(true, 3); else G4C_BX6I_LAUNCH(false, 3);
#undef G4C_BX6I_LAUNCH
    return g4c::check_launch("g4c_mlp_forward (bx6i)");
}

}  // namespace g4cm

extern "C" int g4c_mlp_bx6i_enable(int on) { return g4cm::bx6i_enable(on); }
