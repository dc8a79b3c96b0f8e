// Persistent, role-specialised form of the fused MLP ("px6"): gather -> concat -> Linear/SELU chain -> LayerNorm -> activation
// (-> residual, -> heads, -> per-target aggregation) -> store, same arithmetic as mlp_bx6_kernel (exact three-way bf16 split
// of both operands, six partial products per multiply-add on v_mfma_f32_32x32x16_bf16, fp32 accumulate), reorganised so
// that ONE weight fetch serves 128 rows and nothing on the matrix pipe's critical path is a memory round trip.
//
// Replaces the same reference op sequences as mlp_fused.hip: MLP.forward (graphs4cfd/nn/blocks.py:117-144) with the
// torch.cat / index ops in front of it and the F.selu / tanh / residual behind it (nn/blocks.py:181,185,229,285,328,332;
// nn/mus_gnn.py:178-218), and — with AGG — the scatter(e', col, reduce) of GNBlock / EdgeMP (nn/blocks.py:183,330).
//
// Structure (one workgroup of 8 waves per CU, persistent over a contiguous range of the launch's 32-row units):
//   * waves 0-3 are MATRIX waves, waves 4-7 HELPER waves; wave ct and wave ct + 4 share a SIMD and a column tile
//     (output features [32 ct, 32 ct + 32) of every 128-wide block).
//   * a launch is a cyclic program of STAGES, one per 128 x 128 weight block: the layer-0 blocks (one per weighted input
//     block), the hidden / last layers, the heads.  A workgroup works on a tile of four row tiles (rt, <= 32 rows each);
//     a matrix wave keeps its 32-column slice of the current stage's block in registers (8 steps x 3 planes x 4 VGPRs =
//     96) for all four row tiles and refills each step's slot with the NEXT stage's weights right after its last use:
//     24 KB of L2 -> register traffic per wave per 128 rows (the 32-row-tile kernel: 24 KB per 32 rows, 2 steps of
//     look-ahead), a whole stage of prefetch distance, and weight fetches are the ONLY vector-memory loads these waves
//     wait for (loads return in order per wave: a short-latency wait behind a long-latency prefetch would serialise).
//   * time is cut into intervals by one workgroup barrier each.  In interval k the matrix waves run the 48 MFMAs of unit
//     k = (tile, stage, rt) and drop the accumulators into an LDS hand-over buffer; the helper waves meanwhile finish
//     unit k - 1: bias / gathered additive terms / SELU / operand split into the rt's planes, or LayerNorm + whole-row
//     stores (+ aggregation) of a finished rt, or parking the next input rows.  All global loads of the helpers are issued
//     several intervals before they are consumed, in consumption order (row indices a tile ahead, input rows a tile
//     ahead, gathered additive rows two intervals ahead), so their waits never stall on a younger request.
//   * operand planes live in LDS per row tile as [32 rows][3 planes][136 bf16] (816 B per row: conflict-free
//     ds_read_b128 of the B fragments).  A finished tile's fp32 rows (for LayerNorm / coalesced whole-row stores) alias the
//     SAME rows' plane bytes, so no wave overwrites another wave's pending rows; layers are computed in place.
//   * AGG: row tiles hold whole CSR segments (g4c_plan_tiles); each helper wave normalises the rows of its own segments
//     and then sums them in row order from LDS -> bit-identical to g4c_segment_reduce, no second pass over the messages,
//     and the message rows themselves need not be stored at all (out == nullptr: the last MP layer of a level, whose
//     edge output the reference discards, nn/mus_gnn.py:199-200).
//
// MFMA-bound: per row and block 6 x 2 x 128 x 128 bf16 FLOP; the matrix waves issue MFMAs back to back.
#include "mlp_common.h"
#include <cstdlib>
using namespace g4cm;

// -DG4C_PX_TIMING: cycle stamps of the first phases of workgroup 0 (one lane per group), read back by g4c_px_read_stamps
#ifdef G4C_PX_TIMING
__device__ unsigned long long g4c_px_stamps[2 * 1024];
extern "C" int g4c_px_read_stamps(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g4c_px_stamps), sizeof(unsigned long long) * n);
}
#define PX_STAMP() do { if (blockIdx.x == 0 && ct == 0 && lane == 0 && stamp_k < 1024) g4c_px_stamps[role * 1024 + stamp_k] = __builtin_readcyclecounter(); ++stamp_k; } while (0)
__device__ unsigned long long g4c_px_sub[256 * 8];
extern "C" int g4c_px_read_sub(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g4c_px_sub), sizeof(unsigned long long) * n);
}
// helper-wave sub-stamps: slot s of the current interval (stamp_k / 2)
#define PX_SUB(s) do { if (blockIdx.x == 0 && ct == 0 && lane == 0 && role == 1 && (stamp_k >> 1) < 256) g4c_px_sub[(stamp_k >> 1) * 8 + (s)] = __builtin_readcyclecounter(); } while (0)
#else
#define PX_STAMP() do {} while (0)
#define PX_SUB(s) do {} while (0)
#endif

namespace {

constexpr int PLB = HB * 2;            // bytes of one plane row (136 bf16)
constexpr int ROWB = 3 * PLB;          // bytes per row of a row-tile region: three planes (or the fp32 row, 512 B)
constexpr int RTB = 32 * ROWB;         // one row-tile region
constexpr int NARW_MAX = 16;           // narrow input columns whose W1^T rows are staged in LDS
constexpr int HOROW = 132 * 4;            // row pitch of the start-value form of a hand-over buffer: [32 rows][132 floats] (conflict-free both ways)
constexpr int HOB = 32 * HOROW;           // one hand-over buffer: start values (helpers -> matrix waves, row-major), or a head's
                                          // accumulators (matrix waves -> helpers, [ct][gq][lane] x 16 bytes = 16 KB)
constexpr int IXB = 2 * 2 * 4 * 32 * 4;   // gather rows of the additive terms: [tile parity][source][rt][32 rows] int
constexpr int STB = 2 * 2 * 128 * 4;      // LayerNorm partial statistics: [unit parity][pass][column tile][32 rows] float
constexpr int TRC = 1024;                 // AGG: first rows / first segments of this workgroup's units, cached in LDS (beyond: global)
constexpr int TRB = 2 * (TRC + 1) * 4;
constexpr int PX_LDS = 4 * RTB + 2 * HOB + ((G4C_MAX_LAYERS + 1) * NP + 2 * NP + NARW_MAX * NP) * 4 + 16 + IXB + STB + TRB;

__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// barrier among the four HELPER waves inside an interval (the matrix waves do not take part): an LDS arrival counter;
// every helper wave calls it the same number of times, `epoch` counts the arrivals expected
__device__ __forceinline__ void group_sync(unsigned *cnt, unsigned &epoch, int lane) {
    epoch += 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < epoch) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// Weight fragment refill IN PLACE: the 16 bytes land in the registers that held the slot's previous fragment ("+v": one
// register tuple for the slot's whole life).  Written as a compiler-visible load (W[s] = ldw(...)), hipcc gives every
// refilled fragment registers of its own (96 more), copies them over at the end of the stage behind a vmcnt(0) and spills;
// the price of the asm form is that the wait before the first use is ours (w_wait).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void ldw_inplace(bf16x8 &w, u32x4_t rsrc, unsigned voff, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "+v"(w) : "v"(voff), "s"(rsrc), "s"(soff), "n"(OFF) : "memory");
}
// all refills of this wave have landed (its only vector-memory loads); the fragments are operands of the statement, so no
// MFMA that reads them can be scheduled above it
template <int SP>
__device__ __forceinline__ void w_wait(bf16x8 (&W)[8][SP]) {
    if constexpr (SP == 3) {
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(W[0][0]), "+v"(W[0][1]), "+v"(W[0][2]), "+v"(W[1][0]), "+v"(W[1][1]), "+v"(W[1][2]),
                       "+v"(W[2][0]), "+v"(W[2][1]), "+v"(W[2][2]), "+v"(W[3][0]), "+v"(W[3][1]), "+v"(W[3][2]),
                       "+v"(W[4][0]), "+v"(W[4][1]), "+v"(W[4][2]), "+v"(W[5][0]), "+v"(W[5][1]), "+v"(W[5][2]),
                       "+v"(W[6][0]), "+v"(W[6][1]), "+v"(W[6][2]), "+v"(W[7][0]), "+v"(W[7][1]), "+v"(W[7][2]));
    } else {
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(W[0][0]), "+v"(W[1][0]), "+v"(W[2][0]), "+v"(W[3][0]), "+v"(W[4][0]), "+v"(W[5][0]), "+v"(W[6][0]),
                       "+v"(W[7][0]));
    }
}
template <int SP>
__device__ __forceinline__ void refill_step(bf16x8 (&W)[8][SP], int s, u32x4_t rsrc, unsigned lo_b, unsigned soff) {
    ldw_inplace<0>(W[s][0], rsrc, lo_b, soff);
    if constexpr (SP == 3) {
        ldw_inplace<1024>(W[s][1], rsrc, lo_b, soff);
        ldw_inplace<2048>(W[s][2], rsrc, lo_b, soff);
    }
}

template <int SP>
__device__ __forceinline__ void put_split(unsigned char *d, f32x4 y) {
    bf16x4 vh, vm, vl;
    split3x4<SP>(y, vh, vm, vl);
    *reinterpret_cast<bf16x4 *>(d) = vh;
    if (SP == 3) {
        *reinterpret_cast<bf16x4 *>(d + PLB) = vm;
        *reinterpret_cast<bf16x4 *>(d + 2 * PLB) = vl;
    }
}

// exact three-way bf16 split of two fp32 values -> one packed pair per plane
template <int SP>
__device__ __forceinline__ void put_split2(unsigned char *d, f32x2 y, bool wr = true) {
    if (SP == 1) {
        bf16x2 b;
        b[0] = (__bf16)y[0]; b[1] = (__bf16)y[1];
        if (wr) *reinterpret_cast<bf16x2 *>(d) = b;
        return;
    }
    f32x2 hf, mf, lf;
    const unsigned hu = pack_bf16(y, hf);
    const f32x2 r1 = y - hf;
    const unsigned mu = pack_bf16(r1, mf);
    const f32x2 r2 = r1 - mf;
    const unsigned lu = pack_bf16(r2, lf);
#ifdef G4C_PX_ABLATE_EW                 // timing experiment: no LDS writes of the epilogue
    asm volatile("" :: "v"(hu), "v"(mu), "v"(lu));
#else
    if (wr) {
        *reinterpret_cast<unsigned *>(d) = hu;
        *reinterpret_cast<unsigned *>(d + PLB) = mu;
        *reinterpret_cast<unsigned *>(d + 2 * PLB) = lu;
    }
#endif
}

// SELU of two values
__device__ __forceinline__ f32x2 selu2(f32x2 x) {
    const float sa = 1.6732632423543772848170429916717f * 1.0507009873554804934193349852946f;
    const float scale = 1.0507009873554804934193349852946f;
    f32x2 t, m;
#pragma unroll
    for (int e = 0; e < 2; ++e) { m[e] = fmaxf(x[e], 0.f); t[e] = fminf(x[e], 0.f); }
    t = t * 1.4426950408889634f;
#pragma unroll
    for (int e = 0; e < 2; ++e) t[e] = __builtin_amdgcn_exp2f(t[e]);
    return m * scale + (t * sa - sa);
}

// hidden-layer epilogue of one PAIR of an accumulator (this lane: row i, features fbase + 8 gq + 2 pair + {0, 1}): SELU, exact
// operand split, planes.  Eight of them per unit, one per k step of the next unit's matrix phase.
template <int SP>
__device__ __forceinline__ void epilogue_pair(const f32x16 &accE, unsigned char *dE, int sl, bool wr = true) {
    const int gq = sl >> 1, pr = sl & 1;
    f32x2 x;
    x[0] = accE[4 * gq + 2 * pr]; x[1] = accE[4 * gq + 2 * pr + 1];
#ifdef G4C_PX_ABLATE_EV                 // timing experiment: no SELU / split arithmetic
    *reinterpret_cast<f32x2 *>(dE + 16 * gq + 4 * pr) = x;
#else
    put_split2<SP>(dE + 16 * gq + 4 * pr, selu2(x), wr);
#endif
}

// Last-layer epilogue of one accumulator IN THE MATRIX WAVE (this lane: row i, features fbase + 8 gq + e): LayerNorm from the
// row statistics the four matrix waves exchanged through LDS at the end of the unit's interval, activation, 16-byte stores
// through a buffer descriptor whose bounds end at the row tile's last row (rows beyond it are dropped by the hardware: no
// exec-mask branch inside the MFMA-interleaved region), and what the next consumer of the row tile needs in LDS:
// MODE 0 nothing, 1 the operand planes (the heads multiply the finished rows), 2 the fp32 rows (the helpers aggregate them).
struct FinCtx {
    float mean, rstd;
    __amdgpu_buffer_rsrc_t orsrc;      // the unit's output rows [row0, row0 + n)
    unsigned voff;                     // this lane's byte offset: row i, feature fbase
    unsigned char *reg;                // this lane's row of the row-tile region
    const float *gb;                   // sGB + fbase (gamma; beta NP floats further)
    int fbase;
};
template <int SP, int MODE, int ACT>
__device__ __forceinline__ void final_pair(const f32x16 &accE, int sl, const FinCtx &c, f32x2 &hold) {
    const int gq = sl >> 1, pr = sl & 1;
    const f32x2 g2 = *reinterpret_cast<const f32x2 *>(c.gb + 8 * gq + 2 * pr);
    const f32x2 b2 = *reinterpret_cast<const f32x2 *>(c.gb + NP + 8 * gq + 2 * pr);
    f32x2 x;
    x[0] = accE[4 * gq + 2 * pr]; x[1] = accE[4 * gq + 2 * pr + 1];
    f32x2 y = ((x - c.mean) * c.rstd) * g2 + b2;
    if (ACT == G4C_ACT_SELU) y = selu2(y);
    if (MODE == 1) put_split2<SP>(c.reg + 2 * (c.fbase + 8 * gq + 2 * pr), y);
    if (pr == 0) {
        hold = y;
    } else {
        f32x4 q;
        q[0] = hold[0]; q[1] = hold[1]; q[2] = y[0]; q[3] = y[1];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, q), c.orsrc, (int)(c.voff + 32u * gq), 0, 0);
        if (MODE == 2) *reinterpret_cast<f32x4 *>(c.reg + 4 * (c.fbase + 8 * gq)) = q;
    }
}

// M phase: acc += W(block) * planes(rt) for one row tile.  W[s][pl]: this wave's stationary weight fragments (A operand);
// B fragments of step s + 1 are read from LDS before the six MFMAs of step s issue.  REFILL: slot s is reloaded with the
// next stage's fragments right after its last use.  EPI: the epilogue of the PREVIOUS unit's accumulator accE (bias already
// in it) is interleaved with this unit's MFMAs — one eighth per step, spread over the MFMA issue slots by the scheduler
// (sched_group_barrier: 1 MFMA, then up to 4 vector ALU instructions, ...): the matrix pipe runs while the vector ALUs work on
// the previous tile.  EPI 1: hidden layer (SELU, operand split, planes), EPI 2: last layer (final_pair).
template <int SP, bool REFILL, int EPI, int MODE = 0, int ACT = 0>
__device__ __forceinline__ void m_phase(const unsigned char *pa, bf16x8 (&W)[8][SP], f32x16 &acc, u32x4_t rs,
                                        unsigned lo_b, unsigned nxt_b, const f32x16 &accE, unsigned char *dE, const FinCtx &fc) {
    bf16x8 cur[SP], nx[SP];
    f32x2 hold = {0.f, 0.f};
#pragma unroll
    for (int pl = 0; pl < SP; ++pl) cur[pl] = *reinterpret_cast<const bf16x8 *>(pa + pl * PLB);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        // ---- one scheduling region per step: next fragments, this step's MFMAs, one eighth of the previous unit's epilogue
        if (s < 7) {
#pragma unroll
            for (int pl = 0; pl < SP; ++pl) nx[pl] = *reinterpret_cast<const bf16x8 *>(pa + pl * PLB + 32 * (s + 1));
        }
        if (!EPI) __builtin_amdgcn_sched_barrier(0);      // (the next step's fragments are in flight before this step's MFMAs issue)
        if (EPI == 1) epilogue_pair<SP>(accE, dE, s);
        if (EPI == 2) final_pair<SP, MODE, ACT>(accE, s, fc, hold);
#ifdef G4C_PX_ABLATE_MFMA               // timing experiment: one MFMA per step instead of six
        if (false) {
#else
        if (SP == 3) {          // small terms first
#endif
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[s][0], cur[SP - 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[s][SP - 1], cur[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[s][SP / 2], cur[SP / 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[s][0], cur[SP / 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[s][SP / 2], cur[0], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[s][0], cur[0], acc, 0, 0, 0);
        if (EPI) {
            // issue order inside the region: the fragment reads, then each MFMA followed by a few epilogue instructions
            __builtin_amdgcn_sched_group_barrier(0x100, SP + (EPI == 2 ? 2 : 0), 0);   // DS read (+ gamma / beta)
#pragma unroll
            for (int m = 0; m < (SP == 3 ? 6 : 1); ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, SP == 3 ? 4 : 12, 0);   // VALU
            }
            __builtin_amdgcn_sched_group_barrier(0x200, SP, 0);                 // DS write
        }
        // ---- (the refill loads stay BEHIND the MFMAs that read the slot: otherwise the new fragments get registers of their
        // own, 96 more, and are copied over at the end of the stage behind a full vmcnt(0))
        __builtin_amdgcn_sched_barrier(0);
        if (REFILL) {
            refill_step<SP>(W, s, rs, lo_b, nxt_b + 2u * s * STEP6);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (s < 7) {
#pragma unroll
            for (int pl = 0; pl < SP; ++pl) cur[pl] = nx[pl];
        }
    }
}

// MULTI: layer 0 has more than one weighted input block (its accumulators then persist over several stages: one per row
// tile; otherwise the matrix waves hold a single accumulator)
// HEADS: the finished rows are also the operand of head blocks (their planes are written by the last layer's epilogue)
template <int SP, bool AGG, bool MULTI, bool HEADS>
__global__ __launch_bounds__(512) void mlp_px6_kernel(const Params p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[PX_LDS];
    unsigned char *const sHO = lds + 4 * RTB;                                   // two hand-over buffers
    float *sBias = reinterpret_cast<float *>(lds + 4 * RTB + 2 * HOB);
    float *sGB = sBias + (G4C_MAX_LAYERS + 1) * NP;          // (one extra all-zero bias row: the heads)
    float *sNarW = sGB + 2 * NP;
    unsigned *sCnt = reinterpret_cast<unsigned *>(sNarW + NARW_MAX * NP);      // arrival counter of group_sync
    int *sIx = reinterpret_cast<int *>(sCnt + 4);
    float *sStat = reinterpret_cast<float *>(sIx + IXB / 4);
    int *sTR = reinterpret_cast<int *>(sStat + STB / 4);         // [TRC + 1] tile_rows, then [TRC + 1] tile_seg of units u_begin ...

    // the parameter block is read through a pointer to the kernarg segment
    typedef const __attribute__((address_space(4))) Params *ParamsPtr;
    ParamsPtr pp = (ParamsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    (void)p;
#define P (*pp)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave & 3, role = wave >> 2;         // role 0: matrix wave, 1: helper wave
    // lane-derived indices.  They are re-derived from an opaque copy of the lane id at the head of every interval (remat()):
    // otherwise hipcc hoists every address / mask expression built from them out of the persistent loop and keeps ~100
    // loop-invariant registers alive (spills)
    int i = lane & 31, h = lane >> 5;
    int fbase = ct * 32 + 4 * h;                       // this lane's accumulator features: fbase + 8 gq + e
    int prow = (lane >> 3) + 8 * ct;                   // park layout: row of the row tile, columns 32 q + c4 .. + 3
    int c4 = (lane & 7) * 4;
    int lane_v = lane;
    auto remat = [&]() __attribute__((always_inline)) {
        int l = lane;
        asm volatile("" : "+v"(l));
        lane_v = l; i = l & 31; h = l >> 5; fbase = ct * 32 + 4 * h; prow = (l >> 3) + 8 * ct; c4 = (l & 7) * 4;
    };

    const int n_src = P.n_src, n_layers = P.n_layers, n_heads = P.n_heads, n_add = P.n_add;
    const int NS = n_src + n_layers - 1 + n_heads;     // stages per tile
    const int st_l0 = n_src - 1;                       // stage of the last layer-0 block
    const int st_fin = n_src + n_layers - 2;           // stage of the last layer
    // this workgroup's contiguous range of 32-row units, in tiles of four
    int u_begin, u_end;
    {
        const int G = gridDim.x, q = P.n_tiles / G, rem = P.n_tiles % G, b = blockIdx.x;
        u_begin = b * q + (b < rem ? b : rem);
        u_end = u_begin + q + (b < rem ? 1 : 0);
    }
    const int iters = (u_end - u_begin + 3) >> 2;

    // ---- parameters into LDS (biases, LayerNorm affine, W1^T rows of the narrow input blocks)
    if (tid == 0) sCnt[0] = 0;
    unsigned sync_epoch = 0;
    for (int e = tid; e < (G4C_MAX_LAYERS + 1) * NP; e += 512) sBias[e] = (e < n_layers * NP) ? P.b[e] : 0.f;
    if (tid < 2 * NP) {          // (no LayerNorm: gamma = 1, beta = 0 and the epilogue uses mean 0, rstd 1)
        const int f = tid & (NP - 1), ff = f < P.n_out ? f : 0;
        sGB[tid] = P.gamma ? (tid < NP ? P.gamma[ff] : P.beta[ff]) : (tid < NP ? 1.f : 0.f);
    }
    {
        int base = 0;
        for (int a = 0; a < P.n_nar; ++a) {
            for (int e = tid; e < P.nar[a].width * NP; e += 512) sNarW[base * NP + e] = P.nar[a].w[e];
            base += P.nar[a].width;
        }
    }

    if (AGG) {
        const int cnt = (u_end - u_begin + 1) < (TRC + 1) ? (u_end - u_begin + 1) : (TRC + 1);
        for (int e = tid; e < cnt; e += 512) { sTR[e] = P.tile_rows[u_begin + e]; sTR[TRC + 1 + e] = P.tile_seg[u_begin + e]; }
        __syncthreads();
    }
    // ---- row-tile bookkeeping (wave-uniform)
    auto rt_info = [&](int j, int r, int &row0, int &n) __attribute__((always_inline)) {
        const int ul = 4 * j + r, u = u_begin + ul;
        row0 = 0; n = 0;
        if (j >= 0 && j < iters && u < u_end) {
            if (AGG) {
                if (ul < TRC) { row0 = __builtin_amdgcn_readfirstlane(sTR[ul]); n = __builtin_amdgcn_readfirstlane(sTR[ul + 1]) - row0; }
                else { row0 = P.tile_rows[u]; n = P.tile_rows[u + 1] - row0; }
            } else { row0 = (int)P.row_base + u * 32; n = (int)P.M - row0; n = n < 32 ? n : 32; }
        }
    };
    int stamp_k = 0;
    (void)stamp_k;

    const bool pre_on = (n_add > 0) || (P.n_nar > 0);    // layer 0 has terms the helpers sum into the accumulators' start values
    if (role == 0) {
        // =================================================================================== matrix waves
        // buffer descriptor of the packed weight stream (base, stride 0, no bounds, dword format)
        u32x4_t rs;
        {
            const unsigned long long wb = (unsigned long long)P.w;
            rs[0] = (unsigned)wb; rs[1] = (unsigned)(wb >> 32) & 0xffffu; rs[2] = 0x7fffffffu; rs[3] = 0x00020000u;
        }
        const unsigned lo_b = 2u * (unsigned)(ct * 8 * STEP6 + lane * 8);
        bf16x8 W[8][SP];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int pl = 0; pl < SP; ++pl) W[s][pl] = bf16x8{};
#pragma unroll
        for (int s = 0; s < 8; ++s) refill_step<SP>(W, s, rs, lo_b, 2u * s * STEP6);
        // accumulators: one per row tile when layer 0 accumulates over several stages, otherwise two used alternately (the
        // previous unit's is being split while the current one accumulates)
        constexpr int NACC = MULTI ? 4 : 2;
        f32x16 acc[NACC];
        wg_barrier();                                  // the first tile is parked
        int p3n = 0;                                   // rows of row tile 3 of the previous tile
        for (int j = 0; j <= iters; ++j) {
            int cn[4], crow0;
#pragma unroll
            for (int r = 0; r < 4; ++r) rt_info(j, r, crow0, cn[r]);
            for (int st = 0; st < NS; ++st) {
                if (j == iters && st > 0) break;
                const unsigned nxt_b = 2u * BLOCK6 * (unsigned)(st + 1 == NS ? 0 : st + 1);
                const bool zero = (st == 0) || (st > st_l0);
                const bool hand_over = (st > st_fin);          // heads: the helpers store the accumulators
                const bool hidden = (st >= st_l0) && (st < st_fin);
                const bool final_st = (st == st_fin);
                const int pst = st > 0 ? st - 1 : NS - 1;
                const bool phidden = (pst >= st_l0) && (pst < st_fin);     // stage of the unit before r = 0
                const bool pfinal = (pst == st_fin);
                // layer whose bias the accumulators of this stage start from (heads: the all-zero row)
                const float *bp = sBias + (st <= st_fin ? (st == 0 ? 0 : st - st_l0) : G4C_MAX_LAYERS) * NP + (ct * 32 + 4 * (lane >> 5));
                // one interval; the row tile is a compile-time constant (accumulators and prefetch registers are indexed by it)
                auto interval = [&](auto rc) __attribute__((always_inline)) {
                    constexpr int r = decltype(rc)::value;
                    if (j == iters && r >= 2) return;        // (the drain has two intervals)
                    PX_STAMP();
                    remat();
                    constexpr int pr = (r + 3) & 3;
                    constexpr int ia = MULTI ? r : (r & 1), ie = MULTI ? pr : ((r & 1) ^ 1);
                    if (r == 0) w_wait<SP>(W);              // this stage's weights (requested during the previous stage's last phase)
                    // the previous unit's accumulator still has its epilogue to go through: 1 hidden layer, 2 last layer
                    const int pn = (r == 0) ? ((st == 0) ? p3n : cn[3]) : cn[pr];
                    const bool pvalid = pn > 0 && (r > 0 || j > 0 || st > 0) && !(j == iters && r == 1);
                    const int ek = !pvalid ? 0 : ((r == 0) ? (phidden ? 1 : (pfinal ? 2 : 0)) : (hidden ? 1 : (final_st ? 2 : 0)));
                    unsigned char *dE = lds + pr * RTB + i * ROWB + 2 * fbase;
                    FinCtx fc;
                    fc.mean = 0.f; fc.rstd = 1.f; fc.voff = 0; fc.reg = lds + pr * RTB + i * ROWB; fc.gb = sGB + fbase; fc.fbase = fbase;
                    fc.orsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P.w), 0, 0, 0x00020000);      // (nothing in bounds: every store dropped)
                    if (ek == 2) {
                        int prow0, pnn;
                        rt_info((r == 0 && st == 0) ? j - 1 : j, pr, prow0, pnn);
                        if (P.gamma) {
                            const float *s0 = sStat + ((pr & 1) * 2 + 0) * 128 + i, *s1 = sStat + ((pr & 1) * 2 + 1) * 128 + i;
                            const float inv_n = 1.0f / (float)NP;
                            const float mean = (s0[0] + s0[32] + s0[64] + s0[96]) * inv_n;
                            const float var = fmaxf((s1[0] + s1[32] + s1[64] + s1[96]) * inv_n - mean * mean, 0.f);
                            fc.mean = mean; fc.rstd = rsqrtf(var + P.eps);
                        }
                        if (P.out) {
                            float *ob = P.out + (long long)prow0 * P.out_ld;
                            fc.orsrc = __builtin_amdgcn_make_buffer_rsrc(ob, 0, pnn * P.out_ld * 4, 0x00020000);
                            fc.voff = (unsigned)(i * P.out_ld + fbase) * 4u;
                        }
                    }
                    constexpr int MODE = AGG ? 2 : (HEADS ? 1 : 0);
                    const bool selu_out = P.act == G4C_ACT_SELU;
                    auto serial_epilogue = [&]() __attribute__((always_inline)) {
                        if (ek == 1) {
#pragma unroll
                            for (int sl = 0; sl < 8; ++sl) epilogue_pair<SP>(acc[ie], dE, sl);
                        } else if (ek == 2) {
                            f32x2 hold = {0.f, 0.f};
                            if (selu_out) {
#pragma unroll
                                for (int sl = 0; sl < 8; ++sl) final_pair<SP, MODE, G4C_ACT_SELU>(acc[ie], sl, fc, hold);
                            } else {
#pragma unroll
                                for (int sl = 0; sl < 8; ++sl) final_pair<SP, MODE, G4C_ACT_NONE>(acc[ie], sl, fc, hold);
                            }
                        }
                    };
                    const bool on = cn[r] > 0;
                    if (on) {
                        if (zero) {
#pragma unroll
                            for (int gq = 0; gq < 4; ++gq) {
                                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bp + 8 * gq);
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[ia][4 * gq + e] = b4[e];
                            }
                        }
                        if (st == st_l0 && pre_on) {
                            // gathered additive rows / narrow input blocks, summed by the helpers (hand-over layout)
                            const unsigned char *hs = sHO + (r & 1) * HOB + i * HOROW + fbase * 4;
#pragma unroll
                            for (int gq = 0; gq < 4; ++gq) {
                                const f32x4 v = *reinterpret_cast<const f32x4 *>(hs + gq * 32);
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[ia][4 * gq + e] += v[e];
                            }
                        }
                    }
                    const unsigned char *pa = lds + r * RTB + i * ROWB + 16 * h;
#ifndef G4C_PX_ABLATE_M                 // (timing experiment: matrix waves idle)
                    if (r == 3) {
                        // the phase that refills the weights is ONE code path whatever the row tile's state (an inactive row
                        // tile multiplies stale planes into an unused accumulator): alternatives around the refill make hipcc
                        // keep a second copy of all 96 weight registers.  The previous unit's epilogue therefore runs in front
                        // of it, not interleaved
                        serial_epilogue();
                        if (cn[0] > 0) m_phase<SP, true, 0>(pa, W, acc[ia], rs, lo_b, nxt_b, acc[ie], dE, fc);
                    } else if (on) {
                        if (ek == 1) m_phase<SP, false, 1>(pa, W, acc[ia], rs, lo_b, nxt_b, acc[ie], dE, fc);
                        else if (ek == 2 && selu_out) m_phase<SP, false, 2, MODE, G4C_ACT_SELU>(pa, W, acc[ia], rs, lo_b, nxt_b, acc[ie], dE, fc);
                        else if (ek == 2) m_phase<SP, false, 2, MODE, G4C_ACT_NONE>(pa, W, acc[ia], rs, lo_b, nxt_b, acc[ie], dE, fc);
                        else m_phase<SP, false, 0>(pa, W, acc[ia], rs, lo_b, nxt_b, acc[ie], dE, fc);
                    } else {
                        serial_epilogue();
                    }
#endif
                    if (on && final_st && P.gamma) {
                        // this wave's share of the row statistics (its 32 columns): sum and sum of squares, exchanged at the barrier
                        float s = 0.f, q = 0.f;
#pragma unroll
                        for (int e = 0; e < 16; ++e) { s += acc[ia][e]; q = fmaf(acc[ia][e], acc[ia][e], q); }
                        s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
                        if (h == 0) {
                            sStat[((r & 1) * 2 + 0) * 128 + ct * 32 + i] = s;
                            sStat[((r & 1) * 2 + 1) * 128 + ct * 32 + i] = q;
                        }
                    }
                    if (on && hand_over) {
                        unsigned char *d = sHO + (r & 1) * HOB + (ct * 4 * 64 + lane_v) * 16;
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            f32x4 x;
#pragma unroll
                            for (int e = 0; e < 4; ++e) x[e] = acc[ia][4 * gq + e];
                            *reinterpret_cast<f32x4 *>(d + gq * 1024) = x;
                        }
                    }
                    PX_STAMP();
                    wg_barrier();
                };
                interval(std::integral_constant<int, 0>{});
                interval(std::integral_constant<int, 1>{});
                interval(std::integral_constant<int, 2>{});
                interval(std::integral_constant<int, 3>{});
            }
            p3n = cn[3];
        }
        return;
    }

    // ======================================================================================= helper waves
    // Everything that is not a matrix product or a hidden-layer epilogue: input rows (fetched a tile ahead, parked into the
    // planes), gathered additive rows / narrow input blocks (summed into the matrix waves' start values), the last layer's
    // LayerNorm / stores / aggregation, head stores.  Per-row-tile state is derived from the loop position.
    //
    // Prefetch loads are issued UNCONDITIONALLY, once per interval and register set (the address is selected, not the load:
    // rows that are not needed are requested again or replaced by a dummy row).  A load under a wave-uniform condition
    // merges with the register's previous value at the join, hipcc copies it there, and the copy waits for the data — every
    // prefetch then costs its full memory latency (measured: 3000 cycles per row fetch).
#ifndef G4C_PX_HELPER_PRIO
#define G4C_PX_HELPER_PRIO 3
#endif
    // The SIMD's arbiter prefers the older wave: without this, the matrix wave's dense stretches (a serial epilogue, accumulator
    // initialisation) keep the helper wave from issuing for thousands of cycles (seen as a late first stamp after the barrier);
    // the helpers' work is the critical path of the heavy intervals, the matrix wave only needs one issue slot per MFMA
    __builtin_amdgcn_s_setprio(G4C_PX_HELPER_PRIO);
    f32x4 xp[4][4];                      // xp[r] = this lane's 16 values of row `prow` of the rows row tile r is parked with next
    f32x4 ad[2][2][4];                   // ad[unit parity][source][gq]: gathered additive rows, in flight for two intervals
    // gathers go through buffer descriptors (base in SGPRs, one 32-bit byte offset per lane, the column offsets as immediates):
    // the helper wave shares its SIMD's issue slots with a matrix wave that issues MFMAs and epilogue arithmetic back to back,
    // so every instruction of 64-bit address arithmetic it does not execute is ~10 cycles off the interval
    const __amdgpu_buffer_rsrc_t ars0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(n_add > 0 ? P.add[0].ptr : P.w), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(n_add > 1 ? P.add[1].ptr : P.w), 0, 0x7fffffff, 0x00020000);
    const int ald0 = n_add > 0 ? P.add[0].ld * 4 : 0, ald1 = n_add > 1 ? P.add[1].ld * 4 : 0;
    // fields of the weighted input blocks, read from the kernarg segment once (selected by comparisons: a runtime index into
    // a local array would live in scratch memory)
    int sld_[G4C_MAX_SRC], scol_[G4C_MAX_SRC], swid_[G4C_MAX_SRC], sact_[G4C_MAX_SRC];
    const int *sidx_[G4C_MAX_SRC];
    const float *sptr_[G4C_MAX_SRC];
#pragma unroll
    for (int k = 0; k < G4C_MAX_SRC; ++k) {
        const int kk = k < n_src ? k : 0;
        sld_[k] = P.src[kk].ld; scol_[k] = P.src[kk].col0; swid_[k] = P.src[kk].width; sact_[k] = P.src[kk].pre_act;
        sidx_[k] = P.src[kk].idx; sptr_[k] = P.src[kk].ptr;
    }
#define PX_PICK(a, s) ((s) == 0 ? a[0] : ((s) == 1 ? a[1] : ((s) == 2 ? a[2] : a[3])))
    auto bld16 = [&](f32x4 &dst, __amdgpu_buffer_rsrc_t rsrc, int voff, int imm) __attribute__((always_inline)) {
        dst = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + imm, 0, 0));
    };
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) { xp[r][q] = f32x4{0.f, 0.f, 0.f, 0.f}; ad[r & 1][r >> 1][q] = xp[r][q]; }
    auto ld16 = [&](f32x4 &dst, const float *ptr) __attribute__((always_inline)) { dst = *reinterpret_cast<const f32x4 *>(ptr); };
    auto fetch = [&](auto rc, int j, int s) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        int row0, n;
        rt_info(j, r, row0, n);
        const int rr = prow < n ? prow : n - 1;
        long long gr = row0 + rr;
        if (P.src[s].idx) gr = P.src[s].idx[gr];
        const float *rp = P.src[s].ptr + gr * P.src[s].ld + P.src[s].col0 + c4;
        const int width = P.src[s].width;
#pragma unroll
        for (int q = 0; q < 4; ++q) ld16(xp[r][q], rp + ((q * KC + c4 < width) ? q * KC : -c4));
    };
    // park source s of tile j into row tile r (rows >= n are clamped copies of the last row: never stored)
    auto park = [&](auto rc, int j, int s) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        const int width = PX_PICK(swid_, s);
        const bool act = PX_PICK(sact_, s) != 0;
        unsigned char *d = lds + r * RTB + prow * ROWB + 2 * c4;
        if (width == NP && act) {            // (the MP layers' input block: no column masks, SELU pending on the stored rows)
#pragma unroll
            for (int q = 0; q < 4; ++q) put_split<SP>(d + 2 * q * KC, selu4(xp[r][q]));
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = xp[r][q];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (q * KC + c4 + e < width) ? v[e] : 0.f;
            if (act) v = selu4(v);
            put_split<SP>(d + 2 * q * KC, v);
        }
    };
    // same, for the loop: always issued; (j, s) clamped to rows that exist (requesting rows again is harmless)
    auto fetch_always = [&](auto rc, int j, int s) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        int row0, n;
        rt_info(j, r, row0, n);
        if (n <= 0) { rt_info(0, 0, row0, n); s = 0; }         // (the first row tile always exists)
        const int rr = prow < n ? prow : n - 1;
        int gr = row0 + rr;
        const int *ix = PX_PICK(sidx_, s);
        if (ix) gr = ix[gr];
        const int width = PX_PICK(swid_, s);
        const int voff = (gr * PX_PICK(sld_, s) + PX_PICK(scol_, s) + c4) * 4;
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(PX_PICK(sptr_, s)), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int q = 0; q < 4; ++q) bld16(xp[r][q], srs, (q * KC + c4 < width) ? voff + q * KC * 4 : voff - c4 * 4, 0);
    };
    // the park that follows the latest one of a row tile whose most recent unit is (tile qj, stage qs)
    auto next_park = [&](int qj, int qs, int &nj2, int &ns2) __attribute__((always_inline)) {
        if (qs < st_l0) {
            if (qs + 2 <= st_l0) { nj2 = qj; ns2 = qs + 2; } else { nj2 = qj + 1; ns2 = 0; }
        } else if (qs < NS - 1) { nj2 = qj + 1; ns2 = 0; }
        else if (n_src > 1) { nj2 = qj + 1; ns2 = 1; }
        else { nj2 = qj + 2; ns2 = 0; }
    };

    // gather rows of the additive terms, staged in LDS a tile ahead: helper wave ct loads those of row tile ct
    // (lanes 0-31: source 0, lanes 32-63: source 1)
    auto load_ix = [&](int j) __attribute__((always_inline)) {
        int row0, n;
        rt_info(j, ct, row0, n);
        if (n > 0 && h < n_add) {
            const int gr = row0 + (i < n ? i : n - 1);
            sIx[(((j & 1) * 2 + h) * 4 + ct) * 32 + i] = P.add[h].idx ? P.add[h].idx[gr] : gr;
        }
    };
    // gathers of unit (j, st_l0, r) (need) or of a dummy row (the load itself is unconditional, see above)
    // (lane -> row prow of the row tile, 16-byte chunks at columns c4 + 32 q: eight lanes read 128 contiguous bytes of a row, an
    // instruction touches 8 rows — in the accumulator layout it would touch 32 rows in 32-byte pieces, four times the cache lines,
    // and these gathers share the CU's one address path with the matrix waves' weight refills and row stores)
    auto issue_adds = [&](auto rc, int j, bool need) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        const int *ixp = sIx + ((j & 1) * 2 * 4 + r) * 32 + prow;
        int row0a = 0, row1a = 0;
        if (need) { row0a = ixp[0]; row1a = n_add > 1 ? ixp[4 * 32] : 0; }
        int v0 = row0a * ald0 + c4 * 4, v1 = row1a * ald1 + c4 * 4;
#ifdef G4C_PX_TIMING
        asm volatile("" : "+v"(v0), "+v"(v1));
        __builtin_amdgcn_sched_barrier(0);
        PX_SUB(5);
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int q = 0; q < 4; ++q) bld16(ad[r & 1][0][q], ars0, v0, q * KC * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) bld16(ad[r & 1][1][q], ars1, v1, q * KC * 4);
    };
    // start values of unit (j, st_l0, r) beyond the bias: gathered rows + narrow input blocks, into the hand-over buffer
    auto presum = [&](auto rc, int j) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        int row0, n;
        rt_info(j, r, row0, n);
        f32x4 x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (n_add > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { x[q] = ad[r & 1][0][q]; if (n_add > 1) x[q] += ad[r & 1][1][q]; }
        }
        // narrow input blocks: x[row, k] * W1^T[k, :] in fp32 on the vector ALUs
        int base = 0;
        const int gr = row0 + (prow < n ? prow : n - 1);
        for (int a = 0; a < P.n_nar; ++a) {
            const float *xr = P.nar[a].ptr + (long long)gr * P.nar[a].ld;
            for (int kk = 0; kk < P.nar[a].width; ++kk) {
                const float xv = xr[kk];
                const float *wn = sNarW + (base + kk) * NP + c4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wn + q * KC);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[q][e] = fmaf(xv, w4[e], x[q][e]);
                }
            }
            base += P.nar[a].width;
        }
        unsigned char *d = sHO + (r & 1) * HOB + prow * HOROW + c4 * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4 *>(d + q * KC * 4) = x[q];
    };

    // ---- two intervals after the last layer of row tile r of tile fj (the matrix waves have normalised and stored its rows in the
    // interval between): the per-target aggregation from the fp32 rows they left in the row tile's region, then what refills the
    // region — the next tile's input rows (with heads the region holds the heads' operand planes until the last head is done)
    auto finish = [&](auto rc, int fj) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        int row0, n;
        rt_info(fj, r, row0, n);
        if (n <= 0) return;
        if (AGG) {
            // every helper wave adds up its 32 columns of the row tile's segments in row order (= g4c_segment_reduce):
            // lane -> (feature quad lane & 7, one of 8 segments at a time)
            const int ful = 4 * fj + r, fu = u_begin + ful;
            const int s0 = ful < TRC ? __builtin_amdgcn_readfirstlane(sTR[TRC + 1 + ful]) : P.tile_seg[fu];
            const int s1 = ful < TRC ? __builtin_amdgcn_readfirstlane(sTR[TRC + 2 + ful]) : P.tile_seg[fu + 1];
            const int fq = lane_v & 7;
            const unsigned char *rb = lds + r * RTB + 4 * (32 * ct + 4 * fq);
            for (int sg = s0 + (lane_v >> 3); sg < s1; sg += 8) {
                const int b = P.seg_off[sg] - row0, e = P.seg_off[sg + 1] - row0;
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                for (int row = b; row < e; ++row) a += *reinterpret_cast<const f32x4 *>(rb + row * ROWB);
                if (P.agg_mean) {
                    const float cnt = (float)((e - b) > 1 ? (e - b) : 1);
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[q] /= cnt;
                }
                *reinterpret_cast<f32x4 *>(P.agg + (long long)sg * P.agg_ld + ct * 32 + 4 * fq) = a;
            }
            group_sync(sCnt, sync_epoch, lane_v);      // every helper wave has read the rows before any of them parks over them
        }
        if (!HEADS) {
            int row0n, nn;
            rt_info(fj + 1, r, row0n, nn);
            if (nn > 0) park(rc, fj + 1, 0);
        }
    };

    // ---- prologue: first tile's source 0 rows, fetched and parked just in time; the next rows, indices and gathers in flight
    __builtin_amdgcn_sched_barrier(0);
    if (n_add > 0) { load_ix(0); load_ix(1); }
    auto pro_fetch = [&](auto rc, int s) __attribute__((always_inline)) {
        int row0, n;
        rt_info(0, decltype(rc)::value, row0, n);
        if (n > 0) fetch(rc, 0, s);
    };
    auto pro_park = [&](auto rc) __attribute__((always_inline)) {
        int row0, n;
        rt_info(0, decltype(rc)::value, row0, n);
        if (n > 0) park(rc, 0, 0);
    };
    pro_fetch(std::integral_constant<int, 0>{}, 0); pro_fetch(std::integral_constant<int, 1>{}, 0);
    pro_fetch(std::integral_constant<int, 2>{}, 0); pro_fetch(std::integral_constant<int, 3>{}, 0);
    pro_park(std::integral_constant<int, 0>{}); pro_park(std::integral_constant<int, 1>{});
    pro_park(std::integral_constant<int, 2>{}); pro_park(std::integral_constant<int, 3>{});
    // the rows of the first park inside the loop: the second input block of the first tile, or the next tile's rows
    {
        const int fj0 = n_src > 1 ? 0 : 1, fs0 = n_src > 1 ? 1 : 0;
        fetch_always(std::integral_constant<int, 0>{}, fj0, fs0); fetch_always(std::integral_constant<int, 1>{}, fj0, fs0);
        fetch_always(std::integral_constant<int, 2>{}, fj0, fs0); fetch_always(std::integral_constant<int, 3>{}, fj0, fs0);
    }
    if (n_add > 0) group_sync(sCnt, sync_epoch, lane_v);          // (the index rows are visible to the other helper waves)
    if (pre_on && st_l0 == 0) {
        // gathers of the first three units (every later unit: issued three intervals before its matrix phase), start values
        // of the first (every later unit: summed one interval before its matrix phase)
        auto pro_adds = [&](auto rc) __attribute__((always_inline)) {
            int row0, n;
            rt_info(0, decltype(rc)::value, row0, n);
            if (n_add > 0) issue_adds(rc, 0, n > 0);
        };
        pro_adds(std::integral_constant<int, 0>{}); pro_adds(std::integral_constant<int, 1>{});
        int row0, n;
        rt_info(0, 0, row0, n);
        if (n > 0) presum(std::integral_constant<int, 0>{}, 0);
        pro_adds(std::integral_constant<int, 2>{});            // (into the registers unit 0's gathers have just left)
    }
    wg_barrier();

    for (int j = 0; j <= iters; ++j) {
        // (indices of tile j + 1 go where tile j - 1's were: its last gathers were issued long ago)
        if (j > 0 && n_add > 0) load_ix(j + 1);
        for (int st = 0; st < NS; ++st) {
            if (j == iters && st > 0) break;
            // the previous / next stage pass (row tiles wrap around)
            int pj = j, pst = st - 1;
            if (pst < 0) { pst = NS - 1; pj = j - 1; }
            int nj = j, nst = st + 1;
            if (nst == NS) { nst = 0; nj = j + 1; }
            auto interval = [&](auto rc) __attribute__((always_inline)) {
                constexpr int r = decltype(rc)::value;
                if (j == iters && r >= 2) return;
                PX_STAMP();
                remat();
                PX_SUB(0);
                // ---------------------------------------------------------------- the previous unit (more layer-0 blocks to park, heads to store)
                if constexpr (MULTI || HEADS) {
                    constexpr int ur = (r + 3) & 3;
                    const std::integral_constant<int, ur> urc{};
                    const int uj = (r == 0) ? pj : j, ust = (r == 0) ? pst : st;
                    int urow0, un;
                    rt_info(uj, ur, urow0, un);
                    if (un > 0) {
                        if (ust < st_l0) {
                            park(urc, uj, ust + 1);                   // next input block of layer 0; the accumulator carries on
                        } else if (ust > st_fin) {
                            // head: plain product of the finished rows, stored as it is
                            const unsigned char *hs = sHO + (ur & 1) * HOB + (ct * 4 * 64 + lane_v) * 16;
                            float *ho = P.head_out[ust - st_fin - 1];
                            if (i < un) {
                                float *orow = ho + (long long)(urow0 + i) * P.head_ld + fbase;
#pragma unroll
                                for (int gq = 0; gq < 4; ++gq) *reinterpret_cast<f32x4 *>(orow + 8 * gq) = *reinterpret_cast<const f32x4 *>(hs + gq * 1024);
                            }
                            if (ust + 1 == NS) {
                                int row0n, nn;
                                rt_info(uj + 1, ur, row0n, nn);
                                if (nn > 0) park(urc, uj + 1, 0);
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                PX_SUB(1);
                // ---------------------------------------------------------------- start values of the next unit
                if (pre_on) {
                    constexpr int ar = (r + 1) & 3;
                    const int aj = (r == 3) ? nj : j, ast = (r == 3) ? nst : st;
                    if (ast == st_l0) {
                        int row0a, na;
                        rt_info(aj, ar, row0a, na);
                        if (na > 0) presum(std::integral_constant<int, ar>{}, aj);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                PX_SUB(2);
                // ---------------------------------------------------------------- gathers for the unit three ahead
                // (loads are issued UNCONDITIONALLY, with a selected address: measured alternatives — the same loads under the
                // condition, their registers declared dead first — save ~1000 cycles in the light intervals but make hipcc wait
                // for ALL requests in flight wherever a loaded register is consumed, which costs more in the heavy ones)
                if (n_add > 0) {
                    constexpr int gr = (r + 3) & 3;
                    const int gj = (r == 0) ? j : nj, gst = (r == 0) ? st : nst;
                    int row0g, ng;
                    rt_info(gj, gr, row0g, ng);
                    issue_adds(std::integral_constant<int, gr>{}, gj, gst == st_l0 && ng > 0);
                }
                PX_SUB(3);
                // ---------------------------------------------------------------- next input rows of the row tile two units back
                {
                    constexpr int qr = (r + 2) & 3;
                    const int qj = (r < 2) ? pj : j, qs = (r < 2) ? pst : st;
                    // (its last layer ran two intervals ago, the matrix waves' epilogue of it one interval ago)
                    if (qj >= 0 && qs == st_fin) finish(std::integral_constant<int, qr>{}, qj);
                    __builtin_amdgcn_sched_barrier(0);
                    int fj2 = n_src > 1 ? 0 : 1, fs = n_src > 1 ? 1 : 0;        // (before its first unit: what the prologue requested)
                    if (qj >= 0) next_park(qj, qs, fj2, fs);
                    fetch_always(std::integral_constant<int, qr>{}, fj2, fs);
                }
                PX_SUB(4);
                PX_STAMP();
                wg_barrier();
            };
            interval(std::integral_constant<int, 0>{});
            interval(std::integral_constant<int, 1>{});
            interval(std::integral_constant<int, 2>{});
            interval(std::integral_constant<int, 3>{});
        }
    }
#undef PX_PICK
#undef P
}

}  // namespace

namespace g4cm {

// 1 when the launch described by p can run on the persistent kernel (the launcher falls back to mlp_bx6_kernel otherwise)
static int g_px6_enabled = -1;          // -1: not read yet (G4C_PX6; default 0 while the kernel is slower than the tile kernel)

int px6_enable(int on) {
    if (g_px6_enabled < 0) g_px6_enabled = getenv("G4C_PX6") ? atoi(getenv("G4C_PX6")) : 0;
    const int old = g_px6_enabled;
    if (on >= 0) g_px6_enabled = on ? 1 : 0;
    return old;
}

bool px6_eligible(const Params &p, bool agg, bool save, bool all_vec) {
    if (!px6_enable(-1) || save || !all_vec) return false;
    if (p.n_src < 1 || p.n_add > 2 || p.out_bf16) return false;
    if (agg && p.n_heads) return false;
    if (agg && p.n_out != NP) return false;
    int nar = 0;
    for (int a = 0; a < p.n_nar; ++a) nar += p.nar[a].width;
    if (nar > NARW_MAX) return false;
    for (int s = 0; s < p.n_src; ++s)
        if (p.src[s].seg_off || !p.src[s].vec || p.src[s].bf16) return false;
    for (int a = 0; a < p.n_add; ++a)
        if (p.add[a].width != NP || (p.add[a].ld & 3) || ((uintptr_t)p.add[a].ptr & 15)) return false;
    if (p.n_heads && (p.head_ld & 3)) return false;
    // the last layer's epilogue runs in the matrix waves: whole 128-wide rows through 16-byte buffer stores, LayerNorm over 128
    if (p.n_out != NP || p.resid || p.out_idx || (p.act != G4C_ACT_NONE && p.act != G4C_ACT_SELU)) return false;
    if (p.out && ((p.out_ld & 3) || ((uintptr_t)p.out & 15) || (long long)p.out_ld * 4 * 32 >= (1LL << 31))) return false;
    // the helpers' gathers use 32-bit byte offsets from the tensors' bases
    long long ldmax = 0;
    for (int s = 0; s < p.n_src; ++s) ldmax = p.src[s].ld > ldmax ? p.src[s].ld : ldmax;
    for (int a = 0; a < p.n_add; ++a) ldmax = p.add[a].ld > ldmax ? p.add[a].ld : ldmax;
    if (p.M * ldmax * 4 >= (1LL << 31)) return false;
    return true;
}

int px6_launch(const Params &p, bool round1, bool agg, hipStream_t st) {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return G4C_ELAUNCH;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    static const int max_wg = getenv("G4C_PX6_WGS") ? atoi(getenv("G4C_PX6_WGS")) : 0;      // tuning only
    // one workgroup per CU, each with a contiguous range of 32-row units (>= 1): a launch with fewer units than 4 per CU
    // spreads them over more workgroups rather than filling four-unit tiles
    int wgs = p.n_tiles;
    const int cap = max_wg > 0 ? max_wg : n_cu;
    if (wgs > cap) wgs = cap;
    if (wgs < 1) wgs = 1;
    const dim3 grid(wgs), blk(512);
    const bool multi = p.n_src > 1;
#define PX_LAUNCH(SP, AGG)                                                                    \
    do {                                                                                      \
        if (p.n_heads) {                                                                      \
            if (multi) mlp_px6_kernel<SP, false, true, true><<<grid, blk, 0, st>>>(p);        \
            else mlp_px6_kernel<SP, false, false, true><<<grid, blk, 0, st>>>(p);             \
        } else if (multi) mlp_px6_kernel<SP, AGG, true, false><<<grid, blk, 0, st>>>(p);      \
        else mlp_px6_kernel<SP, AGG, false, false><<<grid, blk, 0, st>>>(p);                  \
    } while (0)
    if (round1) { if (agg) PX_LAUNCH(1, true); else PX_LAUNCH(1, false); }
    else { if (agg) PX_LAUNCH(3, true); else PX_LAUNCH(3, false); }
#undef PX_LAUNCH
    return g4c::check_launch("g4c_mlp_forward (px6)");
}

}  // namespace g4cm

/* Switches the persistent ping-pong kernel on (1) / off (0) for the launches that can use it; -1 only queries.  Returns the
 * previous setting.  (Tests and A/B timing compare it with the 32-row-tile kernel on identical inputs.) */
extern "C" int g4c_mlp_px6_enable(int on) { return g4cm::px6_enable(on); }
