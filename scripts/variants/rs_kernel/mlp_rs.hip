// Row-split persistent form of the fused split-operand MLP ("rs", round 6) for the hoisted message launch of an MP layer on a mesh of
// uniform in-degree — gather -> SELU on load -> three Linear / SELU layers as two-way fp16 split products on v_mfma_f32_16x16x32_f16 ->
// LayerNorm -> store -> per-target aggregation.  Replaces MLP.forward (graphs4cfd/nn/blocks.py:117-144) with the torch.cat / index ops in
// front of it (nn/blocks.py:181) and the scatter(e', col, reduce) behind it (nn/blocks.py:183), like mlp_ws_kernel (mlp_ws.hip).
//
// What is different from mlp_ws_kernel: there a wave owns 16 output FEATURES of every layer for the rows of a tile, so a
// layer's output has to cross the workgroup (LDS planes, a barrier per layer and tile) before it is the next layer's operand, and
// LayerNorm / aggregation need the rows back in LDS — three pipes that in-order waves and seven barriers per 64 rows let overlap by half.
// Here a wave owns 16 ROWS and computes all 128 features of every layer for them:
//   * C = W x X^T on 16x16x32 MFMAs leaves lane (n, g) with features 16 b + 4 g + e (e = 0..3) of row n for each of the eight feature
//     blocks b — which is exactly a B operand of the NEXT layer if that layer's k order inside a 32-k step is {4 g + e, 16 + 4 g + e}
//     instead of {8 g + j}: the binding permutes the columns of the weight matrices accordingly before it packs them, and a layer's output
//     never leaves the registers (SELU, fp16 split and range tracking in place);
//   * all 128 features of a row live in one wave: LayerNorm is 31 in-lane adds and two cross-lane steps, no LDS, no barrier;
//   * the weights cannot be stationary (a wave needs all of a layer: 64 KB): they stream L2 -> LDS by LDS-DMA into a ring of two layers,
//     one barrier per LAYER and round (128 rows per workgroup) instead of one per layer and 32-row tile, and it orders nothing but the ring;
//   * a wave owns a contiguous, segment-aligned range of rows: the aggregation is a segmented scan over the 16 rows of a chunk (DPP row
//     shifts with per-row masks) whose running sum is carried to the wave's next chunk in registers.
// Envelope: f16x3 stream, ONE weighted 128-wide direct block (optional SELU on load), two additive 128-wide blocks through indices, three
// 128-wide layers, LayerNorm, no output activation / residual / heads / output index, fp32 rows; with the fused aggregation only for
// G4C_AGG_UNIFORM(k), 4 <= k <= 8.
#include "mlp_common.h"
using namespace g4cm;

// timing-only ablations (wrong results): 1 the weight ring is never advanced (no DMA, no barrier), 2 no row stores, 4 no input / additive gathers
#ifndef G4C_RS_AHEAD
#define G4C_RS_AHEAD 2          // steps the weight fragments are read ahead of their MFMAs
#endif
#ifndef G4C_RS_SPREAD
#define G4C_RS_SPREAD 0
#endif
#ifndef G4C_RS_ABLATE
#define G4C_RS_ABLATE 0
#endif

namespace {

constexpr int RS_WAVES = 8;                   // waves per workgroup (two per SIMD)
constexpr int LAYER_BYTES = 64 * 1024;        // one layer's two operand planes in LDS: [col tile 4][16-k step 8][plane 2][1 KB]

// 16 bytes per lane global -> LDS (lds_base wave-uniform, lane l lands at lds_base + 16 l).  Inline assembly: the compiler orders every
// LDS-DMA it knows of before the next s_barrier; these it does not see — rs_dma_wait() and the barrier behind it order them.
__device__ __forceinline__ void rs_dma16(const char *g, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void rs_vm_wait0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

// sum over the four lanes n, n + 16, n + 32, n + 48 (the four g of a row), result in all of them: v_permlane16_swap exchanges the odd
// 16-lane rows of its first operand with the even rows of its second, v_permlane32_swap the upper half of the first with the lower half
// of the second — on two copies of v each gives (v[l], v[l ^ 16]) resp. (v[l], v[l ^ 32]) side by side.  (Inline assembly: the builtin
// folded the two results into one register; s_nop: the instruction reads registers a vector instruction has just written.)
__device__ __forceinline__ float sum_over_g(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a += b; b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

// STORE: the rows themselves are written (p.out).  Straight-line memory code throughout the chunk loop — no branch around a load or a
// store — so that hipcc's s_waitcnt vmcnt(N) can count: behind a conditional store it falls back to vmcnt(0), which makes every
// consumer of a prefetched row wait for the previous chunk's stores to be acknowledged (memory returns in order per wave).
template <bool AGG, bool STORE>
__global__ __launch_bounds__(RS_WAVES * 64, 2) void mlp_rs_kernel(const Params p) {
    __shared__ __attribute__((aligned(1024))) char sW[2 * LAYER_BYTES];
    __shared__ __attribute__((aligned(16))) float sBias[3 * NP];
    __shared__ __attribute__((aligned(16))) float sGB[2 * NP];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    f16_range_mode();
    RangeV rng;

    // ---- this wave's rows: a contiguous range cut at segment boundaries (K rows per segment; K = 1 without the aggregation), evenly in
    // segments over all waves of the launch (XCD-aware: consecutive workgroup slots of an XCD hold consecutive ranges)
    const int K = AGG ? p.agg_deg : 1;
    int R0, R1;
    {
        const int G = gridDim.x, b = blockIdx.x;
        const int slot = (G & 7) ? b : (b & 7) * (G >> 3) + (b >> 3);
        const long long n_seg = p.M / K, gw = (long long)slot * RS_WAVES + wave, nw = (long long)G * RS_WAVES;
        R0 = __builtin_amdgcn_readfirstlane((int)((gw * n_seg) / nw) * K);
        R1 = __builtin_amdgcn_readfirstlane((int)(((gw + 1) * n_seg) / nw) * K);
    }
    // rounds of the WORKGROUP (every wave takes part in every round's barriers): the longest range of its waves
    int rounds;
    {
        const int G = gridDim.x, b = blockIdx.x;
        const int slot = (G & 7) ? b : (b & 7) * (G >> 3) + (b >> 3);
        const long long n_seg = p.M / K, nw = (long long)G * RS_WAVES;
        int mx = 0;
        for (int w = 0; w < RS_WAVES; ++w) {
            const long long gw = (long long)slot * RS_WAVES + w;
            const int a = (int)((gw * n_seg) / nw) * K, e = (int)(((gw + 1) * n_seg) / nw) * K;
            const int c = (e - a + 15) >> 4;
            mx = c > mx ? c : mx;
        }
        rounds = __builtin_amdgcn_readfirstlane(mx);
    }
    if (rounds == 0) return;

    if (tid < 3 * NP) sBias[tid] = p.b[tid];
    if (tid < 2 * NP) sGB[tid] = p.gamma ? (tid < NP ? p.gamma[tid] : p.beta[tid - NP]) : (tid < NP ? 1.f : 0.f);

    // ---- weight ring: layer c of the launch's layer sequence (0, 1, 2, 0, 1, 2, ...) sits in slot c % 2.  A layer = 64 pieces of 1 KB
    // (col tile ct, 16-k step st, plane pl) of the packed stream (planes 0 / 1 of its three): piece q = ct * 16 + st * 2 + pl, eight
    // pieces per wave.
    const unsigned lds_w = (unsigned)reinterpret_cast<uintptr_t>(sW);
    auto dma_layer = [&](int layer, int slot) __attribute__((always_inline)) {
        const char *src = reinterpret_cast<const char *>(p.w) + (size_t)layer * (2u * BLOCK6);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = wave * 8 + j;
            const int ct = q >> 4, st = (q >> 1) & 7, pl = q & 1;
            rs_dma16(src + (size_t)((ct * 8 + st) * 3072 + pl * 1024) + lane * 16, lds_w + (unsigned)(slot * LAYER_BYTES + q * 1024));
        }
    };
    auto dma_piece = [&](int layer, int slot, int j) __attribute__((always_inline)) {
        const char *src = reinterpret_cast<const char *>(p.w) + (size_t)layer * (2u * BLOCK6);
        const int q = wave * 8 + j;
        const int ct = q >> 4, st = (q >> 1) & 7, pl = q & 1;
        rs_dma16(src + (size_t)((ct * 8 + st) * 3072 + pl * 1024) + lane * 16, lds_w + (unsigned)(slot * LAYER_BYTES + q * 1024));
    };
    dma_layer(0, 0);
    dma_layer(1, 1);
    rs_vm_wait0();
    __syncthreads();

    // this lane's part of every fragment address: (g >> 1) selects the 16-k half of a 32-k step, (g & 1) the k / 8 parity, n the feature
    const unsigned frag_lane = (unsigned)((g >> 1) * 2048 + ((g & 1) * 32 + n) * 16);
    // fragment (feature block b, 32-k step ks, plane pl) of the layer in `slot`
    auto wfrag = [&](int slot, int b, int ks, int pl) __attribute__((always_inline)) {
        const unsigned off = (unsigned)(slot * LAYER_BYTES + (((b >> 1) * 8 + 2 * ks) * 2 + pl) * 1024 + (b & 1) * 256) + frag_lane;
        return *reinterpret_cast<const bf16x8 *>(sW + off);
    };

    const bool pact = p.src[0].pre_act != 0;
    const int *const ix0 = p.add[0].idx, *const ix1 = p.add[1].idx;          // (the launcher requires both)
    int c_layer = 0;          // the layer-sequence counter (slot = c_layer & 1, the layer after next is requested into the slot just freed)

    // one epilogue value quadruple -> the next layer's operand halves (element positions 4 (b & 1) .. + 3 of 32-k step b >> 1)
    auto to_operand = [&](f32x4 y, bf16x8 &oh, bf16x8 &ol, int half) __attribute__((always_inline)) {
        bf16x4 h, l;
        split2x4(y, h, l, rng);
#pragma unroll
        for (int e = 0; e < 4; ++e) { oh[4 * half + e] = h[e]; ol[4 * half + e] = l[e]; }
    };

    if (R1 <= R0) {          // (more waves than segments: this wave only keeps the ring and the barriers going)
        for (int c = 0; c < 3 * rounds; ++c) {
#if G4C_RS_SPREAD
            if (c > 0) dma_layer((c + 1) % 3, (c + 1) & 1);
            rs_vm_wait0();
            __syncthreads();
#else
            rs_vm_wait0();
            __syncthreads();
            dma_layer((c + 2) % 3, c & 1);
#endif
        }
        rs_vm_wait0();
        return;
    }

    // ---- software pipeline over this wave's chunks of 16 rows.  Rows past the wave's range are CLAMPED to its last row: a lane that
    // has no row of its own recomputes that row bit for bit and stores the same bytes to the same place (no predicate, no branch).  The
    // input row pieces, the gather indices and the first three feature blocks of additive rows of chunk i + 1 are requested while chunk
    // i runs (second layer / before its stores); the other additive blocks three blocks ahead of the epilogue that adds them (they are
    // added BEHIND the products: the MFMAs do not wait for them).
    auto row_of = [&](int rd) __attribute__((always_inline)) {
        const int r = R0 + 16 * rd + n;
        return r < R1 ? r : R1 - 1;
    };
    f32x4 xa[4], xb[4];
    int ir, ic;
    auto request_x = [&](int rd) __attribute__((always_inline)) {
        const int row = row_of(rd);
        const float *xr = p.src[0].ptr + (long long)row * p.src[0].ld + p.src[0].col0 + 4 * g;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            xa[ks] = *reinterpret_cast<const f32x4 *>(xr + 32 * ks);
            xb[ks] = *reinterpret_cast<const f32x4 *>(xr + 32 * ks + 16);
        }
        ir = ix0[row];
        ic = ix1[row];
    };
    bf16x8 inh[4], inl[4];
    auto convert_x = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f32x4 va = xa[ks], vb = xb[ks];
            if (pact) { va = selu4(va); vb = selu4(vb); }
            to_operand(va, inh[ks], inl[ks], 0);
            to_operand(vb, inh[ks], inl[ks], 1);
        }
    };
    const float *pra, *pca;
    f32x4 ar[3], ac[3];          // ring of three feature blocks of additive rows
    auto request_adds = [&]() __attribute__((always_inline)) {
        pra = p.add[0].ptr + (long long)ir * p.add[0].ld + 4 * g;
        pca = p.add[1].ptr + (long long)ic * p.add[1].ld + 4 * g;
#pragma unroll
        for (int b = 0; b < 3; ++b) { ar[b] = *reinterpret_cast<const f32x4 *>(pra + 16 * b); ac[b] = *reinterpret_cast<const f32x4 *>(pca + 16 * b); }
    };
    request_x(0);
    convert_x();
    request_adds();

    f32x4 y[8];          // the last layer's output: features 16 b + 4 g + e of row n (G4C_RS_SPREAD: stored under the next chunk's first layer)
    float *yp = p.out;
    bool have_prev = false;          // (G4C_RS_SPREAD: a barrier has freed a ring slot — from the first layer's end on)
    for (int rd = 0; rd < rounds; ++rd) {
        bf16x8 outh[4], outl[4];
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const int slot = c_layer & 1;
            if (l == 1) request_x(rd + 1 < rounds ? rd + 1 : rd);          // (xa / xb / ir / ic are free since the conversion)
            // One layer = 32 steps (feature block b = s / 4, 32-k step ks = s % 4) of three MFMAs; the weight fragments of a step are
            // read from the LDS ring two steps ahead; the epilogue of block b - 1 (fold, additive rows, SELU, fp16 split into the next
            // layer's operand registers) is issued in four pieces beside the MFMAs of block b, which accumulate into the other
            // accumulator pair.
            constexpr int AH = G4C_RS_AHEAD, RING = AH + 1;
            bf16x8 fh[RING], fl[RING];
#pragma unroll
            for (int q = 0; q < AH; ++q) { fh[q] = wfrag(slot, q >> 2, q & 3, 0); fl[q] = wfrag(slot, q >> 2, q & 3, 1); }
            f32x4 acc[2], acc1[2], v;
            // piece k (0..3) of the epilogue of block `eb`, whose sums are in acc[eb & 1] / acc1[eb & 1]
            auto epilogue = [&](int eb, int k) __attribute__((always_inline)) {
                if (k == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(acc1[eb & 1][e], F16_LO_UNSCALE, acc[eb & 1][e]);
                    if (l == 0) {          // + sender-side + receiver-side product rows (behind the products; the ring slot is refilled)
                        const f32x4 a0 = ar[eb % 3], a1 = ac[eb % 3];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (v[e] + a0[e]) + a1[e];
                        if (eb + 3 < 8) { ar[eb % 3] = *reinterpret_cast<const f32x4 *>(pra + 16 * (eb + 3)); ac[eb % 3] = *reinterpret_cast<const f32x4 *>(pca + 16 * (eb + 3)); }
                    }
                    if (l == 2) y[eb] = v;
                } else if (l < 2 && k == 1) {
                    if (!(G4C_RS_ABLATE & 32)) v = selu4(v);
                } else if (l < 2 && k == 2) {
                    to_operand(v, outh[eb >> 1], outl[eb >> 1], eb & 1);
                }
            };
#pragma unroll
            for (int st = 0; st < 32; ++st) {
                const int b = st >> 2, ks = st & 3;
                if (st + AH < 32 && !((G4C_RS_ABLATE & 16) && st >= 1)) { fh[(st + AH) % RING] = wfrag(slot, (st + AH) >> 2, (st + AH) & 3, 0); fl[(st + AH) % RING] = wfrag(slot, (st + AH) >> 2, (st + AH) & 3, 1); }
#if G4C_RS_SPREAD
                // the ring: the slot of the layer BEFORE this one was freed by the barrier in front of this layer; the layer after this
                // one is requested into it piece by piece under this layer's first MFMAs (eight 1 KB pieces per wave)
                if (!(G4C_RS_ABLATE & 1) && (st & 1) == 0 && st < 16 && have_prev) dma_piece((c_layer + 1) % 3, (c_layer + 1) & 1, st >> 1);
                // the previous chunk's rows: one 16-byte store per lane every second step of the first layer
                if (STORE && !(G4C_RS_ABLATE & 2) && l == 0 && (st & 1) == 1 && st < 16 && rd > 0) *reinterpret_cast<f32x4 *>(yp + 16 * (st >> 1)) = y[st >> 1];
#endif
                if (ks == 0) {
                    acc[b & 1] = *reinterpret_cast<const f32x4 *>(sBias + l * NP + 16 * b + 4 * g);
                    acc1[b & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (b > 0) epilogue(b - 1, ks);
                if (G4C_RS_ABLATE & 8) {
                    asm volatile("" :: "v"(fh[st % RING]), "v"(fl[st % RING]), "v"(inl[ks]), "v"(inh[ks]));
                } else {
                acc1[b & 1] = mfma16(fh[st % RING], inl[ks], acc1[b & 1]);
                acc[b & 1] = mfma16(fh[st % RING], inh[ks], acc[b & 1]);
                acc1[b & 1] = mfma16(fl[st % RING], inh[ks], acc1[b & 1]);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);          // DS read (the fragments two steps ahead)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);          // VALU
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) epilogue(7, k);
            // ---- the ring: everybody is done with this layer's slot and has its pieces of the next layer in LDS -> request the layer
            // after next into the slot just freed
#if !(G4C_RS_ABLATE & 1)
            rs_vm_wait0();
            __syncthreads();
#if !G4C_RS_SPREAD
            dma_layer((c_layer + 2) % 3, slot);
#endif
#endif
            ++c_layer;
            have_prev = true;
            if (l < 2) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) { inh[ks] = outh[ks]; inl[ks] = outl[ks]; }
            }
        }

        // ---- LayerNorm over the row's 128 features (32 in this lane, the rest in the three other lanes of the row), activation
        {
            float s = 0.f;
#pragma unroll
            for (int b = 0; b < 8; ++b) s += (y[b][0] + y[b][1]) + (y[b][2] + y[b][3]);
            const float mean = sum_over_g(s) * (1.0f / NP);
            float q0 = 0.f, q1 = 0.f;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const float d0 = y[b][0] - mean, d1 = y[b][1] - mean, d2 = y[b][2] - mean, d3 = y[b][3] - mean;
                q0 = fmaf(d0, d0, q0); q1 = fmaf(d1, d1, q1); q0 = fmaf(d2, d2, q0); q1 = fmaf(d3, d3, q1);
            }
            const float rstd = rsqrtf(sum_over_g(q0 + q1) * (1.0f / NP) + p.eps);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const f32x4 g4 = *reinterpret_cast<const f32x4 *>(sGB + 16 * b + 4 * g), b4 = *reinterpret_cast<const f32x4 *>(sGB + NP + 16 * b + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[b][e] = fmaf((y[b][e] - mean) * rstd, g4[e], b4[e]);
            }
        }
        // ---- the next chunk's operands and first additive rows (requested long ago / now: in front of this chunk's stores)
        const int row_st = row_of(rd);
        convert_x();
        request_adds();
        if constexpr (STORE && !(G4C_RS_ABLATE & 2)) {
            float *op = p.out + (long long)row_st * p.out_ld + 4 * g;
#if G4C_RS_SPREAD
            yp = op;
            if (rd == rounds - 1) {
#pragma unroll
                for (int b = 0; b < 8; ++b) *reinterpret_cast<f32x4 *>(op + 16 * b) = y[b];
            }
#else
#pragma unroll
            for (int b = 0; b < 8; ++b) *reinterpret_cast<f32x4 *>(op + 16 * b) = y[b];
#endif
        }
    }
    // (the two layers requested last are never read: let them land before the workgroup's LDS is handed on)
    rs_vm_wait0();
    rng.m = rng.m;
    range_report(p, rng);
}

}  // namespace

namespace g4cm {

static int g_rs = -1;
int rs_enable(int on) {
    if (g_rs < 0) g_rs = 0;
    const int old = g_rs;
    if (on >= 0) g_rs = on > 2 ? 2 : on;
    return old;
}

bool rs_eligible(const Params &p, bool agg, long long row_count) {
    if (agg) return false;          // (the aggregation: next step)
    if (p.n_src != 1 || p.n_nar != 0 || p.n_add != 2 || p.n_heads || p.n_layers != 3 || p.n_out != NP || p.resid || p.out_idx || p.out_bf16) return false;
    const Src &s = p.src[0];
    if (s.width != NP || !s.vec || s.seg_off || s.bf16 || s.idx || (s.ld & 3) || (s.col0 & 3) || ((uintptr_t)s.ptr & 15)) return false;
    for (int a = 0; a < 2; ++a)
        if (p.add[a].width != NP || (p.add[a].ld & 3) || ((uintptr_t)p.add[a].ptr & 15) || p.add[a].bf16) return false;
    if (!p.out && !agg) return false;
    if (p.out && ((p.out_ld & 3) || ((uintptr_t)p.out & 15))) return false;
    if (!p.add[0].idx || !p.add[1].idx || p.act != G4C_ACT_NONE) return false;
    if (!p.gamma || ((uintptr_t)p.gamma & 15) || ((uintptr_t)p.beta & 15) || ((uintptr_t)p.b & 15)) return false;
    if (p.M >= (1LL << 31) || p.row_base != 0 || row_count != p.M) return false;
    return true;
}

int rs_launch(const Params &p, bool agg, hipStream_t st) {
    if (p.M == 0) return G4C_OK;
    const int n_cu = g4c::cu_count();
    const long long chunks = (p.M + 15) / 16;
    const long long want = (chunks + RS_WAVES - 1) / RS_WAVES;
    const dim3 grid((unsigned)(want < n_cu ? want : n_cu)), blk(RS_WAVES * 64);
    if (agg) { if (p.out) mlp_rs_kernel<true, true><<<grid, blk, 0, st>>>(p); else mlp_rs_kernel<true, false><<<grid, blk, 0, st>>>(p); }
    else mlp_rs_kernel<false, true><<<grid, blk, 0, st>>>(p);
    return g4c::check_launch("g4c_mlp_forward (rs)");
}

}  // namespace g4cm

extern "C" int g4c_mlp_rs_enable(int on) { return g4cm::rs_enable(on); }
