// Row-split persistent form of the fused MLP in the rounded-bf16 arithmetic ("rs", round 6) for the hoisted message launch of an MP layer
// over receivers of one uniform in-degree (BASELINE config 3: REMuS-GNN's angle launches — every edge of a k-nearest-neighbour graph
// receives k angles): gather -> two or three Linear / SELU layers on v_mfma_f32_16x16x32_bf16 -> LayerNorm -> store -> per-receiver sum /
// mean.  Replaces MLP.forward (graphs4cfd/nn/blocks.py:117-144) with the torch.cat / index ops in front of it (nn/blocks.py:327) and the
// scatter(a', col, reduce) behind it (nn/blocks.py:330), like mlp_ws_kernel<SP = 1> (mlp_ws.hip), which keeps every other launch of the mode.
//
// What is different from mlp_ws_kernel: there a wave owns 16 output FEATURES of every layer for the rows of a tile, so a
// layer's output crosses the workgroup (LDS planes, a barrier per layer and tile) before it is the next layer's operand.  Here a wave
// owns 16 ROWS and computes all 128 features of every layer for them:
//   * C = W x X^T on 16x16x32 MFMAs leaves lane (n, g) with features 16 b + 4 g + e (e = 0..3) of row n for each of the eight feature
//     blocks b — which is exactly a B operand of the NEXT layer if that layer's k order inside a 32-k step is {4 g + e, 16 + 4 g + e}
//     instead of {8 g + j}: the binding permutes the columns of the weight matrices accordingly before it packs them
//     (g4c_mlp_t.w_format = G4C_WFMT_BF16_RS), and a layer's output never leaves the registers;
//   * all 128 features of a row live in one wave: LayerNorm is 31 in-lane adds and two cross-lane steps, no LDS, no barrier;
//   * one operand plane per layer (rounded bf16: one product per multiply-add): ALL layers' weights stay in LDS (32 KB each, brought
//     once by LDS-DMA) — no ring, no barrier behind the prologue, every wave streams its own rows independently;
//   * a wave owns a contiguous, segment-aligned range of rows: the aggregation is a segmented scan over the 16 rows of a chunk (DPP row
//     shifts with per-row masks) whose running sum is carried to the wave's next chunk through 512 bytes of wave-private LDS.
// Straight-line memory code throughout the chunk loop — no branch around a load or a store — so that hipcc's s_waitcnt vmcnt(N) can
// count: behind a conditional store it falls back to vmcnt(0), which makes every consumer of a prefetched row wait for the previous
// chunk's stores to be acknowledged (memory returns in order per wave).
//
// bf16 rows — the weighted block (XB16), the two additive product tables (AB16), the stored rows (OUT 1 / 2) — are in the STREAM'S K
// ORDER (include/g4c.h "row-split order"): position 32 j + 8 g + 4 h + e of a 128-wide row holds feature 32 j + 16 h + 4 g + e, so that
// the eight values a lane needs of a 32-feature step (its operand of one MFMA, its two output blocks of a pair) are 16 contiguous bytes
// and a row group's four lanes read or write 64 contiguous bytes per instruction.  Measured on the 2.5 M-row launch of config 3: 622 us
// with 8-byte pieces in natural order, 511 us with 16-byte pieces (mlp_ws_kernel<SP = 1>: 656 us) — the launch is bound by the number of
// memory instructions and 32-byte pieces the texture path handles, not by HBM bytes (profiles/r06_rs1_*.log).  fp32 rows and the
// aggregate keep the natural order (16 bytes per lane as they are).
//   XB16  the weighted block's rows are bf16 (already activated: the bf16(SELU(a')) rows an earlier launch of this kind stored)
//   AB16  the two additive product tables are bf16
//   OUT   0 fp32 rows, 1 bf16 rows, 2 bf16(SELU(row)) rows (g4c_mlp_forward_bf16_agg out_dtype), 3 rows not stored (aggregate only)
//   AGG   G4C_AGG_UNIFORM(K), 4 <= K <= 8: per chunk of 16 rows the LayerNorm'd fp32 rows are summed per segment by a segmented
//         inclusive scan over the 16 lanes of a row group (v_fmac_f32 with DPP row_shr 1 / 2 / 4 and per-row masks; a segment cut by the
//         chunk's end continues with the previous chunk's running sum), and the lanes that hold a segment's last row store its sum /
//         mean.  Fixed order, deterministic; NOT the sequential order of g4c_segment_reduce (last-bit differences: tests allow 1e-6).
//         AGG = 2 (G4C_AGG_OUT_BF16): the aggregate is stored as bf16 rows in the same column order — what its one reader, the update
//         MLP of the layer, rounds it to on load anyway (same operand, half the bytes of two launches).
// Envelope (rs_eligible): rounded-bf16 stream in this kernel's k order, ONE weighted 128-wide direct block (fp32 with optional SELU on
// load, or bf16), two additive 128-wide blocks through indices, two or three 128-wide layers, LayerNorm, no output activation / residual /
// heads / output index.  A stream in this k order runs on no other kernel: outside the envelope the call fails.
#include "mlp_common.h"
using namespace g4cm;

// timing-only ablations (wrong results; scripts/variants/rs_kernel/README.md): 1 no SELU, 2 no scan, 4 no LayerNorm, 8 no MFMA,
// 16 no row stores, 32 no aggregate stores, 64 no additive rows
#ifndef RS1_ABLATE
#define RS1_ABLATE 0
#endif

namespace {

constexpr int RS_WAVES = 8;                   // waves per workgroup (two per SIMD)

// 16 bytes per lane global -> LDS (lds_base wave-uniform, lane l lands at lds_base + 16 l).  Inline assembly: the compiler orders every
// LDS-DMA it knows of before the next s_barrier with vmcnt(0); these it does not see — rs_vm_wait0() and the barrier behind it order them.
__device__ __forceinline__ void rs_dma16(const char *g, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void rs_vm_wait0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

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

__device__ __forceinline__ f32x4 rs1_selu4(f32x4 v) { if (RS1_ABLATE & 1) return v; return selu4(v); }
__device__ __forceinline__ f32x4 mfma16b(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

template <int NL, bool XB16, bool AB16, int OUT, int AGG>
__global__ __launch_bounds__(RS_WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp_rs1_kernel(const Params p) {
    __shared__ __attribute__((aligned(1024))) char sW1[NL * 32 * 1024];
    __shared__ __attribute__((aligned(16))) float sBias[3 * NP];
    __shared__ __attribute__((aligned(16))) float sGB[2 * NP];
    __shared__ __attribute__((aligned(16))) float sCarry1[AGG ? RS_WAVES * NP : 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int K = AGG ? p.agg_deg : 1;
    int R0, R1;
    {
        const int G = gridDim.x, b = blockIdx.x;
        const int slot = (G & 7) ? b : (b & 7) * (G >> 3) + (b >> 3);
        const long long n_seg = p.M / K, gw = (long long)slot * RS_WAVES + wave, nw = (long long)G * RS_WAVES;
        R0 = __builtin_amdgcn_readfirstlane((int)((gw * n_seg) / nw) * K);
        R1 = __builtin_amdgcn_readfirstlane((int)(((gw + 1) * n_seg) / nw) * K);
    }
    if (tid < NL * NP) sBias[tid] = p.b[tid];
    if (tid < 2 * NP) sGB[tid] = tid < NP ? p.gamma[tid] : p.beta[tid - NP];
    // weights: plane 0 of every (col tile, 16-k step) piece of every layer -> [layer][ct][st][1 KB], four pieces per wave and layer
    {
        const unsigned lds_w = (unsigned)reinterpret_cast<uintptr_t>(sW1);
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const char *src = reinterpret_cast<const char *>(p.w) + (size_t)l * (2u * BLOCK6);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = wave * 4 + j;          // = ct * 8 + st
                rs_dma16(src + (size_t)q * 3072 + lane * 16, lds_w + (unsigned)(l * 32768 + q * 1024));
            }
        }
        rs_vm_wait0();
    }
    __syncthreads();
    if (R1 <= R0) return;
    const unsigned frag_lane = (unsigned)((g >> 1) * 1024 + ((g & 1) * 32 + n) * 16);
    auto wfrag = [&](int l, int b, int ks) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8 *>(sW1 + (unsigned)(l * 32768 + ((b >> 1) * 8 + 2 * ks) * 1024 + (b & 1) * 256) + frag_lane);
    };
    const bool pact = !XB16 && p.src[0].pre_act != 0;
    const int *const ix0 = p.add[0].idx, *const ix1 = p.add[1].idx;
    auto row_of = [&](int rd) __attribute__((always_inline)) { const int r = R0 + 16 * rd + n; return r < R1 ? r : R1 - 1; };
    auto to_op = [&](f32x4 y, bf16x8 &o, int half) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[4 * half + e] = (__bf16)y[e];
    };
    // ---- the next chunk's input pieces: fp32 (two 16-byte loads per 32-k step) or bf16 (two 8-byte loads: the operand halves as they are)
    f32x4 xa[XB16 ? 1 : 4], xb[XB16 ? 1 : 4];
    u32x4v xw[XB16 ? 4 : 1];
    int ir, ic;
    auto request_x = [&](int rd) __attribute__((always_inline)) {
        const int row = row_of(rd);
        if constexpr (XB16) {
            const __bf16 *xr = reinterpret_cast<const __bf16 *>(p.src[0].ptr) + (long long)row * p.src[0].ld + p.src[0].col0 + 8 * g;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) xw[ks] = *reinterpret_cast<const u32x4v *>(xr + 32 * ks);
        } else {
            const float *xr = p.src[0].ptr + (long long)row * p.src[0].ld + p.src[0].col0 + 4 * g;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { xa[ks] = *reinterpret_cast<const f32x4 *>(xr + 32 * ks); xb[ks] = *reinterpret_cast<const f32x4 *>(xr + 32 * ks + 16); }
        }
    };
    auto request_idx = [&](int rd) __attribute__((always_inline)) { const int row = row_of(rd); ir = ix0[row]; ic = ix1[row]; };
    bf16x8 in[4];
    auto convert_x = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr (XB16) {
                in[ks] = __builtin_bit_cast(bf16x8, xw[ks]);
            } else {
                f32x4 va = xa[ks], vb = xb[ks];
                if (pact) { va = rs1_selu4(va); vb = rs1_selu4(vb); }
                to_op(va, in[ks], 0); to_op(vb, in[ks], 1);
            }
        }
    };
    // ---- additive rows: all eight feature blocks of both tables, requested in front of the previous chunk's stores (a request
    // issued behind them could only be waited for by draining them: the counter retires in order)
    const char *pra, *pca;
    f32x4 ar[AB16 ? 1 : 8], ac[AB16 ? 1 : 8];
    u32x4v arw[AB16 ? 4 : 1], acw[AB16 ? 4 : 1];          // bf16 tables: two feature blocks per 16-byte piece
    auto request_adds = [&]() __attribute__((always_inline)) {
        if constexpr (AB16) {
            pra = reinterpret_cast<const char *>(p.add[0].ptr) + ((long long)ir * p.add[0].ld + 8 * g) * 2;
            pca = reinterpret_cast<const char *>(p.add[1].ptr) + ((long long)ic * p.add[1].ld + 8 * g) * 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) { arw[j] = *reinterpret_cast<const u32x4v *>(pra + 64 * j); acw[j] = *reinterpret_cast<const u32x4v *>(pca + 64 * j); }
        } else {
            pra = reinterpret_cast<const char *>(p.add[0].ptr) + ((long long)ir * p.add[0].ld + 4 * g) * 4;
            pca = reinterpret_cast<const char *>(p.add[1].ptr) + ((long long)ic * p.add[1].ld + 4 * g) * 4;
#pragma unroll
            for (int b = 0; b < 8; ++b) { ar[b] = *reinterpret_cast<const f32x4 *>(pra + 64 * b); ac[b] = *reinterpret_cast<const f32x4 *>(pca + 64 * b); }
        }
    };
    auto add_rows = [&](int b, f32x4 &a0, f32x4 &a1) __attribute__((always_inline)) {          // feature block b of both gathered rows
        if constexpr (AB16) {
            u32x2 h0, h1;
            h0[0] = arw[b >> 1][2 * (b & 1)]; h0[1] = arw[b >> 1][2 * (b & 1) + 1];
            h1[0] = acw[b >> 1][2 * (b & 1)]; h1[1] = acw[b >> 1][2 * (b & 1) + 1];
            a0 = widen_bf16x4(h0); a1 = widen_bf16x4(h1);
        } else { a0 = ar[b]; a1 = ac[b]; }
    };
    const int rounds = (R1 - R0 + 15) >> 4;
    // AGG: the previous chunk's last scanned row (lane 15 of each row group: the running sum of the segment the chunk's end cut),
    // kept in LDS — wave-private, 128 floats
    f32x4 *const carry = reinterpret_cast<f32x4 *>(sCarry1) + (wave * 4 + g) * 8;
    // the wave's segments of the aggregate as a raw buffer (offsets past its end are dropped)
    constexpr int AGB = AGG == 2 ? 2 : 4;          // bytes per value of the aggregate
    const __amdgpu_buffer_rsrc_t agg_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        AGG ? (void *)(reinterpret_cast<char *>(p.agg) + (long long)(R0 / K) * p.agg_ld * AGB) : nullptr, 0,
        AGG ? (int)(((R1 - R0) / K) * p.agg_ld * AGB) : 0, 0x00020000);
    if constexpr (AGG) {
#pragma unroll
        for (int b = 0; b < 8; ++b) carry[b] = f32x4{0.f, 0.f, 0.f, 0.f};          // (multiplied by 0 in chunk 0: must be finite)
    }
    const float inv_k = 1.0f / (float)K;
    // Requests run most of a chunk ahead of their use, and always in front of a chunk's stores (a request issued behind stores could
    // only be waited for by draining them: the counter retires in order): after chunk c's first layer, chunk c + 1's input rows and
    // — into the registers that layer has just finished with — its additive rows, then chunk c + 2's two indices.
    request_idx(0);
    request_x(0);
    request_adds();
    request_idx(rounds > 1 ? 1 : 0);
    // Sixteen stores that the bounds check drops, so that the first chunk enters the loop the way every other one does — with its
    // additive rows requested IN FRONT OF a chunk's worth of stores: hipcc's wait for those rows at the top of the loop is then
    // vmcnt(16) on both paths instead of the vmcnt(0) that would drain every chunk's stores before the next chunk starts.
#pragma unroll
    for (int b = 0; b < 16; ++b)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4v{0u, 0u, 0u, 0u}, agg_rsrc, 0x7ffff000u + 16u * b, 0, 0);          // (distinct offsets: identical stores would be merged)
    convert_x();
    for (int rd = 0; rd < rounds; ++rd) {
        asm volatile("" ::: "memory");          // (the weights in LDS are loop-invariant: without this hipcc hoists their reads and spills)
        bf16x8 out[4];
        f32x4 y[8];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            if (l == 1) {
                request_x(rd + 1 < rounds ? rd + 1 : rd);
                request_adds();
                request_idx(rd + 2 < rounds ? rd + 2 : rounds - 1);
            }
            bf16x8 f[4];
#pragma unroll
            for (int q = 0; q < 3; ++q) f[q] = wfrag(l, q >> 2, q & 3);
            f32x4 acc[2], v;
            auto epilogue = [&](int eb, int k) __attribute__((always_inline)) {
                if (k == 0) {
                    v = acc[eb & 1];
                    if (l == 0 && !(RS1_ABLATE & 64)) {
                        f32x4 a0, a1;
                        add_rows(eb, a0, a1);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (v[e] + a0[e]) + a1[e];
                    }
                    if (l == NL - 1) y[eb] = v;
                } else if (l < NL - 1 && k == 1) {
                    v = rs1_selu4(v);
                } else if (l < NL - 1 && k == 2) {
                    to_op(v, out[eb >> 1], eb & 1);
                }
            };
#pragma unroll
            for (int st = 0; st < 32; ++st) {
                const int b = st >> 2, ks = st & 3;
                if (st + 3 < 32) f[(st + 3) % 4] = wfrag(l, (st + 3) >> 2, (st + 3) & 3);
                if (ks == 0) acc[b & 1] = *reinterpret_cast<const f32x4 *>(sBias + l * NP + 16 * b + 4 * g);
                if (b > 0) epilogue(b - 1, ks);
                if (!(RS1_ABLATE & 8)) acc[b & 1] = mfma16b(f[st % 4], in[ks], acc[b & 1]);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) epilogue(7, k);
            if (l < NL - 1) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) in[ks] = out[ks];
            }
        }
        // ---- LayerNorm
        if (!(RS1_ABLATE & 4)) {
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
        // ---- the next chunk's operands and first additive rows (in front of this chunk's stores)
        const int row_st = row_of(rd);
        convert_x();
        // ---- the rows
        if constexpr (OUT != 3 && !(RS1_ABLATE & 16)) {
            if constexpr (OUT == 0) {
                float *op = p.out + (long long)row_st * p.out_ld + 4 * g;
#pragma unroll
                for (int b = 0; b < 8; ++b) *reinterpret_cast<f32x4 *>(op + 16 * b) = y[b];
            } else {
                // bf16 rows in the stream's k order: the lane's feature blocks 2 j and 2 j + 1 side by side, 16 bytes per lane and
                // 64 contiguous bytes per row and store
                __bf16 *op = reinterpret_cast<__bf16 *>(p.out) + (long long)row_st * p.out_ld + 8 * g;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 w0 = OUT == 2 ? rs1_selu4(y[2 * j]) : y[2 * j], w1 = OUT == 2 ? rs1_selu4(y[2 * j + 1]) : y[2 * j + 1];
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = (__bf16)w0[e]; o[4 + e] = (__bf16)w1[e]; }
                    *reinterpret_cast<bf16x8 *>(op + 32 * j) = o;
                }
            }
        }
        // ---- aggregation: segmented scan over the chunk's 16 rows (the un-activated fp32 rows), carried across chunks
        if constexpr (AGG) {
            const int rloc = 16 * rd + n;                                   // row within the wave's range (segment-aligned at 0)
            const int qd = (int)(((float)rloc + 0.5f) * inv_k);             // exact: rloc < 2^22
            const int dist = rloc - qd * K;                                 // position of the row in its segment
            const bool real = R0 + rloc < R1;
            const float m1 = dist >= 1 ? 1.f : 0.f, m2 = dist >= 2 ? 1.f : 0.f, m4 = dist >= 4 ? 1.f : 0.f;
            const float mc = dist > n ? 1.f : 0.f;                          // the segment started in the previous chunk
            // v += m * v[lane - d] (0 in front of the row group's first lane) for d = 1, 2, 4: one DPP multiply-add per step.  Four
            // values per block so that three instructions separate a value's write from its next DPP read (the hazard needs two).
            auto scan4 = [&](f32x4 &v) __attribute__((always_inline)) {
                asm volatile("s_nop 1\n\t"
                             "v_fmac_f32_dpp %0, %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_fmac_f32_dpp %1, %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_fmac_f32_dpp %2, %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_fmac_f32_dpp %3, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_fmac_f32_dpp %0, %0, %5 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_fmac_f32_dpp %1, %1, %5 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_fmac_f32_dpp %2, %2, %5 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_fmac_f32_dpp %3, %3, %5 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_fmac_f32_dpp %0, %0, %6 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_fmac_f32_dpp %1, %1, %6 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_fmac_f32_dpp %2, %2, %6 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_fmac_f32_dpp %3, %3, %6 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(m1), "v"(m2), "v"(m4));
            };
            // the lanes that hold a segment's last row store its sum / mean: a buffer store whose offset is out of range in every other
            // lane (dropped by the bounds check) — no branch around the stores, so the waits on later loads need not drain them
            const bool last = real && dist == K - 1;
            const unsigned aoff = last ? (unsigned)((qd * p.agg_ld + (AGG == 2 ? 8 : 4) * g) * AGB) : 0x7ffffff0u;
            const float km = p.agg_mean ? (float)K : 1.f, ikm = p.agg_mean ? inv_k : 1.f;
            f32x4 a[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const f32x4 pv = carry[b];
                if (!(RS1_ABLATE & 2)) scan4(y[b]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[b][e] = fmaf(pv[e], mc, y[b][e]);                     // the segment's rows in the previous chunk (mc = 0 in chunk 0)
                    const float q0 = y[b][e] * ikm;                         // sum / K, correctly rounded (q0 = the sum itself when km = 1)
                    a[b][e] = fmaf(fmaf(-km, q0, y[b][e]), ikm, q0);
                }
                if constexpr (AGG == 1) {
                    if (!(RS1_ABLATE & 32)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, a[b]), agg_rsrc, aoff + 64u * b, 0, 0);
                } else if (b & 1) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = (__bf16)a[b - 1][e]; o[4 + e] = (__bf16)a[b][e]; }
                    if (!(RS1_ABLATE & 32)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, o), agg_rsrc, aoff + 32u * (b - 1), 0, 0);
                }
            }
            if (n == 15) {
#pragma unroll
                for (int b = 0; b < 8; ++b) carry[b] = y[b];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The update MLP of such a layer in the same form ("rs2"): [aggregate | e] -> Linear (256 -> 128) -> SELU -> Linear -> LayerNorm -> SELU
// -> e' (+ the next layer's two hoisted products W1s e', W1r e' as heads) for REMuS-GNN's edge update (graphs4cfd/nn/blocks.py:331-333
// with the next EdgeMP's nn/blocks.py:327).  Both input blocks are bf16 rows in the row-split order (the aggregate of mlp_rs1_kernel,
// G4C_AGG_OUT_BF16; the compact edge latents of the previous launch of this kernel), every stored bf16 row is in that order too.
// The five 128 x 128 weight blocks (two of the first layer, the second layer, two heads) are 160 KB of bf16 — ALL of a CU's LDS — so
// the bias / LayerNorm vectors live in registers (128 per lane: the four values of each of the eight feature blocks a lane owns).
//   OUT16  e' is stored as bf16 rows in the row-split order (its one reader: the next launch of this kernel) / as fp32 rows in feature order
//   HEADS  the two heads are computed and stored (bf16, row-split order)
//   ENAT   the e block's bf16 rows are in FEATURE order (G4C_WFMT_BF16_RS2N: rows a tile-kernel launch stored — the first update of a
//          run of EdgeMPs): two 8-byte loads per 32-feature step instead of one 16-byte load
template <bool OUT16, bool HEADS, bool ENAT>
__global__ __launch_bounds__(RS_WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp_rs2_kernel(const Params p) {
    constexpr int NB = HEADS ? 5 : 3;          // weight blocks: L0 (aggregate), L0 (e), L1, head 0, head 1
    __shared__ __attribute__((aligned(1024))) char sW2[NB * 32 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    int R0, R1;
    {
        const int G = gridDim.x, b = blockIdx.x;
        const int slot = (G & 7) ? b : (b & 7) * (G >> 3) + (b >> 3);
        const long long chunks = (p.M + 15) >> 4, gw = (long long)slot * RS_WAVES + wave, nw = (long long)G * RS_WAVES;
        R0 = __builtin_amdgcn_readfirstlane((int)((gw * chunks) / nw) * 16);
        const long long r1 = ((gw + 1) * chunks) / nw * 16;
        R1 = __builtin_amdgcn_readfirstlane((int)(r1 < p.M ? r1 : p.M));
    }
    {
        const unsigned lds_w = (unsigned)reinterpret_cast<uintptr_t>(sW2);
#pragma unroll
        for (int l = 0; l < NB; ++l) {
            const char *src = reinterpret_cast<const char *>(p.w) + (size_t)l * (2u * BLOCK6);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = wave * 4 + j;          // = ct * 8 + st
                rs_dma16(src + (size_t)q * 3072 + lane * 16, lds_w + (unsigned)(l * 32768 + q * 1024));
            }
        }
        rs_vm_wait0();
    }
    // the lane's bias / LayerNorm values (features 16 b + 4 g + e)
    f32x4 cb0[8], cb1[8], cg[8], cbe[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        cb0[b] = *reinterpret_cast<const f32x4 *>(p.b + 16 * b + 4 * g);
        cb1[b] = *reinterpret_cast<const f32x4 *>(p.b + NP + 16 * b + 4 * g);
        cg[b] = *reinterpret_cast<const f32x4 *>(p.gamma + 16 * b + 4 * g);
        cbe[b] = *reinterpret_cast<const f32x4 *>(p.beta + 16 * b + 4 * g);
    }
    __syncthreads();
    if (R1 <= R0) return;
    const unsigned frag_lane = (unsigned)((g >> 1) * 1024 + ((g & 1) * 32 + n) * 16);
    auto wfrag = [&](int l, int b, int ks) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8 *>(sW2 + (unsigned)(l * 32768 + ((b >> 1) * 8 + 2 * ks) * 1024 + (b & 1) * 256) + frag_lane);
    };
    auto row_of = [&](int rd) __attribute__((always_inline)) { const int r = R0 + 16 * rd + n; return r < R1 ? r : R1 - 1; };
    auto to_op = [&](f32x4 y, bf16x8 &o, int half) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[4 * half + e] = (__bf16)y[e];
    };
    u32x4v xin[2][4];          // the chunk's two input blocks, 16 bytes per 32-feature step: the MFMA operands as they are
    auto request_x = [&](int rd) __attribute__((always_inline)) {
        const int row = row_of(rd);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (ENAT && s2 == 1) {
                const __bf16 *xr = reinterpret_cast<const __bf16 *>(p.src[1].ptr) + (long long)row * p.src[1].ld + p.src[1].col0 + 4 * g;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const u32x2 lo = *reinterpret_cast<const u32x2 *>(xr + 32 * ks), hi = *reinterpret_cast<const u32x2 *>(xr + 32 * ks + 16);
                    xin[1][ks][0] = lo[0]; xin[1][ks][1] = lo[1]; xin[1][ks][2] = hi[0]; xin[1][ks][3] = hi[1];
                }
            } else {
                const __bf16 *xr = reinterpret_cast<const __bf16 *>(p.src[s2].ptr) + (long long)row * p.src[s2].ld + p.src[s2].col0 + 8 * g;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) xin[s2][ks] = *reinterpret_cast<const u32x4v *>(xr + 32 * ks);
            }
        }
    };
    const int rounds = (R1 - R0 + 15) >> 4;
    const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(nullptr, 0, 0, 0x00020000);
    request_x(0);
    // (stores the bounds check drops: the loop is entered the way its back edge enters it — see mlp_rs1_kernel)
#pragma unroll
    for (int b = 0; b < (HEADS ? 8 : 0) + (OUT16 ? 4 : 8); ++b)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4v{0u, 0u, 0u, 0u}, none, 0x7ffff000u + 16u * b, 0, 0);
    for (int rd = 0; rd < rounds; ++rd) {
        asm volatile("" ::: "memory");          // (the weights in LDS are loop-invariant: without this hipcc hoists their reads and spills)
        bf16x8 in[2][4], out[4];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) in[s2][ks] = __builtin_bit_cast(bf16x8, xin[s2][ks]);
        // ---- layer 0: eight feature blocks x (two input blocks x four 32-k steps)
        {
            bf16x8 f[4];
#pragma unroll
            for (int q = 0; q < 3; ++q) f[q] = wfrag(q >> 2 & 1, 0, q & 3);
            f32x4 acc[2], v;
            auto epilogue = [&](int eb, int k) __attribute__((always_inline)) {
                if (k == 0) v = acc[eb & 1];
                else if (k == 1) v = rs1_selu4(v);
                else if (k == 2) to_op(v, out[eb >> 1], eb & 1);
            };
#pragma unroll
            for (int st = 0; st < 64; ++st) {          // st = 8 b + 4 blk + ks
                const int b = st >> 3, blk = (st >> 2) & 1, ks = st & 3;
                if (st + 3 < 64) { const int t = st + 3; f[t % 4] = wfrag((t >> 2) & 1, t >> 3, t & 3); }
                if ((st & 7) == 0) acc[b & 1] = cb0[b];
                if (b > 0 && (st & 7) < 3) epilogue(b - 1, st & 7);
                acc[b & 1] = mfma16b(f[st % 4], in[blk][ks], acc[b & 1]);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) epilogue(7, k);
        }
        // ---- the next chunk's rows (into the registers layer 0 has just released), in front of this chunk's stores
        request_x(rd + 1 < rounds ? rd + 1 : rd);
        // ---- layer 1
        f32x4 y[8];
        {
            bf16x8 f[4];
#pragma unroll
            for (int q = 0; q < 3; ++q) f[q] = wfrag(2, 0, q);
            f32x4 acc[2];
#pragma unroll
            for (int st = 0; st < 32; ++st) {
                const int b = st >> 2, ks = st & 3;
                if (st + 3 < 32) f[(st + 3) % 4] = wfrag(2, (st + 3) >> 2, (st + 3) & 3);
                if (ks == 0) acc[b & 1] = cb1[b];
                if (b > 0 && ks == 0) y[b - 1] = acc[(b - 1) & 1];
                acc[b & 1] = mfma16b(f[st % 4], out[ks], acc[b & 1]);
            }
            y[7] = acc[1];
        }
        // ---- LayerNorm, activation
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
#pragma unroll
                for (int e = 0; e < 4; ++e) y[b][e] = fmaf((y[b][e] - mean) * rstd, cg[b][e], cbe[b][e]);
                if (p.act == G4C_ACT_SELU) y[b] = rs1_selu4(y[b]);
            }
        }
        const int row_st = row_of(rd);
        // ---- e'
        if constexpr (OUT16) {
            __bf16 *op = reinterpret_cast<__bf16 *>(p.out) + (long long)row_st * p.out_ld + 8 * g;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (__bf16)y[2 * j][e]; o[4 + e] = (__bf16)y[2 * j + 1][e]; }
                *reinterpret_cast<bf16x8 *>(op + 32 * j) = o;
            }
        } else {
            float *op = p.out + (long long)row_st * p.out_ld + 4 * g;
#pragma unroll
            for (int b = 0; b < 8; ++b) *reinterpret_cast<f32x4 *>(op + 16 * b) = y[b];
        }
        // ---- the heads: 128 x 128 products of the activated row, stored as bf16 rows in the row-split order
        if constexpr (HEADS) {
#pragma unroll
            for (int b = 0; b < 8; ++b) to_op(y[b], out[b >> 1], b & 1);
#pragma unroll
            for (int hd = 0; hd < 2; ++hd) {
                __bf16 *hp = reinterpret_cast<__bf16 *>(p.head_out[hd]) + (long long)row_st * p.head_ld + 8 * g;
                bf16x8 f[4];
#pragma unroll
                for (int q = 0; q < 3; ++q) f[q] = wfrag(3 + hd, 0, q);
                f32x4 acc[2];
#pragma unroll
                for (int st = 0; st < 32; ++st) {
                    const int b = st >> 2, ks = st & 3;
                    if (st + 3 < 32) f[(st + 3) % 4] = wfrag(3 + hd, (st + 3) >> 2, (st + 3) & 3);
                    if (ks == 0) acc[b & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc[b & 1] = mfma16b(f[st % 4], out[ks], acc[b & 1]);
                    if (ks == 3 && (b & 1)) {
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { o[e] = (__bf16)acc[0][e]; o[4 + e] = (__bf16)acc[1][e]; }
                        *reinterpret_cast<bf16x8 *>(hp + 16 * (b - 1)) = o;
                    }
                }
            }
        }
    }
}

}  // namespace

namespace g4cm {

bool rs_eligible(const Params &p, bool agg, long long row_count) {
    if (agg && (p.agg_deg < 4 || p.agg_deg > 8)) return false;
    if (p.n_src != 1 || p.n_nar != 0 || p.n_add != 2 || p.n_heads || (p.n_layers != 2 && p.n_layers != 3) || p.n_out != NP || p.resid || p.out_idx) return false;
    const Src &s = p.src[0];
    if (s.width != NP || !s.vec || s.seg_off || s.idx || (s.ld & 7) || (s.col0 & 7) || ((uintptr_t)s.ptr & 15) || (s.bf16 && (s.pre_act || !agg))) return false;
    for (int a = 0; a < 2; ++a)
        if (p.add[a].width != NP || (p.add[a].ld & 7) || ((uintptr_t)p.add[a].ptr & 15) || !p.add[a].idx || p.add[a].bf16 != p.add[0].bf16) return false;
    if (!p.out && !agg) return false;
    if (agg && (!p.agg || (p.agg_ld & (p.agg_bf16 ? 7 : 3)) || ((uintptr_t)p.agg & 15) || p.M % p.agg_deg != 0)) return false;
    if (p.out && ((p.out_ld & 7) || ((uintptr_t)p.out & 15))) return false;
    if (!p.gamma || p.act != G4C_ACT_NONE || ((uintptr_t)p.gamma & 15) || ((uintptr_t)p.beta & 15) || ((uintptr_t)p.b & 15)) return false;
    return p.M < (1LL << 31) && p.row_base == 0 && row_count == p.M;
}

int rs_launch(const Params &p, bool agg, hipStream_t st) {
    if (p.M == 0) return G4C_OK;
    const int n_cu = g4c::cu_count();
    const long long chunks = (p.M + 15) / 16, want = (chunks + RS_WAVES - 1) / RS_WAVES;
    const bool xb = p.src[0].bf16 != 0, ab = p.add[0].bf16 != 0;
    const dim3 grid((unsigned)(want < n_cu ? want : n_cu)), blk(RS_WAVES * 64);
    const int od = p.out ? p.out_bf16 : 3;
#define G4C_RS1(NL, XB, AB, OD) do { if (agg && p.agg_bf16) mlp_rs1_kernel<NL, XB, AB, OD, 2><<<grid, blk, 0, st>>>(p); \
                                     else if (agg) mlp_rs1_kernel<NL, XB, AB, OD, 1><<<grid, blk, 0, st>>>(p); \
                                     else if constexpr (!(XB) && (OD) != 3) mlp_rs1_kernel<NL, XB, AB, OD, 0><<<grid, blk, 0, st>>>(p); \
                                     else { g4c::set_error("rs: bf16 input rows without an aggregation are not built"); return G4C_EUNSUPPORTED; } } while (0)
#define G4C_RS1_XA(NL, XB, AB) do { if (od == 0) G4C_RS1(NL, XB, AB, 0); else if (od == 2) G4C_RS1(NL, XB, AB, 2); else if (od == 3 && agg) G4C_RS1(NL, XB, AB, 3); \
                                  else { g4c::set_error("rs: this combination of row formats is not built"); return G4C_EUNSUPPORTED; } } while (0)
#define G4C_RS1_NL(NL) do { if (!xb && !ab && od == 0) G4C_RS1(NL, false, false, 0); else if (!xb && ab) G4C_RS1_XA(NL, false, true); \
                            else if (xb && ab) G4C_RS1_XA(NL, true, true); else { g4c::set_error("rs: this combination of row formats is not built"); return G4C_EUNSUPPORTED; } } while (0)
    if (p.n_layers == 2) G4C_RS1_NL(2); else G4C_RS1_NL(3);
#undef G4C_RS1_NL
#undef G4C_RS1_XA
#undef G4C_RS1
    return g4c::check_launch("g4c_mlp_forward_bf16 (rs)");
}

bool rs2_eligible(const Params &p, long long row_count) {
    if (p.n_src != 2 || p.n_nar != 0 || p.n_add != 0 || p.n_layers != 2 || p.n_out != NP || p.resid || p.out_idx || p.agg || !p.out) return false;
    if (p.n_heads != 0 && (p.n_heads != 2 || !p.head_bf16 || (p.head_ld & 7) || ((uintptr_t)p.head_out[0] & 15) || ((uintptr_t)p.head_out[1] & 15))) return false;
    for (int s2 = 0; s2 < 2; ++s2) {
        const Src &s = p.src[s2];
        if (s.width != NP || !s.vec || s.seg_off || s.idx || !s.bf16 || s.pre_act || (s.ld & 7) || (s.col0 & 7) || ((uintptr_t)s.ptr & 15)) return false;
    }
    if ((p.out_ld & (p.out_bf16 ? 7 : 3)) || ((uintptr_t)p.out & 15) || p.out_bf16 > 1) return false;
    if (!p.gamma || (p.act != G4C_ACT_NONE && p.act != G4C_ACT_SELU) || ((uintptr_t)p.gamma & 15) || ((uintptr_t)p.beta & 15) || ((uintptr_t)p.b & 15)) return false;
    return p.M < (1LL << 31) && p.row_base == 0 && row_count == p.M;
}

int rs2_launch(const Params &p, bool e_natural, hipStream_t st) {
    if (p.M == 0) return G4C_OK;
    const int n_cu = g4c::cu_count();
    const long long chunks = (p.M + 15) / 16, want = (chunks + RS_WAVES - 1) / RS_WAVES;
    const dim3 grid((unsigned)(want < n_cu ? want : n_cu)), blk(RS_WAVES * 64);
#define G4C_RS2(O16, HD) do { if (e_natural) mlp_rs2_kernel<O16, HD, true><<<grid, blk, 0, st>>>(p); else mlp_rs2_kernel<O16, HD, false><<<grid, blk, 0, st>>>(p); } while (0)
    if (p.n_heads) { if (p.out_bf16) G4C_RS2(true, true); else G4C_RS2(false, true); }
    else { if (p.out_bf16) G4C_RS2(true, false); else G4C_RS2(false, false); }
#undef G4C_RS2
    return g4c::check_launch("g4c_mlp_forward_bf16 (rs2)");
}

}  // namespace g4cm
