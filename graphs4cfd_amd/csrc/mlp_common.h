// Shared declarations of the fused-MLP kernels (mlp_fused.hip: tile kernels, mlp_bx6i.hip: dual-tile kernel, mlp_ws.hip: weight-stationary persistent kernel):
// launch parameter block, LDS / stream constants, operand split and activation helpers.
#pragma once
#include "g4c_common.h"
#include <type_traits>

#ifndef G4C_ABLATE
#define G4C_ABLATE 0
#endif

namespace g4cm {

constexpr int KC = 32;        // K chunk = one revolution of the weight ring (8 steps of 4 k)
constexpr int XS = KC + 2;    // LDS row stride of an input chunk (conflict-free ds_read_b64)
constexpr int HS = 128 + 4;   // LDS row stride of the hidden activations (16-byte aligned rows)
constexpr int NP = 128;       // every layer is computed 128 wide
constexpr int CHUNK_FLOATS = KC * NP;   // packed weights per chunk: [16 kpairs][32 lanes][4 tiles][2]

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Src {
    const float *ptr;
    const int *idx;
    int width, wpad, ld, col0, vec, pre_act;
    const int *seg_off;      // bf16x6 kernel: row r = sum / mean of rows [seg_off[r], seg_off[r+1]) (aggregation on load)
    int seg_mean;
    const int *seg_perm;     // optional row indirection of those positions
    int bf16;                // rows are stored as bf16 (ptr is a __bf16 pointer in disguise, ld / col0 in elements): rounded-bf16 mode only
};

struct NarSrc {          // narrow input block multiplied on the VALUs (g4c_src_t.additive == 2): rows = the tile's own rows
    const float *ptr;    // first used column of row 0
    const float *w;      // fp32 [width][128]
    int width, ld;
};

struct AddSrc {          // pre-multiplied first-layer term, gathered per row and added to the layer-0 output
    const float *ptr;
    const int *idx;
    int width, ld;
    int bf16;            // rounded-bf16 mode: the rows are stored as bf16 (ptr is a __bf16 pointer in disguise, ld in elements; 128 wide)
};

struct Params {
    Src src[G4C_MAX_SRC];
    int n_src;
    AddSrc add[G4C_MAX_SRC];
    int n_add;
    int n_layers;
    int chunks0;              // number of 32-k chunks of layer 0 (sum of source widths padded to 32)
    const float *w;           // all layers packed back to back, chunk after chunk (+ one chunk of slack)
    const float *b;           // [n_layers][128] biases, zero padded
    const float *gamma, *beta;
    float eps;
    int n_out;
    long long M;
    float *out;
    int out_ld;
    int out_bf16;            // the output rows are stored as bf16 (out is a __bf16 pointer in disguise, out_ld in elements)
    const int *out_idx;
    int act;
    const float *resid;
    int resid_ld, resid_col0;
    int n_tiles;
    long long row_base;       // first row of this launch (a call may be split into a 64-row and a 32-row launch)
    // "heads": extra bias-free 128x128 products of the FINAL output tile (after LayerNorm / activation), their
    // weights continuing the packed stream after the last layer.  Used to emit the next MP layer's pre-multiplied
    // node-side terms (W1_row v', W1_col v') from the node-MLP launch that produces v' (column-split kernels only).
    int n_heads;
    float *head_out[G4C_MAX_HEADS];
    int head_ld;
    int head_bf16;           // rounded-bf16 mode: the head rows are stored as bf16 (head_out are __bf16 pointers in disguise, head_ld in elements)
    // fused aggregation (bf16x6 kernel): tiles of whole CSR segments (g4c_plan_tiles) instead of fixed 32-row tiles; after
    // the store, the tile's segments are summed / averaged from the LDS copy of the output rows into agg[segment, :]
    const int *tile_rows, *tile_seg, *seg_off;
    float *agg;
    int agg_ld, agg_mean;
    int agg_deg;              // > 0: every segment has exactly this many rows (G4C_AGG_UNIFORM)
    int agg_bf16;             // the aggregate is stored as bf16 rows in the row-split order (G4C_AGG_OUT_BF16; mlp_rs1_kernel only)
    // narrow input blocks of the first layer, multiplied in fp32 on the vector ALUs (bf16x6 kernel)
    NarSrc nar[G4C_MAX_SRC];
    int n_nar;
    // training forward (g4c_mlp_forward_bx6_save): save[l] (or null) receives layer l's output rows, [M, 128] fp32 — the SELU
    // activations of a hidden layer, the pre-LayerNorm rows of the last one — so the backward pass recomputes nothing
    float *save[G4C_MAX_LAYERS];
    int save_ld;
    // backward chain (same entry point): mul[l] (or null) = the SELU OUTPUT rows a hidden layer l's result is multiplied by the
    // slope of, instead of bias + SELU:  y = x * selu'(.)  — the launch then computes  g_{k-1} = (g_k W_k) * selu'(a_{k-1})  layer
    // after layer, writing every g through save[]
    const float *mul[G4C_MAX_LAYERS];
    int mul_ld;
    // f16x3 arithmetic: range_flag[range_slot] = 1 when a value converted to fp16 reached the end of the fp16 range (g4c_mlp_t)
    int *range_flag;
    int range_slot;
};

// The node update fused behind the message launch of an MP layer (g4c_mp_layer_forward_bx6, mlp_ws_kernel<.., NODE>): after its
// last tile pair a workgroup runs the node MLP ([aggregate | v] -> Linear/SELU chain -> LayerNorm -> activation, + heads) on the targets
// whose aggregates it has just written.  Same depth as the message MLP; f16x3 stream; blocks of the stream in the order
// [aggregate, v], layer 2, (layer 3), heads.
struct NodeParams {
    const float *v;          // [n_targets, v_ld] node latents (input block 1; block 0 is the aggregate the message phase wrote to Params::agg)
    int v_ld;
    const float *w;          // packed stream of the node MLP (heads continue it)
    const float *b;          // [n_layers][128]
    const float *gamma, *beta;
    float eps;
    int act;
    float *out;              // v' [n_targets, out_ld]
    int out_ld;
    int n_heads;
    float *head_out[G4C_MAX_HEADS];
    int head_ld;
    int *range_flag;
    int range_slot;
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int HB = 128 + 8;                 // LDS row stride of a bf16 operand plane (272 B: conflict-free ds_read_b128)
constexpr int STEP6 = 3 * 512;                  // bf16 elements of one 16-k step of one column tile (3 planes)
constexpr int BLOCK6 = 4 * 8 * STEP6;           // one 128-k block of the bf16x6 stream

// 16-byte weight-fragment load through a buffer descriptor: 32-bit lane offset (one VGPR for the whole kernel) + wave-uniform
// byte offset in an SGPR + immediate, instead of a 64-bit VALU address computation (and two address VGPRs) per load.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 ldw(__amdgpu_buffer_rsrc_t rs, unsigned voff_bytes, unsigned soff_bytes) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff_bytes, soff_bytes, 0);
    return __builtin_bit_cast(bf16x8, v);
}

// SELU of four values: scale*max(x,0) + scale*alpha*(exp(min(x,0)) - 1); the second term is exactly 0 for x >= 0
// (g4c::selu_f: the clamp modifier of v_exp_f32 stands in for min(x, 0)).
__device__ __forceinline__ f32x4 selu4(f32x4 x) {
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = g4c::selu_f(x[e]);
    return y;
}
__device__ __forceinline__ f32x2 selu2(f32x2 x) {
    f32x2 y;
    y[0] = g4c::selu_f(x[0]); y[1] = g4c::selu_f(x[1]);
    return y;
}

// exact three-way bf16 split of four fp32 values.  Two values at a time: ONE v_cvt_pk_bf16_f32 gives both bf16 terms, and
// their fp32 values come back with a shift / a mask of that packed word (instead of one extra conversion per element);
// the remainders are vector subtractions (v_pk_add_f32).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(f32x2 x, f32x2 &back) {
    bf16x2 b;
    b[0] = (__bf16)x[0]; b[1] = (__bf16)x[1];
    const unsigned u = __builtin_bit_cast(unsigned, b);
    back[0] = __builtin_bit_cast(float, u << 16);
    back[1] = __builtin_bit_cast(float, u & 0xffff0000u);
    return u;
}
// ---- "f16x3": two-way fp16 split, three products (template parameter SP == 2 of the split kernels) -------------------------------
// x = h + l * 2^-11 with h = fp16(x) (round to nearest) and l = fp16((x - h) * 2^11): 22 significand bits per operand (the
// subtraction is exact; l keeps the magnitude range of x, and v_mfma_f32_32x32x16_f16 honours fp16 subnormals —
// scripts/micro/f16_mfma_modes.hip — so small values lose nothing: DESIGN 4.1).  Products: (Wh, xh) into one accumulator, (Wh, xl) + (Wl, xh) into a second one that is folded in with a
// factor 2^-11 when the layer ends; the dropped (Wl, xl) term is <= 2^-22 relative.  Planes 0 / 1 of the stream and of the LDS tile
// hold h / l as fp16 BIT PATTERNS in the bf16-typed containers (plane 2 of the stream is unused).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr float F16_LO_SCALE = 2048.f, F16_LO_UNSCALE = 1.f / 2048.f;
__device__ __forceinline__ f32x16 mfma_f16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// Range: the kernels set MODE.FP16_OVFL, so an activation beyond +-65504 is CLIPPED to 65504 (1 + 2^-11) by the conversions instead of
// becoming an infinity (scripts/micro/f16_mfma_modes.hip shows both behaviours).  Letting it overflow is not "loud": inf - inf = NaN in
// the accumulators, and the SELU's fminf / fmaxf (IEEE minNum / maxNum) turn a NaN pre-activation into 0 — a wrong finite row.  The
// three-way bf16 split keeps the whole fp32 range and stays selectable.
#ifndef G4C_F16_SATURATE
#define G4C_F16_SATURATE 1
#endif
__device__ __forceinline__ void f16_range_mode() { if (G4C_F16_SATURATE) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1); }
__device__ __forceinline__ unsigned pack_f16(f32x2 x, f32x2 &back) {
    f16x2 b;
    b[0] = (_Float16)x[0]; b[1] = (_Float16)x[1];
    back[0] = (float)b[0]; back[1] = (float)b[1];
    return __builtin_bit_cast(unsigned, b);
}
// Two-way fp16 split of a PAIR in four vector instructions: hu = (fp16(y0), fp16(y1)) by one v_cvt_pk_f16_f32, then
// l = fp16((y - h) * 2^11) as fp16(fma(h, -2^11, y * 2^11)) by v_fma_mixlo_f16 / v_fma_mixhi_f16, which read the fp16 halves of hu
// directly (no widening conversion) and write the fp16 halves of lu directly (no narrowing one).  Bit-identical to converting,
// subtracting and scaling: y * 2^11 and h * 2^11 are exact, y - h is exactly representable (it is the part of y's significand below
// h's), so the fused multiply-add returns exactly (y - h) * 2^11 and the only rounding is the final one to fp16 — as before.
// (Written as inline assembly: hipcc widens the halves with v_cvt_f32_f16 instead of folding them into the fma.)
#ifndef G4C_SPLIT_MIX
#define G4C_SPLIT_MIX 1      // (0: the conversion form, kept for scripts/split_mix_check.py's bitwise comparison of the two builds)
#endif
// (Packed-f32 vector instructions beside MFMAs cost ~4 ns each in isolation — scripts/micro/mfma_fillers.hip — but replacing the three
// this split and the SELU use by plain instructions made a phase of mlp_ws_kernel no faster: 1868 against 1736 cycles)
// Range tracking of the f16x3 arithmetic: every value that is converted to fp16 (MLP inputs when they are parked, hidden
// activations in the epilogues) is also compared with the end of the fp16 range; a wave that saw a clipped value writes the flag word
// once, at the end of the kernel (a plain store of 1: nothing is written on the fast path).  Two trackers:
//   RangeV  one v_max3_f32 per PAIR into a per-lane running maximum of |value| (one live VGPR; mlp_ws_kernel)
//   RangeS  per QUAD a maximum into a temporary, one v_cmp and an s_or into a wave-uniform lane mask (three vector instructions per
//           four values, no live VGPR: mlp_bx6_kernel sits at its register limit and spills 10 - 20 registers with RangeV)
constexpr float F16_RANGE_END = 65504.f;
struct RangeV { float m = 0.f; };
struct RangeS { unsigned long long any = 0ull; };
__device__ __forceinline__ void range_track(RangeV &r, f32x2 y) { r.m = fmaxf(fmaxf(r.m, fabsf(y[0])), fabsf(y[1])); }       // v_max3_f32 |.|
__device__ __forceinline__ void range_track(RangeS &r, f32x2 y) {
    r.any |= __builtin_amdgcn_ballot_w64(fmaxf(fabsf(y[0]), fabsf(y[1])) >= F16_RANGE_END);
}
__device__ __forceinline__ void range_track(RangeV &r, f32x4 y) {
    f32x2 a, b; a[0] = y[0]; a[1] = y[1]; b[0] = y[2]; b[1] = y[3];
    range_track(r, a); range_track(r, b);
}
__device__ __forceinline__ void range_track(RangeS &r, f32x4 y) {
    r.any |= __builtin_amdgcn_ballot_w64(fmaxf(fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fabsf(y[2])), fabsf(y[3])) >= F16_RANGE_END);
}
__device__ __forceinline__ bool range_hit(const RangeV &r) { return __builtin_amdgcn_ballot_w64(r.m >= F16_RANGE_END) != 0ull; }
__device__ __forceinline__ bool range_hit(const RangeS &r) { return r.any != 0ull; }
template <class R>
__device__ __forceinline__ void range_report(const Params &p, const R &r) {
    if (p.range_flag && range_hit(r)) {
        if ((threadIdx.x & 63) == 0) p.range_flag[p.range_slot] = 1;
    }
}
__device__ __forceinline__ void split_pair_f16(f32x2 y, unsigned &hu, unsigned &lu) {
    f16x2 b;
    b[0] = (_Float16)y[0]; b[1] = (_Float16)y[1];
    hu = __builtin_bit_cast(unsigned, b);
    if (G4C_SPLIT_MIX) {
        const f32x2 ys = y * F16_LO_SCALE;
        const float c = -F16_LO_SCALE;
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lu) : "v"(hu), "s"(c), "v"(ys[0]));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lu) : "v"(hu), "s"(c), "v"(ys[1]));
    } else {
        f32x2 hf, lf;
        hf[0] = (float)b[0]; hf[1] = (float)b[1];
        lu = pack_f16((y - hf) * F16_LO_SCALE, lf);
    }
}
template <class R>
__device__ __forceinline__ void split_pair_f16(f32x2 y, unsigned &hu, unsigned &lu, R &rng) {
    range_track(rng, y);
    split_pair_f16(y, hu, lu);
}
struct RangeNone {};
__device__ __forceinline__ void range_track(RangeNone &, f32x4) {}
template <class R>
__device__ __forceinline__ void split2x4(f32x4 x, bf16x4 &h, bf16x4 &l, R &rng) {
    range_track(rng, x);
    unsigned hu[2], lu[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        f32x2 v;
        v[0] = x[2 * j]; v[1] = x[2 * j + 1];
        split_pair_f16(v, hu[j], lu[j]);
    }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 hh, ll;
    hh[0] = hu[0]; hh[1] = hu[1]; ll[0] = lu[0]; ll[1] = lu[1];
    h = __builtin_bit_cast(bf16x4, hh); l = __builtin_bit_cast(bf16x4, ll);
}
__device__ __forceinline__ void split2(float x, __bf16 &h, __bf16 &l) {
    const _Float16 a = (_Float16)x;
    const _Float16 b = (_Float16)((x - (float)a) * F16_LO_SCALE);
    h = __builtin_bit_cast(__bf16, a); l = __builtin_bit_cast(__bf16, b);
}

template <int SP, class R> __device__ __forceinline__ void split3x4(f32x4 x, bf16x4 &h, bf16x4 &m, bf16x4 &l, R &rng);
template <int SP = 3>
__device__ __forceinline__ void split3x4(f32x4 x, bf16x4 &h, bf16x4 &m, bf16x4 &l) { RangeNone none; split3x4<SP>(x, h, m, l, none); }
template <int SP, class R>
__device__ __forceinline__ void split3x4(f32x4 x, bf16x4 &h, bf16x4 &m, bf16x4 &l, R &rng) {
    if (SP == 2) { split2x4(x, h, m, rng); l = m; return; }       // two-way fp16 split: h, l in the first two containers
    if (SP == 1 || (G4C_ABLATE & 512)) {          // SP == 1: round to bf16 (only h is stored)
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = (__bf16)x[e]; m[e] = h[e]; l[e] = h[e]; }
        return;
    }
    unsigned hu[2], mu[2], lu[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        f32x2 v, hf, mf, lf;
        v[0] = x[2 * j]; v[1] = x[2 * j + 1];
        hu[j] = pack_bf16(v, hf);
        const f32x2 r1 = v - hf;
        mu[j] = pack_bf16(r1, mf);
        const f32x2 r2 = r1 - mf;
        lu[j] = pack_bf16(r2, lf);
    }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 hh, mm, ll;
    hh[0] = hu[0]; hh[1] = hu[1]; mm[0] = mu[0]; mm[1] = mu[1]; ll[0] = lu[0]; ll[1] = lu[1];
    h = __builtin_bit_cast(bf16x4, hh); m = __builtin_bit_cast(bf16x4, mm); l = __builtin_bit_cast(bf16x4, ll);
}

// exact three-way bf16 split of an fp32 value
__device__ __forceinline__ void split3(float x, __bf16 &h, __bf16 &m, __bf16 &l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// dual-tile software-pipelined kernel (mlp_bx6i.hip)
int bx6i_enable(int on);
bool bx6i_eligible(const Params &p, bool round1, bool agg, bool save, bool f16x2, long long row_count);
int bx6i_launch(const Params &p, bool agg, bool f16x2, hipStream_t st);

// weight-stationary persistent kernel (mlp_ws.hip): f16x3 stream only
int ws_enable(int on);
bool ws_eligible(const Params &p, bool round1, bool agg, bool save, bool f16x2, long long row_count, bool any_size = false);
int ws_launch(const Params &p, bool agg, bool round1, hipStream_t st, const NodeParams *node = nullptr);


// row-split persistent kernel (mlp_rs.hip, round 6): f16x3 stream, hoisted three-layer message form
bool rs_eligible(const Params &p, bool agg, long long row_count);
int rs_launch(const Params &p, bool agg, hipStream_t st);
bool rs2_eligible(const Params &p, long long row_count);
int rs2_launch(const Params &p, bool e_natural, hipStream_t st);

// four bf16 values (two dwords as loaded) widened to fp32: a shift / a mask each — exact
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 widen_bf16x4(u32x2 w) {
    f32x4 x;
    x[0] = __builtin_bit_cast(float, w[0] << 16); x[1] = __builtin_bit_cast(float, w[0] & 0xffff0000u);
    x[2] = __builtin_bit_cast(float, w[1] << 16); x[3] = __builtin_bit_cast(float, w[1] & 0xffff0000u);
    return x;
}


}  // namespace g4cm
