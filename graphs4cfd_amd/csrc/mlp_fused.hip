// Fused gather -> concat -> MLP (fp32 MFMA) -> LayerNorm -> activation -> residual -> store.
//
// Replaces, in ONE launch, what the reference issues as ~10-25 torch kernels per MLP call:
// `torch.cat((e, v[row], v[col]))` + nn.Linear/nn.SELU chain + nn.LayerNorm (MLP.forward,
// graphs4cfd/nn/blocks.py:117-144) + the F.selu / torch.tanh applied by the caller
// (nn/mus_gnn.py:178-212, nn/blocks.py:233,288) + the residual time step (nn/mus_gnn.py:218).
// The concatenated input is never materialised: each source is gathered row-wise straight
// into LDS.
//
// Tiling (CDNA4, wave64): a workgroup of 4 waves owns TM = 64 rows.  Every layer is a
// [64 x K] x [K x N<=128] product on v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD):
// wave w computes row tile (w&1) and column tiles (w>>1), (w>>1)+2 -> 2 accumulators of 16
// VGPRs.  K is walked in chunks of 32: the weight chunk ([k/2][n][2] packed, 16 KiB) and, for
// the first layer, the gathered input chunk are prefetched into registers while the previous
// chunk is multiplied, then written to single LDS buffers (59 KiB/WG -> 2 WGs per CU, so one
// workgroup's staging overlaps the other's MFMAs).  Hidden activations stay in LDS
// ([64][130] fp32) between layers; the last layer's rows are normalised / activated from LDS by
// 32-lane groups and stored as whole rows.
//
// MFMA-bound in fp32: 2*K*N FLOP per row per layer against ~(K_in + N_out)*4 bytes per row.
#include "g4c_common.h"

namespace {

constexpr int TM = 64;        // rows per workgroup
constexpr int KC = 32;        // K chunk
constexpr int XS = KC + 2;    // LDS row stride of the input chunk (conflict-free ds_read_b64)
constexpr int HS = 128 + 2;   // LDS row stride of the hidden activations
constexpr int NTHREADS = 256;
constexpr int MAXN = 128;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Src {
    const float *ptr;
    const int *idx;
    int width, wpad, ld, col0, vec, pre_act;
};

struct Params {
    Src src[G4C_MAX_SRC];
    int n_src;
    int n_layers;
    int kpad[G4C_MAX_LAYERS], npad[G4C_MAX_LAYERS];
    const float *w[G4C_MAX_LAYERS];
    const float *b[G4C_MAX_LAYERS];
    const float *gamma, *beta;
    float eps;
    int n_out;
    long long M;
    float *out;
    int out_ld;
    const int *out_idx;
    int act;
    const float *resid;
    int resid_ld, resid_col0;
    int n_tiles;
};

constexpr int LDS_FLOATS = TM * XS + KC * MAXN + TM * HS + G4C_MAX_SRC * TM;

// One K-chunk of MFMAs for this wave: A rows from `sA` (row stride `as`, columns [a0, a0+cw)),
// B from the packed weight chunk in sW ([cw/2][np][2]).  Per 4 k: one ds_read_b64 of A
// (lanes<32: k,k+1; lanes>=32: k+2,k+3), one ds_read_b64 of B per column tile, two MFMAs per tile.
// Tile presence is a template parameter so the k loop is branch-free and fully unrolled.
template <bool T0, bool T1>
__device__ __forceinline__ void mma_step(const float *pa, const float *pb, int np, int kk, f32x16 &acc0, f32x16 &acc1) {
    const float2 a = *reinterpret_cast<const float2 *>(pa + kk);
    const float *pbk = pb + (kk >> 1) * np * 2;
    if (T0) {
        const float2 b = *reinterpret_cast<const float2 *>(pbk);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc0, 0, 0, 0);
    }
    if (T1) {
        const float2 b = *reinterpret_cast<const float2 *>(pbk + 64 * 2);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc1, 0, 0, 0);
    }
}

template <bool T0, bool T1>
__device__ __forceinline__ void mma_chunk_t(const float *pa, const float *pb, int np, int cw, f32x16 &acc0, f32x16 &acc1) {
    if (cw == KC) {
#pragma unroll
        for (int kk = 0; kk < KC; kk += 4) mma_step<T0, T1>(pa, pb, np, kk, acc0, acc1);
    } else {
        for (int kk = 0; kk < cw; kk += 4) mma_step<T0, T1>(pa, pb, np, kk, acc0, acc1);
    }
}

__device__ __forceinline__ void mma_chunk(const float *sA, int as, int a0, const float *sW, int np, int cw,
                                          int rt, int ct0, int nt, int i, int h, f32x16 &acc0, f32x16 &acc1) {
    const float *pa = sA + (rt * 32 + i) * as + a0 + 2 * h;
    const float *pb = sW + (h * np + ct0 * 32 + i) * 2;
    if (ct0 + 2 < nt) mma_chunk_t<true, true>(pa, pb, np, cw, acc0, acc1);
    else if (ct0 < nt) mma_chunk_t<true, false>(pa, pb, np, cw, acc0, acc1);
}

// bias (+ SELU unless last layer) of one 32x32 accumulator tile -> hidden buffer.
// C/D layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
__device__ __forceinline__ void store_hidden(const f32x16 &a, float *sH, const float *bias, int ct, int rt,
                                             int i, int h, bool last) {
    const float bv = bias[ct * 32 + i];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int row = rt * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        float x = a[q] + bv;
        if (!last) x = g4c::selu_f(x);
        sH[row * HS + ct * 32 + i] = x;
    }
}

// Prefetch registers are named members (not arrays): an array here gets "promoted" to LDS by the
// AMDGPU backend, costing 16 KiB of LDS per workgroup.
struct WRegs { float4 a, b, c, d; };
struct XRegs { float4 a, b; };

__device__ __forceinline__ float4 ldw1(const float *w, int n4, int f) {
    return reinterpret_cast<const float4 *>(w)[f < n4 ? f : 0];   // unconditional load, clamped
}
__device__ __forceinline__ void load_w(const float *w, int n4, int tid, WRegs &r) {
    r.a = ldw1(w, n4, tid);
    r.b = ldw1(w, n4, tid + NTHREADS);
    r.c = ldw1(w, n4, tid + 2 * NTHREADS);
    r.d = ldw1(w, n4, tid + 3 * NTHREADS);
}

__device__ __forceinline__ void store_w(float *sW, int n4, int tid, const WRegs &r) {
    float4 *d = reinterpret_cast<float4 *>(sW);
    if (tid < n4) d[tid] = r.a;
    if (tid + NTHREADS < n4) d[tid + NTHREADS] = r.b;
    if (tid + 2 * NTHREADS < n4) d[tid + 2 * NTHREADS] = r.c;
    if (tid + 3 * NTHREADS < n4) d[tid + 3 * NTHREADS] = r.d;
}

// gathered input chunk: thread (r = tid>>3 [+32], c4 = tid&7) loads 4 consecutive columns
__device__ __forceinline__ float4 ldx1(const Src &s, int srow, int c) {
    const float *base = s.ptr + (long long)srow * s.ld + s.col0;
    if (s.vec) {
        float4 t = *reinterpret_cast<const float4 *>(base + (c < s.width ? c : 0));
        if (s.pre_act) {
            t.x = g4c::apply_act(t.x, s.pre_act); t.y = g4c::apply_act(t.y, s.pre_act);
            t.z = g4c::apply_act(t.z, s.pre_act); t.w = g4c::apply_act(t.w, s.pre_act);
        }
        return t;
    }
    // narrow / unaligned source: clamped unconditional loads, zero fill by select
    const int w1 = s.width - 1;
    float4 t;
    t.x = base[c + 0 < w1 ? c + 0 : w1];
    t.y = base[c + 1 < w1 ? c + 1 : w1];
    t.z = base[c + 2 < w1 ? c + 2 : w1];
    t.w = base[c + 3 < w1 ? c + 3 : w1];
    if (s.pre_act) {
        t.x = g4c::apply_act(t.x, s.pre_act); t.y = g4c::apply_act(t.y, s.pre_act);
        t.z = g4c::apply_act(t.z, s.pre_act); t.w = g4c::apply_act(t.w, s.pre_act);
    }
    t.x = (c + 0 < s.width) ? t.x : 0.f;
    t.y = (c + 1 < s.width) ? t.y : 0.f;
    t.z = (c + 2 < s.width) ? t.z : 0.f;
    t.w = (c + 3 < s.width) ? t.w : 0.f;
    return t;
}
__device__ __forceinline__ void load_x(const Src &s, const int *sRow, int k0, int tid, XRegs &r) {
    const int c = k0 + (tid & 7) * 4;
    r.a = ldx1(s, sRow[tid >> 3], c);
    r.b = ldx1(s, sRow[(tid >> 3) + 32], c);
}

__device__ __forceinline__ void stx1(float *d, const float4 &v) {   // 8-byte aligned (XS even), not 16
    *reinterpret_cast<float2 *>(d) = make_float2(v.x, v.y);
    *reinterpret_cast<float2 *>(d + 2) = make_float2(v.z, v.w);
}
__device__ __forceinline__ void store_x(float *sX, int cw, int tid, const XRegs &r) {
    const int c = (tid & 7) * 4;
    if (c < cw) {
        stx1(sX + (tid >> 3) * XS + c, r.a);
        stx1(sX + ((tid >> 3) + 32) * XS + c, r.b);
    }
}

__global__ __launch_bounds__(NTHREADS, 2) void mlp_fused_kernel(const Params p) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float *sX = lds;
    float *sW = sX + TM * XS;
    float *sH = sW + KC * MAXN;
    int *sRow = reinterpret_cast<int *>(sH + TM * HS);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int rt = wave & 1, ct0 = wave >> 1;

    // XCD-aware tile order: block b runs on XCD b%8; give each XCD a contiguous range of tiles so
    // neighbouring tiles (which gather the same node rows) share an L2.  Bijective for any n_tiles.
    int tile;
    {
        const int b = blockIdx.x, nt = p.n_tiles;
        const int q = nt >> 3, r = nt & 7, x = b & 7, j = b >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const long long row0 = (long long)tile * TM;

    if (tid < TM) {
        long long r = row0 + tid;
        if (r >= p.M) r = p.M - 1;
        for (int s = 0; s < p.n_src; ++s) sRow[s * TM + tid] = p.src[s].idx ? p.src[s].idx[r] : (int)r;
    }
    __syncthreads();

    f32x16 acc0, acc1;
    WRegs wr;
    XRegs xr;

    // ------------------------------------------------------------------ layer 0 (gathered input)
    {
        const int np = p.npad[0], nt = np >> 5;
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc0[q] = 0.f; acc1[q] = 0.f; }
        // chunk cursor over (source, k0)
        int s = 0, k0 = 0, kbase = 0;   // kbase: row offset of this chunk in the packed weights
        int cw = min(KC, p.src[0].wpad);
        load_x(p.src[0], sRow, 0, tid, xr);
        load_w(p.w[0], cw * np / 4, tid, wr);
        store_x(sX, cw, tid, xr);
        store_w(sW, cw * np / 4, tid, wr);
        __syncthreads();
        while (true) {
            // next chunk (this layer), or the first weight chunk of layer 1
            int ns = s, nk0 = k0 + cw, nkbase = kbase + cw;
            if (nk0 >= p.src[s].wpad) { ns = s + 1; nk0 = 0; }
            const bool more = ns < p.n_src;
            int ncw = 0;
            if (more) {
                ncw = min(KC, p.src[ns].wpad - nk0);
                load_x(p.src[ns], sRow + ns * TM, nk0, tid, xr);
                load_w(p.w[0] + (long long)nkbase * np, ncw * np / 4, tid, wr);
            } else {
                load_w(p.w[1], KC * p.npad[1] / 4, tid, wr);
            }
            mma_chunk(sX, XS, 0, sW, np, cw, rt, ct0, nt, i, h, acc0, acc1);
            __syncthreads();
            if (more) {
                store_x(sX, ncw, tid, xr);
                store_w(sW, ncw * np / 4, tid, wr);
                __syncthreads();
                s = ns; k0 = nk0; kbase = nkbase; cw = ncw;
            } else {
                break;
            }
        }
    }

    // ------------------------------------------------------------------ layers 1..L-1
    for (int l = 0; l < p.n_layers; ++l) {
        const int np = p.npad[l], nt = np >> 5;
        // epilogue of layer l: bias (+ SELU unless last) -> sH.  All waves have passed the barrier
        // that follows the last mma_chunk of this layer, so nobody still reads sH / sW.
        const bool last = (l == p.n_layers - 1);
        if (ct0 < nt) store_hidden(acc0, sH, p.b[l], ct0, rt, i, h, last);
        if (ct0 + 2 < nt) store_hidden(acc1, sH, p.b[l], ct0 + 2, rt, i, h, last);
        if (last) break;
        // layer l+1: K = npad[l] from sH, weight chunks streamed; first chunk is already in `wr`
        const int np1 = p.npad[l + 1], nt1 = np1 >> 5;
        const int K = np;
        store_w(sW, KC * np1 / 4, tid, wr);
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc0[q] = 0.f; acc1[q] = 0.f; }
        __syncthreads();
        for (int k0 = 0; k0 < K; k0 += KC) {
            const bool more = (k0 + KC) < K;
            if (more) {
                load_w(p.w[l + 1] + (long long)(k0 + KC) * np1, KC * np1 / 4, tid, wr);
            } else if (l + 2 < p.n_layers) {
                load_w(p.w[l + 2], KC * p.npad[l + 2] / 4, tid, wr);
            }
            mma_chunk(sH, HS, k0, sW, np1, KC, rt, ct0, nt1, i, h, acc0, acc1);
            __syncthreads();
            if (more) {
                store_w(sW, KC * np1 / 4, tid, wr);
                __syncthreads();
            }
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ LayerNorm / activation / store
    {
        const int g = tid >> 5, l = tid & 31;
        const int n_out = p.n_out;
        const int np = p.npad[p.n_layers - 1];
        const float inv_n = 1.0f / (float)n_out;
        const bool pair_store = ((n_out & 1) == 0) && ((p.out_ld & 1) == 0) && (((uintptr_t)p.out & 7) == 0);
        for (int r = g; r < TM; r += 8) {
            const long long grow = row0 + r;
            if (grow >= p.M) break;   // uniform per 32-lane group; no barriers below
            float x[4];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int c = 2 * l + 64 * m;
                if (c < np) {
                    const float2 t = *reinterpret_cast<const float2 *>(sH + r * HS + c);
                    x[2 * m] = t.x; x[2 * m + 1] = t.y;
                } else {
                    x[2 * m] = 0.f; x[2 * m + 1] = 0.f;
                }
            }
            if (p.gamma) {
                float s = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int c = 2 * l + 64 * m;
                    s += (c < n_out ? x[2 * m] : 0.f) + (c + 1 < n_out ? x[2 * m + 1] : 0.f);
                }
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 32);
                const float mean = s * inv_n;
                float v = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int c = 2 * l + 64 * m;
                    const float d0 = x[2 * m] - mean, d1 = x[2 * m + 1] - mean;
                    v += (c < n_out ? d0 * d0 : 0.f) + (c + 1 < n_out ? d1 * d1 : 0.f);
                }
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
                const float rstd = rsqrtf(v * inv_n + p.eps);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int c = 2 * l + 64 * m;
                    if (c < n_out) x[2 * m] = (x[2 * m] - mean) * rstd * p.gamma[c] + p.beta[c];
                    if (c + 1 < n_out) x[2 * m + 1] = (x[2 * m + 1] - mean) * rstd * p.gamma[c + 1] + p.beta[c + 1];
                }
            }
            const long long orow = p.out_idx ? p.out_idx[grow] : grow;
            float *po = p.out + orow * p.out_ld;
            const float *pr = p.resid ? p.resid + grow * p.resid_ld + p.resid_col0 : nullptr;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int c = 2 * l + 64 * m;
                float y0 = g4c::apply_act(x[2 * m], p.act), y1 = g4c::apply_act(x[2 * m + 1], p.act);
                if (pr) {
                    if (c < n_out) y0 += pr[c];
                    if (c + 1 < n_out) y1 += pr[c + 1];
                }
                if (pair_store) {
                    if (c < n_out) *reinterpret_cast<float2 *>(po + c) = make_float2(y0, y1);
                } else {
                    if (c < n_out) po[c] = y0;
                    if (c + 1 < n_out) po[c + 1] = y1;
                }
            }
        }
    }
}

// W[n_out, k_in] (nn.Linear layout) -> packed [k_pad/2][n_pad][2] with zero padding and optional
// per-block sign flip.  seg tables live in the kernel argument.
struct PackSegs {
    int n_seg;
    int width[G4C_MAX_SRC], wpad[G4C_MAX_SRC], neg[G4C_MAX_SRC];
};

__global__ void pack_layer_kernel(const float *__restrict__ W, int n_out, int k_in, PackSegs segs,
                                  float *__restrict__ packed, int k_pad, int n_pad) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= k_pad * n_pad) return;
    const int kp = gid / n_pad, n = gid % n_pad;
    // map padded k -> source column
    int k = -1, base_p = 0, base = 0, neg = 0;
    for (int s = 0; s < segs.n_seg; ++s) {
        if (kp >= base_p && kp < base_p + segs.wpad[s]) {
            const int j = kp - base_p;
            if (j < segs.width[s]) { k = base + j; neg = segs.neg[s]; }
        }
        base_p += segs.wpad[s];
        base += segs.width[s];
    }
    float v = 0.f;
    if (k >= 0 && n < n_out) v = W[(long long)n * k_in + k];
    if (neg) v = -v;
    packed[((long long)(kp >> 1) * n_pad + n) * 2 + (kp & 1)] = v;
}

int round_npad(int n) { return n <= 32 ? 32 : n <= 64 ? 64 : n <= 128 ? 128 : -1; }

}  // namespace

extern "C" int g4c_mlp_pack_layer(const float *W, int32_t n_out, int32_t k_in, const int32_t *seg_width,
                                  const int32_t *seg_negate, int32_t n_seg, float *packed,
                                  int32_t k_pad, int32_t n_pad, void *stream) {
    G4C_REQUIRE(W && packed && seg_width, G4C_EINVAL, "g4c_mlp_pack_layer: null pointer");
    G4C_REQUIRE(n_seg >= 1 && n_seg <= G4C_MAX_SRC, G4C_EUNSUPPORTED, "g4c_mlp_pack_layer: %d input blocks (max %d)", n_seg, G4C_MAX_SRC);
    PackSegs segs;
    segs.n_seg = n_seg;
    int ksum = 0, kpsum = 0;
    for (int s = 0; s < n_seg; ++s) {
        G4C_REQUIRE(seg_width[s] > 0, G4C_EINVAL, "g4c_mlp_pack_layer: empty input block %d", s);
        segs.width[s] = seg_width[s];
        segs.wpad[s] = (seg_width[s] + 3) / 4 * 4;
        segs.neg[s] = seg_negate ? seg_negate[s] : 0;
        ksum += segs.width[s];
        kpsum += segs.wpad[s];
    }
    G4C_REQUIRE(ksum == k_in, G4C_EINVAL, "g4c_mlp_pack_layer: blocks sum to %d columns, weight has %d", ksum, k_in);
    G4C_REQUIRE(kpsum == k_pad, G4C_EINVAL, "g4c_mlp_pack_layer: k_pad %d != %d", k_pad, kpsum);
    G4C_REQUIRE(n_pad == round_npad(n_out), G4C_EUNSUPPORTED, "g4c_mlp_pack_layer: layer width %d unsupported (max 128)", n_out);
    const int total = k_pad * n_pad;
    pack_layer_kernel<<<dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(W, n_out, k_in, segs, packed, k_pad, n_pad);
    return g4c::check_launch("g4c_mlp_pack_layer");
}

extern "C" int g4c_mlp_forward(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                               float *out, int32_t out_ld, const int32_t *out_idx, int32_t act,
                               const float *resid, int32_t resid_ld, int32_t resid_col0, void *stream) {
    G4C_REQUIRE(mlp && srcs && out, G4C_EINVAL, "g4c_mlp_forward: null pointer");
    G4C_REQUIRE(n_src >= 1 && n_src <= G4C_MAX_SRC, G4C_EUNSUPPORTED, "g4c_mlp_forward: %d sources (max %d)", n_src, G4C_MAX_SRC);
    G4C_REQUIRE(mlp->n_layers >= 2 && mlp->n_layers <= G4C_MAX_LAYERS, G4C_EUNSUPPORTED,
                "g4c_mlp_forward: %d layers (supported 2..%d)", mlp->n_layers, G4C_MAX_LAYERS);
    G4C_REQUIRE(n_rows >= 0 && n_rows < (1LL << 31), G4C_EINVAL, "g4c_mlp_forward: n_rows %lld out of range", (long long)n_rows);
    G4C_REQUIRE(act >= 0 && act <= 2, G4C_EINVAL, "g4c_mlp_forward: bad activation %d", act);
    if (n_rows == 0) return G4C_OK;
    Params p;
    p.n_src = n_src;
    int kp = 0;
    for (int s = 0; s < n_src; ++s) {
        const g4c_src_t &g = srcs[s];
        G4C_REQUIRE(g.ptr && g.width > 0 && g.ld >= g.col0 + g.width && g.col0 >= 0, G4C_EINVAL,
                    "g4c_mlp_forward: bad source %d (width=%d ld=%d col0=%d)", s, g.width, g.ld, g.col0);
        G4C_REQUIRE(g.pre_act >= 0 && g.pre_act <= 2, G4C_EINVAL, "g4c_mlp_forward: source %d bad pre_act %d", s, g.pre_act);
        Src &d = p.src[s];
        d.ptr = g.ptr; d.idx = g.idx; d.width = g.width; d.wpad = (g.width + 3) / 4 * 4; d.ld = g.ld; d.col0 = g.col0; d.pre_act = g.pre_act;
        d.vec = (g.width % 4 == 0) && (g.ld % 4 == 0) && (g.col0 % 4 == 0) && ((uintptr_t)g.ptr % 16 == 0);
        kp += d.wpad;
    }
    for (int s = n_src; s < G4C_MAX_SRC; ++s) p.src[s] = p.src[0];
    G4C_REQUIRE(kp == mlp->k_pad[0], G4C_EINVAL, "g4c_mlp_forward: sources give %d padded columns, layer 1 packed for %d", kp, mlp->k_pad[0]);
    p.n_layers = mlp->n_layers;
    for (int l = 0; l < G4C_MAX_LAYERS; ++l) {
        const bool on = l < mlp->n_layers;
        p.kpad[l] = on ? mlp->k_pad[l] : 0;
        p.npad[l] = on ? mlp->n_pad[l] : 0;
        p.w[l] = on ? mlp->w[l] : nullptr;
        p.b[l] = on ? mlp->b[l] : nullptr;
        if (on) {
            G4C_REQUIRE(p.w[l] && p.b[l], G4C_EINVAL, "g4c_mlp_forward: layer %d has null weights", l);
            G4C_REQUIRE(p.npad[l] == 32 || p.npad[l] == 64 || p.npad[l] == 128, G4C_EUNSUPPORTED,
                        "g4c_mlp_forward: layer %d padded width %d (supported 32/64/128)", l, p.npad[l]);
            if (l > 0) G4C_REQUIRE(p.kpad[l] == p.npad[l - 1], G4C_EINVAL, "g4c_mlp_forward: layer %d k_pad %d != previous n_pad %d", l, p.kpad[l], p.npad[l - 1]);
        }
    }
    p.gamma = mlp->ln_gamma; p.beta = mlp->ln_beta; p.eps = mlp->ln_eps;
    G4C_REQUIRE((p.gamma == nullptr) == (p.beta == nullptr), G4C_EINVAL, "g4c_mlp_forward: LayerNorm needs both gamma and beta");
    p.n_out = mlp->n_out;
    G4C_REQUIRE(p.n_out > 0 && p.n_out <= p.npad[p.n_layers - 1] && out_ld >= p.n_out, G4C_EINVAL,
                "g4c_mlp_forward: n_out=%d out_ld=%d", p.n_out, out_ld);
    p.M = n_rows;
    p.out = out; p.out_ld = out_ld; p.out_idx = out_idx; p.act = act;
    p.resid = resid; p.resid_ld = resid_ld; p.resid_col0 = resid_col0;
    p.n_tiles = (int)((n_rows + TM - 1) / TM);
    mlp_fused_kernel<<<dim3(p.n_tiles), dim3(NTHREADS), 0, (hipStream_t)stream>>>(p);
    return g4c::check_launch("g4c_mlp_forward");
}
