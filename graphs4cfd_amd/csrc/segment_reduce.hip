// Segmented sum / mean over CSR-by-destination (the "scatter-sum" of the hot path).
//
// Replaces torch_geometric.utils.scatter(reduce='sum'|'mean') as called by the reference at
// graphs4cfd/nn/blocks.py:183 (GNBlock), :231 (DownMP), :330 (EdgeMP), :378 (DownEdgeMP), and the
// feature reduction of coalesce(reduce='mean') at :67 (pool_edge).  No atomics: the static mesh
// plan sorts messages by destination once, each destination row is owned by one group of LPR
// lanes that streams its messages as 16-byte loads and adds them in plan order (so a stable plan
// reproduces a sequential scatter_add_ bit for bit).
//
// HBM-bound.  Algorithmic bytes per call: E*W*4 (messages) + S*W*4 (output) + (S+1)*4 (offsets)
// [+ E*4 when a permutation is read].  For W = 128 one message row is 512 B = 32 lanes x float4,
// a wave covers two destination rows per load instruction and keeps up to UNROLL rows in flight.
#include "g4c_common.h"
#include <cstdlib>

namespace {

constexpr int UNROLL = 8;
#ifndef G4C_SEG_SPLIT_BATCH
#define G4C_SEG_SPLIT_BATCH 1
#endif

template <int LPR>
__global__ __launch_bounds__(256) void segment_reduce_kernel(
    const float *__restrict__ src, int src_ld, const int *__restrict__ perm,
    const int *__restrict__ off, int n_seg, int width, int mean, int src_act, int act,
    float *__restrict__ out, int out_ld) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int seg = (int)(gid / LPR);
    const int l = (int)(gid % LPR);
    if (seg >= n_seg) return;
    const int beg = off[seg], end = off[seg + 1];
    const float cnt = (float)((end - beg) > 1 ? (end - beg) : 1);
    for (int c = l * 4; c < width; c += LPR * 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // Loads are issued unconditionally from a clamped row (a per-element "load or not" branch
        // would serialise them); only the adds are predicated.
        for (int p = beg; p < end; p += UNROLL) {
            float4 v[UNROLL];
            // the second half of a batch only when one of its rows exists (coarse edges pool ~4 fine edges: half the row requests)
            const bool more = G4C_SEG_SPLIT_BATCH && (p + UNROLL / 2 >= end);
#pragma unroll
            for (int u = 0; u < UNROLL / 2; ++u) {
                const int pp = (p + u < end) ? (p + u) : (end - 1);
                const long long r = perm ? perm[pp] : pp;
                v[u] = *reinterpret_cast<const float4 *>(src + r * src_ld + c);
            }
            if (!more) {
#pragma unroll
                for (int u = UNROLL / 2; u < UNROLL; ++u) {
                    const int pp = (p + u < end) ? (p + u) : (end - 1);
                    const long long r = perm ? perm[pp] : pp;
                    v[u] = *reinterpret_cast<const float4 *>(src + r * src_ld + c);
                }
            } else {
#pragma unroll
                for (int u = UNROLL / 2; u < UNROLL; ++u) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (src_act) {   // wave-uniform: the activation only on rows that exist (a coarse edge pools ~3 fine edges: a SELU on all
                             // eight slots of the batch was a third of this launch's time); same additions in the same order
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    if (p + u < end) {
                        acc.x += g4c::apply_act(v[u].x, src_act); acc.y += g4c::apply_act(v[u].y, src_act);
                        acc.z += g4c::apply_act(v[u].z, src_act); acc.w += g4c::apply_act(v[u].w, src_act);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const bool on = (p + u < end);
                    acc.x += on ? v[u].x : 0.f; acc.y += on ? v[u].y : 0.f;
                    acc.z += on ? v[u].z : 0.f; acc.w += on ? v[u].w : 0.f;
                }
            }
        }
        if (mean) { acc.x /= cnt; acc.y /= cnt; acc.z /= cnt; acc.w /= cnt; }
        acc.x = g4c::apply_act(acc.x, act); acc.y = g4c::apply_act(acc.y, act);
        acc.z = g4c::apply_act(acc.z, act); acc.w = g4c::apply_act(acc.w, act);
        *reinterpret_cast<float4 *>(out + (long long)seg * out_ld + c) = acc;
    }
}

// generic-width fallback (width or strides not multiples of 4): one lane per column
__global__ __launch_bounds__(256) void segment_reduce_scalar_kernel(
    const float *__restrict__ src, int src_ld, const int *__restrict__ perm,
    const int *__restrict__ off, int n_seg, int width, int mean, int src_act, int act,
    float *__restrict__ out, int out_ld) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int seg = (int)(gid / width);
    const int c = (int)(gid % width);
    if (seg >= n_seg) return;
    const int beg = off[seg], end = off[seg + 1];
    float acc = 0.f;
    for (int p = beg; p < end; ++p) {
        const long long r = perm ? perm[p] : p;
        acc += g4c::apply_act(src[r * src_ld + c], src_act);
    }
    if (mean) acc /= (float)((end - beg) > 1 ? (end - beg) : 1);
    out[(long long)seg * out_ld + c] = g4c::apply_act(acc, act);
}

template <int LPR>
__global__ __launch_bounds__(256) void weighted_segment_mean_kernel(
    const float *__restrict__ x, int x_ld, const int *__restrict__ x_idx, const float *__restrict__ w,
    const int *__restrict__ off, int n_seg, int width, float *__restrict__ out, int out_ld,
    const int *__restrict__ out_idx) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int seg = (int)(gid / LPR);
    const int l = (int)(gid % LPR);
    if (seg >= n_seg) return;
    const int beg = off[seg], end = off[seg + 1];
    const long long orow = out_idx ? out_idx[seg] : seg;
    float den = 0.f;
    for (int p = beg; p < end; ++p) den += w[p];
    for (int c = l; c < width; c += LPR) {
        float num = 0.f;
        for (int p = beg; p < end; ++p) num += x[(long long)x_idx[p] * x_ld + c] * w[p];
        out[orow * out_ld + c] = num / den;
    }
}

// LayerNorm over rows of ANY width (+ activation), in place or out of place: one wave per row, the row's values kept in registers for
// up to 64 * LN_REGS columns (two passes over the values, like F.layer_norm: mean, then the centred sum of squares), re-read from
// memory beyond that.  For the fused MLP kernels' envelope: their own LayerNorm epilogue takes <= 128 columns (MLP._run_stages).
constexpr int LN_REGS = 16;
__global__ __launch_bounds__(256) void layer_norm_rows_kernel(const float *__restrict__ x, int x_ld, long long n_rows, int width,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta, float eps, int act,
                                                             float *__restrict__ out, int out_ld) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    const float *xr = x + row * x_ld;
    float v[LN_REGS];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < LN_REGS; ++j) {
        const int c = lane + 64 * j;
        v[j] = c < width ? xr[c] : 0.f;
        sum += v[j];
    }
    for (int c = lane + 64 * LN_REGS; c < width; c += 64) sum += xr[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)width;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < LN_REGS; ++j) {
        const float d = v[j] - mean;
        var += (lane + 64 * j < width) ? d * d : 0.f;
    }
    for (int c = lane + 64 * LN_REGS; c < width; c += 64) { const float d = xr[c] - mean; var += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
    const float rstd = rsqrtf(var / (float)width + eps);
    float *orow = out + row * out_ld;
#pragma unroll
    for (int j = 0; j < LN_REGS; ++j) {
        const int c = lane + 64 * j;
        if (c < width) orow[c] = g4c::apply_act(fmaf((v[j] - mean) * rstd, gamma ? gamma[c] : 1.f, beta ? beta[c] : 0.f), act);
    }
    for (int c = lane + 64 * LN_REGS; c < width; c += 64)
        orow[c] = g4c::apply_act(fmaf((xr[c] - mean) * rstd, gamma ? gamma[c] : 1.f, beta ? beta[c] : 0.f), act);
}

}  // namespace

extern "C" int g4c_segment_reduce(const float *src, int32_t src_ld, const int32_t *perm, const int32_t *off,
                                  int32_t n_seg, int32_t width, int32_t mean, int32_t src_act, int32_t act,
                                  float *out, int32_t out_ld, void *stream) {
    G4C_REQUIRE(n_seg >= 0 && width > 0 && src_ld >= width && out_ld >= width, G4C_EINVAL,
                "g4c_segment_reduce: bad sizes n_seg=%d width=%d src_ld=%d out_ld=%d", n_seg, width, src_ld, out_ld);
    if (n_seg == 0) return G4C_OK;
    G4C_REQUIRE(off != nullptr && out != nullptr, G4C_EINVAL, "g4c_segment_reduce: null pointer");
    g4c::DeviceGuard on_device(out);
    G4C_REQUIRE(src != nullptr || perm == nullptr, G4C_EINVAL, "g4c_segment_reduce: null src with a permutation");
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (width % 4 == 0) && (src_ld % 4 == 0) && (out_ld % 4 == 0) &&
                     ((uintptr_t)src % 16 == 0) && ((uintptr_t)out % 16 == 0);
    if (!vec) {
        const long long total = (long long)n_seg * width;
        segment_reduce_scalar_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(
            src, src_ld, perm, off, n_seg, width, mean, src_act, act, out, out_ld);
        return g4c::check_launch("g4c_segment_reduce");
    }
    // (32 lanes per 128-wide row: 16 / 8 lanes with two / four pieces each measured 15 / 45 % slower on pool_edge's short segments)
    const int q = width / 4;
    if (q > 16) {
        const long long total = (long long)n_seg * 32;
        segment_reduce_kernel<32><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(
            src, src_ld, perm, off, n_seg, width, mean, src_act, act, out, out_ld);
    } else if (q > 8) {
        const long long total = (long long)n_seg * 16;
        segment_reduce_kernel<16><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(
            src, src_ld, perm, off, n_seg, width, mean, src_act, act, out, out_ld);
    } else {
        const long long total = (long long)n_seg * 8;
        segment_reduce_kernel<8><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(
            src, src_ld, perm, off, n_seg, width, mean, src_act, act, out, out_ld);
    }
    return g4c::check_launch("g4c_segment_reduce");
}

extern "C" int g4c_weighted_segment_mean(const float *x, int32_t x_ld, const int32_t *x_idx, const float *w,
                                         const int32_t *off, int32_t n_seg, int32_t width,
                                         float *out, int32_t out_ld, const int32_t *out_idx, void *stream) {
    G4C_REQUIRE(n_seg >= 0 && width > 0 && x_ld >= width && out_ld >= width, G4C_EINVAL,
                "g4c_weighted_segment_mean: bad sizes n_seg=%d width=%d", n_seg, width);
    G4C_REQUIRE(x && x_idx && w && off && out, G4C_EINVAL, "g4c_weighted_segment_mean: null pointer");
    g4c::DeviceGuard on_device(out);
    if (n_seg == 0) return G4C_OK;
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)n_seg * 64;
    weighted_segment_mean_kernel<64><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(
        x, x_ld, x_idx, w, off, n_seg, width, out, out_ld, out_idx);
    return g4c::check_launch("g4c_weighted_segment_mean");
}

// test hook: the quotient routine of the fused aggregation's mean (g4c::mean_div4) on arrays — out[i] = a[i] / count[i / 4]
__global__ void mean_div_check_kernel(const float *__restrict__ a, const int *__restrict__ count, float *__restrict__ out, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const g4c::mean_f32x4 v = *reinterpret_cast<const g4c::mean_f32x4 *>(a + 4 * i);
    *reinterpret_cast<g4c::mean_f32x4 *>(out + 4 * i) = g4c::mean_div4(v, count[i]);
}
extern "C" int g4c_debug_mean_div(const float *a, const int32_t *count, float *out, int64_t n4, void *stream) {
    G4C_REQUIRE(a && count && out && n4 >= 0, G4C_EINVAL, "g4c_debug_mean_div: null argument");
    if (n4 == 0) return G4C_OK;
    g4c::DeviceGuard guard(a);
    mean_div_check_kernel<<<dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(a, count, out, n4);
    return g4c::check_launch("g4c_debug_mean_div");
}

extern "C" int g4c_layer_norm(const float *x, int32_t x_ld, int64_t n_rows, int32_t width, const float *gamma, const float *beta, float eps,
                              int32_t act, float *out, int32_t out_ld, void *stream) {
    G4C_REQUIRE(n_rows >= 0 && width > 0 && x_ld >= width && out_ld >= width, G4C_EINVAL,
                "g4c_layer_norm: bad sizes n_rows=%lld width=%d x_ld=%d out_ld=%d", (long long)n_rows, width, x_ld, out_ld);
    G4C_REQUIRE(act == G4C_ACT_NONE || act == G4C_ACT_SELU || act == G4C_ACT_TANH, G4C_EINVAL, "g4c_layer_norm: unknown activation %d", act);
    if (n_rows == 0) return G4C_OK;
    G4C_REQUIRE(x && out, G4C_EINVAL, "g4c_layer_norm: null pointer");
    // (in place is fine at any width: every read of the statistics passes precedes the wave's first write, and the last pass re-reads a
    // column in the lane that then writes it)
    g4c::DeviceGuard on_device(out);
    layer_norm_rows_kernel<<<dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(x, x_ld, n_rows, width, gamma, beta, eps, act, out, out_ld);
    return g4c::check_launch("g4c_layer_norm");
}
