// Backward-pass kernels of the training path (gfx950).  All HBM-bound: one pass over their operands, coalesced rows,
// deterministic reductions (no atomics), so two runs of a training step give bit-identical gradients.
// The weight / bias gradients of the 128-wide layers are weight_grad_kernel below (MFMA, one pass over both operands); the
// input-gradient products run as launches of the fused-MLP kernel (mlp_fused.hip), rocBLAS only for shapes outside that
// envelope (graphs4cfd_amd/autograd.py); everything around them is here.
//
// Reference being differentiated: graphs4cfd/nn/blocks.py:117-144 (MLP), :175-186 (GNBlock), :219-237 (DownMP),
// :265-290 (UpMP); the reference relies on torch autograd over cat / index / Linear / SELU / LayerNorm / scatter.
#include "g4c_common.h"

namespace {

using g4c::apply_act;

// d act(x) / dx given either the activation's output (from_input = 0) or its input (from_input = 1)
__device__ __forceinline__ float act_slope(float ref, int act, int from_input) {
    const float alpha = 1.6732632423543772848170429916717f;
    const float scale = 1.0507009873554804934193349852946f;
    if (act == G4C_ACT_SELU) {
        const float y = from_input ? g4c::selu_f(ref) : ref;
        return y > 0.f ? scale : y + scale * alpha;              // x <= 0: d/dx scale*alpha*(e^x - 1) = y + scale*alpha
    }
    if (act == G4C_ACT_TANH) {
        const float y = from_input ? g4c::tanh_f(ref) : ref;
        return 1.f - y * y;
    }
    return 1.f;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// V = 4: four consecutive columns per thread (16-byte loads / stores; the launcher checks alignment), V = 1: any shape.
// ACC: dst += (the recomputed first layer adds the pre-multiplied node-side terms, gathered through their index).
template <int V, bool ACC>
__global__ __launch_bounds__(256) void train_gather_kernel(const float *__restrict__ src, int src_ld, int scol0,
                                                           const int *__restrict__ idx, int pre_act, float sign,
                                                           float *__restrict__ dst, int dst_ld, int dcol0, int width,
                                                           long long n_rows) {
    const int wv = width / V;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long r = t / wv;
    const int c = (int)(t - r * wv) * V;
    if (r >= n_rows) return;
    const long long sr = idx ? (long long)idx[r] : r;
    const float *sp = src + sr * src_ld + scol0 + c;
    float *dp = dst + r * dst_ld + dcol0 + c;
    if (V == 4) {
        f32x4 x = *reinterpret_cast<const f32x4 *>(sp);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = sign * apply_act(x[e], pre_act);
        if (ACC) x += *reinterpret_cast<const f32x4 *>(dp);
        *reinterpret_cast<f32x4 *>(dp) = x;
    } else {
        const float x = sign * apply_act(*sp, pre_act);
        *dp = ACC ? *dp + x : x;
    }
}

template <int V>
__global__ __launch_bounds__(256) void act_grad_kernel(const float *__restrict__ dy, int dy_ld, const float *__restrict__ ref,
                                                       int ref_ld, int from_input, int act, float *__restrict__ dz, int dz_ld,
                                                       int width, long long n_rows) {
    const int wv = width / V;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long r = t / wv;
    const int c = (int)(t - r * wv) * V;
    if (r >= n_rows) return;
    if (V == 4) {
        const f32x4 g = *reinterpret_cast<const f32x4 *>(dy + r * dy_ld + c);
        const f32x4 x = *reinterpret_cast<const f32x4 *>(ref + r * ref_ld + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = g[e] * act_slope(x[e], act, from_input);
        *reinterpret_cast<f32x4 *>(dz + r * dz_ld + c) = o;
    } else {
        dz[r * dz_ld + c] = dy[r * dy_ld + c] * act_slope(ref[r * ref_ld + c], act, from_input);
    }
}

inline bool vec4_ok(const void *p, int ld, int col0) { return ((uintptr_t)p % 16 == 0) && ld % 4 == 0 && col0 % 4 == 0; }

// LayerNorm backward, one wave per row (width <= 256: up to 4 columns per lane), 4 rows per workgroup iteration.
//   xhat = (z - mean) * rstd;  g = dy * gamma;  dz = rstd * (g - mean(g) - xhat * mean(g * xhat))
// dgamma / dbeta: every workgroup keeps running column sums over its rows and writes one partial row
// [dgamma(width) | dbeta(width)]; g4c_colsum adds the partial rows in a fixed order.
constexpr int LN_MAXC = 4;
__global__ __launch_bounds__(256) void layernorm_grad_kernel(const float *__restrict__ z, int z_ld, const float *__restrict__ gamma,
                                                             const float *__restrict__ dy, int dy_ld, float *__restrict__ dz,
                                                             int dz_ld, float *__restrict__ partial, int width,
                                                             long long n_rows, float eps) {
    __shared__ float red[4][2 * 64 * LN_MAXC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dg[LN_MAXC], db[LN_MAXC], gam[LN_MAXC];
#pragma unroll
    for (int j = 0; j < LN_MAXC; ++j) {
        dg[j] = db[j] = 0.f;
        const int c = lane + 64 * j;
        gam[j] = c < width ? gamma[c] : 0.f;
    }
    const float inv_w = 1.f / (float)width;
    for (long long r = (long long)blockIdx.x * 4 + wave; r < n_rows; r += (long long)gridDim.x * 4) {
        float zv[LN_MAXC], gy[LN_MAXC], s = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXC; ++j) {
            const int c = lane + 64 * j;
            zv[j] = c < width ? z[r * z_ld + c] : 0.f;
            gy[j] = c < width ? dy[r * dy_ld + c] : 0.f;
            s += zv[j];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * inv_w;
        float var = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAXC; ++j) {
            const int c = lane + 64 * j;
            const float d = c < width ? zv[j] - mean : 0.f;
            var += d * d;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
        const float rstd = rsqrtf(var * inv_w + eps);
        float m1 = 0.f, m2 = 0.f, xh[LN_MAXC];
#pragma unroll
        for (int j = 0; j < LN_MAXC; ++j) {
            const int c = lane + 64 * j;
            xh[j] = c < width ? (zv[j] - mean) * rstd : 0.f;
            const float g = gy[j] * gam[j];
            m1 += g;
            m2 += g * xh[j];
            dg[j] += gy[j] * xh[j];
            db[j] += gy[j];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m1 += __shfl_xor(m1, o, 64); m2 += __shfl_xor(m2, o, 64); }
        m1 *= inv_w; m2 *= inv_w;
#pragma unroll
        for (int j = 0; j < LN_MAXC; ++j) {
            const int c = lane + 64 * j;
            if (c < width) dz[r * dz_ld + c] = rstd * (gy[j] * gam[j] - m1 - xh[j] * m2);
        }
    }
#pragma unroll
    for (int j = 0; j < LN_MAXC; ++j) { red[wave][lane + 64 * j] = dg[j]; red[wave][64 * LN_MAXC + lane + 64 * j] = db[j]; }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * width; c += 256) {
        const int k = c < width ? c : 64 * LN_MAXC + (c - width);
        partial[(long long)blockIdx.x * 2 * width + c] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    }
}

// column sums of x[n_rows, width].  Stage 1: workgroup (g, cb) adds rows g*chunk .. (g+1)*chunk of column block cb in a fixed
// order — a 256-thread workgroup covers RPI = 256 / min(width, 256) rows at a time (thread = (row offset, column), so every
// wave reads whole contiguous row pieces), eight loads in flight per thread; the RPI partial sums of a column are then added
// in order through LDS.  Stage 2 adds the partial rows, again in order.
__global__ __launch_bounds__(256) void colsum_stage_kernel(const float *__restrict__ x, int ld, int width, long long n_rows,
                                                           long long chunk, float *__restrict__ out, int out_ld) {
    __shared__ float red[256];
    const int wb = width < 256 ? width : 256;                 // columns handled by this workgroup
    const int rpi = 256 / wb;                                 // rows per iteration
    const int roff = threadIdx.x / wb, cl = threadIdx.x - roff * wb;
    const int c = blockIdx.y * 256 + cl;
    const long long r0 = (long long)blockIdx.x * chunk;
    const long long r1 = r0 + chunk < n_rows ? r0 + chunk : n_rows;
    float s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = 0.f;
    if (roff < rpi && c < width) {
        long long r = r0 + roff;
        for (; r + 7LL * rpi < r1; r += 8LL * rpi) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += x[(r + (long long)u * rpi) * ld + c];
        }
        for (; r < r1; r += rpi) s[0] += x[r * ld + c];
    }
    red[threadIdx.x] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (roff == 0 && c < width) {
        float t = red[cl];
        for (int j = 1; j < rpi; ++j) t += red[j * wb + cl];
        out[(long long)blockIdx.x * out_ld + c] = t;
    }
}

// Weight gradient dW[n, k] = sum_m g[m, n] a[m, k] (and db[n] = sum_m g[m, n]) for n, k = 128: a contraction over up to
// millions of rows into ONE 128 x 128 tile.  HBM-bound: g and a are each read exactly once (600k rows: 614 MB; fp32 MFMA time
// for the same rows 125 us = 4.9 TB/s, so the two rooflines meet).  Workgroup w owns a contiguous chunk of rows; per 32-row slab
// it stages g and a in LDS (16-byte coalesced loads, prefetched into registers one slab ahead) and runs
// v_mfma_f32_32x32x2_f32 with A = g^T (wave = one 32-row block of n), B = a (four 32-column blocks of k): the row index m is the
// MFMA's contraction index, two rows per instruction.  Every workgroup writes its partial tile [128*128 | 128 column sums of
// g]; g4c_colsum adds the partial tiles in a fixed order (deterministic, no atomics).
constexpr int WG_N = 128, WG_SLAB = 32;
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Eight waves per workgroup: wave = (32-row block of n, half of k), so a wave holds 2 x 16 accumulators and the 512 threads
// stage a slab with two 16-byte loads per operand each — few enough registers to keep TWO slabs in flight (the global loads
// of slab s+2 are issued before the MFMAs of slab s), which one slab of look-ahead (1.7 us of MFMA work) did not cover.
constexpr int WG_THREADS = 512;
__global__ __launch_bounds__(WG_THREADS, 2) void weight_grad_kernel(const float *__restrict__ g, int g_ld, const float *__restrict__ a,
                                                                    int a_ld, long long n_rows, long long chunk,
                                                                    float *__restrict__ partial, int with_bias) {
    __shared__ __attribute__((aligned(16))) float sG[WG_SLAB * WG_N];
    __shared__ __attribute__((aligned(16))) float sA[WG_SLAB * WG_N];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = wave & 3, kh = wave >> 2;
    const int i = lane & 31, h = lane >> 5;
    const long long r0 = (long long)blockIdx.x * chunk;
    const long long r1 = r0 + chunk < n_rows ? r0 + chunk : n_rows;
    const int c4 = (tid & 31) * 4, rr = tid >> 5;               // this thread's 4 columns / first row inside a slab (rows rr, rr + 16)
    f32x16 acc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[kb][j] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    f32x4 pg[2][2], pa[2][2];                                   // [slab parity][row of the pair]
    auto fetch = [&](int q, long long m0) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const long long r = m0 + rr + 16 * it;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            pg[q][it] = r < r1 ? *reinterpret_cast<const f32x4 *>(g + r * g_ld + c4) : z;
            pa[q][it] = r < r1 ? *reinterpret_cast<const f32x4 *>(a + r * a_ld + c4) : z;
        }
    };
    auto slab = [&](int q, long long m0) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            *reinterpret_cast<f32x4 *>(sG + (rr + 16 * it) * WG_N + c4) = pg[q][it];
            *reinterpret_cast<f32x4 *>(sA + (rr + 16 * it) * WG_N + c4) = pa[q][it];
            bsum += pg[q][it];
        }
        __syncthreads();
        if (m0 + 2 * WG_SLAB < r1) fetch(q, m0 + 2 * WG_SLAB);   // two slabs ahead
        // operands of step s+1 are read from LDS before the MFMAs of step s are issued (LDS latency under the matrix pipe)
        const float *pG = sG + h * WG_N + nb * 32 + i, *pA = sA + h * WG_N + kh * 64 + i;
        float av = pG[0], bv[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) bv[kb] = pA[kb * 32];
#pragma unroll
        for (int s = 0; s < WG_SLAB / 2; ++s) {
            float av_n = 0.f, bv_n[2] = {0.f, 0.f};
            if (s + 1 < WG_SLAB / 2) {
                av_n = pG[(2 * s + 2) * WG_N];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) bv_n[kb] = pA[(2 * s + 2) * WG_N + kb * 32];
            }
            __builtin_amdgcn_sched_barrier(0);        // (the scheduler otherwise sinks these reads to just before their use)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) acc[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[kb], acc[kb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            av = av_n;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) bv[kb] = bv_n[kb];
        }
        __syncthreads();
    };
    if (r0 < r1) fetch(0, r0);
    if (r0 + WG_SLAB < r1) fetch(1, r0 + WG_SLAB);
    for (long long m0 = r0; m0 < r1; m0 += 2 * WG_SLAB) {
        slab(0, m0);
        if (m0 + WG_SLAB < r1) slab(1, m0 + WG_SLAB);
    }
    float *out = partial + (long long)blockIdx.x * (WG_N * WG_N + WG_N);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int n = nb * 32 + 4 * h + 8 * (j >> 2) + (j & 3);
            out[n * WG_N + kh * 64 + kb * 32 + i] = acc[kb][j];
        }
    if (with_bias) {                                            // column sums of g: the sixteen threads sharing c4, in order
        *reinterpret_cast<f32x4 *>(sG + rr * WG_N + c4) = bsum;
        __syncthreads();
        if (tid < WG_N) {
            float t = sG[tid];
#pragma unroll
            for (int q = 1; q < 16; ++q) t += sG[q * WG_N + tid];
            out[WG_N * WG_N + tid] = t;
        }
    }
}

// adjoint of the segmented sum / mean: every row of a segment receives the segment's gradient (/ max(count, 1))
__global__ __launch_bounds__(256) void segment_broadcast_kernel(const float *__restrict__ dout, int dout_ld,
                                                                const int *__restrict__ off, const int *__restrict__ perm,
                                                                int n_seg, int width, int mean, float *__restrict__ dsrc,
                                                                int dsrc_ld) {
    const int lane = threadIdx.x & 63;
    const int seg = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= n_seg) return;
    const int b = off[seg], e = off[seg + 1];
    const float scale = mean ? 1.f / (float)(e - b > 1 ? e - b : 1) : 1.f;
    for (int c = lane; c < width; c += 64) {
        const float g = dout[(long long)seg * dout_ld + c] * scale;
        for (int p = b; p < e; ++p) {
            const long long r = perm ? (long long)perm[p] : (long long)p;
            dsrc[r * dsrc_ld + c] = g;
        }
    }
}

}  // namespace

extern "C" int g4c_train_gather(const float *src, int32_t src_ld, int32_t scol0, const int32_t *idx, int32_t pre_act,
                                int32_t negate, float *dst, int32_t dst_ld, int32_t dcol0, int32_t width, int64_t n_rows,
                                int32_t accumulate, void *stream) {
    G4C_REQUIRE(src && dst, G4C_EINVAL, "g4c_train_gather: null pointer");
    g4c::DeviceGuard on_device(dst);
    G4C_REQUIRE(width > 0 && n_rows >= 0 && scol0 >= 0 && dcol0 >= 0 && src_ld >= scol0 + width && dst_ld >= dcol0 + width &&
                    pre_act >= 0 && pre_act <= 2,
                G4C_EINVAL, "g4c_train_gather: bad arguments width=%d src_ld=%d dst_ld=%d pre_act=%d", width, src_ld, dst_ld, pre_act);
    if (n_rows == 0) return G4C_OK;
    const bool v4 = width % 4 == 0 && vec4_ok(src, src_ld, scol0) && vec4_ok(dst, dst_ld, dcol0);
    const long long total = n_rows * (width / (v4 ? 4 : 1));
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    const float sign = negate ? -1.f : 1.f;
    hipStream_t st = (hipStream_t)stream;
#define G4C_TG(V, ACC) train_gather_kernel<V, ACC><<<grid, blk, 0, st>>>(src, src_ld, scol0, idx, pre_act, sign, dst, dst_ld, dcol0, width, n_rows)
    if (v4) { if (accumulate) G4C_TG(4, true); else G4C_TG(4, false); }
    else { if (accumulate) G4C_TG(1, true); else G4C_TG(1, false); }
#undef G4C_TG
    return g4c::check_launch("g4c_train_gather");
}

extern "C" int g4c_act_grad(const float *dy, int32_t dy_ld, const float *ref, int32_t ref_ld, int32_t from_input, int32_t act,
                            float *dz, int32_t dz_ld, int32_t width, int64_t n_rows, void *stream) {
    G4C_REQUIRE(dy && ref && dz, G4C_EINVAL, "g4c_act_grad: null pointer");
    g4c::DeviceGuard on_device(dz);
    G4C_REQUIRE(width > 0 && n_rows >= 0 && dy_ld >= width && ref_ld >= width && dz_ld >= width && act >= 0 && act <= 2, G4C_EINVAL,
                "g4c_act_grad: bad arguments width=%d lds=%d,%d,%d act=%d", width, dy_ld, ref_ld, dz_ld, act);
    if (n_rows == 0) return G4C_OK;
    const bool v4 = width % 4 == 0 && vec4_ok(dy, dy_ld, 0) && vec4_ok(ref, ref_ld, 0) && vec4_ok(dz, dz_ld, 0);
    const long long total = n_rows * (width / (v4 ? 4 : 1));
    const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
    if (v4) act_grad_kernel<4><<<grid, blk, 0, (hipStream_t)stream>>>(dy, dy_ld, ref, ref_ld, from_input, act, dz, dz_ld, width, n_rows);
    else act_grad_kernel<1><<<grid, blk, 0, (hipStream_t)stream>>>(dy, dy_ld, ref, ref_ld, from_input, act, dz, dz_ld, width, n_rows);
    return g4c::check_launch("g4c_act_grad");
}

extern "C" int32_t g4c_layernorm_grad_partials(int64_t n_rows) {
    const long long want = (n_rows + 63) / 64;       // >= 16 rows per wave before a second workgroup pays off
    return (int32_t)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
}

extern "C" int g4c_layernorm_grad(const float *z, int32_t z_ld, const float *gamma, const float *dy, int32_t dy_ld, float *dz,
                                  int32_t dz_ld, float *partial, int32_t width, int64_t n_rows, float eps, void *stream) {
    G4C_REQUIRE(z && gamma && dy && dz && partial, G4C_EINVAL, "g4c_layernorm_grad: null pointer");
    g4c::DeviceGuard on_device(dz);
    G4C_REQUIRE(width > 0 && width <= 64 * LN_MAXC && n_rows >= 0 && z_ld >= width && dy_ld >= width && dz_ld >= width,
                G4C_EUNSUPPORTED, "g4c_layernorm_grad: width %d (max %d) / leading dimensions", width, 64 * LN_MAXC);
    const int n_wg = g4c_layernorm_grad_partials(n_rows);
    layernorm_grad_kernel<<<dim3(n_wg), dim3(256), 0, (hipStream_t)stream>>>(z, z_ld, gamma, dy, dy_ld, dz, dz_ld, partial, width,
                                                                              n_rows, eps);
    return g4c::check_launch("g4c_layernorm_grad");
}

extern "C" int32_t g4c_colsum_partials(int64_t n_rows) {
    const long long want = (n_rows + 127) / 128;
    return (int32_t)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
}

extern "C" int g4c_colsum(const float *x, int32_t ld, int32_t width, int64_t n_rows, float *scratch, float *out, void *stream) {
    G4C_REQUIRE(x && scratch && out, G4C_EINVAL, "g4c_colsum: null pointer");
    g4c::DeviceGuard on_device(out);
    G4C_REQUIRE(width > 0 && n_rows >= 0 && ld >= width, G4C_EINVAL, "g4c_colsum: bad sizes width=%d ld=%d", width, ld);
    const int g = g4c_colsum_partials(n_rows);
    const long long chunk = (n_rows + g - 1) / g;
    hipStream_t s = (hipStream_t)stream;
    const unsigned cb = (unsigned)((width + 255) / 256);
    colsum_stage_kernel<<<dim3(g, cb), dim3(256), 0, s>>>(x, ld, width, n_rows, chunk > 0 ? chunk : 1, scratch, width);
    colsum_stage_kernel<<<dim3(1, cb), dim3(256), 0, s>>>(scratch, width, width, g, g, out, width);
    return g4c::check_launch("g4c_colsum");
}

extern "C" int g4c_segment_broadcast(const float *dout, int32_t dout_ld, const int32_t *off, const int32_t *perm, int32_t n_seg,
                                     int32_t width, int32_t mean, float *dsrc, int32_t dsrc_ld, void *stream) {
    G4C_REQUIRE(dout && off && dsrc, G4C_EINVAL, "g4c_segment_broadcast: null pointer");
    g4c::DeviceGuard on_device(dsrc);
    G4C_REQUIRE(width > 0 && n_seg >= 0 && dout_ld >= width && dsrc_ld >= width, G4C_EINVAL,
                "g4c_segment_broadcast: bad sizes width=%d dout_ld=%d dsrc_ld=%d", width, dout_ld, dsrc_ld);
    if (n_seg == 0) return G4C_OK;
    segment_broadcast_kernel<<<dim3((unsigned)((n_seg + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
        dout, dout_ld, off, perm, n_seg, width, mean, dsrc, dsrc_ld);
    return g4c::check_launch("g4c_segment_broadcast");
}

static inline int wg_stage2_rows(int G) { return (G + 15) / 16; }

extern "C" int32_t g4c_weight_grad_partials(int64_t n_rows) {
    const long long slabs = (n_rows + WG_SLAB - 1) / WG_SLAB;
    const long long want = slabs / 6;
    // at most one resident round: 256 CUs x 2 workgroups of 8 waves (127 registers per lane) — a second, partly filled round
    // would idle CUs
    return (int32_t)(want < 1 ? 1 : (want > 512 ? 512 : want));
}

extern "C" int g4c_weight_grad(const float *g, int32_t g_ld, const float *a, int32_t a_ld, int64_t n_rows, float *scratch,
                               float *out, int32_t with_bias, void *stream) {
    G4C_REQUIRE(g && a && scratch && out, G4C_EINVAL, "g4c_weight_grad: null pointer");
    g4c::DeviceGuard on_device(out);
    G4C_REQUIRE(n_rows >= 0 && g_ld >= WG_N && a_ld >= WG_N && g_ld % 4 == 0 && a_ld % 4 == 0 && (uintptr_t)g % 16 == 0 &&
                    (uintptr_t)a % 16 == 0,
                G4C_EINVAL, "g4c_weight_grad: 128-wide, 16-byte aligned operands expected (g_ld=%d a_ld=%d)", g_ld, a_ld);
    const int G = g4c_weight_grad_partials(n_rows);
    const long long slabs = (n_rows + WG_SLAB - 1) / WG_SLAB;
    const long long chunk = ((slabs + G - 1) / G) * WG_SLAB;
    hipStream_t st = (hipStream_t)stream;
    weight_grad_kernel<<<dim3(G), dim3(WG_THREADS), 0, st>>>(g, g_ld, a, a_ld, n_rows, chunk > 0 ? chunk : WG_SLAB, scratch, with_bias);
    // fixed-order sum of the partial tiles ([dW | db] is contiguous in every partial row)
    const int width = WG_N * WG_N + (with_bias ? WG_N : 0), ld = WG_N * WG_N + WG_N;
    float *scratch2 = scratch + (long long)G * ld;
    const int g2 = wg_stage2_rows(G);          // 16 partial tiles per first-stage workgroup: both stages are short
    const long long chunk2 = (G + g2 - 1) / g2;
    const unsigned cb = (unsigned)((width + 255) / 256);
    colsum_stage_kernel<<<dim3(g2, cb), dim3(256), 0, st>>>(scratch, ld, width, G, chunk2 > 0 ? chunk2 : 1, scratch2, ld);
    colsum_stage_kernel<<<dim3(1, cb), dim3(256), 0, st>>>(scratch2, ld, width, g2, g2, out, ld);
    return g4c::check_launch("g4c_weight_grad");
}

extern "C" int64_t g4c_weight_grad_scratch_floats(int64_t n_rows) {
    const long long ld = WG_N * WG_N + WG_N;
    const long long G = g4c_weight_grad_partials(n_rows);
    return (G + wg_stage2_rows((int)G)) * ld;
}
