// 2 x 2 register-block form of the fused bf16x6 MLP ("bx6w") for the MP layers' message launch: the same arithmetic as
// mlp_bx6_kernel (mlp_fused.hip) — gather -> [SELU on load] -> Linear/SELU chain on v_mfma_f32_32x32x16_bf16 with the exact
// three-way operand split -> LayerNorm -> activation -> store (-> per-target aggregation) — replacing MLP.forward
// (graphs4cfd/nn/blocks.py:117-144) with the torch.cat / index ops in front of it (nn/blocks.py:181,328) and, with AGG, the
// scatter(e', col, reduce) behind it (nn/blocks.py:183,330).
//
// What is different: the operand traffic per MFMA.  In mlp_bx6_kernel a wave owns a 32 x 32 output block, so every
// v_mfma_f32_32x32x16_bf16 is fed by one 1 KB weight fragment (L1) and one 1 KB activation fragment (LDS): per 32 rows and layer
// 96 KB of each.  Here a workgroup of TWO waves owns a 64-row tile and wave w the 64 columns [64 w, 64 w + 64): four 32 x 32
// accumulators (row half x column half); per 16-k step it loads two weight fragments and two activation fragments per plane (12
// loads) and issues 24 MFMAs on four independent accumulators — half the LDS reads and half the L1 weight stream per row, and no
// dependent-accumulator chain inside a step.  DESIGN.md §9.1 has the per-pipe accounting that led here.
//
// Envelope (everything else runs mlp_bx6_kernel / mlp_bx6i_kernel): exact-split mode, ONE weighted 128-wide 16-byte aligned input block
// (rows direct or through an index, optional SELU on load), 0 or 2 additive 128-wide blocks, no narrow blocks, three layers, 128-wide
// output rows without residual / output index / heads.
#include "mlp_common.h"
#include <cstdlib>
using namespace g4cm;

#ifdef G4C_BX6W_TIMING
__device__ unsigned long long g4c_bx6w_stamps[256 * 32];
extern "C" int g4c_bx6w_read_stamps(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g4c_bx6w_stamps), sizeof(unsigned long long) * n);
}
#define BW_STAMP(k) do { if (pair < 256 && tid == 0) g4c_bx6w_stamps[pair * 32 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define BW_STAMP(k) do {} while (0)
#endif

namespace {

constexpr int ROWS = 64;
constexpr int PLW = ROWS * HB;               // bf16 elements of one operand plane of the tile [64][136]
constexpr int TILEW = 3 * PLW;               // three planes (the fp32 final rows [64][132] alias them: 33 792 B <= 52 224 B)

// One 128-k block: acc[r][c] += W(block, column half c) x planes(row half r).  Weight fragments stream through a two-step ring
// (slot s & 1 refilled with step s + 2 right after its last use; the last two steps refill from `wnext`).
struct RingW { bf16x8 w[2][2][3]; };      // [slot][column half][plane]

__device__ __forceinline__ void m_block_w(const __bf16 *pa, RingW &g, __amdgpu_buffer_rsrc_t rs, const unsigned (&lo_b)[2], unsigned wcur,
                                          unsigned wnext, f32x16 (&acc)[2][2]) {
    bf16x8 cur[2][3], nx[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) cur[r][pl] = *reinterpret_cast<const bf16x8 *>(pa + r * 32 * HB + pl * PLW);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s < 7) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) nx[r][pl] = *reinterpret_cast<const bf16x8 *>(pa + r * 32 * HB + pl * PLW + 16 * (s + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
        // six products per (row half, column half), small terms first as mlp_bx6_kernel; the four accumulators in rotation
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            constexpr int WP[6] = {0, 2, 1, 0, 1, 0}, XP[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g.w[s & 1][c][WP[k]], cur[r][XP[k]], acc[r][c], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const unsigned so = (s < 6) ? wcur + 2u * (s + 2) * STEP6 : wnext + 2u * (s - 6) * STEP6;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                g.w[s & 1][c][0] = ldw(rs, lo_b[c], so);
                g.w[s & 1][c][1] = ldw(rs, lo_b[c] + 1024u, so);
                g.w[s & 1][c][2] = ldw(rs, lo_b[c] + 2048u, so);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s < 7) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) cur[r][pl] = nx[r][pl];
        }
    }
}

template <bool AGG>
__global__ __launch_bounds__(128) void mlp_bx6w_kernel(const Params p) {
    __shared__ __attribute__((aligned(16))) __bf16 sB[TILEW];           // 52 224 B
    __shared__ int sIdx[3][ROWS];                                        // [weighted block, additive 0, additive 1][row]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int prow = (lane >> 3) + 8 * wave, c4 = (lane & 7) * 4;        // park layout: rows prow + 16 k, columns 32 q + c4 .. + 3

    // 64-row tile of this workgroup = two consecutive 32-row tiles of the launch's tile table (XCD-aware order)
    const int n_pairs = (p.n_tiles + 1) >> 1;
    int pair;
    {
        const int b = blockIdx.x, q = n_pairs >> 3, r = n_pairs & 7, x = b & 7, j = b >> 3;
        pair = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    // row half t = 32-row tile 2 pair + t (with aggregation: a tile of whole segments, up to 32 rows; the row halves keep the
    // 32-row slots of the planes, rows past a half's end are clamped copies that are never stored)
    int row0[2], nrow[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int tile = 2 * pair + t;
        if (tile >= p.n_tiles) { row0[t] = 0; nrow[t] = 0; }
        else if (AGG) { row0[t] = p.tile_rows[tile]; nrow[t] = p.tile_rows[tile + 1] - row0[t]; }
        else { row0[t] = (int)p.row_base + tile * 32; const int lim = (int)p.M - row0[t]; nrow[t] = lim < 32 ? lim : 32; }
    }
    if (nrow[1] == 0) row0[1] = row0[0];
    BW_STAMP(0);

    // ---- gather indices (192 entries, 128 threads)
    for (int e = tid; e < 3 * ROWS; e += 128) {
        const int k = e >> 6, rr = e & 63, t = rr >> 5, r = rr & 31;
        const int nn = nrow[t] > 0 ? nrow[t] : nrow[0];
        const int gr = row0[t] + (r < nn ? r : nn - 1);
        const int *ix = (k == 0) ? p.src[0].idx : ((k - 1 < p.n_add) ? p.add[k - 1].idx : nullptr);
        sIdx[k][rr] = ix ? ix[gr] : gr;
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, 0x7fffffff, 0x00020000);
    unsigned lo_b[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) lo_b[c] = 2u * (unsigned)((2 * wave + c) * 8 * STEP6 + lane * 8);
    RingW ring;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            ring.w[s][c][0] = ldw(rs, lo_b[c], 2u * s * STEP6);
            ring.w[s][c][1] = ldw(rs, lo_b[c] + 1024u, 2u * s * STEP6);
            ring.w[s][c][2] = ldw(rs, lo_b[c] + 2048u, 2u * s * STEP6);
        }
    __syncthreads();
    BW_STAMP(1);

    // ---- input rows (park layout: 4 row groups of 16), start values of the four accumulators (bias + additive rows)
    const bool pact = p.src[0].pre_act != 0;
    {
        f32x4 xp[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float *rp = p.src[0].ptr + (long long)sIdx[0][prow + 16 * k] * p.src[0].ld + p.src[0].col0 + c4;
#pragma unroll
            for (int q = 0; q < 4; ++q) xp[k][q] = *reinterpret_cast<const f32x4 *>(rp + q * KC);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            __bf16 *d = sB + (prow + 16 * k) * HB + c4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = xp[k][q];
                if (pact) v = selu4(v);
                bf16x4 vh, vm, vl;
                split3x4<3>(v, vh, vm, vl);
                *reinterpret_cast<bf16x4 *>(d + q * KC) = vh;
                *reinterpret_cast<bf16x4 *>(d + PLW + q * KC) = vm;
                *reinterpret_cast<bf16x4 *>(d + 2 * PLW + q * KC) = vl;
            }
        }
    }
    BW_STAMP(2);
    f32x16 acc[2][2];
    int fb[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) fb[c] = 64 * wave + 32 * c + 4 * h;       // this lane's features of column half c: fb + 8 gq + e
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(p.b + fb[c] + 8 * gq);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[r][c][4 * gq + e] = b4[e];
            }
            if (p.n_add == 2) {
                const float *p0 = p.add[0].ptr + (long long)sIdx[1][i + 32 * r] * p.add[0].ld + fb[c];
                const float *p1 = p.add[1].ptr + (long long)sIdx[2][i + 32 * r] * p.add[1].ld + fb[c];
                f32x4 a0[4], a1[4];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) { a0[gq] = *reinterpret_cast<const f32x4 *>(p0 + 8 * gq); a1[gq] = *reinterpret_cast<const f32x4 *>(p1 + 8 * gq); }
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[r][c][4 * gq + e] = (acc[r][c][4 * gq + e] + a0[gq][e]) + a1[gq][e];
            }
        }
    BW_STAMP(3);
    __syncthreads();
    BW_STAMP(4);

    const __bf16 *pa = sB + i * HB + 8 * h;
    constexpr unsigned WB = 2u * BLOCK6;
    float *sH = reinterpret_cast<float *>(sB);
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        m_block_w(pa, ring, rs, lo_b, WB * (unsigned)l, WB * (unsigned)(l < 2 ? l + 1 : l), acc);
        BW_STAMP(5 + 4 * l);
        __syncthreads();                                    // everybody has read the planes
        BW_STAMP(6 + 4 * l);
        if (l < 2) {
            // hidden layer: SELU (the bias was the start value), exact split, planes; next start values = next bias
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    __bf16 *d = sB + (i + 32 * r) * HB + fb[c];
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        f32x4 x;
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] = acc[r][c][4 * gq + e];
                        bf16x4 vh, vm, vl;
                        split3x4<3>(selu4(x), vh, vm, vl);
                        *reinterpret_cast<bf16x4 *>(d + 8 * gq) = vh;
                        *reinterpret_cast<bf16x4 *>(d + PLW + 8 * gq) = vm;
                        *reinterpret_cast<bf16x4 *>(d + 2 * PLW + 8 * gq) = vl;
                        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(p.b + (l + 1) * NP + fb[c] + 8 * gq);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[r][c][4 * gq + e] = b4[e];
                    }
                }
        } else {
            // last layer: fp32 rows (aliasing the planes everybody finished reading)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        f32x4 x;
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] = acc[r][c][4 * gq + e];
                        *reinterpret_cast<f32x4 *>(sH + (i + 32 * r) * HS + fb[c] + 8 * gq) = x;
                    }
        }
        BW_STAMP(7 + 4 * l);
        __syncthreads();
        BW_STAMP(8 + 4 * l);
    }
    // ---- LayerNorm / activation / stores: wave w owns rows [32 w, 32 w + 32) = row half w, in four groups of 8
    // (lane = part * 8 + row_local, each part = 16 consecutive columns)
    {
        const int t = wave;
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            const int rloc = lane & 7, part = lane >> 3, myrow = 8 * k + rloc, cbk = part * 16;
            float *rowp = sH + (32 * t + myrow) * HS + cbk;
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
                    const f32x4 g4 = *reinterpret_cast<const f32x4 *>(p.gamma + cbk + c), b4 = *reinterpret_cast<const f32x4 *>(p.beta + cbk + c);
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
            if (AGG) {
#pragma unroll
                for (int c = 0; c < 16; c += 4) {
                    f32x4 v;
                    v[0] = x[c]; v[1] = x[c + 1]; v[2] = x[c + 2]; v[3] = x[c + 3];
                    *reinterpret_cast<f32x4 *>(rowp + c) = v;
                }
            }
            if (p.out && myrow < nrow[t]) {
                float *op = p.out + (long long)(row0[t] + myrow) * p.out_ld + cbk;
#pragma unroll
                for (int c = 0; c < 16; c += 4) {
                    f32x4 v;
                    v[0] = x[c]; v[1] = x[c + 1]; v[2] = x[c + 2]; v[3] = x[c + 3];
                    *reinterpret_cast<f32x4 *>(op + c) = v;
                }
            }
        }
    }
    BW_STAMP(20);
    if (AGG) {
        // aggregation of the targets whose messages the tile holds (rows in CSR order): same summation order and the same mean
        // formula as segment_reduce_kernel, so the result is bit-identical to the separate launch.  Wave t reduces row half t
        // (the rows it normalised itself: no barrier needed), lane -> 2 of the 128 columns
        const int t = wave;
        if (nrow[t] > 0) {
            const int tile = 2 * pair + t;
            const int s0 = p.tile_seg[tile], s1 = p.tile_seg[tile + 1];
            const float *base = sH + (32 * t) * HS + 2 * lane;
            for (int sg = s0; sg < s1; ++sg) {
                const int bb = p.seg_off[sg] - row0[t], ee = p.seg_off[sg + 1] - row0[t];
                float a0 = 0.f, a1 = 0.f;
                for (int r = bb; r < ee; ++r) { const float2 v = *reinterpret_cast<const float2 *>(base + r * HS); a0 += v.x; a1 += v.y; }
                if (p.agg_mean) { const float cnt = (float)((ee - bb) > 1 ? (ee - bb) : 1); a0 /= cnt; a1 /= cnt; }
                float2 o; o.x = a0; o.y = a1;
                *reinterpret_cast<float2 *>(p.agg + (long long)sg * p.agg_ld + 2 * lane) = o;
            }
        }
    }
}

}  // namespace

namespace g4cm {

// 0 off (default while it is being measured; environment G4C_BX6W), 1 launches of at least G4C_BX6W_MIN_ROWS rows, 2 every eligible launch
static int g_bx6w = -1;
int bx6w_enable(int on) {
    if (g_bx6w < 0) g_bx6w = getenv("G4C_BX6W") ? atoi(getenv("G4C_BX6W")) : 0;
    const int old = g_bx6w;
    if (on >= 0) g_bx6w = on > 2 ? 2 : on;
    return old;
}

bool bx6w_eligible(const Params &p, bool round1, bool agg, bool save, long long row_count) {
    static const long long min_rows = getenv("G4C_BX6W_MIN_ROWS") ? atoll(getenv("G4C_BX6W_MIN_ROWS")) : 200000;
    const int mode = bx6w_enable(-1);
    if (!mode || round1 || save) return false;
    if (mode == 1 && row_count < min_rows) return false;
    if (p.n_src != 1 || p.n_nar != 0 || (p.n_add != 0 && p.n_add != 2) || p.n_heads) return false;
    if (p.n_layers != 3 || p.n_out != NP || p.resid || p.out_idx || p.out_bf16) return false;
    const Src &s = p.src[0];
    if (s.width != NP || !s.vec || s.seg_off || s.bf16) return false;
    for (int a = 0; a < p.n_add; ++a)
        if (p.add[a].width != NP || (p.add[a].ld & 3) || ((uintptr_t)p.add[a].ptr & 15)) return false;
    if (p.out && ((p.out_ld & 3) || ((uintptr_t)p.out & 15))) return false;
    if (p.gamma && (((uintptr_t)p.gamma & 15) || ((uintptr_t)p.beta & 15))) return false;
    if (((uintptr_t)p.b & 15) || (p.agg && ((p.agg_ld & 1) || ((uintptr_t)p.agg & 7)))) return false;
    if (p.M >= (1LL << 31)) return false;
    (void)agg;
    return true;
}

int bx6w_launch(const Params &p, bool agg, hipStream_t st) {
    const int n_pairs = (p.n_tiles + 1) / 2;
    if (n_pairs == 0) return G4C_OK;
    if (agg) mlp_bx6w_kernel<true><<<dim3(n_pairs), dim3(128), 0, st>>>(p);
    else mlp_bx6w_kernel<false><<<dim3(n_pairs), dim3(128), 0, st>>>(p);
    return g4c::check_launch("g4c_mlp_forward (bx6w)");
}

}  // namespace g4cm

extern "C" int g4c_mlp_bx6w_enable(int on) { return g4cm::bx6w_enable(on); }
