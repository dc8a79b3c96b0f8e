// Fused gather -> concat -> MLP (fp32 MFMA) -> LayerNorm -> activation -> residual -> store.
//
// Replaces, in ONE launch, what the reference issues as ~10-25 torch kernels per MLP call:
// `torch.cat((e, v[row], v[col]))` + nn.Linear/nn.SELU chain + nn.LayerNorm (MLP.forward,
// graphs4cfd/nn/blocks.py:117-144) + the F.selu / torch.tanh applied by the caller
// (nn/mus_gnn.py:178-212, nn/blocks.py:233,288) + the residual time step (nn/mus_gnn.py:218).
// The concatenated input is never materialised: each source is gathered row-wise into LDS.
//
// Design (CDNA4, wave64): ONE WAVE owns RT*32 rows (RT = 2 for large launches, 1 otherwise) through
// every layer, so there is no workgroup barrier anywhere and waves never wait for each other.
//   * every layer is computed 128 wide (narrower layers are zero-padded when packed) on
//     v_mfma_f32_32x32x2_f32: RT x 4 accumulators of 16 VGPRs; per 4 k: RT ds_read_b64 of A,
//     two 16-byte global loads of B, RT*8 MFMAs (1024 cycles at RT = 2);
//   * weights (B) stream L2 -> registers through an 8-deep ring (one ring slot per 4 k, refilled
//     for the next 32-k chunk right after use): no LDS staging of weights, ~3.5 us of prefetch;
//   * the gathered input chunk [RT*32 x 32] goes global -> registers -> wave-private LDS, one chunk
//     ahead; hidden activations live in wave-private LDS ([RT*32][132] fp32, aliasing the input
//     buffers) between layers;
//   * LayerNorm / activation / residual: one lane per (row, half), one cross-half exchange, then
//     whole-row 16-byte stores.
// 64-thread workgroups, 34 KiB LDS at RT = 2 -> 4 waves per CU (one per SIMD, 512 VGPRs each).
//
// MFMA-bound in fp32: 2*K*128 FLOP per row per layer against ~(K_in + N_out)*4 bytes per row.
#include "mlp_common.h"
#include <atomic>
#include <cstdlib>
#include <type_traits>
using namespace g4cm;

// Timing-only ablations, compile-time so that the production kernel has no extra control flow
// (a runtime switch between two mma_chunk instantiations makes hipcc reconcile the weight-ring
// registers at the join, i.e. wait for the in-flight loads at every chunk):
//   -DG4C_ABLATE=1 no ring refill, 2 no input staging, 4 no epilogues / final pass (bits can be OR-ed)
#ifndef G4C_ABLATE
#define G4C_ABLATE 0
#endif

// -DG4C_TIMING: per-phase s_memtime stamps of the first 4096 tiles into g4c_dbg_stamps (debug builds only)
#ifdef G4C_TIMING
__device__ unsigned long long g4c_dbg_stamps[4096 * 16];
#define G4C_STAMP(k) do { if (tile < 4096 && lane == 0) g4c_dbg_stamps[tile * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
extern "C" int g4c_debug_read_stamps(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g4c_dbg_stamps), sizeof(unsigned long long) * n);
}
#define G4C_STAMPW(k) do { if (tile < 4096 && tid == 0) g4c_dbg_stamps[tile * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define G4C_STAMP(k) do {} while (0)
#define G4C_STAMPW(k) do {} while (0)
#endif

namespace {


// offset of 4-k step U inside a chunk of the weight stream: steps are stored in pairs (see pack_layer_kernel)
__host__ __device__ constexpr int step_off(int U) { return (U >> 1) * 1024 + (U & 1) * 2; }


// ---- gathered input chunk: lane (r = lane>>3 [+8q], c4 = lane&7) loads 4 consecutive columns.
// VEC: every source is 16-byte addressable (width, ld, col0 multiples of 4, aligned base) -> one
// unconditional 16-byte load per row; the generic path does 4 clamped scalar loads.
template <bool VEC>
__device__ __forceinline__ f32x4 ldx1(const Src &s, int srow, int c) {
    const float *base = s.ptr + (long long)srow * s.ld + s.col0;
    f32x4 t;
    if (VEC) {
        // columns beyond the source width (padding up to 32) are zero-filled by select; the load
        // itself is unconditional from a clamped column (no path-dependent load count: hipcc can
        // then wait for these loads with an exact vmcnt instead of draining the weight ring)
        // (the zero fill itself happens in finish_x, after the MFMAs: touching the loaded value here
        // would make the wave wait for the gather before it starts computing)
        t = *reinterpret_cast<const f32x4 *>(base + (c < s.width ? c : 0));
    } else {
        const int w1 = s.width - 1;
        t[0] = base[c + 0 < w1 ? c + 0 : w1];
        t[1] = base[c + 1 < w1 ? c + 1 : w1];
        t[2] = base[c + 2 < w1 ? c + 2 : w1];
        t[3] = base[c + 3 < w1 ? c + 3 : w1];
    }
    return t;
}

// ======================================================================================================
// The fp32-MFMA kernel (precision "fp32", and the launches the bf16x6 kernel cannot take): NW = 4 waves share
// one 32-row tile, each computing NCT = 4/NW of the four 32-column tiles.  (Round 1 also carried single-wave 64- / 32-row
// tiles, a 2-wave split, a 64-row split and a small-launch variant; the 4-wave split was the fastest or within 2 % of the
// fastest at every launch size, and the others were removed.)
// The waves of a tile share the gathered input chunk and the hidden activations through LDS:
// one barrier per layer-0 chunk, two per layer.  Weights still stream L2 -> registers (each wave only its
// own column tiles), accumulators stay in registers.
template <int NCT> struct BVec;
template <> struct BVec<2> { typedef f32x4 type; };
template <> struct BVec<1> { typedef float2 type; };

__device__ __forceinline__ void load_bn(float2 &b, const float *wstep, unsigned lane_off) {
    b = *reinterpret_cast<const float2 *>(wstep + lane_off);
}
__device__ __forceinline__ void load_bn(f32x4 &b, const float *wstep, unsigned lane_off) {
    const float2 c0 = *reinterpret_cast<const float2 *>(wstep + lane_off);
    const float2 c1 = *reinterpret_cast<const float2 *>(wstep + lane_off + 256);
    b[0] = c0.x; b[1] = c0.y; b[2] = c1.x; b[3] = c1.y;
}

template <int NCT> struct AccN { f32x16 t[NCT]; };
template <int NCT> struct RingN { typename BVec<NCT>::type s0, s1, s2, s3, s4, s5, s6, s7; };

__device__ __forceinline__ float bget(const f32x4 &b, int k) { return b[k]; }
__device__ __forceinline__ float bget(const float2 &b, int k) { return k ? b.y : b.x; }

// (tried on this kernel and measured neutral or worse in round 1, now constants: s_setprio around the MFMAs, LayerNorm parameters
// from global memory at 8 workgroups per CU ("slim"), 7 waves per SIMD)
constexpr int G4C_SPLIT_PRIO = 0;
constexpr int G4C_SPLIT_SLIM = 0;
template <int NCT>
__device__ __forceinline__ void mma_chunk_n(const float *pa, RingN<NCT> &g, const float *wnext, unsigned lo, AccN<NCT> &acc) {
    float2 a = *reinterpret_cast<const float2 *>(pa);
#define G4C_STEP(U, SLOT)                                                                  \
    {                                                                                      \
        const float2 an = (G4C_ABLATE & 64) ? a : *reinterpret_cast<const float2 *>(pa + (((U) + 1) & 7) * 4); \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        if (G4C_SPLIT_PRIO) __builtin_amdgcn_s_setprio(1);                                 \
        _Pragma("unroll") for (int c = 0; c < NCT; ++c) {                                  \
            acc.t[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bget(g.SLOT, 2 * c), acc.t[c], 0, 0, 0);     \
        }                                                                                  \
        _Pragma("unroll") for (int c = 0; c < NCT; ++c) {                                  \
            acc.t[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bget(g.SLOT, 2 * c + 1), acc.t[c], 0, 0, 0); \
        }                                                                                  \
        if (G4C_SPLIT_PRIO) __builtin_amdgcn_s_setprio(0);                                 \
        if (!(G4C_ABLATE & 32) && !((G4C_ABLATE & 256) && ((U) & 1))) load_bn(g.SLOT, wnext + step_off(U), lo); \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        a = an;                                                                            \
    }
    G4C_STEP(0, s0) G4C_STEP(1, s1) G4C_STEP(2, s2) G4C_STEP(3, s3)
    G4C_STEP(4, s4) G4C_STEP(5, s5) G4C_STEP(6, s6) G4C_STEP(7, s7)
#undef G4C_STEP
}

// LayerNorm / activation / store of a finished 32-row tile held in sH; rows split over the NW waves of the workgroup.
// Shared by the column-split kernels.  Needs: all waves' last-layer columns visible in sH (barrier done by the caller).
template <int NW, int ROWS = 32>
__device__ __forceinline__ void split_finish(const Params &p, float *sH, const float *sGB, int wave, int lane, long long row0,
                                             long long mlim = -1) {
    if (mlim < 0) mlim = p.M;          // rows >= mlim are not stored (mlim < p.M: tiles of whole segments)
    const int i = lane & 31, h = lane >> 5;
    // ---------------------------------------------------------------- LayerNorm / activation: rows split over the waves
    // wave w owns rows [w*RPW, (w+1)*RPW); lane = part * RPW + row_local, each part = NC consecutive columns
    constexpr int RPW = ROWS / NW, PARTS = 64 / RPW, NC = NP / PARTS;
    const int n_out = p.n_out;
    const int rloc = lane % RPW, part = lane / RPW;
    const int myrow = wave * RPW + rloc;
    const int cb = part * NC;
    if ((p.gamma || p.act) && !(G4C_ABLATE & 4)) {
        float *rowp = sH + myrow * HS + cb;
        const float inv_n = 1.0f / (float)n_out;
        float x[NC];
#pragma unroll
        for (int c = 0; c < NC; c += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(rowp + c);
            x[c] = t[0]; x[c + 1] = t[1]; x[c + 2] = t[2]; x[c + 3] = t[3];
        }
        if (p.gamma) {
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) sum += (cb + c < n_out) ? x[c] : 0.f;
#pragma unroll
            for (int o = RPW; o < 64; o <<= 1) sum += __shfl_xor(sum, o);
            const float mean = sum * inv_n;
            float var = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) { const float d = x[c] - mean; var += (cb + c < n_out) ? d * d : 0.f; }
#pragma unroll
            for (int o = RPW; o < 64; o <<= 1) var += __shfl_xor(var, o);
            const float rstd = rsqrtf(var * inv_n + p.eps);
#pragma unroll
            for (int c = 0; c < NC; c += 4) {
                f32x4 g4, b4;
                if (G4C_SPLIT_SLIM) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int cc = (cb + c + u < n_out) ? cb + c + u : 0;
                        g4[u] = p.gamma[cc]; b4[u] = p.beta[cc];
                    }
                } else {
                    g4 = *reinterpret_cast<const f32x4 *>(sGB + cb + c);
                    b4 = *reinterpret_cast<const f32x4 *>(sGB + NP + cb + c);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) x[c + u] = fmaf((x[c + u] - mean) * rstd, g4[u], b4[u]);
            }
        }
        if (p.act == G4C_ACT_SELU) {
#pragma unroll
            for (int c = 0; c < NC; ++c) x[c] = g4c::selu_f(x[c]);
        } else if (p.act == G4C_ACT_TANH) {
#pragma unroll
            for (int c = 0; c < NC; ++c) x[c] = g4c::tanh_f(x[c]);
        }
#pragma unroll
        for (int c = 0; c < NC; c += 4) {
            f32x4 t;
            t[0] = x[c]; t[1] = x[c + 1]; t[2] = x[c + 2]; t[3] = x[c + 3];
            *reinterpret_cast<f32x4 *>(rowp + c) = t;
        }
    }
    // (each wave reads back only the rows it normalised itself: no barrier needed)

    // ---------------------------------------------------------------- store this wave's rows
    if (p.out == nullptr) return;          // (only the aggregate of the rows is wanted: g4c_mlp_forward_bx6_agg with out == NULL)
    const bool fast = (n_out == NP) && ((p.out_ld & 3) == 0) && (((uintptr_t)p.out & 15) == 0) && (p.resid == nullptr);
    if (p.out_bf16) {
        // rows kept in bf16 (the rounded-bf16 mode's message tensors: their consumer rounds them to bf16 anyway): 8 bytes per lane
        __bf16 *o16 = reinterpret_cast<__bf16 *>(p.out);
#pragma unroll
        for (int r = h; r < RPW; r += 2) {
            const int row = wave * RPW + r;
            const long long grow = row0 + row;
            if (grow < mlim) {
                f32x4 t = *reinterpret_cast<const f32x4 *>(sH + row * HS + 4 * i);
                if (p.out_bf16 == 2) t = selu4(t);          // (G4C_DTYPE_BF16_SELU: the reader's pending activation, applied before the one rounding)
                bf16x4 b;
#pragma unroll
                for (int u = 0; u < 4; ++u) b[u] = (__bf16)t[u];
                *reinterpret_cast<bf16x4 *>(o16 + grow * p.out_ld + 4 * i) = b;
            }
        }
    } else if (fast) {
#pragma unroll
        for (int r = h; r < RPW; r += 2) {
            const int row = wave * RPW + r;
            const long long grow = row0 + row;
            if (grow < mlim) {
                const long long orow = p.out_idx ? p.out_idx[grow] : grow;
                const f32x4 t = *reinterpret_cast<const f32x4 *>(sH + row * HS + 4 * i);
                *reinterpret_cast<f32x4 *>(p.out + orow * p.out_ld + 4 * i) = t;
            }
        }
    } else {
        for (int e = lane; e < RPW * n_out; e += 64) {
            const int r = e / n_out, c = e - r * n_out;
            const int row = wave * RPW + r;
            const long long grow = row0 + row;
            if (grow < mlim) {
                const long long orow = p.out_idx ? p.out_idx[grow] : grow;
                float y = sH[row * HS + c];
                if (p.resid) y += p.resid[grow * p.resid_ld + p.resid_col0 + c];
                p.out[orow * p.out_ld + c] = y;
            }
        }
    }
}

constexpr int G4C_SPLIT_B4 = 1;        // 16-byte weight ring slots (two steps per load)
// ring of four 16-byte slots, each holding this lane's B operands of TWO consecutive steps (NCT == 1)
struct Ring4 { f32x4 p0, p1, p2, p3; };
__device__ __forceinline__ void ring4_fill(Ring4 &g, const float *w, unsigned lo4) {
    g.p0 = *reinterpret_cast<const f32x4 *>(w + 0 * 1024 + lo4); g.p1 = *reinterpret_cast<const f32x4 *>(w + 1 * 1024 + lo4);
    g.p2 = *reinterpret_cast<const f32x4 *>(w + 2 * 1024 + lo4); g.p3 = *reinterpret_cast<const f32x4 *>(w + 3 * 1024 + lo4);
}
__device__ __forceinline__ void mma_chunk_4(const float *pa, Ring4 &g, const float *wnext, unsigned lo4, AccN<1> &acc) {
    float2 a = *reinterpret_cast<const float2 *>(pa);
#define G4C_PAIR(V, SLOT)                                                                             \
    {                                                                                                  \
        const float2 a1 = *reinterpret_cast<const float2 *>(pa + (2 * (V) + 1) * 4);                   \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        acc.t[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, g.SLOT[0], acc.t[0], 0, 0, 0);            \
        acc.t[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, g.SLOT[1], acc.t[0], 0, 0, 0);            \
        const float2 a2 = *reinterpret_cast<const float2 *>(pa + ((2 * (V) + 2) & 7) * 4);             \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        acc.t[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, g.SLOT[2], acc.t[0], 0, 0, 0);           \
        acc.t[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, g.SLOT[3], acc.t[0], 0, 0, 0);           \
        g.SLOT = *reinterpret_cast<const f32x4 *>(wnext + (V) * 1024 + lo4);                           \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        a = a2;                                                                                        \
    }
    G4C_PAIR(0, p0) G4C_PAIR(1, p1) G4C_PAIR(2, p2) G4C_PAIR(3, p3)
#undef G4C_PAIR
}

constexpr int G4C_SPLIT_MINW = 1;
template <int NW, bool VEC>
__global__ __launch_bounds__(64 * NW, G4C_SPLIT_MINW) void mlp_split_kernel(const Params p) {
    constexpr int ROWS = 32, NCT = 4 / NW, NPIECE = 4 / NW;   // pieces (8 rows x 32 cols) of a chunk gathered per wave
    // LDS budget: 8 workgroups of NW = 4 waves per CU (= the 32-wave limit) need <= 20 KiB each, so the LayerNorm
    // parameters are read from global memory (L1/L2 hits) instead of being staged
    constexpr int GB_ROWS = G4C_SPLIT_SLIM ? 0 : 2;
    __shared__ __attribute__((aligned(16))) float lds[ROWS * HS + 2 * G4C_MAX_SRC * ROWS + (G4C_MAX_LAYERS + GB_ROWS) * NP];
    float *sH = lds;
    float *sX0 = lds;
    float *sX1 = lds + ROWS * XS;
    int *sRow = reinterpret_cast<int *>(lds + ROWS * HS);
    int *sRowAdd = sRow + G4C_MAX_SRC * ROWS;
    float *sBias = lds + ROWS * HS + 2 * G4C_MAX_SRC * ROWS;
    float *sGB = sBias + G4C_MAX_LAYERS * NP;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int ct0 = wave * NCT;

    int tile;
    {
        const int b = blockIdx.x, nt = p.n_tiles;
        const int q = nt >> 3, r = nt & 7, x = b & 7, j = b >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const long long row0 = p.row_base + (long long)tile * ROWS;
    G4C_STAMPW(0);

    // row indices of every input block: one (block, row) per thread, so all index loads are in flight together
    for (int e = tid; e < 2 * G4C_MAX_SRC * ROWS; e += 64 * NW) {
        const int slot = e / ROWS, r = e % ROWS;
        long long gr = row0 + r;
        if (gr >= p.M) gr = p.M - 1;
        const int *ix = nullptr;
        bool used;
        if (slot < G4C_MAX_SRC) { used = slot < p.n_src; if (used) ix = p.src[slot].idx; }
        else { used = slot - G4C_MAX_SRC < p.n_add; if (used) ix = p.add[slot - G4C_MAX_SRC].idx; }
        if (used) sRow[e] = ix ? ix[gr] : (int)gr;      // sRowAdd == sRow + G4C_MAX_SRC * ROWS
    }
    for (int e = tid; e < p.n_layers * NP; e += 64 * NW) sBias[e] = p.b[e];
    if (p.gamma && !G4C_SPLIT_SLIM) {
        for (int e = tid; e < NP; e += 64 * NW) {
            const int ee = e < p.n_out ? e : 0;
            sGB[e] = p.gamma[ee];
            sGB[NP + e] = p.beta[ee];
        }
    }
    __syncthreads();
    G4C_STAMPW(1);

    AccN<NCT> acc;
    constexpr bool B4 = G4C_SPLIT_B4 && NCT == 1;
    RingN<NCT> ring;
    Ring4 ring4;
    const unsigned lo4 = (unsigned)(ct0 * 256 + lane * 4);
    const float *w = p.w;
    const unsigned lo = (unsigned)(ct0 * 256 + lane * 4);
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc.t[c][q] = 0.f;
    // pre-multiplied node-side terms of the first layer: one source at a time, its 16*NCT loads issued together
    // (16 temporaries keep the kernel at 7 waves per SIMD; batching both sources at once costs occupancy and is slower)
    for (int a = 0; a < ((G4C_ABLATE & 8) ? 0 : p.n_add); ++a) {
        float t[16][NCT];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
            const float *pr = p.add[a].ptr + (long long)sRowAdd[a * ROWS + row] * p.add[a].ld;
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const int col = (ct0 + c) * 32 + i;
                t[q][c] = pr[col < p.add[a].width ? col : 0];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc.t[c][q] += ((ct0 + c) * 32 + i < p.add[a].width) ? t[q][c] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
    }
    G4C_STAMPW(2);
    if constexpr (B4) {
        ring4_fill(ring4, w, lo4);
    } else {
        load_bn(ring.s0, w + step_off(0), lo); load_bn(ring.s1, w + step_off(1), lo);
        load_bn(ring.s2, w + step_off(2), lo); load_bn(ring.s3, w + step_off(3), lo);
        load_bn(ring.s4, w + step_off(4), lo); load_bn(ring.s5, w + step_off(5), lo);
        load_bn(ring.s6, w + step_off(6), lo); load_bn(ring.s7, w + step_off(7), lo);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---------------------------------------------------------------- layer 0: shared gather, one barrier per chunk
    {
        const int c4 = (lane & 7) * 4;
        f32x4 xp[NPIECE];
        const float *rp[NPIECE];
        int s = 0, k0 = 0;
        int cur_width = p.src[0].width, cur_wpad = p.src[0].wpad, cur_act = p.src[0].pre_act;
        auto set_rows = [&](int sidx) {
#pragma unroll
            for (int q = 0; q < NPIECE; ++q)
                rp[q] = p.src[sidx].ptr + (long long)sRow[sidx * ROWS + (lane >> 3) + 8 * (wave + NW * q)] * p.src[sidx].ld + p.src[sidx].col0;
        };
        auto gather = [&](int kk) {
            const int c = kk + c4;
#pragma unroll
            for (int q = 0; q < NPIECE; ++q) {
                if (VEC) {
                    xp[q] = *reinterpret_cast<const f32x4 *>(rp[q] + (c < cur_width ? c : 0));
                } else {
                    const int w1 = cur_width - 1;
                    xp[q][0] = rp[q][c + 0 < w1 ? c + 0 : w1]; xp[q][1] = rp[q][c + 1 < w1 ? c + 1 : w1];
                    xp[q][2] = rp[q][c + 2 < w1 ? c + 2 : w1]; xp[q][3] = rp[q][c + 3 < w1 ? c + 3 : w1];
                }
            }
        };
        auto park = [&](float *dst, int kk) {
            const int c = kk + c4;
#pragma unroll
            for (int q = 0; q < NPIECE; ++q) {
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = (c + e < cur_width) ? xp[q][e] : 0.f;
                if (cur_act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = g4c::selu_f(t[e]);
                }
                float *d = dst + ((lane >> 3) + 8 * (wave + NW * q)) * XS + c4;
                *reinterpret_cast<float2 *>(d) = make_float2(t[0], t[1]);
                *reinterpret_cast<float2 *>(d + 2) = make_float2(t[2], t[3]);
            }
        };
        set_rows(0);
        gather(0);
        park(sX0, 0);
        __syncthreads();
        G4C_STAMPW(3);
        for (int c = 0; c < p.chunks0; ++c) {
            int nk0 = k0 + KC;
            if (nk0 >= cur_wpad) {
                if (s + 1 < p.n_src) {
                    ++s; nk0 = 0;
                    cur_width = p.src[s].width; cur_wpad = p.src[s].wpad; cur_act = p.src[s].pre_act;
                    set_rows(s);
                } else {
                    nk0 = k0;
                }
            }
            if (!(G4C_ABLATE & 2)) gather(nk0);
            __builtin_amdgcn_sched_barrier(0);
            w += CHUNK_FLOATS;
            if constexpr (B4) mma_chunk_4(((c & 1) ? sX1 : sX0) + i * XS + 2 * h, ring4, w, lo4, acc);
            else mma_chunk_n<NCT>(((c & 1) ? sX1 : sX0) + i * XS + 2 * h, ring, w, lo, acc);
            if (!(G4C_ABLATE & 2)) park((c & 1) ? sX0 : sX1, nk0);
            k0 = nk0;
            if (!(G4C_ABLATE & 16)) __syncthreads();
        }
    }

    G4C_STAMPW(4);
    // ---------------------------------------------------------------- layers 1..L-1
    for (int l = 0;; ++l) {
        const bool last = (l == p.n_layers - 1);
        {   // this wave's column tiles of the layer output -> shared hidden buffer
            float *base = sH + (4 * h) * HS + i;
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const float bv = sBias[l * NP + (ct0 + c) * 32 + i];
#pragma unroll
                for (int q0 = 0; q0 < 16; q0 += 4) {   // 4 at a time: more temporaries cost a wave of occupancy
                    float x[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[q] = acc.t[c][q0 + q] + bv;
                    if (!last && !(G4C_ABLATE & 4)) {   // uniform branch per group, not per element
#pragma unroll
                        for (int q = 0; q < 4; ++q) x[q] = g4c::selu_f(x[q]);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) base[(q + 2 * q0) * HS + (ct0 + c) * 32] = x[q];
                }
            }
        }
        __syncthreads();
        G4C_STAMPW(5 + 2 * l);
        if (last) break;
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc.t[c][q] = 0.f;
#pragma unroll 1
        for (int k0 = 0; k0 < NP; k0 += KC) {
            w += CHUNK_FLOATS;
            if constexpr (B4) mma_chunk_4(sH + i * HS + k0 + 2 * h, ring4, w, lo4, acc);
            else mma_chunk_n<NCT>(sH + i * HS + k0 + 2 * h, ring, w, lo, acc);
        }
        __syncthreads();   // everybody is done reading sH before the next layer's output overwrites it
        G4C_STAMPW(6 + 2 * l);
    }

    G4C_STAMPW(12);
    split_finish<NW>(p, sH, sGB, wave, lane, row0);
    G4C_STAMPW(13);
    if (p.n_heads) {
        __syncthreads();                   // every wave's rows of the final tile are in sH
        for (int hd = 0; hd < p.n_heads; ++hd) {
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc.t[c][q] = 0.f;
#pragma unroll 1
            for (int k0 = 0; k0 < NP; k0 += KC) {
                w += CHUNK_FLOATS;
                if constexpr (B4) mma_chunk_4(sH + i * HS + k0 + 2 * h, ring4, w, lo4, acc);
                else mma_chunk_n<NCT>(sH + i * HS + k0 + 2 * h, ring, w, lo, acc);
            }
            float *ho = p.head_out[hd];
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const long long grow = row0 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    if (grow < p.M) ho[grow * p.head_ld + (ct0 + c) * 32 + i] = acc.t[c][q];
                }
        }
    }
}

// W[n_out, k_in] (nn.Linear layout) -> this layer's chunks of the packed stream:
// chunk c = k/32 (4096 floats), step U = (k%32)/4, step pair V = U/2 (1024 floats), then
// [ct = n/32][lane = ((k%4)/2)*32 + n%32][u = U%2][e = k%2]:
// packed[c*4096 + V*1024 + ct*256 + lane*4 + u*2 + e] = W^T[k][n], zero padded to k_pad x 128,
// with an optional per-block sign flip.  seg tables live in the kernel argument.
struct PackSegs {
    int n_seg;
    int width[G4C_MAX_SRC], wpad[G4C_MAX_SRC], neg[G4C_MAX_SRC];
};

__global__ void pack_layer_kernel(const float *__restrict__ W, int n_out, int k_in, PackSegs segs,
                                  float *__restrict__ packed, int k_pad) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= k_pad * NP) return;
    const int kp = gid / NP, n = gid % NP;
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
    const int U = (kp & 31) >> 2;
    packed[(long long)(kp >> 5) * CHUNK_FLOATS + (U >> 1) * 1024 + (n >> 5) * 256 + (((kp >> 1) & 1) * 32 + (n & 31)) * 4 + (U & 1) * 2 + (kp & 1)] = v;
}


// ======================================================================================================
// fp32-accurate MLP on the bf16 matrix pipe ("bf16x6", opt-in: g4c_mlp_forward_bx6).  Every fp32 operand is split
// EXACTLY into three bf16 terms x = h + m + l (8 + 8 + 8 significand bits); of the nine partial products the six
// largest are formed (hh, hm, mh, mm, hl, lh — each exact in fp32) and accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  The dropped terms (ml, lm, ll) are <= 2^-23 relative: the same order as ONE fp32 rounding,
// so the result is as accurate as the fp32-MFMA kernel (tests compare both against fp64), at 6 x 32 = 192 MFMA cycles
// per 16 k instead of 8 x 64 = 512.  A whole 128-k block is staged per barrier pair; weights are split at pack time
// (g4c_mlp_pack_layer_bx6: [128-k block][column tile][16-k step][plane][lane][8], 6 bytes per weight).
// Template parameter SP = 3: the above.  SP = 1: only the leading terms (operands ROUNDED to bf16, one product) — the
// opt-in "bf16" mode of BASELINE config 3 ("bf16 edge-MLP MFMA", ~1e-2 deviation) on the same stream and structure.
// weight ring depth in 16-k steps.  2 where the launch fills the chip (4 / 8 measured slower: 483 / 563 us against 446 on the level-1
// launch — registers, i.e. waves per SIMD, are what hides the L2 latency there).  8 (a whole 128-k block in flight: the NEXT block's
// weights are requested while this block multiplies) for SMALL launches, <= g_bx6_deep_tiles tiles: with one wave per SIMD nothing else
// covers the L2 round trip, and a ring of 2 exposes it four times per block (2 800 cycles per block against 770 of MFMAs;
// scripts/small_launch_stamps.py)
template <int RD6> struct Ring6 { bf16x8 h[RD6], m[RD6], l[RD6]; };

// one 128-k block for RT row tiles of 32: per 16-k step and row tile 6 MFMAs from the three LDS planes (plane stride
// `plane`, row-tile stride 32*HB); the weight fragments of a step are shared by the row tiles; ring slot s % RD6 is
// refilled RD6 steps ahead.  A fragments are double-buffered per (step, row tile) item: the next item's three
// ds_read_b128 are issued before this item's MFMAs.

constexpr int G4C_BX6_TUNE = 0;      // (1 = s_setprio around the MFMAs, 2 = no sched_barriers in the MFMA loop: both measured neutral)
// SP == 2 (two-way fp16 split, mlp_common.h): planes h / l, three products per step — (Wh, xl) and (Wl, xh) into acc1 (the terms
// that carry the factor 2^-11), (Wh, xh) into acc.
// SWAP (the heads): the two MFMA operands trade places, so the accumulator comes out untransposed — a lane holds ONE output feature
// (lane & 31 of the wave's 32-column slice) of the 16 sample rows 8 (q / 4) + 4 (lane / 32) + q % 4 — and a store instruction writes
// 128 contiguous bytes of each of two rows (same products, same order of k inside the MFMA).
template <int RT, int SP, bool SWAP = false, int RD6 = 2>
__device__ __forceinline__ void mma_block_bx6(const __bf16 *pa, int plane, Ring6<RD6> &g, __amdgpu_buffer_rsrc_t rs, unsigned wofs, unsigned lo_b,
                                              f32x16 (&acc)[RT], f32x16 (&acc1)[RT]) {
    bf16x8 ah = *reinterpret_cast<const bf16x8 *>(pa), am = ah, al = ah;
    if (SP >= 2) am = *reinterpret_cast<const bf16x8 *>(pa + plane);
    if (SP == 3) al = *reinterpret_cast<const bf16x8 *>(pa + 2 * plane);
    // ROLLED over groups of RD6 steps (ring slots are compile-time inside a group): unrolling all 8 steps lets hipcc give
    // every refill fresh registers, which costs a wave of occupancy
    unsigned so = wofs + 2u * RD6 * STEP6;          // byte offset of the step that refills slot 0 (RD6 steps ahead)
#pragma unroll 1
    for (int j = 0; j < 8 / RD6; ++j) {
        const __bf16 *pj = pa + j * RD6 * 16;
        if (j == 8 / RD6 - 1) so = wofs + 2u * BLOCK6;      // the last group refills from the NEXT block's first steps
#pragma unroll
        for (int r = 0; r < RD6; ++r) {
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                // next (row tile, step) item; after the last step of the block the prefetch wraps to step 0 (unused values)
                const int nt = (t + 1) % RT, nr = (t + 1 == RT) ? r + 1 : r;
                const __bf16 *pn = ((nr == RD6 && j == 8 / RD6 - 1) ? pa : pj + nr * 16) + nt * 32 * HB;
                const bf16x8 nh = (G4C_ABLATE & 64) ? ah : *reinterpret_cast<const bf16x8 *>(pn),
                             nm = (SP == 1 || (G4C_ABLATE & 64)) ? am : *reinterpret_cast<const bf16x8 *>(pn + plane),
                             nl = (SP != 3 || (G4C_ABLATE & 64)) ? al : *reinterpret_cast<const bf16x8 *>(pn + 2 * plane);
                if (!(G4C_BX6_TUNE & 2)) __builtin_amdgcn_sched_barrier(0);
                if (G4C_BX6_TUNE & 1) __builtin_amdgcn_s_setprio(1);
                if (SP == 3 && !(G4C_ABLATE & 128)) {
                if (SWAP) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, g.h[r], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, g.l[r], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, g.m[r], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, g.h[r], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, g.m[r], acc[t], 0, 0, 0);
                } else {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g.h[r], al, acc[t], 0, 0, 0);     // small terms first
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g.l[r], ah, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g.m[r], am, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g.h[r], am, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g.m[r], ah, acc[t], 0, 0, 0);
                }
                } else if (SP == 3) {      // (ablation) keep every operand live with one cheap VALU op instead of five MFMAs
                    acc[t][0] += (float)al[0] + (float)am[0] + (float)g.l[r][0] + (float)g.m[r][0];
                }
                if (SP == 2) {
                    if (SWAP) {
                        acc1[t] = mfma_f16(am, g.h[r], acc1[t]);
                        acc1[t] = mfma_f16(ah, g.m[r], acc1[t]);
                        acc[t] = mfma_f16(ah, g.h[r], acc[t]);
                    } else {
                        acc1[t] = mfma_f16(g.h[r], am, acc1[t]);
                        acc1[t] = mfma_f16(g.m[r], ah, acc1[t]);
                        acc[t] = mfma_f16(g.h[r], ah, acc[t]);
                    }
                } else
                acc[t] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, g.h[r], acc[t], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(g.h[r], ah, acc[t], 0, 0, 0);
                if (G4C_BX6_TUNE & 1) __builtin_amdgcn_s_setprio(0);
                if (t + 1 == RT && !(G4C_ABLATE & 32)) {
                    const unsigned sx = (G4C_ABLATE & 1024) ? 0u : so + 2u * r * STEP6;     // (1024: always the same 3 KB -> L1 hits)
                    g.h[r] = ldw(rs, lo_b, sx);
                    if (SP >= 2) g.m[r] = ldw(rs, lo_b + 1024u, sx);
                    if (SP == 3) g.l[r] = ldw(rs, lo_b + 2048u, sx);
                }
                if (!(G4C_BX6_TUNE & 2)) __builtin_amdgcn_sched_barrier(0);
                ah = nh; am = nm; al = nl;
            }
        }
        so += 2u * RD6 * STEP6;
    }
}


constexpr int G4C_SEG_CHUNKS = 1;
constexpr int G4C_SEG_INFLIGHT = 6;     // rows of a segment in flight per column chunk when a source is aggregated on load
constexpr int G4C_BX6_MINW = 4;         // workgroups per CU the instantiations are register-limited for (launch_bounds)
constexpr int G4C_F16_MINW = 4;
// RT = 1: 32-row tile.  RT = 2: 64-row tile — every weight fragment feeds two row tiles (half the L2 -> register weight
// traffic per row, which is what this kernel stalls on) and every memory round trip of the tile's critical path serves
// twice the rows.
// FULL: every weighted input block and every additive block is exactly 128 wide and 16-byte aligned (the MP layers):
// no column masks anywhere.
template <int RT, bool VEC, bool FULL, int SP, bool SAVE = false, int RD6 = 2>
__global__ __launch_bounds__(256, RD6 > 2 ? 2 : (SP == 2 ? G4C_F16_MINW : G4C_BX6_MINW)) void mlp_bx6_kernel(const Params p) {
    static_assert(RT == 1, "64-row tiles (RT = 2) measured slower in every arithmetic (split streams, round 3: 480 against 446 us; rounded-bf16 mode, round 6, "
                           "500k-row edge update with heads: 288 against 255 us) and cannot take the fused aggregation: not instantiated");
    constexpr int ROWS = 32 * RT, NW = 4;
    constexpr int PLN = ROWS * HB;              // one bf16 operand plane [ROWS][136]
    // three operand planes; the fp32 final tile [ROWS][132] aliases them
    constexpr int BUF_FLOATS = (SP == 2 ? 2 : 3) * PLN / 2;      // (SP == 2: two planes; [ROWS][132] floats still fit in [2][ROWS][136] halves)
    static_assert(BUF_FLOATS >= ROWS * HS, "final tile must fit");
    // RT = 2 keeps three workgroups per CU (<= 54.6 KB of LDS each): two index slots per kind (the launcher checks) and the
    // biases read from global memory (L1 hits) instead of an LDS copy
    constexpr int NSLOT = RT == 2 ? 2 : G4C_MAX_SRC;
    constexpr bool LDS_BIAS = RT == 1;
    constexpr int BIAS_FLOATS = LDS_BIAS ? G4C_MAX_LAYERS * NP : 0;
    // SMALL (the deep-ring instantiation of launches of few tiles: one or two workgroups per CU, LDS to spare): the fp32 final tile
    // has its own buffer instead of aliasing the planes — one barrier less in front of the heads — and both heads' MFMA blocks run
    // back to back on separate accumulators, all head stores behind them
    constexpr bool SMALL = RD6 > 2;
    constexpr int FIN_FLOATS = SMALL ? ROWS * HS : 0;
    __shared__ __attribute__((aligned(16))) float lds[BUF_FLOATS + 2 * NSLOT * ROWS + BIAS_FLOATS + 2 * NP + FIN_FLOATS];
    float *sH = SMALL ? lds + BUF_FLOATS + 2 * NSLOT * ROWS + BIAS_FLOATS + 2 * NP : lds;
    __bf16 *sB = reinterpret_cast<__bf16 *>(lds);
    int *sRow = reinterpret_cast<int *>(lds + BUF_FLOATS);
    int *sRowAdd = sRow + NSLOT * ROWS;
    float *sBias = lds + BUF_FLOATS + 2 * NSLOT * ROWS;
    float *sGB = sBias + BIAS_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int ct0 = wave;
    if (SP == 2) f16_range_mode();
    RangeS rng;                       // lanes that converted a value beyond the fp16 range (SP == 2: mlp_common.h range_track)

    int tile;
    {
        const int b = blockIdx.x, nt = p.n_tiles;
        const int q = nt >> 3, r = nt & 7, x = b & 7, j = b >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    // (wave-uniform 32-bit values in scalar registers — the launcher checks n_rows < 2^31: as 64-bit per-lane values they were the
    // registers the SP == 2 instantiations spilled, and a spilled uniform costs 64 lanes of scratch traffic per wave)
    int row0 = __builtin_amdgcn_readfirstlane((int)(p.row_base + (long long)tile * ROWS)), mlim = __builtin_amdgcn_readfirstlane((int)p.M);
    if (p.tile_rows) {      // tile of whole segments (<= ROWS rows)
        row0 = __builtin_amdgcn_readfirstlane(p.tile_rows[tile]); mlim = __builtin_amdgcn_readfirstlane(p.tile_rows[tile + 1]);
    }
    G4C_STAMPW(0);

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, 0x7fffffff, 0x00020000);
    const unsigned lo_b = 2u * (unsigned)(ct0 * 8 * STEP6 + lane * 8);    // this lane's byte offset inside a block of the stream
    unsigned wofs = 0;                                                     // byte offset of the current block (wave-uniform)
    Ring6<RD6> ring;

    // this wave's rows of an input block: lane -> row (lane>>3) + 8*wave (+ 32 per row tile), 4 floats at column
    // 4*(lane&7) of each 32-k chunk.  A block whose rows are not gathered through an index can start right away.
    const int grow_l = (lane >> 3) + 8 * wave, c4 = (lane & 7) * 4;
    f32x4 xp[RT][4];
    auto gather = [&](int sidx, bool direct) __attribute__((always_inline)) {
        const int width = p.src[sidx].width;
        if (p.src[sidx].seg_off) {
            // aggregation on load: this lane's row is the sum / mean of a CSR segment of the source's rows, added in
            // order (bit-identical to segment_reduce_kernel); four rows in flight per column chunk
            const int *so = p.src[sidx].seg_off, *sp = p.src[sidx].seg_perm;
            const int ld = p.src[sidx].ld, sact = p.src[sidx].pre_act;
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                long long gr = row0 + grow_l + 32 * t;
                if (gr >= mlim) gr = mlim - 1;
                const int b = so[gr], e = so[gr + 1];
                const float *rp = p.src[sidx].ptr + p.src[sidx].col0 + c4;
                // G4C_SEG_CHUNKS 32-column chunks at a time, G4C_SEG_INFLIGHT rows of each in flight (bounded registers; the
                // row order of the additions is the segment order whichever way the loads are batched)
#pragma unroll
                for (int q0 = 0; q0 < 4; q0 += G4C_SEG_CHUNKS) {
                    f32x4 a[G4C_SEG_CHUNKS];
#pragma unroll
                    for (int qq = 0; qq < G4C_SEG_CHUNKS; ++qq) a[qq] = f32x4{0.f, 0.f, 0.f, 0.f};
                    for (int r = b; r < e; r += G4C_SEG_INFLIGHT) {
                        f32x4 v[G4C_SEG_INFLIGHT][G4C_SEG_CHUNKS];
#pragma unroll
                        for (int u = 0; u < G4C_SEG_INFLIGHT; ++u) {
                            int rr = (r + u < e) ? r + u : e - 1;
                            if (sp) rr = sp[rr];
#pragma unroll
                            for (int qq = 0; qq < G4C_SEG_CHUNKS; ++qq)
                                v[u][qq] = *reinterpret_cast<const f32x4 *>(rp + (long long)rr * ld + (q0 + qq) * KC);
                        }
                        if (sact) {          // the pending activation of the stored rows applies BEFORE the reduction
#pragma unroll
                            for (int u = 0; u < G4C_SEG_INFLIGHT; ++u)
#pragma unroll
                                for (int qq = 0; qq < G4C_SEG_CHUNKS; ++qq) v[u][qq] = selu4(v[u][qq]);
                        }
#pragma unroll
                        for (int u = 0; u < G4C_SEG_INFLIGHT; ++u) {
                            const bool on = r + u < e;
#pragma unroll
                            for (int qq = 0; qq < G4C_SEG_CHUNKS; ++qq)
#pragma unroll
                                for (int el = 0; el < 4; ++el) a[qq][el] += on ? v[u][qq][el] : 0.f;
                        }
                    }
#pragma unroll
                    for (int qq = 0; qq < G4C_SEG_CHUNKS; ++qq) xp[t][q0 + qq] = a[qq];
                }
                if (p.src[sidx].seg_mean) {
                    const float cnt = (float)((e - b) > 1 ? (e - b) : 1);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int el = 0; el < 4; ++el) xp[t][q][el] /= cnt;
                }
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            long long gr;
            if (direct) { gr = row0 + grow_l + 32 * t; if (gr >= mlim) gr = mlim - 1; }
            else gr = sRow[sidx * ROWS + grow_l + 32 * t];
            const float *rp = p.src[sidx].ptr + gr * p.src[sidx].ld + p.src[sidx].col0;
            if (SP == 1 && p.src[sidx].bf16) {        // bf16 rows (128 wide, 8-byte aligned: the launcher checks): widen by a shift / a mask
                const __bf16 *rp16 = reinterpret_cast<const __bf16 *>(p.src[sidx].ptr) + gr * p.src[sidx].ld + p.src[sidx].col0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 w = *reinterpret_cast<const u32x2 *>(rp16 + q * KC + c4);
                    xp[t][q][0] = __builtin_bit_cast(float, w[0] << 16); xp[t][q][1] = __builtin_bit_cast(float, w[0] & 0xffff0000u);
                    xp[t][q][2] = __builtin_bit_cast(float, w[1] << 16); xp[t][q][3] = __builtin_bit_cast(float, w[1] & 0xffff0000u);
                }
            } else
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = q * KC + c4;
                if (FULL) {
                    xp[t][q] = *reinterpret_cast<const f32x4 *>(rp + c);
                } else if (VEC) {
                    xp[t][q] = *reinterpret_cast<const f32x4 *>(rp + (c < width ? c : 0));
                } else {
                    const int w1 = width - 1;
                    xp[t][q][0] = rp[c + 0 < w1 ? c + 0 : w1]; xp[t][q][1] = rp[c + 1 < w1 ? c + 1 : w1];
                    xp[t][q][2] = rp[c + 2 < w1 ? c + 2 : w1]; xp[t][q][3] = rp[c + 3 < w1 ? c + 3 : w1];
                }
            }
        }
    };
    auto park_impl = [&](int sidx, auto act_tag) __attribute__((always_inline)) {
        constexpr bool ACT = decltype(act_tag)::value;
        const int width = p.src[sidx].width;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            __bf16 *d = sB + (grow_l + 32 * t) * HB + c4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = q * KC + c4;
                f32x4 v = xp[t][q];
                if (!FULL) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (c + e < width) ? v[e] : 0.f;
                }
                if (ACT) v = selu4(v);
                bf16x4 vh, vm, vl;
                split3x4<SP>(v, vh, vm, vl, rng);
                *reinterpret_cast<bf16x4 *>(d + q * KC) = vh;
                if (SP >= 2) *reinterpret_cast<bf16x4 *>(d + PLN + q * KC) = vm;
                if (SP == 3) *reinterpret_cast<bf16x4 *>(d + 2 * PLN + q * KC) = vl;
                __builtin_amdgcn_sched_barrier(0);      // one group of four at a time: bounds the live temporaries
            }
        }
    };
    auto park = [&](int sidx) __attribute__((always_inline)) {            // one uniform branch per block, not one per element
        if (p.src[sidx].pre_act && !p.src[sidx].seg_off) park_impl(sidx, std::true_type{});
        else park_impl(sidx, std::false_type{});
    };
    // Memory instructions return in order per wave, so the loads that head a dependent chain go FIRST: row indices (the
    // additive gathers wait for them), then the parameters, then the long-latency streams (weights, directly indexed input).
    int idx_v[(2 * NSLOT * ROWS + 64 * NW - 1) / (64 * NW)];
#pragma unroll
    for (int it = 0; it < (2 * NSLOT * ROWS + 64 * NW - 1) / (64 * NW); ++it) {
        const int e = tid + it * 64 * NW;
        const int slot = e / ROWS, r = e % ROWS;
        long long gr = row0 + r;
        if (gr >= mlim) gr = mlim - 1;
        // selects among the (uniform) index pointers: p.src[slot] with the per-lane slot would be a load from the kernel argument
        // segment — one more dependent round trip in front of every tile's index loads
        const int *ix = nullptr;
#pragma unroll
        for (int s2 = 0; s2 < NSLOT; ++s2) {
            const int *is = s2 < p.n_src ? p.src[s2].idx : nullptr, *ia = s2 < p.n_add ? p.add[s2].idx : nullptr;
            ix = slot == s2 ? is : (slot == NSLOT + s2 ? ia : ix);
        }
        idx_v[it] = ix ? ix[gr] : (int)gr;
    }
    float bias_v[(G4C_MAX_LAYERS * NP) / (64 * NW)], gb_v = 0.f;
#pragma unroll
    for (int it = 0; it < (G4C_MAX_LAYERS * NP) / (64 * NW); ++it) {
        const int e = tid + it * 64 * NW;
        bias_v[it] = LDS_BIAS ? p.b[e < p.n_layers * NP ? e : 0] : 0.f;
    }
    if (p.gamma) {
        const int e = tid & (NP - 1);
        const int ee = e < p.n_out ? e : 0;
        gb_v = tid < NP ? p.gamma[ee] : p.beta[ee];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < RD6; ++s) {
        ring.h[s] = ldw(rs, lo_b, 2u * s * STEP6);
        if (SP >= 2) ring.m[s] = ldw(rs, lo_b + 1024u, 2u * s * STEP6);
        if (SP == 3) ring.l[s] = ldw(rs, lo_b + 2048u, 2u * s * STEP6);
    }
    const bool direct0 = p.n_src > 0 && (p.src[0].idx == nullptr);
    if (direct0) gather(0, true);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < (2 * NSLOT * ROWS + 64 * NW - 1) / (64 * NW); ++it) {
        const int e = tid + it * 64 * NW;
        if (e < 2 * NSLOT * ROWS) sRow[e] = idx_v[it];
    }
    if (LDS_BIAS) {
#pragma unroll
        for (int it = 0; it < (G4C_MAX_LAYERS * NP) / (64 * NW); ++it) sBias[tid + it * 64 * NW] = bias_v[it];
    }
    if (p.gamma) sGB[tid] = gb_v;          // [gamma(128) | beta(128)] = 256 threads
    __syncthreads();
    G4C_STAMPW(1);
    if (!direct0 && p.n_src > 0) gather(0, false);
    __builtin_amdgcn_sched_barrier(0);

    if (p.n_src > 0) park(0);          // (before the additive gathers: the input registers are free again while those are in flight)
    G4C_STAMPW(2);
    // Operands are swapped in the MFMAs (weights as A, activations as B), so the accumulators are TRANSPOSED: this lane
    // holds sample row i (= lane & 31, + 32 per row tile) and the 16 output features 32*ct0 + 8*(q>>2) + 4*h + (q&3):
    // four runs of four consecutive features -> 16-byte gathers / LDS accesses instead of 16 scalar ones.
    f32x16 acc[RT], acc1[RT];       // acc1: the 2^-11 terms of the two-way fp16 split (SP == 2 only)
    const int fbase = ct0 * 32 + 4 * h;
    // a layer's accumulators start at its bias (then the additive rows, then the products — the order mlp_ws_kernel adds them in):
    // no bias add in the epilogues
    auto bias_start = [&](int l) __attribute__((always_inline)) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>((LDS_BIAS ? sBias : p.b) + l * NP + fbase + 8 * gq);
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[t][4 * gq + e] = b4[e]; acc1[t][4 * gq + e] = 0.f; }
        }
    };
    bias_start(0);
    for (int a = 0; a < p.n_add; ++a) {
        const int width = p.add[a].width;
        const bool vec = FULL || (((p.add[a].ld & 3) == 0) && ((width & 3) == 0) && (((uintptr_t)p.add[a].ptr & 15) == 0));
        f32x4 tt[RT][4];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const float *pr = p.add[a].ptr + (long long)sRowAdd[a * ROWS + i + 32 * t] * p.add[a].ld;
            if (SP == 1 && p.add[a].bf16) {          // bf16 rows (128 wide, 8-byte aligned: the launcher checks), widened by a shift / a mask
                const __bf16 *pr16 = reinterpret_cast<const __bf16 *>(p.add[a].ptr) + (long long)sRowAdd[a * ROWS + i + 32 * t] * p.add[a].ld;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) tt[t][gq] = widen_bf16x4(*reinterpret_cast<const u32x2 *>(pr16 + fbase + 8 * gq));
            } else
            if (vec) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int f = fbase + 8 * gq;
                    tt[t][gq] = *reinterpret_cast<const f32x4 *>(pr + ((FULL || f < width) ? f : 0));
                }
            } else {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int f = fbase + 8 * gq + e;
                        tt[t][gq][e] = pr[f < width ? f : 0];
                    }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t][4 * gq + e] += (FULL || fbase + 8 * gq + e < width) ? tt[t][gq][e] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
    }
    // narrow input blocks (2..8 columns): x[row, k] * W1^T[k, :] in fp32 on the vector ALUs — a padded 128-k block of
    // six-product MFMAs for 2 columns of input would cost 48 MFMAs per wave; this costs 16 FMAs per column
    for (int a = 0; a < p.n_nar; ++a) {
        const float *wn = p.nar[a].w + fbase;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            long long gr = row0 + i + 32 * t;
            if (gr >= mlim) gr = mlim - 1;
            const float *xr = p.nar[a].ptr + gr * p.nar[a].ld;
            for (int kk = 0; kk < p.nar[a].width; ++kk) {
                const float x = xr[kk];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4 *>(wn + kk * NP + 8 * gq);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][4 * gq + e] = fmaf(x, w4[e], acc[t][4 * gq + e]);
                }
            }
        }
    }
    __syncthreads();
    G4C_STAMPW(3);

    // ---------------------------------------------------------------- layer 0: one (padded) 128-k input block at a time
    const __bf16 *pa = sB + i * HB + 8 * h;
    for (int s = 0; s < p.n_src; ++s) {
        const bool more = s + 1 < p.n_src;
        if (more && RT == 1) gather(s + 1, false);          // (RT = 2: 32 more live registers would cost a wave per SIMD)
        __builtin_amdgcn_sched_barrier(0);
        mma_block_bx6<RT, SP, false, RD6>(pa, PLN, ring, rs, wofs, lo_b, acc, acc1);
        wofs += 2u * BLOCK6;
        __syncthreads();                   // everybody is done reading the planes
        if (more) {
            if (RT != 1) gather(s + 1, false);
            park(s + 1);
            __syncthreads();
        }
    }
    G4C_STAMPW(4);
    for (int l = 0;; ++l) {
        const bool last = (l == p.n_layers - 1);
        if (last) {
            // final tile in fp32 for the LayerNorm / store epilogue (aliases the operand planes: everybody finished
            // reading them at the barrier that closed the previous block)
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    f32x4 x;
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = SP == 2 ? fmaf(acc1[t][4 * gq + e], F16_LO_UNSCALE, acc[t][4 * gq + e]) : acc[t][4 * gq + e];
                    *reinterpret_cast<f32x4 *>(sH + (i + 32 * t) * HS + fbase + 8 * gq) = x;
                    if (SAVE) {
                        const long long gr = row0 + i + 32 * t;
                        if (p.save[l] && gr < mlim) *reinterpret_cast<f32x4 *>(p.save[l] + gr * p.save_ld + fbase + 8 * gq) = x;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            __syncthreads();
            G4C_STAMPW(5 + 2 * l);
            break;
        }
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                f32x4 x;
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = SP == 2 ? fmaf(acc1[t][4 * gq + e], F16_LO_UNSCALE, acc[t][4 * gq + e]) : acc[t][4 * gq + e];
                bf16x4 vh, vm, vl;
                f32x4 y;
                if (SAVE && p.mul[l]) {
                    const long long gr = row0 + i + 32 * t;
                    const f32x4 r = *reinterpret_cast<const f32x4 *>(p.mul[l] + (gr < mlim ? gr : mlim - 1) * p.mul_ld + fbase + 8 * gq);
                    const float sc = 1.0507009873554804934193349852946f, sa = 1.7580993408473768599402175208123f;   // scale, scale * alpha
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = x[e] * (r[e] > 0.f ? sc : r[e] + sa);
                } else {
                    y = selu4(x);
                }
                if (SAVE) {
                    const long long gr = row0 + i + 32 * t;
                    if (p.save[l] && gr < mlim) *reinterpret_cast<f32x4 *>(p.save[l] + gr * p.save_ld + fbase + 8 * gq) = y;
                }
                split3x4<SP>(y, vh, vm, vl, rng);
                __bf16 *d = sB + (i + 32 * t) * HB + fbase + 8 * gq;
                *reinterpret_cast<bf16x4 *>(d) = vh;
                if (SP >= 2) *reinterpret_cast<bf16x4 *>(d + PLN) = vm;
                if (SP == 3) *reinterpret_cast<bf16x4 *>(d + 2 * PLN) = vl;
                __builtin_amdgcn_sched_barrier(0);
            }
        __syncthreads();
        G4C_STAMPW(5 + 2 * l);
        bias_start(l + 1);
        mma_block_bx6<RT, SP, false, RD6>(pa, PLN, ring, rs, wofs, lo_b, acc, acc1);
        wofs += 2u * BLOCK6;
        __syncthreads();
        G4C_STAMPW(6 + 2 * l);
    }
    G4C_STAMPW(12);
    split_finish<NW, ROWS>(p, sH, sGB, wave, lane, row0, mlim);
    G4C_STAMPW(13);
    if (p.agg) {
        // aggregation of the targets whose messages this tile holds (rows in CSR order): same summation order and the
        // same mean formula as segment_reduce_kernel, so the result is bit-identical to the separate launch
        __syncthreads();
        const int s0 = p.tile_seg[tile], s1 = p.tile_seg[tile + 1];
        const int col = tid & (NP - 1);
        for (int sg = s0 + (tid >> 7); sg < s1; sg += (64 * NW) >> 7) {
            const int b = p.seg_off[sg] - (int)row0, e = p.seg_off[sg + 1] - (int)row0;
            float a = 0.f;
            for (int r = b; r < e; ++r) a += sH[r * HS + col];
            if (p.agg_mean) a /= (float)((e - b) > 1 ? (e - b) : 1);
            p.agg[(long long)sg * p.agg_ld + col] = a;
        }
    }
    if (p.n_heads) {
        // heads (see Params): the finished fp32 tile -> three operand planes (they alias it: read everything, barrier,
        // then overwrite), then one 128-k block per head whose weights continue the stream
        __syncthreads();
        f32x4 v[RT][4];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[t][q] = *reinterpret_cast<const f32x4 *>(sH + (grow_l + 32 * t) * HS + q * KC + c4);
        if (!SMALL) __syncthreads();          // (SMALL: the planes do not alias the tile; their last readers are behind older barriers)
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16x4 vh, vm, vl;
                split3x4<SP>(v[t][q], vh, vm, vl, rng);
                __bf16 *d = sB + (grow_l + 32 * t) * HB + q * KC + c4;
                *reinterpret_cast<bf16x4 *>(d) = vh;
                if (SP >= 2) *reinterpret_cast<bf16x4 *>(d + PLN) = vm;
                if (SP == 3) *reinterpret_cast<bf16x4 *>(d + 2 * PLN) = vl;
            }
        __syncthreads();
        G4C_STAMPW(14);
        // untransposed accumulator (SWAP): lane (i, h) holds column ct0 * 32 + i of the rows 32 t + 8 gq + 4 h + e; one dword
        // store per value, 32 lanes = one 128-byte line of a row.  The rows of the tile past mlim fall outside the buffer
        // descriptor's range (num_records) and are dropped by the hardware.
        const long long left = mlim - row0;
        // (wave-uniform by construction, but a value the vector ALU produced — the clamp below is a v_med3 — lives in a vector register,
        // and a buffer descriptor built from vector registers makes hipcc wrap EVERY store in a readfirstlane "waterfall" loop: twelve
        // instructions per stored dword, 1 450 cycles per head of a tile.  Explicitly scalar:)
        const int nrows = __builtin_amdgcn_readfirstlane(left >= ROWS ? ROWS : (left > 0 ? (int)left : 0));
        const unsigned vo = (unsigned)(4 * h * p.head_ld + ct0 * 32 + i) * 4u;
        auto head_store = [&](int hd, const f32x16 (&a)[RT], const f32x16 (&a1)[RT]) __attribute__((always_inline)) {
            // (rounded-bf16 mode, head_bf16: the rows are stored as bf16 — 2-byte elements: a store instruction writes 64 contiguous
            // bytes of each of two rows; their reader adds them to fp32 accumulators after widening)
            const int esz = (SP == 1 && p.head_bf16) ? 2 : 4;
            const unsigned long long hb = reinterpret_cast<unsigned long long>(p.head_out[hd]) + (unsigned long long)row0 * p.head_ld * esz;
            float *const hbase = reinterpret_cast<float *>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(hb >> 32)) << 32) |
                                                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)hb));
            const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(hbase, 0, nrows * p.head_ld * esz, 0x00020000);
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = SP == 2 ? fmaf(a1[t][4 * gq + e], F16_LO_UNSCALE, a[t][4 * gq + e]) : a[t][4 * gq + e];
                        // (the row offset is part of the VECTOR offset: the scalar offset is not range-checked)
                        if (SP == 1 && p.head_bf16)
                            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)x), rh,
                                                                  (vo >> 1) + (unsigned)((32 * t + 8 * gq + e) * p.head_ld) * 2u, 0, 0);
                        else
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), rh, vo + (unsigned)((32 * t + 8 * gq + e) * p.head_ld) * 4u, 0, 0);
                    }
        };
        auto head_zero = [&](f32x16 (&a)[RT], f32x16 (&a1)[RT]) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int q = 0; q < 16; ++q) { a[t][q] = 0.f; a1[t][q] = 0.f; }
        };
        if constexpr (SMALL) {
            static_assert(G4C_MAX_HEADS == 2, "the small-launch head path multiplies both heads before it stores either");
            f32x16 acc2[RT], acc21[RT];
            head_zero(acc, acc1); head_zero(acc2, acc21);
            mma_block_bx6<RT, SP, true, RD6>(pa, PLN, ring, rs, wofs, lo_b, acc, acc1);
            wofs += 2u * BLOCK6;
            if (p.n_heads > 1) mma_block_bx6<RT, SP, true, RD6>(pa, PLN, ring, rs, wofs, lo_b, acc2, acc21);
            head_store(0, acc, acc1);
            if (p.n_heads > 1) head_store(1, acc2, acc21);
        } else {
            for (int hd = 0; hd < p.n_heads; ++hd) {
                head_zero(acc, acc1);
                mma_block_bx6<RT, SP, true, RD6>(pa, PLN, ring, rs, wofs, lo_b, acc, acc1);
                wofs += 2u * BLOCK6;
                head_store(hd, acc, acc1);
            }
        }
    }
    G4C_STAMPW(15);
    if (SP == 2) range_report(p, rng);
}

// bf16x6 image of one layer: three planes (h, m, l) of the exact split of every weight.  F16: the two-way fp16 split (h, l * 2^11)
// in planes 0 / 1 of the same layout, plane 2 zero.
template <bool F16>
__global__ void pack_layer_bx6_kernel(const float *__restrict__ W, int n_out, int k_in, PackSegs segs,
                                      __bf16 *__restrict__ packed, int k_pad) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= k_pad * NP) return;
    const int kp = gid / NP, n = gid % NP;
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
    __bf16 a, b, c;
    if (F16) { split2(v, a, b); c = (__bf16)0.f; }
    else split3(v, a, b, c);
    const int kk = kp & 127;
    __bf16 *d = packed + (long long)(kp >> 7) * BLOCK6 + (n >> 5) * 8 * STEP6 + (kk >> 4) * STEP6 + (((kk >> 3) & 1) * 32 + (n & 31)) * 8 + (kk & 7);
    d[0] = a; d[512] = b; d[1024] = c;
}

}  // namespace

extern "C" int g4c_mlp_pack_layer(const float *W, int32_t n_out, int32_t k_in, const int32_t *seg_width,
                                  const int32_t *seg_negate, int32_t n_seg, float *packed,
                                  int32_t k_pad, int32_t n_pad, void *stream) {
    G4C_REQUIRE(W && packed && seg_width, G4C_EINVAL, "g4c_mlp_pack_layer: null pointer");
    G4C_REQUIRE(n_seg >= 1 && n_seg <= G4C_MAX_SRC, G4C_EUNSUPPORTED, "g4c_mlp_pack_layer: %d input blocks (max %d)", n_seg, G4C_MAX_SRC);
    G4C_REQUIRE(n_out >= 1 && n_out <= NP, G4C_EUNSUPPORTED, "g4c_mlp_pack_layer: layer width %d unsupported (max 128)", n_out);
    G4C_REQUIRE(n_pad == NP, G4C_EINVAL, "g4c_mlp_pack_layer: n_pad must be 128, got %d", n_pad);
    PackSegs segs;
    segs.n_seg = n_seg;
    int ksum = 0, kpsum = 0;
    for (int s = 0; s < n_seg; ++s) {
        G4C_REQUIRE(seg_width[s] > 0, G4C_EINVAL, "g4c_mlp_pack_layer: empty input block %d", s);
        segs.width[s] = seg_width[s];
        segs.wpad[s] = (seg_width[s] + KC - 1) / KC * KC;
        segs.neg[s] = seg_negate ? seg_negate[s] : 0;
        ksum += segs.width[s];
        kpsum += segs.wpad[s];
    }
    G4C_REQUIRE(ksum == k_in, G4C_EINVAL, "g4c_mlp_pack_layer: blocks sum to %d columns, weight has %d", ksum, k_in);
    G4C_REQUIRE(kpsum <= k_pad && k_pad % KC == 0, G4C_EINVAL, "g4c_mlp_pack_layer: k_pad %d too small for %d (or not a multiple of 32)", k_pad, kpsum);
    const int total = k_pad * NP;
    g4c::DeviceGuard on_device(packed);
    pack_layer_kernel<<<dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(W, n_out, k_in, segs, packed, k_pad);
    return g4c::check_launch("g4c_mlp_pack_layer");
}

static int pack_layer_16(bool f16, const float *W, int32_t n_out, int32_t k_in, const int32_t *seg_width,
                         const int32_t *seg_negate, int32_t n_seg, void *packed, int32_t k_pad, int32_t n_pad, void *stream);

extern "C" int g4c_mlp_pack_layer_bx6(const float *W, int32_t n_out, int32_t k_in, const int32_t *seg_width,
                                      const int32_t *seg_negate, int32_t n_seg, void *packed,
                                      int32_t k_pad, int32_t n_pad, void *stream) {
    return pack_layer_16(false, W, n_out, k_in, seg_width, seg_negate, n_seg, packed, k_pad, n_pad, stream);
}

extern "C" int g4c_mlp_pack_layer_f16x3(const float *W, int32_t n_out, int32_t k_in, const int32_t *seg_width,
                                        const int32_t *seg_negate, int32_t n_seg, void *packed,
                                        int32_t k_pad, int32_t n_pad, void *stream) {
    return pack_layer_16(true, W, n_out, k_in, seg_width, seg_negate, n_seg, packed, k_pad, n_pad, stream);
}

static int pack_layer_16(bool f16, const float *W, int32_t n_out, int32_t k_in, const int32_t *seg_width,
                         const int32_t *seg_negate, int32_t n_seg, void *packed, int32_t k_pad, int32_t n_pad, void *stream) {
    G4C_REQUIRE(W && packed && seg_width, G4C_EINVAL, "g4c_mlp_pack_layer_bf16: null pointer");
    G4C_REQUIRE(n_seg >= 1 && n_seg <= G4C_MAX_SRC, G4C_EUNSUPPORTED, "g4c_mlp_pack_layer_bf16: %d input blocks (max %d)", n_seg, G4C_MAX_SRC);
    G4C_REQUIRE(n_out >= 1 && n_out <= NP, G4C_EUNSUPPORTED, "g4c_mlp_pack_layer_bf16: layer width %d unsupported (max 128)", n_out);
    G4C_REQUIRE(n_pad == NP && k_pad == NP * n_seg, G4C_EINVAL, "g4c_mlp_pack_layer_bf16: n_pad must be 128 and k_pad 128 per input block (got %d, %d)", n_pad, k_pad);
    PackSegs segs;
    segs.n_seg = n_seg;
    int ksum = 0;
    for (int s = 0; s < n_seg; ++s) {
        G4C_REQUIRE(seg_width[s] > 0 && seg_width[s] <= NP, G4C_EUNSUPPORTED, "g4c_mlp_pack_layer_bf16: input block %d is %d wide (1..128)", s, seg_width[s]);
        segs.width[s] = seg_width[s];
        segs.wpad[s] = NP;
        segs.neg[s] = seg_negate ? seg_negate[s] : 0;
        ksum += segs.width[s];
    }
    G4C_REQUIRE(ksum == k_in, G4C_EINVAL, "g4c_mlp_pack_layer_bf16: blocks sum to %d columns, weight has %d", ksum, k_in);
    const int total = k_pad * NP;
    g4c::DeviceGuard on_device(packed);
    if (f16) pack_layer_bx6_kernel<true><<<dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(W, n_out, k_in, segs, (__bf16 *)packed, k_pad);
    else pack_layer_bx6_kernel<false><<<dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(W, n_out, k_in, segs, (__bf16 *)packed, k_pad);
    return g4c::check_launch("g4c_mlp_pack_layer_bx6");
}

extern "C" int g4c_mlp_forward_rows(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                                    int64_t row_begin, int64_t row_count, int32_t tile_rows,
                                    float *out, int32_t out_ld, const int32_t *out_idx, int32_t act,
                                    const float *resid, int32_t resid_ld, int32_t resid_col0, void *stream);

extern "C" int g4c_mlp_forward(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                               float *out, int32_t out_ld, const int32_t *out_idx, int32_t act,
                               const float *resid, int32_t resid_ld, int32_t resid_col0, void *stream) {
    // fp32 MFMA: one kernel family, the 4-wave column split on 32-row tiles (measured the fastest or within 2 % of the fastest of
    // the variants tried in round 1 at every launch size from 8k to 1M rows)
    return g4c_mlp_forward_rows(mlp, srcs, n_src, n_rows, 0, n_rows, 324, out, out_ld, out_idx, act, resid, resid_ld, resid_col0, stream);
}

struct SaveArgs {          // training forward / backward chain (g4c_mlp_forward_bx6_save)
    float *const *ptr;     // n_layers entries, each may be null
    int32_t ld;
    const float *const *mul;   // null, or n_layers entries, each may be null
    int32_t mul_ld;
};

struct AggArgs {           // fused aggregation (g4c_mlp_forward_bx6_agg); all null / 0 otherwise
    const int32_t *tile_rows, *tile_seg, *seg_off;
    int32_t n_tiles;
    float *out;
    int32_t out_ld, mean;
    int32_t rows_bf16;       // the MLP's output rows are stored as bf16 (g4c_mlp_forward_bf16_agg, out_dtype)
};
// rounded-bf16 mode, plain / heads launches: bf16 output rows (g4c_mlp_forward_bf16_out) and bf16 head rows (g4c_mlp_forward_heads_bf16_out)
static thread_local int g_out_dtype = 0, g_head_dtype = 0;

struct NodeArgs {          // the node update fused behind the message launch (g4c_mp_layer_forward_bx6)
    const g4c_mlp_t *upd;
    const float *v;
    int32_t v_ld, act;
    float *v_out;
    int32_t v_out_ld;
    const void *head_w;
    int32_t n_heads;
    float *const *head_out;
    int32_t head_ld;
};

static int mlp_launch(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                      int64_t row_begin, int64_t row_count, int32_t tile_rows,
                      float *out, int32_t out_ld, const int32_t *out_idx, int32_t act,
                      const float *resid, int32_t resid_ld, int32_t resid_col0,
                      const float *head_w, int32_t n_heads, float *const *head_out, int32_t head_ld, void *stream,
                      const AggArgs *agg = nullptr, const SaveArgs *save = nullptr, const NodeArgs *node = nullptr);

extern "C" int g4c_mlp_forward_rows(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                                    int64_t row_begin, int64_t row_count, int32_t tile_rows,
                                    float *out, int32_t out_ld, const int32_t *out_idx, int32_t act,
                                    const float *resid, int32_t resid_ld, int32_t resid_col0, void *stream) {
    return mlp_launch(mlp, srcs, n_src, n_rows, row_begin, row_count, tile_rows, out, out_ld, out_idx, act, resid, resid_ld,
                      resid_col0, nullptr, 0, nullptr, 0, stream);
}

extern "C" int g4c_mlp_forward_heads(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                                     float *out, int32_t out_ld, int32_t act,
                                     const float *head_w, int32_t n_heads, float *const *head_out, int32_t head_ld, void *stream) {
    G4C_REQUIRE(n_heads >= 1 && n_heads <= G4C_MAX_HEADS && head_w && head_out, G4C_EINVAL, "g4c_mlp_forward_heads: bad heads (n=%d)", n_heads);
    return mlp_launch(mlp, srcs, n_src, n_rows, 0, n_rows, 324, out, out_ld, nullptr, act, nullptr, 0, 0,
                      head_w, n_heads, head_out, head_ld, stream);
}

extern "C" int g4c_mlp_forward_bf16(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                                    float *out, int32_t out_ld, const int32_t *out_idx, int32_t act,
                                    const float *resid, int32_t resid_ld, int32_t resid_col0, void *stream) {
    return mlp_launch(mlp, srcs, n_src, n_rows, 0, n_rows, 3216, out, out_ld, out_idx, act, resid, resid_ld, resid_col0,
                      nullptr, 0, nullptr, 0, stream);
}

extern "C" int g4c_mlp_forward_heads_bx6(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                                         float *out, int32_t out_ld, int32_t act,
                                         const void *head_w, int32_t n_heads, float *const *head_out, int32_t head_ld, void *stream) {
    G4C_REQUIRE(n_heads >= 1 && n_heads <= G4C_MAX_HEADS && head_w && head_out, G4C_EINVAL, "g4c_mlp_forward_heads_bx6: bad heads (n=%d)", n_heads);
    return mlp_launch(mlp, srcs, n_src, n_rows, 0, n_rows, 3248, out, out_ld, nullptr, act, nullptr, 0, 0,
                      (const float *)head_w, n_heads, head_out, head_ld, stream);
}

extern "C" int g4c_mlp_forward_heads_bf16(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                                          float *out, int32_t out_ld, int32_t act,
                                          const void *head_w, int32_t n_heads, float *const *head_out, int32_t head_ld, void *stream) {
    G4C_REQUIRE(n_heads >= 1 && n_heads <= G4C_MAX_HEADS && head_w && head_out, G4C_EINVAL, "g4c_mlp_forward_heads_bf16: bad heads (n=%d)", n_heads);
    return mlp_launch(mlp, srcs, n_src, n_rows, 0, n_rows, 3216, out, out_ld, nullptr, act, nullptr, 0, 0,
                      (const float *)head_w, n_heads, head_out, head_ld, stream);
}

extern "C" int g4c_mlp_forward_heads_bf16_out(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                                              float *out, int32_t out_ld, int32_t act,
                                              const void *head_w, int32_t n_heads, void *const *head_out, int32_t head_ld, int32_t head_dtype,
                                              void *stream) {
    G4C_REQUIRE(n_heads >= 1 && n_heads <= G4C_MAX_HEADS && head_w && head_out, G4C_EINVAL, "g4c_mlp_forward_heads_bf16_out: bad heads (n=%d)", n_heads);
    G4C_REQUIRE(head_dtype == G4C_DTYPE_F32 || head_dtype == G4C_DTYPE_BF16, G4C_EINVAL, "g4c_mlp_forward_heads_bf16_out: unknown head_dtype %d", head_dtype);
    g_head_dtype = head_dtype;
    const int rc = mlp_launch(mlp, srcs, n_src, n_rows, 0, n_rows, 3216, out, out_ld, nullptr, act, nullptr, 0, 0,
                              (const float *)head_w, n_heads, (float *const *)head_out, head_ld, stream);
    g_head_dtype = 0;
    return rc;
}

extern "C" int g4c_mlp_forward_heads_bf16_rows(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                                               void *out, int32_t out_ld, int32_t out_dtype, int32_t act,
                                               const void *head_w, int32_t n_heads, void *const *head_out, int32_t head_ld, int32_t head_dtype,
                                               void *stream) {
    G4C_REQUIRE(n_heads >= 1 && n_heads <= G4C_MAX_HEADS && head_w && head_out, G4C_EINVAL, "g4c_mlp_forward_heads_bf16_rows: bad heads (n=%d)", n_heads);
    G4C_REQUIRE((head_dtype == G4C_DTYPE_F32 || head_dtype == G4C_DTYPE_BF16) && (out_dtype == G4C_DTYPE_F32 || out_dtype == G4C_DTYPE_BF16), G4C_EINVAL,
                "g4c_mlp_forward_heads_bf16_rows: unknown dtype (out %d, heads %d)", out_dtype, head_dtype);
    g_head_dtype = head_dtype; g_out_dtype = out_dtype;
    const int rc = mlp_launch(mlp, srcs, n_src, n_rows, 0, n_rows, 3216, (float *)out, out_ld, nullptr, act, nullptr, 0, 0,
                              (const float *)head_w, n_heads, (float *const *)head_out, head_ld, stream);
    g_head_dtype = 0; g_out_dtype = 0;
    return rc;
}

extern "C" int g4c_mlp_forward_bf16_out(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                                        void *out, int32_t out_ld, int32_t out_dtype, int32_t act, void *stream) {
    G4C_REQUIRE(out_dtype == G4C_DTYPE_F32 || out_dtype == G4C_DTYPE_BF16, G4C_EINVAL, "g4c_mlp_forward_bf16_out: unknown out_dtype %d", out_dtype);
    g_out_dtype = out_dtype;
    const int rc = mlp_launch(mlp, srcs, n_src, n_rows, 0, n_rows, 3216, (float *)out, out_ld, nullptr, act, nullptr, 0, 0,
                              nullptr, 0, nullptr, 0, stream);
    g_out_dtype = 0;
    return rc;
}

extern "C" int g4c_mlp_forward_bx6_agg(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                                       float *out, int32_t out_ld, int32_t act,
                                       const int32_t *tile_rows, const int32_t *tile_seg, const int32_t *seg_off, int32_t n_tiles,
                                       float *agg, int32_t agg_ld, int32_t agg_mean, void *stream) {
    G4C_REQUIRE(tile_rows && tile_seg && seg_off && agg && n_tiles >= 0 && agg_ld >= NP, G4C_EINVAL, "g4c_mlp_forward_bx6_agg: bad aggregation plan");
    const AggArgs a{tile_rows, tile_seg, seg_off, n_tiles, agg, agg_ld, agg_mean, 0};
    return mlp_launch(mlp, srcs, n_src, n_rows, 0, n_rows, 3248, out, out_ld, nullptr, act, nullptr, 0, 0,
                      nullptr, 0, nullptr, 0, stream, &a);
}

extern "C" int g4c_mp_layer_forward_bx6(const g4c_mlp_t *msg, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                                        float *e_out, int32_t e_ld,
                                        const int32_t *tile_rows, const int32_t *tile_seg, const int32_t *seg_off, int32_t n_tiles,
                                        float *agg, int32_t agg_ld, int32_t agg_mean,
                                        const g4c_mlp_t *upd, const float *v, int32_t v_ld, int32_t act, float *v_out, int32_t v_out_ld,
                                        const void *head_w, int32_t n_heads, float *const *head_out, int32_t head_ld, void *stream) {
    G4C_REQUIRE(tile_rows && tile_seg && seg_off && agg && n_tiles >= 0 && agg_ld >= NP, G4C_EINVAL, "g4c_mp_layer_forward_bx6: bad aggregation plan");
    const AggArgs a{tile_rows, tile_seg, seg_off, n_tiles, agg, agg_ld, agg_mean, 0};
    const NodeArgs nd{upd, v, v_ld, act, v_out, v_out_ld, head_w, n_heads, head_out, head_ld};
    return mlp_launch(msg, srcs, n_src, n_rows, 0, n_rows, 3248, e_out, e_ld ? e_ld : NP, nullptr, G4C_ACT_NONE, nullptr, 0, 0,
                      nullptr, 0, nullptr, 0, stream, &a, nullptr, &nd);
}

extern "C" int g4c_mlp_forward_bf16_agg(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                                        void *out, int32_t out_ld, int32_t out_dtype, int32_t act,
                                        const int32_t *tile_rows, const int32_t *tile_seg, const int32_t *seg_off, int32_t n_tiles,
                                        float *agg, int32_t agg_ld, int32_t agg_mean, void *stream) {
    G4C_REQUIRE(tile_rows && tile_seg && seg_off && agg && n_tiles >= 0 && agg_ld >= NP, G4C_EINVAL, "g4c_mlp_forward_bf16_agg: bad aggregation plan");
    G4C_REQUIRE(out_dtype == G4C_DTYPE_F32 || out_dtype == G4C_DTYPE_BF16 || out_dtype == G4C_DTYPE_BF16_SELU, G4C_EINVAL,
                "g4c_mlp_forward_bf16_agg: unknown out_dtype %d", out_dtype);
    G4C_REQUIRE(out_dtype != G4C_DTYPE_BF16_SELU || act == G4C_ACT_NONE, G4C_EINVAL, "g4c_mlp_forward_bf16_agg: G4C_DTYPE_BF16_SELU with an output activation");
    const AggArgs a{tile_rows, tile_seg, seg_off, n_tiles, agg, agg_ld, agg_mean, out_dtype};
    return mlp_launch(mlp, srcs, n_src, n_rows, 0, n_rows, 3216, (float *)out, out_ld, nullptr, act, nullptr, 0, 0,
                      nullptr, 0, nullptr, 0, stream, &a);
}

extern "C" int g4c_mlp_forward_bx6_save(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                                        float *out, int32_t out_ld, int32_t act, const float *resid, int32_t resid_ld,
                                        int32_t resid_col0, float *const *save, int32_t save_ld, const float *const *mul,
                                        int32_t mul_ld, void *stream) {
    G4C_REQUIRE(save, G4C_EINVAL, "g4c_mlp_forward_bx6_save: null save list");
    const SaveArgs sv{save, save_ld, mul, mul_ld};
    return mlp_launch(mlp, srcs, n_src, n_rows, 0, n_rows, 3248, out, out_ld, nullptr, act, resid, resid_ld, resid_col0,
                      nullptr, 0, nullptr, 0, stream, nullptr, &sv);
}

extern "C" int g4c_mlp_forward_bx6(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                                   float *out, int32_t out_ld, const int32_t *out_idx, int32_t act,
                                   const float *resid, int32_t resid_ld, int32_t resid_col0, void *stream) {
    return mlp_launch(mlp, srcs, n_src, n_rows, 0, n_rows, 3248, out, out_ld, out_idx, act, resid, resid_ld, resid_col0,
                      nullptr, 0, nullptr, 0, stream);
}

static thread_local int g_last_kernel = G4C_KERNEL_NONE;
extern "C" int g4c_mlp_last_kernel(void) { return g_last_kernel; }

// launches of at most this many 32-row tiles run the tile kernel's deep-ring instantiation (Ring6)
static std::atomic<int> g_bx6_deep_tiles{512};
extern "C" int g4c_mlp_small_launch_tiles(int n_tiles) {
    const int prev = g_bx6_deep_tiles.load(std::memory_order_relaxed);
    if (n_tiles >= 0) g_bx6_deep_tiles.store(n_tiles, std::memory_order_relaxed);
    return prev;
}

static int mlp_launch(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                      int64_t row_begin, int64_t row_count, int32_t tile_rows,
                      float *out, int32_t out_ld, const int32_t *out_idx, int32_t act,
                      const float *resid, int32_t resid_ld, int32_t resid_col0,
                      const float *head_w, int32_t n_heads, float *const *head_out, int32_t head_ld, void *stream,
                      const AggArgs *agg, const SaveArgs *save, const NodeArgs *node) {
    g_last_kernel = G4C_KERNEL_NONE;
    const bool force_tiles = (tile_rows == 3249) || (tile_rows == 3217);     // 3248 / 3216 on the 32-row-tile kernel only (tests, A/B)
    if (force_tiles) tile_rows -= 1;
    const bool round1 = (tile_rows == 3216);     // operands rounded to bf16: only the leading plane of the stream is used
    const bool bx6 = (tile_rows == 3248) || round1;   // weights: the three-plane stream of g4c_mlp_pack_layer_bx6
    const bool rs_fmt = round1 && mlp && mlp->w_format == G4C_WFMT_BF16_RS;     // the rounded-bf16 stream in the row-split kernel's k order: that kernel only
    const bool rs2_fmt = round1 && mlp && (mlp->w_format == G4C_WFMT_BF16_RS2 || mlp->w_format == G4C_WFMT_BF16_RS2N);     // ... its update-MLP form
    const bool f16x2 = bx6 && mlp && mlp->w_format == G4C_WFMT_F16X2;     // the stream holds the two-way fp16 split (g4c_mlp_pack_layer_f16x3)
    G4C_REQUIRE(!(f16x2 && round1), G4C_EINVAL, "g4c_mlp_forward_bf16: the weights were packed by g4c_mlp_pack_layer_f16x3 (fp16 planes)");
    G4C_REQUIRE(!mlp || mlp->w_format == 0 || f16x2 || rs_fmt || rs2_fmt, G4C_EINVAL, "g4c_mlp_forward: w_format %d does not match this entry point", mlp->w_format);
    const bool bf16 = bx6;                       // input blocks padded to 128 k
    const int wbytes = bx6 ? 6 : 4;
    if (bf16) tile_rows = 324;
    G4C_REQUIRE(tile_rows == 324, G4C_EINVAL, "g4c_mlp_forward_rows: tile_rows must be 324 (32-row tiles on 4 waves; the one fp32-MFMA kernel), got %d", tile_rows);
    G4C_REQUIRE(row_begin >= 0 && row_count >= 0 && row_begin + row_count <= n_rows && row_begin % 32 == 0, G4C_EINVAL,
                "g4c_mlp_forward_rows: bad row range [%lld, +%lld) of %lld", (long long)row_begin, (long long)row_count, (long long)n_rows);
    G4C_REQUIRE(mlp && srcs, G4C_EINVAL, "g4c_mlp_forward: null pointer");
    G4C_REQUIRE(n_src >= 1 && n_src <= G4C_MAX_SRC, G4C_EUNSUPPORTED, "g4c_mlp_forward: %d sources (max %d)", n_src, G4C_MAX_SRC);
    G4C_REQUIRE(mlp->n_layers >= 1 && mlp->n_layers <= G4C_MAX_LAYERS, G4C_EUNSUPPORTED,
                "g4c_mlp_forward: %d layers (supported 1..%d)", mlp->n_layers, G4C_MAX_LAYERS);
    G4C_REQUIRE(n_rows >= 0 && n_rows < (1LL << 31), G4C_EINVAL, "g4c_mlp_forward: n_rows %lld out of range", (long long)n_rows);
    G4C_REQUIRE(act >= 0 && act <= 2, G4C_EINVAL, "g4c_mlp_forward: bad activation %d", act);
    if (n_rows == 0) return G4C_OK;
    G4C_REQUIRE(out || agg, G4C_EINVAL, "g4c_mlp_forward: null output");
    g4c::DeviceGuard on_device(mlp->w[0]);
    Params p;
    int kp = 0;
    bool all_vec = true;
    int nk = 0;
    p.n_add = 0;
    p.n_nar = 0;
    for (int s = 0; s < n_src; ++s) {
        const g4c_src_t &g = srcs[s];
        G4C_REQUIRE(g.ptr && g.width > 0 && g.ld >= g.col0 + g.width && g.col0 >= 0, G4C_EINVAL,
                    "g4c_mlp_forward: bad source %d (width=%d ld=%d col0=%d)", s, g.width, g.ld, g.col0);
        if (g.additive == 2) {
            G4C_REQUIRE(bx6, G4C_EUNSUPPORTED, "g4c_mlp_forward: narrow sources (additive == 2) need the bf16x6 kernels");
            G4C_REQUIRE(g.width <= G4C_NARROW_MAX && !g.idx && g.pre_act == G4C_ACT_NONE && g.w && ((uintptr_t)g.w & 15) == 0, G4C_EINVAL,
                        "g4c_mlp_forward: bad narrow source %d (width %d <= %d, no index, no pre_act, 16-byte aligned weights)", s,
                        g.width, G4C_NARROW_MAX);
            NarSrc &a = p.nar[p.n_nar++];
            a.ptr = g.ptr + g.col0; a.w = g.w; a.width = g.width; a.ld = g.ld;
            continue;
        }
        if (g.additive) {
            G4C_REQUIRE(g.pre_act == G4C_ACT_NONE && g.width <= NP && (g.dtype == G4C_DTYPE_F32 || g.dtype == G4C_DTYPE_BF16), G4C_EINVAL,
                        "g4c_mlp_forward: bad additive source %d", s);
            AddSrc &a = p.add[p.n_add++];
            a.idx = g.idx; a.width = g.width; a.ld = g.ld; a.bf16 = g.dtype == G4C_DTYPE_BF16;
            if (a.bf16) {
                G4C_REQUIRE(round1 && g.width == NP && g.ld % 4 == 0 && g.col0 % 4 == 0 && (uintptr_t)g.ptr % 8 == 0, G4C_EUNSUPPORTED,
                            "g4c_mlp_forward: bf16 additive rows need the rounded-bf16 mode (g4c_mlp_forward_bf16*) and a 128-wide, 8-byte aligned block");
                a.ptr = reinterpret_cast<const float *>(reinterpret_cast<const __bf16 *>(g.ptr) + g.col0);
            } else {
                a.ptr = g.ptr + g.col0;
            }
            continue;
        }
        G4C_REQUIRE(g.pre_act == G4C_ACT_NONE || g.pre_act == G4C_ACT_SELU, G4C_EUNSUPPORTED,
                    "g4c_mlp_forward: source %d pre_act %d (only NONE / SELU can be applied on load)", s, g.pre_act);
        Src &d = p.src[nk++];
        if (bf16) G4C_REQUIRE(g.width <= NP, G4C_EUNSUPPORTED, "g4c_mlp_forward_bf16: input block %d is %d wide (max 128)", s, g.width);
        d.ptr = g.ptr; d.idx = g.idx; d.width = g.width; d.wpad = bf16 ? NP : (g.width + KC - 1) / KC * KC; d.ld = g.ld; d.col0 = g.col0;
        d.pre_act = g.pre_act;
        d.bf16 = g.dtype == G4C_DTYPE_BF16;
        if (d.bf16)
            G4C_REQUIRE(round1 && g.width == NP && !g.seg_off && g.ld % 4 == 0 && g.col0 % 4 == 0 && (uintptr_t)g.ptr % 8 == 0, G4C_EUNSUPPORTED,
                        "g4c_mlp_forward: bf16 rows need the rounded-bf16 mode (g4c_mlp_forward_bf16*) and a 128-wide, 8-byte aligned block");
        else
            G4C_REQUIRE(g.dtype == G4C_DTYPE_F32, G4C_EINVAL, "g4c_mlp_forward: source %d has unknown dtype %d", s, g.dtype);
        d.seg_off = g.seg_off; d.seg_mean = g.seg_mean; d.seg_perm = g.seg_off ? g.seg_perm : nullptr;
        if (g.seg_off)
            G4C_REQUIRE(bx6 && !g.idx && g.width == NP && g.ld % 4 == 0 && g.col0 % 4 == 0 && (uintptr_t)g.ptr % 16 == 0, G4C_EUNSUPPORTED,
                        "g4c_mlp_forward: aggregation on load needs the bf16x6 kernels and a 128-wide aligned block without gather index");
        d.vec = (g.width % 4 == 0) && (g.ld % 4 == 0) && (g.col0 % 4 == 0) && ((uintptr_t)g.ptr % (d.bf16 ? 8 : 16) == 0);
        all_vec = all_vec && d.vec;
        kp += d.wpad;
    }
    G4C_REQUIRE(nk >= 1 || p.n_nar >= 1, G4C_EINVAL, "g4c_mlp_forward: no input block goes through the weights");
    p.n_src = nk;
    if (nk == 0) p.src[0] = Src{nullptr, nullptr, 0, 0, 0, 0, 1, 0, nullptr, 0, nullptr, 0};
    for (int s = (nk ? nk : 1); s < G4C_MAX_SRC; ++s) p.src[s] = p.src[0];
    for (int s = p.n_nar; s < G4C_MAX_SRC; ++s) p.nar[s] = NarSrc{nullptr, nullptr, 0, 0};
    for (int s = p.n_add; s < G4C_MAX_SRC; ++s) p.add[s] = AddSrc{nullptr, nullptr, 0, 0, 0};
    G4C_REQUIRE(kp == mlp->k_pad[0], G4C_EINVAL, "g4c_mlp_forward: sources give %d padded columns, layer 1 packed for %d", kp, mlp->k_pad[0]);
    p.n_layers = mlp->n_layers;
    p.chunks0 = kp / KC;
    for (int l = 0; l < mlp->n_layers; ++l) {
        G4C_REQUIRE(mlp->n_pad[l] == NP, G4C_EINVAL, "g4c_mlp_forward: layer %d n_pad %d (must be 128)", l, mlp->n_pad[l]);
        if (l > 0) G4C_REQUIRE(mlp->k_pad[l] == NP, G4C_EINVAL, "g4c_mlp_forward: layer %d k_pad %d (must be 128)", l, mlp->k_pad[l]);
        // one contiguous stream: layer l starts where layer l-1 ends
        if (l > 0) G4C_REQUIRE((const char *)mlp->w[l] == (const char *)mlp->w[l - 1] + (size_t)mlp->k_pad[l - 1] * NP * wbytes, G4C_EINVAL,
                               "g4c_mlp_forward: packed layers must be contiguous (layer %d)", l);
        if (l > 0) G4C_REQUIRE((const float *)mlp->b[l] == (const float *)mlp->b[l - 1] + NP, G4C_EINVAL,
                               "g4c_mlp_forward: padded biases must be contiguous (layer %d)", l);
    }
    p.w = (const float *)mlp->w[0];
    p.b = (const float *)mlp->b[0];
    G4C_REQUIRE(p.w && p.b, G4C_EINVAL, "g4c_mlp_forward: null weights");
    p.gamma = mlp->ln_gamma; p.beta = mlp->ln_beta; p.eps = mlp->ln_eps;
    G4C_REQUIRE((p.gamma == nullptr) == (p.beta == nullptr), G4C_EINVAL, "g4c_mlp_forward: LayerNorm needs both gamma and beta");
    p.n_out = mlp->n_out;
    G4C_REQUIRE(p.n_out > 0 && p.n_out <= NP && out_ld >= p.n_out, G4C_EINVAL, "g4c_mlp_forward: n_out=%d out_ld=%d", p.n_out, out_ld);
    p.M = n_rows;
    p.out = out; p.out_ld = out_ld; p.out_idx = out_idx; p.act = act;
    p.out_bf16 = 0;
    if (g_out_dtype) {
        G4C_REQUIRE(round1 && !agg && !resid && !out_idx && p.n_out == NP && (out_ld & 3) == 0 && ((uintptr_t)out & 7) == 0, G4C_EUNSUPPORTED,
                    "g4c_mlp_forward_bf16_out: bf16 output rows need the rounded-bf16 mode, a plain 128-wide output, out_ld a multiple of 4 and an 8-byte aligned out");
        p.out_bf16 = g_out_dtype;
    }
    p.resid = resid; p.resid_ld = resid_ld; p.resid_col0 = resid_col0;
    p.tile_rows = p.tile_seg = p.seg_off = nullptr; p.agg = nullptr; p.agg_ld = 0; p.agg_mean = 0; p.agg_deg = 0; p.agg_bf16 = 0;
    if (agg) {
        G4C_REQUIRE(bx6 && mlp->n_out == NP && !out_idx && !resid, G4C_EUNSUPPORTED, "g4c_mlp_forward_bx6_agg: needs the bf16x6 kernel and a plain 128-wide output");
        p.tile_rows = agg->tile_rows; p.tile_seg = agg->tile_seg; p.seg_off = agg->seg_off;
        p.agg = agg->out; p.agg_ld = agg->out_ld; p.agg_mean = agg->mean & 1; p.agg_deg = (agg->mean >> 8) & 0xff; p.agg_bf16 = (agg->mean >> 16) & 1;
        G4C_REQUIRE((agg->mean >> 17) == 0 && (!p.agg_bf16 || rs_fmt) && p.agg_deg <= 32 && (p.agg_deg == 0 || row_count % p.agg_deg == 0), G4C_EINVAL,
                    "fused aggregation: agg_mean = %d is not 0 / 1 [| G4C_AGG_UNIFORM(k), 1 <= k <= 32, k dividing the %lld rows]", agg->mean,
                    (long long)row_count);
        if (agg->rows_bf16) {
            G4C_REQUIRE(round1 && (!out || ((out_ld & 3) == 0 && ((uintptr_t)out & 7) == 0)), G4C_EUNSUPPORTED,
                        "g4c_mlp_forward_bf16_agg: bf16 output rows need the rounded-bf16 mode, out_ld a multiple of 4 and an 8-byte aligned out");
            p.out_bf16 = agg->rows_bf16;
        }
    }
    for (int l = 0; l < G4C_MAX_LAYERS; ++l) { p.save[l] = nullptr; p.mul[l] = nullptr; }
    p.save_ld = 0; p.mul_ld = 0;
    if (save) {
        G4C_REQUIRE(bx6 && !round1 && !agg && !n_heads && !out_idx && save->ptr && save->ld >= NP && (save->ld & 3) == 0, G4C_EUNSUPPORTED,
                    "g4c_mlp_forward_bx6_save: needs the bf16x6 kernel without heads / aggregation / output index, save_ld >= 128 and a multiple of 4");
        for (int l = 0; l < mlp->n_layers; ++l) {
            G4C_REQUIRE(((uintptr_t)save->ptr[l] & 15) == 0, G4C_EINVAL, "g4c_mlp_forward_bx6_save: save[%d] is not 16-byte aligned", l);
            p.save[l] = save->ptr[l];
        }
        p.save_ld = save->ld;
        if (save->mul) {
            G4C_REQUIRE(save->mul_ld >= NP && (save->mul_ld & 3) == 0, G4C_EINVAL, "g4c_mlp_forward_bx6_save: mul_ld=%d", save->mul_ld);
            for (int l = 0; l + 1 < mlp->n_layers; ++l) {
                G4C_REQUIRE(((uintptr_t)save->mul[l] & 15) == 0, G4C_EINVAL, "g4c_mlp_forward_bx6_save: mul[%d] is not 16-byte aligned", l);
                p.mul[l] = save->mul[l];
            }
            p.mul_ld = save->mul_ld;
        }
    }
    p.range_flag = f16x2 ? mlp->range_flag : nullptr; p.range_slot = mlp->range_slot;
    G4C_REQUIRE(!p.range_flag || p.range_slot >= 0, G4C_EINVAL, "g4c_mlp_forward: negative range_slot");
    p.n_heads = n_heads; p.head_ld = head_ld; p.head_bf16 = 0;
    for (int hd = 0; hd < G4C_MAX_HEADS; ++hd) p.head_out[hd] = hd < n_heads ? head_out[hd] : nullptr;
    if (n_heads && g_head_dtype) {
        G4C_REQUIRE(round1 && (head_ld & 1) == 0, G4C_EUNSUPPORTED, "g4c_mlp_forward_heads_bf16_out: bf16 head rows need the rounded-bf16 mode and an even head_ld");
        for (int hd = 0; hd < n_heads; ++hd)
            G4C_REQUIRE(((uintptr_t)head_out[hd] & 3) == 0, G4C_EINVAL, "g4c_mlp_forward_heads_bf16_out: head output %d is not 4-byte aligned", hd);
        p.head_bf16 = 1;
    }
    if (n_heads) {
        G4C_REQUIRE((head_ld & 3) == 0 || !bx6, G4C_EINVAL, "g4c_mlp_forward_heads: head outputs need a leading dimension that is a multiple of 4");
        G4C_REQUIRE(p.n_out == NP && !resid && !out_idx && head_ld >= NP, G4C_EINVAL,
                    "g4c_mlp_forward_heads: heads need a 128-wide output without residual / output index (n_out=%d)", p.n_out);
        const int last = mlp->n_layers - 1;
        G4C_REQUIRE((const char *)head_w == (const char *)mlp->w[last] + (size_t)mlp->k_pad[last] * NP * wbytes, G4C_EINVAL,
                    "g4c_mlp_forward_heads: head weights must continue the packed stream");
        for (int hd = 0; hd < n_heads; ++hd) G4C_REQUIRE(head_out[hd], G4C_EINVAL, "g4c_mlp_forward_heads: null head output %d", hd);
    }
    hipStream_t st = (hipStream_t)stream;
    if (row_count == 0) return G4C_OK;
    p.row_base = row_begin;
    p.M = row_begin + row_count;          // rows past the range are neither gathered nor stored
    if (node) {
        // one launch per MP layer: the message MLP on the weight-stationary kernel, the node update behind it (mlp_ws.hip, NODE)
        const g4c_mlp_t *u = node->upd;
        G4C_REQUIRE(f16x2 && agg && agg->out && u && u->w_format == G4C_WFMT_F16X2, G4C_EUNSUPPORTED,
                    "g4c_mp_layer_forward_bx6: both MLPs need the f16x3 stream (g4c_mlp_pack_layer_f16x3) and the aggregation plan");
        G4C_REQUIRE(u->n_layers == mlp->n_layers && u->k_pad[0] == 2 * NP && u->n_out == NP && u->w[0] && u->b[0], G4C_EUNSUPPORTED,
                    "g4c_mp_layer_forward_bx6: the node MLP must have the message MLP's depth (%d), two 128-wide input blocks and a 128-wide output",
                    mlp->n_layers);
        for (int l = 0; l < u->n_layers; ++l) {
            G4C_REQUIRE(u->n_pad[l] == NP && (l == 0 || u->k_pad[l] == NP), G4C_EUNSUPPORTED, "g4c_mp_layer_forward_bx6: node layer %d is not 128 wide", l);
            if (l > 0) G4C_REQUIRE((const char *)u->w[l] == (const char *)u->w[l - 1] + (size_t)u->k_pad[l - 1] * NP * 6 &&
                                   (const float *)u->b[l] == (const float *)u->b[l - 1] + NP, G4C_EINVAL,
                                   "g4c_mp_layer_forward_bx6: the node MLP's packed layers / biases must be contiguous (layer %d)", l);
        }
        G4C_REQUIRE((u->ln_gamma == nullptr) == (u->ln_beta == nullptr), G4C_EINVAL, "g4c_mp_layer_forward_bx6: LayerNorm needs both gamma and beta");
        G4C_REQUIRE(!u->ln_gamma || (((uintptr_t)u->ln_gamma & 15) == 0 && ((uintptr_t)u->ln_beta & 15) == 0), G4C_EINVAL,
                    "g4c_mp_layer_forward_bx6: LayerNorm parameters must be 16-byte aligned");
        G4C_REQUIRE(node->v && node->v_out && (node->v_ld & 3) == 0 && node->v_ld >= NP && (node->v_out_ld & 3) == 0 && node->v_out_ld >= NP &&
                    ((uintptr_t)node->v & 15) == 0 && ((uintptr_t)node->v_out & 15) == 0 && (agg->out_ld & 3) == 0 && ((uintptr_t)agg->out & 15) == 0,
                    G4C_EINVAL, "g4c_mp_layer_forward_bx6: v / v_out / agg need 16-byte aligned rows of at least 128 columns");
        G4C_REQUIRE(node->act >= 0 && node->act <= 2 && node->n_heads >= 0 && node->n_heads <= G4C_MAX_HEADS, G4C_EINVAL, "g4c_mp_layer_forward_bx6: bad activation / head count");
        NodeParams q{};
        q.v = node->v; q.v_ld = node->v_ld; q.w = (const float *)u->w[0]; q.b = (const float *)u->b[0];
        q.gamma = u->ln_gamma; q.beta = u->ln_beta; q.eps = u->ln_eps; q.act = node->act;
        q.out = node->v_out; q.out_ld = node->v_out_ld; q.n_heads = node->n_heads; q.head_ld = node->head_ld;
        q.range_flag = u->range_flag; q.range_slot = u->range_slot;
        G4C_REQUIRE(!q.range_flag || q.range_slot >= 0, G4C_EINVAL, "g4c_mp_layer_forward_bx6: negative range_slot");
        for (int hd = 0; hd < G4C_MAX_HEADS; ++hd) q.head_out[hd] = nullptr;
        if (node->n_heads) {
            const int last = u->n_layers - 1;
            G4C_REQUIRE(node->head_w && node->head_out && (const char *)node->head_w == (const char *)u->w[last] + (size_t)u->k_pad[last] * NP * 6, G4C_EINVAL,
                        "g4c_mp_layer_forward_bx6: head weights must continue the node MLP's packed stream");
            G4C_REQUIRE((node->head_ld & 3) == 0 && node->head_ld >= NP, G4C_EINVAL, "g4c_mp_layer_forward_bx6: head_ld must be a multiple of 4 and >= 128");
            for (int hd = 0; hd < node->n_heads; ++hd) {
                G4C_REQUIRE(node->head_out[hd] && ((uintptr_t)node->head_out[hd] & 15) == 0, G4C_EINVAL, "g4c_mp_layer_forward_bx6: bad head output %d", hd);
                q.head_out[hd] = node->head_out[hd];
            }
        }
        G4C_REQUIRE(ws_eligible(p, false, true, false, true, row_count, true), G4C_EUNSUPPORTED,
                    "g4c_mp_layer_forward_bx6: the message launch is outside the weight-stationary kernel's envelope (one 128-wide weighted block, "
                    "two 128-wide additive blocks, two or three 128-wide layers, aligned rows)");
        p.n_tiles = agg->n_tiles;
        if (p.n_tiles == 0) return G4C_OK;
        g_last_kernel = G4C_KERNEL_MLP_WS;
        return ws_launch(p, true, false, st, &q);
    }
    if (rs_fmt) {
        // row-split persistent kernel (mlp_rs.hip): a wave owns 16 rows through all layers, the weights stay in LDS, bf16 rows in its
        // own column order.  A stream in its k order can run on no other kernel: outside its envelope the call fails instead of
        // computing something else.
        G4C_REQUIRE(!save && !node && !force_tiles && rs_eligible(p, agg != nullptr, row_count), G4C_EUNSUPPORTED,
                    "g4c_mlp_forward_bf16: weights packed for the row-split kernel (G4C_WFMT_BF16_RS), launch outside its envelope");
        g_last_kernel = G4C_KERNEL_MLP_RS;
        return rs_launch(p, agg != nullptr, st);
    }
    if (rs2_fmt) {
        G4C_REQUIRE(!save && !node && !force_tiles && !agg && rs2_eligible(p, row_count), G4C_EUNSUPPORTED,
                    "g4c_mlp_forward_bf16: weights packed for the row-split update kernel (G4C_WFMT_BF16_RS2), launch outside its envelope");
        g_last_kernel = G4C_KERNEL_MLP_RS2;
        return rs2_launch(p, mlp->w_format == G4C_WFMT_BF16_RS2N, st);
    }
    if (bx6 && !force_tiles && ws_eligible(p, round1, agg != nullptr, save != nullptr, f16x2, row_count)) {
        // weight-stationary persistent kernel (mlp_ws.hip): pairs of 32-row tiles (whole segments with aggregation), one workgroup per CU
        p.n_tiles = agg ? agg->n_tiles : (int)((row_count + 31) / 32);
        if (p.n_tiles == 0) return G4C_OK;          // (nothing launches: g4c_mlp_last_kernel stays G4C_KERNEL_NONE — ADVICE r04)
        g_last_kernel = G4C_KERNEL_MLP_WS;
        return ws_launch(p, agg != nullptr, round1, st);
    } else if (bx6 && !force_tiles && bx6i_eligible(p, round1, agg != nullptr, save != nullptr, f16x2, row_count)) {
        // dual-tile software-pipelined kernel (mlp_bx6i.hip): pairs of 32-row tiles (whole segments with aggregation)
        p.n_tiles = agg ? agg->n_tiles : (int)((row_count + 31) / 32);
        if (p.n_tiles == 0) return G4C_OK;
        g_last_kernel = G4C_KERNEL_MLP_BX6I;
        return bx6i_launch(p, agg != nullptr, f16x2, st);
    } else if (bx6) {
        bool full = all_vec;
        for (int s2 = 0; s2 < p.n_src; ++s2) full = full && p.src[s2].width == NP;
        for (int a = 0; a < p.n_add; ++a)
            full = full && p.add[a].width == NP && (p.add[a].ld & 3) == 0 && ((uintptr_t)p.add[a].ptr & (p.add[a].bf16 ? 7 : 15)) == 0;
        const dim3 blk(256);
        p.n_tiles = agg ? agg->n_tiles : (int)((row_count + 31) / 32);
        if (p.n_tiles == 0) return G4C_OK;
        g_last_kernel = G4C_KERNEL_MLP_BX6;
        const dim3 grid(p.n_tiles);
        const bool deep = p.n_tiles <= g_bx6_deep_tiles;          // (small launch: whole-block weight ring, see Ring6)
#define G4C_BX6_LAUNCH(RT, SP)                                                                         \
        do {                                                                                           \
            if (deep) {                                                                                \
                if (full) mlp_bx6_kernel<RT, true, true, SP, false, 8><<<grid, blk, 0, st>>>(p);       \
                else if (all_vec) mlp_bx6_kernel<RT, true, false, SP, false, 8><<<grid, blk, 0, st>>>(p); \
                else mlp_bx6_kernel<RT, false, false, SP, false, 8><<<grid, blk, 0, st>>>(p);          \
            } else if (full) mlp_bx6_kernel<RT, true, true, SP><<<grid, blk, 0, st>>>(p);              \
            else if (all_vec) mlp_bx6_kernel<RT, true, false, SP><<<grid, blk, 0, st>>>(p);            \
            else mlp_bx6_kernel<RT, false, false, SP><<<grid, blk, 0, st>>>(p);                        \
        } while (0)
        if (save && f16x2) {
            if (full) mlp_bx6_kernel<1, true, true, 2, true><<<grid, blk, 0, st>>>(p);
            else if (all_vec) mlp_bx6_kernel<1, true, false, 2, true><<<grid, blk, 0, st>>>(p);
            else mlp_bx6_kernel<1, false, false, 2, true><<<grid, blk, 0, st>>>(p);
        } else if (save) {
            if (full) mlp_bx6_kernel<1, true, true, 3, true><<<grid, blk, 0, st>>>(p);
            else if (all_vec) mlp_bx6_kernel<1, true, false, 3, true><<<grid, blk, 0, st>>>(p);
            else mlp_bx6_kernel<1, false, false, 3, true><<<grid, blk, 0, st>>>(p);
        }
        else if (f16x2) G4C_BX6_LAUNCH(1, 2);
        else if (round1) G4C_BX6_LAUNCH(1, 1);
        else G4C_BX6_LAUNCH(1, 3);
#undef G4C_BX6_LAUNCH
    } else {
        p.n_tiles = (int)((row_count + 31) / 32);
        g_last_kernel = G4C_KERNEL_MLP_SPLIT;
        if (all_vec) mlp_split_kernel<4, true><<<dim3(p.n_tiles), dim3(256), 0, st>>>(p);
        else mlp_split_kernel<4, false><<<dim3(p.n_tiles), dim3(256), 0, st>>>(p);
    }
    return g4c::check_launch("g4c_mlp_forward");
}
