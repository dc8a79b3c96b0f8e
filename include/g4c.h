/*
 * g4c.h — C-ABI of libg4c.so: the MI355X (gfx950) implementation of graphs4cfd's
 * message-passing hot path.
 *
 * The reference (mario-linov/graphs4cfd) has no FFI / operator registry: its hot path is
 * Python calling torch + torch_geometric ops.  Each entry point below
 * therefore names the reference *op sequence* (file:line under graphs4cfd/) that it
 * replaces.  All pointers are raw device pointers unless marked "host"; `stream` is a
 * hipStream_t passed as void*; indices are int32 (the reference's int64 index tensors are
 * narrowed once, when the static mesh plan is built).  Every function returns 0 on
 * success or a negative G4C_E* code; g4c_last_error() returns a thread-local message.
 * No entry point synchronises the stream or allocates device memory, so a whole rollout
 * step can be captured in a hipGraph.
 *
 * Reference-side binding: see INTEGRATION.md (ctypes stub for graphs4cfd/nn/blocks.py).
 */
#ifndef G4C_H
#define G4C_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define G4C_OK 0
#define G4C_EINVAL (-1)   /* bad argument (the Python binding raises ValueError) */
#define G4C_ELAUNCH (-2)  /* HIP launch / runtime failure (RuntimeError) */
#define G4C_EUNSUPPORTED (-3) /* shape outside the kernels' envelope (NotImplementedError) */

#define G4C_ACT_NONE 0
#define G4C_ACT_SELU 1
#define G4C_ACT_TANH 2

#define G4C_MAX_SRC 4
#define G4C_MAX_LAYERS 4
#define G4C_MAX_HEADS 2
#define G4C_NARROW_MAX 8

int g4c_version(void);
/* Where a launch on `device_ptr`'s buffers goes: the ordinal of the device that owns it (every launching entry point switches to that
 * device for the call, whatever the caller's current device is) and the compute-unit count the persistent kernels size their grids
 * with there (cached per ordinal).  A multi-GPU job checks both on every rank before it times anything (bench.py partition_check). */
int g4c_device_info(const void *device_ptr, int32_t *device /*host, out*/, int32_t *cu_count /*host, out*/);
const char *g4c_last_error(void);

/* ---------------------------------------------------------------- static mesh plan (host)
 * The reference recomputes topology every step with host syncs: `scatter(dim_size=None)`,
 * `idx.max().item()`, `remove_self_loops`, `coalesce` (nn/blocks.py:45,63-67,109,231).  The
 * plan builders do that work once per mesh.  Host pointers in, host pointers out. */

/* Stable counting sort of `n` keys in [0, n_seg): perm[p] = original position of the p-th
 * entry in key order, off[s]..off[s+1] = its segment.  Replaces the index handling inside
 * torch_geometric.utils.scatter as called at nn/blocks.py:183,231,330,378. */
int g4c_plan_csr(const int64_t *keys /*host*/, int64_t n, int64_t n_seg,
                 int32_t *perm /*host, n*/, int32_t *off /*host, n_seg+1*/);

/* Topology part of pool_edge (nn/blocks.py:51-68): remap endpoints through idx_hr_to_lr,
 * drop intra-cluster edges, merge duplicates; coarse edges ordered by (row, col) exactly as
 * torch_geometric.utils.coalesce orders them.  Outputs: coarse edge_index (2 x n_coarse,
 * row-major), `perm` = surviving fine edges grouped by coarse edge (stable), `off` =
 * n_coarse+1 segment offsets into perm.  Returns n_coarse (>= 0) or a negative error. */
int64_t g4c_plan_pool_edge(const int64_t *idx_hr_to_lr /*host, n_hr*/, int64_t n_hr,
                           const int64_t *edge_index /*host, 2 x n_edges*/, int64_t n_edges,
                           int64_t *coarse_edge_index /*host, 2 x n_edges capacity*/,
                           int32_t *perm /*host, n_edges capacity*/,
                           int32_t *off /*host, n_edges+1 capacity*/,
                           int64_t *n_kept /*host, 1*/);
/* Same, with the order of the coarse edges selectable: target_major = 0 is the function above; target_major = 1 groups the
 * coarse edges by TARGET (sorted by (col, row)) — the order the models use internally: the reference never exposes the
 * coarse edge order and in this order the coarse MP layers need no permutation.  O(n) (two stable
 * counting-sort passes). */
int64_t g4c_plan_pool_edge_ordered(const int64_t *idx_hr_to_lr, int64_t n_hr, const int64_t *edge_index, int64_t n_edges,
                                   int32_t target_major, int64_t *coarse_edge_index, int32_t *perm, int32_t *off,
                                   int64_t *n_kept);

/* ---------------------------------------------------------------- aggregation (HBM-bound)
 * out[s, :] = act( reduce_{p in [off[s], off[s+1])} src_act( src[perm ? perm[p] : p, :] ) )
 * mean = sum / max(count, 1) (empty segments give 0).  Summation runs in p order, so with a
 * stable plan the result equals a sequential scatter_add_.  Replaces
 * torch_geometric.utils.scatter(reduce='sum'|'mean') at nn/blocks.py:183,231,330,378 and the
 * feature part of coalesce(reduce='mean') at nn/blocks.py:67. */
int g4c_segment_reduce(const float *src, int32_t src_ld, const int32_t *perm, const int32_t *off,
                       int32_t n_seg, int32_t width, int32_t mean, int32_t src_act, int32_t act,
                       float *out, int32_t out_ld, void *stream);

/* knn_interpolate (nn/blocks.py:34-48) with fixed, y-sorted segments:
 * out[o(s), :] = sum_p w[p] * x[x_idx[p], :] / sum_p w[p],  o(s) = out_idx ? out_idx[s] : s. */
int g4c_weighted_segment_mean(const float *x, int32_t x_ld, const int32_t *x_idx, const float *w,
                              const int32_t *off, int32_t n_seg, int32_t width,
                              float *out, int32_t out_ld, const int32_t *out_idx, void *stream);

/* ---------------------------------------------------------------- fused MLP (MFMA-bound)
 * One kernel for: gather + concatenate up to 4 sources -> Linear -> (SELU -> Linear)* ->
 * [LayerNorm] -> [SELU|tanh] -> [+ residual] -> store (optionally row-scattered).
 * Replaces `MLP.forward` (nn/blocks.py:117-144) together with the torch.cat / index ops that
 * feed it and the F.selu / torch.tanh / residual add that follow it at every call site
 * (nn/blocks.py:181,185,229,285,328,332,373,380,456; nn/mus_gnn.py:178-218). */
#define G4C_DTYPE_F32 0
#define G4C_DTYPE_BF16 1
#define G4C_DTYPE_BF16_SELU 2   /* g4c_mlp_forward_bf16_agg out_dtype only: rows stored as bf16(SELU(row)) */

typedef struct {
    const float *ptr;   /* [rows, ld] row-major */
    const int32_t *idx; /* NULL: row r of the tile reads row r; else reads row idx[r] */
    int32_t width;      /* columns taken from this source */
    int32_t ld;         /* row stride in floats */
    int32_t col0;       /* first column taken */
    int32_t pre_act;    /* G4C_ACT_*: applied to the values as they are loaded (lets a producer store the
                           un-activated tensor its aggregation needs, nn/blocks.py:181-183 vs nn/mus_gnn.py:182) */
    int32_t additive;   /* 0: a column block of the concatenated input (multiplied by the first layer's weights);
                           1: a term ALREADY multiplied by its block of the first layer's weights, at the row count of
                           the tensor it was gathered from; row idx[r] is added to row r of the first layer's output.
                           Linearity: W1 [e | v[row] | v[col]] = W1e e + (W1r v)[row] + (W1c v)[col], so the node-side
                           products cost N rows instead of E (nn/blocks.py:181, :328, :373).
                           2 (bf16x6 kernels only): a NARROW column block (width <= G4C_NARROW_MAX, idx == NULL, no pre_act)
                           multiplied in fp32 on the vector ALUs by its rows of the first layer's weight, `w`, instead of
                           being padded to a 128-k block of the matrix-pipe stream (the 2..5-wide encoder / DownMP / UpMP
                           inputs: nn/mus_gnn.py:71,176-177, nn/blocks.py:229,285); such blocks are NOT part of the packed
                           stream (k_pad[0] counts the other blocks only). */
    int32_t seg_mean;   /* with seg_off: 1 = mean over the segment (divided by max(count, 1)), 0 = sum */
    const float *w;     /* additive == 2: fp32 [width][128] = W1^T rows of this block (sign folded in), zero padded */
    const int32_t *seg_off; /* bf16x6 kernels, additive == 0, idx == NULL, width 128: row r of this block is the sum / mean of rows
                           [seg_off[r], seg_off[r+1]) of ptr — the aggregation `scatter(e', col, reduce)` (nn/blocks.py:183) done
                           while the node MLP gathers its input, in the order and with the formula of g4c_segment_reduce, instead
                           of a separate pass that writes and re-reads the aggregate.  `pre_act` is then applied to every
                           source row BEFORE it is added (g4c_segment_reduce's src_act). */
    const int32_t *seg_perm; /* with seg_off: NULL = the segment's rows are [seg_off[r], seg_off[r+1]) themselves; else those are
                           positions in seg_perm, which holds the row numbers (pool_edge's fine -> coarse edge plan,
                           nn/blocks.py:67: the pooled coarse edge latents are formed while the first coarse edge MLP gathers them). */
    int32_t dtype;      /* G4C_DTYPE_F32 (0): the rows are fp32.  G4C_DTYPE_BF16 (1), rounded-bf16 mode only (g4c_mlp_forward_bf16*),
                           width 128, no seg_off: the rows are bf16 (ptr is a bf16 pointer, ld / col0 in elements) — additive == 0: the
                           message rows a g4c_mlp_forward_bf16_agg launch stored with out_dtype = G4C_DTYPE_BF16; additive == 1 (round 5):
                           the first-layer products a g4c_mlp_forward_heads_bf16_out / g4c_mlp_forward_bf16_out launch stored as bf16
                           (widened exactly and added to the fp32 accumulators: half the bytes of the largest gather stream). */
} g4c_src_t;

typedef struct {
    int32_t n_layers;                /* number of Linear layers, 1..G4C_MAX_LAYERS */
    int32_t k_pad[G4C_MAX_LAYERS];   /* padded input width of each layer (see g4c_mlp_pack_layer) */
    int32_t n_pad[G4C_MAX_LAYERS];   /* padded output width: 32, 64 or 128 */
    const float *w[G4C_MAX_LAYERS];  /* packed weights, k_pad*n_pad floats each */
    const float *b[G4C_MAX_LAYERS];  /* bias padded with zeros to n_pad */
    const float *ln_gamma;           /* NULL: no LayerNorm */
    const float *ln_beta;
    float ln_eps;
    int32_t n_out;                   /* true output width of the last layer */
    int32_t w_format;                /* 0: the format the entry point names (fp32 stream / three bf16 planes).  G4C_WFMT_F16X2: the
                                        stream was written by g4c_mlp_pack_layer_f16x3 — the g4c_mlp_forward_bx6* entry points then run
                                        their "f16x3" arithmetic (below). */
    int32_t *range_flag;             /* NULL, or a device array of int32: a launch in the f16x3 arithmetic writes 1 into
                                        range_flag[range_slot] when an MLP input or a hidden activation it converted to fp16 reached
                                        the end of fp16's range (|x| >= 65504: the value was CLIPPED there) — never written otherwise,
                                        never cleared by the library.  The reference computes in fp32 (nn/model.py:303-321), so a set
                                        slot means the result may differ from it: rerun with the bf16x6 stream (fp32 exponent range). */
    int32_t range_slot;
} g4c_mlp_t;
#define G4C_WFMT_F16X2 1
#define G4C_WFMT_BF16_RS 3       /* the bf16 stream of g4c_mlp_pack_layer_bx6 (rounded-bf16 mode: its leading plane) with the row-split
                                  * kernel's k order, see "row-split order" at g4c_mlp_forward_bf16 */
#define G4C_WFMT_BF16_RS2 4      /* ... the update MLP of such a layer on the row-split update kernel (csrc/mlp_rs.hip, mlp_rs2_kernel;
                                  * g4c_mlp_forward_bf16_out / g4c_mlp_forward_heads_bf16_rows): TWO weighted 128-wide bf16 blocks
                                  * [aggregate | e] in the row-split order, two layers (256 -> 128 -> 128), LayerNorm, activation none /
                                  * SELU, no heads or two bf16 heads; the columns of every 128-wide block of every layer AND of the heads
                                  * are packed in the row-split order (the heads' ROWS are not permuted: the kernel's stores put them in
                                  * that order); bf16 output rows and head rows come out in the row-split order, fp32 rows in feature order */
#define G4C_WFMT_BF16_RS2N 5     /* the same with the e block's rows in FEATURE order (rows a launch of another kernel stored) */

/* Packs one nn.Linear weight W[n_out, k_in] (row-major, device) for the kernel.  The input
 * dimension is the concatenation of `n_seg` column blocks of widths seg_width[] (each padded
 * to a multiple of 4); seg_negate[s] != 0 folds a sign flip of that block into the weights
 * (UpMP's `-e_hl`, nn/blocks.py:283).  Layout: [k_pad/2][n_pad][2].  k_pad = sum of padded
 * block widths, n_pad = n_out rounded up to 32/64/128.  `packed` must hold k_pad*n_pad floats. */
int g4c_mlp_pack_layer(const float *W, int32_t n_out, int32_t k_in, const int32_t *seg_width /*host*/,
                       const int32_t *seg_negate /*host*/, int32_t n_seg, float *packed,
                       int32_t k_pad, int32_t n_pad, void *stream);

int g4c_mlp_forward(const g4c_mlp_t *mlp /*host*/, const g4c_src_t *srcs /*host*/, int32_t n_src,
                    int64_t n_rows, float *out, int32_t out_ld, const int32_t *out_idx,
                    int32_t act, const float *resid, int32_t resid_ld, int32_t resid_col0,
                    void *stream);

/* g4c_mlp_forward on rows [row_begin, row_begin + row_count) only (row_begin a multiple of 32): lets a caller split one
 * MLP over several launches (profiling, overlap with a halo exchange).  tile_rows names the kernel: 324 = the fp32-MFMA
 * kernel (32-row tiles, the 128 output columns split over 4 waves; the only fp32 variant since round 2). */
int g4c_mlp_forward_rows(const g4c_mlp_t *mlp /*host*/, const g4c_src_t *srcs /*host*/, int32_t n_src,
                         int64_t n_rows, int64_t row_begin, int64_t row_count, int32_t tile_rows,
                         float *out, int32_t out_ld, const int32_t *out_idx, int32_t act,
                         const float *resid, int32_t resid_ld, int32_t resid_col0, void *stream);

/* g4c_mlp_forward plus up to G4C_MAX_HEADS "heads": head_out[h][r, :] = W_h y[r, :], y = the MLP's final 128-wide output
 * row (after LayerNorm / activation), W_h a bias-free 128x128 layer packed with g4c_mlp_pack_layer (k_pad = n_pad = 128)
 * whose packed image continues the MLP's stream: head_w == w[last] + k_pad[last]*128, heads back to back, then the
 * usual chunk of slack.  One launch of the node MLP (nn/blocks.py:185) thereby also emits the two node-side first-layer
 * terms W1[:, H:2H] v', W1[:, 2H:3H] v' of the NEXT GNBlock's edge MLP (nn/blocks.py:181), which that edge MLP
 * gathers as additive sources. */
int g4c_mlp_forward_heads(const g4c_mlp_t *mlp /*host*/, const g4c_src_t *srcs /*host*/, int32_t n_src,
                          int64_t n_rows, float *out, int32_t out_ld, int32_t act,
                          const float *head_w, int32_t n_heads, float *const *head_out /*host*/, int32_t head_ld,
                          void *stream);

/* fp32-accurate variant on the bf16 matrix pipe ("bf16x6", opt-in): both operands of every Linear are split exactly
 * into three bf16 terms (x = h + m + l), the six largest partial products are accumulated in fp32; dropped terms are
 * <= 2^-23 relative.  g4c_mlp_pack_layer_bx6 writes the three-plane weight stream (6 bytes per weight: k_pad * n_pad * 3
 * bf16 per layer, every input block padded to 128 k: k_pad = 128 * n_seg; one 128-k block of slack after the last
 * layer); g4c_mlp_forward_bx6 takes a g4c_mlp_t whose w[] point into that stream. */
int g4c_mlp_pack_layer_bx6(const float *W, int32_t n_out, int32_t k_in, const int32_t *seg_width /*host*/,
                           const int32_t *seg_negate /*host*/, int32_t n_seg, void *packed,
                           int32_t k_pad, int32_t n_pad, void *stream);
int g4c_mlp_forward_bx6(const g4c_mlp_t *mlp /*host*/, const g4c_src_t *srcs /*host*/, int32_t n_src,
                        int64_t n_rows, float *out, int32_t out_ld, const int32_t *out_idx,
                        int32_t act, const float *resid, int32_t resid_ld, int32_t resid_col0, void *stream);
/* "f16x3": the same kernels with a TWO-way fp16 split of both operands, x = h + l * 2^-11 (h = fp16(x) rounded to nearest,
 * l = fp16((x - h) * 2^11): 22 significand bits per operand), and three products per MAC — (Wh, xh) in one fp32 accumulator,
 * (Wh, xl) + (Wl, xh) in a second one folded in with 2^-11 at the end of the layer; the dropped (Wl, xl) term and the operand
 * representation are <= 2^-22 relative each: the result is within the rounding error of an fp32 GEMM of the same shape (measured
 * against fp64: scripts/mlp_accuracy.py, test_mlp_precisions_vs_fp64) at half the matrix-pipe work and two thirds of the operand
 * traffic of the six-product form.  Range: an input or hidden activation with |x| > 65504 is clipped to +-65504 (1 + 2^-11) when it
 * is converted (MODE.FP16_OVFL; no infinities or NaNs are produced); small values lose nothing (the matrix pipe honours fp16
 * subnormals and l keeps x's magnitude).  The bf16 three-way split keeps the whole fp32 range and stays selectable.
 * g4c_mlp_pack_layer_f16x3 writes planes 0 / 1 of the g4c_mlp_pack_layer_bx6 layout (same sizes, plane 2 zero); a g4c_mlp_t over
 * such a stream sets w_format = G4C_WFMT_F16X2 and goes through g4c_mlp_forward_bx6 / _heads_bx6 / _bx6_agg / _bx6_save. */
int g4c_mlp_pack_layer_f16x3(const float *W, int32_t n_out, int32_t k_in, const int32_t *seg_width /*host*/,
                             const int32_t *seg_negate /*host*/, int32_t n_seg, void *packed,
                             int32_t k_pad, int32_t n_pad, void *stream);
/* The dual-tile software-pipelined form of the exact-split kernel (mlp_bx6i.hip: a workgroup alternates between two 32-row tiles,
 * the vector work of one running under the MFMAs of the other, the layer's weights stationary in registers for both) takes the
 * launches of the MP layers' message MLP (one weighted 128-wide block + 0 or 2 additive blocks, three layers, plain 128-wide output
 * rows — through out_idx too — with or without the fused aggregation) in the bf16x6 stream: 0 = never, 1 = launches of at least
 * 400 000 rows (the default mode), 2 = every launch it can take (tests); -1 only queries.  Returns the previous setting. */
int g4c_mlp_bx6i_enable(int on);

/* Weight-stationary persistent form of the same launches for the f16x3 stream (mlp_ws.hip: one 8-wave workgroup per CU, every wave
 * keeps its 16-column slice of all three layers' weights in registers for the whole launch, the loop over tile pairs prefetches the
 * next pair's indices and rows): same envelope and the same per-element arithmetic as g4c_mlp_bx6i_enable's kernel (sums over k in a
 * different association: equal to it within fp32 rounding, the fused aggregation still bit-identical to g4c_segment_reduce of the
 * stored rows).  0 = never, 1 = launches of at least 20 000 rows (the default mode),
 * 2 = every launch it can take (tests); -1 only queries.  Returns the previous setting.  The dual-tile kernel of
 * g4c_mlp_bx6i_enable takes the bf16x6 stream only since round 3. */
int g4c_mlp_ws_enable(int on);

/* Small launches of the tile kernel (at most n_tiles 32-row tiles; default 512 = two workgroups per CU) run an instantiation that
 * keeps a whole 128-k block of weights in flight per wave — the next block's weights are requested while this block multiplies —
 * instead of the two-step ring the chip-filling launches use: with one or two waves per SIMD nothing else hides the L2 round trip.
 * Same arithmetic, bit-identical results.  n_tiles >= 0 sets the limit (0 = never), -1 only queries.  Returns the previous limit. */
int g4c_mlp_small_launch_tiles(int n_tiles);

/* Row-wise LayerNorm (+ activation G4C_ACT_*) over rows of any width: out[r, :] = act((x[r, :] - mean) * rsqrt(var + eps) * gamma + beta),
 * mean / biased variance over the row's `width` columns in two passes, as torch.nn.functional.layer_norm (nn/blocks.py:137-141: the
 * reference's MLP puts a LayerNorm of the output width behind its last Linear layer, whatever that width is).  The fused MLP kernels
 * normalise up to 128 columns in their own epilogue; wider outputs are produced without it and normalised by this launch.
 * gamma / beta may be NULL (1 / 0).  In place (out == x) is allowed. */
int g4c_layer_norm(const float *x, int32_t x_ld, int64_t n_rows, int32_t width, const float *gamma, const float *beta, float eps,
                   int32_t act, float *out, int32_t out_ld, void *stream);

/* Test hook: out[4 i .. 4 i + 3] = a[4 i .. 4 i + 3] / count[i] by the quotient routine the fused aggregation's mean uses (a shared
 * reciprocal + one correction per value, with the division itself as the fallback): must equal the IEEE quotient bit for bit
 * (tests/test_gpu_parity.py::test_mean_div_is_the_ieee_quotient).  count[i] >= 1; a, out 16-byte aligned. */
int g4c_debug_mean_div(const float *a, const int32_t *count, float *out, int64_t n4, void *stream);

/* Which kernel family the calling thread's most recent fused-MLP launch (any g4c_mlp_forward* entry point) ran on — the library
 * picks it per launch (arithmetic, shape, row count), so a profiler-free caller that times launches with events (bench.py's
 * roofline leg) can label them by the kernel that executed instead of by the entry point: G4C_KERNEL_NONE (no launch yet, or the
 * last call launched nothing), _MLP_SPLIT (mlp_split_kernel: fp32 MFMA), _MLP_BX6 (mlp_bx6_kernel: split-operand tile kernel),
 * _MLP_BX6I (mlp_bx6i_kernel: dual-tile), _MLP_WS (mlp_ws_kernel: weight-stationary persistent). */
#define G4C_KERNEL_NONE 0
#define G4C_KERNEL_MLP_SPLIT 1
#define G4C_KERNEL_MLP_BX6 2
#define G4C_KERNEL_MLP_BX6I 3
#define G4C_KERNEL_MLP_WS 4
#define G4C_KERNEL_MLP_RS 5      /* mlp_rs1_kernel: row-split persistent kernel of the rounded-bf16 mode (round 6) */
#define G4C_KERNEL_MLP_RS2 6     /* mlp_rs2_kernel: its update-MLP form (G4C_WFMT_BF16_RS2) */
int g4c_mlp_last_kernel(void);

/* Rounded-bf16 variant (opt-in only; BASELINE config 3 "bf16 edge-MLP MFMA"): the same stream and kernel structure, but
 * only the LEADING bf16 term of every operand is used (one product per multiply-add): weights and the activations
 * entering each Linear are rounded to bf16, accumulation / bias / SELU / LayerNorm / additive sources / residual stay
 * fp32.  Expected deviation from the fp32 result: ~1e-2 on LayerNorm-scale outputs. */
int g4c_mlp_forward_bf16(const g4c_mlp_t *mlp /*host*/, const g4c_src_t *srcs /*host*/, int32_t n_src,
                         int64_t n_rows, float *out, int32_t out_ld, const int32_t *out_idx,
                         int32_t act, const float *resid, int32_t resid_ld, int32_t resid_col0, void *stream);
/* Edge MLP + aggregation in one launch.  g4c_plan_tiles (host) cuts the CSR-ordered rows into tiles of whole segments
 * (<= max_rows rows; -1 if a segment is longer): tile t = segments [tile_seg[t], tile_seg[t+1]) = rows
 * [tile_rows[t], tile_rows[t+1]).  g4c_mlp_forward_bx6_agg runs the MLP on those tiles (max_rows must be 32) and, from
 * the on-chip copy of each tile's output rows, writes agg[s, :] = sum or mean (agg_mean) of the rows of segment s —
 * same order and formula as g4c_segment_reduce, i.e. the `scatter(e', col, reduce)` of nn/blocks.py:183 without
 * re-reading e' from HBM.  tile_rows / tile_seg / seg_off are device int32 arrays.  out == NULL: the rows themselves are
 * not stored, only their aggregate — the last MP layer of a level, whose edge output the reference discards
 * (nn/mus_gnn.py:199-200,211-212).
 * agg_mean: 0 sum, 1 mean; OR-ed with G4C_AGG_UNIFORM(k) the caller promises that EVERY segment has exactly k rows (1 <= k <= 32:
 * the in-degree of a kNN mesh) — the weight-stationary kernel then aggregates with static addressing instead of reading segment
 * offsets (same sums in the same order; the mean as the correctly rounded quotient by Markstein's correction, which differs from
 * the IEEE division only below 2^-100). */
#define G4C_AGG_UNIFORM(k) ((int32_t)(k) << 8)
/* OR-ed into agg_mean (g4c_mlp_forward_bf16_agg with a G4C_WFMT_BF16_RS stream only): `agg` points to bf16 rows (agg_ld in elements, a
 * multiple of 8) and the aggregate is stored rounded to bf16, its 128 values in the row-split order — what the layer's update MLP, its
 * one reader in the rounded-bf16 mode, rounds it to on load anyway. */
#define G4C_AGG_OUT_BF16 ((int32_t)1 << 16)
int64_t g4c_plan_tiles(const int32_t *off /*host*/, int32_t n_seg, int32_t max_rows, int32_t *tile_rows /*host, out*/,
                       int32_t *tile_seg /*host, out*/, int64_t capacity);
int g4c_mlp_forward_bx6_agg(const g4c_mlp_t *mlp /*host*/, const g4c_src_t *srcs /*host*/, int32_t n_src,
                            int64_t n_rows, float *out, int32_t out_ld, int32_t act,
                            const int32_t *tile_rows, const int32_t *tile_seg, const int32_t *seg_off, int32_t n_tiles,
                            float *agg, int32_t agg_ld, int32_t agg_mean, void *stream);
/* the same for the rounded-bf16 mode (g4c_mlp_forward_bf16: leading plane of the stream only).  out_dtype = G4C_DTYPE_BF16 stores
 * the rows as bf16 (out is then a bf16 pointer, out_ld in elements): in this mode the consumer of the rows rounds them to bf16 when
 * it loads them anyway, and the launch is HBM-bound on exactly these rows (REMuS-GNN's angle latents, BASELINE config 3); the
 * aggregate is computed from the fp32 tile and stays fp32.  out_dtype = G4C_DTYPE_BF16_SELU stores bf16(SELU(row)) — the activation
 * the model applies to the messages after the aggregation (nn/blocks.py:331-333, nn/remus_gnn.py:150-190), which their one reader
 * would otherwise apply on load (g4c_src_t.pre_act) BEFORE rounding to bf16: stored this way the reader gets bit for bit the operand
 * it would have formed from fp32 rows (one rounding, after the activation), the aggregate still sees the un-activated fp32 rows. */
/* Row-split order (round 6; g4c_mlp_t.w_format = G4C_WFMT_BF16_RS, with g4c_mlp_forward_bf16 / _bf16_agg only).  The message launch
 * of an MP layer whose receivers all have the same in-degree k, 4 <= k <= 8 (agg_mean | G4C_AGG_UNIFORM(k); REMuS-GNN: every edge of
 * a k-nearest-neighbour graph receives k angles) — ONE weighted 128-wide direct block (fp32, optional SELU on load, or bf16), two
 * additive 128-wide blocks through indices, two or three 128-wide layers, LayerNorm, no output activation — runs on a kernel in which a
 * wave owns 16 rows through all layers (csrc/mlp_rs.hip).  Its caller packs the weights with the COLUMNS of every layer permuted:
 * position 32 j + 8 g + 4 h + e (j < 4, g < 4, h < 2, e < 4) of the 128 input columns takes column 32 j + 16 h + 4 g + e, and every
 * bf16 row of such a launch — the bf16 weighted block, bf16 additive tables, the rows it stores with out_dtype G4C_DTYPE_BF16 /
 * _BF16_SELU — has its 128 values in the SAME order (position -> feature): producers of the additive tables permute the ROWS of the
 * weight that makes them (g4c_mlp_forward_bf16_out / _heads_bf16_out), readers other than this kernel must undo the order.  fp32
 * rows, the bias / LayerNorm vectors and the aggregate are in feature order.  The aggregate is a fixed-order segmented scan, not the
 * sequential order of g4c_segment_reduce: last-bit differences.  A launch outside this envelope fails with G4C_EUNSUPPORTED. */
int g4c_mlp_forward_bf16_agg(const g4c_mlp_t *mlp /*host*/, const g4c_src_t *srcs /*host*/, int32_t n_src,
                             int64_t n_rows, void *out, int32_t out_ld, int32_t out_dtype, int32_t act,
                             const int32_t *tile_rows, const int32_t *tile_seg, const int32_t *seg_off, int32_t n_tiles,
                             float *agg, int32_t agg_ld, int32_t agg_mean, void *stream);
/* g4c_mlp_forward_heads for the bf16x6 stream (heads packed with g4c_mlp_pack_layer_bx6 right after the last layer) */
int g4c_mlp_forward_heads_bx6(const g4c_mlp_t *mlp /*host*/, const g4c_src_t *srcs /*host*/, int32_t n_src,
                              int64_t n_rows, float *out, int32_t out_ld, int32_t act,
                              const void *head_w, int32_t n_heads, float *const *head_out /*host*/, int32_t head_ld,
                              void *stream);

/* the same in the rounded-bf16 mode (stream of g4c_mlp_forward_bf16): head j = bf16(act(y)) x bf16(head weights j), fp32 accumulate —
 * the operands the consumer's own first layer would form from the gathered rows of y, so its hoisted first layer
 * (g4c_src_t.additive) adds the same products in another order. */
int g4c_mlp_forward_heads_bf16(const g4c_mlp_t *mlp /*host*/, const g4c_src_t *srcs /*host*/, int32_t n_src,
                               int64_t n_rows, float *out, int32_t out_ld, int32_t act,
                               const void *head_w, int32_t n_heads, float *const *head_out /*host*/, int32_t head_ld,
                               void *stream);

/* Round 5: ONE launch per MP layer — `GNBlock.forward` (nn/blocks.py:175-186): e' = edge_mlp([e | v[row] | v[col]]), aggregation of e'
 * per target, v' = act(node_mlp([aggr | v])) — for the f16x3 stream.  The message part is g4c_mlp_forward_bx6_agg's launch on the
 * weight-stationary kernel (one 128-wide weighted block `e`, the two hoisted node-side products as additive sources, two or three
 * 128-wide layers; e_out may be NULL: the rows are then not stored); a persistent workgroup's tile pairs cover a contiguous range of
 * targets, so once they are done it runs the node MLP `upd` (same depth; input blocks [aggregate | v], both 128 wide) on exactly those
 * targets — the aggregates go through `agg` ([n_targets, agg_ld] scratch, L2-resident) — and stores v' = act(LayerNorm(...)) to v_out
 * and, with n_heads > 0, the heads of g4c_mlp_forward_heads_bx6 (the NEXT layer's node-side products).  Same arithmetic per element as
 * the two separate launches (g4c_mlp_forward_bx6_agg, g4c_mlp_forward_heads_bx6); sums over k are associated as in the
 * weight-stationary kernel.  Small and medium levels of a multi-scale model are bound by the dependent chain inside each launch, not
 * by throughput: this halves the number of chains per MP layer. */
int g4c_mp_layer_forward_bx6(const g4c_mlp_t *msg /*host*/, const g4c_src_t *srcs /*host*/, int32_t n_src, int64_t n_rows,
                             float *e_out, int32_t e_ld,
                             const int32_t *tile_rows, const int32_t *tile_seg, const int32_t *seg_off, int32_t n_tiles,
                             float *agg, int32_t agg_ld, int32_t agg_mean,
                             const g4c_mlp_t *upd /*host*/, const float *v, int32_t v_ld, int32_t act, float *v_out, int32_t v_out_ld,
                             const void *head_w, int32_t n_heads, float *const *head_out /*host*/, int32_t head_ld, void *stream);

/* Round 5, rounded-bf16 mode (BASELINE config 3): the first-layer products of a hoisted message MLP stored as bf16.  A product row
 * is a pre-activation term of a layer whose operands are already rounded to bf16 (relative 2^-9 each); stored as bf16 it is rounded
 * once more at the same relative size, and REMuS-GNN's level-1 angle launch — 2.5 M rows, two gathered product rows each, the
 * launch's largest stream — reads half the bytes (nn/blocks.py:322-333: the `torch.cat` of gathered sender / receiver rows it
 * replaces).  g4c_mlp_forward_heads_bf16_out = g4c_mlp_forward_heads_bf16 with head_dtype: G4C_DTYPE_BF16 stores the head rows as
 * bf16 (head_out are then bf16 pointers, head_ld in elements, even).  g4c_mlp_forward_bf16_out = g4c_mlp_forward_bf16 (no output
 * index / residual) with out_dtype: G4C_DTYPE_BF16 stores the output rows as bf16 (a 128-wide output, out_ld a multiple of 4, out
 * 8-byte aligned) — the launch that multiplies the node-side inputs by their block of the first layer when no producer emitted them. */
int g4c_mlp_forward_heads_bf16_out(const g4c_mlp_t *mlp /*host*/, const g4c_src_t *srcs /*host*/, int32_t n_src,
                                   int64_t n_rows, float *out, int32_t out_ld, int32_t act,
                                   const void *head_w, int32_t n_heads, void *const *head_out /*host*/, int32_t head_ld,
                                   int32_t head_dtype, void *stream);
/* Round 6: g4c_mlp_forward_heads_bf16_out with out_dtype as well — G4C_DTYPE_BF16 stores the launch's own output rows (after LayerNorm
 * and the activation) as bf16 (out is then a bf16 pointer, out_ld in elements, a multiple of 4): the edge latents between consecutive
 * EdgeMPs of a level, whose only reader is the next update MLP — which rounds them to bf16 on load (same operand, half the bytes in
 * both launches).  The heads are computed from the fp32 rows either way. */
int g4c_mlp_forward_heads_bf16_rows(const g4c_mlp_t *mlp /*host*/, const g4c_src_t *srcs /*host*/, int32_t n_src,
                                    int64_t n_rows, void *out, int32_t out_ld, int32_t out_dtype, int32_t act,
                                    const void *head_w, int32_t n_heads, void *const *head_out /*host*/, int32_t head_ld,
                                    int32_t head_dtype, void *stream);
int g4c_mlp_forward_bf16_out(const g4c_mlp_t *mlp /*host*/, const g4c_src_t *srcs /*host*/, int32_t n_src,
                             int64_t n_rows, void *out, int32_t out_ld, int32_t out_dtype, int32_t act, void *stream);

/* ---------------------------------------------------------------- REMuS helpers (HBM-bound)
 * out[e, f] = v[node[e], 2f]*U[e,0] + v[node[e], 2f+1]*U[e,1]
 * (nn/remus_gnn.py:124-126, nn/blocks.py:454). node == NULL reads row e. */
int g4c_project_to_edges(const float *v, int32_t v_ld, const int32_t *node, const float *unit,
                         int64_t n_edges, int32_t n_feat, float *out, int32_t out_ld, void *stream);

/* edgeScalarToNodeVector with edgeUnitVectorInverse (nn/blocks.py:88-114):
 * out[n, 2f+c] = sum_j unit_inv[n, c, j] * e[n*k + j, f]. */
int g4c_edge_scalar_to_node_vector(const float *e, int32_t e_ld, const float *unit_inv, int32_t k,
                                   int64_t n_nodes, int32_t n_feat, float *out, int32_t out_ld,
                                   void *stream);

/* ---------------------------------------------------------------- pre-processing
 * The neighbour search of connect_knn (transforms/connect.py:9-72; torch_cluster.knn / k-d tree on the host in the
 * reference): for every point its k nearest OTHER points, ascending distance, exact.  The caller bins the cloud into a
 * uniform grid of `cell_size` cells starting at `origin` (cell id = x + n_cells[0]*(y + n_cells[1]*z)) and hands over the
 * points in cell-sorted order: pos_sorted [n, dim] fp32, cell_sorted [n] their cell ids, order [n] their original
 * indices, cell_start [prod(n_cells)+1] the first sorted point of each cell (all device, int32).  n_cells and origin are
 * host arrays of 3.  out [n, k] int64 (device): row = original index of the query, entries = original indices. */
int g4c_knn_grid(const float *pos_sorted, const int32_t *cell_sorted, const int32_t *order,
                 const int32_t *cell_start, int64_t n, int32_t dim, const int32_t *n_cells, const float *origin,
                 float cell_size, int32_t k, int64_t *out, void *stream);

/* The same search for m separate query points (get_knn_interpolate_weights, transforms/interpolate.py:110-131: the k
 * nearest nodes of pos_x for every node of pos_y): q_pos [m, dim] fp32 and q_cell [m] (each query's cell in the cloud's
 * grid, coordinates clamped into it) in any order; no point is excluded; out [m, k] int64, row = query. */
int g4c_knn_grid_query(const float *pos_sorted, const int32_t *order, const int32_t *cell_start, int64_t n, int32_t dim,
                       const int32_t *n_cells, const float *origin, float cell_size, const float *q_pos,
                       const int32_t *q_cell, int64_t m, int32_t k, int64_t *out, void *stream);

/* ---------------------------------------------------------------- rollout (nn/model.py:303-327)
 * One step's bookkeeping without host involvement: t = *step;
 * outputs[:, nf*t : nf*(t+1)] = pred;  field = roll(field, -nf, dim=1); field[:, -nf:] = pred;
 * then step[0] = t + 1, written by the last workgroup of the launch to finish.  `step` points to TWO int32: [0] the step index, [1] a
 * ticket counter the launch uses for that (zero before the first launch; it leaves it zero).
 * out_ld == 0: `outputs` is step-major, [steps][n_nodes][nf] contiguous — outputs[t] = pred (one contiguous block per step; the
 * caller transposes once at the end of the rollout). */
int g4c_rollout_advance(float *field, int32_t field_cols, const float *pred, int32_t nf,
                        float *outputs, int32_t out_ld, int32_t *step, int64_t n_nodes, void *stream);

/* out[r, c] = a[r, a_col0 + c] + b[r, c]: the residual time step `field[:, -nf:] + output`
 * (nn/remus_gnn.py:199; the MuS-GNN decoder fuses it into g4c_mlp_forward's epilogue instead). */
int g4c_add_cols(const float *a, int32_t a_ld, int32_t a_col0, const float *b, int32_t b_ld,
                 float *out, int32_t out_ld, int32_t width, int64_t n_rows, void *stream);

/* dst[r, dcol0 : dcol0+width] = src[r, scol0 : scol0+width]  (torch.cat of the narrow node inputs,
 * nn/mus_gnn.py:71; also used to assemble halo send buffers when idx != NULL: reads src[idx[r]]). */
/* x[i] = act(x[i]) in place, n contiguous floats (F.selu / torch.tanh on a block output). */
int g4c_activation_inplace(float *x, int64_t n, int32_t act, void *stream);

int g4c_copy_cols(const float *src, int32_t src_ld, int32_t scol0, const int32_t *idx,
                  float *dst, int32_t dst_ld, int32_t dcol0, int32_t width, int64_t n_rows, void *stream);

/* g4c_mlp_forward_bx6 for the training forward: in addition to the output, save[l] (l < n_layers; an entry may be NULL) receives
 * the rows layer l produces — SELU(hidden) for l < n_layers-1, the pre-LayerNorm rows for the last layer — as fp32
 * [n_rows, 128] (leading dimension save_ld >= 128, multiple of 4; columns past the layer's width are padding).  With them
 * the backward pass of the block recomputes nothing (autograd.py).  No output index / heads / fused aggregation.
 * `mul` (NULL for the forward): the same launch as the BACKWARD chain of a block.  With mul[l] != NULL, hidden layer l's result
 * is multiplied by the SELU slope of the rows mul[l] holds (SELU outputs, [n_rows, 128], leading dimension mul_ld) instead of
 * bias + SELU.  Packing the transposed weights last layer first (zero biases) and passing the kept activations as `mul` gives
 *   g_{k-1} = (g_k W_k) * selu'(a_{k-1})   for every hidden layer, each g written through save[], and the input gradient as
 * the launch's output — one launch instead of a product + an elementwise pass per layer. */
int g4c_mlp_forward_bx6_save(const g4c_mlp_t *mlp, const g4c_src_t *srcs, int32_t n_src, int64_t n_rows,
                             float *out, int32_t out_ld, int32_t act, const float *resid, int32_t resid_ld, int32_t resid_col0,
                             float *const *save, int32_t save_ld, const float *const *mul, int32_t mul_ld, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Training path: the backward pass of the fused blocks.  The reference differentiates
 * cat / index / nn.Linear / SELU / LayerNorm / scatter with torch autograd (GNN.fit, nn/model.py:152-301); here the forward
 * is the fused launch above, the backward recomputes the block's activations (nothing but the block's inputs and output is
 * kept between the passes) with rocBLAS for the plain GEMMs and these kernels for everything else.  All reductions run
 * in a fixed order: gradients are bit-reproducible.
 * ------------------------------------------------------------------------------------------------------------------ */

/* dst[r, dcol0 : dcol0+width] (+)= (negate ? -1 : 1) * pre_act(src[idx ? idx[r] : r, scol0 : scol0+width]): one column block
 * of the concatenated MLP input (nn/blocks.py:181,185,229,285), recomputed for the backward pass; `accumulate` != 0 adds
 * into dst (the recomputed first layer: pre-multiplied node-side terms gathered through their index). */
int g4c_train_gather(const float *src, int32_t src_ld, int32_t scol0, const int32_t *idx, int32_t pre_act, int32_t negate,
                     float *dst, int32_t dst_ld, int32_t dcol0, int32_t width, int64_t n_rows, int32_t accumulate, void *stream);

/* dz = dy * act'(.)  — `ref` holds the activation's OUTPUT (from_input = 0) or its INPUT (from_input = 1).  dz may alias dy. */
int g4c_act_grad(const float *dy, int32_t dy_ld, const float *ref, int32_t ref_ld, int32_t from_input, int32_t act,
                 float *dz, int32_t dz_ld, int32_t width, int64_t n_rows, void *stream);

/* LayerNorm backward (nn/blocks.py:141, eps 1e-5, affine): dz from the pre-norm rows z, gamma and dy; `partial` receives
 * g4c_layernorm_grad_partials(n_rows) rows of [dgamma(width) | dbeta(width)] partial sums — add them with g4c_colsum. */
int32_t g4c_layernorm_grad_partials(int64_t n_rows);
int g4c_layernorm_grad(const float *z, int32_t z_ld, const float *gamma, const float *dy, int32_t dy_ld, float *dz,
                       int32_t dz_ld, float *partial, int32_t width, int64_t n_rows, float eps, void *stream);

/* out[c] = sum_r x[r, c] (bias gradients; the LayerNorm partials).  `scratch`: g4c_colsum_partials(n_rows) * width floats. */
int32_t g4c_colsum_partials(int64_t n_rows);
int g4c_colsum(const float *x, int32_t ld, int32_t width, int64_t n_rows, float *scratch, float *out, void *stream);

/* Weight / bias gradient of one nn.Linear with 128 inputs and 128 outputs (every hidden layer of every published arch):
 * out[0 : 128*128] = dW[n, k] = sum_r g[r, n] a[r, k] (row-major [128, 128]), and, if with_bias, out[128*128 : +128] =
 * db[n] = sum_r g[r, n].  g = dL/d(layer output) [n_rows, 128], a = the layer's input rows [n_rows, 128] (a 128-column window
 * of a wider tensor is fine: a_ld).  One pass over g and a (HBM-bound), fp32 MFMA, partial tiles per workgroup added in a
 * fixed order.  `scratch`: g4c_weight_grad_scratch_floats(n_rows) floats; `out`: 128*128 + 128 floats. */
int32_t g4c_weight_grad_partials(int64_t n_rows);
int64_t g4c_weight_grad_scratch_floats(int64_t n_rows);
int g4c_weight_grad(const float *g, int32_t g_ld, const float *a, int32_t a_ld, int64_t n_rows, float *scratch, float *out,
                    int32_t with_bias, void *stream);

/* Adjoint of g4c_segment_reduce: dsrc[perm ? perm[p] : p] = dout[s] (/ max(count_s, 1) if mean) for p in segment s.
 * Rows of dsrc that belong to no segment are left untouched (zero them first when perm is not a full permutation). */
int g4c_segment_broadcast(const float *dout, int32_t dout_ld, const int32_t *off, const int32_t *perm, int32_t n_seg,
                          int32_t width, int32_t mean, float *dsrc, int32_t dsrc_ld, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* G4C_H */
