// Host-side builders of the static mesh plan (run once per mesh, not per step).
//
// The reference recomputes all of this inside every forward, with device->host syncs:
//   * PyG scatter(dim_size=None) -> int(index.max())+1           (call at nn/blocks.py:231)
//   * pool_edge: idx.max().item(), remove_self_loops (mask compaction), coalesce (sort+unique)
//                                                                (nn/blocks.py:63-67)
// Topology is loop-invariant in a rollout, so it is hoisted here.
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

#include "g4c_common.h"

extern "C" int g4c_plan_csr(const int64_t *keys, int64_t n, int64_t n_seg, int32_t *perm, int32_t *off) {
    G4C_REQUIRE(n >= 0 && n_seg >= 0 && n < (1LL << 31) && n_seg < (1LL << 31), G4C_EINVAL,
                "g4c_plan_csr: sizes out of int32 range (n=%lld n_seg=%lld)", (long long)n, (long long)n_seg);
    G4C_REQUIRE((keys || n == 0) && off && (perm || n == 0), G4C_EINVAL, "g4c_plan_csr: null pointer");
    std::fill(off, off + n_seg + 1, 0);
    for (int64_t p = 0; p < n; ++p) {
        const int64_t k = keys[p];
        G4C_REQUIRE(k >= 0 && k < n_seg, G4C_EINVAL, "g4c_plan_csr: key %lld at %lld outside [0, %lld)",
                    (long long)k, (long long)p, (long long)n_seg);
        off[k + 1]++;
    }
    for (int64_t s = 0; s < n_seg; ++s) off[s + 1] += off[s];
    std::vector<int32_t> cur(off, off + n_seg);
    for (int64_t p = 0; p < n; ++p) perm[cur[keys[p]]++] = (int32_t)p;   // stable
    return G4C_OK;
}

// Tiles of whole segments: tile t covers segments [tile_seg[t], tile_seg[t+1]) = rows [tile_rows[t], tile_rows[t+1]), at
// most max_rows rows (greedy, in order; empty segments ride along).  Lets the edge-MLP kernel reduce the messages of the
// targets it has just computed (g4c_mlp_forward_bx6_agg).  Returns the tile count, -1 if a segment exceeds max_rows (then
// the caller keeps the separate g4c_segment_reduce launch), or a negative G4C_E* code.
extern "C" int64_t g4c_plan_tiles(const int32_t *off, int32_t n_seg, int32_t max_rows, int32_t *tile_rows, int32_t *tile_seg,
                                  int64_t capacity) {
    G4C_REQUIRE(off && tile_rows && tile_seg && n_seg >= 0 && max_rows > 0 && capacity >= 1, G4C_EINVAL, "g4c_plan_tiles: bad arguments");
    int64_t nt = 0;
    int32_t s = 0;
    tile_rows[0] = off[0];
    tile_seg[0] = 0;
    while (s < n_seg) {
        const int32_t r0 = off[s];
        int32_t e = s;
        while (e < n_seg && off[e + 1] - r0 <= max_rows) ++e;
        if (e == s) return -1;                          // one segment alone is larger than a tile
        G4C_REQUIRE(nt + 1 < capacity, G4C_EINVAL, "g4c_plan_tiles: capacity %lld too small", (long long)capacity);
        ++nt;
        tile_rows[nt] = off[e];
        tile_seg[nt] = e;
        s = e;
    }
    return nt;
}

extern "C" int64_t g4c_plan_pool_edge_ordered(const int64_t *idx_hr_to_lr, int64_t n_hr, const int64_t *edge_index,
                                              int64_t n_edges, int32_t target_major, int64_t *coarse_edge_index, int32_t *perm,
                                              int32_t *off, int64_t *n_kept) {
    G4C_REQUIRE(n_hr >= 0 && n_edges >= 0 && n_edges < (1LL << 31), G4C_EINVAL, "g4c_plan_pool_edge: bad sizes");
    G4C_REQUIRE((idx_hr_to_lr || n_hr == 0) && (edge_index || n_edges == 0) && off && n_kept, G4C_EINVAL,
                "g4c_plan_pool_edge: null pointer");
    int64_t n_lr = 0;
    for (int64_t i = 0; i < n_hr; ++i) {
        // (-1 is the 'empty' value of the reference's mask2idx tables: a fine node without a coarse node cannot be pooled)
        G4C_REQUIRE(idx_hr_to_lr[i] >= 0, G4C_EINVAL, "g4c_plan_pool_edge: idx_hr_to_lr[%lld] = %lld is negative", (long long)i,
                    (long long)idx_hr_to_lr[i]);
        n_lr = std::max(n_lr, idx_hr_to_lr[i] + 1);
    }
    G4C_REQUIRE(n_lr < (1LL << 31), G4C_EINVAL, "g4c_plan_pool_edge: %lld coarse nodes do not fit int32", (long long)n_lr);
    // surviving fine edges with their coarse endpoints (remove_self_loops), in fine-edge order
    std::vector<int32_t> cr, cc, id;
    cr.reserve((size_t)n_edges); cc.reserve((size_t)n_edges); id.reserve((size_t)n_edges);
    const int64_t *row = edge_index, *col = edge_index + n_edges;
    for (int64_t e = 0; e < n_edges; ++e) {
        G4C_REQUIRE(row[e] >= 0 && row[e] < n_hr && col[e] >= 0 && col[e] < n_hr, G4C_EINVAL,
                    "g4c_plan_pool_edge: edge %lld endpoint outside [0, %lld)", (long long)e, (long long)n_hr);
        const int64_t r = idx_hr_to_lr[row[e]], c = idx_hr_to_lr[col[e]];
        if (r != c) { cr.push_back((int32_t)r); cc.push_back((int32_t)c); id.push_back((int32_t)e); }
    }
    const int64_t kept = (int64_t)id.size();
    *n_kept = kept;
    // stable two-pass counting sort (minor key, then major key): coalesce order = (row, col), target-major = (col, row);
    // ties keep fine-edge order, as the stable comparison sort on row * n_lr + col did
    const std::vector<int32_t> &major = target_major ? cc : cr, &minor = target_major ? cr : cc;
    std::vector<int32_t> order((size_t)kept), tmp((size_t)kept);
    std::vector<int64_t> cnt((size_t)n_lr + 1);
    auto pass = [&](const std::vector<int32_t> &key, const int32_t *in, int32_t *out) {
        std::fill(cnt.begin(), cnt.end(), 0);
        for (int64_t p = 0; p < kept; ++p) ++cnt[(size_t)key[(size_t)in[p]] + 1];
        for (int64_t b = 0; b < n_lr; ++b) cnt[(size_t)b + 1] += cnt[(size_t)b];
        for (int64_t p = 0; p < kept; ++p) out[cnt[(size_t)key[(size_t)in[p]]]++] = in[p];
    };
    for (int64_t p = 0; p < kept; ++p) tmp[(size_t)p] = (int32_t)p;
    pass(minor, tmp.data(), order.data());
    pass(major, order.data(), tmp.data());          // tmp: positions (into cr / cc / id) in final order
    int64_t n_coarse = 0;
    for (int64_t p = 0; p < kept; ++p) {
        const int32_t q = tmp[(size_t)p];
        if (p == 0 || cr[(size_t)q] != cr[(size_t)tmp[(size_t)p - 1]] || cc[(size_t)q] != cc[(size_t)tmp[(size_t)p - 1]]) ++n_coarse;
    }
    int64_t s = -1;
    for (int64_t p = 0; p < kept; ++p) {
        const int32_t q = tmp[(size_t)p];
        if (p == 0 || cr[(size_t)q] != cr[(size_t)tmp[(size_t)p - 1]] || cc[(size_t)q] != cc[(size_t)tmp[(size_t)p - 1]]) {
            ++s;
            off[s] = (int32_t)p;
            coarse_edge_index[s] = cr[(size_t)q];
            coarse_edge_index[n_coarse + s] = cc[(size_t)q];
        }
        perm[p] = id[(size_t)q];
    }
    off[n_coarse] = (int32_t)kept;
    return n_coarse;
}

extern "C" int64_t g4c_plan_pool_edge(const int64_t *idx_hr_to_lr, int64_t n_hr, const int64_t *edge_index,
                                      int64_t n_edges, int64_t *coarse_edge_index, int32_t *perm, int32_t *off,
                                      int64_t *n_kept) {
    return g4c_plan_pool_edge_ordered(idx_hr_to_lr, n_hr, edge_index, n_edges, 0, coarse_edge_index, perm, off, n_kept);
}
