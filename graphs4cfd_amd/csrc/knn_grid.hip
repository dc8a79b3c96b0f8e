// k nearest neighbours of every point of a cloud among the cloud itself (the search inside `connect_knn`,
// reference graphs4cfd/transforms/connect.py:9-72, which calls torch_cluster.knn / a k-d tree on the host).
//
// MI355X form: the points are binned into a uniform cell grid (a few points per cell; the binning — cell ids, a stable
// sort, the cells' offsets — is done by the caller on the device), and one lane per query point scans the block of
// cells within R rings of its own cell, keeping its k best candidates in registers.  The answer is exact: a query is
// finished only when its k-th distance is no larger than the distance to the nearest face of the scanned block that
// still has cells behind it; otherwise R grows and the block is scanned again (rare: R = 1 suffices for almost every
// point of a quasi-uniform cloud).  Distances are accumulated in fp64 from the fp32 coordinates, like the host k-d
// tree does, so that the ascending-distance order of the result is the host's.  HBM-bound integer / compare work:
// sorted points of neighbouring cells are contiguous, a wavefront's 64 queries are 64 consecutive sorted points.
#include "g4c_common.h"

namespace {

constexpr int KMAX = 16;   // candidate registers: instantiated for 8 (k <= 8: half the swap chain) and 16

// SELF: the queries are the cloud's own points (query s = sorted point s, itself excluded, result row = its original
// index); otherwise m separate query points qpos / qcell (their cell in the cloud's grid, clamped), result row = s.
template <int DIM, bool SELF, int KM>
__global__ __launch_bounds__(256) void knn_grid_kernel(
    const float *__restrict__ pos,        // [n, DIM] points in cell-sorted order
    const int *__restrict__ cell,         // [n] cell id of each sorted point (SELF)
    const int *__restrict__ order,        // [n] original index of each sorted point
    const int *__restrict__ cell_start,   // [n_cells + 1] first sorted point of each cell
    const float *__restrict__ qpos, const int *__restrict__ qcell, long long m,
    int nc0, int nc1, int nc2, float o0, float o1, float o2, float h, int k,
    int64_t *__restrict__ out) {          // [m, k] original indices, ascending distance
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= m) return;
    if (SELF) { qpos = pos; qcell = cell; }
    const int nc[3] = {nc0, nc1, nc2};
    const float org[3] = {o0, o1, o2};
    double q[DIM];
    int c[3] = {0, 0, 0};
    {
        int id = qcell[s];
        c[0] = id % nc0; id /= nc0;
        c[1] = id % nc1; id /= nc1;
        c[2] = id;
    }
#pragma unroll
    for (int a = 0; a < DIM; ++a) q[a] = (double)qpos[s * DIM + a];

    double best_d[KM];
    int best_j[KM];
    int max_r = 0;
#pragma unroll
    for (int a = 0; a < DIM; ++a) max_r = max(max_r, max(c[a], nc[a] - 1 - c[a]));

    for (int R = 1;; ++R) {
#pragma unroll
        for (int u = 0; u < KM; ++u) { best_d[u] = 1e300; best_j[u] = -1; }
        int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
#pragma unroll
        for (int a = 0; a < DIM; ++a) { lo[a] = max(c[a] - R, 0); hi[a] = min(c[a] + R, nc[a] - 1); }
        for (int z = lo[2]; z <= hi[2]; ++z)
            for (int y = lo[1]; y <= hi[1]; ++y) {
                // the cells lo[0] .. hi[0] of one grid line are consecutive cell ids: one contiguous run of sorted points
                const long long line = ((long long)z * nc1 + y) * nc0;
                const int beg = cell_start[line + lo[0]], end = cell_start[line + hi[0] + 1];
                for (int j = beg; j < end; ++j) {
                    if (SELF && j == s) continue;
                    double d = 0.0;
#pragma unroll
                    for (int a = 0; a < DIM; ++a) {
                        const double t = (double)pos[(long long)j * DIM + a] - q[a];
                        d += t * t;
                    }
                    int jj = j;
                    // sorted insertion by a swap chain: static register indexing only
#pragma unroll
                    for (int u = 0; u < KM; ++u) {
                        if (u < k && d < best_d[u]) {
                            const double td = best_d[u]; best_d[u] = d; d = td;
                            const int tj = best_j[u]; best_j[u] = jj; jj = tj;
                        }
                    }
                }
            }
        double kth = 1e300;
#pragma unroll
        for (int u = 0; u < KM; ++u)
            if (u == k - 1) kth = best_d[u];
        if (R >= max_r) break;   // the block is the whole grid
        // distance to the nearest face of the block with cells behind it (shrunk by a rounding margin)
        double safe = 1e300;
#pragma unroll
        for (int a = 0; a < DIM; ++a) {
            if (c[a] - R > 0) safe = fmin(safe, q[a] - ((double)org[a] + (double)(c[a] - R) * (double)h));
            if (c[a] + R < nc[a] - 1) safe = fmin(safe, ((double)org[a] + (double)(c[a] + R + 1) * (double)h) - q[a]);
        }
        safe -= 1e-5 * (double)h;
        if (safe > 0.0 && kth <= safe * safe) break;
    }
    const long long row = SELF ? (long long)order[s] : s;
#pragma unroll
    for (int u = 0; u < KM; ++u)
        if (u < k) out[row * k + u] = best_j[u] >= 0 ? (int64_t)order[best_j[u]] : (int64_t)-1;
}

}  // namespace

extern "C" int g4c_knn_grid(const float *pos_sorted, const int32_t *cell_sorted, const int32_t *order,
                            const int32_t *cell_start, int64_t n, int32_t dim, const int32_t *n_cells /*host[3]*/,
                            const float *origin /*host[3]*/, float cell_size, int32_t k, int64_t *out, void *stream) {
    G4C_REQUIRE(pos_sorted && cell_sorted && order && cell_start && n_cells && origin && out, G4C_EINVAL,
                "g4c_knn_grid: null pointer");
    g4c::DeviceGuard on_device(out);
    G4C_REQUIRE(dim == 2 || dim == 3, G4C_EINVAL, "g4c_knn_grid: dim=%d, must be 2 or 3", dim);
    G4C_REQUIRE(k >= 1 && k <= KMAX, G4C_EINVAL, "g4c_knn_grid: k=%d outside [1, %d]", k, KMAX);
    G4C_REQUIRE(n > k && n < (1LL << 31), G4C_EINVAL, "g4c_knn_grid: n=%lld needs more than k=%d points", (long long)n, k);
    G4C_REQUIRE(cell_size > 0.f && n_cells[0] >= 1 && n_cells[1] >= 1 && n_cells[2] >= 1 && (dim == 3 || n_cells[2] == 1) &&
                    (long long)n_cells[0] * n_cells[1] * n_cells[2] < (1LL << 31),
                G4C_EINVAL, "g4c_knn_grid: bad grid %d x %d x %d, cell %g", n_cells[0], n_cells[1], n_cells[2], (double)cell_size);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
#define G4C_KNN_LAUNCH(DIM_, SELF_, KM_, ...) knn_grid_kernel<DIM_, SELF_, KM_><<<grid, block, 0, (hipStream_t)stream>>>(__VA_ARGS__)
#define G4C_KNN_DISPATCH(SELF_, ...)                                                  \
    do {                                                                              \
        if (dim == 2 && k <= 8) G4C_KNN_LAUNCH(2, SELF_, 8, __VA_ARGS__);             \
        else if (dim == 2) G4C_KNN_LAUNCH(2, SELF_, 16, __VA_ARGS__);                 \
        else if (k <= 8) G4C_KNN_LAUNCH(3, SELF_, 8, __VA_ARGS__);                    \
        else G4C_KNN_LAUNCH(3, SELF_, 16, __VA_ARGS__);                               \
    } while (0)
    G4C_KNN_DISPATCH(true, pos_sorted, cell_sorted, order, cell_start, nullptr, nullptr, n, n_cells[0], n_cells[1],
                     dim == 3 ? n_cells[2] : 1, origin[0], origin[1], dim == 3 ? origin[2] : 0.f, cell_size, k, out);
    return g4c::check_launch("g4c_knn_grid");
}

extern "C" int g4c_knn_grid_query(const float *pos_sorted, const int32_t *order, const int32_t *cell_start, int64_t n, int32_t dim,
                                  const int32_t *n_cells /*host[3]*/, const float *origin /*host[3]*/, float cell_size,
                                  const float *q_pos, const int32_t *q_cell, int64_t m, int32_t k, int64_t *out, void *stream) {
    G4C_REQUIRE(pos_sorted && order && cell_start && n_cells && origin && (m == 0 || (q_pos && q_cell && out)), G4C_EINVAL,
                "g4c_knn_grid_query: null pointer");
    g4c::DeviceGuard on_device(pos_sorted);
    G4C_REQUIRE(dim == 2 || dim == 3, G4C_EINVAL, "g4c_knn_grid_query: dim=%d, must be 2 or 3", dim);
    G4C_REQUIRE(k >= 1 && k <= KMAX, G4C_EINVAL, "g4c_knn_grid_query: k=%d outside [1, %d]", k, KMAX);
    G4C_REQUIRE(n >= k && n < (1LL << 31) && m >= 0 && m < (1LL << 31), G4C_EINVAL,
                "g4c_knn_grid_query: n=%lld points for k=%d neighbours, m=%lld queries", (long long)n, k, (long long)m);
    G4C_REQUIRE(cell_size > 0.f && n_cells[0] >= 1 && n_cells[1] >= 1 && n_cells[2] >= 1 && (dim == 3 || n_cells[2] == 1) &&
                    (long long)n_cells[0] * n_cells[1] * n_cells[2] < (1LL << 31),
                G4C_EINVAL, "g4c_knn_grid_query: bad grid %d x %d x %d, cell %g", n_cells[0], n_cells[1], n_cells[2], (double)cell_size);
    if (m == 0) return G4C_OK;
    const dim3 grid((unsigned)((m + 255) / 256)), block(256);
    G4C_KNN_DISPATCH(false, pos_sorted, nullptr, order, cell_start, q_pos, q_cell, m, n_cells[0], n_cells[1],
                     dim == 3 ? n_cells[2] : 1, origin[0], origin[1], dim == 3 ? origin[2] : 0.f, cell_size, k, out);
    return g4c::check_launch("g4c_knn_grid_query");
}
