// Small bandwidth-bound helpers of the REMuS-GNN path and of the rollout loop.
#include "g4c_common.h"

namespace {

// out[e, f] = v[node[e], 2f] * U[e,0] + v[node[e], 2f+1] * U[e,1]
// reference: (field[col].reshape(E,-1,2) * edgeUnitVector.unsqueeze(1)).sum(-1)
//            graphs4cfd/nn/remus_gnn.py:124-126 and nn/blocks.py:454.
// Products and the sum are rounded separately (no fma contraction) like the torch ops.
__global__ __launch_bounds__(256) void project_to_edges_kernel(
    const float *__restrict__ v, int v_ld, const int *__restrict__ node, const float *__restrict__ unit,
    long long n_edges, int n_feat, float *__restrict__ out, int out_ld) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long e = gid / n_feat;
    const int f = (int)(gid % n_feat);
    if (e >= n_edges) return;
    const long long n = node ? node[e] : e;
    const float2 x = *reinterpret_cast<const float2 *>(v + n * v_ld + 2 * f);
    const float u0 = unit[2 * e], u1 = unit[2 * e + 1];
    out[e * out_ld + f] = __fadd_rn(__fmul_rn(x.x, u0), __fmul_rn(x.y, u1));
}

// out[n, 2f+c] = sum_j unit_inv[n, c, j] * e[n*k + j, f]   (nn/blocks.py:110-114)
__global__ __launch_bounds__(256) void edge_scalar_to_node_vector_kernel(
    const float *__restrict__ e, int e_ld, const float *__restrict__ unit_inv, int k,
    long long n_nodes, int n_feat, float *__restrict__ out, int out_ld) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = gid / n_feat;
    const int f = (int)(gid % n_feat);
    if (n >= n_nodes) return;
    const float *ui = unit_inv + n * 2 * k;
    float s0 = 0.f, s1 = 0.f;
    for (int j = 0; j < k; ++j) {
        const float x = e[(n * k + j) * e_ld + f];
        s0 = fmaf(ui[j], x, s0);
        s1 = fmaf(ui[k + j], x, s1);
    }
    *reinterpret_cast<float2 *>(out + n * out_ld + 2 * f) = make_float2(s0, s1);
}

__global__ __launch_bounds__(256) void copy_cols_kernel(
    const float *__restrict__ src, int src_ld, int scol0, const int *__restrict__ idx,
    float *__restrict__ dst, int dst_ld, int dcol0, int width, long long n_rows) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long r = gid / width;
    const int c = (int)(gid % width);
    if (r >= n_rows) return;
    const long long sr = idx ? idx[r] : r;
    dst[r * dst_ld + dcol0 + c] = src[sr * src_ld + scol0 + c];
}

// GNN.solve bookkeeping (graphs4cfd/nn/model.py:316-327) with the step index read on device.
__global__ __launch_bounds__(256) void rollout_advance_kernel(
    float *__restrict__ field, int field_cols, const float *__restrict__ pred, int nf,
    float *__restrict__ outputs, int out_ld, int *__restrict__ step, long long n_nodes) {
    const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = __builtin_nontemporal_load(step);
    if (n < n_nodes) {
        float *fr = field + n * field_cols;
        // roll left by nf, then append pred (row-local, in place, ascending order is safe)
        for (int c = 0; c + nf < field_cols; ++c) fr[c] = fr[c + nf];
        for (int c = 0; c < nf; ++c) {
            const float y = pred[n * nf + c];
            fr[field_cols - nf + c] = y;
            // out_ld == 0: step-major outputs [steps][n_nodes][nf] — the step's slice is one contiguous block (row-major rows of many
            // steps take one 4 nf-byte store per row and step, each into a line of its own: 23 us for 100k nodes instead of 3)
            if (out_ld) outputs[n * out_ld + (long long)nf * t + c] = y;
            else outputs[((long long)t * n_nodes + n) * nf + c] = y;
        }
    }
    // *step = t + 1 by the LAST workgroup to get here (step[1]: a ticket counter, zero between launches): every workgroup has read
    // the step index before it takes its ticket, so nobody can see the new value — no trailing single-thread launch (4 us per step)
    // (round 6: no __threadfence() in front of the ticket — nothing this workgroup WROTE has to be visible to the others, only its read
    // of the step index has to have completed, which the wait below says; 391 workgroups each writing back and invalidating their XCD's
    // L2 was most of the launch's 16 us)
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (atomicAdd(step + 1, 1) == (int)gridDim.x - 1) { step[1] = 0; step[0] = t + 1; }
    }
}

__global__ __launch_bounds__(256) void activation_kernel(float *__restrict__ x, long long n, int act) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 v = *reinterpret_cast<float4 *>(x + i);
        v.x = g4c::apply_act(v.x, act); v.y = g4c::apply_act(v.y, act);
        v.z = g4c::apply_act(v.z, act); v.w = g4c::apply_act(v.w, act);
        *reinterpret_cast<float4 *>(x + i) = v;
    } else {
        for (long long j = i; j < n; ++j) x[j] = g4c::apply_act(x[j], act);
    }
}

// out[r, c] = a[r, a_col0 + c] + b[r, c]   (residual time step, nn/remus_gnn.py:199)
__global__ __launch_bounds__(256) void add_cols_kernel(const float *__restrict__ a, int a_ld, int a_col0,
                                                       const float *__restrict__ b, int b_ld,
                                                       float *__restrict__ out, int out_ld, int width, long long n_rows) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long r = gid / width;
    const int c = (int)(gid % width);
    if (r >= n_rows) return;
    out[r * out_ld + c] = a[r * a_ld + a_col0 + c] + b[r * b_ld + c];
}

}  // namespace

extern "C" int g4c_project_to_edges(const float *v, int32_t v_ld, const int32_t *node, const float *unit,
                                    int64_t n_edges, int32_t n_feat, float *out, int32_t out_ld, void *stream) {
    G4C_REQUIRE(v && unit && out, G4C_EINVAL, "g4c_project_to_edges: null pointer");
    g4c::DeviceGuard on_device(out);
    G4C_REQUIRE(n_edges >= 0 && n_feat > 0 && v_ld >= 2 * n_feat && out_ld >= n_feat && v_ld % 2 == 0 && ((uintptr_t)v % 8 == 0),
                G4C_EINVAL, "g4c_project_to_edges: bad sizes n_feat=%d v_ld=%d out_ld=%d", n_feat, v_ld, out_ld);
    if (n_edges == 0) return G4C_OK;
    const long long total = n_edges * n_feat;
    project_to_edges_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        v, v_ld, node, unit, n_edges, n_feat, out, out_ld);
    return g4c::check_launch("g4c_project_to_edges");
}

extern "C" int g4c_edge_scalar_to_node_vector(const float *e, int32_t e_ld, const float *unit_inv, int32_t k,
                                              int64_t n_nodes, int32_t n_feat, float *out, int32_t out_ld, void *stream) {
    G4C_REQUIRE(e && unit_inv && out, G4C_EINVAL, "g4c_edge_scalar_to_node_vector: null pointer");
    g4c::DeviceGuard on_device(out);
    G4C_REQUIRE(n_nodes >= 0 && k > 0 && n_feat > 0 && e_ld >= n_feat && out_ld >= 2 * n_feat && out_ld % 2 == 0 && ((uintptr_t)out % 8 == 0),
                G4C_EINVAL, "g4c_edge_scalar_to_node_vector: bad sizes k=%d n_feat=%d e_ld=%d out_ld=%d", k, n_feat, e_ld, out_ld);
    if (n_nodes == 0) return G4C_OK;
    const long long total = n_nodes * n_feat;
    edge_scalar_to_node_vector_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        e, e_ld, unit_inv, k, n_nodes, n_feat, out, out_ld);
    return g4c::check_launch("g4c_edge_scalar_to_node_vector");
}

extern "C" int g4c_activation_inplace(float *x, int64_t n, int32_t act, void *stream) {
    G4C_REQUIRE(x || n == 0, G4C_EINVAL, "g4c_activation_inplace: null pointer");
    g4c::DeviceGuard on_device(x);
    G4C_REQUIRE(n >= 0 && act >= 0 && act <= 2 && ((uintptr_t)x % 16 == 0), G4C_EINVAL, "g4c_activation_inplace: bad arguments");
    if (n == 0 || act == G4C_ACT_NONE) return G4C_OK;
    const long long nthreads = (n + 3) / 4;
    activation_kernel<<<dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(x, n, act);
    return g4c::check_launch("g4c_activation_inplace");
}

extern "C" int g4c_add_cols(const float *a, int32_t a_ld, int32_t a_col0, const float *b, int32_t b_ld,
                            float *out, int32_t out_ld, int32_t width, int64_t n_rows, void *stream) {
    G4C_REQUIRE(a && b && out, G4C_EINVAL, "g4c_add_cols: null pointer");
    g4c::DeviceGuard on_device(out);
    G4C_REQUIRE(width > 0 && n_rows >= 0 && a_col0 >= 0 && a_ld >= a_col0 + width && b_ld >= width && out_ld >= width,
                G4C_EINVAL, "g4c_add_cols: bad sizes width=%d a_ld=%d b_ld=%d out_ld=%d", width, a_ld, b_ld, out_ld);
    if (n_rows == 0) return G4C_OK;
    const long long total = n_rows * width;
    add_cols_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        a, a_ld, a_col0, b, b_ld, out, out_ld, width, n_rows);
    return g4c::check_launch("g4c_add_cols");
}

extern "C" int g4c_copy_cols(const float *src, int32_t src_ld, int32_t scol0, const int32_t *idx,
                             float *dst, int32_t dst_ld, int32_t dcol0, int32_t width, int64_t n_rows, void *stream) {
    G4C_REQUIRE(src && dst, G4C_EINVAL, "g4c_copy_cols: null pointer");
    g4c::DeviceGuard on_device(dst);
    G4C_REQUIRE(width > 0 && n_rows >= 0 && src_ld >= scol0 + width && dst_ld >= dcol0 + width && scol0 >= 0 && dcol0 >= 0,
                G4C_EINVAL, "g4c_copy_cols: bad sizes width=%d src_ld=%d dst_ld=%d", width, src_ld, dst_ld);
    if (n_rows == 0) return G4C_OK;
    const long long total = n_rows * width;
    copy_cols_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        src, src_ld, scol0, idx, dst, dst_ld, dcol0, width, n_rows);
    return g4c::check_launch("g4c_copy_cols");
}

extern "C" int g4c_rollout_advance(float *field, int32_t field_cols, const float *pred, int32_t nf,
                                   float *outputs, int32_t out_ld, int32_t *step, int64_t n_nodes, void *stream) {
    G4C_REQUIRE(field && pred && outputs && step, G4C_EINVAL, "g4c_rollout_advance: null pointer");
    g4c::DeviceGuard on_device(field);
    G4C_REQUIRE(nf > 0 && field_cols >= nf && (out_ld >= nf || out_ld == 0) && n_nodes >= 0, G4C_EINVAL,
                "g4c_rollout_advance: bad sizes nf=%d field_cols=%d out_ld=%d", nf, field_cols, out_ld);
    hipStream_t s = (hipStream_t)stream;
    const long long blocks = n_nodes > 0 ? (n_nodes + 255) / 256 : 1;          // (no nodes: one workgroup, for the step index)
    rollout_advance_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(field, field_cols, pred, nf, outputs, out_ld, step, n_nodes);
    return g4c::check_launch("g4c_rollout_advance");
}
