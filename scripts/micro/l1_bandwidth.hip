// L1 (TCP) -> register bandwidth per CU for the access pattern of the fused MLP kernels' weight stream: every lane loads
// 16 bytes (one wave instruction = 1 KB contiguous) from a small buffer that stays L1 / L2 resident.
// hipcc -O3 --offload-arch=gfx950 -o /tmp/l1bw scripts/micro/l1_bandwidth.hip && /tmp/l1bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void bw_kernel(const u32x4 *__restrict__ w, int span_vec, int iters, unsigned *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each wave walks its own 1 KB-granular slice of a `span_vec`-element window (16 B per element)
    u32x4 acc = {0, 0, 0, 0};
    int pos = wave * 64 + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const u32x4 v = __builtin_nontemporal_load(&w[pos]) ;
            acc ^= v;
            pos += 256;
            if (pos >= span_vec) pos -= span_vec;
        }
    }
    if (acc[0] == 0x12345678u) out[0] = acc[1] ^ acc[2] ^ acc[3];
}

template <int UNROLL>
__global__ __launch_bounds__(256) void bw_kernel_plain(const u32x4 *__restrict__ w, int span_vec, int iters, unsigned *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 acc = {0, 0, 0, 0};
    int pos = wave * 64 + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const u32x4 v = w[pos];
            acc ^= v;
            pos += 256;
            if (pos >= span_vec) pos -= span_vec;
        }
    }
    if (acc[0] == 0x12345678u) out[0] = acc[1] ^ acc[2] ^ acc[3];
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    const size_t bytes = 8u << 20;
    u32x4 *w; unsigned *out;
    hipMalloc(&w, bytes); hipMalloc(&out, 4);
    hipMemset(w, 1, bytes);
    printf("%d CUs, %.2f GHz nominal\n", n_cu, ghz);
    for (int wg_per_cu : {1, 2, 4}) {
        for (size_t span : {(size_t)12 << 10, (size_t)96 << 10, (size_t)1 << 20, (size_t)8 << 20}) {      // window per workgroup walk
            const int span_vec = (int)(span / 16), iters = 4000, UN = 8;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            bw_kernel_plain<UN><<<n_cu * wg_per_cu, 256>>>(w, span_vec, 100, out);
            hipDeviceSynchronize();
            hipEventRecord(a);
            bw_kernel_plain<UN><<<n_cu * wg_per_cu, 256>>>(w, span_vec, iters, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double total = (double)n_cu * wg_per_cu * 256 * 16.0 * iters * UN;
            printf("wgs/CU %d  window %7zu KB: %8.1f GB/s total = %6.1f B/clk/CU (at %.2f GHz)\n", wg_per_cu, span >> 10, total / ms * 1e-6,
                   total / (ms * 1e-3) / n_cu / (ghz * 1e9), ghz);
        }
    }
    return 0;
}
