// Can a workgroup own all 160 KB of a CU's LDS (static allocation of exactly 163840 bytes)?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned *out) {
    __shared__ __attribute__((aligned(1024))) unsigned w[163840 / 4];
    for (int i = threadIdx.x; i < 163840 / 4; i += 512) w[i] = i * 2654435761u;
    __syncthreads();
    unsigned s = 0;
    for (int i = threadIdx.x; i < 163840 / 4; i += 512) s ^= w[(i * 7 + 3) % (163840 / 4)];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    unsigned *d; hipMalloc(&d, 256 * 512 * 4);
    k<<<256, 512>>>(d);
    hipError_t e = hipDeviceSynchronize();
    printf("launch with 163840 B of static LDS: %s / %s\n", hipGetErrorString(hipGetLastError()), hipGetErrorString(e));
    return 0;
}
