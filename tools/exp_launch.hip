// Workgroup dispatch-rate probe (development aid): how long does a grid of N small 256-thread
// workgroups take when each does almost nothing?  Bounds kernels that map one 32x32 tile to a
// workgroup (the affine family launches 518k workgroups for 64 4K frames).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void tiny(uint32_t* out) {
    if (threadIdx.x == 0) out[blockIdx.x & 1023] = blockIdx.x;
}
__global__ __launch_bounds__(256) void store12(uint32_t* out) {   // every lane writes 12 bytes (one tile's output)
    uint32_t* p = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 3;
    p[0] = blockIdx.x; p[1] = threadIdx.x; p[2] = 7;
}
int main() {
    uint32_t* buf; hipMalloc(&buf, 518400ull * 256 * 12);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int nb : {16200, 129600, 518400}) {
        for (int which = 0; which < 2; ++which) {
            for (int w = 0; w < 2; ++w) { if (which) hipLaunchKernelGGL(store12, dim3(nb), dim3(256), 0, 0, buf); else hipLaunchKernelGGL(tiny, dim3(nb), dim3(256), 0, 0, buf); }
            hipEventRecord(a);
            for (int i = 0; i < 10; ++i) { if (which) hipLaunchKernelGGL(store12, dim3(nb), dim3(256), 0, 0, buf); else hipLaunchKernelGGL(tiny, dim3(nb), dim3(256), 0, 0, buf); }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%-8s blocks=%-7d %8.4f ms  (%.2f ns per workgroup)\n", which ? "store12" : "tiny", nb, ms / 10, ms / 10 * 1e6 / nb);
        }
    }
    return 0;
}
