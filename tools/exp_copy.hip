// Copy-bandwidth ceiling probe on MI355X (development aid): what a pure 1:1 read/write
// stream reaches with different per-lane unrolls, grid shapes and cache-policy hints.
// build: hipcc -O3 --offload-arch=gfx950 tools/exp_copy.hip -o /tmp/exp_copy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int U, int NT>
__global__ __launch_bounds__(256) void copyk(const u32x4* __restrict__ a, u32x4* __restrict__ b, size_t n) {
    // each block moves U*256 consecutive 16-B items; grid-stride over such tiles
    for (size_t base = (size_t)blockIdx.x * (256 * U); base < n; base += (size_t)gridDim.x * (256 * U)) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t i = base + u * 256 + threadIdx.x;
            if (i < n) v[u] = (NT & 1) ? __builtin_nontemporal_load(a + i) : a[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t i = base + u * 256 + threadIdx.x;
            if (i < n) { if (NT & 2) __builtin_nontemporal_store(v[u], b + i); else b[i] = v[u]; }
        }
    }
}
__global__ void fill(uint32_t* p, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 7);
}
template <int U, int NT>
void run(const char* tag, const u32x4* a, u32x4* b, size_t n, int blocks) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    if (blocks <= 0) blocks = (int)((n + 256 * U - 1) / (256 * U));
    hipLaunchKernelGGL((copyk<U, NT>), dim3(blocks), dim3(256), 0, 0, a, b, n);
    hipEventRecord(s);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((copyk<U, NT>), dim3(blocks), dim3(256), 0, 0, a, b, n);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= 10;
    printf("%-28s U=%d NT=%d blocks=%-8d %7.3f ms %7.1f GB/s\n", tag, U, NT, blocks, ms, 2.0 * n * 16 / ms / 1e6); fflush(stdout);
}
int main() {
    size_t bytes = 64ull * 2160 * 3840 * 3, n = bytes / 16;
    u32x4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)a, bytes / 4);
    run<1, 0>("oneshot", a, b, n, 0); run<2, 0>("oneshot", a, b, n, 0); run<4, 0>("oneshot", a, b, n, 0); run<8, 0>("oneshot", a, b, n, 0);
    run<1, 2>("oneshot", a, b, n, 0); run<4, 2>("oneshot", a, b, n, 0); run<4, 1>("oneshot", a, b, n, 0); run<4, 3>("oneshot", a, b, n, 0);
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) { run<4, 0>("gridstride", a, b, n, blocks); run<1, 0>("gridstride", a, b, n, blocks); }
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(s); for (int i = 0; i < 10; ++i) hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= 10; printf("hipMemcpyAsync d2d %7.3f ms %7.1f GB/s\n", ms, 2.0 * bytes / ms / 1e6);
    return 0;
}
