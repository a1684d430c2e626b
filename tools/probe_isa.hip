// One-off ISA probe for gfx950 facts the kernels rely on (results recorded in DESIGN.md):
//   - direction of v_mov_b32_dpp wave_shr:1 / wave_shl:1 and what the edge lanes receive
//   - rounding of v_cvt_pk_u8_f32 (needed: round-half-even + saturation for saturate_cast<uchar>)
// build: hipcc -O2 --offload-arch=gfx950 tools/probe_isa.hip -o /tmp/probe_isa && /tmp/probe_isa
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint32_t* shr, uint32_t* shl, const float* f, uint32_t* q, int n) {
    int i = threadIdx.x;
    uint32_t v = 100 + i;
    shr[i] = __builtin_amdgcn_update_dpp(7777u, v, 0x138, 0xf, 0xf, false);
    shl[i] = __builtin_amdgcn_update_dpp(7777u, v, 0x130, 0xf, 0xf, false);
    if (i < n) q[i] = __builtin_amdgcn_cvt_pk_u8_f32(f[i], 0, 0u);
}
int main() {
    const float vals[] = {0.5f, 1.5f, 2.5f, 3.5f, 0.49999f, 0.50001f, 254.5f, 255.5f, 300.f, -0.5f, -3.f, 127.5f, 128.5f, 1.4999f, 2.500001f, 254.49f};
    const int n = sizeof(vals) / sizeof(float);
    uint32_t *shr, *shl, *q; float* f;
    hipMalloc(&shr, 256); hipMalloc(&shl, 256); hipMalloc(&q, 256); hipMalloc(&f, 256);
    hipMemcpy(f, vals, sizeof(vals), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, shr, shl, f, q, n);
    uint32_t a[64], b[64], c[64];
    hipMemcpy(a, shr, 256, hipMemcpyDeviceToHost); hipMemcpy(b, shl, 256, hipMemcpyDeviceToHost);
    hipMemcpy(c, q, 256, hipMemcpyDeviceToHost);
    printf("wave_shr:1 lane0=%u lane1=%u lane31=%u lane32=%u lane63=%u\n", a[0], a[1], a[31], a[32], a[63]);
    printf("wave_shl:1 lane0=%u lane1=%u lane31=%u lane32=%u lane62=%u lane63=%u\n", b[0], b[1], b[31], b[32], b[62], b[63]);
    for (int i = 0; i < n; ++i) printf("cvt_pk_u8_f32(%g) = %u\n", vals[i], c[i] & 0xff);
    return 0;
}
