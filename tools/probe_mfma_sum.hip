// Probe (development aid): how exactly does v_mfma_f32_32x32x16_f16 sum products whose A operand is an f16
// denormal (the byte-as-denormal trick of sepconv_mfma.inc)?  A = constant byte v, B = the hi / lo halves of a
// 13-tap Gaussian at scale 2^e; prints the result of hi only, lo only, hi then lo chained, and the exact sums.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_sum.hip -o _exp/probe_mfma_sum
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(int byte, const _Float16* hi, const _Float16* lo, float* out) {
    const int l = threadIdx.x, h = l >> 5;
    union { f16x8 v; uint16_t u[8]; } a;
    f16x8 bh, bl;
    for (int j = 0; j < 8; ++j) { a.u[j] = (uint16_t)byte; bh[j] = hi[8 * h + j]; bl[j] = lo[8 * h + j]; }
    const f32x16 z = {};
    const f32x16 dh = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, bh, z, 0, 0, 0);
    const f32x16 dl = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, bl, z, 0, 0, 0);
    const f32x16 dc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, bl, dh, 0, 0, 0);
    const f32x16 dr = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, bh, dl, 0, 0, 0);
    if (l == 0) { out[0] = dh[0]; out[1] = dl[0]; out[2] = dc[0]; out[3] = dr[0]; }
}

int main() {
    const int K = 13;
    double g[16] = {0}, sum = 0;
    for (int i = 0; i < K; ++i) { const double x = i - K / 2; g[i] = exp(-x * x / 8.0); sum += g[i]; }
    for (int i = 0; i < K; ++i) g[i] = (double)(float)(g[i] / sum);
    _Float16 *dhi, *dlo; float* dout;
    hipMalloc(&dhi, 32); hipMalloc(&dlo, 32); hipMalloc(&dout, 16);
    for (int e = 15; e <= 17; e += 2)
        for (int byte : {1, 2, 3, 7, 200}) {
            _Float16 hi[16], lo[16]; double shi = 0, slo = 0;
            for (int i = 0; i < 16; ++i) {
                const float ws = (float)g[i] * (float)(1 << e);
                hi[i] = (_Float16)ws; lo[i] = (_Float16)(ws - (float)hi[i]);
                shi += (double)(float)hi[i]; slo += (double)(float)lo[i];
            }
            hipMemcpy(dhi, hi, 32, hipMemcpyHostToDevice); hipMemcpy(dlo, lo, 32, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, byte, dhi, dlo, dout);
            float o[4]; hipMemcpy(o, dout, 16, hipMemcpyDeviceToHost);
            const double s = 16777216.0 / byte;     // to units of weight * 1
            printf("scale 2^%d byte %3d: hi %.6f (exact %.6f)  lo %.6f (exact %.6f)  hi->lo %.6f  lo->hi %.6f (exact %.6f)\n", e, byte,
                   o[0] * s, shi, o[1] * s, slo, o[2] * s, o[3] * s, shi + slo);
        }
    return 0;
}
