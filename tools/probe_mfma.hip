// Probe (development aid): v_mfma_f32_32x32x16_f16 operand layouts and f16-denormal inputs on gfx950.
//   A: 32 x 16 bytes encoded as f16 bit patterns 0x00bb (= b * 2^-24, an f16 denormal for b < 1024)
//   B: 16 x 32 f16 weights
// Checks D = A.B against the host, i.e. (a) lane l holds A[i = l&31][k = 8 (l>>5) + j], B[k = 8 (l>>5) + j][n = l&31],
// D reg g -> row (g&3) + 8 (g>>2) + 4 (l>>5), col l&31; (b) denormal inputs are not flushed.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe_mfma.hip -o _exp/probe_mfma
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const uint8_t* A, const _Float16* B, float* D, float scale) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    union { f16x8 v; uint16_t u[8]; } a;
    f16x8 b;
    for (int j = 0; j < 8; ++j) { a.u[j] = A[i * 16 + 8 * h + j]; b[j] = B[(8 * h + j) * 32 + i]; }
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b, acc, 0, 0, 0);
    for (int g = 0; g < 16; ++g) D[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + i] = acc[g] * scale;
}

int main() {
    uint8_t hA[32 * 16]; _Float16 hB[16 * 32]; float hD[32 * 32];
    for (int i = 0; i < 32; ++i) for (int kk = 0; kk < 16; ++kk) hA[i * 16 + kk] = (uint8_t)((i * 37 + kk * 11 + 5) & 255);
    for (int kk = 0; kk < 16; ++kk) for (int n = 0; n < 32; ++n) hB[kk * 32 + n] = (_Float16)(0.001f * (1 + ((kk * 7 + n * 3) % 29)));
    uint8_t* dA; _Float16* dB; float* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, 16777216.0f);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double maxrel = 0; int bad = 0;
    for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) {
        double ref = 0;
        for (int kk = 0; kk < 16; ++kk) ref += (double)hA[i * 16 + kk] * (double)(float)hB[kk * 32 + n];
        const double rel = fabs(hD[i * 32 + n] - ref) / fmax(fabs(ref), 1e-9);
        if (rel > maxrel) maxrel = rel;
        if (rel > 1e-6) ++bad;
    }
    printf("mfma f16 denormal-byte probe: max rel err %.3g, mismatches %d of 1024 (sample D[3][5] = %.6f)\n", maxrel, bad, hD[3 * 32 + 5]);
    return bad != 0;
}
