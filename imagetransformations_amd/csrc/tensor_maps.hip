// Float-tensor corruption maps of the patch pipelines (/root/reference/pipenline/angellic.py:34-46,
// angellic2.py:47-50): unnormalised images in [0,1], fp32, any shape (treated as a flat array).
//   add_brightness      y = clamp(x + f, 0, 1)
//   add_contrast        y = clamp((x - 0.5) * f + 0.5, 0, 1)
//   add_gaussian_noise  y = clamp(x + (z * std + mean), 0, 1)     z = torch.randn_like(x), drawn by the caller
// Each fp32 operation is the one torch's eager kernels perform, in the same order (the library
// is built with -ffp-contract=off), so results are bit-identical; NaN passes through clamp as in
// torch.  `mask` (optional) receives 1 where the value before the clamp lay in [0,1] — the set on
// which torch.clamp's backward lets the gradient through.  16 bytes per lane, grid-stride.
#include "imgxf_common.h"

namespace imgxf {

__device__ __forceinline__ float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

template <int MODE>
__device__ __forceinline__ float map1(float x, float z, float p0, float p1) {
    if (MODE == IMGXF_F32_BRIGHTNESS) return x + p0;
    if (MODE == IMGXF_F32_CONTRAST) return (x - 0.5f) * p0 + 0.5f;
    return x + (z * p0 + p1);                                   // noise: std = p0, mean = p1
}

template <int MODE, bool MASK>
__global__ __launch_bounds__(256) void f32_map_kernel(const float* __restrict__ src, const float* __restrict__ noise,
                                                      float* __restrict__ dst, u8* __restrict__ mask,
                                                      int64_t count, float p0, float p1, int vec) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        const int64_t n4 = count >> 2;
        for (; i < n4; i += stride) {
            const float4 x = ((const float4*)src)[i];
            float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE == IMGXF_F32_NOISE) z = ((const float4*)noise)[i];
            const float v[4] = {map1<MODE>(x.x, z.x, p0, p1), map1<MODE>(x.y, z.y, p0, p1),
                                map1<MODE>(x.z, z.z, p0, p1), map1<MODE>(x.w, z.w, p0, p1)};
            ((float4*)dst)[i] = make_float4(clamp01(v[0]), clamp01(v[1]), clamp01(v[2]), clamp01(v[3]));
            if (MASK) {
                u32 m = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) m |= (u32)(v[e] >= 0.0f && v[e] <= 1.0f) << (8 * e);
                ((u32*)mask)[i] = m;
            }
        }
        i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x;       // tail elements
    }
    for (; i < count; i += stride) {
        const float v = map1<MODE>(src[i], MODE == IMGXF_F32_NOISE ? noise[i] : 0.0f, p0, p1);
        dst[i] = clamp01(v);
        if (MASK) mask[i] = (u8)(v >= 0.0f && v <= 1.0f);
    }
}

template <int MODE>
static int launch_f32_map(const float* src, const float* noise, float* dst, u8* mask, int64_t count,
                          float p0, float p1, hipStream_t st) {
    const bool vec = ((((uintptr_t)src) | ((uintptr_t)dst) | ((uintptr_t)noise)) & 15) == 0 && (((uintptr_t)mask) & 3) == 0;
    int64_t blocks = ((vec ? (count + 3) / 4 : count) + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks);
    if (mask) hipLaunchKernelGGL((f32_map_kernel<MODE, true>), dim3((unsigned)blocks), dim3(256), 0, st, src, noise, dst, mask, count, p0, p1, vec ? 1 : 0);
    else hipLaunchKernelGGL((f32_map_kernel<MODE, false>), dim3((unsigned)blocks), dim3(256), 0, st, src, noise, dst, mask, count, p0, p1, vec ? 1 : 0);
    return launch_status();
}

// ToTensor (+ Normalize): uint8 HWC -> float32 CHW, x/255 correctly rounded (Tensor.div(255)),
// then (x - mean[c]) / std[c] as two fp32 operations (Tensor.sub_ / div_), the model-input step
// that follows the transformations in every evaluation script of the reference
// (e.g. fall_2025/transformations_code:57 ToTensor; T.Normalize at 68 call sites).
struct NormArgs { float mean[4], std[4]; int normalize; };

__device__ __forceinline__ float unit255f(u32 b) {
    const float v = (float)b, r = __uint_as_float(0x3b808081u);      // RN(1/255); one residual step makes it exact
    const float q = v * r;
    return fmaf(fmaf(-255.0f, q, v), r, q);
}

template <int C>
__global__ __launch_bounds__(256) void to_tensor_kernel(View s, float* __restrict__ dst, NormArgs a) {
    const int64_t plane = (int64_t)s.h * s.w;
    const int64_t total = (int64_t)s.n * plane;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % s.w);
        const int64_t r = i / s.w;
        const int y = (int)(r % s.h), f = (int)(r / s.h);
        const u8* sp = s.row(f, y) + (int64_t)x * C;
        float* dp = dst + (int64_t)f * C * plane + (int64_t)y * s.w + x;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float v = unit255f(sp[c]);
            if (a.normalize) v = (v - a.mean[c]) / a.std[c];
            dp[c * plane] = v;
        }
    }
}

} // namespace imgxf

using namespace imgxf;

IMGXF_API int imgxf_to_tensor_f32(const imgxf_view* src, float* dst, const float* mean, const float* std,
                                  void* stream) {
    IMGXF_CHECK(check_view(src));
    if (empty_view(src)) return IMGXF_OK;
    if (!dst) return IMGXF_ERR_NULL;
    if (((uintptr_t)dst) & 3) return IMGXF_ERR_ARG;
    if ((mean == nullptr) != (std == nullptr)) return IMGXF_ERR_NULL;
    if (src->c == 2) return IMGXF_ERR_UNSUPPORTED;
    NormArgs a;
    a.normalize = mean != nullptr;
    for (int c = 0; c < 4; ++c) {
        a.mean[c] = a.normalize && c < src->c ? mean[c] : 0.0f;
        a.std[c] = a.normalize && c < src->c ? std[c] : 1.0f;
    }
    const View s = make_view(src);
    const int64_t total = (int64_t)s.n * s.h * s.w;
    int64_t blocks = (total + 255) / 256;
    blocks = blocks > 65536 ? 65536 : blocks;
    hipStream_t st = (hipStream_t)stream;
    switch (s.c) {
        case 1: hipLaunchKernelGGL(to_tensor_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, s, dst, a); break;
        case 3: hipLaunchKernelGGL(to_tensor_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, st, s, dst, a); break;
        case 4: hipLaunchKernelGGL(to_tensor_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, s, dst, a); break;
    }
    return launch_status();
}

IMGXF_API int imgxf_f32_map(const float* src, const float* noise, float* dst, uint8_t* mask,
                            int64_t count, int mode, float p0, float p1, void* stream) {
    if (count < 0) return IMGXF_ERR_SHAPE;
    if (count == 0) return IMGXF_OK;
    if (!src || !dst) return IMGXF_ERR_NULL;
    if ((((uintptr_t)src) | ((uintptr_t)dst) | ((uintptr_t)noise)) & 3) return IMGXF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (mode) {
        case IMGXF_F32_BRIGHTNESS: return launch_f32_map<IMGXF_F32_BRIGHTNESS>(src, nullptr, dst, mask, count, p0, p1, st);
        case IMGXF_F32_CONTRAST: return launch_f32_map<IMGXF_F32_CONTRAST>(src, nullptr, dst, mask, count, p0, p1, st);
        case IMGXF_F32_NOISE:
            if (!noise) return IMGXF_ERR_NULL;
            return launch_f32_map<IMGXF_F32_NOISE>(src, noise, dst, mask, count, p0, p1, st);
        default: return IMGXF_ERR_ARG;
    }
}
