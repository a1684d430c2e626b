// Colour-space maps and histogram equalisation of TransformationPool.histogram_equalization
// (/root/reference/pipenline/cifar_image_transformations.py:122-129):
//   cv2.cvtColor(RGB2YUV) -> cv2.equalizeHist on Y -> cv2.cvtColor(YUV2RGB).
// PARITY UNPINNED: OpenCV is not installed here.  The arithmetic follows OpenCV's 8-bit integer
// definitions (imgproc color_yuv: yuv_shift = 14, R2Y/G2Y/B2Y = 4899/9617/1868, B2U = 8061,
// R2V = 14369, U2B/U2G/V2G/V2R = 33292/-6472/-9519/18678, CV_DESCALE rounding, saturate_cast;
// histogram.cpp equalizeHist: scale = 255.f / (total - hist[first]), lut = saturate(sum * scale)).
#include "imgxf_common.h"

namespace imgxf {

__device__ __forceinline__ int descale14(int x) { return (x + (1 << 13)) >> 14; }
__device__ __forceinline__ u32 sat8(int v) { return (u32)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

struct Rgb2Yuv {
    __device__ __forceinline__ void operator()(const u32 (&c)[3], u32 (&o)[3]) const {
        const int R = (int)c[0], G = (int)c[1], B = (int)c[2];
        const int Y = descale14(R * 4899 + G * 9617 + B * 1868);
        const int V = descale14((R - Y) * 14369 + (128 << 14));
        const int U = descale14((B - Y) * 8061 + (128 << 14));
        o[0] = sat8(Y); o[1] = sat8(U); o[2] = sat8(V);
    }
};
struct Yuv2Rgb {
    __device__ __forceinline__ void operator()(const u32 (&c)[3], u32 (&o)[3]) const {
        const int Y = (int)c[0], U = (int)c[1] - 128, V = (int)c[2] - 128;
        o[2] = sat8(Y + descale14(U * 33292));
        o[1] = sat8(Y + descale14(U * -6472 + V * -9519));
        o[0] = sat8(Y + descale14(V * 18678));
    }
};

// 16 pixels (48 bytes) per lane, dense 16-byte accesses when the view allows it
template <class Op>
__global__ __launch_bounds__(256) void pixel3_map_kernel(View s, View d, Op op) {
    const int ngrp = (d.w + 15) >> 4;
    const int64_t total = (int64_t)d.n * d.h * ngrp;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int g = (int)(t % ngrp);
        const int64_t r = t / ngrp;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const int x0 = g << 4;
        const int np = min(16, d.w - x0);
        const u8* sp = s.row(f, y) + x0 * 3;
        u8* dp = d.row(f, y) + x0 * 3;
        u32 in[12], out[12];
        const bool vec = np == 16 && ((((uintptr_t)sp) | ((uintptr_t)dp)) & 15) == 0;
        if (vec) {
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const uint4 q = ((const uint4*)sp)[b];
                in[4 * b] = q.x; in[4 * b + 1] = q.y; in[4 * b + 2] = q.z; in[4 * b + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) in[k] = 0;
            for (int e = 0; e < np * 3; ++e) {
                const u32 v = sp[e];
#pragma unroll
                for (int k = 0; k < 12; ++k) if ((e >> 2) == k) in[k] |= v << (8 * (e & 3));
            }
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) out[k] = 0;
#pragma unroll
        for (int px = 0; px < 16; ++px) {
            u32 c[3], o[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) { const int b = px * 3 + j; c[j] = (in[b >> 2] >> (8 * (b & 3))) & 0xffu; }
            op(c, o);
#pragma unroll
            for (int j = 0; j < 3; ++j) { const int b = px * 3 + j; out[b >> 2] |= o[j] << (8 * (b & 3)); }
        }
        if (vec) {
#pragma unroll
            for (int b = 0; b < 3; ++b) ((uint4*)dp)[b] = make_uint4(out[4 * b], out[4 * b + 1], out[4 * b + 2], out[4 * b + 3]);
        } else {
            for (int e = 0; e < np * 3; ++e) {
                u32 w = 0;
#pragma unroll
                for (int k = 0; k < 12; ++k) if ((e >> 2) == k) w = out[k];
                dp[e] = (u8)(w >> (8 * (e & 3)));
            }
        }
    }
}

// cv2.equalizeHist table of channel `ch` from hist[n][c][256]; the other channels get the
// identity so one table-apply pass maps the whole interleaved frame
__global__ void cv_equalize_lut_kernel(const u32* hist, u8* lut, int nframes, int c, int ch) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nframes) return;
    for (int j = 0; j < c; ++j)
        for (int i = 0; i < 256; ++i) lut[((int64_t)f * c + j) * 256 + i] = (u8)i;
    const u32* h = hist + ((int64_t)f * c + ch) * 256;
    u8* l = lut + ((int64_t)f * c + ch) * 256;
    long long total = 0;
    for (int i = 0; i < 256; ++i) total += h[i];
    int i = 0;
    while (i < 255 && !h[i]) ++i;
    if ((long long)h[i] == total) {                         // one level: dst.setTo(i)
        for (int k = 0; k < 256; ++k) l[k] = (u8)i;
        return;
    }
    const float scale = 255.0f / (float)(total - (long long)h[i]);
    int sum = 0;
    l[i++] = 0;
    for (; i < 256; ++i) {
        sum += (int)h[i];
        l[i] = (u8)sat_u8_rne((float)sum * scale);          // saturate_cast<uchar>(float): cvRound
    }
}

static inline unsigned cs_grid(int64_t total) {
    int64_t b = (total + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

} // namespace imgxf

using namespace imgxf;

extern "C" int imgxf_channel_histogram_u8(const imgxf_view* src, uint32_t* hist, void* stream);
extern "C" int imgxf_lut_device_u8(const imgxf_view* src, const imgxf_view* dst, const uint8_t* lut_dev, void* stream);

IMGXF_API int imgxf_rgb2yuv_u8(const imgxf_view* src, const imgxf_view* dst, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!same_geometry(src, dst) || src->c != 3) return IMGXF_ERR_SHAPE;
    if (empty_view(dst)) return IMGXF_OK;
    const View d = make_view(dst);
    hipLaunchKernelGGL((pixel3_map_kernel<Rgb2Yuv>), dim3(cs_grid((int64_t)d.n * d.h * ((d.w + 15) >> 4))), dim3(256), 0,
                       (hipStream_t)stream, make_view(src), d, Rgb2Yuv());
    return launch_status();
}

IMGXF_API int imgxf_yuv2rgb_u8(const imgxf_view* src, const imgxf_view* dst, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!same_geometry(src, dst) || src->c != 3) return IMGXF_ERR_SHAPE;
    if (empty_view(dst)) return IMGXF_OK;
    const View d = make_view(dst);
    hipLaunchKernelGGL((pixel3_map_kernel<Yuv2Rgb>), dim3(cs_grid((int64_t)d.n * d.h * ((d.w + 15) >> 4))), dim3(256), 0,
                       (hipStream_t)stream, make_view(src), d, Yuv2Rgb());
    return launch_status();
}

IMGXF_API int imgxf_equalize_hist_cv_u8(const imgxf_view* src, const imgxf_view* dst, int channel, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!same_geometry(src, dst)) return IMGXF_ERR_SHAPE;
    if (channel < 0 || channel >= src->c) return IMGXF_ERR_ARG;
    if (empty_view(dst)) return IMGXF_OK;
    if (!workspace) return IMGXF_ERR_NULL;
    const size_t ntab = (size_t)dst->n * dst->c;
    if (workspace_bytes < ntab * 256 * 5 || (((uintptr_t)workspace) & 3)) return IMGXF_ERR_WORKSPACE;
    u32* hist = (u32*)workspace;
    u8* lut = (u8*)workspace + ntab * 256 * 4;
    IMGXF_CHECK(imgxf_channel_histogram_u8(src, hist, stream));
    hipLaunchKernelGGL(cv_equalize_lut_kernel, dim3((unsigned)((dst->n + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                       hist, lut, dst->n, dst->c, channel);
    IMGXF_CHECK(launch_status());
    return imgxf_lut_device_u8(src, dst, lut, stream);
}
