// Lookup-table point ops of the AugMix operation set (/root/reference/fall_2025/AugMix.py:31,36,37:
// ImageOps.posterize / equalize / solarize) and the per-channel histogram behind
// ImageOps.equalize and the Shannon-entropy feature
// (/root/reference/fall_2025/Initial_Experiments.py:95-113).  All integer, bit-exact.
//   chist_kernel        [n][c][256] uint32 histogram, LDS-privatised per workgroup
//   equalize_lut_kernel ImageOps.equalize's table from one channel histogram (one lane per table)
//   lut_apply_kernel    dst = lut[frame][channel][src]; tables staged in LDS, 16 bytes per lane
#include "imgxf_common.h"
#include <string.h>

namespace imgxf {

__global__ __launch_bounds__(256) void chist_kernel(View s, u32* hist) {
    // 4 sub-histograms per channel (lane & 3 picks one) spread same-bin atomics of neighbouring lanes
    constexpr int NS = 4;
    __shared__ u32 h[NS * 4 * 256];
    const int C = s.c;
    for (int i = threadIdx.x; i < NS * C * 256; i += 256) h[i] = 0;
    __syncthreads();
    const int f = blockIdx.y;
    u32* hs = h + (threadIdx.x & (NS - 1)) * C * 256;
    const int rowbytes = s.w * C;
    const int64_t total = (int64_t)s.h * rowbytes;
    const bool vec = (rowbytes % 4 == 0) && ((((uintptr_t)s.p) | (uintptr_t)s.rs | (uintptr_t)s.fs) & 3) == 0;
    if (vec && (int64_t)s.h * (rowbytes >> 2) < 0x7fffffff) {
        const u32 rowwords = (u32)rowbytes >> 2, words = (u32)s.h * rowwords;       // 32-bit index arithmetic
        for (u32 t = blockIdx.x * 256u + threadIdx.x; t < words; t += gridDim.x * 256u) {
            const u32 y = t / rowwords, xw = t - y * rowwords;
            const u32 v = ((const u32*)s.row(f, (int)y))[xw];
            int ch = (int)((xw * 4u) % (u32)C);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                atomicAdd(&hs[ch * 256 + ((v >> (8 * b)) & 0xffu)], 1u);
                ch = ch + 1 == C ? 0 : ch + 1;
            }
        }
    } else if (vec) {
        const int rowwords = rowbytes >> 2;
        const int64_t words = (int64_t)s.h * rowwords;
        for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < words; t += (int64_t)gridDim.x * 256) {
            const int xw = (int)(t % rowwords), y = (int)(t / rowwords);
            const u32 v = ((const u32*)s.row(f, y))[xw];
            int ch = (xw * 4) % C;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                atomicAdd(&hs[ch * 256 + ((v >> (8 * b)) & 0xffu)], 1u);
                ch = ch + 1 == C ? 0 : ch + 1;
            }
        }
    } else {
        for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
            const int xb = (int)(t % rowbytes), y = (int)(t / rowbytes);
            atomicAdd(&hs[(xb % C) * 256 + s.row(f, y)[xb]], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 256; i += 256) {
        u32 sum = 0;
#pragma unroll
        for (int k = 0; k < NS; ++k) sum += h[k * C * 256 + i];
        if (sum) atomicAdd(&hist[(int64_t)f * C * 256 + i], sum);
    }
}

// PIL/ImageOps.py equalize(): histo = non-zero bins; step = (sum(histo) - histo[-1]) // 255;
// identity when there is at most one non-zero bin or step == 0; else n = step // 2 and
// lut[i] = n // step, n += h[i].  image.point() clips table entries to 0..255 (getlist/CLIP8).
__global__ void equalize_lut_kernel(const u32* hist, u8* lut, int ntables) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntables) return;
    const u32* h = hist + (int64_t)t * 256;
    u8* l = lut + (int64_t)t * 256;
    unsigned long long sum = 0;
    u32 lastnz = 0;
    int nnz = 0;
    for (int i = 0; i < 256; ++i) {
        sum += h[i];
        if (h[i]) { lastnz = h[i]; ++nnz; }
    }
    const unsigned long long step = nnz <= 1 ? 0 : (sum - lastnz) / 255;
    if (step == 0) {
        for (int i = 0; i < 256; ++i) l[i] = (u8)i;
        return;
    }
    unsigned long long n = step / 2;
    for (int i = 0; i < 256; ++i) {
        const unsigned long long v = n / step;
        l[i] = v > 255 ? (u8)255 : (u8)v;
        n += h[i];
    }
}

struct LutArg { u8 t[4 * 256]; };

// lut: device tables [n][c][256] (per-frame) when `dev` is set, else the by-value table for all frames
__global__ __launch_bounds__(256) void lut_apply_kernel(View s, View d, const u8* dev, LutArg arg) {
    __shared__ __attribute__((aligned(16))) u8 tab[4 * 256];
    const int C = d.c;
    const int f = blockIdx.y;
    for (int i = threadIdx.x; i < C * 256; i += 256) tab[i] = dev ? dev[(int64_t)f * C * 256 + i] : arg.t[i];
    __syncthreads();
    const int rowbytes = d.w * C;
    const bool vec = (rowbytes % 4 == 0) &&
                     ((((uintptr_t)s.p) | (uintptr_t)s.rs | (uintptr_t)s.fs | (uintptr_t)d.p | (uintptr_t)d.rs | (uintptr_t)d.fs) & 3) == 0;
    const bool vec16 = (rowbytes % 16 == 0) && (int64_t)d.h * (rowbytes >> 4) < 0x7fffffff &&
                       ((((uintptr_t)s.p) | (uintptr_t)s.rs | (uintptr_t)s.fs | (uintptr_t)d.p | (uintptr_t)d.rs | (uintptr_t)d.fs) & 15) == 0;
    if (vec16) {
        // 16 bytes per lane and 32-bit index arithmetic (the dword loop below spends more on its 64-bit
        // division per 4 bytes than on the look-ups); the table's 64 dwords sit in 64 different banks, so the
        // byte reads never conflict
        const u32 nch = (u32)rowbytes >> 4, total = (u32)d.h * nch;
        for (u32 t = blockIdx.x * 256u + threadIdx.x; t < total; t += gridDim.x * 256u) {
            const u32 y = t / nch, ck = t - y * nch;
            const uint4 q = *(const uint4*)(s.row(f, (int)y) + (ck << 4));
            const u32 w[4] = {q.x, q.y, q.z, q.w};
            int ch = (int)((ck << 4) % (u32)C);
            u32 o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u32 acc = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    acc |= (u32)tab[ch * 256 + ((w[k] >> (8 * b)) & 0xffu)] << (8 * b);
                    ch = ch + 1 == C ? 0 : ch + 1;
                }
                o[k] = acc;
            }
            *(uint4*)(d.row(f, (int)y) + (ck << 4)) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    } else if (vec) {
        const int rowwords = rowbytes >> 2;
        const int64_t words = (int64_t)d.h * rowwords;
        for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < words; t += (int64_t)gridDim.x * 256) {
            const int xw = (int)(t % rowwords), y = (int)(t / rowwords);
            const u32 v = ((const u32*)s.row(f, y))[xw];
            int ch = (xw * 4) % C;
            u32 o = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                o |= (u32)tab[ch * 256 + ((v >> (8 * b)) & 0xffu)] << (8 * b);
                ch = ch + 1 == C ? 0 : ch + 1;
            }
            ((u32*)d.row(f, y))[xw] = o;
        }
    } else {
        const int64_t total = (int64_t)d.h * rowbytes;
        for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
            const int xb = (int)(t % rowbytes), y = (int)(t / rowbytes);
            d.row(f, y)[xb] = tab[(xb % C) * 256 + s.row(f, y)[xb]];
        }
    }
}

static inline unsigned blocks_for(int64_t items) {
    int64_t b = (items + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

} // namespace imgxf

using namespace imgxf;

IMGXF_API int imgxf_channel_histogram_u8(const imgxf_view* src, uint32_t* hist, void* stream) {
    IMGXF_CHECK(check_view(src));
    if (!hist) return IMGXF_ERR_NULL;
    if (src->n == 0) return IMGXF_OK;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(hist, 0, (size_t)src->n * src->c * 256 * sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
    if (empty_view(src)) return IMGXF_OK;
    const View s = make_view(src);
    if (s.n > 65535) return IMGXF_ERR_SHAPE;
    // (c * 256 global atomics per workgroup: a few thousand workgroups per launch, not per frame)
    unsigned hb = blocks_for((int64_t)s.h * s.rowbytes() / 16);
    const unsigned hcap = (unsigned)(4096 / s.n < 32 ? 32 : (4096 / s.n > 512 ? 512 : 4096 / s.n));
    if (hb > hcap) hb = hcap;
    hipLaunchKernelGGL(chist_kernel, dim3(hb, (unsigned)s.n), dim3(256), 0, st, s, hist);
    return launch_status();
}

IMGXF_API int imgxf_lut_u8(const imgxf_view* src, const imgxf_view* dst, const uint8_t* lut, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!lut) return IMGXF_ERR_NULL;
    if (!same_geometry(src, dst)) return IMGXF_ERR_SHAPE;
    if (empty_view(dst)) return IMGXF_OK;
    const View s = make_view(src), d = make_view(dst);
    if (d.n > 65535) return IMGXF_ERR_SHAPE;
    LutArg arg;
    memset(&arg, 0, sizeof(arg));
    memcpy(arg.t, lut, (size_t)d.c * 256);
    hipLaunchKernelGGL(lut_apply_kernel, dim3(blocks_for((int64_t)d.h * d.rowbytes() / 16), (unsigned)d.n), dim3(256), 0,
                       (hipStream_t)stream, s, d, (const u8*)nullptr, arg);
    return launch_status();
}

IMGXF_API int imgxf_equalize_u8(const imgxf_view* src, const imgxf_view* dst, void* workspace, size_t workspace_bytes,
                                void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!same_geometry(src, dst)) return IMGXF_ERR_SHAPE;
    if (empty_view(dst)) return IMGXF_OK;
    const View s = make_view(src), d = make_view(dst);
    if (d.n > 65535) return IMGXF_ERR_SHAPE;
    const size_t ntab = (size_t)d.n * d.c;
    if (!workspace) return IMGXF_ERR_NULL;
    if (workspace_bytes < ntab * 256 * 5 || (((uintptr_t)workspace) & 3)) return IMGXF_ERR_WORKSPACE;
    u32* hist = (u32*)workspace;
    u8* lut = (u8*)workspace + ntab * 256 * 4;
    IMGXF_CHECK(imgxf_channel_histogram_u8(src, hist, stream));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(equalize_lut_kernel, dim3((unsigned)((ntab + 63) / 64)), dim3(64), 0, st, hist, lut, (int)ntab);
    LutArg arg;
    memset(&arg, 0, sizeof(arg));
    hipLaunchKernelGGL(lut_apply_kernel, dim3(blocks_for((int64_t)d.h * d.rowbytes() / 16), (unsigned)d.n), dim3(256), 0, st,
                       s, d, (const u8*)lut, arg);
    return launch_status();
}

// internal (not in imgxf.h): table apply with per-frame device tables [n][c][256]
extern "C" __attribute__((visibility("hidden"))) int imgxf_lut_device_u8(const imgxf_view* src, const imgxf_view* dst,
                                                                         const uint8_t* lut_dev, void* stream) {
    const View s = make_view(src), d = make_view(dst);
    if (d.n > 65535) return IMGXF_ERR_SHAPE;
    LutArg arg;
    memset(&arg, 0, sizeof(arg));
    hipLaunchKernelGGL(lut_apply_kernel, dim3(blocks_for((int64_t)d.h * d.rowbytes() / 16), (unsigned)d.n), dim3(256), 0,
                       (hipStream_t)stream, s, d, (const u8*)lut_dev, arg);
    return launch_status();
}
