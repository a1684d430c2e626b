// Integer geometry: fill / crop+paste / quarter-turn rotations.  Bodies behind
//   Image.new + Image.crop + Image.paste   /root/reference/transformation.py:187-193,287-305
//   Image.transpose(ROTATE_90/180/270)     fast paths of Image.rotate, PIL/Image.py:2513-2521
#include "imgxf_common.h"
#include <string.h>

namespace imgxf {

struct Fill4 { u8 v[4]; };

// The byte pattern of a filled row has period lcm(16, c) = 16 or 48 bytes: the host passes the
// three 16-byte chunks of one period and the kernel only picks chunk (index % 3)
struct FillPat { uint4 q[3]; };

__global__ __launch_bounds__(256) void fill_kernel(View d, FillPat pat) {
    const int rowbytes = d.w * d.c;
    const int nchunks = (rowbytes + 15) >> 4;
    const int64_t total = (int64_t)d.n * d.h * nchunks;
    const bool al = ((((uintptr_t)d.p) | (uintptr_t)d.rs | (uintptr_t)d.fs) & 15) == 0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int ck = (int)(t % nchunks);
        const int64_t r = t / nchunks;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const int xb = ck << 4;
        u8* dp = d.row(f, y) + xb;
        const int nv = min(16, rowbytes - xb);
        const int ph = ck % 3;
        const uint4 v = ph == 0 ? pat.q[0] : (ph == 1 ? pat.q[1] : pat.q[2]);
        if (nv == 16 && al) {
            *(uint4*)dp = v;
        } else {
            const u32 o[4] = {v.x, v.y, v.z, v.w};
            for (int e = 0; e < nv; ++e) dp[e] = (u8)(o[e >> 2] >> (8 * (e & 3)));
        }
    }
}

// rectangle copy expressed on byte runs: rbytes = rw*c bytes per row
__global__ __launch_bounds__(256) void copy_rect_kernel(View s, View d, int sxb, int sy, int dxb,
                                                        int dy, int rbytes, int rh) {
    const int nchunks = (rbytes + 15) >> 4;
    const int64_t total = (int64_t)d.n * rh * nchunks;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int ck = (int)(t % nchunks);
        const int64_t r = t / nchunks;
        const int y = (int)(r % rh), f = (int)(r / rh);
        const int xb = ck << 4;
        const u8* sp = s.row(f, sy + y) + sxb + xb;
        u8* dp = d.row(f, dy + y) + dxb + xb;
        const int nv = min(16, rbytes - xb);
        if (nv == 16 && ((((uintptr_t)sp) | ((uintptr_t)dp)) & 15) == 0) {
            *(uint4*)dp = *(const uint4*)sp;
        } else if (nv == 16 && ((((uintptr_t)sp) | ((uintptr_t)dp)) & 3) == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) ((u32*)dp)[k] = ((const u32*)sp)[k];
        } else {
            for (int e = 0; e < nv; ++e) dp[e] = sp[e];
        }
    }
}

// dst(y,x) = src(sy,sx) with quarter turns counter-clockwise (PIL ROTATE_90 = ccw)
__global__ __launch_bounds__(256) void rot90_kernel(View s, View d, int turns) {
    const int64_t total = (int64_t)d.n * d.h * d.w;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int x = (int)(t % d.w);
        const int64_t r = t / d.w;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        int sx, sy;
        if (turns == 1) { sx = s.w - 1 - y; sy = x; }            // np.rot90(a, 1)
        else if (turns == 2) { sx = s.w - 1 - x; sy = s.h - 1 - y; }
        else { sx = y; sy = s.h - 1 - x; }                        // np.rot90(a, 3)
        const u8* sp = s.row(f, sy) + sx * s.c;
        u8* dp = d.row(f, y) + x * d.c;
        for (int j = 0; j < d.c; ++j) dp[j] = sp[j];
    }
}

// Image.transpose(FLIP_LEFT_RIGHT / FLIP_TOP_BOTTOM): dst(y,x) = src(y, w-1-x) or src(h-1-y, x)
__global__ __launch_bounds__(256) void flip_kernel(View s, View d, int mode) {
    const int64_t total = (int64_t)d.n * d.h * d.w;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int x = (int)(t % d.w);
        const int64_t r = t / d.w;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const int sx = mode == 0 ? s.w - 1 - x : x, sy = mode == 0 ? y : s.h - 1 - y;
        const u8* sp = s.row(f, sy) + sx * s.c;
        u8* dp = d.row(f, y) + x * d.c;
        for (int j = 0; j < d.c; ++j) dp[j] = sp[j];
    }
}

static inline unsigned grid_for(int64_t total) {
    int64_t blocks = (total + 255) / 256;
    return (unsigned)(blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks));
}

} // namespace imgxf

using namespace imgxf;

IMGXF_API int imgxf_fill_u8(const imgxf_view* dst, const uint8_t* color, void* stream) {
    IMGXF_CHECK(check_view(dst));
    if (!color) return IMGXF_ERR_NULL;
    if (empty_view(dst)) return IMGXF_OK;
    FillPat pat;
    uint8_t bytes[48];
    for (int i = 0; i < 48; ++i) bytes[i] = color[i % dst->c];     // 48 is a multiple of c = 1..4
    memcpy(&pat, bytes, sizeof(bytes));
    const View d = make_view(dst);
    const int64_t total = (int64_t)d.n * d.h * ((d.rowbytes() + 15) >> 4);
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, d, pat);
    return launch_status();
}

IMGXF_API int imgxf_copy_rect_u8(const imgxf_view* src, const imgxf_view* dst, int sx, int sy,
                                 int dx, int dy, int rw, int rh, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (src->n != dst->n || src->c != dst->c) return IMGXF_ERR_SHAPE;
    if (rw < 0 || rh < 0 || sx < 0 || sy < 0 || dx < 0 || dy < 0) return IMGXF_ERR_ARG;
    if (sx + rw > src->w || sy + rh > src->h || dx + rw > dst->w || dy + rh > dst->h) return IMGXF_ERR_ARG;
    if (rw == 0 || rh == 0 || dst->n == 0) return IMGXF_OK;
    const int c = src->c;
    const int64_t total = (int64_t)dst->n * rh * (((int64_t)rw * c + 15) >> 4);
    hipLaunchKernelGGL(copy_rect_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       make_view(src), make_view(dst), sx * c, sy, dx * c, dy, rw * c, rh);
    return launch_status();
}

IMGXF_API int imgxf_rot90_u8(const imgxf_view* src, const imgxf_view* dst, int quarter_turns_ccw,
                             void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (quarter_turns_ccw < 1 || quarter_turns_ccw > 3) return IMGXF_ERR_ARG;
    if (src->n != dst->n || src->c != dst->c) return IMGXF_ERR_SHAPE;
    const bool swap = quarter_turns_ccw != 2;
    if ((swap && (dst->h != src->w || dst->w != src->h)) || (!swap && (dst->h != src->h || dst->w != src->w)))
        return IMGXF_ERR_SHAPE;
    if (empty_view(dst)) return IMGXF_OK;
    const View d = make_view(dst);
    hipLaunchKernelGGL(rot90_kernel, dim3(grid_for((int64_t)d.n * d.h * d.w)), dim3(256), 0,
                       (hipStream_t)stream, make_view(src), d, quarter_turns_ccw);
    return launch_status();
}

IMGXF_API int imgxf_flip_u8(const imgxf_view* src, const imgxf_view* dst, int mode, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (mode < 0 || mode > 1) return IMGXF_ERR_ARG;
    if (!same_geometry(src, dst)) return IMGXF_ERR_SHAPE;
    if (empty_view(dst)) return IMGXF_OK;
    const View d = make_view(dst);
    hipLaunchKernelGGL(flip_kernel, dim3(grid_for((int64_t)d.n * d.h * d.w)), dim3(256), 0,
                       (hipStream_t)stream, make_view(src), d, mode);
    return launch_status();
}
