// Shared host/device helpers for libimgxf (gfx950 only — no other backend exists).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "imgxf.h"
#include "knobs.h"

#define IMGXF_API extern "C" __attribute__((visibility("default")))

namespace imgxf {

typedef uint8_t u8;
typedef uint32_t u32;

// Device-side copy of imgxf_view with a typed pointer.
struct View {
    u8* p;
    int n, h, w, c;
    int64_t rs, fs;
    __host__ __device__ inline int64_t rowbytes() const { return (int64_t)w * c; }
    __device__ inline u8* row(int f, int y) const { return p + (int64_t)f * fs + (int64_t)y * rs; }
};

inline View make_view(const imgxf_view* v) {
    View o;
    o.p = (u8*)v->data; o.n = v->n; o.h = v->h; o.w = v->w; o.c = v->c;
    o.rs = v->row_stride; o.fs = v->frame_stride;
    return o;
}

// Validate one view: non-null, positive dims, strides large enough for `elem` bytes/sample.
inline int check_view(const imgxf_view* v, int elem = 1) {
    if (!v) return IMGXF_ERR_NULL;
    if (v->n < 0 || v->h < 0 || v->w < 0 || v->c < 1 || v->c > 4) return IMGXF_ERR_SHAPE;
    if (!v->data && v->n != 0 && v->h != 0 && v->w != 0) return IMGXF_ERR_NULL;   // empty views may be NULL
    if (v->w > 32767 || v->h > 32767) return IMGXF_ERR_SHAPE; // 16.16 fixed-point samplers
    int64_t rb = (int64_t)v->w * v->c * elem;
    if (v->row_stride < rb) return IMGXF_ERR_SHAPE;
    if (v->n > 1 && v->frame_stride < v->row_stride * (int64_t)v->h) return IMGXF_ERR_SHAPE;
    return IMGXF_OK;
}

inline bool same_geometry(const imgxf_view* a, const imgxf_view* b) {
    return a->n == b->n && a->h == b->h && a->w == b->w && a->c == b->c;
}
inline bool same_nhw(const imgxf_view* a, const imgxf_view* b) {
    return a->n == b->n && a->h == b->h && a->w == b->w;
}
inline bool empty_view(const imgxf_view* v) { return v->n == 0 || v->h == 0 || v->w == 0; }

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? IMGXF_OK : (int)e;
}

#define IMGXF_CHECK(expr)                 \
    do {                                  \
        int _rc = (expr);                 \
        if (_rc != IMGXF_OK) return _rc;  \
    } while (0)

__device__ __forceinline__ int reflect101(int i, int n) {
    // valid for any i when n >= 1 (period 2n-2)
    if (n == 1) return 0;
    int p = 2 * n - 2;
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - i : i;
}
__device__ __forceinline__ int reflect_sym(int i, int n) {
    int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - 1 - i : i;
}
__device__ __forceinline__ int border_index(int i, int n, int border) {
    return border == IMGXF_BORDER_REFLECT ? reflect_sym(i, n) : reflect101(i, n);
}

// saturate_cast<uchar>(float): round half to even, clamp to [0,255]
__device__ __forceinline__ u32 sat_u8_rne(float v) {
    float r = __builtin_rintf(v);
    r = fminf(fmaxf(r, 0.0f), 255.0f);
    return (u32)r;
}

} // namespace imgxf
