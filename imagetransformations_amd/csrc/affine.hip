// Affine resampler: the body behind Image.transform(size, AFFINE, m, resample, fillcolor)
//   NEAREST  -> Image.rotate      /root/reference/transformation.py:200   (libImaging affine_fixed)
//   BICUBIC  -> apply_shear       /root/reference/transformation.py:217-224
//   BILINEAR -> benchmark configs[3] (rotate 30 deg + 1.5x), SURVEY §8a row a2'
// Output-stationary gather: every lane owns 4 consecutive output pixels of one row (12
// bytes for RGB, written as three aligned dwords), a wave owns 256 consecutive pixels, a
// workgroup a 256 x 4 pixel patch, so the source footprint of a workgroup is a thin
// rotated strip that stays in the XCD's L2 while neighbouring patches reuse it.
#include "imgxf_common.h"
#include <math.h>
#include <string.h>

namespace imgxf {

struct AffineParams {
    double m[6];
    int fx[6];   // 16.16 fixed-point matrix for NEAREST (affine_fixed)
    u8 fill[4];
};

// ---- arithmetic policies -------------------------------------------------------------
// Precise: fp64 with every multiply/add rounded separately (no FMA contraction), i.e. the
// exact sequence libImaging's C code performs on x86-64 -> bit-identical to Pillow.
struct PreciseArith {
    typedef double T;
    static __device__ __forceinline__ T mul(T a, T b) { return __dmul_rn(a, b); }
    static __device__ __forceinline__ T add(T a, T b) { return __dadd_rn(a, b); }
    static __device__ __forceinline__ T sub(T a, T b) { return __dsub_rn(a, b); }
};
// Fast: fp32 (<= 1e-5 relative before truncation).
struct FastArith {
    typedef float T;
    static __device__ __forceinline__ T mul(T a, T b) { return a * b; }
    static __device__ __forceinline__ T add(T a, T b) { return a + b; }
    static __device__ __forceinline__ T sub(T a, T b) { return a - b; }
};

// BILINEAR(v,a,b,d) = a + (b-a)*d
__device__ __forceinline__ double lerp_t(PreciseArith, double a, double b, double d) {
    return a + (b - a) * d;          // built with -ffp-contract=off: two roundings, as in C on x86-64
}
__device__ __forceinline__ float lerp_t(FastArith, float a, float b, float d) {
    return fmaf(b - a, d, a);
}
template <class A>
__device__ __forceinline__ typename A::T lerp(typename A::T a, typename A::T b, typename A::T d) {
    return lerp_t(A(), a, b, d);
}

template <class A>
__device__ __forceinline__ typename A::T cubic(typename A::T v1, typename A::T v2, typename A::T v3,
                                               typename A::T v4, typename A::T d) {
    typedef typename A::T T;
    const T p1 = v2;
    const T p2 = A::add(-v1, v3);
    const T p3 = A::sub(A::add(A::mul((T)2, A::sub(v1, v2)), v3), v4);
    const T p4 = A::add(A::sub(A::add(-v1, v2), v3), v4);
    return A::add(p1, A::mul(d, A::add(p2, A::mul(d, A::add(p3, A::mul(d, p4))))));
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <int C, int FILTER, class A>
__global__ __launch_bounds__(256) void affine_kernel(View s, View d, AffineParams P, View dbg) {
    typedef typename A::T T;
    const int xg = blockIdx.x * 64 + threadIdx.x;   // group of 4 output pixels
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int f = blockIdx.z;
    const int x0 = xg * 4;
    if (y >= d.h || x0 >= d.w) return;
    const u8* sp = s.p + (int64_t)f * s.fs;
    u8 out[4 * C];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = x0 + k;
        bool ok;
        u8 px[C];
        if (FILTER == IMGXF_FILTER_NEAREST) {
            // int arithmetic wraps exactly like the C `int` accumulators of affine_fixed
            const int xx = (int)((u32)P.fx[2] + (u32)P.fx[1] * (u32)y + (u32)P.fx[0] * (u32)x);
            const int yy = (int)((u32)P.fx[5] + (u32)P.fx[4] * (u32)y + (u32)P.fx[3] * (u32)x);
            const int xin = xx >> 16, yin = yy >> 16;
            ok = xin >= 0 && xin < s.w && yin >= 0 && yin < s.h;
            if (ok) {
                const u8* q = sp + (int64_t)yin * s.rs + xin * C;
#pragma unroll
                for (int j = 0; j < C; ++j) px[j] = q[j];
            }
        } else {
            // coordinates always in fp64, un-contracted: a0*xin + a1*yin + a2
            const double xc = (double)x + 0.5, yc = (double)y + 0.5;
            double xin = __dadd_rn(__dadd_rn(__dmul_rn(P.m[0], xc), __dmul_rn(P.m[1], yc)), P.m[2]);
            double yin = __dadd_rn(__dadd_rn(__dmul_rn(P.m[3], xc), __dmul_rn(P.m[4], yc)), P.m[5]);
            ok = xin >= 0.0 && xin < (double)s.w && yin >= 0.0 && yin < (double)s.h;
            if (ok) {
                xin -= 0.5; yin -= 0.5;
                const double xfl = floor(xin), yfl = floor(yin);
                const int xi = (int)xfl, yi = (int)yfl;
                const T dx = (T)(xin - xfl), dy = (T)(yin - yfl);
                T v[C];
                if (FILTER == IMGXF_FILTER_BILINEAR) {
                    const int xa = clampi(xi, 0, s.w - 1) * C, xb = clampi(xi + 1, 0, s.w - 1) * C;
                    const u8* r0 = sp + (int64_t)clampi(yi, 0, s.h - 1) * s.rs;
                    const bool has1 = (yi + 1 >= 0) && (yi + 1 < s.h);
                    const u8* r1 = sp + (int64_t)(has1 ? yi + 1 : 0) * s.rs;
#pragma unroll
                    for (int j = 0; j < C; ++j) {
                        const T v1 = lerp<A>((T)r0[xa + j], (T)r0[xb + j], dx);
                        const T v2 = has1 ? lerp<A>((T)r1[xa + j], (T)r1[xb + j], dx) : v1;
                        v[j] = lerp<A>(v1, v2, dy);
                    }
#pragma unroll
                    for (int j = 0; j < C; ++j) px[j] = (u8)(int)v[j];   // (UINT8)v truncation
                } else {
                    int xs[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) xs[t] = clampi(xi - 1 + t, 0, s.w - 1) * C;
                    T rowv[4][C];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int yy = yi - 1 + t;
                        const bool inr = (t == 0) || (yy >= 0 && yy < s.h);
                        const u8* r = sp + (int64_t)clampi(yy, 0, s.h - 1) * s.rs;
#pragma unroll
                        for (int j = 0; j < C; ++j) {
                            if (inr)
                                rowv[t][j] = cubic<A>((T)r[xs[0] + j], (T)r[xs[1] + j],
                                                      (T)r[xs[2] + j], (T)r[xs[3] + j], dx);
                            else
                                rowv[t][j] = rowv[t - 1 < 0 ? 0 : t - 1][j];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < C; ++j) {
                        v[j] = cubic<A>(rowv[0][j], rowv[1][j], rowv[2][j], rowv[3][j], dy);
                        px[j] = v[j] <= (T)0 ? (u8)0 : (v[j] >= (T)255 ? (u8)255 : (u8)(int)v[j]);
                    }
                }
                if (dbg.p && x < d.w) {
                    float* fp = (float*)dbg.row(f, y) + x * C;
#pragma unroll
                    for (int j = 0; j < C; ++j) fp[j] = (float)v[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < C; ++j) out[k * C + j] = ok ? px[j] : P.fill[j];
        if (!ok && FILTER != IMGXF_FILTER_NEAREST && dbg.p && x < d.w) {
            float* fp = (float*)dbg.row(f, y) + x * C;
#pragma unroll
            for (int j = 0; j < C; ++j) fp[j] = (float)P.fill[j];
        }
    }
    u8* dp = d.row(f, y) + x0 * C;
    const int npx = min(4, d.w - x0);
    if (npx == 4 && ((((uintptr_t)dp) & 3) == 0)) {
#pragma unroll
        for (int q = 0; q < C; ++q) {
            ((u32*)dp)[q] = (u32)out[4 * q] | ((u32)out[4 * q + 1] << 8) | ((u32)out[4 * q + 2] << 16) |
                            ((u32)out[4 * q + 3] << 24);
        }
    } else {
        for (int e = 0; e < npx * C; ++e) {
            u8 v = 0;
#pragma unroll
            for (int k = 0; k < 4 * C; ++k) if (k == e) v = out[k];
            dp[e] = v;
        }
    }
}

static inline int fix16(double v) {
    const double t = v * 65536.0 + 0.5;
    return t < 0.0 ? (int)floor(t) : (int)t;   // libImaging FLOOR()
}

template <int C, class A>
static int launch_affine_filter(int filter, const View& s, const View& d, const AffineParams& P,
                                const View& dbg, hipStream_t st) {
    dim3 block(64, 4), grid((unsigned)((d.w + 255) / 256), (unsigned)((d.h + 3) / 4), (unsigned)d.n);
    switch (filter) {
        case IMGXF_FILTER_NEAREST:
            hipLaunchKernelGGL((affine_kernel<C, IMGXF_FILTER_NEAREST, A>), grid, block, 0, st, s, d, P, dbg);
            break;
        case IMGXF_FILTER_BILINEAR:
            hipLaunchKernelGGL((affine_kernel<C, IMGXF_FILTER_BILINEAR, A>), grid, block, 0, st, s, d, P, dbg);
            break;
        case IMGXF_FILTER_BICUBIC:
            hipLaunchKernelGGL((affine_kernel<C, IMGXF_FILTER_BICUBIC, A>), grid, block, 0, st, s, d, P, dbg);
            break;
        default: return IMGXF_ERR_ARG;
    }
    return launch_status();
}

int run_affine(const imgxf_view* src, const imgxf_view* dst, const double* m, int filter,
               const uint8_t* fill, int precise, const imgxf_view* dbg_f32, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!m) return IMGXF_ERR_NULL;
    if (src->n != dst->n || src->c != dst->c) return IMGXF_ERR_SHAPE;
    if (filter < 0 || filter > 2) return IMGXF_ERR_ARG;
    if (filter == IMGXF_FILTER_NEAREST && m[1] == 0.0 && m[3] == 0.0) return IMGXF_ERR_UNSUPPORTED;
    if (filter != IMGXF_FILTER_NEAREST && src->c == 4) return IMGXF_ERR_UNSUPPORTED; // Pillow premultiplies RGBA
    if (empty_view(dst)) return IMGXF_OK;
    if (empty_view(src)) return IMGXF_ERR_SHAPE;
    AffineParams P; memset(&P, 0, sizeof(P));
    for (int i = 0; i < 6; ++i) P.m[i] = m[i];
    P.fx[0] = fix16(m[0]); P.fx[1] = fix16(m[1]); P.fx[3] = fix16(m[3]); P.fx[4] = fix16(m[4]);
    P.fx[2] = fix16(m[2] + m[0] * 0.5 + m[1] * 0.5);
    P.fx[5] = fix16(m[5] + m[3] * 0.5 + m[4] * 0.5);
    if (fill) for (int j = 0; j < dst->c; ++j) P.fill[j] = fill[j];
    View dbg; memset(&dbg, 0, sizeof(dbg));
    if (dbg_f32) {
        IMGXF_CHECK(check_view(dbg_f32, 4));
        if (!same_geometry(dst, dbg_f32)) return IMGXF_ERR_SHAPE;
        dbg = make_view(dbg_f32);
    }
    const View s = make_view(src), d = make_view(dst);
    hipStream_t st = (hipStream_t)stream;
    const bool pr = precise != 0 || filter == IMGXF_FILTER_NEAREST;
    switch (src->c) {
        case 1: return pr ? launch_affine_filter<1, PreciseArith>(filter, s, d, P, dbg, st)
                          : launch_affine_filter<1, FastArith>(filter, s, d, P, dbg, st);
        case 3: return pr ? launch_affine_filter<3, PreciseArith>(filter, s, d, P, dbg, st)
                          : launch_affine_filter<3, FastArith>(filter, s, d, P, dbg, st);
        case 4: return launch_affine_filter<4, PreciseArith>(filter, s, d, P, dbg, st);
        default: return IMGXF_ERR_UNSUPPORTED;
    }
}

// ---- ImagingScaleAffine: NEAREST with m1 == m3 == 0 ------------------------------------
// libImaging walks the source coordinate with repeated double additions (xo += a0), so the
// index tables are produced by ONE lane per axis doing the same serial additions (bit-exact,
// <= 32767 steps), then a gather kernel copies pixels.  No host round trip.
__global__ void scale_tables_kernel(int* xtab, int* ytab, int* meta, int ow, int oh, int sw, int sh,
                                    double a0, double a2, double a4, double a5) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double xo = __dadd_rn(a2, __dmul_rn(a0, 0.5));
        int xmin = ow, xmax = 0;
        for (int x = 0; x < ow; ++x) {
            const int xin = xo < 0.0 ? -1 : (int)xo;   // COORD()
            if (xin >= 0 && xin < sw) {
                xmax = x + 1;
                if (x < xmin) xmin = x;
            }
            xtab[x] = xin;
            xo = __dadd_rn(xo, a0);
        }
        meta[0] = xmin; meta[1] = xmax;
    }
    if (threadIdx.x == 0 && blockIdx.x == 1) {
        double yo = __dadd_rn(a5, __dmul_rn(a4, 0.5));
        for (int y = 0; y < oh; ++y) {
            const int yin = yo < 0.0 ? -1 : (int)yo;
            ytab[y] = (yin >= 0 && yin < sh) ? yin : -1;
            yo = __dadd_rn(yo, a4);
        }
    }
}

struct Fill4 { u8 v[4]; };

__global__ __launch_bounds__(256) void scale_nearest_kernel(View s, View d, const int* xtab,
                                                            const int* ytab, const int* meta,
                                                            Fill4 fillc) {
    const int64_t total = (int64_t)d.n * d.h * d.w;
    const int xmin = meta[0], xmax = meta[1];
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int x = (int)(t % d.w);
        const int64_t r = t / d.w;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        u8* dp = d.row(f, y) + x * d.c;
        const int yi = ytab[y];
        if (yi >= 0 && x >= xmin && x < xmax) {
            int xi = xtab[x];
            xi = xi < 0 ? 0 : (xi >= s.w ? s.w - 1 : xi);
            const u8* q = s.row(f, yi) + xi * s.c;
            for (int j = 0; j < d.c; ++j) dp[j] = q[j];
        } else {
            for (int j = 0; j < d.c; ++j) dp[j] = fillc.v[j];
        }
    }
}

} // namespace imgxf

using namespace imgxf;

IMGXF_API int imgxf_affine_u8(const imgxf_view* src, const imgxf_view* dst, const double* m,
                              int filter, const uint8_t* fill, int precise,
                              const imgxf_view* dst_f32, void* stream) {
    return run_affine(src, dst, m, filter, fill, precise, dst_f32, stream);
}

IMGXF_API int imgxf_affine_scale_nearest_u8(const imgxf_view* src, const imgxf_view* dst,
                                            const double* m, const uint8_t* fill, void* workspace,
                                            size_t workspace_bytes, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!m || !workspace) return IMGXF_ERR_NULL;
    if (src->n != dst->n || src->c != dst->c) return IMGXF_ERR_SHAPE;
    if (m[1] != 0.0 || m[3] != 0.0) return IMGXF_ERR_ARG;
    if (workspace_bytes < sizeof(int) * ((size_t)dst->w + dst->h + 2)) return IMGXF_ERR_WORKSPACE;
    if (((uintptr_t)workspace) & 3) return IMGXF_ERR_ARG;
    if (empty_view(dst)) return IMGXF_OK;
    if (empty_view(src)) return IMGXF_ERR_SHAPE;
    int* xtab = (int*)workspace;
    int* ytab = xtab + dst->w;
    int* meta = ytab + dst->h;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(scale_tables_kernel, dim3(2), dim3(64), 0, st, xtab, ytab, meta, dst->w, dst->h,
                       src->w, src->h, m[0], m[2], m[4], m[5]);
    Fill4 fc; memset(&fc, 0, sizeof(fc));
    if (fill) for (int j = 0; j < dst->c; ++j) fc.v[j] = fill[j];
    const View d = make_view(dst);
    int64_t total = (int64_t)d.n * d.h * d.w;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scale_nearest_kernel, dim3((unsigned)blocks), dim3(256), 0, st, make_view(src), d,
                       xtab, ytab, meta, fc);
    return launch_status();
}
