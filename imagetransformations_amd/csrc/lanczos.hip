// Lanczos-3 resize: the body behind img.resize((nw,nh), Image.Resampling.LANCZOS)
// (/root/reference/transformation.py:179), i.e. libImaging Resample.c for 8-bit images:
// host-side double-precision coefficient windows normalised to 22-bit integers, then a
// horizontal integer MAC pass into a uint8 intermediate followed by a vertical pass.
#include "imgxf_common.h"
#include <math.h>
#include <string.h>
#include <vector>

#define PRECISION_BITS (32 - 8 - 2)

struct imgxf_lanczos_plan {
    int in_h, in_w, out_h, out_w, c, max_frames;
    int ksx, ksy;            // coefficients per output sample (row length of the tables)
    int* d_bounds_x;         // [out_w][2] (xmin, count)
    int* d_kk_x;             // [out_w][ksx]
    int* d_bounds_y;         // [out_h][2]
    int* d_kk_y;             // [out_h][ksy]
    uint8_t* d_tmp;          // [max_frames][in_h][out_w][c], only when both passes run
    int need_h, need_v;
};

namespace imgxf {

static inline double sinc_filter(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return sin(x) / x;
}
static inline double lanczos_filter(double x) {
    if (-3.0 <= x && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3);
    return 0.0;
}

// precompute_coeffs + normalize_coeffs_8bpc (whole-image box)
static int build_coeffs(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& kk) {
    double scale, filterscale;
    filterscale = scale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 3.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = lanczos_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) k[x] /= ww;
            const double v = k[x];
            kk[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS))
                                               : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return ksize;
}

__device__ __forceinline__ u8 clip8(int v) {
    v >>= PRECISION_BITS;
    return (u8)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// out(y, xx, ch) = clip8(2^21 + sum_x in(y, xmin+x, ch) * k[xx][x])
template <int C>
__global__ __launch_bounds__(256) void resample_h_kernel(View s, View d, const int* bounds,
                                                         const int* kk, int ksize) {
    const int64_t total = (int64_t)d.n * d.h * d.w;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int xx = (int)(t % d.w);
        const int64_t r = t / d.w;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
        const int* k = kk + (int64_t)xx * ksize;
        const u8* sp = s.row(f, y) + xmin * C;
        int acc[C];
#pragma unroll
        for (int j = 0; j < C; ++j) acc[j] = 1 << (PRECISION_BITS - 1);
        for (int x = 0; x < cnt; ++x) {
            const int w = k[x];
#pragma unroll
            for (int j = 0; j < C; ++j) acc[j] += (int)sp[x * C + j] * w;
        }
        u8* dp = d.row(f, y) + xx * C;
#pragma unroll
        for (int j = 0; j < C; ++j) dp[j] = clip8(acc[j]);
    }
}

// out(yy, x, :) = clip8(2^21 + sum_y in(ymin+y, x, :) * k[yy][y]); lanes walk bytes of a row
__global__ __launch_bounds__(256) void resample_v_kernel(View s, View d, const int* bounds,
                                                         const int* kk, int ksize) {
    const int rowbytes = d.w * d.c;
    const int64_t total = (int64_t)d.n * d.h * rowbytes;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int xb = (int)(t % rowbytes);
        const int64_t r = t / rowbytes;
        const int yy = (int)(r % d.h), f = (int)(r / d.h);
        const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
        const int* k = kk + (int64_t)yy * ksize;
        int acc = 1 << (PRECISION_BITS - 1);
        const u8* sp = s.row(f, ymin) + xb;
        for (int y = 0; y < cnt; ++y) acc += (int)sp[(int64_t)y * s.rs] * k[y];
        d.row(f, yy)[xb] = clip8(acc);
    }
}

static inline unsigned grid_for(int64_t total) {
    int64_t blocks = (total + 255) / 256;
    return (unsigned)(blocks > 16384 ? 16384 : (blocks < 1 ? 1 : blocks));
}

static int launch_h(const View& s, const View& d, const int* b, const int* k, int ks, hipStream_t st) {
    const unsigned g = grid_for((int64_t)d.n * d.h * d.w);
    switch (s.c) {
        case 1: hipLaunchKernelGGL((resample_h_kernel<1>), dim3(g), dim3(256), 0, st, s, d, b, k, ks); break;
        case 3: hipLaunchKernelGGL((resample_h_kernel<3>), dim3(g), dim3(256), 0, st, s, d, b, k, ks); break;
        case 4: hipLaunchKernelGGL((resample_h_kernel<4>), dim3(g), dim3(256), 0, st, s, d, b, k, ks); break;
        default: return IMGXF_ERR_UNSUPPORTED;
    }
    return launch_status();
}

} // namespace imgxf

using namespace imgxf;

static int upload(const std::vector<int>& v, int** dptr) {
    hipError_t e = hipMalloc((void**)dptr, v.size() * sizeof(int));
    if (e != hipSuccess) return (int)e;
    e = hipMemcpy(*dptr, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice);
    return e == hipSuccess ? IMGXF_OK : (int)e;
}

IMGXF_API int imgxf_lanczos_plan_create(imgxf_lanczos_plan** plan, int in_h, int in_w, int out_h,
                                        int out_w, int c, int max_frames) {
    if (!plan) return IMGXF_ERR_NULL;
    *plan = nullptr;
    if (in_h < 1 || in_w < 1 || out_h < 1 || out_w < 1 || max_frames < 1) return IMGXF_ERR_ARG;
    if (c != 1 && c != 3 && c != 4) return IMGXF_ERR_UNSUPPORTED;
    imgxf_lanczos_plan* p = new imgxf_lanczos_plan();
    memset(p, 0, sizeof(*p));
    p->in_h = in_h; p->in_w = in_w; p->out_h = out_h; p->out_w = out_w; p->c = c;
    p->max_frames = max_frames;
    p->need_h = out_w != in_w;   // ImagingResample: a pass is skipped when the size is unchanged
    p->need_v = out_h != in_h;
    int rc = IMGXF_OK;
    if (p->need_h) {
        std::vector<int> b, k;
        p->ksx = build_coeffs(in_w, out_w, b, k);
        if ((rc = upload(b, &p->d_bounds_x)) == IMGXF_OK) rc = upload(k, &p->d_kk_x);
    }
    if (rc == IMGXF_OK && p->need_v) {
        std::vector<int> b, k;
        p->ksy = build_coeffs(in_h, out_h, b, k);
        if ((rc = upload(b, &p->d_bounds_y)) == IMGXF_OK) rc = upload(k, &p->d_kk_y);
    }
    if (rc == IMGXF_OK && p->need_h && p->need_v) {
        hipError_t e = hipMalloc((void**)&p->d_tmp, (size_t)max_frames * in_h * out_w * c);
        if (e != hipSuccess) rc = (int)e;
    }
    if (rc != IMGXF_OK) { imgxf_lanczos_plan_destroy(p); return rc; }
    *plan = p;
    return IMGXF_OK;
}

IMGXF_API int imgxf_lanczos_plan_destroy(imgxf_lanczos_plan* p) {
    if (!p) return IMGXF_OK;
    if (p->d_bounds_x) (void)hipFree(p->d_bounds_x);
    if (p->d_kk_x) (void)hipFree(p->d_kk_x);
    if (p->d_bounds_y) (void)hipFree(p->d_bounds_y);
    if (p->d_kk_y) (void)hipFree(p->d_kk_y);
    if (p->d_tmp) (void)hipFree(p->d_tmp);
    delete p;
    return IMGXF_OK;
}

IMGXF_API int imgxf_resize_lanczos_u8(const imgxf_lanczos_plan* p, const imgxf_view* src,
                                      const imgxf_view* dst, void* stream) {
    if (!p) return IMGXF_ERR_NULL;
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (src->n != dst->n || src->c != p->c || dst->c != p->c) return IMGXF_ERR_SHAPE;
    if (src->h != p->in_h || src->w != p->in_w || dst->h != p->out_h || dst->w != p->out_w)
        return IMGXF_ERR_SHAPE;
    if (src->n > p->max_frames) return IMGXF_ERR_WORKSPACE;
    if (src->n == 0) return IMGXF_OK;
    hipStream_t st = (hipStream_t)stream;
    const View s = make_view(src), d = make_view(dst);
    if (!p->need_h && !p->need_v) {   // same size: Image.resize returns a copy
        const size_t rb = (size_t)s.w * s.c;
        for (int f = 0; f < s.n; ++f) {
            hipError_t e = hipMemcpy2DAsync(d.p + f * d.fs, d.rs, s.p + f * s.fs, s.rs, rb, s.h,
                                            hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) return (int)e;
        }
        return IMGXF_OK;
    }
    if (p->need_h && !p->need_v) return launch_h(s, d, p->d_bounds_x, p->d_kk_x, p->ksx, st);
    View mid = s;
    if (p->need_h) {
        mid.p = p->d_tmp; mid.n = s.n; mid.h = p->in_h; mid.w = p->out_w; mid.c = p->c;
        mid.rs = (int64_t)p->out_w * p->c; mid.fs = mid.rs * p->in_h;
        IMGXF_CHECK(launch_h(s, mid, p->d_bounds_x, p->d_kk_x, p->ksx, st));
    }
    const int64_t total = (int64_t)d.n * d.h * d.rowbytes();
    hipLaunchKernelGGL(resample_v_kernel, dim3(grid_for(total)), dim3(256), 0, st, mid, d,
                       p->d_bounds_y, p->d_kk_y, p->ksy);
    return launch_status();
}
