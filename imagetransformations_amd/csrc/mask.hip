// Mask stage of apply_background_change (/root/reference/transformation.py:340-341):
//   np.percentile(edges, 70)  -> 256-bin histogram (LDS-privatised, one atomic per bin per
//                                workgroup) + a one-lane order-statistic walk;
//   edges > threshold         -> 0/255 mask;
//   binary_dilation(mask, iterations=k) with the 4-connected cross and border 0, which for
//   this structuring element equals "some set pixel within L1 distance k".
#include "imgxf_common.h"
#include <math.h>

namespace imgxf {

__global__ __launch_bounds__(256) void hist_kernel(View s, u32* hist) {
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int f = blockIdx.y;
    const int64_t total = (int64_t)s.h * s.w;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int x = (int)(t % s.w), y = (int)(t / s.w);
        atomicAdd(&h[s.row(f, y)[x]], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[f * 256 + threadIdx.x], h[threadIdx.x]);
}

// numpy percentile, method 'linear' (lib/_function_base_impl.py: _compute_virtual_index with
// alpha=beta=1, _lerp), evaluated on the sorted multiset described by the histogram.
__global__ void percentile_kernel(const u32* hist, int nframes, int64_t count, double q,
                                  double* thr) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nframes) return;
    const u32* h = hist + f * 256;
    const double quant = q / 100.0;
    const double n = (double)count;
    double virt = __dsub_rn(__dadd_rn(__dmul_rn(n, quant),
                                      __dadd_rn(1.0, __dmul_rn(quant, (1.0 - 1.0 - 1.0)))), 1.0);
    double lof = floor(virt);
    double gamma = __dsub_rn(virt, lof);
    int64_t lo = (int64_t)lof;
    if (lo < 0) lo = 0;
    if (lo > count - 1) lo = count - 1;
    int64_t hi = lo + 1 > count - 1 ? count - 1 : lo + 1;
    // value at sorted index i = first bin whose cumulative count exceeds i
    int va = 255, vb = 255;
    int64_t cum = 0;
    bool ga = false, gb = false;
    for (int b = 0; b < 256; ++b) {
        cum += h[b];
        if (!ga && cum > lo) { va = b; ga = true; }
        if (!gb && cum > hi) { vb = b; gb = true; }
    }
    const double a = (double)va, bb = (double)vb;
    const double diff = __dsub_rn(bb, a);
    double out = __dadd_rn(a, __dmul_rn(diff, gamma));
    if (gamma >= 0.5) out = __dsub_rn(bb, __dmul_rn(diff, __dsub_rn(1.0, gamma)));
    if (diff == 0.0) out = a;
    thr[f] = out;
}

__global__ __launch_bounds__(256) void gt_mask_kernel(View s, View d, const double* thr) {
    const int64_t total = (int64_t)d.n * d.h * d.w;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int x = (int)(t % d.w);
        const int64_t r = t / d.w;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        d.row(f, y)[x] = ((double)s.row(f, y)[x] > thr[f]) ? 255 : 0;
    }
}

__global__ __launch_bounds__(256) void dilate_kernel(View s, View d, int k) {
    const int64_t total = (int64_t)d.n * d.h * d.w;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int x = (int)(t % d.w);
        const int64_t r = t / d.w;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        u8 hit = 0;
        for (int dy = -k; dy <= k && !hit; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= s.h) continue;
            const int span = k - (dy < 0 ? -dy : dy);
            const u8* rp = s.row(f, yy);
            for (int dx = -span; dx <= span; ++dx) {
                const int xx = x + dx;
                if (xx >= 0 && xx < s.w && rp[xx]) { hit = 255; break; }
            }
        }
        d.row(f, y)[x] = hit;
    }
}

static inline unsigned grid_for(int64_t total) {
    int64_t blocks = (total + 255) / 256;
    return (unsigned)(blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks));
}

} // namespace imgxf

using namespace imgxf;

IMGXF_API int imgxf_histogram_u8(const imgxf_view* src, uint32_t* hist, void* stream) {
    IMGXF_CHECK(check_view(src));
    if (!hist) return IMGXF_ERR_NULL;
    if (src->c != 1) return IMGXF_ERR_SHAPE;
    if (src->n == 0) return IMGXF_OK;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(hist, 0, (size_t)src->n * 256 * sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
    if (empty_view(src)) return IMGXF_OK;
    const View s = make_view(src);
    int64_t bx = ((int64_t)s.h * s.w + 256 * 16 - 1) / (256 * 16);
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(hist_kernel, dim3((unsigned)bx, (unsigned)s.n), dim3(256), 0, st, s, hist);
    return launch_status();
}

IMGXF_API int imgxf_percentile_mask_u8(const imgxf_view* src, const uint32_t* hist, double q,
                                       const imgxf_view* dst, double* thr_out, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!hist || !thr_out) return IMGXF_ERR_NULL;
    if (!same_geometry(src, dst) || src->c != 1) return IMGXF_ERR_SHAPE;
    if (!(q >= 0.0 && q <= 100.0)) return IMGXF_ERR_ARG;
    if (empty_view(src)) return IMGXF_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(percentile_kernel, dim3((unsigned)((src->n + 63) / 64)), dim3(64), 0, st, hist,
                       src->n, (int64_t)src->h * src->w, q, thr_out);
    const View d = make_view(dst);
    hipLaunchKernelGGL(gt_mask_kernel, dim3(grid_for((int64_t)d.n * d.h * d.w)), dim3(256), 0, st,
                       make_view(src), d, (const double*)thr_out);
    return launch_status();
}

IMGXF_API int imgxf_dilate_cross_u8(const imgxf_view* src, const imgxf_view* dst, int iterations,
                                    void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!same_geometry(src, dst) || src->c != 1) return IMGXF_ERR_SHAPE;
    if (iterations < 1 || iterations > 16) return IMGXF_ERR_ARG;
    if (empty_view(src)) return IMGXF_OK;
    const View d = make_view(dst);
    hipLaunchKernelGGL(dilate_kernel, dim3(grid_for((int64_t)d.n * d.h * d.w)), dim3(256), 0,
                       (hipStream_t)stream, make_view(src), d, iterations);
    return launch_status();
}
