// C-ABI entry points for the separable filters (Gaussian blur).
// Replaces cv2.GaussianBlur at /root/reference/transformation.py:249.
#include "sepconv_tile.inc"
#include <math.h>
#include <string.h>

namespace imgxf {
int sepconv_c1(int R, const View&, const View&, const View&, const Taps&, int, hipStream_t);
int sepconv_c3(int R, const View&, const View&, const View&, const Taps&, int, hipStream_t);
int sepconv_c4(int R, const View&, const View&, const View&, const Taps&, int, hipStream_t);
int sepconv_fx_c1(int R, const View&, const View&, const View&, const Taps&, int, hipStream_t);
int sepconv_fx_c3(int R, const View&, const View&, const View&, const Taps&, int, hipStream_t);
int sepconv_fx_c4(int R, const View&, const View&, const View&, const Taps&, int, hipStream_t);

static int run_sepconv(const imgxf_view* src, const imgxf_view* dst, const float* kx, int nkx,
                       const float* ky, int nky, int border, const imgxf_view* dst_f32,
                       void* stream, bool fixed = false) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!kx || !ky) return IMGXF_ERR_NULL;
    if (!same_geometry(src, dst)) return IMGXF_ERR_SHAPE;
    if (nkx < 1 || nky < 1 || !(nkx & 1) || !(nky & 1) || nkx > 31 || nky > 31) return IMGXF_ERR_ARG;
    if (border != IMGXF_BORDER_REFLECT_101 && border != IMGXF_BORDER_REFLECT) return IMGXF_ERR_ARG;
    View df; memset(&df, 0, sizeof(df));
    if (dst_f32) {
        IMGXF_CHECK(check_view(dst_f32, 4));
        if (!same_geometry(src, dst_f32)) return IMGXF_ERR_SHAPE;
        if (((uintptr_t)dst_f32->data & 3) || (dst_f32->row_stride & 3) || (dst_f32->frame_stride & 3))
            return IMGXF_ERR_ARG;
        df = make_view(dst_f32);
    }
    if (empty_view(src)) return IMGXF_OK;
    int R = (nkx > nky ? nkx : nky) / 2;
    if (R < 1) R = 1;
    Taps taps; memset(&taps, 0, sizeof(taps));
    for (int i = 0; i < nkx; ++i) taps.x[R - nkx / 2 + i] = kx[i];
    for (int i = 0; i < nky; ++i) taps.y[R - nky / 2 + i] = ky[i];
    const View s = make_view(src), d = make_view(dst);
    hipStream_t st = (hipStream_t)stream;
    if (fixed) {
        switch (src->c) {
            case 1: return sepconv_fx_c1(R, s, d, df, taps, border, st);
            case 3: return sepconv_fx_c3(R, s, d, df, taps, border, st);
            case 4: return sepconv_fx_c4(R, s, d, df, taps, border, st);
            default: return IMGXF_ERR_UNSUPPORTED;
        }
    }
    switch (src->c) {
        case 1: return sepconv_c1(R, s, d, df, taps, border, st);
        case 3: return sepconv_c3(R, s, d, df, taps, border, st);
        case 4: return sepconv_c4(R, s, d, df, taps, border, st);
        default: return IMGXF_ERR_UNSUPPORTED;
    }
}
} // namespace imgxf

using namespace imgxf;

IMGXF_API int imgxf_sepconv_u8(const imgxf_view* src, const imgxf_view* dst, const float* kx,
                               int nkx, const float* ky, int nky, int border,
                               const imgxf_view* dst_f32, void* stream) {
    return run_sepconv(src, dst, kx, nkx, ky, nky, border, dst_f32, stream);
}

// cv::getGaussianKernel: binomial kernels for sigma <= 0 and ksize in {1,3,5,7}
static double small_gaussian_tab(int ksize, int i) {
    static const double t3[3] = {0.25, 0.5, 0.25}, t5[5] = {0.0625, 0.25, 0.375, 0.25, 0.0625};
    static const double t7[7] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125};
    return ksize == 1 ? 1.0 : ksize == 3 ? t3[i] : ksize == 5 ? t5[i] : t7[i];
}

IMGXF_API int imgxf_gaussian_u8(const imgxf_view* src, const imgxf_view* dst, int ksize,
                                double sigma, const imgxf_view* dst_f32, void* stream) {
    if (ksize < 1 || !(ksize & 1) || ksize > 31) return IMGXF_ERR_ARG;
    float kf[31];
    if (sigma <= 0 && ksize <= 7) {            // cv::getGaussianKernel's small_gaussian_tab
        for (int i = 0; i < ksize; ++i) kf[i] = (float)small_gaussian_tab(ksize, i);
    } else {
        if (sigma <= 0) sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8;
        double kd[31], sum = 0.0;
        for (int i = 0; i < ksize; ++i) {
            const double x = i - (ksize - 1) * 0.5;
            kd[i] = exp(-(x * x) / (2.0 * sigma * sigma));
            sum += kd[i];
        }
        for (int i = 0; i < ksize; ++i) kf[i] = (float)(kd[i] / sum);
    }
    return run_sepconv(src, dst, kf, ksize, kf, ksize, IMGXF_BORDER_REFLECT_101, dst_f32, stream);
}

// 8.8 fixed-point separable filter (OpenCV's uint8 path: ufixedpoint16 rows, ufixedpoint32
// columns, (v + 2^15) >> 16).  Taps are integers n/256; each axis must sum to <= 256 so that no
// intermediate saturates.  Computed in fp32, where every product and partial sum is exact.
static int fixed_taps(const uint16_t* k, int n, float* out) {
    if (!k) return IMGXF_ERR_NULL;
    if (n < 1 || !(n & 1) || n > 31) return IMGXF_ERR_ARG;
    unsigned sum = 0;
    for (int i = 0; i < n; ++i) { sum += k[i]; out[i] = (float)k[i] * (1.0f / 256.0f); }
    return sum <= 256 ? IMGXF_OK : IMGXF_ERR_ARG;
}

IMGXF_API int imgxf_sepconv_fixed_u8(const imgxf_view* src, const imgxf_view* dst, const uint16_t* kx,
                                     int nkx, const uint16_t* ky, int nky, int border, void* stream) {
    float fx[31], fy[31];
    IMGXF_CHECK(fixed_taps(kx, nkx, fx));
    IMGXF_CHECK(fixed_taps(ky, nky, fy));
    return run_sepconv(src, dst, fx, nkx, fy, nky, border, nullptr, stream, true);
}

IMGXF_API int imgxf_gaussian_cv_fixed_u8(const imgxf_view* src, const imgxf_view* dst, int ksize,
                                         double sigma, void* stream) {
    if (ksize < 1 || !(ksize & 1) || ksize > 31) return IMGXF_ERR_ARG;
    if (sigma <= 0 && ksize <= 7) {            // the binomial tables are exact multiples of 1/256
        uint16_t kt[7];
        for (int i = 0; i < ksize; ++i) kt[i] = (uint16_t)(small_gaussian_tab(ksize, i) * 256.0);
        return imgxf_sepconv_fixed_u8(src, dst, kt, ksize, kt, ksize, IMGXF_BORDER_REFLECT_101, stream);
    }
    if (sigma <= 0) sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8;
    // getGaussianKernelFixedPoint_ED: float kernel * 256, rounded with error diffusion from the
    // ends inward (round half to even), centre = 256 - the rest
    const int n2 = (ksize - 1) / 2;
    const double scale2x = -0.125 / (sigma * sigma);
    double vals[16], sum = 0.0;
    for (int i = 0; i < n2; ++i) {
        const double x = (double)(1 - ksize + 2 * i);
        vals[i] = exp(x * x * scale2x);
        sum += vals[i];
    }
    const double mul1 = 1.0 / (2.0 * sum + 1.0);
    uint16_t k[31];
    double err = 0.0;
    long tot = 0;
    for (int i = 0; i < n2; ++i) {
        const double adj = vals[i] * mul1 * 256.0 + err;
        const double v0 = nearbyint(adj);
        err = adj - v0;
        if (v0 < 0 || v0 > 256) return IMGXF_ERR_ARG;
        k[i] = k[ksize - 1 - i] = (uint16_t)v0;
        tot += (long)v0;
    }
    if (2 * tot > 256) return IMGXF_ERR_ARG;
    k[n2] = (uint16_t)(256 - 2 * tot);
    return imgxf_sepconv_fixed_u8(src, dst, k, ksize, k, ksize, IMGXF_BORDER_REFLECT_101, stream);
}
