/*
 * imgxf_oracle.c — plain-C CPU restatement of the benchmark leg of the hot path
 * (TEST INFRASTRUCTURE ONLY: used by tests/ to cross-check the NumPy oracle and by
 * bench.py's `cpu_baseline` leg; never linked into or called by the product library).
 *
 * Semantics are those of oracle/imgxf_oracle.py, which cites and is pinned against the
 * third-party calls the reference makes:
 *   gaussian_blur_u8   cv2.GaussianBlur      /root/reference/transformation.py:249 (parity unpinned: no cv2 here)
 *   affine_u8          Image.transform(AFFINE, NEAREST|BILINEAR)  transformation.py:200, libImaging Geometry.c
 *   rgb2l_u8           Image.convert('L')    transformation.py:336
 *   sobel_x_wrap_u8    scipy.ndimage.sobel   transformation.py:339
 *   sobel_mag_u8       benchmark configs[2]
 * Rows are independent, so every loop is `omp parallel for` over rows; the thread count is
 * whatever the caller sets through omp_set_num_threads (reported as cpu_baseline.cores).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    int p = 2 * n - 2;
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - i : i;
}
static inline int reflect_sym(int i, int n) {
    int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - 1 - i : i;
}

int oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* Separable Gaussian, float64, horizontal then vertical, REFLECT_101, rint + clamp.
 * src/dst: h x w x c interleaved, contiguous.  tmp: caller-provided h*w*c doubles. */
int oracle_gaussian_blur_u8(const uint8_t* src, uint8_t* dst, double* tmp, int h, int w, int c,
                            int ksize, double sigma) {
    if (ksize < 1 || !(ksize & 1) || ksize > 63) return -1;
    double k[63], sum = 0.0;
    /* cv::getGaussianKernel: binomial tables for sigma <= 0 and ksize <= 7 */
    static const double tab3[3] = {0.25, 0.5, 0.25}, tab5[5] = {0.0625, 0.25, 0.375, 0.25, 0.0625};
    static const double tab7[7] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125};
    if (sigma <= 0 && ksize <= 7) {
        const double* t = ksize == 3 ? tab3 : ksize == 5 ? tab5 : tab7;
        if (ksize == 1) k[0] = 1.0; else memcpy(k, t, (size_t)ksize * sizeof(double));
        sum = 1.0;
    } else {
        if (sigma <= 0) sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8;
        for (int i = 0; i < ksize; ++i) {
            double x = i - (ksize - 1) * 0.5;
            k[i] = exp(-(x * x) / (2.0 * sigma * sigma));
            sum += k[i];
        }
        for (int i = 0; i < ksize; ++i) k[i] /= sum;
    }
    const int r = ksize / 2;
    const int rb = w * c;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const uint8_t* sp = src + (size_t)y * rb;
        double* tp = tmp + (size_t)y * rb;
        for (int x = 0; x < w; ++x) {
            for (int ch = 0; ch < c; ++ch) {
                double acc = 0.0;
                if (x >= r && x + r < w) {
                    for (int i = 0; i < ksize; ++i) acc += k[i] * sp[(x + i - r) * c + ch];
                } else {
                    for (int i = 0; i < ksize; ++i) acc += k[i] * sp[reflect101(x + i - r, w) * c + ch];
                }
                tp[x * c + ch] = acc;
            }
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const double* rows[63];
        for (int i = 0; i < ksize; ++i) rows[i] = tmp + (size_t)reflect101(y + i - r, h) * rb;
        uint8_t* dp = dst + (size_t)y * rb;
        for (int xb = 0; xb < rb; ++xb) {
            double acc = 0.0;
            for (int i = 0; i < ksize; ++i) acc += k[i] * rows[i][xb];
            double v = rint(acc);
            dp[xb] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    return 0;
}

static inline int fix16(double v) {
    double t = v * 65536.0 + 0.5;
    return t < 0.0 ? (int)floor(t) : (int)t;
}

/* Image.transform(size, AFFINE, m, filter, fillcolor): filter 0 = NEAREST (affine_fixed,
 * requires m1 or m3 non-zero), 1 = BILINEAR (float64, truncation). */
int oracle_affine_u8(const uint8_t* src, int h, int w, int c, uint8_t* dst, int oh, int ow,
                     const double* m, int filter, const uint8_t* fill) {
    uint8_t fz[4] = {0, 0, 0, 0};
    if (fill) memcpy(fz, fill, (size_t)c);
    if (filter == 0) {
        if (m[1] == 0.0 && m[3] == 0.0) return -2;
        const int a0 = fix16(m[0]), a1 = fix16(m[1]), a3 = fix16(m[3]), a4 = fix16(m[4]);
        const int a2 = fix16(m[2] + m[0] * 0.5 + m[1] * 0.5), a5 = fix16(m[5] + m[3] * 0.5 + m[4] * 0.5);
#pragma omp parallel for schedule(static)
        for (int y = 0; y < oh; ++y) {
            uint8_t* dp = dst + (size_t)y * ow * c;
            for (int x = 0; x < ow; ++x) {
                int xx = (int)((uint32_t)a2 + (uint32_t)a1 * (uint32_t)y + (uint32_t)a0 * (uint32_t)x);
                int yy = (int)((uint32_t)a5 + (uint32_t)a4 * (uint32_t)y + (uint32_t)a3 * (uint32_t)x);
                int xin = xx >> 16, yin = yy >> 16;
                const uint8_t* q = fz;
                if (xin >= 0 && xin < w && yin >= 0 && yin < h) q = src + ((size_t)yin * w + xin) * c;
                for (int j = 0; j < c; ++j) dp[x * c + j] = q[j];
            }
        }
        return 0;
    }
    if (filter != 1) return -3;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < oh; ++y) {
        uint8_t* dp = dst + (size_t)y * ow * c;
        for (int x = 0; x < ow; ++x) {
            double xin = m[0] * (x + 0.5) + m[1] * (y + 0.5) + m[2];
            double yin = m[3] * (x + 0.5) + m[4] * (y + 0.5) + m[5];
            if (xin < 0.0 || xin >= w || yin < 0.0 || yin >= h) {
                for (int j = 0; j < c; ++j) dp[x * c + j] = fz[j];
                continue;
            }
            xin -= 0.5; yin -= 0.5;
            int xi = (int)floor(xin), yi = (int)floor(yin);
            double dx = xin - xi, dy = yin - yi;
            int xa = xi < 0 ? 0 : (xi > w - 1 ? w - 1 : xi);
            int xb = xi + 1 < 0 ? 0 : (xi + 1 > w - 1 ? w - 1 : xi + 1);
            int ya = yi < 0 ? 0 : (yi > h - 1 ? h - 1 : yi);
            const uint8_t* r0 = src + (size_t)ya * w * c;
            int has1 = (yi + 1 >= 0 && yi + 1 < h);
            const uint8_t* r1 = src + (size_t)(has1 ? yi + 1 : 0) * w * c;
            for (int j = 0; j < c; ++j) {
                double v1 = r0[xa * c + j] + (r0[xb * c + j] - r0[xa * c + j]) * dx;
                double v2 = has1 ? r1[xa * c + j] + (r1[xb * c + j] - r1[xa * c + j]) * dx : v1;
                double v = v1 + (v2 - v1) * dy;
                dp[x * c + j] = (uint8_t)v;
            }
        }
    }
    return 0;
}

int oracle_rgb2l_u8(const uint8_t* src, uint8_t* dst, int h, int w, int c) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const uint8_t* p = src + ((size_t)y * w + x) * c;
            dst[(size_t)y * w + x] = (uint8_t)((p[0] * 19595u + p[1] * 38470u + p[2] * 7471u + 0x8000u) >> 16);
        }
    return 0;
}

/* variant 0: scipy x-derivative mod 256; 1: y-derivative mod 256; 2: magnitude (rint, saturate) */
int oracle_sobel_u8(const uint8_t* g, uint8_t* dst, int h, int w, int variant) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const uint8_t* r0 = g + (size_t)reflect_sym(y - 1, h) * w;
        const uint8_t* r1 = g + (size_t)y * w;
        const uint8_t* r2 = g + (size_t)reflect_sym(y + 1, h) * w;
        for (int x = 0; x < w; ++x) {
            int xl = reflect_sym(x - 1, w), xr = reflect_sym(x + 1, w);
            int gx = (r0[xr] - r0[xl]) + 2 * (r1[xr] - r1[xl]) + (r2[xr] - r2[xl]);
            int gy = (r2[xl] - r0[xl]) + 2 * (r2[x] - r0[x]) + (r2[xr] - r0[xr]);
            int out;
            if (variant == 0) out = gx & 0xff;
            else if (variant == 1) out = gy & 0xff;
            else {
                double v = rint(sqrt((double)(gx * gx + gy * gy)));
                out = (int)(v > 255 ? 255 : v);
            }
            dst[(size_t)y * w + x] = (uint8_t)out;
        }
    }
    return 0;
}
