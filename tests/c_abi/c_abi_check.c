/* TEST: the C-ABI of libimgxf.so driven from plain C — hipMalloc'd buffers, a caller-created
 * stream, no Python and no torch anywhere — checked against oracle/c (the CPU restatement).
 * This is the shape of the binding a non-Python host would write.
 * Built by tests/test_gpu_c_abi.py with gcc; prints "c_abi ok" and exits 0 on success. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "imgxf.h"

int oracle_gaussian_blur_u8(const uint8_t* src, uint8_t* dst, double* tmp, int h, int w, int c, int ksize, double sigma);
int oracle_affine_u8(const uint8_t* src, int h, int w, int c, uint8_t* dst, int oh, int ow, const double* m, int filter, const uint8_t* fill);
int oracle_rgb2l_u8(const uint8_t* src, uint8_t* dst, int h, int w, int c);
int oracle_sobel_u8(const uint8_t* g, uint8_t* dst, int h, int w, int variant);

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define IMGXF(x) do { int r_ = (x); if (r_ != IMGXF_OK) { fprintf(stderr, "%s -> %d (%s)\n", #x, r_, imgxf_strerror(r_)); return 3; } } while (0)

static imgxf_view view(void* p, int n, int h, int w, int c) {
    imgxf_view v; v.data = p; v.n = n; v.h = h; v.w = w; v.c = c;
    v.row_stride = (int64_t)w * c; v.frame_stride = (int64_t)h * w * c;
    return v;
}

static long diff(const uint8_t* a, const uint8_t* b, size_t n, int* maxabs) {
    long bad = 0; *maxabs = 0;
    for (size_t i = 0; i < n; ++i) if (a[i] != b[i]) { ++bad; int d = abs((int)a[i] - (int)b[i]); if (d > *maxabs) *maxabs = d; }
    return bad;
}

int main(void) {
    if (imgxf_version() != IMGXF_VERSION) { fprintf(stderr, "version mismatch\n"); return 1; }
    if (imgxf_device_count() < 1) { fprintf(stderr, "no gfx950 device\n"); return 1; }
    const int n = 3, h = 270, w = 480, c = 3;          /* 16-byte aligned rows: the marching kernels run */
    const size_t fb = (size_t)h * w * c, total = fb * n;
    uint8_t* host = (uint8_t*)malloc(total);
    uint8_t* got = (uint8_t*)malloc(total);
    uint8_t* want = (uint8_t*)malloc(fb);
    double* tmp = (double*)malloc(fb * sizeof(double));
    uint32_t s = 12345u;
    for (size_t i = 0; i < total; ++i) { s = s * 1664525u + 1013904223u; host[i] = (uint8_t)(s >> 24); }

    hipStream_t st;
    HIP(hipStreamCreate(&st));
    void *dsrc, *ddst, *dgray, *dedge;
    HIP(hipMalloc(&dsrc, total)); HIP(hipMalloc(&ddst, total));
    HIP(hipMalloc(&dgray, (size_t)n * h * w)); HIP(hipMalloc(&dedge, (size_t)n * h * w));
    HIP(hipMemcpyAsync(dsrc, host, total, hipMemcpyHostToDevice, st));
    const imgxf_view vs = view(dsrc, n, h, w, c), vd = view(ddst, n, h, w, c);
    const imgxf_view vg = view(dgray, n, h, w, 1), ve = view(dedge, n, h, w, 1);
    int maxabs; long bad;

    /* a1: 5x5 Gaussian, sigma 5/6 (float definition: <= 1 LSB on rare ties against the fp64 oracle) */
    IMGXF(imgxf_gaussian_u8(&vs, &vd, 5, 5.0 / 6.0, NULL, st));
    HIP(hipMemcpyAsync(got, ddst, total, hipMemcpyDeviceToHost, st)); HIP(hipStreamSynchronize(st));
    for (int f = 0; f < n; ++f) {
        oracle_gaussian_blur_u8(host + f * fb, want, tmp, h, w, c, 5, 5.0 / 6.0);
        bad = diff(got + f * fb, want, fb, &maxabs);
        if (maxabs > 1 || bad > (long)(fb / 1000)) { fprintf(stderr, "gaussian frame %d: %ld bad, max %d\n", f, bad, maxabs); return 4; }
    }
    /* a2 / a2': rotate 30 deg about the centre, 1.5x zoom — NEAREST and BILINEAR, bit-exact */
    const double a = 30.0 * M_PI / 180.0, cs = cos(a) / 1.5, sn = sin(a) / 1.5, cx = w / 2.0, cy = h / 2.0;
    const double m[6] = {cs, -sn, cx - (cs * cx - sn * cy), sn, cs, cy - (sn * cx + cs * cy)};
    const uint8_t fill[4] = {7, 8, 9, 0};
    for (int filter = 0; filter <= 1; ++filter) {
        IMGXF(imgxf_affine_u8(&vs, &vd, m, filter, fill, 1, NULL, st));
        HIP(hipMemcpyAsync(got, ddst, total, hipMemcpyDeviceToHost, st)); HIP(hipStreamSynchronize(st));
        for (int f = 0; f < n; ++f) {
            oracle_affine_u8(host + f * fb, h, w, c, want, h, w, m, filter, fill);
            bad = diff(got + f * fb, want, fb, &maxabs);
            if (bad) { fprintf(stderr, "affine filter %d frame %d: %ld bad\n", filter, f, bad); return 5; }
        }
    }
    /* a4 / a6: convert('L') then the Sobel magnitude, bit-exact */
    IMGXF(imgxf_rgb2l_u8(&vs, &vg, st));
    IMGXF(imgxf_sobel_u8(&vg, &ve, IMGXF_SOBEL_MAGNITUDE, st));
    HIP(hipMemcpyAsync(got, dedge, (size_t)n * h * w, hipMemcpyDeviceToHost, st)); HIP(hipStreamSynchronize(st));
    uint8_t* gray = (uint8_t*)malloc((size_t)h * w);
    for (int f = 0; f < n; ++f) {
        oracle_rgb2l_u8(host + f * fb, gray, h, w, c);
        oracle_sobel_u8(gray, want, h, w, 2);
        bad = diff(got + (size_t)f * h * w, want, (size_t)h * w, &maxabs);
        if (bad) { fprintf(stderr, "sobel frame %d: %ld bad\n", f, bad); return 6; }
    }
    /* error behaviour: mismatched geometry and NULL views are reported, not faulted on */
    imgxf_view vbad = vd; vbad.w = w - 1;
    if (imgxf_gaussian_u8(&vs, &vbad, 5, 1.0, NULL, st) != IMGXF_ERR_SHAPE) return 7;
    if (imgxf_gaussian_u8(NULL, &vd, 5, 1.0, NULL, st) != IMGXF_ERR_NULL) return 7;
    if (imgxf_gaussian_u8(&vs, &vd, 4, 1.0, NULL, st) != IMGXF_ERR_ARG) return 7;

    HIP(hipFree(dsrc)); HIP(hipFree(ddst)); HIP(hipFree(dgray)); HIP(hipFree(dedge));
    HIP(hipStreamDestroy(st));
    free(host); free(got); free(want); free(tmp); free(gray);
    printf("c_abi ok\n");
    return 0;
}
