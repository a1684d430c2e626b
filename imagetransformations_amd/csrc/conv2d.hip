// Dense k x k correlation and the Sobel family.  Bodies behind
//   cv2.filter2D(img,-1,kernel)        /root/reference/pipenline/cifar_image_transformations.py:118
//   scipy.ndimage.sobel(gray_u8)       /root/reference/transformation.py:339
//   benchmark configs[2] (RGB -> L -> Gx,Gy -> |G|), SURVEY §8a row a4
// conv2d: one workgroup stages a (16+kh-1) x (128+(kw-1)*C) byte tile (+halo, border
// resolved while loading) into LDS as fp32, each lane then owns one byte column x 8 rows.
// sobel:  LDS-staged L tile (the RGB->L conversion is fused into the tile load for the
// benchmark variant), exact int32 gradients.
#include "sobel_march.inc"
#include <string.h>
#include <algorithm>
#include <stdlib.h>
#include <math.h>

namespace imgxf {

struct KernelTaps { float w[15 * 15]; };

__global__ __launch_bounds__(256) void conv2d_kernel(View s, View d, KernelTaps K, int kh, int kw,
                                                     int border) {
    constexpr int TWB = 128, TH = 16;
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int C = s.c;
    const int ay = kh / 2, ax = kw / 2;
    const int tw = TWB + (kw - 1) * C;   // tile width in bytes
    const int th = TH + kh - 1;
    const int x0b = blockIdx.x * TWB, y0 = blockIdx.y * TH, f = blockIdx.z;
    const int rowbytes = s.w * C;
    for (int t = threadIdx.x; t < tw * th; t += 256) {
        const int ty = t / tw, tx = t - ty * tw;
        int gy = y0 - ay + ty;
        if (gy < 0 || gy >= s.h) gy = border_index(gy, s.h, border);
        const int bx = x0b - ax * C + tx;
        int px = bx >= 0 ? bx / C : -((-bx + C - 1) / C);
        const int ch = bx - px * C;
        if (px < 0 || px >= s.w) px = border_index(px, s.w, border);
        tile[t] = (float)s.row(f, gy)[px * C + ch];
    }
    __syncthreads();
    const int col = threadIdx.x & 127, rg = threadIdx.x >> 7;   // 2 row groups of 8 rows
    const int xb = x0b + col;
    if (xb >= rowbytes) return;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = 0.f;
    for (int j = 0; j < kh; ++j) {
        for (int i = 0; i < kw; ++i) {
            const float wgt = K.w[j * kw + i];
            if (wgt == 0.0f) continue;                       // uniform: sparse kernels pay for their non-zeros only
            const float* tp = &tile[(rg * 8 + j) * tw + col + i * C];
#pragma unroll
            for (int o = 0; o < 8; ++o) acc[o] = fmaf(wgt, tp[o * tw], acc[o]);
        }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        const int y = y0 + rg * 8 + o;
        if (y < s.h) d.row(f, y)[xb] = (u8)sat_u8_rne(acc[o]);
    }
}

// ---------------------------------------------------------------------------------------
// Sobel.  Tile: 64 x 16 output pixels; LDS holds the (64+2) x (16+2) L tile.
// FROM_RGB: the tile loader converts RGB -> L with Pillow's fixed-point weights.
// ---------------------------------------------------------------------------------------
template <bool FROM_RGB>
__global__ __launch_bounds__(256) void sobel_kernel(View s, View d, int variant) {
    constexpr int TW = 64, THS = 16;
    __shared__ int L[(THS + 2) * (TW + 2)];
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * THS, f = blockIdx.z;
    for (int t = threadIdx.x; t < (THS + 2) * (TW + 2); t += 256) {
        const int ty = t / (TW + 2), tx = t - ty * (TW + 2);
        int gy = y0 - 1 + ty, gx = x0 - 1 + tx;
        if (gy < 0 || gy >= s.h) gy = reflect_sym(gy, s.h);
        if (gx < 0 || gx >= s.w) gx = reflect_sym(gx, s.w);
        const u8* p = s.row(f, gy) + gx * s.c;
        int v;
        if (FROM_RGB) v = (int)(((u32)p[0] * 19595u + (u32)p[1] * 38470u + (u32)p[2] * 7471u + 0x8000u) >> 16);
        else v = p[0];
        L[t] = v;
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, lg = threadIdx.x >> 6;
    const int x = x0 + lx;
    if (x >= s.w) return;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int ly = lg * 4 + o, y = y0 + ly;
        if (y >= s.h) break;
        const int* c = &L[(ly + 1) * (TW + 2) + lx + 1];
        const int a00 = c[-(TW + 2) - 1], a01 = c[-(TW + 2)], a02 = c[-(TW + 2) + 1];
        const int a10 = c[-1], a12 = c[1];
        const int a20 = c[(TW + 2) - 1], a21 = c[(TW + 2)], a22 = c[(TW + 2) + 1];
        const int gx = (a02 - a00) + 2 * (a12 - a10) + (a22 - a20);
        const int gy = (a20 - a00) + 2 * (a21 - a01) + (a22 - a02);
        u32 out;
        if (variant == IMGXF_SOBEL_X_WRAP) out = (u32)gx & 0xffu;
        else if (variant == IMGXF_SOBEL_Y_WRAP) out = (u32)gy & 0xffu;
        else out = sat_u8_rne(__fsqrt_rn((float)(gx * gx + gy * gy)));
        d.row(f, y)[x] = (u8)out;
    }
}

// ---------------------------------------------------------------------------------------
// libImaging ImagingFilter3x3 (Image.filter with a 3x3 ImageFilter.Kernel, e.g. SMOOTH behind
// ImageEnhance.Sharpness, cifar_image_transformations.py:95-99): float32, the exact operation
// order of Filter.c, the one-pixel frame copied from the input.  A lane owns 4 bytes of a row.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int clampi_(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
struct K9 { float k[9]; float off; };
__global__ __launch_bounds__(256) void filter3x3_kernel(View s, View d, K9 K) {
    const int C = s.c;
    const int rowbytes = s.w * C;
    const int nq = (rowbytes + 3) >> 2;
    const int64_t total = (int64_t)s.n * s.h * nq;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int qd = (int)(t % nq);
        const int64_t r = t / nq;
        const int y = (int)(r % s.h), f = (int)(r / s.h);
        const u8* r0 = s.row(f, y);
        const bool inner_row = y > 0 && y < s.h - 1 && s.w >= 3;
        const u8* rm = inner_row ? s.row(f, y - 1) : r0;
        const u8* rp = inner_row ? s.row(f, y + 1) : r0;
        u8* dp = d.row(f, y);
        for (int e = 0; e < 4; ++e) {
            const int b = qd * 4 + e;
            if (b >= rowbytes) break;
            u8 v = r0[b];
            if (inner_row && b >= C && b < rowbytes - C) {
                float ss = K.off;
                ss += ((float)rp[b - C] * K.k[0] + (float)rp[b] * K.k[1]) + (float)rp[b + C] * K.k[2];
                ss += ((float)r0[b - C] * K.k[3] + (float)r0[b] * K.k[4]) + (float)r0[b + C] * K.k[5];
                ss += ((float)rm[b - C] * K.k[6] + (float)rm[b] * K.k[7]) + (float)rm[b + C] * K.k[8];
                v = ss <= 0.0f ? (u8)0 : (ss >= 255.0f ? (u8)255 : (u8)(int)ss);
            }
            dp[b] = v;
        }
    }
}

// 16 output bytes per lane (16-byte aligned rows): per source row one aligned 16-byte load plus the dword on
// either side; every source byte is converted once (16 + 2C conversions per row instead of 48) and the three taps
// of output byte i are elements i, i + C, i + 2C of that row's float window.  Same float32 operation order as
// filter3x3_kernel (Filter.c); the frame's first / last row and the first / last chunk of a row (image border
// pixels are copies) take the per-byte code.
template <int C>
__global__ __launch_bounds__(256) void filter3x3_rows16_kernel(View s, View d, K9 K) {
    const int rowbytes = s.w * C;
    const int nch = (rowbytes + 15) >> 4;
    const int64_t total = (int64_t)s.n * s.h * nch;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int ck = (int)(t % nch);
        const int64_t r = t / nch;
        const int y = (int)(r % s.h), f = (int)(r / s.h);
        const int b0 = ck << 4;
        const u8* r0 = s.row(f, y);
        u8* dp = d.row(f, y);
        const bool inner_row = y > 0 && y < s.h - 1 && s.w >= 3;
        if (inner_row && ck > 0 && b0 + 16 + 4 <= rowbytes) {
            float ss[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) ss[i] = K.off;
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {                      // Filter.c: in1 = row below (y+1), in0 = row, in_1 = row above
                const u8* rp = rr == 0 ? s.row(f, y + 1) : (rr == 1 ? r0 : s.row(f, y - 1));
                const uint4 cv = *(const uint4*)(rp + b0);
                const u32 lv = *(const u32*)(rp + b0 - 4), rv = *(const u32*)(rp + b0 + 16);
                float w[16 + 2 * C];
#pragma unroll
                for (int j = 0; j < C; ++j) w[j] = (float)((lv >> (8 * (4 - C + j))) & 0xffu);
                const u32 cw[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
                for (int j = 0; j < 16; ++j) w[C + j] = (float)((cw[j >> 2] >> (8 * (j & 3))) & 0xffu);
#pragma unroll
                for (int j = 0; j < C; ++j) w[C + 16 + j] = (float)((rv >> (8 * j)) & 0xffu);
                const float k0 = K.k[3 * rr], k1 = K.k[3 * rr + 1], k2 = K.k[3 * rr + 2];
#pragma unroll
                for (int i = 0; i < 16; ++i) ss[i] += (w[i] * k0 + w[i + C] * k1) + w[i + 2 * C] * k2;
            }
            u32 o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = ss[i];
                const u32 q = v <= 0.0f ? 0u : (v >= 255.0f ? 255u : (u32)(int)v);
                o[i >> 2] |= q << (8 * (i & 3));
            }
            *(uint4*)(dp + b0) = make_uint4(o[0], o[1], o[2], o[3]);
            continue;
        }
        const u8* rm = inner_row ? s.row(f, y - 1) : r0;
        const u8* rp = inner_row ? s.row(f, y + 1) : r0;
        for (int e = 0; e < 16; ++e) {
            const int b = b0 + e;
            if (b >= rowbytes) break;
            u8 v = r0[b];
            if (inner_row && b >= C && b < rowbytes - C) {
                float a = K.off;
                a += ((float)rp[b - C] * K.k[0] + (float)rp[b] * K.k[1]) + (float)rp[b + C] * K.k[2];
                a += ((float)r0[b - C] * K.k[3] + (float)r0[b] * K.k[4]) + (float)r0[b + C] * K.k[5];
                a += ((float)rm[b - C] * K.k[6] + (float)rm[b] * K.k[7]) + (float)rm[b + C] * K.k[8];
                v = a <= 0.0f ? (u8)0 : (a >= 255.0f ? (u8)255 : (u8)(int)a);
            }
            dp[b] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------
// libImaging BoxBlur.c (ImageFilter.BoxBlur / GaussianBlur, TransformationPool.defocus_blur,
// cifar_image_transformations.py:72-77).  One pass = ImagingLineBoxBlur: exact uint32
// arithmetic out = (window_sum*ww + (far_l + far_r)*fw + 2^23) >> 24 with replicated edges;
// evaluated directly per output byte (the C code's running sum gives the same integers).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void box_pass_kernel(View s, View d, int radius, u32 ww, u32 fw, int vertical) {
    const int C = s.c;
    const int rowbytes = s.w * C;
    const int64_t total = (int64_t)s.n * s.h * rowbytes;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int b = (int)(t % rowbytes);
        const int64_t r = t / rowbytes;
        const int y = (int)(r % s.h), f = (int)(r / s.h);
        u32 acc = 0, far;
        if (!vertical) {
            const u8* rp = s.row(f, y);
            const int x = b / C, ch = b - x * C;
            for (int i = -radius; i <= radius; ++i) acc += rp[clampi_(x + i, 0, s.w - 1) * C + ch];
            far = (u32)rp[clampi_(x - radius - 1, 0, s.w - 1) * C + ch] + (u32)rp[clampi_(x + radius + 1, 0, s.w - 1) * C + ch];
        } else {
            for (int i = -radius; i <= radius; ++i) acc += s.row(f, clampi_(y + i, 0, s.h - 1))[b];
            far = (u32)s.row(f, clampi_(y - radius - 1, 0, s.h - 1))[b] + (u32)s.row(f, clampi_(y + radius + 1, 0, s.h - 1))[b];
        }
        const u32 bulk = acc * ww + far * fw;
        d.row(f, y)[b] = (u8)((bulk + (1u << 23)) >> 24);
    }
}

// Wide-lane box passes (16-byte aligned rows, radius <= BOX_RMAX): 16 output bytes per lane, whole-dword loads.
//   vertical:   the 2r+3 source rows of the lane's 16-byte column block are summed as packed 16-bit pairs
//               (even / odd bytes of every dword; 13 x 255 fits 16 bits), far rows separately;
//   horizontal: the lane's window (16 + 2 C (r+1) bytes, three aligned 16-byte loads) is unpacked once and the
//               window sums of outputs i, i+C, i+2C ... follow each other by one add and one subtract.
// Same integers as box_pass_kernel: out = (acc * ww + far * fw + 2^23) >> 24 in uint32.
constexpr int BOX_RMAX = 10;    // round 3: up to the radii of TransformationPool.defocus_blur (GaussianBlur radius 10 -> boxes of radius 8)

__global__ __launch_bounds__(256) void box_v16_kernel(View s, View d, int radius, u32 ww, u32 fw) {
    const int nch = (int)(s.rowbytes() >> 4);
    const int64_t total = (int64_t)s.n * s.h * nch;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int ck = (int)(t % nch);
        const int64_t r = t / nch;
        const int y = (int)(r % s.h), f = (int)(r / s.h);
        u32 ae[4] = {0u, 0u, 0u, 0u}, ao[4] = {0u, 0u, 0u, 0u};          // even / odd bytes as 16-bit pairs
        for (int i = -radius; i <= radius; ++i) {
            const uint4 v = *(const uint4*)(s.row(f, clampi_(y + i, 0, s.h - 1)) + (ck << 4));
            const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) { ae[q] += w[q] & 0x00ff00ffu; ao[q] += (w[q] >> 8) & 0x00ff00ffu; }
        }
        const uint4 va = *(const uint4*)(s.row(f, clampi_(y - radius - 1, 0, s.h - 1)) + (ck << 4));
        const uint4 vb = *(const uint4*)(s.row(f, clampi_(y + radius + 1, 0, s.h - 1)) + (ck << 4));
        const u32 wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
        u32 o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32 fe = (wa[q] & 0x00ff00ffu) + (wb[q] & 0x00ff00ffu), fo = ((wa[q] >> 8) & 0x00ff00ffu) + ((wb[q] >> 8) & 0x00ff00ffu);
            const u32 b0 = ((ae[q] & 0xffffu) * ww + (fe & 0xffffu) * fw + (1u << 23)) >> 24;
            const u32 b1 = ((ao[q] & 0xffffu) * ww + (fo & 0xffffu) * fw + (1u << 23)) >> 24;
            const u32 b2 = ((ae[q] >> 16) * ww + (fe >> 16) * fw + (1u << 23)) >> 24;
            const u32 b3 = ((ao[q] >> 16) * ww + (fo >> 16) * fw + (1u << 23)) >> 24;
            o[q] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
        }
        *(uint4*)(d.row(f, y) + (ck << 4)) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// Vertical pass, a RUN of rows per lane (round 3): the window sum of the lane's 16-byte column block is carried from row to
// row — window(y+1) = window(y) + src[clamp(y+1+R)] - src[clamp(y-R)], termwise true with the replicated edges — so a row
// costs two 16-byte loads (the row entering the window, which is also this row's lower far tap, and the row leaving it,
// which is the next row's upper far tap) and one unpacking each, instead of 2R+3 loads and unpackings.  Same integers.
constexpr int BOX_VRUN = 32;

__global__ __launch_bounds__(256) void box_v16_run_kernel(View s, View d, int radius, u32 ww, u32 fw) {
    const int nch = (int)(s.rowbytes() >> 4), nruns = (s.h + BOX_VRUN - 1) / BOX_VRUN;
    const int64_t total = (int64_t)s.n * nruns * nch;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int ck = (int)(t % nch);
        const int64_t r = t / nch;
        const int run = (int)(r % nruns), f = (int)(r / nruns);
        const int ya = run * BOX_VRUN, yb = min(s.h, ya + BOX_VRUN);
        const u8* col = s.p + (int64_t)f * s.fs + (ck << 4);
        auto load = [&](int y, u32 (&e)[4], u32 (&o)[4]) {
            const uint4 v = *(const uint4*)(col + (int64_t)clampi_(y, 0, s.h - 1) * s.rs);
            const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) { e[q] = w[q] & 0x00ff00ffu; o[q] = (w[q] >> 8) & 0x00ff00ffu; }
        };
        u32 ae[4] = {0u, 0u, 0u, 0u}, ao[4] = {0u, 0u, 0u, 0u};          // window sum of row y: even / odd bytes as 16-bit pairs
        for (int i = -radius; i <= radius; ++i) {
            u32 e[4], o[4];
            load(ya + i, e, o);
#pragma unroll
            for (int q = 0; q < 4; ++q) { ae[q] += e[q]; ao[q] += o[q]; }
        }
        u32 le[4], lo[4];                                                 // upper far tap of row y: src[clamp(y - R - 1)]
        load(ya - radius - 1, le, lo);
        for (int y = ya; y < yb; ++y) {
            u32 he[4], ho[4], se[4], so[4];
            load(y + radius + 1, he, ho);                                 // lower far tap, and the row entering the window
            load(y - radius, se, so);                                     // the row leaving the window = the next row's upper far tap
            u32 o4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32 fe = le[q] + he[q], fo = lo[q] + ho[q];
                const u32 b0 = ((ae[q] & 0xffffu) * ww + (fe & 0xffffu) * fw + (1u << 23)) >> 24;
                const u32 b1 = ((ao[q] & 0xffffu) * ww + (fo & 0xffffu) * fw + (1u << 23)) >> 24;
                const u32 b2 = ((ae[q] >> 16) * ww + (fe >> 16) * fw + (1u << 23)) >> 24;
                const u32 b3 = ((ao[q] >> 16) * ww + (fo >> 16) * fw + (1u << 23)) >> 24;
                o4[q] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
                ae[q] += he[q] - se[q]; ao[q] += ho[q] - so[q];           // (16-bit fields: the window sum never goes negative, no borrow crosses)
                le[q] = se[q]; lo[q] = so[q];
            }
            *(uint4*)(d.p + (int64_t)f * d.fs + (int64_t)y * d.rs + (ck << 4)) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
        }
    }
}

template <int C, int R>
__global__ __launch_bounds__(256) void box_h16_kernel(View s, View d, u32 ww, u32 fw) {
    constexpr int HALO = C * (R + 1);                        // bytes of window on either side of the lane's 16
    constexpr int NB = (HALO + 15) / 16;                     // 16-byte blocks on either side (1 for the radii of round 2, up to 3)
    const int rowbytes = s.w * C;
    const int nch = (rowbytes + 15) >> 4;
    const int64_t total = (int64_t)s.n * s.h * nch;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int ck = (int)(t % nch);
        const int64_t r = t / nch;
        const int y = (int)(r % s.h), f = (int)(r / s.h);
        const int b0 = ck << 4;
        const u8* rp = s.row(f, y);
        u8* dp = d.row(f, y);
        if (ck >= NB && b0 + 16 * (NB + 1) <= rowbytes) {    // the 2 NB + 1 blocks around b0 lie inside the row
            u32 src[4 * (2 * NB + 1)];
#pragma unroll
            for (int q = 0; q < 2 * NB + 1; ++q) {
                const uint4 v = *(const uint4*)(rp + b0 + 16 * (q - NB));
                src[4 * q] = v.x; src[4 * q + 1] = v.y; src[4 * q + 2] = v.z; src[4 * q + 3] = v.w;
            }
            u32 w[16 + 2 * HALO];                            // w[j] = byte b0 - HALO + j
#pragma unroll
            for (int j = 0; j < 16 + 2 * HALO; ++j) {
                const int p = 16 * NB - HALO + j;
                w[j] = (src[p >> 2] >> (8 * (p & 3))) & 0xffu;
            }
            u32 o[4] = {0u, 0u, 0u, 0u};
            u32 acc[C];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ph = i % C;
                if (i < C) {
                    u32 a = 0;
#pragma unroll
                    for (int k = -R; k <= R; ++k) a += w[HALO + i + C * k];
                    acc[ph] = a;
                } else {
                    acc[ph] += w[HALO + i + C * R] - w[HALO + i - C * (R + 1)];
                }
                const u32 far = w[HALO + i - C * (R + 1)] + w[HALO + i + C * (R + 1)];
                o[i >> 2] |= ((acc[ph] * ww + far * fw + (1u << 23)) >> 24) << (8 * (i & 3));
            }
            *(uint4*)(dp + b0) = make_uint4(o[0], o[1], o[2], o[3]);
            continue;
        }
        for (int e = 0; e < 16; ++e) {                       // row ends: replicated edge pixels, per byte
            const int b = b0 + e;
            if (b >= rowbytes) break;
            const int x = b / C, ch = b - x * C;
            u32 a = 0;
            for (int i = -R; i <= R; ++i) a += rp[clampi_(x + i, 0, s.w - 1) * C + ch];
            const u32 far = (u32)rp[clampi_(x - R - 1, 0, s.w - 1) * C + ch] + (u32)rp[clampi_(x + R + 1, 0, s.w - 1) * C + ch];
            dp[b] = (u8)((a * ww + far * fw + (1u << 23)) >> 24);
        }
    }
}

template <int C>
static bool launch_box_h16(const View& s, const View& d, int radius, u32 ww, u32 fw, unsigned blocks, hipStream_t st) {
    if ((int64_t)s.rowbytes() < 16 * (2 * ((C * (radius + 1) + 15) / 16) + 1)) return false;    // no lane would take the wide path
    switch (radius) {
#define IMGXF_BH(r) case r: hipLaunchKernelGGL((box_h16_kernel<C, r>), dim3(blocks), dim3(256), 0, st, s, d, ww, fw); return true;
        IMGXF_BH(0) IMGXF_BH(1) IMGXF_BH(2) IMGXF_BH(3) IMGXF_BH(4) IMGXF_BH(5) IMGXF_BH(6) IMGXF_BH(7) IMGXF_BH(8) IMGXF_BH(9) IMGXF_BH(10)
#undef IMGXF_BH
        default: return false;
    }
}

} // namespace imgxf

using namespace imgxf;

// BoxBlur.c _gaussian_blur_radius: float variables, double sqrt / floor (built un-contracted)
static float gaussian_box_radius(float radius, int passes) {
    float sigma2, L, l, a;
    sigma2 = radius * radius / passes;
    L = sqrt(12.0 * sigma2 + 1.0);
    l = floor((L - 1.0) / 2.0);
    a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2);
    a /= 6 * (sigma2 - (l + 1) * (l + 1));
    return l + a;
}

IMGXF_API int imgxf_box_blur_u8(const imgxf_view* src, const imgxf_view* dst, float xradius, float yradius,
                                int passes, void* workspace, size_t workspace_bytes, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!same_geometry(src, dst)) return IMGXF_ERR_SHAPE;
    if (passes < 1 || passes > 16 || !(xradius >= 0.0f) || !(yradius >= 0.0f) || xradius > 16384.f || yradius > 16384.f)
        return IMGXF_ERR_ARG;
    if (empty_view(src)) return IMGXF_OK;
    const View s = make_view(src), d = make_view(dst);
    const size_t need = (size_t)s.n * s.h * s.rowbytes();
    const int total_passes = (xradius != 0.0f ? passes : 0) + (yradius != 0.0f ? passes : 0);
    if (total_passes > 1 && (!workspace || workspace_bytes < need)) return IMGXF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    View ws = s;                       // contiguous scratch image with the same geometry
    ws.p = (u8*)workspace; ws.rs = s.rowbytes(); ws.fs = ws.rs * s.h;
    if (total_passes == 0) {
        for (int f = 0; f < s.n; ++f) {
            hipError_t e = hipMemcpy2DAsync(d.p + f * d.fs, d.rs, s.p + f * s.fs, s.rs, (size_t)s.rowbytes(), s.h,
                                            hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) return (int)e;
        }
        return IMGXF_OK;
    }
    int64_t blocks = ((int64_t)need + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    View cur = s;
    int done = 0;
    for (int axis = 0; axis < 2; ++axis) {
        const float fr = axis == 0 ? xradius : yradius;
        if (fr == 0.0f) continue;
        const int radius = (int)fr;
        const u32 ww = (u32)((float)(1u << 24) / (fr * 2 + 1));
        const u32 fw = ((u32)(1 << 24) - (u32)(radius * 2 + 1) * ww) / 2;
        for (int p = 0; p < passes; ++p) {
            // the last pass must land in dst: alternate so that parity works out
            const bool to_dst = ((total_passes - 1 - done) & 1) == 0;
            const View out = to_dst ? d : ws;
            // wide-lane kernels for 16-byte aligned images and the radii GaussianBlur produces (radius <= 4)
            const bool al16 = ((((uintptr_t)cur.p) | (uintptr_t)cur.rs | (uintptr_t)cur.fs | ((uintptr_t)out.p) | (uintptr_t)out.rs | (uintptr_t)out.fs) & 15) == 0 &&
                              s.rowbytes() % 16 == 0 && s.rowbytes() >= 48 && !knob_set(K_BOX_BYTES);
            const unsigned b16 = (unsigned)std::min<int64_t>(32768, ((int64_t)s.n * s.h * (s.rowbytes() >> 4) + 255) / 256);
            bool launched = false;
            if (al16 && radius <= BOX_RMAX) {
                if (axis == 1 && !knob_set(K_NO_FAST_LEFTOVERS)) {
                    const unsigned bv = (unsigned)std::min<int64_t>(32768, ((int64_t)s.n * ((s.h + BOX_VRUN - 1) / BOX_VRUN) * (s.rowbytes() >> 4) + 255) / 256);
                    hipLaunchKernelGGL(box_v16_run_kernel, dim3(bv), dim3(256), 0, st, cur, out, radius, ww, fw); launched = true;
                }
                else if (axis == 1) { hipLaunchKernelGGL(box_v16_kernel, dim3(b16), dim3(256), 0, st, cur, out, radius, ww, fw); launched = true; }
                else if (s.c == 1) launched = launch_box_h16<1>(cur, out, radius, ww, fw, b16, st);
                else if (s.c == 3) launched = launch_box_h16<3>(cur, out, radius, ww, fw, b16, st);
                else if (s.c == 4) launched = launch_box_h16<4>(cur, out, radius, ww, fw, b16, st);
            }
            if (!launched)
                hipLaunchKernelGGL(box_pass_kernel, dim3((unsigned)blocks), dim3(256), 0, st, cur, out, radius, ww, fw, axis);
            cur = out;
            ++done;
        }
    }
    return launch_status();
}

IMGXF_API int imgxf_gaussian_blur_pil_u8(const imgxf_view* src, const imgxf_view* dst, float radius,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    if (!(radius >= 0.0f)) return IMGXF_ERR_ARG;
    const float r = gaussian_box_radius(radius, 3);
    return imgxf_box_blur_u8(src, dst, r, r, 3, workspace, workspace_bytes, stream);
}

IMGXF_API int imgxf_filter3x3_u8(const imgxf_view* src, const imgxf_view* dst, const float* kernel9,
                                 float scale, float offset, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!kernel9) return IMGXF_ERR_NULL;
    if (!same_geometry(src, dst)) return IMGXF_ERR_SHAPE;
    if (scale == 0.0f) return IMGXF_ERR_ARG;
    if (empty_view(src)) return IMGXF_OK;
    K9 K;
    for (int i = 0; i < 9; ++i) K.k[i] = kernel9[i] / scale;      // FLOAT32 division, as _imaging.c does
    K.off = offset + 0.5f;
    const View s = make_view(src), d = make_view(dst);
    if (((((uintptr_t)s.p) | (uintptr_t)s.rs | (uintptr_t)s.fs | ((uintptr_t)d.p) | (uintptr_t)d.rs | (uintptr_t)d.fs) & 15) == 0 &&
        s.rowbytes() >= 48 && !knob_set(K_FILTER3X3_BYTES)) {
        const int64_t total16 = (int64_t)s.n * s.h * ((s.rowbytes() + 15) >> 4);
        int64_t blocks16 = (total16 + 255) / 256;
        if (blocks16 > 32768) blocks16 = 32768;
        switch (s.c) {
            case 1: hipLaunchKernelGGL((filter3x3_rows16_kernel<1>), dim3((unsigned)blocks16), dim3(256), 0, (hipStream_t)stream, s, d, K); return launch_status();
            case 3: hipLaunchKernelGGL((filter3x3_rows16_kernel<3>), dim3((unsigned)blocks16), dim3(256), 0, (hipStream_t)stream, s, d, K); return launch_status();
            case 4: hipLaunchKernelGGL((filter3x3_rows16_kernel<4>), dim3((unsigned)blocks16), dim3(256), 0, (hipStream_t)stream, s, d, K); return launch_status();
            default: break;
        }
    }
    const int64_t total = (int64_t)s.n * s.h * ((s.rowbytes() + 3) >> 2);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(filter3x3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, s, d, K);
    return launch_status();
}

IMGXF_API int imgxf_conv2d_u8(const imgxf_view* src, const imgxf_view* dst, const float* kernel,
                              int kh, int kw, int border, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!kernel) return IMGXF_ERR_NULL;
    if (!same_geometry(src, dst)) return IMGXF_ERR_SHAPE;
    if (kh < 1 || kw < 1 || !(kh & 1) || !(kw & 1) || kh > 15 || kw > 15) return IMGXF_ERR_ARG;
    if (border != IMGXF_BORDER_REFLECT_101 && border != IMGXF_BORDER_REFLECT) return IMGXF_ERR_ARG;
    if (empty_view(src)) return IMGXF_OK;
    // Rank-1 kernels — TransformationPool.motion_blur's single row of 1/size
    // (/root/reference/pipenline/cifar_image_transformations.py:113-118), box filters, any outer product — are
    // separable: K = ky (x) kx with kx = the pivot row and ky = the pivot column / pivot.  They run on the
    // separable kernels (marching / matrix-core Gaussians: 0.6 - 1.1 ms per 64 4K frames instead of 4 - 14 ms
    // here); the factorisation reproduces K to 1e-6 of its largest entry, inside the 1e-5 contract of this
    // float filter, and both evaluate sum(w p) in fp32 with one rounding to uint8.
    if (!knob_set(K_CONV2D_NO_SEPARABLE)) {
        int pj = 0, pi = 0;
        double pmax = 0.0;
        for (int j = 0; j < kh; ++j)
            for (int i = 0; i < kw; ++i)
                if (fabs((double)kernel[j * kw + i]) > pmax) { pmax = fabs((double)kernel[j * kw + i]); pj = j; pi = i; }
        bool rank1 = pmax > 0.0;
        const double P = kernel[pj * kw + pi];
        for (int j = 0; j < kh && rank1; ++j)
            for (int i = 0; i < kw; ++i)
                if (fabs((double)kernel[j * kw + i] * P - (double)kernel[j * kw + pi] * (double)kernel[pj * kw + i]) > 1e-6 * P * P) { rank1 = false; break; }
        if (rank1) {
            float kx[15], ky[15];
            for (int i = 0; i < kw; ++i) kx[i] = kernel[pj * kw + i];
            for (int j = 0; j < kh; ++j) ky[j] = (float)((double)kernel[j * kw + pi] / P);
            const int rc = imgxf_sepconv_u8(src, dst, kx, kw, ky, kh, border, nullptr, stream);
            if (rc != IMGXF_ERR_UNSUPPORTED) return rc;
        }
    }
    KernelTaps K; memset(&K, 0, sizeof(K));
    for (int i = 0; i < kh * kw; ++i) K.w[i] = kernel[i];
    const View s = make_view(src), d = make_view(dst);
    const size_t lds = sizeof(float) * (size_t)(128 + (kw - 1) * s.c) * (16 + kh - 1);
    dim3 grid((unsigned)((s.rowbytes() + 127) / 128), (unsigned)((s.h + 15) / 16), (unsigned)s.n);
    hipLaunchKernelGGL(conv2d_kernel, grid, dim3(256), lds, (hipStream_t)stream, s, d, K, kh, kw, border);
    return launch_status();
}

IMGXF_API int imgxf_sobel_u8(const imgxf_view* src, const imgxf_view* dst, int variant, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!same_geometry(src, dst) || src->c != 1) return IMGXF_ERR_SHAPE;
    if (variant < 0 || variant > 2) return IMGXF_ERR_ARG;
    if (empty_view(src)) return IMGXF_OK;
    const View s = make_view(src), d = make_view(dst);
    const bool no_march = knob_set(K_NO_MARCH);
    if (!no_march && sobel_march_eligible(s, d, 1)) {
        switch (variant) {
            case IMGXF_SOBEL_X_WRAP: return launch_sobel_march<1, IMGXF_SOBEL_X_WRAP>(s, d, (hipStream_t)stream);
            case IMGXF_SOBEL_Y_WRAP: return launch_sobel_march<1, IMGXF_SOBEL_Y_WRAP>(s, d, (hipStream_t)stream);
            default: return launch_sobel_march<1, IMGXF_SOBEL_MAGNITUDE>(s, d, (hipStream_t)stream);
        }
    }
    dim3 grid((unsigned)((s.w + 63) / 64), (unsigned)((s.h + 15) / 16), (unsigned)s.n);
    hipLaunchKernelGGL((sobel_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, s, d, variant);
    return launch_status();
}

IMGXF_API int imgxf_rgb_sobel_u8(const imgxf_view* src, const imgxf_view* dst, int variant, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!same_nhw(src, dst) || src->c != 3 || dst->c != 1) return IMGXF_ERR_SHAPE;
    if (variant < 0 || variant > 2) return IMGXF_ERR_ARG;
    if (empty_view(src)) return IMGXF_OK;
    const View s = make_view(src), d = make_view(dst);
    const bool no_march = knob_set(K_NO_MARCH);
    if (!no_march && sobel_march_eligible(s, d, 3)) {
        switch (variant) {
            case IMGXF_SOBEL_X_WRAP: return launch_sobel_march<3, IMGXF_SOBEL_X_WRAP>(s, d, (hipStream_t)stream);
            case IMGXF_SOBEL_Y_WRAP: return launch_sobel_march<3, IMGXF_SOBEL_Y_WRAP>(s, d, (hipStream_t)stream);
            default: return launch_sobel_march<3, IMGXF_SOBEL_MAGNITUDE>(s, d, (hipStream_t)stream);
        }
    }
    dim3 grid((unsigned)((s.w + 63) / 64), (unsigned)((s.h + 15) / 16), (unsigned)s.n);
    hipLaunchKernelGGL((sobel_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, s, d, variant);
    return launch_status();
}

IMGXF_API int imgxf_rgb_sobel_mag_u8(const imgxf_view* src, const imgxf_view* dst, void* stream) {
    return imgxf_rgb_sobel_u8(src, dst, IMGXF_SOBEL_MAGNITUDE, stream);
}
