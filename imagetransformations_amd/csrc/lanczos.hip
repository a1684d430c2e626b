// Lanczos-3 resize: the body behind img.resize((nw,nh), Image.Resampling.LANCZOS)
// (/root/reference/transformation.py:179), i.e. libImaging Resample.c for 8-bit images:
// host-side double-precision coefficient windows normalised to 22-bit integers, then a
// horizontal integer MAC pass into a uint8 intermediate followed by a vertical pass.
#include "imgxf_common.h"
#include <math.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <stdlib.h>
#include <stdio.h>

#define PRECISION_BITS (32 - 8 - 2)

struct imgxf_lanczos_plan {
    int in_h, in_w, out_h, out_w, c, max_frames;
    int ksx, ksy;            // coefficients per output sample (row length of the tables)
    int* d_bounds_x;         // [out_w][2] (xmin, count)
    int* d_kk_x;             // [out_w][ksx]
    int* d_bounds_y;         // [out_h][2]
    int* d_kk_y;             // [out_h][ksy]
    // window-normalised copies for the fast kernels: every window is KP samples wide and lies
    // fully inside the source (start = min(xmin, in - KP)), coefficients shifted accordingly
    int kpx, kpy;            // padded window widths (kpx in {8,12,16}; 0 = fast kernel not usable)
    int hspan;               // max over 1024-column groups of the 16-B aligned source byte span (+16), 0 = unknown
    int* d_start_x;          // [out_w]
    int* d_pk_x;             // [out_w][kpx]
    int* d_start_y;          // [out_h]
    int* d_pk_y;             // [out_h][kpy]
    // 4-row groups for resample_v4_kernel: union window of KU4 input rows per group, coefficients of
    // each of the group's rows re-indexed to it (0 = not usable)
    int ku4;
    int* d_start4_y;         // [ceil(out_h/4)]
    int* d_pk4_y;            // [ceil(out_h/4)][4][ku4]
    uint8_t* d_tmp;          // [max_frames][in_h][out_w][c], only when both passes run
    int need_h, need_v;
    // output window (crop fused into the resize): the plan produces rows [wy, wy+wh) x columns
    // [wx, wx+ww) of the virtual out_h x out_w result; the H pass then only filters source rows
    // [ry0, ry0+rh), the ones the window's vertical taps touch
    int wx, wy, ww, wh, ry0, rh;
    // digit tables of the fused matrix-core kernel (resample_mfma.inc); mf_nkh == 0: not usable
    int mf_nkh, mf_ng, mf_nob, mf_nchunks;
    int *d_mf_ws, *d_mf_kcol, *d_mf_sched, *d_mf_chunks, *d_mf_krow;
    int8_t *d_mf_wh, *d_mf_wv;
};

namespace imgxf {

typedef uint32_t u32_ua __attribute__((aligned(1)));
typedef uint32_t u32x2_ua __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t u32x4_ua __attribute__((ext_vector_type(4), aligned(1)));

static inline double sinc_filter(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return sin(x) / x;
}
static inline double lanczos_filter(double x) {
    if (-3.0 <= x && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3);
    return 0.0;
}
// the other filters of libImaging Resample.c (Image.resize's default is BICUBIC)
static inline double box_filter(double x) { return (x > -0.5 && x <= 0.5) ? 1.0 : 0.0; }
static inline double bilinear_filter(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}
static inline double hamming_filter(double x) {
    if (x < 0.0) x = -x;
    if (x == 0.0) return 1.0;
    if (x >= 1.0) return 0.0;
    x = x * M_PI;
    return sin(x) / x * (0.54f + 0.46f * cos(x));      // float literals, as in Resample.c
}
static inline double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
static inline double filter_support(int filter) {
    switch (filter) {
        case IMGXF_RESAMPLE_BOX: return 0.5;
        case IMGXF_RESAMPLE_BILINEAR: case IMGXF_RESAMPLE_HAMMING: return 1.0;
        case IMGXF_RESAMPLE_BICUBIC: return 2.0;
        default: return 3.0;
    }
}
static inline double filter_value(int filter, double x) {
    switch (filter) {
        case IMGXF_RESAMPLE_BOX: return box_filter(x);
        case IMGXF_RESAMPLE_BILINEAR: return bilinear_filter(x);
        case IMGXF_RESAMPLE_HAMMING: return hamming_filter(x);
        case IMGXF_RESAMPLE_BICUBIC: return bicubic_filter(x);
        default: return lanczos_filter(x);
    }
}

// precompute_coeffs + normalize_coeffs_8bpc (whole-image box)
static int build_coeffs(int in_size, int out_size, int filter, std::vector<int>& bounds, std::vector<int>& kk) {
    double scale, filterscale;
    filterscale = scale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = filter_support(filter) * filterscale;
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
            const double w = filter_value(filter, (x + xmin - center + 0.5) * ss);
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

// window-normalised tables: start[i] = min(xmin, in_size - KP); pk[i][start-relative] = coeff
static void build_padded(int in_size, int out_size, int ksize, int KP, const std::vector<int>& bounds,
                         const std::vector<int>& kk, std::vector<int>& start, std::vector<int>& pk) {
    start.assign(out_size, 0);
    pk.assign((size_t)out_size * KP, 0);
    for (int i = 0; i < out_size; ++i) {
        const int xmin = bounds[2 * i], cnt = bounds[2 * i + 1];
        int st = xmin;
        if (st > in_size - KP) st = in_size - KP;
        start[i] = st;
        for (int x = 0; x < cnt; ++x) pk[(size_t)i * KP + (xmin - st) + x] = kk[(size_t)i * ksize + x];
    }
}

// pixel (8 bits) x 22-bit coefficient: the 24-bit multiplier runs at full rate, a 32-bit
// v_mul_lo_u32 at a quarter of it (the compiler cannot see the coefficient's range)
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }

__device__ __forceinline__ u8 clip8(int v) {
    v >>= PRECISION_BITS;
    // keep the shift and the clamp apart: hipcc (ROCm 7.2) otherwise fuses pairs of them into
    // v_ashr_pk_u8_i32 and ORs further bytes into its result as if bits 31:16 were zero, which
    // they are not on gfx950 (observed: every third byte of a packed dword corrupted)
    asm volatile("" : "+v"(v));
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
            for (int j = 0; j < C; ++j) acc[j] += mul24((int)sp[x * C + j], w);
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
        for (int y = 0; y < cnt; ++y) acc += mul24((int)sp[(int64_t)y * s.rs], k[y]);
        d.row(f, yy)[xb] = clip8(acc);
    }
}

// ---- fast horizontal pass (RGB): a lane owns 4 adjacent output columns, keeps their 4 x KP
// coefficients in registers and marches down ROWS_PER_BLOCK rows; each column's window is
// KP*3 bytes = KP*3/4 unaligned dword loads that start exactly at the window's first pixel
// (no shifting), every byte feeds one v_mad_i32_i24.  Windows never leave the row (tables are
// start-clamped on the host) and KP*3 is a multiple of 4, so no load passes the row end.
template <int KP>
__global__ __launch_bounds__(256) void resample_h_fast_kernel(View s, View d, const int* start, const int* pk,
                                                              int rows_per_block) {
    constexpr int C = 3, ND = KP * 3 / 4;
    const int xg = blockIdx.x * 256 + threadIdx.x;          // group of 4 output columns
    const int x0 = xg * 4;
    const int f = blockIdx.z;
    const int y_begin = blockIdx.y * rows_per_block, y_end = min(d.h, y_begin + rows_per_block);
    if (x0 >= d.w) return;
    int st[4];
    int kc[4][KP];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xx = min(x0 + j, d.w - 1);
        st[j] = start[xx] * C;
#pragma unroll
        for (int t = 0; t < KP; ++t) kc[j][t] = pk[(int64_t)xx * KP + t];
    }
    const int npx = min(4, d.w - x0);
    for (int y = y_begin; y < y_end; ++y) {
        const u8* rowp = s.row(f, y);
        u32 out[4 * C];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32 w[ND];
            const u8* wp = rowp + st[j];
            // widest loads that tile the window exactly (byte-aligned addresses are fine for
            // global loads): 24 B = x4 + x2, 36 B = 2 x4 + x1, 48 B = 3 x4.  The L1 address path
            // charges per wave-level load instruction, so 2-3 wide loads beat 6-12 dword loads.
#pragma unroll
            for (int q = 0; q + 4 <= ND; q += 4) {
                const u32x4_ua v = *(const u32x4_ua*)(wp + 4 * q);
                w[q] = v.x; w[q + 1] = v.y; w[q + 2] = v.z; w[q + 3] = v.w;
            }
            if constexpr (ND % 4 >= 2) {
                const u32x2_ua v = *(const u32x2_ua*)(wp + 4 * (ND & ~3));
                w[ND & ~3] = v.x; w[(ND & ~3) + 1] = v.y;
            }
            if constexpr (ND % 2 == 1) w[ND - 1] = *(const u32_ua*)(wp + 4 * (ND - 1));
            int acc[C];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) acc[ch] = 1 << (PRECISION_BITS - 1);
#pragma unroll
            for (int t = 0; t < KP; ++t) {
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    const int b = t * C + ch;
                    acc[ch] += mul24((int)((w[b >> 2] >> (8 * (b & 3))) & 0xffu), kc[j][t]);
                }
            }
#pragma unroll
            for (int ch = 0; ch < C; ++ch) out[j * C + ch] = clip8(acc[ch]);
        }
        u8* dp = d.row(f, y) + x0 * C;
        if (npx == 4 && (((uintptr_t)dp) & 3) == 0) {
#pragma unroll
            for (int q = 0; q < C; ++q)
                ((u32*)dp)[q] = out[4 * q] | (out[4 * q + 1] << 8) | (out[4 * q + 2] << 16) | (out[4 * q + 3] << 24);
        } else {
            for (int e = 0; e < npx * C; ++e) {
                u32 v = 0;
#pragma unroll
                for (int k = 0; k < 4 * C; ++k) if (k == e) v = out[k];
                dp[e] = (u8)v;
            }
        }
    }
}

// ---- horizontal pass through LDS (RGB, 16-byte aligned source rows).  PMC on the kernel above:
// the four windows of a lane overlap almost completely and neighbouring lanes' windows do too,
// so a wave requests ~6 KiB per row to use ~0.7 KiB of it, and the L1 address path is charged per
// lane request (~58 TCP accesses per wave-level x4 load, TCP 80 % busy, VALU 31 %).  Here the
// workgroup copies the byte span its 1024 output columns need into LDS once per row with
// coalesced 16-byte loads (next row's chunks are in flight while the current row is filtered;
// two LDS buffers, one barrier per row), and every window is read from LDS as aligned dwords
// plus a funnel shift.
template <int KP>
__global__ __launch_bounds__(256) void resample_h_lds_kernel(View s, View d, const int* start, const int* pk,
                                                             int rows_per_block, int bufbytes) {
    constexpr int C = 3, ND = KP * 3 / 4, MAXCH = 2;         // <= 2 chunks of 16 B per lane and row
    extern __shared__ __attribute__((aligned(16))) u8 hbuf[];  // 2 x bufbytes
    const int tid = threadIdx.x;
    const int x_first = blockIdx.x * 1024;
    const int x0 = x_first + tid * 4;
    const int f = blockIdx.z;
    const int y_begin = blockIdx.y * rows_per_block, y_end = min(d.h, y_begin + rows_per_block);
    const int x_last = min(x_first + 1023, d.w - 1);
    // byte span of the source row this workgroup needs, from a 16-byte aligned base
    const int base = (start[x_first] * C) & ~15;
    const int span_end = (start[x_last] + KP) * C;             // exclusive; <= row bytes (start-clamped tables)
    const int nchunks = (span_end - base + 15) >> 4;           // <= bufbytes / 16 (host-checked)
    const int rowbytes = s.w * C;

    const bool live = x0 < d.w;
    int so[4], sr[4];                                          // window offset in the buffer: dword index, byte shift
    int kc[4][KP];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xx = min(x0 + j, d.w - 1);
        const int o = start[xx] * C - base;
        so[j] = o & ~3; sr[j] = o & 3;
#pragma unroll
        for (int t = 0; t < KP; ++t) kc[j][t] = pk[(int64_t)xx * KP + t];
    }
    const int npx = min(4, d.w - x0);

    uint4 pre[MAXCH];
    auto fetch = [&](int y) {
        const u8* rowp = s.row(f, y) + base;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            const int ck = tid + 256 * c;
            pre[c] = make_uint4(0, 0, 0, 0);
            if (ck < nchunks && base + 16 * ck + 16 <= rowbytes) pre[c] = *(const uint4*)(rowp + 16 * ck);
        }
    };
    auto deposit = [&](u8* buf) {
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            const int ck = tid + 256 * c;
            if (ck < nchunks) *(uint4*)(buf + 16 * ck) = pre[c];
        }
    };

    if (y_begin < y_end) fetch(y_begin);
    for (int y = y_begin; y < y_end; ++y) {
        u8* buf = hbuf + ((y - y_begin) & 1) * bufbytes;
        deposit(buf);
        __syncthreads();
        if (y + 1 < y_end) fetch(y + 1);                       // lands while this row is filtered
        if (live) {
            u32 out[4 * C];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32 raw[ND + 1];
                const u32* wp = (const u32*)(buf + so[j]);
#pragma unroll
                for (int q = 0; q <= ND; ++q) raw[q] = wp[q];
                u32 w[ND];
#pragma unroll
                for (int q = 0; q < ND; ++q) w[q] = __builtin_amdgcn_alignbyte(raw[q + 1], raw[q], (u32)sr[j]);
                int acc[C];
#pragma unroll
                for (int ch = 0; ch < C; ++ch) acc[ch] = 1 << (PRECISION_BITS - 1);
#pragma unroll
                for (int t = 0; t < KP; ++t) {
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) {
                        const int b = t * C + ch;
                        acc[ch] += mul24((int)((w[b >> 2] >> (8 * (b & 3))) & 0xffu), kc[j][t]);
                    }
                }
#pragma unroll
                for (int ch = 0; ch < C; ++ch) out[j * C + ch] = clip8(acc[ch]);
            }
            u8* dp = d.row(f, y) + x0 * C;
            if (npx == 4 && (((uintptr_t)dp) & 3) == 0) {
#pragma unroll
                for (int q = 0; q < C; ++q)
                    ((u32*)dp)[q] = out[4 * q] | (out[4 * q + 1] << 8) | (out[4 * q + 2] << 16) | (out[4 * q + 3] << 24);
            } else {
                for (int e = 0; e < npx * C; ++e) {
                    u32 v = 0;
#pragma unroll
                    for (int k = 0; k < 4 * C; ++k) if (k == e) v = out[k];
                    dp[e] = (u8)v;
                }
            }
        }
    }
}

// ---- fast vertical pass: a lane owns 16 bytes of an output row; the KP window rows are read
// with coalesced 16-byte loads, the coefficient of a row is wave-uniform (scalar load), rows
// with a zero coefficient are skipped.
__global__ __launch_bounds__(256) void resample_v_fast_kernel(View s, View d, const int* start, const int* pk, int KP) {
    const int rowbytes = d.w * d.c;                         // multiple of 16 (host-checked)
    const int nchunks = rowbytes >> 4;
    const int ck = blockIdx.x * 256 + threadIdx.x;
    const int yy = blockIdx.y, f = blockIdx.z;
    if (ck >= nchunks) return;
    const int y0 = start[yy];
    const int* k = pk + (int64_t)yy * KP;
    int acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 1 << (PRECISION_BITS - 1);
    const u8* sp = s.row(f, y0) + (ck << 4);
    for (int t = 0; t < KP; ++t) {
        const int c = k[t];
        if (c != 0) {
            const uint4 q = *(const uint4*)(sp + (int64_t)t * s.rs);
            const u32 w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += mul24((int)((w[e >> 2] >> (8 * (e & 3))) & 0xffu), c);
        }
    }
    u32 o[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e >> 2] |= (u32)clip8(acc[e]) << (8 * (e & 3));
    *(uint4*)(d.row(f, yy) + (ck << 4)) = make_uint4(o[0], o[1], o[2], o[3]);
}

// ---- vertical pass, 4 output rows per lane.  The one-row kernel re-reads its KP input rows
// for every output row (7x the output size through L2).  Four consecutive output rows share a
// union window of KU input rows (KP + their window shifts); the host re-indexes each row's
// coefficients to that window (zero outside its own), so the lane reads each input row of the
// window once and feeds up to four accumulator sets: ~3 loads per output row instead of 7, no
// dynamic register indexing.  Coefficients are wave-uniform (scalar loads), zero ones skipped.
template <int KU>
__global__ __launch_bounds__(256) void resample_v4_kernel(View s, View d, const int* start4, const int* pk4) {
    const int rowbytes = d.w * d.c;                         // multiple of 16 (host-checked)
    const int nchunks = rowbytes >> 4;
    const int ck = blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y, f = blockIdx.z;
    if (ck >= nchunks) return;
    const int y0 = start4[g];
    const int* k = pk4 + (int64_t)g * 4 * KU;
    const int nrows = min(4, d.h - 4 * g);
    int acc[4][16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 1 << (PRECISION_BITS - 1);
    const u8* sp = s.row(f, y0) + (ck << 4);
#pragma unroll
    for (int t = 0; t < KU; ++t) {
        const uint4 q = *(const uint4*)(sp + (int64_t)t * s.rs);     // window rows lie inside the image (host-clamped)
        const u32 w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = k[j * KU + t];
            if (c != 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] += mul24((int)((w[e >> 2] >> (8 * (e & 3))) & 0xffu), c);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < nrows) {
            u32 o[4] = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e >> 2] |= (u32)clip8(acc[j][e]) << (8 * (e & 3));
            *(uint4*)(d.row(f, 4 * g + j) + (ck << 4)) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

#include "resample_mfma.inc"

static inline unsigned grid_for(int64_t total) {
    int64_t blocks = (total + 255) / 256;
    return (unsigned)(blocks > 16384 ? 16384 : (blocks < 1 ? 1 : blocks));
}

static int launch_h_fast(const imgxf_lanczos_plan* p, const View& s, const View& d, hipStream_t st) {
    const int rpb = 32;
    dim3 grid((unsigned)((d.w + 1023) / 1024), (unsigned)((d.h + rpb - 1) / rpb), (unsigned)d.n);
    const bool no_lds = knob_set(K_LANCZOS_NO_LDS);
    const bool aligned = ((((uintptr_t)s.p) | (uintptr_t)s.rs | (uintptr_t)s.fs) & 15) == 0 && (s.w * 3) % 16 == 0;
    if (!no_lds && aligned && p->hspan > 0 && p->hspan <= 8192) {
        const int bufbytes = (p->hspan + 15) & ~15;
        const size_t lds = 2 * (size_t)bufbytes;
        switch (p->kpx) {
            case 8: hipLaunchKernelGGL((resample_h_lds_kernel<8>), grid, dim3(256), lds, st, s, d, p->d_start_x, p->d_pk_x, rpb, bufbytes); break;
            case 12: hipLaunchKernelGGL((resample_h_lds_kernel<12>), grid, dim3(256), lds, st, s, d, p->d_start_x, p->d_pk_x, rpb, bufbytes); break;
            default: hipLaunchKernelGGL((resample_h_lds_kernel<16>), grid, dim3(256), lds, st, s, d, p->d_start_x, p->d_pk_x, rpb, bufbytes); break;
        }
        return launch_status();
    }
    switch (p->kpx) {
        case 8: hipLaunchKernelGGL((resample_h_fast_kernel<8>), grid, dim3(256), 0, st, s, d, p->d_start_x, p->d_pk_x, rpb); break;
        case 12: hipLaunchKernelGGL((resample_h_fast_kernel<12>), grid, dim3(256), 0, st, s, d, p->d_start_x, p->d_pk_x, rpb); break;
        default: hipLaunchKernelGGL((resample_h_fast_kernel<16>), grid, dim3(256), 0, st, s, d, p->d_start_x, p->d_pk_x, rpb); break;
    }
    return launch_status();
}

static bool v_fast_ok(const imgxf_lanczos_plan* p, const View& s, const View& d) {
    return p->kpy > 0 && (d.rowbytes() % 16) == 0 &&
           ((((uintptr_t)s.p) | (uintptr_t)s.rs | (uintptr_t)s.fs | ((uintptr_t)d.p) | (uintptr_t)d.rs | (uintptr_t)d.fs) & 15) == 0;
}

static int launch_v_fast(const imgxf_lanczos_plan* p, const View& s, const View& d, hipStream_t st) {
    static const bool no_v4 = knob_set(K_LANCZOS_NO_V4);
    if (p->ku4 == 12 && !no_v4) {
        dim3 grid4((unsigned)(((d.rowbytes() >> 4) + 255) / 256), (unsigned)((d.h + 3) / 4), (unsigned)d.n);
        hipLaunchKernelGGL((resample_v4_kernel<12>), grid4, dim3(256), 0, st, s, d, p->d_start4_y, p->d_pk4_y);
        return launch_status();
    }
    dim3 grid((unsigned)(((d.rowbytes() >> 4) + 255) / 256), (unsigned)d.h, (unsigned)d.n);
    hipLaunchKernelGGL(resample_v_fast_kernel, grid, dim3(256), 0, st, s, d, p->d_start_y, p->d_pk_y, p->kpy);
    return launch_status();
}

// the fused kernel reads and writes dwords at 4-byte aligned addresses and keeps row offsets in 32 bits
static bool rs_mf_ok(const imgxf_lanczos_plan* p, const View& s, const View& d) {
    if (!p->mf_nkh) return false;
    if (((((uintptr_t)s.p) | (uintptr_t)s.rs | (uintptr_t)s.fs | ((uintptr_t)d.p) | (uintptr_t)d.rs | (uintptr_t)d.fs) & 3) != 0) return false;
    // the workgroup count of the launch (the widest setting of IMGXF_RESAMPLE_MFMA_WAVES gives the most) must fit
    // the grid: decided HERE so that the workspace query and the run agree and an oversized call takes the
    // two-pass path with its intermediate
    const int64_t total = (int64_t)((p->mf_ng + 3) / 4) * p->mf_nchunks * s.n;
    if (total > (int64_t(1) << 30)) return false;
    return s.rs > 0 && s.rs < (1 << 24) && p->rh < (1 << 24) && (int64_t)(p->ry0 + p->rh) * s.rs < (int64_t(1) << 31);
}

template <int NW>
static void launch_rs_mf_t(int nkh, dim3 grid, hipStream_t st, const View& s, const View& d, const RsMfArgs& a) {
    const dim3 block(NW * 64);
    const size_t lds = rsmf_lds_bytes(NW);
    switch (nkh) {
        case 1: hipLaunchKernelGGL((resample_mfma_kernel<1, NW>), grid, block, lds, st, s, d, a); break;
        case 2: hipLaunchKernelGGL((resample_mfma_kernel<2, NW>), grid, block, lds, st, s, d, a); break;
        default: hipLaunchKernelGGL((resample_mfma_kernel<3, NW>), grid, block, lds, st, s, d, a); break;
    }
}

static int launch_rs_mf(const imgxf_lanczos_plan* p, const View& s, const View& d, hipStream_t st) {
    const int nw_env = knob_int(K_RESAMPLE_MFMA_WAVES, 4);         // waves per workgroup, 4 or 8 (A/B knob; a wash at 4K)
    const int nw = nw_env == 8 ? 8 : 4;
    RsMfArgs a;
    a.ws = p->d_mf_ws; a.wh = (const rs_v4i*)p->d_mf_wh; a.kcol = p->d_mf_kcol; a.sched = p->d_mf_sched;
    a.chunks = p->d_mf_chunks; a.wv = (const u8*)p->d_mf_wv; a.krow = p->d_mf_krow;
    a.ng = p->mf_ng; a.nsets = (p->mf_ng + nw - 1) / nw; a.nchunks = p->mf_nchunks;
    a.ry0 = p->ry0; a.rh = p->rh; a.out_h = d.h; a.out_rb = (int)d.rowbytes();
    const int64_t total = (int64_t)a.nsets * a.nchunks * s.n;
    if (total > (int64_t(1) << 30)) return IMGXF_ERR_UNSUPPORTED;
    a.total = (unsigned)total;
    const dim3 grid((unsigned)((total + 7) & ~int64_t(7)));
    if (nw == 8) launch_rs_mf_t<8>(p->mf_nkh, grid, st, s, d, a);
    else launch_rs_mf_t<4>(p->mf_nkh, grid, st, s, d, a);
    return launch_status();
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

static int upload_bytes(const std::vector<int8_t>& v, int8_t** dptr) {
    hipError_t e = hipMalloc((void**)dptr, v.size());
    if (e != hipSuccess) return (int)e;
    e = hipMemcpy(*dptr, v.data(), v.size(), hipMemcpyHostToDevice);
    return e == hipSuccess ? IMGXF_OK : (int)e;
}

IMGXF_API int imgxf_lanczos_plan_create(imgxf_lanczos_plan** plan, int in_h, int in_w, int out_h,
                                        int out_w, int c, int max_frames) {
    return imgxf_resample_plan_create(plan, in_h, in_w, out_h, out_w, c, max_frames, IMGXF_RESAMPLE_LANCZOS);
}

IMGXF_API int imgxf_resample_plan_create(imgxf_lanczos_plan** plan, int in_h, int in_w, int out_h,
                                         int out_w, int c, int max_frames, int filter) {
    return imgxf_resample_plan_create_window(plan, in_h, in_w, out_h, out_w, c, max_frames, filter, 0, 0, out_w, out_h);
}

// keep rows [first, first + count) of a (bounds, coefficients) table pair
static void slice_tables(std::vector<int>& b, std::vector<int>& k, int ksize, int first, int count) {
    std::vector<int> b2(b.begin() + 2 * (size_t)first, b.begin() + 2 * (size_t)(first + count));
    std::vector<int> k2(k.begin() + (size_t)first * ksize, k.begin() + (size_t)(first + count) * ksize);
    b.swap(b2); k.swap(k2);
}

IMGXF_API int imgxf_resample_plan_create_window(imgxf_lanczos_plan** plan, int in_h_full, int in_w, int out_h_full,
                                                int out_w_full, int c, int max_frames, int filter,
                                                int wx, int wy, int ww, int wh) {
    if (!plan) return IMGXF_ERR_NULL;
    *plan = nullptr;
    if (in_h_full < 1 || in_w < 1 || out_h_full < 1 || out_w_full < 1 || max_frames < 0) return IMGXF_ERR_ARG;
    if (filter < IMGXF_RESAMPLE_LANCZOS || filter > IMGXF_RESAMPLE_HAMMING) return IMGXF_ERR_ARG;
    if (c != 1 && c != 3 && c != 4) return IMGXF_ERR_UNSUPPORTED;
    if (wx < 0 || wy < 0 || ww < 1 || wh < 1 || wx + ww > out_w_full || wy + wh > out_h_full) return IMGXF_ERR_ARG;
    const bool windowed = wx != 0 || wy != 0 || ww != out_w_full || wh != out_h_full;
    // a window needs both passes (a skipped pass would have to become a crop)
    if (windowed && (out_w_full == in_w || out_h_full == in_h_full)) return IMGXF_ERR_UNSUPPORTED;
    imgxf_lanczos_plan* p = new imgxf_lanczos_plan();
    memset(p, 0, sizeof(*p));
    p->in_h = in_h_full; p->in_w = in_w; p->c = c;
    p->out_h = wh; p->out_w = ww;              // what the resize call writes
    p->wx = wx; p->wy = wy; p->ww = ww; p->wh = wh; p->ry0 = 0; p->rh = in_h_full;
    p->max_frames = max_frames;
    p->need_h = out_w_full != in_w;   // ImagingResample: a pass is skipped when the size is unchanged
    p->need_v = out_h_full != in_h_full;
    int rc = IMGXF_OK;
    // vertical coefficients first: they decide which source rows the horizontal pass must produce
    std::vector<int> by, ky;
    int in_h = in_h_full;                       // rows the vertical pass sees (the H pass output)
    const int out_h = wh, out_w = ww;
    if (p->need_v) {
        p->ksy = build_coeffs(in_h_full, out_h_full, filter, by, ky);
        slice_tables(by, ky, p->ksy, wy, wh);
        if (windowed) {
            int lo = by[0], hi = 0;
            for (int i = 0; i < wh; ++i) { lo = std::min(lo, by[2 * i]); hi = std::max(hi, by[2 * i] + by[2 * i + 1]); }
            p->ry0 = lo; p->rh = hi - lo;
            for (int i = 0; i < wh; ++i) by[2 * i] -= lo;
            in_h = p->rh;
        }
    }
    std::vector<int> bxv, kxv;
    if (p->need_h) {
        std::vector<int>& b = bxv;
        std::vector<int>& k = kxv;
        p->ksx = build_coeffs(in_w, out_w_full, filter, b, k);
        slice_tables(b, k, p->ksx, wx, ww);
        if ((rc = upload(b, &p->d_bounds_x)) == IMGXF_OK) rc = upload(k, &p->d_kk_x);
        const int kp = p->ksx <= 8 ? 8 : (p->ksx <= 12 ? 12 : (p->ksx <= 16 ? 16 : 0));
        if (rc == IMGXF_OK && c == 3 && kp && in_w >= kp) {
            std::vector<int> st, pk;
            build_padded(in_w, out_w, p->ksx, kp, b, k, st, pk);
            p->kpx = kp;
            // source byte span of every group of 1024 output columns (what resample_h_lds_kernel stages);
            // + 4: the window reader fetches one dword past the window for the funnel shift
            int span = 0;
            for (int x0 = 0; x0 < out_w; x0 += 1024) {
                const int x1 = std::min(x0 + 1023, out_w - 1);
                const int base = (st[x0] * 3) & ~15;
                span = std::max(span, (st[x1] + kp) * 3 - base + 4);
            }
            p->hspan = span;
            if ((rc = upload(st, &p->d_start_x)) == IMGXF_OK) rc = upload(pk, &p->d_pk_x);
        }
    }
    if (rc == IMGXF_OK && p->need_v) {
        std::vector<int>& b = by;
        std::vector<int>& k = ky;
        if ((rc = upload(b, &p->d_bounds_y)) == IMGXF_OK) rc = upload(k, &p->d_kk_y);
        const int kp = p->ksy;
        if (rc == IMGXF_OK && in_h >= kp) {
            std::vector<int> st, pk;
            build_padded(in_h, out_h, p->ksy, kp, b, k, st, pk);
            p->kpy = kp;
            if ((rc = upload(st, &p->d_start_y)) == IMGXF_OK) rc = upload(pk, &p->d_pk_y);
            // 4-row groups: union window of KU = 12 rows when every group's windows fit it
            constexpr int KU = 12;
            if (rc == IMGXF_OK && in_h >= KU) {
                const int ng = (out_h + 3) / 4;
                std::vector<int> st4(ng, 0), pk4((size_t)ng * 4 * KU, 0);
                bool fits = true;
                for (int g = 0; g < ng && fits; ++g) {
                    int lo = b[2 * (4 * g)], hi = 0;
                    for (int j = 0; j < 4 && 4 * g + j < out_h; ++j) {
                        lo = std::min(lo, b[2 * (4 * g + j)]);
                        hi = std::max(hi, b[2 * (4 * g + j)] + b[2 * (4 * g + j) + 1]);
                    }
                    if (hi - lo > KU) { fits = false; break; }
                    const int base = std::min(lo, in_h - KU);
                    st4[g] = base;
                    for (int j = 0; j < 4 && 4 * g + j < out_h; ++j) {
                        const int yy = 4 * g + j, xmin = b[2 * yy], cnt = b[2 * yy + 1];
                        for (int x = 0; x < cnt; ++x) pk4[((size_t)g * 4 + j) * KU + (xmin - base) + x] = k[(size_t)yy * p->ksy + x];
                    }
                }
                if (fits) {
                    p->ku4 = KU;
                    if ((rc = upload(st4, &p->d_start4_y)) == IMGXF_OK) rc = upload(pk4, &p->d_pk4_y);
                }
            }
        }
    }
    if (rc == IMGXF_OK && p->need_h && p->need_v) {
        const int oc = knob_int(K_RESAMPLE_MFMA_OC, 16);
        RsMfTables t;
        if (oc > 0 && oc <= RSMF_MAX_OC && build_rs_mf_tables(bxv, kxv, p->ksx, by, ky, p->ksy, in_w, c, in_h, out_w, out_h, oc, t)) {
            if ((rc = upload(t.ws, &p->d_mf_ws)) == IMGXF_OK && (rc = upload(t.kcol, &p->d_mf_kcol)) == IMGXF_OK &&
                (rc = upload(t.sched, &p->d_mf_sched)) == IMGXF_OK && (rc = upload(t.chunks, &p->d_mf_chunks)) == IMGXF_OK &&
                (rc = upload(t.krow, &p->d_mf_krow)) == IMGXF_OK && (rc = upload_bytes(t.wh, &p->d_mf_wh)) == IMGXF_OK &&
                (rc = upload_bytes(t.wv, &p->d_mf_wv)) == IMGXF_OK) {
                p->mf_nkh = t.nkh; p->mf_ng = t.ng; p->mf_nob = t.nob; p->mf_nchunks = t.nchunks;
            }
        }
    }
    if (rc == IMGXF_OK && p->need_h && p->need_v && max_frames > 0) {     // max_frames == 0: workspace-only plan
        hipError_t e = hipMalloc((void**)&p->d_tmp, (size_t)max_frames * p->rh * out_w * c);
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
    if (p->d_start_x) (void)hipFree(p->d_start_x);
    if (p->d_pk_x) (void)hipFree(p->d_pk_x);
    if (p->d_start_y) (void)hipFree(p->d_start_y);
    if (p->d_pk_y) (void)hipFree(p->d_pk_y);
    if (p->d_start4_y) (void)hipFree(p->d_start4_y);
    if (p->d_pk4_y) (void)hipFree(p->d_pk4_y);
    if (p->d_tmp) (void)hipFree(p->d_tmp);
    if (p->d_mf_ws) (void)hipFree(p->d_mf_ws);
    if (p->d_mf_kcol) (void)hipFree(p->d_mf_kcol);
    if (p->d_mf_sched) (void)hipFree(p->d_mf_sched);
    if (p->d_mf_chunks) (void)hipFree(p->d_mf_chunks);
    if (p->d_mf_krow) (void)hipFree(p->d_mf_krow);
    if (p->d_mf_wh) (void)hipFree(p->d_mf_wh);
    if (p->d_mf_wv) (void)hipFree(p->d_mf_wv);
    delete p;
    return IMGXF_OK;
}

// Bytes of the H -> V intermediate for n frames (0 when the plan runs a single pass)
static size_t resample_tmp_bytes(const imgxf_lanczos_plan* p, int n) {
    return (p->need_h && p->need_v) ? (size_t)n * p->rh * p->out_w * p->c : 0;
}

IMGXF_API int imgxf_resample_workspace_bytes(const imgxf_lanczos_plan* p, int n, size_t* bytes) {
    if (!p || !bytes) return IMGXF_ERR_NULL;
    if (n < 0) return IMGXF_ERR_ARG;
    *bytes = resample_tmp_bytes(p, n);
    return IMGXF_OK;
}

static int run_resample(const imgxf_lanczos_plan* p, const imgxf_view* src, const imgxf_view* dst,
                        uint8_t* tmp, void* stream);

IMGXF_API int imgxf_resample_plan_kernel(const imgxf_lanczos_plan* p, int* ksteps) {
    if (!p || !ksteps) return IMGXF_ERR_NULL;
    *ksteps = p->mf_nkh;
    return IMGXF_OK;
}

IMGXF_API int imgxf_resize_lanczos_u8(const imgxf_lanczos_plan* p, const imgxf_view* src,
                                      const imgxf_view* dst, void* stream) {
    if (!p || !src) return IMGXF_ERR_NULL;
    if (src->n > p->max_frames) return IMGXF_ERR_WORKSPACE;
    return run_resample(p, src, dst, p->d_tmp, stream);
}

// does this call run the fused matrix-core kernel (no intermediate)?
static bool resample_runs_fused(const imgxf_lanczos_plan* p, const imgxf_view* src, const imgxf_view* dst) {
    if (!p->need_h || !p->need_v || !src || !dst) return false;
    // the knobs that pick a variant of the two-pass kernels imply the two-pass path
    if (knob_set(K_LANCZOS_SLOW) || knob_set(K_RESAMPLE_NO_MFMA) || knob_set(K_LANCZOS_NO_LDS) || knob_set(K_LANCZOS_NO_V4))
        return false;
    return rs_mf_ok(p, make_view(src), make_view(dst));
}

IMGXF_API int imgxf_resample_workspace_bytes_for(const imgxf_lanczos_plan* p, const imgxf_view* src,
                                                 const imgxf_view* dst, size_t* bytes) {
    if (!p || !src || !dst || !bytes) return IMGXF_ERR_NULL;
    *bytes = resample_runs_fused(p, src, dst) ? 0 : resample_tmp_bytes(p, src->n);
    return IMGXF_OK;
}

// Same resample with the intermediate in a caller-provided, stream-ordered workspace: the plan
// holds only immutable coefficient tables, so one plan serves any number of streams (and any
// batch size) concurrently and nothing is allocated or freed at call time (graph-capture safe).
IMGXF_API int imgxf_resample_ws_u8(const imgxf_lanczos_plan* p, const imgxf_view* src, const imgxf_view* dst,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    if (!p || !src) return IMGXF_ERR_NULL;
    const size_t need = resample_runs_fused(p, src, dst) ? 0 : resample_tmp_bytes(p, src->n);
    if (need && (!workspace || workspace_bytes < need)) return IMGXF_ERR_WORKSPACE;
    return run_resample(p, src, dst, (uint8_t*)workspace, stream);
}

static int run_resample(const imgxf_lanczos_plan* p, const imgxf_view* src, const imgxf_view* dst,
                        uint8_t* tmp, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (src->n != dst->n || src->c != p->c || dst->c != p->c) return IMGXF_ERR_SHAPE;
    if (src->h != p->in_h || src->w != p->in_w || dst->h != p->out_h || dst->w != p->out_w)
        return IMGXF_ERR_SHAPE;
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
    const bool slow = knob_set(K_LANCZOS_SLOW);
    auto run_h = [&](const View& a, const View& b) {
        if (p->kpx && !slow) return launch_h_fast(p, a, b, st);
        return launch_h(a, b, p->d_bounds_x, p->d_kk_x, p->ksx, st);
    };
    if (p->need_h && !p->need_v) return run_h(s, d);
    if (resample_runs_fused(p, src, dst)) return launch_rs_mf(p, s, d, st);
    View mid = s;
    if (p->need_h) {
        View sub = s;                                  // the source rows the window's vertical taps touch
        sub.p = s.p + (int64_t)p->ry0 * s.rs; sub.h = p->rh;
        if (!tmp) return IMGXF_ERR_WORKSPACE;
        mid.p = tmp; mid.n = s.n; mid.h = p->rh; mid.w = p->out_w; mid.c = p->c;
        mid.rs = (int64_t)p->out_w * p->c; mid.fs = mid.rs * p->rh;
        IMGXF_CHECK(run_h(sub, mid));
    }
    if (!slow && v_fast_ok(p, mid, d)) return launch_v_fast(p, mid, d, st);
    const int64_t total = (int64_t)d.n * d.h * d.rowbytes();
    hipLaunchKernelGGL(resample_v_kernel, dim3(grid_for(total)), dim3(256), 0, st, mid, d,
                       p->d_bounds_y, p->d_kk_y, p->ksy);
    return launch_status();
}
