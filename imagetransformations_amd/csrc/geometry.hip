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

typedef u32 u32x4_any __attribute__((ext_vector_type(4), aligned(1)));   // 16 bytes at any address (global loads need no alignment)

// rectangle copy expressed on byte runs: rbytes = rw*c bytes per row.  Chunks are cut on the DESTINATION's
// 16-byte grid (aligned stores), the source side is one unaligned 16-byte load: a paste at a pixel offset
// (3 dx bytes) is almost never aligned on both sides, and the byte-wise path that used to take those ran at
// a third of a copy's speed.
__global__ __launch_bounds__(256) void copy_rect_kernel(View s, View d, int sxb, int sy, int dxb,
                                                        int dy, int rbytes, int rh) {
    const int nchunks = (rbytes + 30) >> 4;                   // a row's run touches at most this many aligned blocks
    const int64_t total = (int64_t)d.n * rh * nchunks;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int ck = (int)(t % nchunks);
        const int64_t r = t / nchunks;
        const int y = (int)(r % rh), f = (int)(r / rh);
        const u8* sp0 = s.row(f, sy + y) + sxb;
        u8* dp0 = d.row(f, dy + y) + dxb;
        const int lead = (int)((uintptr_t)dp0 & 15);          // bytes of the run's first block that precede it
        const int b0 = ck * 16 - lead, b1 = b0 + 16;          // this block = run bytes [b0, b1)
        if (b0 >= 0 && b1 <= rbytes) {
            *(uint4*)(dp0 + b0) = __builtin_bit_cast(uint4, *(const u32x4_any*)(sp0 + b0));
        } else {
            for (int e = max(b0, 0); e < min(b1, rbytes); ++e) dp0[e] = sp0[e];
        }
    }
}

// dst(x, y) = src(x - dx, y - dy) where that exists, else the fill colour: Image.new + crop + paste of
// apply_translation (/root/reference/transformation.py:284-307) in ONE pass over the destination
// (6 instead of 9 bytes per pixel, aligned 16-byte stores, unaligned 16-byte loads).
__global__ __launch_bounds__(256) void translate_kernel(View s, View d, int dxb, int dy, FillPat pat) {
    const int rowbytes = d.w * d.c;
    const int nchunks = (rowbytes + 15) >> 4;
    const int64_t total = (int64_t)d.n * d.h * nchunks;
    const bool al = ((((uintptr_t)d.p) | (uintptr_t)d.rs | (uintptr_t)d.fs) & 15) == 0;
    const int c0 = max(dxb, 0), c1 = min(rowbytes, rowbytes + dxb);      // destination bytes that have a source
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int ck = (int)(t % nchunks);
        const int64_t r = t / nchunks;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const int xb = ck << 4;
        u8* dp = d.row(f, y) + xb;
        const int nv = min(16, rowbytes - xb);
        const int ys = y - dy;
        const bool row_in = ys >= 0 && ys < s.h;
        const int ph = ck % 3;
        const uint4 fv = ph == 0 ? pat.q[0] : (ph == 1 ? pat.q[1] : pat.q[2]);
        if (al && nv == 16 && (!row_in || xb + 16 <= c0 || xb >= c1)) {
            *(uint4*)dp = fv;                                             // entirely fill
        } else if (al && nv == 16 && xb >= c0 && xb + 16 <= c1) {
            *(uint4*)dp = __builtin_bit_cast(uint4, *(const u32x4_any*)(s.row(f, ys) + xb - dxb));
        } else {
            const u32 o[4] = {fv.x, fv.y, fv.z, fv.w};
            const u8* sp = row_in ? s.row(f, ys) - dxb : nullptr;
            for (int e = 0; e < nv; ++e) {
                const int b = xb + e;
                dp[e] = (row_in && b >= c0 && b < c1) ? sp[b] : (u8)(o[e >> 2] >> (8 * (e & 3)));
            }
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

// Quarter turns through a 32 x 32 pixel LDS tile: the source tile is read along its rows and the destination
// tile written along its rows (the per-pixel kernel above strides one of the two sides by a whole image row per
// lane).  Pixels sit in LDS as dwords at pitch 33, so the transposed read is conflict free.
template <int C>
__global__ __launch_bounds__(256) void rot90_tile_kernel(View s, View d, int turns, int ntx, int nty) {
    __shared__ u32 tile[32][33];
    const int bid = blockIdx.x, txb = bid % ntx, tyb = (bid / ntx) % nty, f = bid / (ntx * nty);
    const int x0 = txb * 32, y0 = tyb * 32;                  // destination tile origin
    const int j = threadIdx.x & 31, i0 = threadIdx.x >> 5;
    // source tile: rows sy0 .. sy0+31, columns sx0 .. sx0+31
    const int sy0 = turns == 1 ? x0 : s.h - 1 - x0 - 31;
    const int sx0 = turns == 1 ? s.w - 1 - y0 - 31 : y0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = i0 + 8 * k, sy = sy0 + r, sx = sx0 + j;
        u32 v = 0;
        if ((u32)sy < (u32)s.h && (u32)sx < (u32)s.w) {
            const u8* sp = s.row(f, sy) + sx * C;
#pragma unroll
            for (int c = 0; c < C; ++c) v |= (u32)sp[c] << (8 * c);
        }
        tile[r][j] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int yy = i0 + 8 * k, xx = j;
        const int y = y0 + yy, x = x0 + xx;
        if (y < d.h && x < d.w) {
            const u32 v = turns == 1 ? tile[xx][31 - yy] : tile[31 - xx][yy];
            u8* dp = d.row(f, y) + x * C;
#pragma unroll
            for (int c = 0; c < C; ++c) dp[c] = (u8)(v >> (8 * c));
        }
    }
}

// (Round 3 tried 128 x 16 DESTINATION tiles for packed RGB — whole-line dword-triple stores, one unaligned dword per source
// pixel, the 128-row x 16-pixel source tile parked transposed in LDS: 0.586 ms per 16 4K frames against 0.375 ms for the
// 32 x 32 tiles above.  A wave-level load then touches 64 source rows, and unaligned dwords cost the L1 address path as
// much as the three byte loads they replace; profiles/r03_experiments/leftovers_round3.txt.)
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

// Mirrors on whole dwords.  FLIP_TOP_BOTTOM is a row copy in reverse row order (16-byte chunks cut on the
// destination grid, unaligned loads); FLIP_LEFT_RIGHT of packed RGB moves groups of 4 pixels = 3 dwords:
// destination group g is source group G-1-g with its pixels reversed, four v_perm_b32 per group
// (a = R0G0B0R1  b = G1B1R2G2  c = B2R3G3B3  ->  R3G3B3R2  G2B2R1G1  B1R0G0B0).
typedef u32 u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));

__global__ __launch_bounds__(256) void flip_rows_kernel(View s, View d) {
    const int rowbytes = d.w * d.c;
    const int nchunks = (rowbytes + 30) >> 4;
    const int64_t total = (int64_t)d.n * d.h * nchunks;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int ck = (int)(t % nchunks);
        const int64_t r = t / nchunks;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const u8* sp0 = s.row(f, s.h - 1 - y);
        u8* dp0 = d.row(f, y);
        const int lead = (int)((uintptr_t)dp0 & 15);
        const int b0 = ck * 16 - lead, b1 = b0 + 16;
        if (b0 >= 0 && b1 <= rowbytes) *(uint4*)(dp0 + b0) = __builtin_bit_cast(uint4, *(const u32x4_any*)(sp0 + b0));
        else for (int e = max(b0, 0); e < min(b1, rowbytes); ++e) dp0[e] = sp0[e];
    }
}

// rows: false = same row (FLIP_LEFT_RIGHT), true = reversed rows as well (ROTATE_180); w % 4 == 0, rows 4-byte aligned
__global__ __launch_bounds__(256) void mirror_rgb4_kernel(View s, View d, int rows) {
    const int G = d.w >> 2;
    const int64_t total = (int64_t)d.n * d.h * G;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int g = (int)(t % G);
        const int64_t r = t / G;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const u32x3_a4 v = *(const u32x3_a4*)(s.row(f, rows ? s.h - 1 - y : y) + 12 * (G - 1 - g));
        const u32 a = v.x, b = v.y, c = v.z;
        u32x3_a4 o;
        o.x = __builtin_amdgcn_perm(b, c, 0x06030201u);                       // c1 c2 c3 b2
        o.y = __builtin_amdgcn_perm(b, __builtin_amdgcn_perm(a, c, 0x00070000u), 0x04020107u);   // b3 c0 a3 b0 (via . c0 a3 .)
        o.z = __builtin_amdgcn_perm(b, a, 0x02010005u);                       // b1 a0 a1 a2
        *(u32x3_a4*)(d.row(f, y) + 12 * g) = o;
    }
}

static inline unsigned grid_for(int64_t total) {
    int64_t blocks = (total + 255) / 256;
    return (unsigned)(blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks));
}

static inline bool mirror_rgb4_ok(const View& s, const View& d) {
    return d.c == 3 && (d.w & 3) == 0 &&
           ((((uintptr_t)s.p) | (uintptr_t)s.rs | (uintptr_t)s.fs | ((uintptr_t)d.p) | (uintptr_t)d.rs | (uintptr_t)d.fs) & 3) == 0;
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
    const int64_t total = (int64_t)dst->n * rh * (((int64_t)rw * c + 30) >> 4);
    hipLaunchKernelGGL(copy_rect_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       make_view(src), make_view(dst), sx * c, sy, dx * c, dy, rw * c, rh);
    return launch_status();
}

IMGXF_API int imgxf_translate_u8(const imgxf_view* src, const imgxf_view* dst, int dx, int dy,
                                 const uint8_t* fill, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!fill) return IMGXF_ERR_NULL;
    if (!same_geometry(src, dst)) return IMGXF_ERR_SHAPE;
    if (empty_view(dst)) return IMGXF_OK;
    if (dx <= -src->w || dx >= src->w || dy <= -src->h || dy >= src->h) return imgxf_fill_u8(dst, fill, stream);
    FillPat pat;
    uint8_t bytes[48];
    for (int i = 0; i < 48; ++i) bytes[i] = fill[i % dst->c];
    memcpy(&pat, bytes, sizeof(bytes));
    const View s = make_view(src), d = make_view(dst);
    const int64_t total = (int64_t)d.n * d.h * ((d.rowbytes() + 15) >> 4);
    hipLaunchKernelGGL(translate_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, s, d, dx * src->c, dy, pat);
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
    if (quarter_turns_ccw == 2 && mirror_rgb4_ok(make_view(src), d)) {
        hipLaunchKernelGGL(mirror_rgb4_kernel, dim3(grid_for((int64_t)d.n * d.h * (d.w >> 2))), dim3(256), 0, (hipStream_t)stream,
                           make_view(src), d, 1);
        return launch_status();
    }
    if (quarter_turns_ccw != 2 && d.c <= 4) {
        const int ntx = (d.w + 31) / 32, nty = (d.h + 31) / 32;
        const int64_t nb = (int64_t)ntx * nty * d.n;
        if (nb <= 0x7fffffff) {
            const View sv = make_view(src);
            switch (d.c) {
                case 1: hipLaunchKernelGGL((rot90_tile_kernel<1>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, sv, d, quarter_turns_ccw, ntx, nty); break;
                case 2: hipLaunchKernelGGL((rot90_tile_kernel<2>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, sv, d, quarter_turns_ccw, ntx, nty); break;
                case 3: hipLaunchKernelGGL((rot90_tile_kernel<3>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, sv, d, quarter_turns_ccw, ntx, nty); break;
                default: hipLaunchKernelGGL((rot90_tile_kernel<4>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, sv, d, quarter_turns_ccw, ntx, nty); break;
            }
            return launch_status();
        }
    }
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
    const View d = make_view(dst), s = make_view(src);
    if (mode == 1) {
        hipLaunchKernelGGL(flip_rows_kernel, dim3(grid_for((int64_t)d.n * d.h * ((d.rowbytes() + 30) >> 4))), dim3(256), 0,
                           (hipStream_t)stream, s, d);
        return launch_status();
    }
    if (mirror_rgb4_ok(s, d)) {
        hipLaunchKernelGGL(mirror_rgb4_kernel, dim3(grid_for((int64_t)d.n * d.h * (d.w >> 2))), dim3(256), 0, (hipStream_t)stream, s, d, 0);
        return launch_status();
    }
    hipLaunchKernelGGL(flip_kernel, dim3(grid_for((int64_t)d.n * d.h * d.w)), dim3(256), 0,
                       (hipStream_t)stream, s, d, mode);
    return launch_status();
}
