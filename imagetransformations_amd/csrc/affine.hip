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
#include <algorithm>
#include <string.h>
#include <stdlib.h>

namespace imgxf {

struct AffineParams {
    double m[6];
    int fx[6];       // 16.16 fixed-point matrix for NEAREST (affine_fixed)
    int64_t q0, q3;  // m0, m3 in 2^-40 fixed point (per-pixel x increments of the BILINEAR fast path)
    int64_t q1, q4;  // m1, m4 likewise (per-row increments, LDS-staged path)
    int64_t x00, y00; // xin-.5, yin-.5 of output pixel (0,0) in 2^-40 fixed point (Pillow's fp64 value)
    u32 ntx_magic;    // ceil(2^32 / ntx) of the LDS kernels' tile grid
    int sx[4], sy[4]; // round(k*q0 / 2^16), round(k*q3 / 2^16): 8.24 steps to pixel k of a lane (LDS fast loop)
    u8 fill[4];
    int strip_w;      // NEAREST DMA kernel: > 0 = every XCD owns a vertical strip of this many tile columns
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

// HONLY (BICUBIC): the matrix is a pure horizontal shear/shift (m3 == 0, m4 == 1, m5 integral),
// so yin - 0.5 is an integer, dy == 0 exactly and libImaging's column cubic p1 + 0*(...) returns
// the row-y value bit-for-bit: only row y's horizontal cubic is evaluated (apply_shear).
template <int C, int FILTER, class A, bool HONLY = false>
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
                    for (int t = (HONLY ? 1 : 0); t < (HONLY ? 2 : 4); ++t) {
                        const int yy = yi - 1 + t;
                        const bool inr = (t == 0) || HONLY || (yy >= 0 && yy < s.h);
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
                        v[j] = HONLY ? rowv[1][j] : cubic<A>(rowv[0][j], rowv[1][j], rowv[2][j], rowv[3][j], dy);
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

// ---------------------------------------------------------------------------------------
// Coalesced store of a wave's output.  Every lane holds 4 consecutive pixels = 4*C bytes;
// written straight from the lane that is C dword stores at a 4*C-byte lane stride (each
// store instruction touches every line of the span).  Instead the wave parks its bytes in a
// private LDS strip (conflict-free b32 writes) and reads them back as 16-byte chunks, so one
// dwordx4 store per lane covers the span densely.  `seg_lanes` lanes form one contiguous
// output segment (a tile row); segments need 16-byte aligned global addresses.
// ---------------------------------------------------------------------------------------
template <int C, int SEG_LANES>
__device__ __forceinline__ void staged_store(u32* wlds, const u32 (&o)[C], int lane, u8* row_seg_base) {
    // wlds: this wave's 64*C dwords.  row_seg_base: global address of the first byte of the
    // segment (tile row) this lane belongs to — identical for all SEG_LANES lanes of a segment.
    constexpr int SEG_BYTES = SEG_LANES * 4 * C;            // multiple of 16 (host/tile checked)
    constexpr int CHUNKS = SEG_BYTES / 16;                  // 16-byte chunks per segment
    constexpr int NSEG = 64 / SEG_LANES;
#pragma unroll
    for (int j = 0; j < C; ++j) wlds[lane * C + j] = o[j];
    // lane t re-reads chunk (t % CHUNKS) of segment (t / CHUNKS); segment base pointers are
    // exchanged through a second tiny LDS table (one 8-byte slot per segment)
    u8** segtab = (u8**)(wlds + 64 * C);
    if ((lane % SEG_LANES) == 0) segtab[lane / SEG_LANES] = row_seg_base;
    if (lane < NSEG * CHUNKS) {
        const int sg = lane / CHUNKS, ck = lane - sg * CHUNKS;
        const uint4 v = *(const uint4*)(wlds + sg * (SEG_BYTES / 4) + ck * 4);
        u8* gp = segtab[sg];
        if (gp) *(uint4*)(gp + ck * 16) = v;
    }
}

// ---------------------------------------------------------------------------------------
// BILINEAR fast path (benchmark configs[3]).  Same output-stationary mapping, but
//   - the 2x2 neighbourhood of a pixel is fetched with TWO unaligned 8-byte loads (one per
//     source row: 2 pixels x C bytes each) instead of 4*C byte loads;
//   - interpolation runs in fp32 FMAs on v_cvt_f32_ubyteN outputs; coordinates stay in
//     un-contracted fp64 exactly as libImaging computes them (floor / bounds decisions and
//     dx,dy must agree with Pillow's doubles);
//   - PRECISE: the fp32 value is within GUARD of Pillow's fp64 value (bound derived in
//     DESIGN.md), so (UINT8)v can only differ when the fp32 value sits within GUARD of an
//     integer; exactly those pixels are recomputed with libImaging's fp64 sequence from the
//     bytes already in registers.  Result: bit-identical to Pillow at close to fp32 speed.
// Pixels whose 2x2 support touches the image border (clamped neighbours) take the generic
// code below the fast branch.
// ---------------------------------------------------------------------------------------
typedef uint64_t u64_unaligned __attribute__((aligned(1)));
typedef uint16_t u16_unaligned __attribute__((aligned(1)));
typedef uint32_t u32_unaligned __attribute__((aligned(1)));

template <int C>
__device__ __forceinline__ uint64_t load_pair(const u8* p) {
    if constexpr (C == 1) return (uint64_t)(*(const u16_unaligned*)p);
    else return *(const u64_unaligned*)p;
}
__device__ __forceinline__ float byte_f(uint64_t w, int i) {   // static i -> v_cvt_f32_ubyteN
    return (float)((u32)(w >> (8 * i)) & 0xffu);
}

// Lane -> pixel mapping: a wave owns a TXG*4 x (64/TXG) pixel tile (lanes along x first),
// a workgroup WX x (4/WX) such tiles.  A compact 2-D tile keeps the rotated source
// footprint of one load instruction on a few cache lines (a 256x1 strip touches ~64 lines
// per instruction at 30 degrees, a 32x8 tile ~15).  Workgroups are numbered x-fastest and
// remapped so that each XCD (blockIdx % 8 under round-robin dispatch — speed only) walks a
// contiguous range of tiles and re-uses its own L2 for the footprint overlap.
template <int C, bool PRECISE, int TXG, int WX>
__global__ __launch_bounds__(256) void affine_bilinear_kernel(View s, View d, AffineParams P, View dbg,
                                                              int ntx, int nty, int nblocks) {
    constexpr float GUARD = 1.0e-4f;
    constexpr int TR = 64 / TXG, WY = 4 / WX;
    constexpr int BW = WX * TXG * 4, BH = WY * TR;
    // bijective XCD remap (cdna guide T1): xcd k gets a contiguous range of logical ids
    const int orig = blockIdx.x, xcd = orig & 7, q = nblocks >> 3, r = nblocks & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int per_frame = ntx * nty;
    const int f = logical / per_frame;
    const int rem = logical - f * per_frame;
    const int tyb = rem / ntx, txb = rem - tyb * ntx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = txb * BW + ((wave % WX) * TXG + (lane % TXG)) * 4;
    const int y = tyb * BH + (wave / WX) * TR + lane / TXG;
    // wave-uniform: the whole wave tile is inside the image and its rows start 16-B aligned
    const int wx0 = txb * BW + (wave % WX) * TXG * 4, wy0 = tyb * BH + (wave / WX) * TR;
    const bool staged = (TXG * 4 * C) % 16 == 0 && wx0 + TXG * 4 <= d.w && wy0 + TR <= d.h &&
                        ((((uintptr_t)d.p) | (uintptr_t)d.rs | (uintptr_t)d.fs | (uintptr_t)(wx0 * C)) & 15) == 0;
    if (y >= d.h || x0 >= d.w) return;
    const u8* sp = s.p + (int64_t)f * s.fs;
    const double yc = (double)y + 0.5;
    const double tx = P.m[1] * yc, ty = P.m[4] * yc;    // a1*yin, a4*yin (row constants)

    // libImaging's exact fp64 evaluation of one pixel (coordinates, clamped 2x2 support, lerps):
    // used for the lane's first pixel coordinate, and for every pixel the fast path hands back.
    auto exact_pixel = [&](int x, u8 (&px)[C], float (&vv)[C]) {
        const double xc = (double)x + 0.5;
        double xin = (P.m[0] * xc + tx) + P.m[2];        // a0*xin + a1*yin + a2, un-contracted
        double yin = (P.m[3] * xc + ty) + P.m[5];
        if (!(xin >= 0.0 && xin < (double)s.w && yin >= 0.0 && yin < (double)s.h)) {
#pragma unroll
            for (int j = 0; j < C; ++j) { px[j] = P.fill[j]; vv[j] = (float)P.fill[j]; }
            return;
        }
        xin -= 0.5; yin -= 0.5;
        const double xfl = floor(xin), yfl = floor(yin);
        const int xq = (int)xfl, yq = (int)yfl;
        const double dxd = xin - xfl, dyd = yin - yfl;
        const int xa = clampi(xq, 0, s.w - 1) * C, xb = clampi(xq + 1, 0, s.w - 1) * C;
        const u8* r0 = sp + (int64_t)clampi(yq, 0, s.h - 1) * s.rs;
        const bool has1 = (yq + 1 >= 0) && (yq + 1 < s.h);
        const u8* r1 = sp + (int64_t)(has1 ? yq + 1 : 0) * s.rs;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const double v1 = (double)r0[xa + j] + ((double)r0[xb + j] - (double)r0[xa + j]) * dxd;
            const double v2 = has1 ? (double)r1[xa + j] + ((double)r1[xb + j] - (double)r1[xa + j]) * dxd : v1;
            const double v = v1 + (v2 - v1) * dyd;
            vv[j] = (float)v; px[j] = (u8)(int)v;
        }
    };

    // ---- phase 1: source coordinates in 2^-40 fixed point.  Pixel 0 of the lane is Pillow's
    // own fp64 value; pixels 1..3 add k*m0 (k*m3) as 64-bit integers.  That differs from the
    // fp64 evaluation by < 2^-37, so floor() and the bounds test can only disagree when a
    // coordinate lies within CG = 2^-32 of an integer or half-integer: such pixels (and the
    // ones whose 2x2 support touches the border) are handed to exact_pixel() in phase 3.
    constexpr int64_t ONE = (int64_t)1 << 40, FMASK = ONE - 1, CG = (int64_t)1 << 8;
    auto to_fixed = [&](double v) -> int64_t {          // |v| < 2^21 (host-checked)
        const double fl = floor(v);
        const double fr = (v - fl) * 256.0;              // [0, 256)
        const double frh = floor(fr);
        const u32 lo = (u32)((fr - frh) * 4294967296.0);
        return ((int64_t)(int)fl << 40) + ((int64_t)(u32)(int)frh << 32) + (int64_t)lo;
    };
    const double xc0 = (double)x0 + 0.5;
    const int64_t X0 = to_fixed((P.m[0] * xc0 + tx) + P.m[2]);
    const int64_t Y0 = to_fixed((P.m[3] * xc0 + ty) + P.m[5]);
    bool ok[4], inner[4];
    float dxf[4], dyf[4];
    u32 tl[4], th[4], bl[4], bh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t X = X0 + k * P.q0, Y = Y0 + k * P.q3;
        const int xh = (int)(X >> 40), yh = (int)(Y >> 40);
        ok[k] = xh >= 0 && xh < s.w && yh >= 0 && yh < s.h;     // 0 <= xin < w, 0 <= yin < h
        const int64_t Xs = X - (ONE >> 1), Ys = Y - (ONE >> 1);  // xin - 0.5
        const int xi = (int)(Xs >> 40), yi = (int)(Ys >> 40);
        const int64_t fX = X & FMASK, fXs = Xs & FMASK, fY = Y & FMASK, fYs = Ys & FMASK;
        const bool sure = fX >= CG && fX < ONE - CG && fXs >= CG && fXs < ONE - CG &&
                          fY >= CG && fY < ONE - CG && fYs >= CG && fYs < ONE - CG;
        // xin, yin far outside: the fixed-point value may have wrapped, but then it is not `ok`
        // only if ... keep it simple: out-of-range pixels that are `sure` are final (fill)
        inner[k] = ok[k] && sure && xi >= 0 && xi + 2 < s.w && yi >= 0 && yi + 1 < s.h;
        if (!sure) ok[k] = true;                                // undecided: let exact_pixel decide
        dxf[k] = (float)(u32)(fXs >> 8) * 2.3283064365386963e-10f;   // 2^-32
        dyf[k] = (float)(u32)(fYs >> 8) * 2.3283064365386963e-10f;
        const int64_t off = inner[k] ? (int64_t)yi * s.rs + xi * C : 0;
        const uint64_t t = load_pair<C>(sp + off), b = load_pair<C>(sp + off + s.rs);
        tl[k] = (u32)t; th[k] = (u32)(t >> 32); bl[k] = (u32)b; bh[k] = (u32)(b >> 32);
    }
    // ---- phase 2: fp32 interpolation.  |fp32 - fp64| < GUARD, so (UINT8)v is
    // certain unless v is within GUARD of an integer; those pixels go to phase 3 when PRECISE.
    u8 out[4 * C];
    float vf[4][C];
    bool redo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        redo[k] = ok[k] && !inner[k];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            // static byte positions -> v_cvt_f32_ubyteN
            const float a = (float)((tl[k] >> (8 * j)) & 0xffu);
            const float b = (C + j < 4) ? (float)((tl[k] >> (8 * (C + j))) & 0xffu) : (float)((th[k] >> (8 * (C + j - 4))) & 0xffu);
            const float c = (float)((bl[k] >> (8 * j)) & 0xffu);
            const float e = (C + j < 4) ? (float)((bl[k] >> (8 * (C + j))) & 0xffu) : (float)((bh[k] >> (8 * (C + j - 4))) & 0xffu);
            const float v1 = fmaf(b - a, dxf[k], a), v2 = fmaf(e - c, dxf[k], c);
            const float v = fmaf(v2 - v1, dyf[k], v1);
            vf[k][j] = inner[k] ? v : (float)P.fill[j];
            out[k * C + j] = inner[k] ? (u8)(int)v : P.fill[j];
            if (PRECISE) {
                const float fr = v - floorf(v);
                redo[k] |= inner[k] && (fr < GUARD || fr > 1.0f - GUARD) && !(a == b && c == e && a == c);
            }
        }
    }
    // ---- phase 3: exact fp64 evaluation of the handed-back pixels (rare)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (redo[k]) {
            u8 px[C]; float vv[C];
            exact_pixel(x0 + k, px, vv);
#pragma unroll
            for (int j = 0; j < C; ++j) { out[k * C + j] = px[j]; vf[k][j] = vv[j]; }
        }
    }
    if (dbg.p) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (x0 + k < d.w) {
                float* fp = (float*)dbg.row(f, y) + (x0 + k) * C;
#pragma unroll
                for (int j = 0; j < C; ++j) fp[j] = vf[k][j];
            }
        }
    }
    // ---- phase 4: store.  Full tile rows on 16-byte aligned addresses go through the LDS
    // transpose (one dense dwordx4 store per lane); ragged right edges / unaligned views
    // fall back to per-lane dword or byte stores.
    u32 od[C];
#pragma unroll
    for (int q = 0; q < C; ++q)
        od[q] = (u32)out[4 * q] | ((u32)out[4 * q + 1] << 8) | ((u32)out[4 * q + 2] << 16) | ((u32)out[4 * q + 3] << 24);
    u8* dp = d.row(f, y) + x0 * C;
    if (staged) {
        __shared__ __attribute__((aligned(16))) u32 stage[4][64 * C + 2 * (64 / TXG) + 4];
        u8* seg = d.row(f, y) + (x0 - (lane % TXG) * 4) * C;
        staged_store<C, TXG>(stage[wave], od, lane, seg);
        return;
    }
    const int npx = min(4, d.w - x0);
    if (npx == 4 && ((((uintptr_t)dp) & 3) == 0)) {
#pragma unroll
        for (int q = 0; q < C; ++q) ((u32*)dp)[q] = od[q];
    } else {
        for (int e = 0; e < npx * C; ++e) {
            u8 v = 0;
#pragma unroll
            for (int kk = 0; kk < 4 * C; ++kk) if (kk == e) v = out[kk];
            dp[e] = v;
        }
    }
}

// ---- stage the source bounding box of a tile into LDS (RGBX dword per pixel, row pitch PITCH):
// wave w copies rows w, w+4, ...; lanes walk consecutive pixels of a source row (unaligned 4-byte
// loads at a 3-byte lane stride: two cache lines per instruction).  All rows of a wave are in
// flight at once (one memory latency per batch of NB rows); row pointers are scalar, the lane
// contributes a 32-bit byte offset; rows past the box re-read its last row and are not written.
// The frame's very last pixel cannot be read with a 4-byte load, so the one tile that owns it
// reads that pixel from 1 byte earlier and shifts.
template <int PITCH, int NB>
__device__ __forceinline__ void stage_bbox(const View& s, const u8* sp, u32* srct, int lane, int wave,
                                           int sx_lo, int sy_lo, int sx_hi, int sy_hi, int bwc, int bhc) {
    const bool owns_last = sy_hi == s.h - 1 && sx_hi == s.w - 1;           // block-uniform
    const int64_t gstep = 4 * s.rs;
    for (int cc = lane; cc < bwc; cc += 64) {
        u32* lp = srct + wave * PITCH + cc;
        if (!owns_last) {
            const u8* rowp = sp + (int64_t)(sy_lo + wave) * s.rs;           // scalar (wave-uniform)
            const u32 loff = (u32)(sx_lo + cc) * 3u;
            for (int r0 = wave; r0 < bhc; r0 += 4 * NB) {
                u32 v[NB];
                const int last = (bhc - 1 - r0) >> 2;                       // index of the wave's last row in this batch
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    v[i] = *(const u32_unaligned*)(rowp + (int64_t)min(i, last) * gstep + loff);
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    if (i <= last) lp[i * 4 * PITCH] = v[i];
                rowp += NB * gstep;
                lp += NB * 4 * PITCH;
            }
        } else {
            const u8* gp = sp + (int64_t)(sy_lo + wave) * s.rs + (sx_lo + cc) * 3;
            const u8* frame_last4 = sp + (int64_t)(s.h - 1) * s.rs + (int64_t)s.w * 3 - 4;
            for (int rr = wave; rr < bhc; rr += 4) {
                const bool tail = gp > frame_last4;
                const u32 v = *(const u32_unaligned*)(tail ? gp - 1 : gp);
                *lp = tail ? v >> 8 : v;
                gp += gstep;
                lp += 4 * PITCH;
            }
        }
    }
}

#ifndef IMGXF_NEAREST_NB
#define IMGXF_NEAREST_NB 13
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
// byte B of a packed RGBX dword as float; asm so the compiler cannot turn (float)b - (float)a
// into integer byte extraction + subtract + convert (3 extra VALU ops per tap pair)
template <int B>
__device__ __forceinline__ float ubyte_f(u32 w) {
    float r;
    if (B == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(r) : "v"(w));
    else if (B == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(w));
    else if (B == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(r) : "v"(w));
    else asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(r) : "v"(w));
    return r;
}
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// ---------------------------------------------------------------------------------------
// BILINEAR, LDS-staged source tile (RGB).  The three gather kernels above all cost ~79 cycles
// per 64-lane load instruction on the texture-address path (PMC: every lane that touches a
// different cache line is serialised), independent of their VALU work.  Here a workgroup
// (32x32 output pixels) first copies the bounding box of its rotated source footprint from
// global memory into LDS with coalesced row-wise loads (lanes walk consecutive pixels of a
// source row), expanded to one dword per pixel (RGBX), and then gathers its 2x2 supports
// from LDS with one ds_read2_b32 per source row.  Coordinates, guard logic and the exact
// fp64 hand-back are the same as in affine_bilinear_kernel.
// ---------------------------------------------------------------------------------------
// One output tile of the LDS-staged bilinear kernels.  BH = 32: the square tile (smallest rotated
// bounding box), interior fast path or general per-pixel path.  BH = 64 (INTERIOR): interior
// tiles only, a lane computes 8 pixels in two pieces — halves the per-tile setup / staging
// bookkeeping per pixel and stages 1.03 instead of 1.67 source pixels per output pixel at
// 30 deg / 1.5x; returns false BEFORE touching LDS when the tile is not interior, and the caller
// then runs its two 32x32 halves through the BH = 32 body.
template <bool PRECISE, int PITCH, bool DBG, int BH, bool INTERIOR>
__device__ __forceinline__ bool bilinear_tile(const View& s, const View& d, const AffineParams& P, const View& dbg,
                                              u32* srct, u32 (*stage)[64 * 3 + 2 * (64 / 8) + 4],
                                              int f, int txb, int tyb, int lane, int wave) {
    constexpr int C = 3, TXG = 8, WX = 1, TR = 8, BW = 32, NH = BH / 32;
    constexpr float GUARD = 1.2e-4f;
    constexpr int FONE = 1 << 24, FHALF = 1 << 23, CG32 = 1 << 6;   // 8.24 tile-relative coordinates
    const int lx = ((wave % WX) * TXG + (lane % TXG)) * 4;
    int ly = (wave / WX) * TR * NH + lane / TXG;                  // MODE 1 revisits ly / y per half
    const int x0 = txb * BW + lx;
    int y = tyb * BH + ly;
    const int wx0 = txb * BW + (wave % WX) * TXG * 4, wy0 = tyb * BH + (wave / WX) * TR;
    const bool staged = wx0 + TXG * 4 <= d.w && wy0 + TR <= d.h &&
                        ((((uintptr_t)d.p) | (uintptr_t)d.rs | (uintptr_t)d.fs | (uintptr_t)(wx0 * C)) & 15) == 0;
    const bool valid = y < d.h && x0 < d.w;
    const u8* sp = s.p + (int64_t)f * s.fs;

    // ---- tile-origin source coordinate (xin-.5, yin-.5) in 2^-40 fixed point: the host converts
    // Pillow's fp64 value for output pixel (0,0) once; tiles add integer multiples of the matrix
    // (scalar 64-bit arithmetic, error < 2^-27 px over a 32k image, far inside the 2^-18 guard)
    const int64_t XT = P.x00 + (int64_t)(txb * BW) * P.q0 + (int64_t)(tyb * BH) * P.q1;
    const int64_t YT = P.y00 + (int64_t)(txb * BW) * P.q3 + (int64_t)(tyb * BH) * P.q4;
    // bounding box of floor(xin-.5) .. +1 over the tile: the map is affine, extremes are corners
    const int64_t ax = (BW - 1) * P.q0, bx = (BH - 1) * P.q1, ay = (BW - 1) * P.q3, by = (BH - 1) * P.q4;
    const int64_t xlo = XT + min(ax, (int64_t)0) + min(bx, (int64_t)0), xhi = XT + max(ax, (int64_t)0) + max(bx, (int64_t)0);
    const int64_t ylo = YT + min(ay, (int64_t)0) + min(by, (int64_t)0), yhi = YT + max(ay, (int64_t)0) + max(by, (int64_t)0);
    const int ux_lo = (int)(xlo >> 40), uy_lo = (int)(ylo >> 40);              // unclipped origin
    const int sx_lo = max(ux_lo, 0), sx_hi = min((int)(xhi >> 40) + 1, s.w - 1);
    const int sy_lo = max(uy_lo, 0), sy_hi = min((int)(yhi >> 40) + 1, s.h - 1);
    const int bwc = sx_hi - sx_lo + 1, bhc = sy_hi - sy_lo + 1;    // <= 0: tile sees no source pixel
    // interior tile (block-uniform): inside the output, and the 2x2 support of every pixel inside the source
    const bool clean = ux_lo >= 0 && uy_lo >= 0 && (int)(xhi >> 40) + 1 <= s.w - 1 && (int)(yhi >> 40) + 1 <= s.h - 1 &&
                       txb * BW + BW <= d.w && tyb * BH + BH <= d.h;
    // (INTERIOR: the one tile whose box holds the frame's last pixel is left to the list kernel too — same test as the host's)
    if (INTERIOR && (!clean || ((int)(xhi >> 40) + 1 == s.w - 1 && (int)(yhi >> 40) + 1 == s.h - 1))) return false;

    // ---- stage the bounding box: wave w copies rows w, w+4, ...; lanes walk consecutive pixels
    // of a source row (unaligned 4-byte loads at a 3-byte lane stride: two cache lines per
    // instruction).  The frame's very last pixel cannot be read with a 4-byte load, so the one
    // tile that owns it reads that pixel from 1 byte earlier and shifts.
    stage_bbox<PITCH, BH == 128 ? 21 : (BH == 64 ? 13 : 9)>(s, sp, srct, lane, wave, sx_lo, sy_lo, sx_hi, sy_hi, bwc, bhc);
    __syncthreads();

    auto exact_pixel = [&](int x, u8 (&px)[C], float (&vv)[C]) {
        const double xc = (double)x + 0.5, yc = (double)y + 0.5;
        double xin = (P.m[0] * xc + P.m[1] * yc) + P.m[2];
        double yin = (P.m[3] * xc + P.m[4] * yc) + P.m[5];
        if (!(xin >= 0.0 && xin < (double)s.w && yin >= 0.0 && yin < (double)s.h)) {
#pragma unroll
            for (int j = 0; j < C; ++j) { px[j] = P.fill[j]; vv[j] = (float)P.fill[j]; }
            return;
        }
        xin -= 0.5; yin -= 0.5;
        const double xfl = floor(xin), yfl = floor(yin);
        const int xq = (int)xfl, yq = (int)yfl;
        const double dxd = xin - xfl, dyd = yin - yfl;
        const int xa = clampi(xq, 0, s.w - 1) * C, xb = clampi(xq + 1, 0, s.w - 1) * C;
        const u8* r0 = sp + (int64_t)clampi(yq, 0, s.h - 1) * s.rs;
        const bool has1 = (yq + 1 >= 0) && (yq + 1 < s.h);
        const u8* r1 = sp + (int64_t)(has1 ? yq + 1 : 0) * s.rs;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const double v1 = (double)r0[xa + j] + ((double)r0[xb + j] - (double)r0[xa + j]) * dxd;
            const double v2 = has1 ? (double)r1[xa + j] + ((double)r1[xb + j] - (double)r1[xa + j]) * dxd : v1;
            const double v = v1 + (v2 - v1) * dyd;
            vv[j] = (float)v; px[j] = (u8)(int)v;
        }
    };

    // ---- per-pixel coordinates: 64-bit adds, then 8.24 fixed point relative to the UNCLIPPED
    // bbox origin (|rel| < 128 px, host-checked).  The integer part indexes the LDS tile, the
    // fraction is dx / dy.  A coordinate within 2^-18 of an integer or half-integer (where the
    // truncation to 24 fraction bits could change floor() or the bounds test) hands the pixel
    // to exact_pixel(); so does a 2x2 support that touches the image border.
    const int64_t XL = XT + lx * P.q0 + ly * P.q1 - ((int64_t)ux_lo << 40);
    const int64_t YL = YT + lx * P.q3 + ly * P.q4 - ((int64_t)uy_lo << 40);
    const int offx = ux_lo - sx_lo, offy = uy_lo - sy_lo;          // <= 0 where the bbox was clipped

    // ---- interior tiles (block-uniform): the tile lies inside the output and the 2x2 support of
    // every pixel inside the source, so there is no bounds test and no fill.  Straight-line code
    // for the lane's 4 pixels (the compiler interleaves their LDS reads); pixel k's coordinate is
    // pixel 0's rounded 8.24 value plus the host-rounded step sx[k] (|error| <= 2^-24 px in total,
    // so |fp32 - fp64| <= 2*255*6e-8 + 3*7.6e-6 = 5.3e-5 < GUARD).  Pixels that need libImaging's
    // fp64 sequence are collected in a lane mask and handled after the loop.
    if (clean) {
      const bool dst16 = ((((uintptr_t)d.p) | (uintptr_t)d.rs | (uintptr_t)d.fs | (uintptr_t)(wx0 * C)) & 15) == 0;
#pragma unroll
      for (int half = 0; half < NH; ++half) {
        if (NH > 1) { ly = (wave * NH + half) * TR + lane / TXG; y = tyb * BH + ly; }
        const int64_t XLh = NH > 1 ? XT + lx * P.q0 + ly * P.q1 - ((int64_t)ux_lo << 40) : XL;
        const int64_t YLh = NH > 1 ? YT + lx * P.q3 + ly * P.q4 - ((int64_t)uy_lo << 40) : YL;
        const bool staged_h = NH > 1 ? dst16 : staged;
        const int Xb = (int)((XLh + 32768) >> 16), Yb = (int)((YLh + 32768) >> 16);   // rounded: |error| <= 2^-25
        u32 od[C] = {0u, 0u, 0u};
        u32 need = 0;                                    // bit k: pixel k is handed to the slow path
        float vf[DBG ? 4 : 1][C];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int X32 = Xb + P.sx[k], Y32 = Yb + P.sy[k];
            const u32 fx = (u32)X32 & (FONE - 1), fy = (u32)Y32 & (FONE - 1);
            const f32x2 dd = f32x2{(float)fx, (float)fy} * f32x2{5.9604644775390625e-08f, 5.9604644775390625e-08f};   // 2^-24
            const float dxf = dd.x, dyf = dd.y;
            const int li = (Y32 >> 24) * PITCH + (X32 >> 24);
            const u32 p00 = srct[li], p01 = srct[li + 1], p10 = srct[li + PITCH], p11 = srct[li + PITCH + 1];
            const f32x2 dx2 = {dxf, dxf}, dy2 = {dyf, dyf};
            // R,G as a pair through both lerps; B's two rows as a pair through the first lerp
            const f32x2 a = {ubyte_f<0>(p00), ubyte_f<1>(p00)}, b = {ubyte_f<0>(p01), ubyte_f<1>(p01)};
            const f32x2 c = {ubyte_f<0>(p10), ubyte_f<1>(p10)}, e = {ubyte_f<0>(p11), ubyte_f<1>(p11)};
            const f32x2 v1 = fma2(b - a, dx2, a), v2 = fma2(e - c, dx2, c);
            const f32x2 vrg = fma2(v2 - v1, dy2, v1);
            const f32x2 ab = {ubyte_f<2>(p00), ubyte_f<2>(p10)}, bb = {ubyte_f<2>(p01), ubyte_f<2>(p11)};
            const f32x2 vb12 = fma2(bb - ab, dx2, ab);
            const float vb = fmaf(vb12.y - vb12.x, dyf, vb12.x);
            // v >= 0: (UINT8)v == floor(v); |frac(v) - .5| close to .5 <=> v close to an integer
            const f32x2 flrg = {floorf(vrg.x), floorf(vrg.y)};
            const float flb = floorf(vb);
            od[(k * C + 0) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(flrg.x, (k * C + 0) & 3, od[(k * C + 0) >> 2]);
            od[(k * C + 1) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(flrg.y, (k * C + 1) & 3, od[(k * C + 1) >> 2]);
            od[(k * C + 2) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(flb, (k * C + 2) & 3, od[(k * C + 2) >> 2]);
            // one float test flags both kinds of hand-back: a coordinate whose fraction is within
            // GUARD of 0 or 1 (floor() of the 8.24 value could differ from libImaging's; interior
            // tiles have no bounds test, so half-integers do not matter) and, when PRECISE, a value
            // within GUARD of an integer.  The slow path below sorts out which it was.
            const f32x2 cd = dd - f32x2{0.5f, 0.5f};
            float dist = fmaxf(fabsf(cd.x), fabsf(cd.y));
            if (PRECISE) {
                const f32x2 urg = (vrg - flrg) - f32x2{0.5f, 0.5f};
                const float ub = (vb - flb) - 0.5f;
                dist = fmaxf(dist, fmaxf(fmaxf(fabsf(urg.x), fabsf(urg.y)), fabsf(ub)));
            }
            if (DBG) { vf[k][0] = vrg.x; vf[k][1] = vrg.y; vf[k][2] = vb; }
            need |= dist > 0.5f - GUARD ? 1u << k : 0u;
        }
        if (need) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if ((need >> k) & 1u) {
                    u8 px[C]; float vv[C];
                    bool have = true;
                    const int X32 = Xb + P.sx[k], Y32 = Yb + P.sy[k];
                    const u32 gx = (((u32)X32 + CG32) & (FONE - 1)), gy = (((u32)Y32 + CG32) & (FONE - 1));
                    if (min(gx, gy) < 2u * CG32) {
                        // the 8.24 coordinate is within 2^-18 of an integer: its floor is not certain
                        exact_pixel(x0 + k, px, vv);
                    } else if (!PRECISE) {
                        have = false;                    // certain support, fp32 value stands
                    } else {
                        // the support is certain: only dx, dy and the lerps need libImaging's fp64
                        // sequence, on the taps re-read from LDS.  A flat support is exact in fp32.
                        const int li = (Y32 >> 24) * PITCH + (X32 >> 24);
                        const u32 p00 = srct[li], p01 = srct[li + 1], p10 = srct[li + PITCH], p11 = srct[li + PITCH + 1];
                        have = (((p00 ^ p01) | (p10 ^ p11) | (p00 ^ p10)) << 8) != 0;
                        if (have) {
                            const double xc = (double)(x0 + k) + 0.5, yc = (double)y + 0.5;
                            const double xin = ((P.m[0] * xc + P.m[1] * yc) + P.m[2]) - 0.5;
                            const double yin = ((P.m[3] * xc + P.m[4] * yc) + P.m[5]) - 0.5;
                            const double dxd = xin - floor(xin), dyd = yin - floor(yin);
#pragma unroll
                            for (int j = 0; j < C; ++j) {
                                const double ta = (double)((p00 >> (8 * j)) & 0xffu), tb = (double)((p01 >> (8 * j)) & 0xffu);
                                const double tc = (double)((p10 >> (8 * j)) & 0xffu), te = (double)((p11 >> (8 * j)) & 0xffu);
                                const double w1 = ta + (tb - ta) * dxd, w2 = tc + (te - tc) * dxd;
                                const double w = w1 + (w2 - w1) * dyd;
                                vv[j] = (float)w; px[j] = (u8)(int)w;
                            }
                        }
                    }
                    if (have) {
#pragma unroll
                        for (int j = 0; j < C; ++j) {
                            const int q = (k * C + j) >> 2, sh = ((k * C + j) & 3) * 8;
                            od[q] = (od[q] & ~(0xffu << sh)) | ((u32)px[j] << sh);
                            if (DBG) vf[k][j] = vv[j];
                        }
                    }
                }
            }
        }
        if (DBG) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float* fp = (float*)dbg.row(f, y) + (x0 + k) * C;
#pragma unroll
                for (int j = 0; j < C; ++j) fp[j] = vf[k][j];
            }
        }
        if (staged_h) {
            u8* seg = d.row(f, y) + (x0 - (lane % TXG) * 4) * C;
            staged_store<C, TXG>(stage[wave], od, lane, seg);
        } else {
            u8* dp = d.row(f, y) + x0 * C;
            if ((((uintptr_t)dp) & 3) == 0) {
#pragma unroll
                for (int qq = 0; qq < C; ++qq) ((u32*)dp)[qq] = od[qq];
            } else {
#pragma unroll
                for (int e = 0; e < 4 * C; ++e) dp[e] = (u8)(od[e >> 2] >> ((e & 3) * 8));
            }
        }
      }   // half
      return true;
    }
    if (INTERIOR) return true;      // (unreachable: non-interior tiles left before staging)
    u8 out[4 * C];
    float vf[4][C];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int X32 = (int)((XL + k * P.q0) >> 16), Y32 = (int)((YL + k * P.q3) >> 16);
        const int xi = (X32 >> 24) + offx, yi = (Y32 >> 24) + offy;          // relative to the clipped bbox
        const u32 fx = (u32)X32 & (FONE - 1), fy = (u32)Y32 & (FONE - 1);
        const bool sure = ((fx & (FHALF - 1)) - CG32) < (u32)(FHALF - 2 * CG32) &&
                          ((fy & (FHALF - 1)) - CG32) < (u32)(FHALF - 2 * CG32);
        const bool inner = valid && sure && bwc > 1 && bhc > 1 && (u32)xi < (u32)(bwc - 1) && (u32)yi < (u32)(bhc - 1);
        const float dxf = (float)fx * 5.9604644775390625e-08f;               // 2^-24
        const float dyf = (float)fy * 5.9604644775390625e-08f;
        const int li = inner ? yi * PITCH + xi : 0;
        const u32 p00 = srct[li], p01 = srct[li + 1], p10 = srct[li + PITCH], p11 = srct[li + PITCH + 1];
        bool near_int = false;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const float a = (float)((p00 >> (8 * j)) & 0xffu), b = (float)((p01 >> (8 * j)) & 0xffu);
            const float c = (float)((p10 >> (8 * j)) & 0xffu), e = (float)((p11 >> (8 * j)) & 0xffu);
            const float v1 = fmaf(b - a, dxf, a), v2 = fmaf(e - c, dxf, c);
            const float v = fmaf(v2 - v1, dyf, v1);
            const float fl = floorf(v);                                      // v >= 0: (UINT8)v == floor(v)
            vf[k][j] = inner ? v : (float)P.fill[j];
            out[k * C + j] = inner ? (u8)(int)fl : P.fill[j];
            if (PRECISE) near_int |= fabsf((v - fl) - 0.5f) > 0.5f - GUARD;
        }
        if (inner) {
            if (PRECISE) {
                // a flat 2x2 support is exact in both arithmetics: not worth the fp64 redo
                const bool flat = (((p00 ^ p01) | (p10 ^ p11) | (p00 ^ p10)) << 8) == 0;
                if (near_int && !flat) {
                    // the support is certain (`sure`): only dx, dy and the lerps need libImaging's
                    // fp64 sequence, on the taps already in registers
                    const double xc = (double)(x0 + k) + 0.5, yc = (double)y + 0.5;
                    const double xin = ((P.m[0] * xc + P.m[1] * yc) + P.m[2]) - 0.5;
                    const double yin = ((P.m[3] * xc + P.m[4] * yc) + P.m[5]) - 0.5;
                    const double dxd = xin - floor(xin), dyd = yin - floor(yin);
#pragma unroll
                    for (int j = 0; j < C; ++j) {
                        const double a = (double)((p00 >> (8 * j)) & 0xffu), b = (double)((p01 >> (8 * j)) & 0xffu);
                        const double c = (double)((p10 >> (8 * j)) & 0xffu), e = (double)((p11 >> (8 * j)) & 0xffu);
                        const double v1 = a + (b - a) * dxd, v2 = c + (e - c) * dxd;
                        const double v = v1 + (v2 - v1) * dyd;
                        vf[k][j] = (float)v; out[k * C + j] = (u8)(int)v;
                    }
                }
            }
        } else if (valid && x0 + k < d.w) {
            // far outside the image: plain fill; anything else that is not `inner` goes to fp64
            const int xa = xi + sx_lo, ya = yi + sy_lo;
            const bool far_out = sure && (xa < -1 || xa > s.w || ya < -1 || ya > s.h);
            if (!far_out) {
                u8 px[C]; float vv[C];
                exact_pixel(x0 + k, px, vv);
#pragma unroll
                for (int j = 0; j < C; ++j) { out[k * C + j] = px[j]; vf[k][j] = vv[j]; }
            }
        }
    }
    if (dbg.p && valid) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (x0 + k < d.w) {
                float* fp = (float*)dbg.row(f, y) + (x0 + k) * C;
#pragma unroll
                for (int j = 0; j < C; ++j) fp[j] = vf[k][j];
            }
        }
    }
    u32 od[C];
#pragma unroll
    for (int qq = 0; qq < C; ++qq)
        od[qq] = (u32)out[4 * qq] | ((u32)out[4 * qq + 1] << 8) | ((u32)out[4 * qq + 2] << 16) | ((u32)out[4 * qq + 3] << 24);
    if (staged) {
        u8* seg = d.row(f, y) + (x0 - (lane % TXG) * 4) * C;
        staged_store<C, TXG>(stage[wave], od, lane, seg);
        return true;
    }
    if (!valid) return true;
    u8* dp = d.row(f, y) + x0 * C;
    const int npx = min(4, d.w - x0);
    if (npx == 4 && ((((uintptr_t)dp) & 3) == 0)) {
#pragma unroll
        for (int qq = 0; qq < C; ++qq) ((u32*)dp)[qq] = od[qq];
    } else {
        for (int e = 0; e < npx * C; ++e) {
            u8 v = 0;
#pragma unroll
            for (int kk = 0; kk < 4 * C; ++kk) if (kk == e) v = out[kk];
            dp[e] = v;
        }
    }
    return true;
}

template <bool PRECISE, int PITCH, bool DBG>
__global__ __launch_bounds__(256) void affine_bilinear_lds_kernel(View s, View d, AffineParams P, View dbg,
                                                                  int ntx, int nty) {
    extern __shared__ __attribute__((aligned(16))) u32 srct[];      // bbox pixels, RGBX, row pitch PITCH
    __shared__ __attribute__((aligned(16))) u32 stage[4][64 * 3 + 2 * (64 / 8) + 4];
    // tiles of one frame are numbered x-fastest and remapped so each XCD walks a contiguous range
    const int nblocks = ntx * nty;
    const int orig = blockIdx.x, xcd = orig & 7, q = nblocks >> 3, r = nblocks & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int f = blockIdx.y;
    // logical / ntx by the host's reciprocal (exact: logical * ntx < 2^32) keeps the tile setup scalar
    const int tyb = P.ntx_magic ? (int)__umulhi((u32)logical, P.ntx_magic) : logical, txb = logical - tyb * ntx;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bilinear_tile<PRECISE, PITCH, DBG, 32, false>(s, d, P, dbg, srct, stage, f, txb, tyb, lane, wave);
}

// Interior 32x64 tiles (8 pixels per lane); tiles that are not interior leave at once and are
// done by affine_bilinear_lds_list_kernel from the host's list (nty counts 64-row bands).
template <bool PRECISE, int PITCH, bool DBG, int BHT>
__global__ __launch_bounds__(256) void affine_bilinear_lds_interior_kernel(View s, View d, AffineParams P, View dbg,
                                                                           int ntx, int nty) {
    extern __shared__ __attribute__((aligned(16))) u32 srct[];
    __shared__ __attribute__((aligned(16))) u32 stage[4][64 * 3 + 2 * (64 / 8) + 4];
    const int nblocks = ntx * nty;
    const int orig = blockIdx.x, xcd = orig & 7, q = nblocks >> 3, r = nblocks & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int f = blockIdx.y;
    const int tyb = P.ntx_magic ? (int)__umulhi((u32)logical, P.ntx_magic) : logical, txb = logical - tyb * ntx;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bilinear_tile<PRECISE, PITCH, DBG, BHT, true>(s, d, P, dbg, srct, stage, f, txb, tyb, lane, wave);
}

// The 32x64 tiles the interior pass skipped, as 32x32 tiles: workgroup 2e + c is half c of list entry e
constexpr int BILINEAR_LIST_MAX = 896;
struct TileList { int n; u32 idx[BILINEAR_LIST_MAX]; };

template <bool PRECISE, int PITCH, bool DBG, int BHT>
__global__ __launch_bounds__(256) void affine_bilinear_lds_list_kernel(View s, View d, AffineParams P, View dbg,
                                                                       int ntx, TileList list) {
    constexpr int NC = BHT / 32;                     // 32x32 children per listed tile
    extern __shared__ __attribute__((aligned(16))) u32 srct[];
    __shared__ __attribute__((aligned(16))) u32 stage[4][64 * 3 + 2 * (64 / 8) + 4];
    const int parent = (int)list.idx[blockIdx.x / NC];
    const int f = blockIdx.y;
    const int ty64 = P.ntx_magic ? (int)__umulhi((u32)parent, P.ntx_magic) : parent, txb = parent - ty64 * ntx;
    const int tyb = NC * ty64 + (int)(blockIdx.x % NC);
    if (tyb * 32 >= d.h) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bilinear_tile<PRECISE, PITCH, DBG, 32, false>(s, d, P, dbg, srct, stage, f, txb, tyb, lane, wave);
}

// ---------------------------------------------------------------------------------------
// NEAREST (Image.rotate, /root/reference/transformation.py:200), LDS-staged like the bilinear
// kernel: libImaging's affine_fixed integers (16.16, C-int wrap) give the source pixel of
// every output pixel exactly, so there is no guard and no fp64 at all — stage the bounding
// box of the tile's source pixels, read one RGBX dword per output pixel from LDS.
// ---------------------------------------------------------------------------------------
template <int PITCH>
__global__ __launch_bounds__(256) void affine_nearest_lds_kernel(View s, View d, AffineParams P, int ntx, int nty) {
    constexpr int C = 3, TXG = 8, WX = 1, TR = 8, BW = 32, BH = 32;   // square tile: smallest rotated bbox
    extern __shared__ __attribute__((aligned(16))) u32 srct[];
    __shared__ __attribute__((aligned(16))) u32 stage[4][64 * C + 2 * (64 / TXG) + 4];
    const int nblocks = ntx * nty;
    const int orig = blockIdx.x, xcd = orig & 7, q = nblocks >> 3, r = nblocks & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int f = blockIdx.y;
    // logical / ntx by the host's reciprocal (exact: logical * ntx < 2^32) keeps the tile setup scalar
    const int tyb = P.ntx_magic ? (int)__umulhi((u32)logical, P.ntx_magic) : logical, txb = logical - tyb * ntx;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lx = ((wave % WX) * TXG + (lane % TXG)) * 4, ly = (wave / WX) * TR + lane / TXG;
    const int x0 = txb * BW + lx, y = tyb * BH + ly;
    const int wx0 = txb * BW + (wave % WX) * TXG * 4, wy0 = tyb * BH + (wave / WX) * TR;
    const bool staged = wx0 + TXG * 4 <= d.w && wy0 + TR <= d.h &&
                        ((((uintptr_t)d.p) | (uintptr_t)d.rs | (uintptr_t)d.fs | (uintptr_t)(wx0 * C)) & 15) == 0;
    const bool valid = y < d.h && x0 < d.w;
    const u8* sp = s.p + (int64_t)f * s.fs;

    // source coordinate of the tile origin and its extremes over the tile (host guarantees no
    // 32-bit wrap inside the output rectangle, so plain int arithmetic equals libImaging's)
    const int XT = P.fx[2] + P.fx[1] * (tyb * BH) + P.fx[0] * (txb * BW);
    const int YT = P.fx[5] + P.fx[4] * (tyb * BH) + P.fx[3] * (txb * BW);
    const int ax = (BW - 1) * P.fx[0], bx = (BH - 1) * P.fx[1], ay = (BW - 1) * P.fx[3], by = (BH - 1) * P.fx[4];
    const int sx_lo = max((XT + min(ax, 0) + min(bx, 0)) >> 16, 0), sx_hi = min((XT + max(ax, 0) + max(bx, 0)) >> 16, s.w - 1);
    const int sy_lo = max((YT + min(ay, 0) + min(by, 0)) >> 16, 0), sy_hi = min((YT + max(ay, 0) + max(by, 0)) >> 16, s.h - 1);
    const int bwc = sx_hi - sx_lo + 1, bhc = sy_hi - sy_lo + 1;

    stage_bbox<PITCH, IMGXF_NEAREST_NB>(s, sp, srct, lane, wave, sx_lo, sy_lo, sx_hi, sy_hi, bwc, bhc);
    __syncthreads();

    const u32 fillw = (u32)P.fill[0] | ((u32)P.fill[1] << 8) | ((u32)P.fill[2] << 16);
    const int XL = XT + lx * P.fx[0] + ly * P.fx[1], YL = YT + lx * P.fx[3] + ly * P.fx[4];
    u32 px[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xi = ((XL + k * P.fx[0]) >> 16) - sx_lo, yi = ((YL + k * P.fx[3]) >> 16) - sy_lo;
        const bool in = bwc > 0 && bhc > 0 && (u32)xi < (u32)bwc && (u32)yi < (u32)bhc;
        const u32 v = srct[in ? yi * PITCH + xi : 0];
        px[k] = in ? (v & 0xffffffu) : fillw;
    }
    // 4 RGBX pixels -> 12 packed bytes
    u32 od[3];
    od[0] = px[0] | (px[1] << 24);
    od[1] = (px[1] >> 8) | (px[2] << 16);
    od[2] = (px[2] >> 16) | (px[3] << 8);
    if (staged) {
        u8* seg = d.row(f, y) + (x0 - (lane % TXG) * 4) * C;
        staged_store<C, TXG>(stage[wave], od, lane, seg);
        return;
    }
    if (!valid) return;
    u8* dp = d.row(f, y) + x0 * C;
    const int npx = min(4, d.w - x0);
    if (npx == 4 && ((((uintptr_t)dp) & 3) == 0)) {
#pragma unroll
        for (int qq = 0; qq < C; ++qq) ((u32*)dp)[qq] = od[qq];
    } else {
        for (int e = 0; e < npx * C; ++e) dp[e] = (u8)(px[e / 3] >> (8 * (e % 3)));
    }
}

// ---------------------------------------------------------------------------------------
// NEAREST with LDS-DMA staging (16-byte aligned sources).  The kernel above is bound by the L1
// address path: its staging loads are dwords at a 3-byte lane stride and cost ~16 TCP accesses
// per wave-level load for 192 useful bytes.  Here the bounding box is fetched as packed RGB rows
// in 16-byte chunks straight into LDS (global_load_lds_dwordx4: 1 KiB per wave-level load, no
// VGPR round trip): lane i of the workgroup's DMA round j owns chunk idx = 256 j + i of the
// [row][DCH] chunk grid, whose LDS image is simply idx * 16.  The gather then reads the two
// aligned dwords around byte 3*x of the packed row and funnel-shifts.
// Preconditions (host): src data / row stride / frame stride / row bytes all multiples of 16,
// bbox <= 49 pixels wide (<= DCH chunks whatever its alignment) and <= 52 rows.
// ---------------------------------------------------------------------------------------
#define IMGXF_AFF_GLDS16(gptr, lptr)                                                           \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),    \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// BH = 32 with DCHT = 11 chunks per staged row (box <= 49 pixels wide), or BH = 64 with DCHT = 13
// (box <= 64 wide): the taller tile halves the per-tile setup per pixel, a lane then gathers
// two groups of 4 pixels.
template <int BH, int DCHT>
__global__ __launch_bounds__(256) void affine_nearest_dma_kernel(View s, View d, AffineParams P, int ntx, int nty) {
    constexpr int C = 3, TXG = 8, TR = 8, BW = 32, NH = BH / 32, DPITCH = DCHT * 16;
    constexpr u32 DIVM = (65536u + DCHT - 1) / DCHT;                  // idx / DCHT == (idx * DIVM) >> 16 for idx < 2^12
    extern __shared__ __attribute__((aligned(16))) char srcb[];      // bhc x DPITCH bytes (+16)
    __shared__ __attribute__((aligned(16))) u32 stage[4][64 * C + 2 * (64 / TXG) + 4];
    const int nblocks = ntx * nty;
    const int orig = blockIdx.x, xcd = orig & 7, q = nblocks >> 3, r = nblocks & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int f = blockIdx.y;
    int tyb = P.ntx_magic ? (int)__umulhi((u32)logical, P.ntx_magic) : logical, txb = logical - tyb * ntx;
    if (P.strip_w) {
        // vertical strips: a tile's box overlaps its neighbours' (2x for a pure rotation); inside a strip
        // walked row by row both neighbours are a few workgroups away on the same L2
        const int l = orig >> 3;
        tyb = l / P.strip_w; txb = xcd * P.strip_w + (l - tyb * P.strip_w);
        if (txb >= ntx || tyb >= nty) return;
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lx = (lane % TXG) * 4;
    const int x0 = txb * BW + lx;
    const int wx0 = txb * BW;
    const u8* sp = s.p + (int64_t)f * s.fs;

    const int XT = P.fx[2] + P.fx[1] * (tyb * BH) + P.fx[0] * (txb * BW);
    const int YT = P.fx[5] + P.fx[4] * (tyb * BH) + P.fx[3] * (txb * BW);
    const int ax = (BW - 1) * P.fx[0], bx = (BH - 1) * P.fx[1], ay = (BW - 1) * P.fx[3], by = (BH - 1) * P.fx[4];
    const int sx_lo = max((XT + min(ax, 0) + min(bx, 0)) >> 16, 0), sx_hi = min((XT + max(ax, 0) + max(bx, 0)) >> 16, s.w - 1);
    const int sy_lo = max((YT + min(ay, 0) + min(by, 0)) >> 16, 0), sy_hi = min((YT + max(ay, 0) + max(by, 0)) >> 16, s.h - 1);
    const int bwc = sx_hi - sx_lo + 1, bhc = sy_hi - sy_lo + 1;
    const int a0 = (sx_lo * 3) & ~15;                  // first staged byte of every row
    const int last_chunk = s.w * 3 - 16;               // chunks past the row end re-read its last one

    if (bwc > 0 && bhc > 0) {
        const int nchunks = bhc * DCHT;                // host: bhc * DCHT < 4096
        for (int base = wave * 64; base < nchunks; base += 256) {          // wave-uniform trip count
            const int idx = base + lane;
            if (idx < nchunks) {
                const int row = (int)(((u32)idx * DIVM) >> 16);
                const int ch = idx - row * DCHT;
                const u8* gp = sp + (int64_t)(sy_lo + row) * s.rs + min(a0 + ch * 16, last_chunk);
                IMGXF_AFF_GLDS16(gp, srcb + base * 16);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const u32 fillw = (u32)P.fill[0] | ((u32)P.fill[1] << 8) | ((u32)P.fill[2] << 16);
    const bool dst16 = ((((uintptr_t)d.p) | (uintptr_t)d.rs | (uintptr_t)d.fs | (uintptr_t)(wx0 * C)) & 15) == 0;
#pragma unroll
    for (int half = 0; half < NH; ++half) {
        const int ly = (wave * NH + half) * TR + lane / TXG;
        const int y = tyb * BH + ly;
        const int wy0 = tyb * BH + (wave * NH + half) * TR;
        const bool staged = dst16 && wx0 + TXG * 4 <= d.w && wy0 + TR <= d.h;
        const bool valid = y < d.h && x0 < d.w;
        const int XL = XT + lx * P.fx[0] + ly * P.fx[1], YL = YT + lx * P.fx[3] + ly * P.fx[4];
        u32 px[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xi = ((XL + k * P.fx[0]) >> 16) - sx_lo, yi = ((YL + k * P.fx[3]) >> 16) - sy_lo;
            const bool in = bwc > 0 && bhc > 0 && (u32)xi < (u32)bwc && (u32)yi < (u32)bhc;
            const int bo = in ? (int)__umul24((u32)yi, (u32)DPITCH) + (xi + sx_lo) * 3 - a0 : 0;
            const u32* wp = (const u32*)(srcb + (bo & ~3));
            const u32 lo = wp[0], hi = wp[1];
            const u32 t = __builtin_amdgcn_alignbyte(hi, lo, (u32)bo & 3u);
            px[k] = in ? (t & 0xffffffu) : fillw;
        }
        u32 od[3];
        od[0] = px[0] | (px[1] << 24);
        od[1] = (px[1] >> 8) | (px[2] << 16);
        od[2] = (px[2] >> 16) | (px[3] << 8);
        if (staged) {
            u8* seg = d.row(f, y) + (x0 - (lane % TXG) * 4) * C;
            staged_store<C, TXG>(stage[wave], od, lane, seg);
        } else if (valid) {
            u8* dp = d.row(f, y) + x0 * C;
            const int npx = min(4, d.w - x0);
            if (npx == 4 && ((((uintptr_t)dp) & 3) == 0)) {
#pragma unroll
                for (int qq = 0; qq < C; ++qq) ((u32*)dp)[qq] = od[qq];
            } else {
                for (int e = 0; e < npx * C; ++e) dp[e] = (u8)(px[e / 3] >> (8 * (e % 3)));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// BICUBIC horizontal-only (apply_shear, /root/reference/transformation.py:212-226: matrix
// (1, sh, -shift, 0, 1, 0), white fill): m3 == 0, m4 == 1, m5 integral, so the source row is
// y + m5 exactly and only libImaging's row cubic  p1 + d(p2 + d(p3 + d p4))  is evaluated.
// The generic kernel does that in fp64 from 12 byte loads per pixel (VALU-bound on fp64).  Here:
// coordinates stay in un-contracted fp64 (floor / bounds must agree with libImaging), the four
// taps of a pixel are one 12-byte load, and the cubic runs in fp32 — p2, p3, p4 are exact
// (integers <= 1020), the three Horner FMAs and d's rounding give |fp32 - fp64| <= 1.5e-4 +
// 5600 * 2^-25 = 3.2e-4.  PRECISE: a value within GUARD = 4e-4 of an integer (the only place
// where truncation / the 0 and 255 clamps can differ) is re-evaluated with libImaging's fp64
// sequence; so are pixels whose taps touch the left / right image border (clamped taps).
// ---------------------------------------------------------------------------------------
typedef uint32_t u32x2_sh __attribute__((ext_vector_type(2), aligned(1)));
typedef uint32_t u32x4_ua __attribute__((ext_vector_type(4), aligned(1)));

// libImaging's fp64 sequence for one pixel of a horizontal-only bicubic transform (row = source row
// yi clamped, yok = the row passes the bounds test, a1y = m1 * (y + 0.5))
__device__ __forceinline__ void shear_exact_px(const View& s, const AffineParams& P, const u8* row, bool yok, double a1y,
                                               int x, u8 (&px)[3]) {
    constexpr int C = 3;
    const double xc = (double)x + 0.5;
    double xin = __dadd_rn(__dadd_rn(__dmul_rn(P.m[0], xc), a1y), P.m[2]);
    if (!(yok && xin >= 0.0 && xin < (double)s.w)) {
#pragma unroll
        for (int j = 0; j < C; ++j) px[j] = P.fill[j];
        return;
    }
    xin -= 0.5;
    const double xfl = floor(xin);
    const int xi = (int)xfl;
    const double dx = xin - xfl;
    int xs[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) xs[t] = clampi(xi - 1 + t, 0, s.w - 1) * C;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const double v = cubic<PreciseArith>((double)row[xs[0] + j], (double)row[xs[1] + j],
                                             (double)row[xs[2] + j], (double)row[xs[3] + j], dx);
        px[j] = v <= 0.0 ? (u8)0 : (v >= 255.0 ? (u8)255 : (u8)(int)v);
    }
}

// Unit-step rows, second pass: the few 4-pixel groups at both ends of the source row (taps
// clamped at the image edge, or partly outside) that shear_bicubic_kernel<.., true> left out.
// One lane per (row, candidate group); pixels inside the source are evaluated with the fp64
// sequence and stored byte-wise.
__global__ __launch_bounds__(256) void shear_edges_kernel(View s, View d, AffineParams P) {
    constexpr int C = 3;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int slot = t & 7, y = t >> 3;
    const int f = blockIdx.y;
    if (y >= d.h || slot >= 6) return;
    const double yc = (double)y + 0.5;
    const double a1y = __dmul_rn(P.m[1], yc);
    const double yin = __dadd_rn(yc, P.m[5]);
    const bool yok = yin >= 0.0 && yin < (double)s.h;
    if (!yok) return;                                       // whole row is fill: nothing was left out
    const int yi = (int)floor(yin - 0.5);
    const u8* row = s.p + (int64_t)f * s.fs + (int64_t)clampi(yi, 0, s.h - 1) * s.rs;
    const double u = __dadd_rn(a1y, P.m[2]);
    const double fu = floor(u), du = u - fu;
    const double eps = 9.313225746154785e-10;
    const bool row_fast = du > eps && du < 1.0 - eps && fabs(du - 0.5) > eps && fabs(fu) < 1.0e9;
    if (!row_fast) return;                                  // the main kernel did this row pixel by pixel
    const int iu = (int)fu;
    // candidate groups: x0 = 4 * floor(a / 4) + 4 * (slot % 3) around the first (a = -iu - 4) and the
    // last (a = w - 6 - iu) source columns
    const int a = slot < 3 ? -iu - 4 : s.w - 6 - iu;
    const int x0 = ((a >> 2) << 2) + 4 * (slot % 3);
    if (x0 < 0 || x0 + 4 > d.w) return;                     // (ragged last group: done by the main kernel)
    if (slot >= 3) {                                        // do not redo a group the left window already covers
        const int al = -iu - 4, xl = ((al >> 2) << 2);
        if (x0 >= xl && x0 <= xl + 8) return;
    }
    const int xi0 = x0 + iu;
    if (xi0 >= 1 && xi0 + 7 <= s.w && x0 + 4 <= d.w) return;   // an ordinary group: the main kernel did it
    for (int k = 0; k < 4 && x0 + k < d.w; ++k) {
        const int xi = xi0 + k;
        const bool ok = du < 0.5 ? (xi >= 0 && xi <= s.w - 1) : (xi >= -1 && xi <= s.w - 2);
        if (!ok) continue;                                  // fill colour, already written
        u8 px[C];
        shear_exact_px(s, P, row, yok, a1y, x0 + k, px);
        u8* dp = d.row(f, y) + (x0 + k) * C;
        dp[0] = px[0]; dp[1] = px[1]; dp[2] = px[2];
    }
}

template <bool PRECISE, bool DEFER>   // DEFER: unit-step rows leave their edge groups to shear_edges_kernel
__global__ __launch_bounds__(256) void shear_bicubic_kernel(View s, View d, AffineParams P) {
    constexpr int C = 3;
    constexpr float GUARD = 4.0e-4f;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int xg = (blockIdx.x * 4 + wv) * 64 + lane;   // group of 4 output pixels; a workgroup = 1024 pixels of one row
    const int y = blockIdx.y;
    const int f = blockIdx.z;
    const int x0 = xg * 4;
    if (y >= d.h || x0 >= d.w) return;
    const u8* sp = s.p + (int64_t)f * s.fs;
    const double yc = (double)y + 0.5;
    const double a1y = __dmul_rn(P.m[1], yc);
    // yin = m3*xc + m4*yc + m5 = yc + m5 exactly (host-checked m3 == 0, m4 == 1, m5 integral)
    const double yin = __dadd_rn(yc, P.m[5]);
    const bool yok = yin >= 0.0 && yin < (double)s.h;
    const int yi = (int)floor(yin - 0.5);
    const u8* row = sp + (int64_t)clampi(yi, 0, s.h - 1) * s.rs;

    auto exact_px = [&](int x, u8 (&px)[C]) { shear_exact_px(s, P, row, yok, a1y, x, px); };

    u32 od[C] = {0u, 0u, 0u};
    u32 need = 0;
    // ---- unit-step rows (m0 == 1: apply_shear).  Then xin(x) - 0.5 = x + u with u = a1*yc + a2 up
    // to 3 * 2^-40 of rounding, so every pixel of the row has the same fraction d = frac(u) and
    // source column x + floor(u), unless d is within 2^-30 of 0, 0.5 (the bounds test) or 1 — those
    // rows take the per-pixel path below.  The cubic is then the fixed 4-tap filter
    //   w1 = -d + 2d^2 - d^3,  w2 = 1 - 2d^2 + d^3,  w3 = d + d^2 - d^3,  w4 = -d^2 + d^3
    // (libImaging's Horner form expanded; weights from fp64, rounded once), a lane's 4 pixels share
    // 7 source pixels = one 24-byte load and 21 conversions, and |fp32 - fp64| <= 1.26 * 255 * 2^-25
    // + 4 * 1.5e-5 = 7e-5 < GUARD_ROW.
    constexpr float GUARD_ROW = 1.2e-4f;
    bool done = false;
    if (P.m[0] == 1.0) {
        const double u = __dadd_rn(a1y, P.m[2]);
        const double fu = floor(u), du = u - fu;
        const double eps = 9.313225746154785e-10;      // 2^-30
        const bool row_sure = du > eps && du < 1.0 - eps && fabs(du - 0.5) > eps;
        const int xi0 = x0 + (int)fu;
        const bool row_fast = row_sure && fabs(fu) < 1.0e9;
        if (row_fast && !(yok && xi0 >= 1 && xi0 + 7 <= s.w && x0 + 4 <= d.w)) {
            // lanes at the ends of the source row or in the fill triangles: the bounds test is
            // integer here (d is not within 2^-30 of 0 / 0.5 / 1), outside pixels are the fill
            // colour, the few inside ones go to the fp64 hand-back
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int xi = xi0 + k;
                const bool ok = yok && (du < 0.5 ? (xi >= 0 && xi <= s.w - 1) : (xi >= -1 && xi <= s.w - 2));
                if (ok) {
                    if (!DEFER || x0 + 4 > d.w) need |= 1u << k;      // (a ragged last group is never deferred)
                } else {
#pragma unroll
                    for (int j = 0; j < C; ++j) {
                        const int q = (k * C + j) >> 2, sh = ((k * C + j) & 3) * 8;
                        od[q] |= (u32)P.fill[j] << sh;
                    }
                }
            }
            done = true;
        }
        if (row_fast && !done) {
            const double d2 = du * du, d3 = d2 * du;
            const float w1 = (float)(-du + 2.0 * d2 - d3), w2 = (float)(1.0 - 2.0 * d2 + d3);
            const float w3 = (float)(du + d2 - d3), w4 = (float)(-d2 + d3);
            const u8* tp = row + (xi0 - 1) * C;                               // source pixels xi0-1 .. xi0+5 (+3 spare bytes)
            const u32x4_ua q4 = *(const u32x4_ua*)tp;
            const u32x2_sh q2 = *(const u32x2_sh*)(tp + 16);
            const u32 w[6] = {q4.x, q4.y, q4.z, q4.w, q2.x, q2.y};
            float T[7][C];
#pragma unroll
            for (int pp = 0; pp < 7; ++pp)
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    const int b = pp * C + j;
                    T[pp][j] = (float)((w[b >> 2] >> (8 * (b & 3))) & 0xffu);
                }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float dist = 0.0f;
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    const float v = fmaf(w4, T[k + 3][j], fmaf(w3, T[k + 2][j], fmaf(w2, T[k + 1][j], w1 * T[k][j])));
                    const float fl = floorf(v);
                    od[(k * C + j) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(fl, (k * C + j) & 3, od[(k * C + j) >> 2]);
                    if (PRECISE) dist = fmaxf(dist, fabsf((v - fl) - 0.5f));
                }
                need |= (PRECISE && dist > 0.5f - GUARD_ROW) ? 1u << k : 0u;
            }
            done = true;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (done) break;
        const int x = x0 + k;
        const double xc = (double)x + 0.5;
        const double xin = __dadd_rn(__dadd_rn(__dmul_rn(P.m[0], xc), a1y), P.m[2]);
        const bool ok = yok && xin >= 0.0 && xin < (double)s.w;
        const double xs = xin - 0.5, xfl = floor(xs);
        const int xi = (int)xfl;
        const float dx = (float)(xs - xfl);
        const bool inner = ok && xi >= 1 && xi + 2 <= s.w - 1;      // all four taps unclamped
        // taps xi-1 .. xi+2: 12 contiguous bytes (a safe address when the pixel is not `inner`)
        const u8* tp = row + (inner ? (xi - 1) * C : 0);
        const u32x2_sh t2 = *(const u32x2_sh*)tp;                   // exactly 12 bytes: x2 + x1
        const u32 w0 = t2.x, w1 = t2.y, w2 = *(const u32_unaligned*)(tp + 8);
        float v[C];
        float dist = 0.0f;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            // tap t, channel j = byte 3t + j of the 12
            const float v1 = (float)((w0 >> (8 * j)) & 0xffu);
            const float v2 = (float)((j < 1 ? (w0 >> 24) : (w1 >> (8 * (j - 1)))) & 0xffu);
            const float v3 = (float)((j < 2 ? (w1 >> (8 * (j + 2))) : w2) & 0xffu);
            const float v4 = (float)((w2 >> (8 * (j + 1))) & 0xffu);
            const float p2 = v3 - v1;
            const float p3 = fmaf(2.0f, v1 - v2, v3) - v4;
            const float p4 = (v2 - v1) + (v4 - v3);
            v[j] = fmaf(dx, fmaf(dx, fmaf(dx, p4, p3), p2), v2);
            const float fl = floorf(v[j]);
            od[(k * C + j) >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(fl, (k * C + j) & 3, od[(k * C + j) >> 2]);   // saturates: <=0 -> 0, >=255 -> 255
            if (PRECISE) dist = fmaxf(dist, fabsf((v[j] - fl) - 0.5f));
        }
        const bool flag = !inner || (PRECISE && dist > 0.5f - GUARD);
        need |= flag ? 1u << k : 0u;
    }
    if (need) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((need >> k) & 1u) {
                u8 px[C];
                exact_px(x0 + k, px);
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    const int q = (k * C + j) >> 2, sh = ((k * C + j) & 3) * 8;
                    od[q] = (od[q] & ~(0xffu << sh)) | ((u32)px[j] << sh);
                }
            }
        }
    }
    u8* dp = d.row(f, y) + x0 * C;
    const int npx = min(4, d.w - x0);
    // a full wave (256 pixels of one row, 768 contiguous bytes) goes through the LDS transpose:
    // 48 dense 16-byte stores instead of three dword stores at a 12-byte lane stride
    const int xw0 = (blockIdx.x * 4 + wv) * 256;
    const bool dense = xw0 + 256 <= d.w &&
                       ((((uintptr_t)d.p) | (uintptr_t)d.rs | (uintptr_t)d.fs | (uintptr_t)(xw0 * C)) & 15) == 0;
    if (dense) {
        __shared__ __attribute__((aligned(16))) u32 stage[4][64 * C + 2 + 4];
        staged_store<C, 64>(stage[threadIdx.x >> 6], od, lane, d.row(f, y) + xw0 * C);
        return;
    }
    if (npx == 4 && ((((uintptr_t)dp) & 3) == 0)) {
#pragma unroll
        for (int q = 0; q < C; ++q) ((u32*)dp)[q] = od[q];
    } else {
        for (int e = 0; e < npx * C; ++e) {
            u32 vv = 0;
#pragma unroll
            for (int q = 0; q < C; ++q) if (q == (e >> 2)) vv = od[q];
            dp[e] = (u8)(vv >> (8 * (e & 3)));
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
        case IMGXF_FILTER_BICUBIC: {
            const bool honly = P.m[3] == 0.0 && P.m[4] == 1.0 && P.m[5] == floor(P.m[5]) && fabs(P.m[5]) < 1.0e9;
            if (honly) hipLaunchKernelGGL((affine_kernel<C, IMGXF_FILTER_BICUBIC, A, true>), grid, block, 0, st, s, d, P, dbg);
            else hipLaunchKernelGGL((affine_kernel<C, IMGXF_FILTER_BICUBIC, A>), grid, block, 0, st, s, d, P, dbg);
            break;
        }
        default: return IMGXF_ERR_ARG;
    }
    return launch_status();
}

} // namespace imgxf
#include "affine_mf.inc"
#include "affine_wq.inc"
namespace imgxf {

// Tall interior tiles: `affine_bilinear_lds_interior_kernel<.., BHT>` over every 32 x BHT tile (the
// ones that are not interior leave at once) and `affine_bilinear_lds_list_kernel` over the host's
// list of those that are not.  Returns BILINEAR_TALL_NOT_TAKEN when the taller tile's source box
// does not fit PITCH x MAXROWS of LDS or the list does not fit the launch arguments.
constexpr int BILINEAR_TALL_NOT_TAKEN = 1 << 30;

template <int BHT, int PITCH, int MAXROWS>
static int launch_bilinear_tall(const View& s, const View& d, const AffineParams& Pin, const View& dbg, const double* m,
                                int bw, int bh, int ntx, bool pr, hipStream_t st) {
    const AffineParams& P = Pin;
    const int bwt = (int)ceil(fabs(m[0]) * 31 + fabs(m[1]) * (BHT - 1)) + 4;
    const int bht = (int)ceil(fabs(m[3]) * 31 + fabs(m[4]) * (BHT - 1)) + 4;
    const int ntyt = (d.h + BHT - 1) / BHT;
    if (bwt > PITCH || bht > MAXROWS || bw > PITCH || (int64_t)ntx * ntyt * ntx >= ((int64_t)1 << 32))
        return BILINEAR_TALL_NOT_TAKEN;
    // batches: every tile loops over `fpb` frames per workgroup with the per-pixel geometry held in
    // registers (affine_mf.inc); single frames and the fp32 side output keep the per-frame kernels
    const bool want_f32 = dbg.p != nullptr;
    const int fpb = knob_int(K_AFFINE_FPB, 16);
    const int afpb = fpb;
    const bool mf = BHT == MF_TILE_H && !want_f32 && afpb >= 2 && d.n >= 2 && bht <= 64 &&
                    (int64_t)d.h * d.rs < ((int64_t)1 << 32);
    // LDS-DMA staging of packed rows needs 16-byte aligned source rows and a box of <= 52 x 52 pixels; it
    // also takes the border tiles (no list pass)
    const int nch = (bwt * 3 + 15 + 15) / 16;                  // 16-byte chunks per packed box row (any alignment of its start)
    const bool dma = mf && !knob_set(K_AFFINE_NO_DMA) && bht <= 52 && bwt <= 52 && 52 * nch <= 768 &&
                     ((((uintptr_t)s.p) | (uintptr_t)s.rs | (uintptr_t)s.fs) & 15) == 0 && s.w * 3 >= 16 &&
                     (int64_t)s.h * s.rs < ((int64_t)1 << 32);
    // round 3: wave-private boxes + whole-line stores (affine_wq.inc) when a 32 x 16 block's box fits 28 x 27 pixels
    // and both images are 16-byte aligned with 16-byte multiples as rows
    if (mf && !knob_set(K_AFFINE_NO_WQ) && !knob_set(K_AFFINE_NO_DMA)) {
        const int bwq = (int)ceil(fabs(m[0]) * 31 + fabs(m[1]) * 15) + 4, bhq = (int)ceil(fabs(m[3]) * 31 + fabs(m[4]) * 15) + 4;
        const int nchq = (bwq * 3 + 15 + 15) / 16;
        const int ntxq = (d.w + 127) / 128, ntyq = (d.h + 15) / 16;
        const bool aligned = ((((uintptr_t)s.p) | (uintptr_t)s.rs | (uintptr_t)s.fs | ((uintptr_t)d.p) | (uintptr_t)d.rs | (uintptr_t)d.fs) & 15) == 0 &&
                             (d.w * 3) % 16 == 0 && s.w * 3 >= 16;
        if (aligned && bwq <= WQ_PITCH && bhq * nchq <= WQ_MAXCH && bhq * (WQ_PITCH / 4) <= 192 &&
            (int64_t)s.h * s.rs < ((int64_t)1 << 32) && (int64_t)ntxq * ntyq < ((int64_t)1 << 27)) {
            AffineParams Pq = Pin;
            Pq.strip_w = 0;
            // 24 frames per workgroup: the tile set-up (~1500 instructions) is amortised further than at 16, the launch still
            // has > 20 000 workgroups to drain evenly (sweep 8 ... 128 on 128 4K / 512 1080p / 32 4K frames:
            // profiles/r03_experiments/affine_fpb_sweep.txt; 24 is the fastest or within 0.5 % everywhere)
            const int fpbq = knob_int(K_AFFINE_FPB, 24);
            dim3 grid((unsigned)(ntxq * ntyq), (unsigned)((d.n + fpbq - 1) / fpbq));
            // vertical strips of tile columns per XCD from 8 columns on: vertically adjacent tiles (whose boxes overlap 2x in y)
            // then sit a few workgroups apart on ONE L2 — at 1080p (15 columns) row-major ranges fetched every source byte twice
            if (ntxq >= 8 && !knob_set(K_AFFINE_NO_STRIPS)) { Pq.strip_w = (ntxq + 7) / 8; grid.x = (unsigned)(8 * Pq.strip_w * ntyq); }
            const size_t lds = (size_t)4 * ((size_t)bhq * WQ_PITCH * 4 + 2 * ((size_t)bhq * nchq * 16 + 64)) + 2 * 16 * 384;
            if (pr) hipLaunchKernelGGL((affine_bilinear_wq_kernel<true>), grid, dim3(256), lds, st, s, d, Pq, ntxq, ntyq, fpbq, nchq, bhq, knob_int(K_AFFINE_MF_DBG, 0));
            else hipLaunchKernelGGL((affine_bilinear_wq_kernel<false>), grid, dim3(256), lds, st, s, d, Pq, ntxq, ntyq, fpbq, nchq, bhq, knob_int(K_AFFINE_MF_DBG, 0));
            return launch_status();
        }
    }
    // (64 x 32 tiles, IMGXF_AFFINE_MF_WIDE: 192-byte rows; measured slower than 32 x 64 — 1.76 vs 1.63 ms — although their
    // stores alone are faster: kept as an experiment knob)
    if (dma && knob_set(K_AFFINE_MF_WIDE) && !knob_set(K_AFFINE_MF_NARROW)) {
        // 64 x 32 tiles when THEIR source box fits the same LDS image: 192-byte output rows (two of three 128-byte lines
        // written whole by one workgroup) instead of 96-byte ones — see affine_mf.inc, WIDE
        const int bww = (int)ceil(fabs(m[0]) * 63 + fabs(m[1]) * 31) + 4, bhw = (int)ceil(fabs(m[3]) * 63 + fabs(m[4]) * 31) + 4;
        const int nchw = (bww * 3 + 15 + 15) / 16;
        const int ntxw = (d.w + 63) / 64, ntyw = (d.h + 31) / 32;
        if (bww <= 52 && bhw <= 52 && 52 * nchw <= 768 && (int64_t)ntxw * ntyw < ((int64_t)1 << 28)) {
            dim3 grid((unsigned)(ntxw * ntyw), (unsigned)((d.n + afpb - 1) / afpb));
            const int npk = knob_set(K_AFFINE_PK3) ? 3 : 2;
            AffineParams Ps = Pin;
            Ps.ntx_magic = (u32)((((uint64_t)1 << 32) + ntxw - 1) / ntxw);
            if (ntxw == 1) Ps.ntx_magic = 0;
            if (ntxw >= 16 && !knob_set(K_AFFINE_NO_STRIPS)) { Ps.strip_w = (ntxw + 7) / 8; grid.x = (unsigned)(8 * Ps.strip_w * ntyw); }
            const size_t lds = (size_t)52 * 56 * 4 + npk * ((size_t)52 * nchw * 16 + 64);
            if (pr) hipLaunchKernelGGL((affine_bilinear_mf_kernel<true, 56, 13, true, true>), grid, dim3(256), lds, st, s, d, Ps, ntxw, ntyw, fpb, nchw, npk, knob_int(K_AFFINE_MF_DBG, 0));
            else hipLaunchKernelGGL((affine_bilinear_mf_kernel<false, 56, 13, true, true>), grid, dim3(256), lds, st, s, d, Ps, ntxw, ntyw, fpb, nchw, npk, knob_int(K_AFFINE_MF_DBG, 0));
            return launch_status();
        }
    }
    if (dma) {
        dim3 grid((unsigned)(ntx * ntyt), (unsigned)((d.n + afpb - 1) / afpb));
        // RGBX pitch 56: fewest bank conflicts of the multiples of 4 (simulated for 30 deg / 1.5x: 4.0 vs 5.8 LDS cycles per gather read at 52)
        const int npk = knob_set(K_AFFINE_PK3) ? 3 : 2;        // packed-row buffers (A/B knob)
        AffineParams Ps = Pin;
        if (ntx >= 16 && !knob_set(K_AFFINE_NO_STRIPS)) { Ps.strip_w = (ntx + 7) / 8; grid.x = (unsigned)(8 * Ps.strip_w * ntyt); }
        const size_t lds = (size_t)52 * 56 * 4 + npk * ((size_t)52 * nch * 16 + 64);
        if (pr) hipLaunchKernelGGL((affine_bilinear_mf_kernel<true, 56, 13, true>), grid, dim3(256), lds, st, s, d, Ps, ntx, ntyt, fpb, nch, npk, knob_int(K_AFFINE_MF_DBG, 0));
        else hipLaunchKernelGGL((affine_bilinear_mf_kernel<false, 56, 13, true>), grid, dim3(256), lds, st, s, d, Ps, ntx, ntyt, fpb, nch, npk, knob_int(K_AFFINE_MF_DBG, 0));
        return launch_status();
    }
    // the kernel's interior test (bilinear_tile<.., BHT, true>) with the same integers
    TileList list;
    list.n = 0;
    const int64_t ax = 31 * P.q0, bx = (BHT - 1) * P.q1, ay = 31 * P.q3, by = (BHT - 1) * P.q4;
    const int64_t xlo_o = std::min(ax, (int64_t)0) + std::min(bx, (int64_t)0), xhi_o = std::max(ax, (int64_t)0) + std::max(bx, (int64_t)0);
    const int64_t ylo_o = std::min(ay, (int64_t)0) + std::min(by, (int64_t)0), yhi_o = std::max(ay, (int64_t)0) + std::max(by, (int64_t)0);
    for (int ty = 0; ty < ntyt && list.n <= BILINEAR_LIST_MAX; ++ty)
        for (int tx = 0; tx < ntx; ++tx) {
            const int64_t XT = P.x00 + (int64_t)(tx * 32) * P.q0 + (int64_t)(ty * BHT) * P.q1;
            const int64_t YT = P.y00 + (int64_t)(tx * 32) * P.q3 + (int64_t)(ty * BHT) * P.q4;
            const bool clean = (int)((XT + xlo_o) >> 40) >= 0 && (int)((YT + ylo_o) >> 40) >= 0 &&
                               (int)((XT + xhi_o) >> 40) + 1 <= s.w - 1 && (int)((YT + yhi_o) >> 40) + 1 <= s.h - 1 &&
                               tx * 32 + 32 <= d.w && ty * BHT + BHT <= d.h &&
                               !((int)((XT + xhi_o) >> 40) + 1 == s.w - 1 && (int)((YT + yhi_o) >> 40) + 1 == s.h - 1);
            if (!clean) {
                if (list.n < BILINEAR_LIST_MAX) list.idx[list.n] = (u32)(ty * ntx + tx);
                ++list.n;
            }
        }
    if (list.n > BILINEAR_LIST_MAX) return BILINEAR_TALL_NOT_TAKEN;
    if (list.n < ntx * ntyt && mf) {
        const dim3 grid((unsigned)(ntx * ntyt), (unsigned)((d.n + afpb - 1) / afpb));
        const size_t lds = (size_t)2 * PITCH * (bht <= 52 ? 52 : 64) * 4 + 16;     // all 4 * NBR rows are written
        if (bht <= 52) {
            if (pr) hipLaunchKernelGGL((affine_bilinear_mf_kernel<true, PITCH, 13, false>), grid, dim3(256), lds, st, s, d, P, ntx, ntyt, fpb, 0, 0, 0);
            else hipLaunchKernelGGL((affine_bilinear_mf_kernel<false, PITCH, 13, false>), grid, dim3(256), lds, st, s, d, P, ntx, ntyt, fpb, 0, 0, 0);
        } else {
            if (pr) hipLaunchKernelGGL((affine_bilinear_mf_kernel<true, PITCH, 16, false>), grid, dim3(256), lds, st, s, d, P, ntx, ntyt, fpb, 0, 0, 0);
            else hipLaunchKernelGGL((affine_bilinear_mf_kernel<false, PITCH, 16, false>), grid, dim3(256), lds, st, s, d, P, ntx, ntyt, fpb, 0, 0, 0);
        }
        IMGXF_CHECK(launch_status());
    } else if (list.n < ntx * ntyt) {                // at least one interior tile
        const dim3 grid((unsigned)(ntx * ntyt), (unsigned)d.n);
        const size_t lds = (size_t)PITCH * bht * 4 + 16;
        if (pr && want_f32) hipLaunchKernelGGL((affine_bilinear_lds_interior_kernel<true, PITCH, true, BHT>), grid, dim3(256), lds, st, s, d, P, dbg, ntx, ntyt);
        else if (pr) hipLaunchKernelGGL((affine_bilinear_lds_interior_kernel<true, PITCH, false, BHT>), grid, dim3(256), lds, st, s, d, P, dbg, ntx, ntyt);
        else if (want_f32) hipLaunchKernelGGL((affine_bilinear_lds_interior_kernel<false, PITCH, true, BHT>), grid, dim3(256), lds, st, s, d, P, dbg, ntx, ntyt);
        else hipLaunchKernelGGL((affine_bilinear_lds_interior_kernel<false, PITCH, false, BHT>), grid, dim3(256), lds, st, s, d, P, dbg, ntx, ntyt);
        IMGXF_CHECK(launch_status());
    }
    if (list.n > 0) {
        const dim3 grid((unsigned)((BHT / 32) * list.n), (unsigned)d.n);
        const size_t lds = (size_t)PITCH * bh * 4 + 16;
        if (pr && want_f32) hipLaunchKernelGGL((affine_bilinear_lds_list_kernel<true, PITCH, true, BHT>), grid, dim3(256), lds, st, s, d, P, dbg, ntx, list);
        else if (pr) hipLaunchKernelGGL((affine_bilinear_lds_list_kernel<true, PITCH, false, BHT>), grid, dim3(256), lds, st, s, d, P, dbg, ntx, list);
        else if (want_f32) hipLaunchKernelGGL((affine_bilinear_lds_list_kernel<false, PITCH, true, BHT>), grid, dim3(256), lds, st, s, d, P, dbg, ntx, list);
        else hipLaunchKernelGGL((affine_bilinear_lds_list_kernel<false, PITCH, false, BHT>), grid, dim3(256), lds, st, s, d, P, dbg, ntx, list);
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
    if (filter == IMGXF_FILTER_NEAREST && src->c == 3 && !knob_set(K_AFFINE_NO_LDS) &&
        ((((uintptr_t)src->data) | (uintptr_t)src->row_stride | (uintptr_t)src->frame_stride) & 3) == 0) {
        // 16.16 coordinates must not wrap 32 bits anywhere in the (tile-padded) output rectangle
        bool nowrap = true;
        for (int cy = 0; cy < 2; ++cy)
            for (int cx = 0; cx < 2; ++cx) {
                const double X = cx ? dst->w + 32 : 0, Y = cy ? dst->h + 32 : 0;
                const double xs = (double)P.fx[2] + (double)P.fx[1] * Y + (double)P.fx[0] * X;
                const double ys = (double)P.fx[5] + (double)P.fx[4] * Y + (double)P.fx[3] * X;
                if (fabs(xs) > 2.0e9 || fabs(ys) > 2.0e9) nowrap = false;
            }
        const int bw = (int)ceil((fabs((double)P.fx[0]) * 31 + fabs((double)P.fx[1]) * 31) / 65536.0) + 3;
        const int bh = (int)ceil((fabs((double)P.fx[3]) * 31 + fabs((double)P.fx[4]) * 31) / 65536.0) + 3;
        const int ntx = (d.w + 31) / 32, nty = (d.h + 31) / 32;
        if (nowrap && bw <= 97 && bh <= 100 && (int64_t)ntx * nty < 0x7fffffff && d.n <= 65535) {
            dim3 grid((unsigned)(ntx * nty), (unsigned)d.n);
            P.ntx_magic = (u32)((((uint64_t)1 << 32) + ntx - 1) / ntx);   // ntx, nty <= 1024: exact; 0 when ntx == 1
            const bool no_dma = knob_set(K_AFFINE_NO_DMA);
            // round 3: 128 x 16 tiles walked over the frames, wave-private boxes, whole-line stores (affine_wq.inc) for 16-byte
            // aligned images whose 32 x 16 block's box is at most 64 NQ_KW chunks
            {
                const int bwq = (int)ceil((fabs((double)P.fx[0]) * 31 + fabs((double)P.fx[1]) * 15) / 65536.0) + 3;
                const int bhq = (int)ceil((fabs((double)P.fx[3]) * 31 + fabs((double)P.fx[4]) * 15) / 65536.0) + 3;
                const int nchq = (bwq * 3 + 15 + 15) / 16;
                const int kwq = (bhq * nchq + 63) / 64;
                const int ntxq = (d.w + 127) / 128, ntyq = (d.h + 15) / 16;
                const bool aligned = ((((uintptr_t)s.p) | (uintptr_t)s.rs | (uintptr_t)s.fs | ((uintptr_t)d.p) | (uintptr_t)d.rs | (uintptr_t)d.fs) & 15) == 0 &&
                                     (d.w * 3) % 16 == 0 && (s.w * 3) % 16 == 0 && s.w * 3 >= 16 &&
                                     (int64_t)s.h * s.rs < ((int64_t)1 << 32);
                if (aligned && !no_dma && !knob_set(K_AFFINE_NO_WQ) && kwq <= NQ_KW && (int64_t)ntxq * ntyq < ((int64_t)1 << 27)) {
                    AffineParams Pq = P;
                    Pq.strip_w = 0;
                    unsigned gx = (unsigned)(ntxq * ntyq);
                    // measured on 128 4K frames at 22.5 degrees (profiles/r03_experiments/ab_nearest_wq.txt): one frame per workgroup
                    // (one box, one slot: 6 workgroups per CU) 1.43 ms, with XCD strips 1.49, 2 / 3 / 24 frames per workgroup
                    // 1.77 / 1.63 / 1.71, affine_nearest_dma_kernel 1.52
                    const int fpbq = max(1, min(knob_int(K_AFFINE_FPB, 1), d.n));
                    const dim3 grid(gx, (unsigned)((d.n + fpbq - 1) / fpbq));
                    const size_t nbufq = fpbq > 1 ? 2 : 1;
                    const size_t lds = nbufq * ((size_t)4 * ((size_t)bhq * nchq * 16 + 16) + 16 * 384);
#define IMGXF_NQ_LAUNCH(KWV) hipLaunchKernelGGL((affine_nearest_wq_kernel<KWV>), grid, dim3(256), lds, st, s, d, Pq, ntxq, ntyq, fpbq, nchq, bhq)
                    switch (kwq) {
                        case 1: IMGXF_NQ_LAUNCH(1); break; case 2: IMGXF_NQ_LAUNCH(2); break; case 3: IMGXF_NQ_LAUNCH(3); break;
                        case 4: IMGXF_NQ_LAUNCH(4); break; case 5: IMGXF_NQ_LAUNCH(5); break; case 6: IMGXF_NQ_LAUNCH(6); break;
                        case 7: IMGXF_NQ_LAUNCH(7); break; default: IMGXF_NQ_LAUNCH(8); break;
                    }
#undef IMGXF_NQ_LAUNCH
                    return launch_status();
                }
            }
            if (bw <= 49 && bh <= 52 && !no_dma && src->w * 3 >= 16 && (src->w * 3) % 16 == 0 &&
                ((((uintptr_t)src->data) | (uintptr_t)src->row_stride | (uintptr_t)src->frame_stride) & 15) == 0) {
                // 32x64 tiles when their source box fits 13 chunks x 80 rows, else 32x32
                const int bw64 = (int)ceil((fabs((double)P.fx[0]) * 31 + fabs((double)P.fx[1]) * 63) / 65536.0) + 3;
                const int bh64 = (int)ceil((fabs((double)P.fx[3]) * 31 + fabs((double)P.fx[4]) * 63) / 65536.0) + 3;
                const bool no_tall = knob_set(K_AFFINE_NO_TALL);
                if (bw64 <= 64 && bh64 <= 80 && !no_tall) {
                    const int nty64 = (d.h + 63) / 64;
                    unsigned gx = (unsigned)(ntx * nty64);
                    // vertical strips per XCD (3 % at 4K: 0.86 -> 0.83 ms per 64 frames); IMGXF_AFFINE_NO_STRIPS = row-major ranges
                    if (ntx >= 16 && !knob_set(K_AFFINE_NO_STRIPS)) { P.strip_w = (ntx + 7) / 8; gx = (unsigned)(8 * P.strip_w * nty64); }
                    hipLaunchKernelGGL((affine_nearest_dma_kernel<64, 13>), dim3(gx, (unsigned)d.n), dim3(256),
                                       (size_t)13 * 16 * bh64 + 32, st, s, d, P, ntx, nty64);
                } else {
                    hipLaunchKernelGGL((affine_nearest_dma_kernel<32, 11>), grid, dim3(256), (size_t)11 * 16 * bh + 32, st, s, d, P, ntx, nty);
                }
                return launch_status();
            }
            if (bw <= 49) hipLaunchKernelGGL((affine_nearest_lds_kernel<49>), grid, dim3(256), (size_t)49 * bh * 4 + 16, st, s, d, P, ntx, nty);
            else if (bw <= 65) hipLaunchKernelGGL((affine_nearest_lds_kernel<65>), grid, dim3(256), (size_t)65 * bh * 4 + 16, st, s, d, P, ntx, nty);
            else hipLaunchKernelGGL((affine_nearest_lds_kernel<97>), grid, dim3(256), (size_t)97 * bh * 4 + 16, st, s, d, P, ntx, nty);
            return launch_status();
        }
    }
    // fixed-point fast path: needs every source coordinate of the output rectangle below 2^21
    bool fixed_ok = fabs(m[0]) < 64.0 && fabs(m[3]) < 64.0;
    for (int cy = 0; cy < 2 && fixed_ok; ++cy)
        for (int cx = 0; cx < 2; ++cx) {
            const double X = (cx ? dst->w + 4 : 0) + 0.5, Y = (cy ? dst->h : 0) + 0.5;
            const double xs = m[0] * X + m[1] * Y + m[2], ys = m[3] * X + m[4] * Y + m[5];
            if (!(fabs(xs) < 2097152.0 && fabs(ys) < 2097152.0)) fixed_ok = false;
        }
    P.q0 = fixed_ok ? llround(m[0] * 1099511627776.0) : 0;
    P.q3 = fixed_ok ? llround(m[3] * 1099511627776.0) : 0;
    fixed_ok = fixed_ok && fabs(m[1]) < 64.0 && fabs(m[4]) < 64.0;
    P.q1 = fixed_ok ? llround(m[1] * 1099511627776.0) : 0;
    P.q4 = fixed_ok ? llround(m[4] * 1099511627776.0) : 0;
    for (int k = 0; k < 4; ++k) {
        P.sx[k] = (int)((k * P.q0 + (k * P.q0 >= 0 ? 32768 : -32768)) / 65536);   // round to nearest 2^-24
        P.sy[k] = (int)((k * P.q3 + (k * P.q3 >= 0 ? 32768 : -32768)) / 65536);
    }
    if (fixed_ok) {
        const double x0d = (m[0] * 0.5 + m[1] * 0.5) + m[2] - 0.5, y0d = (m[3] * 0.5 + m[4] * 0.5) + m[5] - 0.5;
        P.x00 = (int64_t)floor(x0d * 1099511627776.0);
        P.y00 = (int64_t)floor(y0d * 1099511627776.0);
    }
    if (filter == IMGXF_FILTER_BILINEAR && (src->c == 1 || src->c == 3) && src->w >= 3 && src->h >= 2 && fixed_ok) {
        const int tile_env = knob_int(K_AFFINE_TILE, 0);  // tuning knob
        const bool no_lds = knob_set(K_AFFINE_NO_LDS);
        if (src->c == 3 && !no_lds && tile_env == 0 &&
            ((((uintptr_t)src->data) | (uintptr_t)src->row_stride | (uintptr_t)src->frame_stride) & 3) == 0) {
            // source bounding box of a 32x32 output tile (translation-invariant up to rounding)
            const int bw = (int)ceil(fabs(m[0]) * 31 + fabs(m[1]) * 31) + 4;
            const int bh = (int)ceil(fabs(m[3]) * 31 + fabs(m[4]) * 31) + 4;
            const int ntx = (d.w + 31) / 32, nty = (d.h + 31) / 32;
            if (bw <= 97 && bh <= 100 && (int64_t)ntx * nty < 0x7fffffff && d.n <= 65535) {
                dim3 grid((unsigned)(ntx * nty), (unsigned)d.n);
                P.ntx_magic = (u32)((((uint64_t)1 << 32) + ntx - 1) / ntx);   // ntx, nty <= 1024: exact; 0 when ntx == 1
#define IMGXF_LDS_LAUNCH(KERNEL, PITCH, GRID, NTY, LDSB)                                            \
    do {                                                                                           \
        if (dbg.p) {                                                                               \
            if (pr) hipLaunchKernelGGL((KERNEL<true, PITCH, true>), GRID, dim3(256), LDSB, st, s, d, P, dbg, ntx, NTY); \
            else hipLaunchKernelGGL((KERNEL<false, PITCH, true>), GRID, dim3(256), LDSB, st, s, d, P, dbg, ntx, NTY); \
        } else {                                                                                   \
            if (pr) hipLaunchKernelGGL((KERNEL<true, PITCH, false>), GRID, dim3(256), LDSB, st, s, d, P, dbg, ntx, NTY); \
            else hipLaunchKernelGGL((KERNEL<false, PITCH, false>), GRID, dim3(256), LDSB, st, s, d, P, dbg, ntx, NTY); \
        }                                                                                          \
    } while (0)
#define IMGXF_LDS(PITCH)                                                                           \
    do {                                                                                           \
        const size_t lds = (size_t)PITCH * bh * 4 + 16;                                            \
        IMGXF_LDS_LAUNCH(affine_bilinear_lds_kernel, PITCH, grid, nty, lds);                       \
        return launch_status();                                                                    \
    } while (0)
                // interior tiles as 32x64 (8 pixels per lane) + a host-built list of the others, when
                // the geometry qualifies
                const bool no_tall = knob_set(K_AFFINE_NO_TALL);
                if (!no_tall) {
                    const int rc = launch_bilinear_tall<64, 49, 64>(s, d, P, dbg, m, bw, bh, ntx, pr, st);
                    if (rc != BILINEAR_TALL_NOT_TAKEN) return rc;     // (32x128 tiles measured slower: 1.18 vs 1.08 ms)
                }
                if (bw <= 49) IMGXF_LDS(49);
                if (bw <= 65) IMGXF_LDS(65);
                IMGXF_LDS(97);
#undef IMGXF_LDS
#undef IMGXF_LDS_LAUNCH
            }
        }
#define IMGXF_BIL(CC, PR, TXG, WX)                                                                 \
    do {                                                                                           \
        constexpr int BW = WX * TXG * 4, BH = (4 / WX) * (64 / TXG);                               \
        const int ntx = (d.w + BW - 1) / BW, nty = (d.h + BH - 1) / BH;                            \
        const int64_t nb = (int64_t)ntx * nty * d.n;                                               \
        if (nb > 0x7fffffff) return IMGXF_ERR_SHAPE;                                               \
        hipLaunchKernelGGL((affine_bilinear_kernel<CC, PR, TXG, WX>), dim3((unsigned)nb), dim3(256), 0, st, \
                           s, d, P, dbg, ntx, nty, (int)nb);                                       \
        return launch_status();                                                                    \
    } while (0)
        if (src->c == 3) {
            if (tile_env == 1) { if (pr) IMGXF_BIL(3, true, 64, 1); else IMGXF_BIL(3, false, 64, 1); }
            if (pr) IMGXF_BIL(3, true, 8, 2); else IMGXF_BIL(3, false, 8, 2);
        } else {
            if (pr) IMGXF_BIL(1, true, 8, 2); else IMGXF_BIL(1, false, 8, 2);
        }
#undef IMGXF_BIL
    }
    {
        const bool no_shear = knob_set(K_AFFINE_NO_SHEAR_FAST);
        const bool honly = m[3] == 0.0 && m[4] == 1.0 && m[5] == floor(m[5]) && fabs(m[5]) < 1.0e9;
        if (filter == IMGXF_FILTER_BICUBIC && honly && src->c == 3 && !dbg.p && !no_shear && src->w >= 4) {
            dim3 block(256), grid((unsigned)((d.w + 1023) / 1024), (unsigned)d.h, (unsigned)d.n);
            if (m[0] == 1.0) {
                // unit-step rows (apply_shear): main pass + a tiny pass over the row-end groups
                if (pr) hipLaunchKernelGGL((shear_bicubic_kernel<true, true>), grid, block, 0, st, s, d, P);
                else hipLaunchKernelGGL((shear_bicubic_kernel<false, true>), grid, block, 0, st, s, d, P);
                IMGXF_CHECK(launch_status());
                hipLaunchKernelGGL(shear_edges_kernel, dim3((unsigned)((d.h * 8 + 255) / 256), (unsigned)d.n), dim3(256), 0, st, s, d, P);
                return launch_status();
            }
            if (pr) hipLaunchKernelGGL((shear_bicubic_kernel<true, false>), grid, block, 0, st, s, d, P);
            else hipLaunchKernelGGL((shear_bicubic_kernel<false, false>), grid, block, 0, st, s, d, P);
            return launch_status();
        }
    }
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

// packed RGB, 4 output pixels = 3 dwords per lane (width a multiple of 4, dword-aligned destination rows):
// one dword store triple instead of twelve byte stores, 32-bit index arithmetic; source pixels stay byte loads
// (their addresses come from the tables).
__global__ __launch_bounds__(256) void scale_nearest_rgb4_kernel(View s, View d, const int* xtab, const int* ytab,
                                                                 const int* meta, Fill4 fillc) {
    typedef u32 u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
    const u32 G = (u32)d.w >> 2, per_frame = (u32)d.h * G;
    const int xmin = meta[0], xmax = meta[1];
    const u32 fillw = (u32)fillc.v[0] | ((u32)fillc.v[1] << 8) | ((u32)fillc.v[2] << 16);
    const int f = blockIdx.y;
    for (u32 t = blockIdx.x * 256u + threadIdx.x; t < per_frame; t += gridDim.x * 256u) {
        const u32 y = t / G, g = t - y * G;
        const int yi = ytab[y];
        const u8* srow = s.row(f, yi < 0 ? 0 : yi);
        u32 px[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = (int)(4 * g) + k;
            int xi = xtab[x];
            xi = xi < 0 ? 0 : (xi >= s.w ? s.w - 1 : xi);
            // one unaligned dword per pixel instead of three byte loads (the L1 address path bounds this gather); the
            // last pixel of a row takes the dword ending at its last byte, so nothing past the row is read (w >= 2: host)
            const bool last = xi == s.w - 1;
            u32 raw;
            __builtin_memcpy(&raw, srow + xi * 3 - (last ? 1 : 0), 4);
            const u32 v = last ? raw >> 8 : raw & 0xffffffu;
            px[k] = (yi >= 0 && x >= xmin && x < xmax) ? v : fillw;
        }
        u32x3_a4 o;
        o.x = px[0] | (px[1] << 24);
        o.y = (px[1] >> 8) | (px[2] << 16);
        o.z = (px[2] >> 16) | (px[3] << 8);
        *(u32x3_a4*)(d.row(f, (int)y) + 12 * g) = o;
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
    // unit scale with whole-pixel offsets (AugMix translate_x / translate_y, fall_2025/AugMix.py:34-35): ImagingScaleAffine
    // picks source column floor(x + 0.5 + m2) = x + m2, i.e. dst(x, y) = src(x + m2, y + m5) with the fill outside — the
    // one-pass translation kernel (57 % of the HBM roofline against 11 % for the table-driven gather)
    if (m[0] == 1.0 && m[4] == 1.0 && m[2] == floor(m[2]) && m[5] == floor(m[5]) && fabs(m[2]) < 1.0e9 && fabs(m[5]) < 1.0e9 &&
        same_geometry(src, dst) && !knob_set(K_NO_FAST_LEFTOVERS)) {
        uint8_t f4[4] = {0, 0, 0, 0};
        if (fill) for (int j = 0; j < dst->c; ++j) f4[j] = fill[j];
        const double tx = -m[2], ty = -m[5];
        const int dx = tx <= -32768.0 ? -32768 : (tx >= 32768.0 ? 32768 : (int)tx), dy = ty <= -32768.0 ? -32768 : (ty >= 32768.0 ? 32768 : (int)ty);
        return imgxf_translate_u8(src, dst, dx, dy, f4, stream);
    }
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
    if (d.c == 3 && (d.w & 3) == 0 && src->w >= 2 && d.n <= 65535 && (int64_t)d.h * (d.w >> 2) < 0x7fffffff &&
        ((((uintptr_t)d.p) | (uintptr_t)d.rs | (uintptr_t)d.fs) & 3) == 0) {
        int64_t b4 = ((int64_t)d.h * (d.w >> 2) + 255) / 256;
        if (b4 > 4096) b4 = 4096;
        hipLaunchKernelGGL(scale_nearest_rgb4_kernel, dim3((unsigned)b4, (unsigned)d.n), dim3(256), 0, st, make_view(src), d,
                           xtab, ytab, meta, fc);
        return launch_status();
    }
    hipLaunchKernelGGL(scale_nearest_kernel, dim3((unsigned)blocks), dim3(256), 0, st, make_view(src), d,
                       xtab, ytab, meta, fc);
    return launch_status();
}
