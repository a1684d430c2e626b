// Perspective warp behind apply_perspective_warp (/root/reference/fall_2025/transformations_code:54-66):
//   ToTensor -> torchvision RandomPerspective(p=1) on a float tensor -> ToPILImage
// which for drawn coefficients is torchvision's _perspective_grid + grid_sample(bilinear, zeros,
// align_corners=False) of [image/255 | ones] in fp32, img*mask + (1-mask)*fill with fill = 0, then
// mul(255).byte().  Every fp32 operation below is written in the order (and with the fused
// multiply-adds) that the torch CPU build evaluates them in, so the bytes match; the library is
// built with -ffp-contract=off, so only the fmaf calls fuse.
//
// A 64x16 output tile maps to a convex quadrilateral of the source; its bounding box (from the
// four tile corners, plus a margin for rounding) is staged in LDS already divided by 255 (one
// float per byte: 1.2 conversions per output sample instead of 4), and the four taps are read
// from there; the output tile leaves through LDS as dwords.  Tiles whose box does not fit or
// whose denominator changes sign (extreme coefficients) gather from global memory instead.
// VALU-issue bound (PMC: VALU busy ~100 %): ~77 instructions per output pixel, most of them
// the fp32 sequence above, which cannot be reassociated without changing bytes.
#include "imgxf_common.h"

namespace imgxf {

constexpr int PV_TW = 64, PV_TH = 16;          // output tile
constexpr int PV_LDS_FLOATS = 8 * 1024;        // staged source box, one float per byte (32 KiB)
constexpr int PV_MAX_FRAMES = 48;              // coefficient sets per launch (kernarg budget)

struct PerspCoef { float t[6]; float c6, c7; };   // t = c[0..5] / (0.5*ow | 0.5*oh)
struct PerspArgs {
    PerspCoef k[PV_MAX_FRAMES];
    int per_frame;      // 0: k[0] for every frame
    int frame0;         // first frame of this launch
};

// v / 255 correctly rounded for integer-valued v in [0, 255] (Tensor.div(255) in fp32): one
// multiply by RN(1/255) and one residual correction (exhaustively equal to the IEEE quotient)
__device__ __forceinline__ float unit255(float v) {
    const float r = __uint_as_float(0x3b808081u);
    const float q = v * r;
    return fmaf(fmaf(-255.0f, q, v), r, q);
}

__device__ __forceinline__ void persp_src(const PerspCoef& k, float bx, float by, float fw, float fh,
                                          float& ix, float& iy) {
    const float nx = fmaf(by, k.t[1], bx * k.t[0]) + k.t[2];
    const float ny = fmaf(by, k.t[4], bx * k.t[3]) + k.t[5];
    const float dn = fmaf(by, k.c7, bx * k.c6) + 1.0f;
    const float gx = nx / dn - 1.0f;
    const float gy = ny / dn - 1.0f;
    ix = fmaf(gx + 1.0f, fw, -1.0f) / 2.0f;
    iy = fmaf(gy + 1.0f, fh, -1.0f) / 2.0f;
}

// The same with the two IEEE quotients sharing one refined reciprocal: the compiler's own fdiv
// expansion (rcp, two Newton steps on the quotient, final fma) minus the operand scaling, which
// only acts on exponents beyond +-96 — callers guarantee 1e-3 < dn and finite numerators.
__device__ __forceinline__ float div_by(float n, float d, float r) {
    const float q0 = n * r;
    const float q1 = fmaf(fmaf(-d, q0, n), r, q0);
    return fmaf(fmaf(-d, q1, n), r, q1);
}
__device__ __forceinline__ void persp_src_fast(const PerspCoef& k, float bx, float by, float fw, float fh,
                                               float& ix, float& iy) {
    const float nx = fmaf(by, k.t[1], bx * k.t[0]) + k.t[2];
    const float ny = fmaf(by, k.t[4], bx * k.t[3]) + k.t[5];
    const float dn = fmaf(by, k.c7, bx * k.c6) + 1.0f;
    const float r0 = __builtin_amdgcn_rcpf(dn);
    const float r = fmaf(fmaf(-dn, r0, 1.0f), r0, r0);
    const float gx = div_by(nx, dn, r) - 1.0f;
    const float gy = div_by(ny, dn, r) - 1.0f;
    ix = fmaf(gx + 1.0f, fw, -1.0f) * 0.5f;
    iy = fmaf(gy + 1.0f, fh, -1.0f) * 0.5f;
}

// min over the four lanes of a quad (DPP quad_perm swaps), the same value in all four
__device__ __forceinline__ float quad_min(float v) {
    float o = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v = fminf(v, o);
    o = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));         // quad_perm [2,3,0,1]
    return fminf(v, o);
}

template <int C>
__global__ __launch_bounds__(256) void perspective_kernel(View s, View d, PerspArgs a) {
    __shared__ __attribute__((aligned(16))) float box[PV_LDS_FLOATS];
    __shared__ __attribute__((aligned(16))) u8 outt[PV_TH][PV_TW * C];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.z;
    const PerspCoef& k = a.k[a.per_frame ? f : 0];
    const int fr = a.frame0 + f;
    const int tx0 = blockIdx.x * PV_TW, ty0 = blockIdx.y * PV_TH;
    const float fw = (float)s.w, fh = (float)s.h;

    // source bounding box of the tile from its four corner pixels (uniform across the block)
    const int tx1 = min(tx0 + PV_TW, d.w) - 1, ty1 = min(ty0 + PV_TH, d.h) - 1;
    // wave 0 works out the source box (lane q of every quad evaluates corner q, a quad-wide
    // min / max combines them) and hands it to the other waves through LDS
    __shared__ int boxinfo[8];
    const bool src4 = ((((uintptr_t)s.p) | (uintptr_t)s.rs | (uintptr_t)s.fs) & 3) == 0 &&
                      s.rs * (int64_t)s.h < ((int64_t)1 << 32);       // 32-bit offsets within a frame
    if (wave == 0) {
        const int q = lane & 3;
        const float bx = (float)((q & 1) ? tx1 : tx0) + 0.5f, by = (float)((q & 2) ? ty1 : ty0) + 0.5f;
        float ix, iy;
        persp_src(k, bx, by, fw, fh, ix, iy);
        // the denominator must keep one sign over the tile for the quadrilateral argument to hold
        const float dn = fmaf(by, k.c7, bx * k.c6) + 1.0f;
        const float bad = (fabsf(ix) < 1.0e9f && fabsf(iy) < 1.0e9f) ? 0.0f : 1.0f;   // NaN -> 1
        const float lox = quad_min(ix), hix = -quad_min(-ix);
        const float loy = quad_min(iy), hiy = -quad_min(-iy);
        const float dmin = quad_min(dn);
        const bool finite = quad_min(-bad) == 0.0f;
        bool staged = finite && dmin > 1.0e-3f && dmin < 1.0e6f;
        bool interior = false;
        int bx0 = 0, by0 = 0, bw = 0, bh = 0, pitch = 0, gb0 = 0;
        if (staged) {
            // 1 px of margin for the rounding of the corner estimates, 1 more for the right / bottom tap
            const int ux0 = (int)floorf(lox) - 1, uy0 = (int)floorf(loy) - 1;
            const int ux1 = (int)floorf(hix) + 2, uy1 = (int)floorf(hiy) + 2;
            bx0 = max(ux0, 0); by0 = max(uy0, 0);
            const int bx1 = min(ux1, s.w - 1), by1 = min(uy1, s.h - 1);
            interior = ux0 >= 0 && uy0 >= 0 && ux1 <= s.w - 1 && uy1 <= s.h - 1;
            bw = bx1 - bx0 + 1; bh = by1 - by0 + 1;
            if (bw <= 0 || bh <= 0) { bw = bh = 0; }           // the tile sees no source pixel at all
            gb0 = src4 ? (bx0 * C) & ~3 : bx0 * C;              // first staged byte of a source row
            pitch = bw > 0 ? (((bx1 + 1) * C - gb0 + 3) & ~3) : 0;     // floats per staged row
            staged = (int64_t)pitch * bh <= PV_LDS_FLOATS;
            interior = interior && staged;
        }
        if (lane == 0) {
            boxinfo[0] = (staged ? 1 : 0) | (interior ? 2 : 0);
            boxinfo[1] = bx0; boxinfo[2] = by0; boxinfo[3] = bw; boxinfo[4] = bh;
            boxinfo[5] = pitch; boxinfo[6] = gb0;
        }
    }
    __syncthreads();
    const int flags = __builtin_amdgcn_readfirstlane(boxinfo[0]);
    const bool staged = (flags & 1) != 0, interior = (flags & 2) != 0;
    const int bx0 = __builtin_amdgcn_readfirstlane(boxinfo[1]), by0 = __builtin_amdgcn_readfirstlane(boxinfo[2]);
    const int bw = __builtin_amdgcn_readfirstlane(boxinfo[3]), bh = __builtin_amdgcn_readfirstlane(boxinfo[4]);
    const int pitch = __builtin_amdgcn_readfirstlane(boxinfo[5]), gb0 = __builtin_amdgcn_readfirstlane(boxinfo[6]);
    (void)bx0;
    if (staged && bw > 0) {
        // the box is bh rows of ndw dwords = bh*ndw float4 slots in LDS; thread t owns slots
        // t, t+256, ...  All of a thread's loads are issued before the first conversion.
        const int ndw = pitch >> 2;
        const int total = bh * ndw;
        const u8* sp0 = s.row(fr, by0) + gb0;
        if (src4) {
            constexpr int K = PV_LDS_FLOATS / 4 / 256;           // slots per thread at most
            const float inv = 1.0f / (float)ndw;
            int q256 = (int)(256.0f * inv), m256 = 256 - q256 * ndw;          // 256 = q256*ndw + m256
            if (m256 >= ndw) { m256 -= ndw; ++q256; }
            if (m256 < 0) { m256 += ndw; --q256; }
            int r = (int)((float)tid * inv), c = tid - r * ndw;
            if (c >= ndw) { c -= ndw; ++r; }
            if (c < 0) { c += ndw; --r; }
            const u32 rs32 = (u32)s.rs;
            u32 v[K];
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                const int rr = min(r, bh - 1);                    // clamped: no branch around the load
                v[kk] = *(const u32*)(sp0 + ((u32)rr * rs32 + 4u * (u32)c));   // a frame is < 4 GiB
                c += m256; r += q256;
                if (c >= ndw) { c -= ndw; ++r; }
            }
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                const int i = tid + 256 * kk;
                if (i < total) {
                    const u32 w = v[kk];
                    const float4 o = {unit255((float)(w & 255u)), unit255((float)((w >> 8) & 255u)),
                                      unit255((float)((w >> 16) & 255u)), unit255((float)(w >> 24))};
                    *(float4*)(box + 4 * i) = o;
                }
            }
        } else {
            const int rowbytes = s.w * C;
            for (int r = wave; r < bh; r += 4) {
                const u8* sp = sp0 + (int64_t)r * s.rs;
                float* lp = box + r * pitch;
                for (int i = lane; i < ndw; i += 64) {
                    u32 w = 0;
                    for (int e = 0; e < 4; ++e)
                        if (gb0 + 4 * i + e < rowbytes) w |= (u32)sp[4 * i + e] << (8 * e);
                    const float4 o = {unit255((float)(w & 255u)), unit255((float)((w >> 8) & 255u)),
                                      unit255((float)((w >> 16) & 255u)), unit255((float)(w >> 24))};
                    *(float4*)(lp + 4 * i) = o;
                }
            }
        }
    }
    __syncthreads();

    const int x = tx0 + lane;
    const bool dense = tx0 + PV_TW <= d.w &&
                       ((((uintptr_t)d.p) | (uintptr_t)d.rs | (uintptr_t)d.fs) & 3) == 0;
    if (interior) {
        // every tap of every pixel of the tile lies inside the staged box and inside the image
        const float bx = (float)min(x, d.w - 1) + 0.5f;      // partial tiles: stay inside the box
        const int rel = -by0 * pitch - gb0;
#pragma unroll
        for (int it = 0; it < PV_TH / 4; ++it) {
            const int ly = wave + 4 * it;
            const int y = ty0 + ly;
            float ix, iy;
            persp_src_fast(k, bx, (float)min(y, d.h - 1) + 0.5f, fw, fh, ix, iy);
            const float x0f = floorf(ix), y0f = floorf(iy);
            const float ww = ix - x0f, we = 1.0f - ww, wn = iy - y0f, ws = 1.0f - wn;
            const float w0 = ws * we, w1 = ws * ww, w2 = wn * we, w3 = wn * ww;
            const float msk = ((w0 + w1) + w2) + w3;
            const float* p0 = box + ((int)y0f * pitch + (int)x0f * C + rel);
            const float* p1 = p0 + pitch;
            u8* op = &outt[ly][lane * C];
#pragma unroll
            for (int j = 0; j < C; ++j) {
                float acc = p0[j] * w0;
                acc = fmaf(p0[C + j], w1, acc);
                acc = fmaf(p1[j], w2, acc);
                acc = fmaf(p1[C + j], w3, acc);
                const float o = (acc * msk) * 255.0f;
                op[j] = (u8)min((int)o, 255);
            }
        }
    } else {
#pragma unroll 1
        for (int ly = wave; ly < PV_TH; ly += 4) {
            const int y = ty0 + ly;
            float ix, iy;
            persp_src(k, (float)x + 0.5f, (float)y + 0.5f, fw, fh, ix, iy);
            const float x0f = floorf(ix), y0f = floorf(iy);
            const float ww = ix - x0f, we = 1.0f - ww, wn = iy - y0f, ws = 1.0f - wn;
            const float w4[4] = {ws * we, ws * ww, wn * we, wn * ww};
            // NaN / huge coordinates compare false everywhere below and sample nothing
            const bool sane = fabsf(ix) < 1.0e9f && fabsf(iy) < 1.0e9f;
            const int xi = sane ? (int)x0f : -4, yi = sane ? (int)y0f : -4;
            float acc[C];
            float msk = 0.0f;
#pragma unroll
            for (int j = 0; j < C; ++j) acc[j] = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int xx = xi + (q & 1), yy = yi + (q >> 1);
                const bool ok = xx >= 0 && xx < s.w && yy >= 0 && yy < s.h;
                float v[C];
#pragma unroll
                for (int j = 0; j < C; ++j) v[j] = 0.0f;
                if (ok) {
                    if (staged) {
                        const float* lp = box + (yy - by0) * pitch + (xx * C - gb0);
#pragma unroll
                        for (int j = 0; j < C; ++j) v[j] = lp[j];
                    } else {
                        const u8* sp = s.row(fr, yy) + (int64_t)xx * C;
#pragma unroll
                        for (int j = 0; j < C; ++j) v[j] = unit255((float)sp[j]);
                    }
                }
                const float m = ok ? 1.0f : 0.0f;
                if (q == 0) {
#pragma unroll
                    for (int j = 0; j < C; ++j) acc[j] = v[j] * w4[0];
                    msk = m * w4[0];
                } else {
#pragma unroll
                    for (int j = 0; j < C; ++j) acc[j] = fmaf(v[j], w4[q], acc[j]);
                    msk = fmaf(m, w4[q], msk);
                }
            }
#pragma unroll
            for (int j = 0; j < C; ++j) {
                // img*mask + (1-mask)*0, then mul(255).byte(): truncation of a value in [0, 255.0001]
                const float o = (acc[j] * msk + (1.0f - msk) * 0.0f) * 255.0f;
                outt[ly][lane * C + j] = (u8)min((int)o, 255);
            }
        }
    }
    // the tile's rows leave LDS as dwords when the destination allows, else byte by byte
    __syncthreads();
    if (dense) {
        constexpr int DW = PV_TW * C / 4;
        for (int i = tid; i < PV_TH * DW; i += 256) {
            const int r = i / DW, cdw = i % DW;
            if (ty0 + r < d.h)
                *(u32*)(d.row(fr, ty0 + r) + (int64_t)tx0 * C + 4 * cdw) = *(const u32*)&outt[r][4 * cdw];
        }
    } else {
        const int nb = (min(tx0 + PV_TW, d.w) - tx0) * C;
        for (int r = wave; r < PV_TH && ty0 + r < d.h; r += 4) {
            u8* dp = d.row(fr, ty0 + r) + (int64_t)tx0 * C;
            for (int b = lane; b < nb; b += 64) dp[b] = outt[r][b];
        }
    }
}

template <int C>
static int launch_perspective(const View& s, const View& d, const float* coeffs, int per_frame,
                              hipStream_t st) {
    const float sx = 0.5f * (float)d.w, sy = 0.5f * (float)d.h;
    for (int f0 = 0; f0 < d.n; f0 += PV_MAX_FRAMES) {
        const int nf = per_frame ? min(PV_MAX_FRAMES, d.n - f0) : d.n;
        PerspArgs a;
        a.per_frame = per_frame;
        a.frame0 = per_frame ? f0 : 0;
        for (int i = 0; i < (per_frame ? nf : 1); ++i) {
            const float* c = coeffs + (size_t)(per_frame ? f0 + i : 0) * 8;
            PerspCoef& k = a.k[i];
            k.t[0] = c[0] / sx; k.t[1] = c[1] / sx; k.t[2] = c[2] / sx;
            k.t[3] = c[3] / sy; k.t[4] = c[4] / sy; k.t[5] = c[5] / sy;
            k.c6 = c[6]; k.c7 = c[7];
        }
        const dim3 grid((d.w + PV_TW - 1) / PV_TW, (d.h + PV_TH - 1) / PV_TH, nf);
        hipLaunchKernelGGL(perspective_kernel<C>, grid, dim3(256), 0, st, s, d, a);
        IMGXF_CHECK(launch_status());
        if (!per_frame) break;
    }
    return IMGXF_OK;
}

} // namespace imgxf

using namespace imgxf;

IMGXF_API int imgxf_perspective_bilinear_u8(const imgxf_view* src, const imgxf_view* dst,
                                            const float* coeffs, int per_frame, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!coeffs) return IMGXF_ERR_NULL;
    if (per_frame != 0 && per_frame != 1) return IMGXF_ERR_ARG;
    if (!same_geometry(src, dst)) return IMGXF_ERR_SHAPE;     // torchvision keeps the size
    if (src->c == 2) return IMGXF_ERR_UNSUPPORTED;
    if (src->data == dst->data && !empty_view(dst)) return IMGXF_ERR_ARG;
    if (empty_view(dst)) return IMGXF_OK;
    if ((int64_t)((dst->h + PV_TH - 1) / PV_TH) > 65535) return IMGXF_ERR_SHAPE;
    const int nfr = per_frame ? dst->n : 1;
    for (int i = 0; i < nfr * 8; ++i)
        if (!(coeffs[i] == coeffs[i]) || coeffs[i] - coeffs[i] != 0.0f) return IMGXF_ERR_ARG;   // NaN / inf
    const View s = make_view(src), d = make_view(dst);
    hipStream_t st = (hipStream_t)stream;
    if (!per_frame && d.n > 65535) return IMGXF_ERR_SHAPE;
    switch (d.c) {
        case 1: return launch_perspective<1>(s, d, coeffs, per_frame, st);
        case 3: return launch_perspective<3>(s, d, coeffs, per_frame, st);
        case 4: return launch_perspective<4>(s, d, coeffs, per_frame, st);
    }
    return IMGXF_ERR_UNSUPPORTED;
}
