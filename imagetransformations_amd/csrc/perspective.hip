// Perspective warp behind apply_perspective_warp (/root/reference/fall_2025/transformations_code:54-66):
//   ToTensor -> torchvision RandomPerspective(p=1) on a float tensor -> ToPILImage
// which for drawn coefficients is torchvision's _perspective_grid + grid_sample(bilinear, zeros,
// align_corners=False) of [image/255 | ones] in fp32, img*mask + (1-mask)*fill with fill = 0, then
// mul(255).byte().  Every fp32 operation below is written in the order (and with the fused
// multiply-adds) that the torch CPU build evaluates them in, so the bytes match; the library is
// built with -ffp-contract=off, so only the fmaf calls fuse.
//
// A 64x16 output tile maps to a convex quadrilateral of the source; its bounding box (from the
// four tile corners, plus a margin for rounding) is staged in LDS as packed u8 rows and the four
// taps are gathered from there.  Tiles whose box does not fit (extreme coefficients) gather
// from global memory instead.
#include "imgxf_common.h"

namespace imgxf {

constexpr int PV_TW = 64, PV_TH = 16;          // output tile
constexpr int PV_LDS_BYTES = 40 * 1024;        // staged source box
constexpr int PV_MAX_FRAMES = 48;              // coefficient sets per launch (kernarg budget)

struct PerspCoef { float t[6]; float c6, c7; };   // t = c[0..5] / (0.5*ow | 0.5*oh)
struct PerspArgs {
    PerspCoef k[PV_MAX_FRAMES];
    int per_frame;      // 0: k[0] for every frame
    int frame0;         // first frame of this launch
};

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

template <int C>
__global__ __launch_bounds__(256) void perspective_kernel(View s, View d, PerspArgs a) {
    __shared__ float unit[256];                 // v / 255 correctly rounded, as Tensor.div(255)
    __shared__ __attribute__((aligned(16))) u8 box[PV_LDS_BYTES];
    const int tid = threadIdx.x;
    unit[tid] = (float)tid / 255.0f;
    const int f = blockIdx.z;
    const PerspCoef& k = a.k[a.per_frame ? f : 0];
    const int fr = a.frame0 + f;
    const int tx0 = blockIdx.x * PV_TW, ty0 = blockIdx.y * PV_TH;
    const float fw = (float)s.w, fh = (float)s.h;

    // source bounding box of the tile from its four corner pixels (uniform across the block)
    const int tx1 = min(tx0 + PV_TW, d.w) - 1, ty1 = min(ty0 + PV_TH, d.h) - 1;
    float lox = 3.0e9f, hix = -3.0e9f, loy = 3.0e9f, hiy = -3.0e9f;
    bool finite = true;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float ix, iy;
        persp_src(k, (float)((q & 1) ? tx1 : tx0) + 0.5f, (float)((q & 2) ? ty1 : ty0) + 0.5f, fw, fh, ix, iy);
        finite = finite && (fabsf(ix) < 1.0e9f) && (fabsf(iy) < 1.0e9f);
        lox = fminf(lox, ix); hix = fmaxf(hix, ix);
        loy = fminf(loy, iy); hiy = fmaxf(hiy, iy);
    }
    // the denominator must keep one sign over the tile for the quadrilateral argument to hold
    const float d00 = fmaf((float)ty0 + 0.5f, k.c7, ((float)tx0 + 0.5f) * k.c6) + 1.0f;
    const float d10 = fmaf((float)ty0 + 0.5f, k.c7, ((float)tx1 + 0.5f) * k.c6) + 1.0f;
    const float d01 = fmaf((float)ty1 + 0.5f, k.c7, ((float)tx0 + 0.5f) * k.c6) + 1.0f;
    const float d11 = fmaf((float)ty1 + 0.5f, k.c7, ((float)tx1 + 0.5f) * k.c6) + 1.0f;
    const float dmin = fminf(fminf(d00, d10), fminf(d01, d11));
    bool staged = finite && dmin > 1.0e-3f;
    int bx0 = 0, by0 = 0, bw = 0, bh = 0, pitch = 0;
    if (staged) {
        // +-2 px of margin: one for the right/bottom tap, one for rounding of the corner estimates
        bx0 = max((int)floorf(lox) - 2, 0);
        by0 = max((int)floorf(loy) - 2, 0);
        const int bx1 = min((int)floorf(hix) + 3, s.w - 1);
        const int by1 = min((int)floorf(hiy) + 3, s.h - 1);
        bw = bx1 - bx0 + 1; bh = by1 - by0 + 1;
        if (bw <= 0 || bh <= 0) { bw = bh = 0; }           // the tile sees no source pixel at all
        pitch = (bw * C + 3) & ~3;
        staged = (int64_t)pitch * bh <= PV_LDS_BYTES;
    }
    if (staged && bw > 0) {
        const int nb = bw * C;
        for (int r = tid >> 6; r < bh; r += 4) {
            const u8* sp = s.row(fr, by0 + r) + (int64_t)bx0 * C;
            u8* lp = box + r * pitch;
            for (int b = tid & 63; b < nb; b += 64) lp[b] = sp[b];
        }
    }
    __syncthreads();

    const int lx = tid & 63;
    const int x = tx0 + lx;
#pragma unroll 1
    for (int ly = tid >> 6; ly < PV_TH; ly += 4) {
        const int y = ty0 + ly;
        if (x >= d.w || y >= d.h) continue;
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
            if (ok) {
                if (staged) {
                    const u8* lp = box + (yy - by0) * pitch + (xx - bx0) * C;
#pragma unroll
                    for (int j = 0; j < C; ++j) v[j] = unit[lp[j]];
                } else {
                    const u8* sp = s.row(fr, yy) + (int64_t)xx * C;
#pragma unroll
                    for (int j = 0; j < C; ++j) v[j] = unit[sp[j]];
                }
            } else {
#pragma unroll
                for (int j = 0; j < C; ++j) v[j] = 0.0f;
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
        u8* dp = d.row(fr, y) + (int64_t)x * C;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            // img*mask + (1-mask)*0, then mul(255).byte(): truncation of a value in [0, 255.0001]
            const float o = (acc[j] * msk + (1.0f - msk) * 0.0f) * 255.0f;
            dp[j] = (u8)min((int)o, 255);
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
