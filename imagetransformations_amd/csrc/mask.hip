// Mask stage of apply_background_change (/root/reference/transformation.py:340-341):
//   np.percentile(edges, 70)  -> 256-bin histogram (LDS-privatised, one atomic per bin per
//                                workgroup) + a one-lane order-statistic walk;
//   edges > threshold         -> 0/255 mask;
//   binary_dilation(mask, iterations=k) with the 4-connected cross and border 0, which for
//   this structuring element equals "some set pixel within L1 distance k".
#include "imgxf_common.h"
#include <algorithm>
#include <math.h>
#include <stdlib.h>

namespace imgxf {

__global__ __launch_bounds__(256) void hist_kernel(View s, u32* hist) {
    // 8 sub-histograms (lane & 7 picks one) keep same-bin LDS atomics of neighbouring lanes apart;
    // rows are read as dwords when the view allows it
    __shared__ u32 h[8][256];
    for (int i = threadIdx.x; i < 8 * 256; i += 256) (&h[0][0])[i] = 0;
    __syncthreads();
    const int f = blockIdx.y;
    u32* hs = h[threadIdx.x & 7];
    const bool vec = (s.w % 4 == 0) && ((((uintptr_t)s.p) | (uintptr_t)s.rs | (uintptr_t)s.fs) & 3) == 0;
    if (vec && (int64_t)s.h * (s.w >> 2) < 0x7fffffff) {
        const u32 rowwords = (u32)s.w >> 2, words = (u32)s.h * rowwords;              // 32-bit index arithmetic
        for (u32 t = blockIdx.x * 256u + threadIdx.x; t < words; t += gridDim.x * 256u) {
            const u32 y = t / rowwords, xw = t - y * rowwords;
            const u32 v = ((const u32*)s.row(f, (int)y))[xw];
            atomicAdd(&hs[v & 0xffu], 1u);
            atomicAdd(&hs[(v >> 8) & 0xffu], 1u);
            atomicAdd(&hs[(v >> 16) & 0xffu], 1u);
            atomicAdd(&hs[v >> 24], 1u);
        }
    } else if (vec) {
        const int rowwords = s.w >> 2;
        const int64_t words = (int64_t)s.h * rowwords;
        for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < words; t += (int64_t)gridDim.x * 256) {
            const int xw = (int)(t % rowwords), y = (int)(t / rowwords);
            const u32 v = ((const u32*)s.row(f, y))[xw];
            atomicAdd(&hs[v & 0xffu], 1u);
            atomicAdd(&hs[(v >> 8) & 0xffu], 1u);
            atomicAdd(&hs[(v >> 16) & 0xffu], 1u);
            atomicAdd(&hs[v >> 24], 1u);
        }
    } else {
        const int64_t total = (int64_t)s.h * s.w;
        for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
            const int x = (int)(t % s.w), y = (int)(t / s.w);
            atomicAdd(&hs[s.row(f, y)[x]], 1u);
        }
    }
    __syncthreads();
    u32 sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += h[k][threadIdx.x];
    if (sum) atomicAdd(&hist[f * 256 + threadIdx.x], sum);
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
    // (double)e > thr  <=>  e > floor(thr) for integer e: compare bytes against an integer limit
    const bool vec = (d.w % 4 == 0) &&
                     ((((uintptr_t)s.p) | (uintptr_t)s.rs | (uintptr_t)s.fs | (uintptr_t)d.p | (uintptr_t)d.rs | (uintptr_t)d.fs) & 3) == 0;
    if (vec) {
        const int rowwords = d.w >> 2;
        const int64_t total = (int64_t)d.n * d.h * rowwords;
        for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
            const int xw = (int)(t % rowwords);
            const int64_t r = t / rowwords;
            const int y = (int)(r % d.h), f = (int)(r / d.h);
            const double th = thr[f];
            const int lim = th < 0.0 ? -1 : (th >= 255.0 ? 255 : (int)floor(th));   // e > lim
            const u32 v = ((const u32*)s.row(f, y))[xw];
            u32 o = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) o |= ((int)((v >> (8 * b)) & 0xffu) > lim ? 0xffu : 0u) << (8 * b);
            ((u32*)d.row(f, y))[xw] = o;
        }
        return;
    }
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

// ---- binary_dilation(mask, iterations=K) with the cross element, marching form --------------
// K iterations of the 4-connected cross = "some set pixel within L1 distance K" (border 0):
//   out(y,x) = OR_{|dy|<=K} hd_{K-|dy|}(y+dy, x),   hd_r(y,x) = OR_{|dx|<=r} m(y, x+dx).
// A lane owns 16 bytes of a row and marches down the rows of its chunk: every new row yields
// hd_0..hd_K of its 4 dwords (byte funnel shifts against the neighbouring dwords; the dwords of
// the neighbouring lanes come over DPP wave shifts) and is OR-ed into the 2K+1 pending output
// rows held in registers; the oldest one is then complete.  Strips overlap by one lane on each
// side (lanes 0 and 63 only supply halo bytes).  Any non-zero input byte counts as set.
__device__ __forceinline__ u32 dpp_from_prev_lane(u32 v) {   // lane i <- lane i-1, lane 0 <- 0
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ u32 dpp_from_next_lane(u32 v) {   // lane i <- lane i+1, lane 63 <- 0
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);
}

template <int K>
__global__ __launch_bounds__(256) void dilate_march_kernel(View s, View d, int rows_per_chunk, int nstrips) {
    constexpr int NP = 2 * K + 1;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int strip = blockIdx.x * 4 + wave;
    if (strip >= nstrips) return;                                 // wave-uniform
    const int f = blockIdx.z;
    const int y0 = blockIdx.y * rows_per_chunk, y1 = min(s.h, y0 + rows_per_chunk);
    const int xb = strip * 992 + (lane - 1) * 16;                 // may be -16 or >= w: halo-only / idle lanes
    const bool inrow = xb >= 0 && xb + 16 <= s.w;
    const bool writer = inrow && lane >= 1 && lane <= 62;
    u32 acc[NP][4];
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j][q] = 0;

    // input rows y0-K .. y1+K-1; after consuming row t the output row t-K is complete
    for (int base = y0 - K; base < y1 + K; base += NP) {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int t = base + u;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (inrow && t >= 0 && t < s.h && t < y1 + K) v = *(const uint4*)(s.row(f, t) + xb);
            const u32 c[4] = {v.x, v.y, v.z, v.w};
            const u32 L = dpp_from_prev_lane(c[3]), R = dpp_from_next_lane(c[0]);
            u32 hd[K + 1][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32 p = q == 0 ? L : c[q - 1], n = q == 3 ? R : c[q + 1];
                hd[0][q] = c[q];
#pragma unroll
                for (int r = 1; r <= K; ++r)
                    hd[r][q] = hd[r - 1][q] | __builtin_amdgcn_alignbyte(n, c[q], (u32)r) | __builtin_amdgcn_alignbyte(c[q], p, (u32)(4 - r));
            }
            // pending output j (ring position) is |dy| rows away from this input row
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                // ring: output row (t - K) lives at position (u + 1) % NP after this step's store;
                // position p holds output row  t - K + ((p - (u + 1) + NP) % NP)
                const int rel = (j - (u + 1) + NP + NP) % NP;         // 0 .. 2K: output row t - K + rel
                const int dy = rel - K;                                  // input row t is dy below that output... |dy| <= K
                const int r = K - (dy < 0 ? -dy : dy);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[j][q] |= hd[r][q];
            }
            // output row t-K (ring position (u+1) % NP) has now seen all of its 2K+1 input rows
            const int done = (u + 1) % NP;
            const int yo = t - K;
            if (writer && yo >= y0 && yo < y1) {
                u32 o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32 x = acc[done][q];
                    const u32 hi = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;   // bit 7 of every non-zero byte
                    o[q] = hi | (hi - (hi >> 7));                                           // -> 0xff
                }
                *(uint4*)(d.row(f, yo) + xb) = make_uint4(o[0], o[1], o[2], o[3]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[done][q] = 0;
        }
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
    // every workgroup ends with 256 global atomics, one per bin: with ~1000 workgroups per frame those
    // same-address atomics, not the reads, set the time.  A few thousand workgroups per launch fill the chip.
    int64_t bx = ((int64_t)s.h * s.w + 256 * 16 - 1) / (256 * 16);
    const int64_t cap = std::max<int64_t>(32, std::min<int64_t>(512, 4096 / s.n));
    if (bx > cap) bx = cap;
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
    hipLaunchKernelGGL(gt_mask_kernel, dim3(grid_for((int64_t)d.n * d.h * ((d.w + 3) / 4))), dim3(256), 0, st,
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
    const View d = make_view(dst), sv = make_view(src);
    const bool no_march = knob_set(K_NO_MARCH);
    if (iterations == 3 && !no_march && d.w % 16 == 0 && d.w >= 64 && d.n <= 65535 &&
        ((((uintptr_t)sv.p) | (uintptr_t)sv.rs | (uintptr_t)sv.fs | (uintptr_t)d.p | (uintptr_t)d.rs | (uintptr_t)d.fs) & 15) == 0) {
        const int nstrips = (d.w + 991) / 992;
        int64_t nchunks = (d.h + 127) / 128;
        while (nchunks * ((nstrips + 3) / 4) * d.n < 2048 && (d.h + nchunks - 1) / nchunks > 32) nchunks *= 2;
        const int rpc = (int)((d.h + nchunks - 1) / nchunks);
        nchunks = (d.h + rpc - 1) / rpc;
        if (nchunks <= 65535) {
            hipLaunchKernelGGL((dilate_march_kernel<3>), dim3((unsigned)((nstrips + 3) / 4), (unsigned)nchunks, (unsigned)d.n),
                               dim3(256), 0, (hipStream_t)stream, sv, d, rpc, nstrips);
            return launch_status();
        }
    }
    hipLaunchKernelGGL(dilate_kernel, dim3(grid_for((int64_t)d.n * d.h * d.w)), dim3(256), 0,
                       (hipStream_t)stream, sv, d, iterations);
    return launch_status();
}
