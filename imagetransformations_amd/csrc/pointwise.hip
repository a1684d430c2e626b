// Elementwise colour maps: the bodies behind
//   Image.blend / ImageEnhance.Brightness   /root/reference/transformation.py:266-267,354
//   cv2.convertScaleAbs                      /root/reference/transformation.py:207
//   Image.convert('L')                       /root/reference/transformation.py:336
//   noise add + clip                         /root/reference/transformation.py:275-278
//   cv2.cvtColor channel permutations        /root/reference/transformation.py:206,233-235,252
//   Image.composite                          /root/reference/transformation.py:344
// All are pure HBM streams: one lane moves 16 bytes per access whenever the row base is
// 16-byte aligned, rows are walked grid-stride so a 4K batch launches >= 2048 workgroups.
#include "imgxf_common.h"
#include <string.h>

namespace imgxf {

// A lane owns one chunk of a row: 16 bytes, or 48 bytes (= 16 RGB pixels) when C == 3 so that
// the channel of every byte position is a compile-time constant (per-channel constants then
// cost nothing).  Full, 16-byte aligned chunks move as uint4; ragged row ends and unaligned
// views go byte by byte.  Ops return fp32 that v_cvt_pk_u8_f32 packs (round-half-even,
// saturating); ops with truncating semantics floor() first.
template <int C, class Op>
__global__ __launch_bounds__(256) void map_rows_kernel(View a, View b, View d, Op op) {
    constexpr int NV = (C == 3) ? 3 : 1, CB = 16 * NV;
    const int rowbytes = d.w * C;
    const int nchunks = (rowbytes + CB - 1) / CB;
    const int64_t total = (int64_t)d.n * d.h * nchunks;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int ck = (int)(t % nchunks);
        const int64_t r = t / nchunks;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const int xb = ck * CB;
        const u8* ap = a.p ? a.row(f, y) + xb : nullptr;
        const u8* bp = b.p ? b.row(f, y) + xb : nullptr;
        u8* dp = d.row(f, y) + xb;
        const int nv = min(CB, rowbytes - xb);
        u32 av[4 * NV], bv[4 * NV], ov[4 * NV];
        const bool vec = nv == CB && (((uintptr_t)dp | (uintptr_t)ap | (uintptr_t)bp) & 15) == 0;
        if (vec) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (ap) { const uint4 q = ((const uint4*)ap)[k]; av[4 * k] = q.x; av[4 * k + 1] = q.y; av[4 * k + 2] = q.z; av[4 * k + 3] = q.w; }
                if (bp) { const uint4 q = ((const uint4*)bp)[k]; bv[4 * k] = q.x; bv[4 * k + 1] = q.y; bv[4 * k + 2] = q.z; bv[4 * k + 3] = q.w; }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4 * NV; ++k) { av[k] = 0; bv[k] = 0; }
            for (int e = 0; e < nv; ++e) {
                const u32 x = ap ? ap[e] : 0, z = bp ? bp[e] : 0;
#pragma unroll
                for (int k = 0; k < 4 * NV; ++k)
                    if ((e >> 2) == k) { av[k] |= x << (8 * (e & 3)); bv[k] |= z << (8 * (e & 3)); }
            }
        }
        if (!ap) { for (int k = 0; k < 4 * NV; ++k) av[k] = 0; }
        if (!bp) { for (int k = 0; k < 4 * NV; ++k) bv[k] = 0; }
#pragma unroll
        for (int k = 0; k < 4 * NV; ++k) {
            u32 o = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pa = (float)((av[k] >> (8 * e)) & 0xffu), pb = (float)((bv[k] >> (8 * e)) & 0xffu);
                o = __builtin_amdgcn_cvt_pk_u8_f32(op(pa, pb, (4 * k + e) % C), e, o);
            }
            ov[k] = o;
        }
        if (vec) {
#pragma unroll
            for (int k = 0; k < NV; ++k) ((uint4*)dp)[k] = make_uint4(ov[4 * k], ov[4 * k + 1], ov[4 * k + 2], ov[4 * k + 3]);
        } else {
            for (int e = 0; e < nv; ++e) {
                u32 w = 0;
#pragma unroll
                for (int k = 0; k < 4 * NV; ++k) if ((e >> 2) == k) w = ov[k];
                dp[e] = (u8)(w >> (8 * (e & 3)));
            }
        }
    }
}

struct Color4f { float v[4]; };

// libImaging Blend.c: float32 in1 + alpha*(in2-in1), truncated ((UINT8) cast); outside
// 0 <= alpha <= 1 the value is clipped to [0,255] first.  floor() + the saturating pack
// reproduce both: in range the value is >= 0 so floor == truncation, out of range the pack
// saturates exactly where Blend.c clips (<= 0 -> 0, >= 255 -> 255).
struct BlendOp {
    float alpha; int const1, const2; Color4f c1, c2;
    __device__ __forceinline__ float operator()(float pa, float pb, int ch) const {
        const float i1 = const1 ? c1.v[ch] : pa;
        const float i2 = const2 ? c2.v[ch] : pb;
        const float t = i1 + alpha * (i2 - i1);      // un-contracted (library-wide -ffp-contract=off)
        return floorf(t);
    }
};

// cv2.convertScaleAbs: saturate_cast<uchar>(|alpha*p + beta|)
struct ScaleAbsOp {
    float alpha, beta;
    __device__ __forceinline__ float operator()(float pa, float, int) const {
        return fabsf(pa * alpha + beta);
    }
};

template <class Op>
static int launch_map(const View& a, const View& b, const View& d, const Op& op, hipStream_t st) {
    const int cb = d.c == 3 ? 48 : 16;
    const int64_t total = (int64_t)d.n * d.h * ((d.rowbytes() + cb - 1) / cb);
    if (total == 0) return IMGXF_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    switch (d.c) {
        case 1: hipLaunchKernelGGL((map_rows_kernel<1, Op>), dim3((unsigned)blocks), dim3(256), 0, st, a, b, d, op); break;
        case 2: hipLaunchKernelGGL((map_rows_kernel<2, Op>), dim3((unsigned)blocks), dim3(256), 0, st, a, b, d, op); break;
        case 3: hipLaunchKernelGGL((map_rows_kernel<3, Op>), dim3((unsigned)blocks), dim3(256), 0, st, a, b, d, op); break;
        default: hipLaunchKernelGGL((map_rows_kernel<4, Op>), dim3((unsigned)blocks), dim3(256), 0, st, a, b, d, op); break;
    }
    return launch_status();
}

// ---------------- RGB -> L : 16 pixels (48 B in, 16 B out) per lane ----------------
template <int C>
__global__ __launch_bounds__(256) void rgb2l_kernel(View s, View d) {
    const int ngrp = (d.w + 15) >> 4;
    const int64_t total = (int64_t)d.n * d.h * ngrp;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int g = (int)(t % ngrp);
        const int64_t r = t / ngrp;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const int x0 = g << 4;
        const int np = min(16, d.w - x0);
        const u8* sp = s.row(f, y) + x0 * C;
        u8* dp = d.row(f, y) + x0;
        u32 in[4 * C];
        const bool vec = np == 16 && ((((uintptr_t)sp) | ((uintptr_t)dp)) & 15) == 0;
        if (vec) {
#pragma unroll
            for (int b = 0; b < C; ++b) {
                const uint4 q = *(const uint4*)(sp + 16 * b);
                in[4 * b] = q.x; in[4 * b + 1] = q.y; in[4 * b + 2] = q.z; in[4 * b + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4 * C; ++k) in[k] = 0;
            for (int e = 0; e < np * C; ++e) {
                // dynamic register index avoided: build through a byte loop over a static unroll
                const u32 v = sp[e];
#pragma unroll
                for (int k = 0; k < 4 * C; ++k)
                    if ((e >> 2) == k) in[k] |= v << (8 * (e & 3));
            }
        }
        u32 out[4] = {0, 0, 0, 0};
#pragma unroll
        for (int px = 0; px < 16; ++px) {
            const int b0 = px * C;
            const u32 R = (in[b0 >> 2] >> (8 * (b0 & 3))) & 0xffu;
            const u32 G = (in[(b0 + 1) >> 2] >> (8 * ((b0 + 1) & 3))) & 0xffu;
            const u32 B = (in[(b0 + 2) >> 2] >> (8 * ((b0 + 2) & 3))) & 0xffu;
            const u32 L = (R * 19595u + G * 38470u + B * 7471u + 0x8000u) >> 16;
            out[px >> 2] |= L << (8 * (px & 3));
        }
        if (vec) {
            *(uint4*)dp = make_uint4(out[0], out[1], out[2], out[3]);
        } else {
            for (int e = 0; e < np; ++e) dp[e] = (u8)(out[e >> 2] >> (8 * (e & 3)));
        }
    }
}

// ---------------- add noise: u8 + f32 -> u8 ----------------
__global__ __launch_bounds__(256) void add_noise_kernel(View s, View nz, View d) {
    const int rowbytes = d.w * d.c;
    const int nchunks = (rowbytes + 3) >> 2;
    const int64_t total = (int64_t)d.n * d.h * nchunks;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int ck = (int)(t % nchunks);
        const int64_t r = t / nchunks;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const int xb = ck << 2;
        const int nv = min(4, rowbytes - xb);
        const u8* sp = s.row(f, y) + xb;
        const float* np_ = (const float*)nz.row(f, y) + xb;
        u8* dp = d.row(f, y) + xb;
        float z[4] = {0, 0, 0, 0};
        u32 pv = 0;
        const bool vec = nv == 4 && ((((uintptr_t)sp | (uintptr_t)dp) & 3) == 0) && (((uintptr_t)np_ & 15) == 0);
        if (vec) {
            pv = *(const u32*)sp;
            const float4 q = *(const float4*)np_;
            z[0] = q.x; z[1] = q.y; z[2] = q.z; z[3] = q.w;
        } else {
            for (int e = 0; e < nv; ++e) { pv |= (u32)sp[e] << (8 * e); z[e] = np_[e]; }
        }
        u32 o = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = __fadd_rn((float)((pv >> (8 * e)) & 0xffu), z[e]);
            v = fminf(fmaxf(v, 0.0f), 255.0f);   // np.clip; NaN noise is outside the contract
            o |= ((u32)(int)v) << (8 * e);
        }
        if (vec) *(u32*)dp = o;
        else for (int e = 0; e < nv; ++e) dp[e] = (u8)(o >> (8 * e));
    }
}

// ---------------- channel permutation / drop ----------------
struct Perm4 { int v[4]; };
__global__ __launch_bounds__(256) void permute_kernel(View s, View d, Perm4 pm) {
    const int64_t total = (int64_t)d.n * d.h * d.w;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int x = (int)(t % d.w);
        const int64_t r = t / d.w;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const u8* sp = s.row(f, y) + x * s.c;
        u8* dp = d.row(f, y) + x * d.c;
        u8 px[4];
        for (int j = 0; j < s.c; ++j) px[j] = sp[j];
        for (int j = 0; j < d.c; ++j) {
            const int k = pm.v[j];
            dp[j] = k == 0 ? px[0] : k == 1 ? px[1] : k == 2 ? px[2] : px[3];
        }
    }
}

// ---------------- composite: mask ? im1 : im2 ----------------
// Image.composite(im1, im2, mask) for the 0/255 masks the reference builds
// (/root/reference/transformation.py:343-344); any non-zero mask byte selects im1.
__device__ __forceinline__ u32 nonzero_bytes_to_ff(u32 x) {
    const u32 hi = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;   // bit 7 of every non-zero byte
    return hi | (hi - (hi >> 7));
}

// RGB fast path: a lane owns 16 pixels = 48 bytes of both images and 16 mask bytes; each mask
// dword (4 pixels) is spread over 3 image dwords with v_perm_b32 and selects with v_bfi_b32
__global__ __launch_bounds__(256) void composite_rgb16_kernel(View a, View b, View m, View d) {
    const int ngrp = d.w >> 4;
    const int64_t total = (int64_t)d.n * d.h * ngrp;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int g = (int)(t % ngrp);
        const int64_t r = t / ngrp;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const uint4 mq = *(const uint4*)(m.row(f, y) + g * 16);
        const u32 mw[4] = {mq.x, mq.y, mq.z, mq.w};
        const uint4* ap = (const uint4*)(a.row(f, y) + g * 48);
        const uint4* bp = (const uint4*)(b.row(f, y) + g * 48);
        uint4* dp = (uint4*)(d.row(f, y) + g * 48);
        u32 av[12], bv[12], ov[12];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint4 qa = ap[k], qb = bp[k];
            av[4 * k] = qa.x; av[4 * k + 1] = qa.y; av[4 * k + 2] = qa.z; av[4 * k + 3] = qa.w;
            bv[4 * k] = qb.x; bv[4 * k + 1] = qb.y; bv[4 * k + 2] = qb.z; bv[4 * k + 3] = qb.w;
        }
#pragma unroll
        for (int p4 = 0; p4 < 4; ++p4) {                       // 4 pixels -> 3 dwords
            const u32 mm = nonzero_bytes_to_ff(mw[p4]);
            const u32 s0 = __builtin_amdgcn_perm(0, mm, 0x01000000u);   // bytes of pixels 0,0,0,1
            const u32 s1 = __builtin_amdgcn_perm(0, mm, 0x02020101u);   // 1,1,2,2
            const u32 s2 = __builtin_amdgcn_perm(0, mm, 0x03030302u);   // 2,3,3,3
            ov[3 * p4] = (av[3 * p4] & s0) | (bv[3 * p4] & ~s0);
            ov[3 * p4 + 1] = (av[3 * p4 + 1] & s1) | (bv[3 * p4 + 1] & ~s1);
            ov[3 * p4 + 2] = (av[3 * p4 + 2] & s2) | (bv[3 * p4 + 2] & ~s2);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) dp[k] = make_uint4(ov[4 * k], ov[4 * k + 1], ov[4 * k + 2], ov[4 * k + 3]);
    }
}

// same, im2 = one colour (Image.new(mode, size, colour) never materialised)
__global__ __launch_bounds__(256) void composite_const_rgb16_kernel(View a, View m, View d, u32 c0, u32 c1, u32 c2) {
    const int ngrp = d.w >> 4;
    const int64_t total = (int64_t)d.n * d.h * ngrp;
    const u32 cpat[3] = {c0, c1, c2};                     // the colour as 12 interleaved bytes (period 3 dwords)
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int g = (int)(t % ngrp);
        const int64_t r = t / ngrp;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const uint4 mq = *(const uint4*)(m.row(f, y) + g * 16);
        const u32 mw[4] = {mq.x, mq.y, mq.z, mq.w};
        const uint4* ap = (const uint4*)(a.row(f, y) + g * 48);
        uint4* dp = (uint4*)(d.row(f, y) + g * 48);
        u32 av[12], ov[12];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint4 qa = ap[k];
            av[4 * k] = qa.x; av[4 * k + 1] = qa.y; av[4 * k + 2] = qa.z; av[4 * k + 3] = qa.w;
        }
#pragma unroll
        for (int p4 = 0; p4 < 4; ++p4) {
            const u32 mm = nonzero_bytes_to_ff(mw[p4]);
            const u32 s0 = __builtin_amdgcn_perm(0, mm, 0x01000000u);
            const u32 s1 = __builtin_amdgcn_perm(0, mm, 0x02020101u);
            const u32 s2 = __builtin_amdgcn_perm(0, mm, 0x03030302u);
            ov[3 * p4] = (av[3 * p4] & s0) | (cpat[0] & ~s0);
            ov[3 * p4 + 1] = (av[3 * p4 + 1] & s1) | (cpat[1] & ~s1);
            ov[3 * p4 + 2] = (av[3 * p4 + 2] & s2) | (cpat[2] & ~s2);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) dp[k] = make_uint4(ov[4 * k], ov[4 * k + 1], ov[4 * k + 2], ov[4 * k + 3]);
    }
}

__global__ __launch_bounds__(256) void composite_const_kernel(View a, View m, View d, u32 colour) {
    const int64_t total = (int64_t)d.n * d.h * d.w;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int x = (int)(t % d.w);
        const int64_t r = t / d.w;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const bool sel = m.row(f, y)[x] != 0;
        const u8* src = a.row(f, y) + x * d.c;
        u8* dp = d.row(f, y) + x * d.c;
        for (int j = 0; j < d.c; ++j) dp[j] = sel ? src[j] : (u8)(colour >> (8 * j));
    }
}

__global__ __launch_bounds__(256) void composite_kernel(View a, View b, View m, View d) {
    const int64_t total = (int64_t)d.n * d.h * d.w;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int x = (int)(t % d.w);
        const int64_t r = t / d.w;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const u8* src = (m.row(f, y)[x] ? a.row(f, y) : b.row(f, y)) + x * d.c;
        u8* dp = d.row(f, y) + x * d.c;
        for (int j = 0; j < d.c; ++j) dp[j] = src[j];
    }
}

// ---------------- ImageEnhance.Color: blend(L replicated, image, factor), fused ----------------
// (/root/reference/pipenline/cifar_image_transformations.py:102-106)
__global__ __launch_bounds__(256) void enhance_color_kernel(View s, View d, float factor) {
    const int ngrp = (d.w + 15) >> 4;
    const int64_t total = (int64_t)d.n * d.h * ngrp;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int g = (int)(t % ngrp);
        const int64_t r = t / ngrp;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const int x0 = g << 4;
        const int np = min(16, d.w - x0);
        const u8* sp = s.row(f, y) + x0 * 3;
        u8* dp = d.row(f, y) + x0 * 3;
        u32 in[12], out[12];
        const bool vec = np == 16 && ((((uintptr_t)sp) | ((uintptr_t)dp)) & 15) == 0;
        if (vec) {
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const uint4 q = ((const uint4*)sp)[b];
                in[4 * b] = q.x; in[4 * b + 1] = q.y; in[4 * b + 2] = q.z; in[4 * b + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) in[k] = 0;
            for (int e = 0; e < np * 3; ++e) {
                const u32 v = sp[e];
#pragma unroll
                for (int k = 0; k < 12; ++k) if ((e >> 2) == k) in[k] |= v << (8 * (e & 3));
            }
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) out[k] = 0;
#pragma unroll
        for (int px = 0; px < 16; ++px) {
            u32 ch[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) { const int b = px * 3 + j; ch[j] = (in[b >> 2] >> (8 * (b & 3))) & 0xffu; }
            const float L = (float)((ch[0] * 19595u + ch[1] * 38470u + ch[2] * 7471u + 0x8000u) >> 16);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int b = px * 3 + j;
                const float tv = L + factor * ((float)ch[j] - L);          // Blend.c, un-contracted
                out[b >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(floorf(tv), b & 3, out[b >> 2]);
            }
        }
        if (vec) {
#pragma unroll
            for (int b = 0; b < 3; ++b) ((uint4*)dp)[b] = make_uint4(out[4 * b], out[4 * b + 1], out[4 * b + 2], out[4 * b + 3]);
        } else {
            for (int e = 0; e < np * 3; ++e) {
                u32 w = 0;
#pragma unroll
                for (int k = 0; k < 12; ++k) if ((e >> 2) == k) w = out[k];
                dp[e] = (u8)(w >> (8 * (e & 3)));
            }
        }
    }
}

// ---------------- ImageEnhance.Contrast ----------------
// (/root/reference/pipenline/cifar_image_transformations.py:81-85)
// pass 1: per-frame sum of L (wavefront shuffle reduction, one 64-bit atomic per wave)
template <int C>
__global__ __launch_bounds__(256) void lum_sum_kernel(View s, unsigned long long* sums) {
    const int f = blockIdx.y;
    u32 part = 0;
    // 16 pixels per lane from whole 16-byte blocks when the rows allow it (the per-pixel byte loads below ran at
    // 15 % of a copy's speed); a lane's partial sum stays below 2^32: <= 2^24 pixels per lane x 255
    const bool vec = (s.w & 15) == 0 && ((((uintptr_t)s.p) | (uintptr_t)s.rs | (uintptr_t)s.fs) & 15) == 0;
    if (vec) {
        const int ngrp = s.w >> 4;
        const int64_t groups = (int64_t)s.h * ngrp;
        for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < groups; t += (int64_t)gridDim.x * 256) {
            const int g = (int)(t % ngrp), y = (int)(t / ngrp);
            const u8* sp = s.row(f, y) + (g << 4) * C;
            u32 in[4 * C];
#pragma unroll
            for (int b = 0; b < C; ++b) {
                const uint4 q = *(const uint4*)(sp + 16 * b);
                in[4 * b] = q.x; in[4 * b + 1] = q.y; in[4 * b + 2] = q.z; in[4 * b + 3] = q.w;
            }
#pragma unroll
            for (int px = 0; px < 16; ++px) {
                const int b0 = px * C;
                const u32 R = (in[b0 >> 2] >> (8 * (b0 & 3))) & 0xffu;
                if (C == 1) { part += R; continue; }
                const u32 G = (in[(b0 + 1) >> 2] >> (8 * ((b0 + 1) & 3))) & 0xffu;
                const u32 B = (in[(b0 + 2) >> 2] >> (8 * ((b0 + 2) & 3))) & 0xffu;
                part += (R * 19595u + G * 38470u + B * 7471u + 0x8000u) >> 16;
            }
        }
    } else {
        const int64_t total = (int64_t)s.h * s.w;
        for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
            const int x = (int)(t % s.w), y = (int)(t / s.w);
            const u8* p = s.row(f, y) + x * C;
            part += C == 1 ? (u32)p[0] : ((u32)p[0] * 19595u + (u32)p[1] * 38470u + (u32)p[2] * 7471u + 0x8000u) >> 16;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    // one atomic per workgroup: thousands of same-address atomics per frame serialise in the L2
    __shared__ u32 wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicAdd(&sums[f], (unsigned long long)wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

// pass 2: blend(solid int(mean + 0.5), image, factor); the mean is formed from the frame's sum
struct ContrastOp {
    float factor, mean;
    __device__ __forceinline__ float operator()(float pa, float, int) const {
        return floorf(mean + factor * (pa - mean));
    }
};
template <int C>
__global__ __launch_bounds__(256) void contrast_kernel(View s, View d, float factor, const unsigned long long* sums) {
    constexpr int NV = (C == 3) ? 3 : 1, CB = 16 * NV;
    const int rowbytes = d.w * C;
    const int nchunks = (rowbytes + CB - 1) / CB;
    const int f = blockIdx.y;
    const double cnt = (double)((int64_t)s.h * s.w);
    const float mean = (float)(int)((double)sums[f] / cnt + 0.5);        // int(ImageStat.mean[0] + 0.5)
    const int64_t total = (int64_t)d.h * nchunks;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int ck = (int)(t % nchunks), y = (int)(t / nchunks);
        const int xb = ck * CB;
        const u8* ap = s.row(f, y) + xb;
        u8* dp = d.row(f, y) + xb;
        const int nv = min(CB, rowbytes - xb);
        if (nv == CB && ((((uintptr_t)dp) | ((uintptr_t)ap)) & 15) == 0) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const uint4 q = ((const uint4*)ap)[k];
                const u32 w[4] = {q.x, q.y, q.z, q.w};
                u32 o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    u32 acc = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pa = (float)((w[i] >> (8 * e)) & 0xffu);
                        acc = __builtin_amdgcn_cvt_pk_u8_f32(floorf(mean + factor * (pa - mean)), e, acc);
                    }
                    o[i] = acc;
                }
                ((uint4*)dp)[k] = make_uint4(o[0], o[1], o[2], o[3]);
            }
        } else {
            for (int e = 0; e < nv; ++e) {
                const float tv = floorf(mean + factor * ((float)ap[e] - mean));
                dp[e] = (u8)__builtin_amdgcn_cvt_pk_u8_f32(tv, 0, 0u);
            }
        }
    }
}

// ---------------- TransformationPool noise members (cifar_image_transformations.py:39-70) ----------------
// The random draws stay on the host (NumPy's global generator, as in the reference); the device
// applies them.  MODE 0: gaussian_noise  out = trunc(clip(f64(p) + z, 0, 255))          (z float64)
//                MODE 1: shot_noise      out = trunc(clip(k / lambda * 255.0, 0, 255))   (k = Poisson draw as float64)
template <int MODE>
__global__ __launch_bounds__(256) void noise_f64_kernel(View s, View z, View d, double lambda) {
    const int rowbytes = d.w * d.c;
    const int64_t total = (int64_t)d.n * d.h * rowbytes;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int b = (int)(t % rowbytes);
        const int64_t r = t / rowbytes;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const double zv = ((const double*)z.row(f, y))[b];
        double v = MODE == 0 ? (double)(float)s.row(f, y)[b] + zv : zv / lambda * 255.0;
        v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);                     // np.clip
        d.row(f, y)[b] = (u8)(int)v;                                      // astype(np.uint8)
    }
}
// impulse_noise: pixels with mask < lo become 0, with mask > hi become 255 (all channels)
__global__ __launch_bounds__(256) void impulse_kernel(View s, View m, View d, double lo, double hi) {
    const int64_t total = (int64_t)d.n * d.h * d.w;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int x = (int)(t % d.w);
        const int64_t r = t / d.w;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const double mv = ((const double*)m.row(f, y))[x];
        const u8* sp = s.row(f, y) + x * d.c;
        u8* dp = d.row(f, y) + x * d.c;
        for (int j = 0; j < d.c; ++j) dp[j] = mv < lo ? (u8)0 : (mv > hi ? (u8)255 : sp[j]);
    }
}

static inline unsigned grid_for(int64_t total) {
    int64_t blocks = (total + 255) / 256;
    return (unsigned)(blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks));
}

} // namespace imgxf

using namespace imgxf;

IMGXF_API int imgxf_rgb2l_u8(const imgxf_view* src, const imgxf_view* dst, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!same_nhw(src, dst) || dst->c != 1 || (src->c != 3 && src->c != 4)) return IMGXF_ERR_SHAPE;
    if (empty_view(dst)) return IMGXF_OK;
    const View s = make_view(src), d = make_view(dst);
    const int64_t total = (int64_t)d.n * d.h * ((d.w + 15) >> 4);
    if (src->c == 3)
        hipLaunchKernelGGL((rgb2l_kernel<3>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, s, d);
    else
        hipLaunchKernelGGL((rgb2l_kernel<4>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, s, d);
    return launch_status();
}

IMGXF_API int imgxf_scale_abs_u8(const imgxf_view* src, const imgxf_view* dst, float alpha,
                                 float beta, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!same_geometry(src, dst)) return IMGXF_ERR_SHAPE;
    View none; memset(&none, 0, sizeof(none));
    ScaleAbsOp op{alpha, beta};
    return launch_map(make_view(src), none, make_view(dst), op, (hipStream_t)stream);
}

IMGXF_API int imgxf_blend_u8(const imgxf_view* im1, const uint8_t* color1, const imgxf_view* im2,
                             const uint8_t* color2, const imgxf_view* dst, float alpha,
                             void* stream) {
    IMGXF_CHECK(check_view(dst));
    if ((!im1 && !color1) || (!im2 && !color2)) return IMGXF_ERR_NULL;
    View a, b; memset(&a, 0, sizeof(a)); memset(&b, 0, sizeof(b));
    BlendOp op; memset(&op, 0, sizeof(op));
    op.alpha = alpha;
    if (im1) {
        IMGXF_CHECK(check_view(im1));
        if (!same_geometry(im1, dst)) return IMGXF_ERR_SHAPE;
        a = make_view(im1);
    } else {
        op.const1 = 1;
        for (int j = 0; j < dst->c; ++j) op.c1.v[j] = (float)color1[j];
    }
    if (im2) {
        IMGXF_CHECK(check_view(im2));
        if (!same_geometry(im2, dst)) return IMGXF_ERR_SHAPE;
        b = make_view(im2);
    } else {
        op.const2 = 1;
        for (int j = 0; j < dst->c; ++j) op.c2.v[j] = (float)color2[j];
    }
    return launch_map(a, b, make_view(dst), op, (hipStream_t)stream);
}

IMGXF_API int imgxf_add_noise_u8(const imgxf_view* src, const imgxf_view* noise_f32,
                                 const imgxf_view* dst, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    IMGXF_CHECK(check_view(noise_f32, 4));
    if (!same_geometry(src, dst) || !same_geometry(src, noise_f32)) return IMGXF_ERR_SHAPE;
    if (((uintptr_t)noise_f32->data & 3) || (noise_f32->row_stride & 3) || (noise_f32->frame_stride & 3))
        return IMGXF_ERR_ARG;
    if (empty_view(dst)) return IMGXF_OK;
    const View d = make_view(dst);
    const int64_t total = (int64_t)d.n * d.h * ((d.rowbytes() + 3) >> 2);
    hipLaunchKernelGGL(add_noise_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       make_view(src), make_view(noise_f32), d);
    return launch_status();
}

// 3 -> 3 channel permutations (cv2.cvtColor RGB2BGR / BGR2RGB, transformation.py:233,252) on 16-byte chunks: a thread
// moves 16 pixels = 48 bytes as three uint4; inside every 12-byte group of 4 pixels output dword j collects its four
// bytes from at most three input dwords with two v_perm_b32 (selectors built once per launch on the host).
struct PermSel { u32 s1[3], s2[3]; };
__global__ __launch_bounds__(256) void permute_rgb16_kernel(View s, View d, PermSel ps) {
    const int chunks = (d.w * 3) / 48;                                      // host: row bytes % 48 == 0
    const int64_t total = (int64_t)d.n * d.h * chunks;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int ck = (int)(t % chunks);
        const int64_t r = t / chunks;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const uint4* sp = (const uint4*)(s.row(f, y) + 48 * ck);
        uint4* dp = (uint4*)(d.row(f, y) + 48 * ck);
        const uint4 i0 = sp[0], i1 = sp[1], i2 = sp[2];
        const u32 in[12] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y, i2.z, i2.w};
        u32 o[12];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const u32 t1 = __builtin_amdgcn_perm(in[3 * g + 1], in[3 * g], ps.s1[j]);      // bytes that live in dwords 0 / 1 of the group
                o[3 * g + j] = __builtin_amdgcn_perm(in[3 * g + 2], t1, ps.s2[j]);             // ... and those of dword 2
            }
        dp[0] = make_uint4(o[0], o[1], o[2], o[3]); dp[1] = make_uint4(o[4], o[5], o[6], o[7]); dp[2] = make_uint4(o[8], o[9], o[10], o[11]);
    }
}

IMGXF_API int imgxf_permute_u8(const imgxf_view* src, const imgxf_view* dst, const int32_t* perm,
                               void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!perm) return IMGXF_ERR_NULL;
    if (!same_nhw(src, dst)) return IMGXF_ERR_SHAPE;
    Perm4 pm; memset(&pm, 0, sizeof(pm));
    for (int j = 0; j < dst->c; ++j) {
        if (perm[j] < 0 || perm[j] >= src->c) return IMGXF_ERR_ARG;
        pm.v[j] = perm[j];
    }
    if (empty_view(dst)) return IMGXF_OK;
    const View d = make_view(dst);
    {
        const View sv = make_view(src);
        auto al16 = [](const View& v) { return ((((uintptr_t)v.p) | (uintptr_t)v.rs | (uintptr_t)v.fs) & 15) == 0; };
        if (src->c == 3 && dst->c == 3 && (d.w * 3) % 48 == 0 && al16(sv) && al16(d) && !knob_set(K_NO_FAST_LEFTOVERS)) {
            // output byte 4 j + e of a 12-byte group comes from input byte q = 3 ((4 j + e) / 3) + perm[(4 j + e) % 3] of the
            // group's three dwords: t1 = perm(dword 1, dword 0) gathers the bytes that live there, perm(dword 2, t1) the rest
            PermSel ps; memset(&ps, 0, sizeof(ps));
            for (int j = 0; j < 3; ++j) {
                u32 s1 = 0, s2 = 0;
                for (int e = 0; e < 4; ++e) {
                    const int ob = 4 * j + e, q = 3 * (ob / 3) + pm.v[ob % 3], dw = q >> 2, by = q & 3;
                    const u32 sel1 = dw == 0 ? (u32)by : (dw == 1 ? (u32)(4 + by) : 0x0cu);       // 0x0c: constant zero
                    const u32 sel2 = dw == 2 ? (u32)(4 + by) : (u32)e;                           // byte `by` of dword 2, else keep t1's byte e
                    s1 |= sel1 << (8 * e); s2 |= sel2 << (8 * e);
                }
                ps.s1[j] = s1; ps.s2[j] = s2;
            }
            hipLaunchKernelGGL(permute_rgb16_kernel, dim3(grid_for((int64_t)d.n * d.h * ((d.w * 3) / 48))), dim3(256), 0,
                               (hipStream_t)stream, sv, d, ps);
            return launch_status();
        }
    }
    hipLaunchKernelGGL(permute_kernel, dim3(grid_for((int64_t)d.n * d.h * d.w)), dim3(256), 0,
                       (hipStream_t)stream, make_view(src), d, pm);
    return launch_status();
}

IMGXF_API int imgxf_composite_u8(const imgxf_view* im1, const imgxf_view* im2,
                                 const imgxf_view* mask, const imgxf_view* dst, void* stream) {
    IMGXF_CHECK(check_view(im1));
    IMGXF_CHECK(check_view(im2));
    IMGXF_CHECK(check_view(mask));
    IMGXF_CHECK(check_view(dst));
    if (!same_geometry(im1, dst) || !same_geometry(im2, dst) || !same_nhw(mask, dst) || mask->c != 1)
        return IMGXF_ERR_SHAPE;
    if (empty_view(dst)) return IMGXF_OK;
    const View d = make_view(dst);
    const View va = make_view(im1), vb = make_view(im2), vm = make_view(mask);
    auto al16 = [](const View& v) { return ((((uintptr_t)v.p) | (uintptr_t)v.rs | (uintptr_t)v.fs) & 15) == 0; };
    if (d.c == 3 && d.w % 16 == 0 && al16(va) && al16(vb) && al16(vm) && al16(d)) {
        hipLaunchKernelGGL(composite_rgb16_kernel, dim3(grid_for((int64_t)d.n * d.h * (d.w >> 4))), dim3(256), 0,
                           (hipStream_t)stream, va, vb, vm, d);
        return launch_status();
    }
    hipLaunchKernelGGL(composite_kernel, dim3(grid_for((int64_t)d.n * d.h * d.w)), dim3(256), 0,
                       (hipStream_t)stream, va, vb, vm, d);
    return launch_status();
}

IMGXF_API int imgxf_enhance_color_u8(const imgxf_view* src, const imgxf_view* dst, float factor, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!same_geometry(src, dst) || src->c != 3) return IMGXF_ERR_SHAPE;
    if (empty_view(dst)) return IMGXF_OK;
    const View d = make_view(dst);
    const int64_t total = (int64_t)d.n * d.h * ((d.w + 15) >> 4);
    hipLaunchKernelGGL(enhance_color_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, make_view(src), d, factor);
    return launch_status();
}

IMGXF_API int imgxf_enhance_contrast_u8(const imgxf_view* src, const imgxf_view* dst, float factor,
                                        uint64_t* sums, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!sums) return IMGXF_ERR_NULL;
    if (!same_geometry(src, dst) || (src->c != 3 && src->c != 1)) return IMGXF_ERR_SHAPE;
    if (empty_view(dst)) return IMGXF_OK;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(sums, 0, (size_t)src->n * sizeof(uint64_t), st);
    if (e != hipSuccess) return (int)e;
    const View s = make_view(src), d = make_view(dst);
    int64_t bx = ((int64_t)s.h * s.w + 256 * 16 - 1) / (256 * 16);
    if (bx > 256) bx = 256;
    if (bx < 1) bx = 1;
    dim3 g1((unsigned)bx, (unsigned)s.n);
    const int cb = s.c == 3 ? 48 : 16;
    int64_t b2 = ((int64_t)d.h * ((d.rowbytes() + cb - 1) / cb) + 255) / 256;
    if (b2 > 2048) b2 = 2048;
    dim3 g2((unsigned)b2, (unsigned)s.n);
    if (s.c == 3) {
        hipLaunchKernelGGL((lum_sum_kernel<3>), g1, dim3(256), 0, st, s, (unsigned long long*)sums);
        hipLaunchKernelGGL((contrast_kernel<3>), g2, dim3(256), 0, st, s, d, factor, (const unsigned long long*)sums);
    } else {
        hipLaunchKernelGGL((lum_sum_kernel<1>), g1, dim3(256), 0, st, s, (unsigned long long*)sums);
        hipLaunchKernelGGL((contrast_kernel<1>), g2, dim3(256), 0, st, s, d, factor, (const unsigned long long*)sums);
    }
    return launch_status();
}

IMGXF_API int imgxf_add_noise_f64_u8(const imgxf_view* src, const imgxf_view* noise_f64, const imgxf_view* dst, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    IMGXF_CHECK(check_view(noise_f64, 8));
    if (!same_geometry(src, dst) || !same_geometry(src, noise_f64)) return IMGXF_ERR_SHAPE;
    if (((uintptr_t)noise_f64->data & 7) || (noise_f64->row_stride & 7) || (noise_f64->frame_stride & 7)) return IMGXF_ERR_ARG;
    if (empty_view(dst)) return IMGXF_OK;
    const View d = make_view(dst);
    hipLaunchKernelGGL((noise_f64_kernel<0>), dim3(grid_for((int64_t)d.n * d.h * d.rowbytes())), dim3(256), 0,
                       (hipStream_t)stream, make_view(src), make_view(noise_f64), d, 1.0);
    return launch_status();
}

IMGXF_API int imgxf_shot_noise_u8(const imgxf_view* counts_f64, double lambda, const imgxf_view* dst, void* stream) {
    IMGXF_CHECK(check_view(dst));
    IMGXF_CHECK(check_view(counts_f64, 8));
    if (!same_geometry(dst, counts_f64)) return IMGXF_ERR_SHAPE;
    if (!(lambda > 0.0)) return IMGXF_ERR_ARG;
    if (((uintptr_t)counts_f64->data & 7) || (counts_f64->row_stride & 7) || (counts_f64->frame_stride & 7)) return IMGXF_ERR_ARG;
    if (empty_view(dst)) return IMGXF_OK;
    const View d = make_view(dst);
    hipLaunchKernelGGL((noise_f64_kernel<1>), dim3(grid_for((int64_t)d.n * d.h * d.rowbytes())), dim3(256), 0,
                       (hipStream_t)stream, d, make_view(counts_f64), d, lambda);
    return launch_status();
}

IMGXF_API int imgxf_impulse_noise_u8(const imgxf_view* src, const imgxf_view* mask_f64, double lo, double hi,
                                     const imgxf_view* dst, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    IMGXF_CHECK(check_view(mask_f64, 8));
    if (!same_geometry(src, dst) || !same_nhw(src, mask_f64) || mask_f64->c != 1) return IMGXF_ERR_SHAPE;
    if (((uintptr_t)mask_f64->data & 7) || (mask_f64->row_stride & 7) || (mask_f64->frame_stride & 7)) return IMGXF_ERR_ARG;
    if (empty_view(dst)) return IMGXF_OK;
    const View d = make_view(dst);
    hipLaunchKernelGGL(impulse_kernel, dim3(grid_for((int64_t)d.n * d.h * d.w)), dim3(256), 0, (hipStream_t)stream,
                       make_view(src), make_view(mask_f64), d, lo, hi);
    return launch_status();
}

IMGXF_API int imgxf_composite_const_u8(const imgxf_view* im1, const uint8_t* colour, const imgxf_view* mask,
                                       const imgxf_view* dst, void* stream) {
    IMGXF_CHECK(check_view(im1));
    IMGXF_CHECK(check_view(mask));
    IMGXF_CHECK(check_view(dst));
    if (!colour) return IMGXF_ERR_NULL;
    if (!same_geometry(im1, dst) || !same_nhw(mask, dst) || mask->c != 1) return IMGXF_ERR_SHAPE;
    if (empty_view(dst)) return IMGXF_OK;
    const View d = make_view(dst), va = make_view(im1), vm = make_view(mask);
    auto al16 = [](const View& v) { return ((((uintptr_t)v.p) | (uintptr_t)v.rs | (uintptr_t)v.fs) & 15) == 0; };
    if (d.c == 3 && d.w % 16 == 0 && al16(va) && al16(vm) && al16(d)) {
        uint8_t pat[12];
        for (int i = 0; i < 12; ++i) pat[i] = colour[i % 3];
        u32 c[3];
        memcpy(c, pat, 12);
        hipLaunchKernelGGL(composite_const_rgb16_kernel, dim3(grid_for((int64_t)d.n * d.h * (d.w >> 4))), dim3(256), 0,
                           (hipStream_t)stream, va, vm, d, c[0], c[1], c[2]);
        return launch_status();
    }
    u32 packed = 0;
    for (int j = 0; j < d.c; ++j) packed |= (u32)colour[j] << (8 * j);
    hipLaunchKernelGGL(composite_const_kernel, dim3(grid_for((int64_t)d.n * d.h * d.w)), dim3(256), 0,
                       (hipStream_t)stream, va, vm, d, packed);
    return launch_status();
}
