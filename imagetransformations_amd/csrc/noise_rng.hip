// Opt-in DEVICE random numbers for the noise transform (round 3).
//
// apply_gaussian_noise (/root/reference/transformation.py:272-281) draws np.random.normal(0, sigma * 255, shape) from
// NumPy's global MT19937 stream on the host — for a 375 x 500 image that draw is ~8 of the 8.6 ms the batched driver
// spends per image.  The drop-in default keeps that stream (bit-exact parity).  With IMGXF_NOISE_RNG=device the normals
// are generated inside the add kernel: Philox4x32-10 (Salmon et al., SC'11; counter = element index / 4, key = seed)
// -> Box-Muller in fp32 -> noise.astype(f32); then exactly the reference's arithmetic clip(f32(p) + noise, 0, 255)
// -> uint8 (truncation).  A DIFFERENT random stream: distribution-level parity only, tested on
// moments, a Kolmogorov-Smirnov bound and the clipping behaviour; the uint32 stream itself is pinned by the Random123
// known-answer vectors.  Counter-based, so a frame's noise does not depend on the batch it is launched in.
#include "imgxf_common.h"

namespace imgxf {

struct Philox4 { u32 x, y, z, w; };

__host__ __device__ __forceinline__ Philox4 philox4x32_10(u32 c0, u32 c1, u32 c2, u32 c3, u32 k0, u32 k1) {
    constexpr u32 M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const u32 n0 = (u32)(p1 >> 32) ^ c1 ^ k0, n1 = (u32)p1, n2 = (u32)(p0 >> 32) ^ c3 ^ k1, n3 = (u32)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return Philox4{c0, c1, c2, c3};
}

// two u32 -> two standard normals (Box-Muller, fp32): u1 in (0, 1], u2 in [0, 1)
__device__ __forceinline__ void box_muller(u32 a, u32 b, float& n0, float& n1) {
    const float u1 = ((float)(a >> 8) + 1.0f) * 5.9604644775390625e-08f;      // (0, 1]: the log is finite
    const float u2 = (float)(b >> 8) * 5.9604644775390625e-08f;
    const float r = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    n0 = r * cs; n1 = r * sn;
}

// element e (0 .. n*h*rowbytes) of the batch gets normal number e: counter = e / 4 (+ offset), lane e % 4 of the block
__global__ __launch_bounds__(256) void add_noise_philox_kernel(View s, View d, float scale, u32 k0, u32 k1, uint64_t offset4) {
    const int rowbytes = d.w * d.c;
    const int nchunks = (rowbytes + 3) >> 2;
    const int64_t total = (int64_t)d.n * d.h * nchunks;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int ck = (int)(t % nchunks);
        const int64_t r = t / nchunks;
        const int y = (int)(r % d.h), f = (int)(r / d.h);
        const int xb = ck << 2, nv = min(4, rowbytes - xb);
        const u8* sp = s.row(f, y) + xb;
        u8* dp = d.row(f, y) + xb;
        const uint64_t ctr = (uint64_t)t + offset4;
        const Philox4 q = philox4x32_10((u32)ctr, (u32)(ctr >> 32), 0u, 0u, k0, k1);
        float z[4];
        box_muller(q.x, q.y, z[0], z[1]);
        box_muller(q.z, q.w, z[2], z[3]);
        const bool vec = nv == 4 && ((((uintptr_t)sp | (uintptr_t)dp) & 3) == 0);
        u32 pv = 0;
        if (vec) pv = *(const u32*)sp;
        else for (int e = 0; e < nv; ++e) pv |= (u32)sp[e] << (8 * e);
        u32 o = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = __fadd_rn((float)((pv >> (8 * e)) & 0xffu), __fmul_rn(z[e], scale));     // f32(p) + noise
            v = fminf(fmaxf(v, 0.0f), 255.0f);                                                  // np.clip
            o |= ((u32)(int)v) << (8 * e);                                                      // astype(uint8)
        }
        if (vec) *(u32*)dp = o;
        else for (int e = 0; e < nv; ++e) dp[e] = (u8)(o >> (8 * e));
    }
}

__global__ __launch_bounds__(256) void philox_u32_kernel(u32* out, int64_t nblocks4, u32 k0, u32 k1, uint64_t offset4) {
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < nblocks4; t += (int64_t)gridDim.x * 256) {
        const uint64_t ctr = (uint64_t)t + offset4;
        const Philox4 q = philox4x32_10((u32)ctr, (u32)(ctr >> 32), 0u, 0u, k0, k1);
        out[4 * t + 0] = q.x; out[4 * t + 1] = q.y; out[4 * t + 2] = q.z; out[4 * t + 3] = q.w;
    }
}

static unsigned noise_grid(int64_t total) {
    const int64_t g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

} // namespace imgxf

namespace imgxf {
// MT19937's state sequence (numpy/random/src/mt19937/mt19937.c mt19937_gen restated): block 0 = the generator's current
// key, block b = the 624 state words after b regenerations.  The recurrence runs over the block index, so ONE workgroup
// walks it; inside a block, words 0..226 depend on the old block only, 227..453 on the new words 0..226, 454..622 on
// 227..395, and 623 on the new words 396 and 0: four barriers per block.  The RAW words are written (tempering, the
// doubles and the polar method are data parallel and run afterwards, imagetransformations_amd/numpy_stream.py): the
// generator's state at any stream position is then a slice of the output.
__global__ __launch_bounds__(256) void mt19937_blocks_kernel(const u32* __restrict__ key, u32* __restrict__ out, long long nblocks) {
    __shared__ u32 st[2][624];
    const int tid = threadIdx.x;
    for (int i = tid; i < 624; i += 256) { const u32 v = key[i]; st[0][i] = v; out[i] = v; }
    __syncthreads();
    auto twist = [](u32 u, u32 v) { return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u); };
    // Inside a block, thread t computes new[t], new[227 + t] (from its own new[t]) and new[454 + t] (from its own new[227 + t]):
    // the distance of the recurrence is 227, so the chain stays in the thread's registers.  Only new[623] needs other
    // threads' words (new[396] and new[0]); thread 0 recomputes new[396] from the old block itself.  ONE barrier per block —
    // the next block reads everybody's words — and it waits for the LDS traffic only (__syncthreads() would also drain the
    // global stores: 535 ns per block with four of those).
    for (long long b = 1; b <= nblocks; ++b) {
        const u32* o = st[(b - 1) & 1];
        u32* n = st[b & 1];
        u32* dst = out + b * 624;
        if (tid < 227) {
            const u32 a = o[tid + 397] ^ twist(o[tid], o[tid + 1]);
            const u32 c = a ^ twist(o[227 + tid], o[228 + tid]);
            n[tid] = a; n[227 + tid] = c;
            dst[tid] = a; dst[227 + tid] = c;
            if (tid < 169) {
                const u32 e = c ^ twist(o[454 + tid], o[455 + tid]);
                n[454 + tid] = e; dst[454 + tid] = e;
            }
            if (tid == 0) {
                const u32 n169 = o[566] ^ twist(o[169], o[170]);
                const u32 n396 = n169 ^ twist(o[396], o[397]);
                const u32 z = n396 ^ twist(o[623], a);
                n[623] = z; dst[623] = z;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}
} // namespace imgxf

namespace imgxf {
// Jump-ahead (Haramoto et al.): out_keys[w] = F^((w+1) J) base = g_(w+1)(F) base, where F is the generator's one-word step on its
// canonical state (S[t] .. S[t+623]) and g_m = x^(m J) mod (F's characteristic polynomial), evaluated by Horner's rule:
// r = 0; for i = deg .. 0: r = F r (+ s if g_i).  One wave per jump (no barriers: a wave's LDS operations execute in order): lane 0
// makes the step on a circular buffer — one new word per step — and the 64 lanes add s where the coefficient is set (half of
// the 19937 steps): 9 ms per jump, all jumps of a request in parallel.  coefs: [jump][2496] bytes, bit i of the little-endian bit
// string = coefficient i (tools/make_mt_jump.py).
__global__ __launch_bounds__(64) void mt19937_jump_kernel(const u32* __restrict__ base, u32* __restrict__ out_keys, const u8* __restrict__ coefs) {
    __shared__ u32 s[624], r[624];
    __shared__ u8 g[2496];
    const int lane = threadIdx.x;
    const u8* coef = coefs + (size_t)blockIdx.x * 2496;              // workgroup w: out_keys[w] = g_(w+1)(F) base = F^((w+1) J) base
    for (int i = lane; i < 2496; i += 64) g[i] = coef[i];
    for (int i = lane; i < 624; i += 64) { s[i] = base[i]; r[i] = 0u; }
    __syncthreads();
    int h = 0;                                                       // logical word j of r lives at r[(h + j) % 624]
    for (int i = 19936; i >= 0; --i) {
        if (lane == 0) {
            const u32 a = r[h], b = r[h + 1 < 624 ? h + 1 : 0], c = r[h + 397 < 624 ? h + 397 : h + 397 - 624];
            const u32 y = (a & 0x80000000u) | (b & 0x7fffffffu);
            r[h] = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        h = h + 1 < 624 ? h + 1 : 0;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if ((g[i >> 3] >> (i & 7)) & 1) {
            for (int j = lane; j < 624; j += 64) { const int p = h + j < 624 ? h + j : h + j - 624; r[p] ^= s[j]; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    for (int j = lane; j < 624; j += 64) out_keys[(size_t)blockIdx.x * 624 + j] = r[h + j < 624 ? h + j : h + j - 624];
}

// The state sequence in STRETCHES of `bps` blocks: workgroup m starts from keys[m] (= block m * bps of the stream) and writes
// blocks m * bps .. min((m + 1) * bps, total) - 1 (the loop of mt19937_blocks_kernel).
__global__ __launch_bounds__(256) void mt19937_stretch_kernel(const u32* __restrict__ keys, u32* __restrict__ out, long long bps, long long total) {
    __shared__ u32 st[2][624];
    const int tid = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * bps, b1 = b0 + bps < total ? b0 + bps : total;
    if (b0 >= total) return;
    const u32* key = keys + (size_t)blockIdx.x * 624;
    // (a jumped key's word 0 is state only in its top bit: the stream's word there is written by the stretch before, below)
    for (int i = tid; i < 624; i += 256) { const u32 v = key[i]; st[0][i] = v; if (i || blockIdx.x == 0) out[b0 * 624 + i] = v; }
    __syncthreads();
    auto twist = [](u32 u, u32 v) { return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u); };
    for (long long b = b0 + 1; b < b1; ++b) {
        const u32* o = st[(b - b0 - 1) & 1];
        u32* n = st[(b - b0) & 1];
        u32* dst = out + b * 624;
        if (tid < 227) {
            const u32 a = o[tid + 397] ^ twist(o[tid], o[tid + 1]);
            const u32 c = a ^ twist(o[227 + tid], o[228 + tid]);
            n[tid] = a; n[227 + tid] = c;
            dst[tid] = a; dst[227 + tid] = c;
            if (tid < 169) {
                const u32 e = c ^ twist(o[454 + tid], o[455 + tid]);
                n[454 + tid] = e; dst[454 + tid] = e;
            }
            if (tid == 0) {
                const u32 n169 = o[566] ^ twist(o[169], o[170]);
                const u32 n396 = n169 ^ twist(o[396], o[397]);
                const u32 z = n396 ^ twist(o[623], a);
                n[623] = z; dst[623] = z;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (b1 < total && tid == 0) {                                      // word 0 of the next stretch's first block
        const u32* o = st[(b1 - b0 - 1) & 1];
        out[b1 * 624] = o[397] ^ twist(o[0], o[1]);
    }
}
} // namespace imgxf

namespace imgxf {
// NumPy's legacy_gauss over the word stream, data parallel (imagetransformations_amd/numpy_stream.py states the algorithm):
// group g = words 4 g .. 4 g + 3 of the tempered stream -> (x1, x2, r2); accepted iff 0 < r2 < 1.
__device__ __forceinline__ u32 mt_temper(u32 y) {
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
__device__ __forceinline__ void np_group(const u32* __restrict__ w, double& x1, double& x2, double& r2) {
    const u32 a = mt_temper(w[0]) >> 5, b = mt_temper(w[1]) >> 6, c = mt_temper(w[2]) >> 5, d = mt_temper(w[3]) >> 6;
    const double u1 = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;          // legacy_double
    const double u2 = ((double)c * 67108864.0 + (double)d) / 9007199254740992.0;
    x1 = 2.0 * u1 - 1.0; x2 = 2.0 * u2 - 1.0;
    r2 = x1 * x1 + x2 * x2;
}

__global__ __launch_bounds__(256) void np_accept_kernel(const u32* __restrict__ words, long long ngroups, u8* __restrict__ acc) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= ngroups) return;
    double x1, x2, r2;
    np_group(words + 4 * g, x1, x2, r2);
    acc[g] = (r2 < 1.0 && r2 != 0.0) ? 1 : 0;
}

// rank[g] = inclusive prefix sum of acc.  Accepted group with rank k <= groups yields normals 2 (k - 1) (= f x2) and 2 (k - 1) + 1
// (= f x1) of the stream; normal e belongs to the request whose [begin, end) holds e + lead (lead = 1 if a cached normal comes
// first) and is scaled by its scale: out[e + lead] = float(0.0 + scale * f x).  A sample within `margin` (relative) of a float32
// rounding boundary is appended to risky[] (its index e, for the host's libm); info[0] = index of the groups-th accepted group,
// info[1] = number of risky samples, xr[0 .. 1] = (x1, r2) of that last group (the normal an odd count leaves cached).
struct NpReq { long long begin; double scale; };
__global__ __launch_bounds__(256) void np_normals_kernel(const u32* __restrict__ words, long long ngroups, const long long* __restrict__ rank,
                                                         long long groups, long long n2, int lead, const NpReq* __restrict__ reqs, int nreq,
                                                         double margin, float* __restrict__ out, long long* __restrict__ info,
                                                         long long* __restrict__ risky, long long risky_cap, double* __restrict__ xr) {
    // (xr[0 .. 1]: the last group's (x1, r2); xr[2 + 2 slot ..]: (x, r2) of risky sample `slot`)
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= ngroups) return;
    const long long k = rank[g];
    const long long prev = g ? rank[g - 1] : 0;
    if (k == prev || k > groups) return;                             // rejected, or beyond what the draw consumes
    double x1, x2, r2;
    np_group(words + 4 * g, x1, x2, r2);
    const double f = sqrt(-2.0 * log(r2) / r2);
    if (k == groups) { info[0] = g; xr[0] = x1; xr[1] = r2; }
    const double xs[2] = {x2, x1};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const long long e = 2 * (k - 1) + h;
        if (e >= n2) break;
        const long long pos = e + lead;
        int lo = 0, hi = nreq - 1;                                   // the request whose range holds pos
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (reqs[mid].begin <= pos) lo = mid; else hi = mid - 1; }
        const double nd = 0.0 + reqs[lo].scale * (f * xs[h]);
        out[pos] = (float)nd;
        if ((float)(nd * (1.0 - margin)) != (float)(nd * (1.0 + margin))) {
            const long long slot = (long long)atomicAdd((unsigned long long*)&info[1], 1ull);
            if (slot < risky_cap) { risky[slot] = e; xr[2 + 2 * slot] = xs[h]; xr[3 + 2 * slot] = r2; }
        }
    }
}
} // namespace imgxf

using namespace imgxf;

IMGXF_API int imgxf_add_noise_philox_u8(const imgxf_view* src, const imgxf_view* dst, float sigma, uint64_t seed,
                                        uint64_t offset, void* stream) {
    IMGXF_CHECK(check_view(src));
    IMGXF_CHECK(check_view(dst));
    if (!same_geometry(src, dst)) return IMGXF_ERR_SHAPE;
    if (!(sigma >= 0.0f) || (offset & 3)) return IMGXF_ERR_ARG;       // offset counts normals: whole Philox blocks
    if (empty_view(dst)) return IMGXF_OK;
    const View d = make_view(dst);
    const int64_t total = (int64_t)d.n * d.h * ((d.rowbytes() + 3) >> 2);
    hipLaunchKernelGGL(add_noise_philox_kernel, dim3(noise_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       make_view(src), d, sigma, (u32)seed, (u32)(seed >> 32), offset >> 2);
    return launch_status();
}

IMGXF_API int imgxf_philox4x32_u32(void* dst_u32, int64_t count, uint64_t seed, uint64_t offset, void* stream) {
    if (!dst_u32) return IMGXF_ERR_NULL;
    if (count < 0 || (count & 3) || (offset & 3)) return IMGXF_ERR_ARG;
    if (count == 0) return IMGXF_OK;
    hipLaunchKernelGGL(philox_u32_kernel, dim3(noise_grid(count / 4)), dim3(256), 0, (hipStream_t)stream,
                       (u32*)dst_u32, count / 4, (u32)seed, (u32)(seed >> 32), offset >> 2);
    return launch_status();
}

IMGXF_API int imgxf_mt19937_blocks(const uint32_t* key, uint32_t* out, int64_t nblocks, void* stream) {
    if (!key || !out) return IMGXF_ERR_NULL;
    if (nblocks < 0) return IMGXF_ERR_ARG;
    hipLaunchKernelGGL(mt19937_blocks_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, key, out, (long long)nblocks);
    return launch_status();
}

IMGXF_API int imgxf_mt19937_jump(const uint32_t* base_key, uint32_t* out_keys, int n_out, const uint8_t* coefs, void* stream) {
    if (!base_key || !out_keys || !coefs) return IMGXF_ERR_NULL;
    if (n_out < 0) return IMGXF_ERR_ARG;
    if (n_out > 0) hipLaunchKernelGGL(mt19937_jump_kernel, dim3((unsigned)n_out), dim3(64), 0, (hipStream_t)stream, base_key, out_keys, coefs);
    return launch_status();
}

IMGXF_API int imgxf_mt19937_stretches(const uint32_t* keys, uint32_t* out, int n_stretches, int64_t blocks_per_stretch, int64_t total_blocks,
                                      void* stream) {
    if (!keys || !out) return IMGXF_ERR_NULL;
    if (n_stretches < 1 || blocks_per_stretch < 1 || total_blocks < 1) return IMGXF_ERR_ARG;
    hipLaunchKernelGGL(mt19937_stretch_kernel, dim3((unsigned)n_stretches), dim3(256), 0, (hipStream_t)stream, keys, out,
                       (long long)blocks_per_stretch, (long long)total_blocks);
    return launch_status();
}

IMGXF_API int imgxf_np_accept(const uint32_t* words, int64_t ngroups, uint8_t* acc, void* stream) {
    if (!words || !acc) return IMGXF_ERR_NULL;
    if (ngroups < 0 || ngroups > ((int64_t)1 << 39)) return IMGXF_ERR_ARG;
    if (ngroups) hipLaunchKernelGGL(np_accept_kernel, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, words, (long long)ngroups, acc);
    return launch_status();
}

IMGXF_API int imgxf_np_normals_f32(const uint32_t* words, int64_t ngroups, const int64_t* rank, int64_t groups, int64_t n2, int lead,
                                   const void* reqs, int nreq, double margin, float* out, int64_t* info, int64_t* risky,
                                   int64_t risky_cap, double* xr, void* stream) {
    if (!words || !rank || !reqs || !out || !info || !risky || !xr) return IMGXF_ERR_NULL;
    if (ngroups < 0 || ngroups > ((int64_t)1 << 39) || nreq < 1 || lead < 0 || lead > 1) return IMGXF_ERR_ARG;
    if (ngroups) hipLaunchKernelGGL(np_normals_kernel, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, (hipStream_t)stream, words, (long long)ngroups,
                                    (const long long*)rank, (long long)groups, (long long)n2, lead, (const NpReq*)reqs, nreq, margin, out,
                                    (long long*)info, (long long*)risky, (long long)risky_cap, xr);
    return launch_status();
}
