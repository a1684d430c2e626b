// Load/store skeleton of the marching stencil kernels without their arithmetic (development
// aid): what does the DMA -> LDS ring -> ds_read -> nt store structure copy at, against a
// one-shot uint4 copy, a read-only and a write-only stream on the SAME box, and how does it
// react to the workgroup shape, chunk height, rows in flight and store policy?
// build: hipcc -O3 --offload-arch=gfx950 tools/exp_skeleton.hip -o _exp/exp_skeleton
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef uint32_t u32;
typedef uint8_t u8;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

#define GLDS16A(gptr, lptr, aux) \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, aux)
#define GLDS16(gptr, lptr) GLDS16A(gptr, lptr, 0)

template <int U, int NT>
__global__ __launch_bounds__(256) void copyk(const u32x4* __restrict__ a, u32x4* __restrict__ b, size_t n) {
    const size_t base = (size_t)blockIdx.x * (256 * U);
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        size_t i = base + u * 256 + threadIdx.x;
        if (i < n) v[u] = a[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        size_t i = base + u * 256 + threadIdx.x;
        if (i < n) { if (NT) __builtin_nontemporal_store(v[u], b + i); else b[i] = v[u]; }
    }
}
template <int U>
__global__ __launch_bounds__(256) void readk(const u32x4* __restrict__ a, u32x4* __restrict__ b, size_t n) {
    const size_t base = (size_t)blockIdx.x * (256 * U);
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        size_t i = base + u * 256 + threadIdx.x;
        if (i < n) acc ^= a[i];
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) b[base] = acc;     // never true in practice
}
template <int U, int NT>
__global__ __launch_bounds__(256) void writek(u32x4* __restrict__ b, size_t n) {
    const size_t base = (size_t)blockIdx.x * (256 * U);
    const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        size_t i = base + u * 256 + threadIdx.x;
        if (i < n) { if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v; }
    }
}

// The marching skeleton: a wave owns a 1 KiB strip of the super-row of G frames and walks
// rows_per_wave rows; J rows in flight in a wave-private LDS ring; HALO: a second DMA of two
// 16-byte blocks per row as in the stencil kernels; NT: nontemporal store; WORK: dummy VALU
// instructions per row (dependent fmacs on the loaded data) to emulate an issue-bound body.
template <int J, bool HALO, bool NT, int WORK, int AUX = 0>
__global__ __launch_bounds__(768) void marchk(const u8* __restrict__ src, u8* __restrict__ dst, int h, int64_t rs, int64_t fs,
                                              int rows_per_wave, int nchunks, int gx, int G, int bpr, int order = 0) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int SLOT = 1024 + 32;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
    const int logical = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
    int bx = logical % gx, chunk = (logical / gx) % nchunks, f = (logical / (gx * nchunks)) * G;
    // work orders (round 3): which (strip, chunk, frame group) does workgroup id `orig` take?
    //  0  XCD-contiguous ranges, strips fastest, then chunks, then frame groups (the shipped order)
    //  1  every XCD works on the SAME frame group: XCD k owns the chunks k, k+8, k+16, ... of it
    //  2  plain row-major ids (no XCD decode)
    //  3  XCD-contiguous ranges, chunks fastest, then strips
    //  4  XCD-contiguous ranges, frame groups fastest (the same rows of many frames are in flight)
    //  5  every XCD works on the same frame group: XCD k owns a contiguous band of chunks of it
    if (order != 0) {
        const int cpx = (nchunks + 7) / 8, l = orig >> 3, ngroups = nwg / (8 * gx * cpx);
        if (order == 1 || order == 5) {
            bx = l % gx; const int c8 = (l / gx) % cpx; f = (l / (gx * cpx)) * G;
            chunk = order == 1 ? c8 * 8 + xcd : xcd * cpx + c8;
            if (chunk >= nchunks || l / (gx * cpx) >= ngroups) return;
        } else if (order == 2) {
            bx = orig % gx; chunk = (orig / gx) % nchunks; f = (orig / (gx * nchunks)) * G;
        } else if (order == 3) {
            chunk = logical % nchunks; bx = (logical / nchunks) % gx; f = (logical / (gx * nchunks)) * G;
        } else if (order == 4) {
            const int ng = nwg / (gx * nchunks);
            f = (logical % ng) * G; bx = (logical / ng) % gx; chunk = logical / (ng * gx);
        }
    }
    const int total = G * bpr;
    const int strip = bx * (blockDim.x >> 6) + wave;
    if (strip * 64 >= total) return;
    const int gb = strip * 64 + lane;
    const bool active = gb < total;
    const int gbc = active ? gb : total - 1;
    const int fr = gbc / bpr, bi = gbc - fr * bpr;
    u32 xo = (u32)fr * (u32)fs + (u32)bi * 16u;
    u32 eo = lane == 0 ? (xo >= 16 ? xo - 16 : xo) : xo;            // some neighbouring block
    const int y_begin = chunk * rows_per_wave, y_end = min(h, y_begin + rows_per_wave);
    const u8* sbase = src + (int64_t)f * fs;
    char* wl = lds + wave * (16 + J * SLOT) + 16;
    const u32 a_main = (u32)(uintptr_t)(__attribute__((address_space(3))) char*)wl + lane * 16;
    auto issue = [&](int y, int slot) {
        const u8* rowp = sbase + (int64_t)y * rs;
        asm volatile("" : "+v"(xo), "+v"(eo));
        GLDS16A(rowp + xo, wl + slot * SLOT, AUX);
        if (HALO) { if (lane < 2) GLDS16(rowp + eo, wl + slot * SLOT + 1024); }
    };
#pragma unroll
    for (int j = 0; j < J; ++j) issue(min(y_begin + j, y_end - 1), j);
    u8* drow = dst + (int64_t)f * fs + (int64_t)y_begin * rs;
    float carry = 0.0f;
    for (int base = y_begin; base < y_end; base += J) {
#pragma unroll
        for (int s = 0; s < J; ++s) {
            const int y = base + s;
            u32x4 q;
            asm volatile("s_waitcnt vmcnt(%2)\n\tds_read_b128 %0, %1 offset:%3\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(q) : "v"(a_main), "n"((HALO ? 2 : 1) * (J - 1)), "n"(s * SLOT) : "memory");
            issue(min(y + J, y_end - 1), s);
            if (WORK > 0) {
                float t0 = __uint_as_float(q.x & 0x3fffffffu), t1 = __uint_as_float(q.y & 0x3fffffffu);
                float t2 = __uint_as_float(q.z & 0x3fffffffu), t3 = __uint_as_float(q.w & 0x3fffffffu);
#pragma unroll
                for (int k = 0; k < WORK; ++k) {
                    t0 = fmaf(t0, 1.0001f, carry); t1 = fmaf(t1, 0.9999f, t0); t2 = fmaf(t2, 1.0002f, t1); t3 = fmaf(t3, 0.9998f, t2);
                }
                carry = t3 * 1.0e-30f;
                q.x ^= (u32)(carry != carry);       // NaN-only perturbation keeps the chain alive
            }
            if (y < y_end && active) {
                u32 so = xo;
                asm volatile("" : "+v"(so));
                if (NT) __builtin_nontemporal_store(q, (u32x4*)(drow + so)); else *(u32x4*)(drow + so) = q;
            }
            drow += rs;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// every workgroup copies one 4 KiB piece from a pseudo-random place to the same place of dst
__global__ __launch_bounds__(256) void scatter4k(const u32x4* __restrict__ a, u32x4* __restrict__ b, size_t npieces) {
    const size_t piece = ((size_t)blockIdx.x * 2654435761ull) % npieces;     // npieces odd-ish: a permutation-like spread
    const size_t i = piece * 256 + threadIdx.x;
    __builtin_nontemporal_store(a[i], b + i);
}
__global__ void fillk(uint32_t* p, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 7);
}

static hipEvent_t ev0, ev1;
template <class Fn> static float timeit(Fn fn, int iters = 8) {
    fn(); fn();
    hipEventRecord(ev0);
    for (int i = 0; i < iters; ++i) fn();
    hipEventRecord(ev1); hipEventSynchronize(ev1);
    float ms; hipEventElapsedTime(&ms, ev0, ev1);
    return ms / iters;
}

struct Geo { int n, h; int64_t rs, fs; };

template <int J, bool HALO, bool NT, int WORK, int AUX = 0>
static void run_march(const char* tag, const u8* a, u8* b, Geo g, int G, int spb, int rpw, int pad_lds = 0, int order = 0) {
    const int bpr = (int)(g.rs / 16);
    const int nstrips = (G * bpr + 63) / 64;
    const int gx = (nstrips + spb - 1) / spb;
    const int nchunks = (g.h + rpw - 1) / rpw;
    const int rpw2 = (g.h + nchunks - 1) / nchunks;
    const int groups = g.n / G;
    const size_t lds = (size_t)spb * (16 + J * 1056) + pad_lds;
    const unsigned nwg = (order == 1 || order == 5) ? (unsigned)(8 * gx * ((nchunks + 7) / 8) * groups) : (unsigned)(gx * nchunks * groups);
    float ms = timeit([&] {
        hipLaunchKernelGGL((marchk<J, HALO, NT, WORK, AUX>), dim3(nwg), dim3(64 * spb), lds, 0, a, b, g.h, g.rs, g.fs, rpw2, nchunks, gx, G, bpr, order);
    });
    if (order) printf("order %d  ", order);
    const double bytes = 2.0 * g.n * g.h * g.rs;
    printf("march %-22s aux=%-2d lds=%-6zu J=%d halo=%d nt=%d work=%-3d G=%d spb=%d rpw=%-4d wgs=%-6u %7.3f ms %7.1f GB/s\n", tag, AUX, lds, J, (int)HALO, (int)NT, WORK, G, spb,
           rpw2, nwg, ms, bytes / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 128;
    const int H = argc > 2 ? atoi(argv[2]) : 2160, W = argc > 3 ? atoi(argv[3]) : 3840;
    Geo g{F, H, (int64_t)W * 3, (int64_t)W * 3 * H};
    const size_t bytes = (size_t)F * g.fs, n = bytes / 16;
    u8 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipEventCreate(&ev0); hipEventCreate(&ev1);
    hipLaunchKernelGGL(fillk, dim3(4096), dim3(256), 0, 0, (uint32_t*)a, bytes / 4);
    hipDeviceSynchronize();
    auto rep = [&](const char* tag, float ms, double mult) { printf("%-40s %7.3f ms %7.1f GB/s\n", tag, ms, mult * bytes / ms / 1e6); fflush(stdout); };
    for (int rep_i = 0; rep_i < 2; ++rep_i) {
        rep("oneshot copy U=4", timeit([&] { hipLaunchKernelGGL((copyk<4, 0>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n); }), 2);
        rep("oneshot copy U=4 nt store", timeit([&] { hipLaunchKernelGGL((copyk<4, 1>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n); }), 2);
        rep("oneshot copy U=1", timeit([&] { hipLaunchKernelGGL((copyk<1, 0>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n); }), 2);
        rep("read only U=4", timeit([&] { hipLaunchKernelGGL((readk<4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n); }), 1);
        rep("write only U=4", timeit([&] { hipLaunchKernelGGL((writek<4, 0>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, (u32x4*)b, n); }), 1);
        rep("write only U=4 nt", timeit([&] { hipLaunchKernelGGL((writek<4, 1>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, (u32x4*)b, n); }), 1);
        rep("hipMemcpyAsync d2d", timeit([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }), 2);
        if (getenv("SKEL_POLICY")) {
            rep("scattered 4 KiB pieces copy", timeit([&] { hipLaunchKernelGGL(scatter4k, dim3((unsigned)(n / 256)), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n / 256); }), 2);
            run_march<5, true, true, 0, 0>("G1 aux0", a, b, g, 1, 4, 94);
            run_march<5, true, true, 0, 2>("G1 aux2 nt", a, b, g, 1, 4, 94);
            run_march<5, true, true, 0, 1>("G1 aux1 sc0", a, b, g, 1, 4, 94);
            run_march<5, true, true, 0, 16>("G1 aux16 sc1", a, b, g, 1, 4, 94);
            run_march<5, true, true, 0, 18>("G1 aux18 sc1 nt", a, b, g, 1, 4, 94);
            run_march<12, true, true, 0, 2>("G1 J12 nt", a, b, g, 1, 4, 94);
            run_march<10, true, true, 0, 2>("G1 J10 nt", a, b, g, 1, 4, 94);
            run_march<10, true, true, 0, 0>("G1 J10", a, b, g, 1, 4, 94);
            run_march<5, true, true, 0, 2>("G4 nt", a, b, g, 4, 3, 94);
            run_march<10, true, true, 0, 2>("G4 J10 nt", a, b, g, 4, 3, 94);
            run_march<5, true, true, 0, 0>("G1 4wg/cu", a, b, g, 1, 4, 94, 40960 - 4 * 5296);
            run_march<5, true, true, 0, 0>("G1 3wg/cu", a, b, g, 1, 4, 94, 53248 - 4 * 5296);
            run_march<5, true, true, 0, 0>("G1 2wg/cu", a, b, g, 1, 4, 94, 80000 - 4 * 5296);
            run_march<5, true, true, 35, 2>("G1 work140 nt", a, b, g, 1, 4, 94);
            run_march<5, true, true, 50, 2>("G1 work200 nt", a, b, g, 1, 4, 94);
            run_march<5, true, true, 50, 0>("G1 work200", a, b, g, 1, 4, 94);
            printf("----\n");
            continue;
        }
        if (getenv("SKEL_ORDER")) {
            // the shipped launch shape (super-rows of 4 frames, single-wave workgroups, 64-row chunks) and the
            // full-row-workgroup shape (all 45 strips of a super-row in one workgroup... 12 strips of one frame) in every order
            for (int order : {0, 1, 5, 2, 3, 4}) run_march<5, true, true, 0>("G4 spb1 rpw64", a, b, g, 4, 1, 64, 0, order);
            for (int order : {0, 1, 5, 2, 3, 4}) run_march<5, true, true, 0>("G1 spb12 rpw64", a, b, g, 1, 12, 64, 0, order);
            for (int order : {0, 1, 5, 3}) run_march<5, true, true, 0>("G4 spb9 rpw64", a, b, g, 4, 9, 64, 0, order);
            for (int order : {0, 1, 5}) run_march<5, true, true, 0>("G4 spb1 rpw32", a, b, g, 4, 1, 32, 0, order);
            for (int order : {0, 1, 5}) run_march<5, true, true, 0>("G4 spb1 rpw135", a, b, g, 4, 1, 135, 0, order);
            for (int order : {0, 1, 5}) run_march<5, true, true, 35>("G4 spb1 rpw64 work140", a, b, g, 4, 1, 64, 0, order);
            printf("----\n");
            continue;
        }
        if (getenv("SKEL_RPW")) {
            for (int rpw : {1, 2, 4, 8, 16, 32, 64, 94}) run_march<5, true, true, 0>("G1 spb4", a, b, g, 1, 4, rpw);
            for (int rpw : {1, 4, 16, 94}) run_march<5, true, true, 0>("G1 spb12", a, b, g, 1, 12, rpw);
            for (int rpw : {4, 16, 94}) run_march<2, true, true, 0>("G1 spb4 J2", a, b, g, 1, 4, rpw);
            for (int rpw : {4, 16, 94}) run_march<5, false, true, 0>("G1 spb4 nohalo", a, b, g, 1, 4, rpw);
            printf("----\n");
            continue;
        }
        if (getenv("SKEL_WIDE")) {
            run_march<5, true, true, 0>("G1 spb4", a, b, g, 1, 4, 94);
            run_march<5, true, true, 0>("G1 spb6", a, b, g, 1, 6, 94);
            run_march<5, true, true, 0>("G1 spb12", a, b, g, 1, 12, 94);
            run_march<5, true, true, 0>("G1 spb12 rpw 47", a, b, g, 1, 12, 47);
            run_march<5, true, true, 0>("G1 spb12 rpw 24", a, b, g, 1, 12, 24);
            run_march<10, true, true, 0>("G1 spb12 J10", a, b, g, 1, 12, 94);
            run_march<5, true, true, 0>("G4 spb3", a, b, g, 4, 3, 94);
            run_march<5, true, true, 0>("G4 spb9", a, b, g, 4, 9, 94);
            run_march<5, true, true, 50>("G1 spb4 work200", a, b, g, 1, 4, 94);
            run_march<5, true, true, 50>("G1 spb12 work200", a, b, g, 1, 12, 94);
            run_march<5, true, true, 50>("G4 spb3 work200", a, b, g, 4, 3, 94);
            printf("----\n");
            continue;
        }
        if (getenv("SKEL_MATRIX")) {
            for (int G : {1, 2, 4, 8})
                for (int spb : {1, 2, 3, 4}) run_march<5, true, true, 0>("matrix", a, b, g, G, spb, 94);
            run_march<5, true, true, 0>("G1 rpw47", a, b, g, 1, 4, 47);
            run_march<5, true, true, 0>("G1 rpw180", a, b, g, 1, 4, 180);
            run_march<3, true, true, 0>("G1 J3", a, b, g, 1, 4, 94);
            run_march<12, true, true, 0>("G1 J12", a, b, g, 1, 4, 94);
            run_march<5, false, true, 0>("G1 nohalo", a, b, g, 1, 4, 94);
            run_march<5, true, true, 35>("G1 work140", a, b, g, 1, 4, 94);
            run_march<5, true, true, 50>("G1 work200", a, b, g, 1, 4, 94);
            printf("----\n");
            continue;
        }
        const int G = (W * 3 / 16) % 64 == 0 ? 1 : ((W == 3840) ? 4 : (W == 1920 ? 8 : 1));
        run_march<5, true, true, 0>("base", a, b, g, G, 3, 94);
        run_march<5, true, true, 0>("G=1 spb4", a, b, g, 1, 4, 94);
        run_march<5, true, true, 0>("spb1", a, b, g, G, 1, 94);
        run_march<5, true, true, 0>("spb5", a, b, g, G, 5, 94);
        run_march<5, true, true, 0>("spb8 (wait: 45%8)", a, b, g, G, 8, 94);
        run_march<5, true, true, 0>("rpw 47", a, b, g, G, 3, 47);
        run_march<5, true, true, 0>("rpw 180", a, b, g, G, 3, 180);
        run_march<5, true, true, 0>("rpw 540", a, b, g, G, 3, 540);
        run_march<5, false, true, 0>("no halo", a, b, g, G, 3, 94);
        run_march<5, true, false, 0>("plain store", a, b, g, G, 3, 94);
        run_march<3, true, true, 0>("J3", a, b, g, G, 3, 94);
        run_march<8, true, true, 0>("J8", a, b, g, G, 3, 94);
        run_march<12, true, true, 0>("J12", a, b, g, G, 3, 94);
        run_march<5, true, true, 20>("work 80 fma", a, b, g, G, 3, 94);
        run_march<5, true, true, 35>("work 140 fma", a, b, g, G, 3, 94);
        run_march<5, true, true, 50>("work 200 fma", a, b, g, G, 3, 94);
        printf("----\n");
    }
    return 0;
}
