// Baseline JPEG writer on the device: the byte stream Pillow's `Image.save(fp, "JPEG")` (libjpeg-turbo: 4:2:0, islow
// DCT, Annex-K Huffman tables, no restart markers) writes for an RGB image — the save step of the reference driver,
// transformation.py:161-162.  A batch of n frames → n independent files.
//
//   jpeg_transform_kernel   RGB → YCbCr (16-bit fixed point) → 2×2 chroma averaging → 8×8 forward DCT → quantise;
//                           16 MCUs (16×256 px) per workgroup staged through LDS, one thread per 8×8 block, zigzag
//                           int16 coefficients in MCU order (6 blocks per MCU), 16-byte pieces interleaved over groups
//                           of 64 blocks, plus, per block, its DC value and the bits its AC symbols will take
//   jpeg_lens_kernel        bits per block (DC difference code + AC bits)               → exclusive scan = bit offsets
//   jpeg_zero_kernel        clears the stream words the emit kernel ORs into
//   jpeg_emit_kernel        one thread per block walks its coefficients; a workgroup's 256 blocks form one contiguous
//                           span of the stream, merged in LDS and stored whole (atomics only on the two shared words)
//   jpeg_ffcount / jpeg_stuff_kernel   0xFF → 0xFF 0x00 byte stuffing (count per 32-byte chunk on words, scan, expand
//                           in LDS, coalesced stores), header, padding of the last byte with 1-bits, EOI, file size
//
// Integer arithmetic throughout: bit-identical to the library (tests/test_gpu_jpeg.py compares whole files).
#include "imgxf_common.h"
#include <string.h>

namespace imgxf {

constexpr int JM = 16;                       // MCUs per transform workgroup: 4·JM luminance + 2·JM chrominance blocks
constexpr int JT = 128;                      // threads (>= 6·JM blocks; 32 MCUs × 192 threads measured slower: 460 vs 430 µs)
constexpr int JPX = 16 * JM;                 // pixels per row of the workgroup's strip
constexpr int JCHUNK = 32;                   // bytes per stuffing thread
constexpr unsigned JLW = 4096;               // words of an emit workgroup's span merged in LDS (16 KB)

struct JpegQuant {                           // per coefficient (natural order): |c| → (((|c| + half) << sh) · m) >> 32, 24-bit operands
    u32 m[2][64];                            // ceil(2^32 / (8q << sh)) < 2^24, sh = the shift that brings 8q above 256
    u32 half[2][64];                         // 4q | sh << 16
    u8 aclen[2][256];                        // bits of the AC symbol (run << 4) | size: Huffman code length + size
};
struct JpegHuff {                            // code | len << 16
    u32 dc[2][16];
    u32 ac[2][256];
};
struct JpegHeader {
    u8 b[1024];
    int len;
};

// jfdctint.c, one 1-D pass over eight values (FIRST: the row pass, results scaled up by 4).  Same sums as the library's
// (32-bit two's complement, any association); the rounding constant of DESCALE rides in the shared terms z1 / z5 so
// that every output is one multiply-add chain and a shift.
// (the multiplies as explicit 24-bit instructions: left to the compiler, the second pass — whose operand ranges it cannot
// bound — comes out as quarter-rate v_mul_lo_u32 / v_mad_u64_u32)
__device__ __forceinline__ int mul24c(int a, int c) {
    int r;
    asm("v_mul_i32_i24_e32 %0, %1, %2" : "=v"(r) : "s"(c), "v"(a));
    return r;
}
__device__ __forceinline__ int mad24c(int a, int c, int acc) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(c), "v"(acc));
    return r;
}

template <bool FIRST>
__device__ __forceinline__ void fdct8(int& d0, int& d1, int& d2, int& d3, int& d4, int& d5, int& d6, int& d7) {
    constexpr int N = FIRST ? 11 : 15, R = 1 << (N - 1);
    const int t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6;
    const int t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    if (FIRST) {
        d0 = (t10 + t11) << 2;
        d4 = (t10 - t11) << 2;
    } else {
        d0 = (t10 + t11 + 2) >> 2;
        d4 = (t10 - t11 + 2) >> 2;
    }
    const int z1r = mad24c(t12 + t13, 4433, R);
    d2 = mad24c(t13, 6270, z1r) >> N;
    d6 = mad24c(t12, -15137, z1r) >> N;
    const int s1 = t4 + t7, s2 = t5 + t6, s3 = t4 + t6, s4 = t5 + t7;
    const int z5r = mad24c(s3 + s4, 9633, R);
    const int z3 = mad24c(s3, -16069, z5r), z4 = mad24c(s4, -3196, z5r);
    const int z1 = mul24c(s1, -7373), z2 = mul24c(s2, -20995);
    d7 = mad24c(t4, 2446, z1 + z3) >> N;
    d5 = mad24c(t5, 16819, z2 + z4) >> N;
    d3 = mad24c(t6, 25172, z2 + z3) >> N;
    d1 = mad24c(t7, 12299, z1 + z4) >> N;
}

// jccolor.c: Y = (19595 R + 38470 G + 7471 B + 32768) >> 16 on a dword holding R, G, B in its low three bytes: the 16-bit
// constants split into bytes for two v_dot4_u32_u8 (76·256+139, 150·256+70, 29·256+47); the fourth byte has weight 0.
__device__ __forceinline__ u32 ycc_y_dot(u32 rgbx) {
    const u32 hi = __builtin_amdgcn_udot4(rgbx, 76u | (150u << 8) | (29u << 16), 0u, false);
    const u32 lo = __builtin_amdgcn_udot4(rgbx, 139u | (70u << 8) | (47u << 16), 32768u, false);
    return ((hi << 8) + lo) >> 16;
}
__device__ __forceinline__ u32 ycc_cb(int r, int g, int b) { return (u32)(-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16; }
__device__ __forceinline__ u32 ycc_cr(int r, int g, int b) { return (u32)(32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16; }

constexpr int zz(int i) {
    constexpr int t[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                           41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                           15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55,
                           62, 63};
    return t[i];
}

// One workgroup: rows y0 .. y0+15, columns x0 .. x0+16·JM-1 of frame f.  All threads stage and convert, then one thread
// per block transforms — wave 0 the 64 luminance blocks, half of wave 1 the 32 chrominance blocks, so that the
// quantiser table is wave-uniform (scalar loads).
__global__ __launch_bounds__(JT) void jpeg_transform_kernel(View s, int16_t* __restrict__ coef, int64_t coef_fs,
                                                             int16_t* __restrict__ dcs, uint16_t* __restrict__ acbits, int nblk,
                                                             int mw, int bw, int bh, JpegQuant q) {
    __shared__ __attribute__((aligned(4))) u8 slen[2][256];
    __shared__ __attribute__((aligned(16))) u8 rgb[16][JPX * 3];
    __shared__ __attribute__((aligned(16))) u8 yp[16][JPX + 8];
    __shared__ __attribute__((aligned(16))) u8 cp[2][8][JPX / 2 + 8];
    const int tid = threadIdx.x, f = blockIdx.z, my = blockIdx.y, mx0 = blockIdx.x * JM;
    const int y0 = my * 16, x0 = mx0 * 16;
    if (tid < 128) ((u32*)slen)[tid] = ((const u32*)q.aclen)[tid];
    const u8* base = s.p + (int64_t)f * s.fs;
    const bool fast = (x0 + JPX <= s.w) && (((uintptr_t)base | (uintptr_t)s.rs) & 15) == 0;
    constexpr int CPR = JPX * 3 / 16;                          // 16-byte pieces per row
    for (int i = tid; i < 16 * CPR; i += JT) {
        const int r = i / CPR, ch = i - r * CPR;
        const u8* row = base + (int64_t)min(y0 + r, s.h - 1) * s.rs;
        if (fast) {
            *(uint4*)&rgb[r][ch * 16] = *(const uint4*)(row + x0 * 3 + ch * 16);
        } else {
            for (int b = 0; b < 16; ++b) {
                const int o = ch * 16 + b, px = o / 3, cc = o - px * 3;
                rgb[r][o] = row[min(x0 + px, s.w - 1) * 3 + cc];
            }
        }
    }
    __syncthreads();
    // luminance: four pixels (three dwords) per task
    constexpr int G4 = JPX / 4;                                // groups of four pixels per row
    for (int i = tid; i < 16 * G4; i += JT) {
        const int r = i / G4, g4 = i - r * G4;
        const u32* p = (const u32*)&rgb[r][g4 * 12];
        const u32 a = p[0], b = p[1], c = p[2];
        const u32 y0v = ycc_y_dot(a);
        const u32 y1v = ycc_y_dot(__builtin_amdgcn_alignbit(b, a, 24));
        const u32 y2v = ycc_y_dot(__builtin_amdgcn_alignbit(c, b, 16));
        const u32 y3v = ycc_y_dot(c >> 8);
        *(u32*)&yp[r][g4 * 4] = y0v | (y1v << 8) | (y2v << 16) | (y3v << 24);
    }
    // chrominance: two samples (4×2 pixels) per task; rows past the image repeat the last DOWNSAMPLED row
    const int crows = (s.h + 1) >> 1;
    for (int i = tid; i < 8 * G4; i += JT) {
        const int j = i / G4, g4 = i - j * G4;
        const int ce = min(y0 / 2 + j, crows - 1);
        const int ra = 2 * ce - y0, rb = min(2 * ce + 1, s.h - 1) - y0;
        const u32* pa = (const u32*)&rgb[ra][g4 * 12];
        const u32* pb = (const u32*)&rgb[rb][g4 * 12];
        u32 cb[2] = {1, 2}, cr[2] = {1, 2};                      // h2v2_downsample's alternating bias
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u32* p = h ? pb : pa;
            const u32 a = p[0], b = p[1], c = p[2];
            const int r0 = a & 255, g0 = (a >> 8) & 255, b0 = (a >> 16) & 255;
            const int r1 = a >> 24, g1 = b & 255, b1 = (b >> 8) & 255;
            const int r2 = (b >> 16) & 255, g2 = b >> 24, b2 = c & 255;
            const int r3 = (c >> 8) & 255, g3 = (c >> 16) & 255, b3 = c >> 24;
            cb[0] += ycc_cb(r0, g0, b0) + ycc_cb(r1, g1, b1);
            cb[1] += ycc_cb(r2, g2, b2) + ycc_cb(r3, g3, b3);
            cr[0] += ycc_cr(r0, g0, b0) + ycc_cr(r1, g1, b1);
            cr[1] += ycc_cr(r2, g2, b2) + ycc_cr(r3, g3, b3);
        }
        *(uint16_t*)&cp[0][j][g4 * 2] = (uint16_t)((cb[0] >> 2) | ((cb[1] >> 2) << 8));
        *(uint16_t*)&cp[1][j][g4 * 2] = (uint16_t)((cr[0] >> 2) | ((cr[1] >> 2) << 8));
    }
    __syncthreads();
    if (tid >= 6 * JM) return;
    const int chroma = __builtin_amdgcn_readfirstlane(tid >= 4 * JM ? 1 : 0);
    int ml, k;
    const u8* origin;
    int stride;
    bool real;
    if (!chroma) {
        ml = tid >> 2;
        k = tid & 3;
        origin = &yp[(k >> 1) * 8][ml * 16 + (k & 1) * 8];
        stride = JPX + 8;
        real = (2 * my + (k >> 1) < bh) && (2 * (mx0 + ml) + (k & 1) < bw);
    } else {
        const int c = (tid - 4 * JM) / JM;
        ml = (tid - 4 * JM) % JM;
        k = 4 + c;
        origin = &cp[c][0][ml * 8];
        stride = JPX / 2 + 8;
        real = true;
    }
    if (mx0 + ml >= mw || !real) return;
    int d[64];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint2 v = *(const uint2*)(origin + r * stride);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            d[r * 8 + x] = (int)((v.x >> (8 * x)) & 255) - 128;
            d[r * 8 + 4 + x] = (int)((v.y >> (8 * x)) & 255) - 128;
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
        fdct8<true>(d[r * 8], d[r * 8 + 1], d[r * 8 + 2], d[r * 8 + 3], d[r * 8 + 4], d[r * 8 + 5], d[r * 8 + 6], d[r * 8 + 7]);
#pragma unroll
    for (int c = 0; c < 8; ++c)
        fdct8<false>(d[c], d[8 + c], d[16 + c], d[24 + c], d[32 + c], d[40 + c], d[48 + c], d[56 + c]);
#pragma unroll
    for (int i = 0; i < 64; ++i) {                                // jcdctmgr.c quantize: sign · ((|c| + 4q) / 8q)
        const int v = d[i], sg = v >> 31;
        const u32 a = (u32)((v ^ sg) - sg);
        const u32 hs = q.half[chroma][i];
        const u32 x = (a + (hs & 0xffff)) << (hs >> 16);         // < 2^23
        const u32 qq = (u32)(((unsigned long long)(x & 0xffffffu) * (q.m[chroma][i] & 0xffffffu)) >> 32);   // v_mul_hi_u32_u24: full rate
        d[i] = ((int)qq ^ sg) - sg;
    }
    const int64_t blk = ((int64_t)my * mw + mx0 + ml) * 6 + k;
    {   // bits of the AC part of this block (jchuff.c encode_one_block), so that only the DC term needs the neighbours
        const u8* lt = slen[chroma];
        u32 acc = 0, run16 = 0;                                 // acc: bits | ZRL symbols << 16; run16: 16 · zero run
#pragma unroll
        for (int i = 1; i < 64; ++i) {
            const int c = d[zz(i)];
            const u32 a = (u32)max(c, -c);
            const u32 cat = 32 - (u32)__clz((int)a);          // 0 for a == 0; <= 11 for 8-bit samples
            const u32 add = lt[(run16 & 0xf0) | cat] + ((run16 & 0xff00) << 8);   // code length + size; runs of 16 zeros
            acc += a ? add : 0u;
            run16 = a ? 0u : run16 + 16;
        }
        u32 bits = (acc & 0xffff) + (acc >> 16) * lt[0xF0];
        if (run16) bits += lt[0];
        acbits[(int64_t)f * nblk + blk] = (uint16_t)bits;
        dcs[(int64_t)f * nblk + blk] = (int16_t)d[0];
    }
    // zigzag order, eight coefficients (16 bytes) at a time, interleaved over groups of 64 blocks: piece g of block b
    // at 16-byte slot (b >> 6)·512 + g·64 + (b & 63), so that the emit kernel's one-thread-per-block walk reads
    // consecutive 16-byte pieces across a wave
    uint4* out = (uint4*)(coef + (int64_t)f * coef_fs) + (blk >> 6) * 512 + (blk & 63);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        uint4 v;
        v.x = (u32)(d[zz(g * 8 + 0)] & 0xffff) | ((u32)d[zz(g * 8 + 1)] << 16);
        v.y = (u32)(d[zz(g * 8 + 2)] & 0xffff) | ((u32)d[zz(g * 8 + 3)] << 16);
        v.z = (u32)(d[zz(g * 8 + 4)] & 0xffff) | ((u32)d[zz(g * 8 + 5)] << 16);
        v.w = (u32)(d[zz(g * 8 + 6)] & 0xffff) | ((u32)d[zz(g * 8 + 7)] << 16);
        out[g * 64] = v;
    }
}

// ---- entropy coding -------------------------------------------------------------------------------------------------

struct JpegGeom {
    int mw, mh, bw, bh, nblk;
};

// jccoefct.c compress_data: block k of an MCU is a dummy (zero AC, DC of the block before it) when it lies past the
// component's last real block row / column; returns the block whose DC it carries.
__device__ __forceinline__ int dc_source(const JpegGeom& g, int mx, int my, int k, bool& dummy) {
    dummy = false;
    if (k >= 4) return k;
    const int yi = k >> 1, xi = k & 1;
    const bool rowok = 2 * my + yi < g.bh, colok = 2 * mx + xi < g.bw;
    if (rowok && colok) return k;
    dummy = true;
    if (!rowok) return (2 * mx + 1 < g.bw) ? 1 : 0;          // a whole dummy row: DC of the top row's last real block
    return k - 1;                                              // right edge: the block to its left
}

// DC value carried by block k of MCU (mx, my), and the one the DC difference is taken against (the block of the same
// component before it in scan order; 0 at the start of the frame).
__device__ __forceinline__ int block_dc(const int16_t* __restrict__ dcs, const JpegGeom& g, int mcu, int mx, int my, int k, bool& dummy) {
    return dcs[(int64_t)mcu * 6 + dc_source(g, mx, my, k, dummy)];
}
__device__ __forceinline__ int block_pred(const int16_t* __restrict__ dcs, const JpegGeom& g, int mcu, int mx, int my, int k) {
    bool pd;
    if (k >= 4) return mcu > 0 ? dcs[(int64_t)(mcu - 1) * 6 + k] : 0;
    if (k > 0) return block_dc(dcs, g, mcu, mx, my, k - 1, pd);
    if (mcu == 0) return 0;
    const int pm = mcu - 1, pmy = pm / g.mw, pmx = pm - pmy * g.mw;
    return block_dc(dcs, g, pm, pmx, pmy, 3, pd);
}

// bits of every block: DC category code + magnitude bits + the AC bits the transform kernel counted (EOB for a dummy)
__global__ __launch_bounds__(256) void jpeg_lens_kernel(const int16_t* __restrict__ dcs, const uint16_t* __restrict__ acbits,
                                                        u32* __restrict__ lens, JpegGeom g, JpegHuff hf) {
    const int j = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (j >= g.nblk) return;
    const int mcu = j / 6, k = j - mcu * 6, my = mcu / g.mw, mx = mcu - my * g.mw;
    const int16_t* dd = dcs + (int64_t)f * g.nblk;
    bool dummy;
    const int diff = block_dc(dd, g, mcu, mx, my, k, dummy) - block_pred(dd, g, mcu, mx, my, k);
    const int sg = diff >> 31, t = k >= 4 ? 1 : 0;
    const u32 cat = 32 - (u32)__clz((diff ^ sg) - sg);
    lens[(int64_t)f * g.nblk + j] = (hf.dc[t][cat] >> 16) + cat + (dummy ? hf.ac[t][0] >> 16 : (u32)acbits[(int64_t)f * g.nblk + j]);
}

// One thread per block: the coefficients come interleaved over 64 blocks (see jpeg_transform_kernel), so a wave's
// loads are consecutive dwords; every thread walks its block and writes the codes MSB-first at the block's bit offset:
// the first word it touches is shared with the previous block (atomic OR), the words after it are its own (plain
// stores), the last partial word is shared with the next block (atomic OR).
__global__ __launch_bounds__(256) void jpeg_emit_kernel(const int16_t* __restrict__ coef, int64_t coef_fs, const int16_t* __restrict__ dcs,
                                                        const u32* __restrict__ offs, u32* __restrict__ stream, int64_t stream_fs_words,
                                                        const u32* __restrict__ total_bits, JpegGeom g, JpegHuff hf) {
    constexpr u32 LW = JLW;
    __shared__ u32 sdc[2][16];
    __shared__ u32 sac[2][256];
    __shared__ u32 lbuf[LW];
    const int f = blockIdx.y, j0 = blockIdx.x * 256, j = j0 + threadIdx.x;
    if (((unsigned long long)total_bits[f] + 31) / 32 > (unsigned long long)stream_fs_words) return;   // reported by jpeg_stuff_kernel
    for (int i = threadIdx.x; i < 32; i += 256) sdc[i >> 4][i & 15] = hf.dc[i >> 4][i & 15];
    for (int i = threadIdx.x; i < 512; i += 256) sac[i >> 8][i & 255] = hf.ac[i >> 8][i & 255];
    // the bits of this workgroup's 256 blocks are contiguous: [offs[j0], offs[j0 + 256]).  When that span fits LW words
    // the codes are merged in LDS (ds_or) and leave as coalesced stores, with global atomics only on the two words
    // shared with the neighbouring workgroups; longer spans (≈ > 500 bits per block) write to the stream directly.
    const int j1 = min(j0 + 256, g.nblk);
    const u32 sbit = offs[(int64_t)f * g.nblk + j0];
    const u32 ebit = j1 < g.nblk ? offs[(int64_t)f * g.nblk + j1] : total_bits[f];
    const u32 wlo = sbit >> 5, nw = ((ebit + 31) >> 5) - wlo;
    const bool merged = nw <= LW;
    if (merged)
        for (u32 i = threadIdx.x; i < nw; i += 256) lbuf[i] = 0;
    __syncthreads();
    u32* gs = stream + (int64_t)f * stream_fs_words;
    if (j < g.nblk) {
    const int mcu = j / 6, k = j - mcu * 6, my = mcu / g.mw, mx = mcu - my * g.mw;
    const int16_t* dd = dcs + (int64_t)f * g.nblk;
    bool dummy;
    const int diff = block_dc(dd, g, mcu, mx, my, k, dummy) - block_pred(dd, g, mcu, mx, my, k);
    const int t = k >= 4 ? 1 : 0;
    const u32 off = offs[(int64_t)f * g.nblk + j];
    unsigned long long acc = 0;
    u32 nb = off & 31;
    u32 wi = off >> 5;
    bool first = true;
    auto put = [&](u32 code, u32 len) {
        acc |= (unsigned long long)code << (64 - nb - len);
        nb += len;
        if (nb >= 32) {
            if (merged) atomicOr(&lbuf[wi - wlo], (u32)(acc >> 32));
            else if (first) atomicOr(gs + wi, (u32)(acc >> 32));
            else gs[wi] = (u32)(acc >> 32);
            first = false;
            ++wi;
            acc <<= 32;
            nb -= 32;
        }
    };
    {
        const int sg = diff >> 31;
        const u32 cat = 32 - (u32)__clz((diff ^ sg) - sg);
        const u32 e = sdc[t][cat];
        put(((e & 0xffff) << cat) | ((u32)(diff + sg) & ((1u << cat) - 1)), (e >> 16) + cat);
    }
    if (!dummy) {
        const uint4* blk = (const uint4*)(coef + (int64_t)f * coef_fs) + (j >> 6) * 512 + (j & 63);
        const u32 zrl = sac[t][0xF0];
        u32 run = 0;
        uint4 nxt = blk[0];
        for (int g8 = 0; g8 < 8; ++g8) {
            const uint4 cur = nxt;
            if (g8 < 7) nxt = blk[(g8 + 1) * 64];
            const u32 pairs[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32 pair = pairs[u];
                const bool dcpair = g8 == 0 && u == 0;
                if ((dcpair ? pair >> 16 : pair) == 0) {          // most of a photograph's coefficients
                    run += dcpair ? 1 : 2;
                    continue;
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (dcpair && h == 0) continue;
                    const int c = (int)(int16_t)(pair >> (16 * h));
                    if (c == 0) {
                        ++run;
                        continue;
                    }
                    for (u32 z = run >> 4; z > 0; --z) put(zrl & 0xffff, zrl >> 16);
                    const int sg = c >> 31;
                    const u32 cat = 32 - (u32)__clz((c ^ sg) - sg);
                    const u32 e = sac[t][((run & 15) << 4) | cat];
                    put(((e & 0xffff) << cat) | ((u32)(c + sg) & ((1u << cat) - 1)), (e >> 16) + cat);
                    run = 0;
                }
            }
        }
        if (run) {
            const u32 e = sac[t][0];
            put(e & 0xffff, e >> 16);
        }
    } else {
        const u32 e = sac[t][0];
        put(e & 0xffff, e >> 16);
    }
    if (nb) {
        if (merged) atomicOr(&lbuf[wi - wlo], (u32)(acc >> 32));
        else atomicOr(gs + wi, (u32)(acc >> 32));
    }
    }
    if (merged) {
        __syncthreads();
        for (u32 i = threadIdx.x; i < nw; i += 256) {
            const u32 v = lbuf[i];
            if (i == 0 || i + 1 == nw) {
                if (v) atomicOr(gs + wlo + i, v);
            } else {
                gs[wlo + i] = v;
            }
        }
    }
}

// ---- exclusive scan of u32 rows (in place), 1024 elements per workgroup ---------------------------------------------

__device__ __forceinline__ u32 wg_exclusive_scan(u32 v, u32* total) {          // 256 threads
    __shared__ u32 wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u32 x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    __syncthreads();
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    u32 basev = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < wv) basev += wsum[i];
        tot += wsum[i];
    }
    *total = tot;
    return basev + x - v;
}

__global__ __launch_bounds__(256) void scan_partials_kernel(const u32* __restrict__ data, int64_t fs, int len, u32* __restrict__ part,
                                                            int nparts) {
    const int f = blockIdx.y;
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const u32* p = data + (int64_t)f * fs;
    u32 s = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) s += (i0 + e < len) ? p[i0 + e] : 0u;
    u32 tot;
    wg_exclusive_scan(s, &tot);
    if (threadIdx.x == 0) part[(int64_t)f * nparts + blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void scan_spine_kernel(u32* __restrict__ part, int nparts, u32* __restrict__ totals) {
    const int f = blockIdx.x;
    u32* p = part + (int64_t)f * nparts;
    u32 carry = 0;
    for (int b = 0; b < nparts; b += 256) {
        const int i = b + threadIdx.x;
        const u32 v = i < nparts ? p[i] : 0u;
        u32 tot;
        const u32 ex = wg_exclusive_scan(v, &tot);
        if (i < nparts) p[i] = carry + ex;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[f] = carry;
}

__global__ __launch_bounds__(256) void scan_apply_kernel(u32* __restrict__ data, int64_t fs, int len, const u32* __restrict__ part,
                                                         int nparts) {
    const int f = blockIdx.y;
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    u32* p = data + (int64_t)f * fs;
    u32 v[4], s = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = (i0 + e < len) ? p[i0 + e] : 0u;
        s += v[e];
    }
    u32 tot;
    u32 ex = wg_exclusive_scan(s, &tot) + part[(int64_t)f * nparts + blockIdx.x];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (i0 + e < len) p[i0 + e] = ex;
        ex += v[e];
    }
}

static int scan_rows(u32* data, int64_t fs, int len, int n, u32* part, u32* totals, hipStream_t st) {
    const int nparts = (len + 1023) / 1024;
    hipLaunchKernelGGL(scan_partials_kernel, dim3((unsigned)nparts, (unsigned)n), dim3(256), 0, st, data, fs, len, part, nparts);
    hipLaunchKernelGGL(scan_spine_kernel, dim3((unsigned)n), dim3(256), 0, st, part, nparts, totals);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nparts, (unsigned)n), dim3(256), 0, st, data, fs, len, part, nparts);
    return launch_status();
}

// ---- byte stuffing and the file around the entropy-coded segment ----------------------------------------------------

// Before the emit kernel: the words it ORs into must start at zero.  A workgroup whose span is merged in LDS touches
// only its first and last word that way (everything between is stored whole); a span too long for LDS is cleared entirely.
__global__ __launch_bounds__(256) void jpeg_zero_kernel(u32* __restrict__ stream, int64_t fs_words, const u32* __restrict__ offs,
                                                        const u32* __restrict__ total_bits, int nblk) {
    const int f = blockIdx.y, j0 = blockIdx.x * 256;
    const u32 tb = total_bits[f];
    if (((unsigned long long)tb + 31) / 32 > (unsigned long long)fs_words) return;
    const int j1 = min(j0 + 256, nblk);
    const u32 sbit = offs[(int64_t)f * nblk + j0];
    const u32 ebit = j1 < nblk ? offs[(int64_t)f * nblk + j1] : tb;
    const u32 wlo = sbit >> 5, nw = ((ebit + 31) >> 5) - wlo;
    u32* gs = stream + (int64_t)f * fs_words + wlo;
    if (nw <= JLW) {
        if (threadIdx.x == 0) gs[0] = 0;
        if (threadIdx.x == 1) gs[nw - 1] = 0;
    } else {
        for (u32 i = threadIdx.x; i < nw; i += 256) gs[i] = 0;
    }
}

// The eight MSB-first words of chunk ci of a frame's unstuffed stream, bytes past the end cleared and the last byte
// completed with 1-bits (jchuff.c flush_bits).
__device__ __forceinline__ void chunk_words(const u32* __restrict__ w, int ci, int64_t nbytes, u32 tb, u32 (&ws)[8]) {
    const uint4 a = ((const uint4*)w)[ci * 2], b = ((const uint4*)w)[ci * 2 + 1];
    ws[0] = a.x; ws[1] = a.y; ws[2] = a.z; ws[3] = a.w;
    ws[4] = b.x; ws[5] = b.y; ws[6] = b.z; ws[7] = b.w;
    const int64_t left = nbytes - (int64_t)ci * JCHUNK;           // > 0
    if (left < JCHUNK) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int v = (int)left - 4 * e;                       // valid bytes of word e
            ws[e] = v >= 4 ? ws[e] : (v <= 0 ? 0u : ws[e] & (0xffffffffu << (32 - 8 * v)));
        }
    }
    if ((tb & 7) && left <= JCHUNK) {                              // the stream's last byte lives in this chunk
        const int lb = (int)left - 1;
        const u32 pad = ((1u << (8 - (tb & 7))) - 1) << (24 - 8 * (lb & 3));
#pragma unroll
        for (int e = 0; e < 8; ++e) ws[e] |= (e == (lb >> 2)) ? pad : 0u;
    }
}

__device__ __forceinline__ u32 ff_bytes(u32 w) {                  // number of 0xFF bytes in a word
    u32 t = w & (w >> 4) & 0x0f0f0f0fu;
    t &= t >> 2;
    t &= t >> 1;
    return __popc(t & 0x01010101u);
}

__global__ __launch_bounds__(256) void jpeg_ffcount_kernel(const u32* __restrict__ stream, int64_t fs_words, const u32* __restrict__ total_bits,
                                                           u32* __restrict__ cnt, int64_t cnt_fs, int nchunks) {
    const int f = blockIdx.y;
    const u32 tb = total_bits[f];
    const bool over = ((unsigned long long)tb + 31) / 32 > (unsigned long long)fs_words;
    const int64_t nbytes = over ? 0 : ((int64_t)tb + 7) >> 3;
    const u32* w = stream + (int64_t)f * fs_words;
    for (int ci = blockIdx.x * 256 + threadIdx.x; ci < nchunks; ci += gridDim.x * 256) {   // the capacity, mostly unused
        u32 c = 0;
        if ((int64_t)ci * JCHUNK < nbytes) {
            u32 ws[8];
            chunk_words(w, ci, nbytes, tb, ws);
#pragma unroll
            for (int e = 0; e < 8; ++e) c += ff_bytes(ws[e]);
        }
        cnt[(int64_t)f * cnt_fs + ci] = c;
    }
}

// 256 chunks (8 KB of stream) per workgroup pass: every thread expands its chunk into LDS at its stuffed offset (byte
// writes), then the workgroup copies its contiguous piece of the file out — whole dwords where the piece covers them,
// single bytes at its two ends (the neighbouring workgroups own the rest of those dwords).
__global__ __launch_bounds__(256) void jpeg_stuff_kernel(const u32* __restrict__ stream, int64_t fs_words, const u32* __restrict__ total_bits,
                                                         const u32* __restrict__ cnt, int64_t cnt_fs, int nchunks,
                                                         const u32* __restrict__ ff_total, u8* __restrict__ out, int64_t out_fs,
                                                         u32* __restrict__ sizes, JpegHeader hd) {
    __shared__ __attribute__((aligned(4))) u8 lb[256 * 2 * JCHUNK + 8];
    const int f = blockIdx.y;
    const u32 tb = total_bits[f];
    const bool over = ((unsigned long long)tb + 31) / 32 > (unsigned long long)fs_words;
    const int64_t nbytes = ((int64_t)tb + 7) >> 3;
    const u32 nff = ff_total[f];
    const int64_t fsize = (int64_t)hd.len + nbytes + nff + 2;
    const bool fits = !over && fsize <= out_fs;
    u8* o = out + (int64_t)f * out_fs;
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) sizes[f] = fits ? (u32)fsize : 0xffffffffu;
        if (fits) {
            for (int i = threadIdx.x; i < hd.len; i += 256) o[i] = hd.b[i];
            if (threadIdx.x == 0) {
                o[fsize - 2] = 0xff;
                o[fsize - 1] = 0xd9;
            }
        }
    }
    if (!fits) return;
    const u32* w = stream + (int64_t)f * fs_words;
    const u32* cf = cnt + (int64_t)f * cnt_fs;
    const int nvc = (int)((nbytes + JCHUNK - 1) / JCHUNK);         // chunks that hold stream bytes
    for (int c0 = blockIdx.x * 256; c0 < nvc; c0 += gridDim.x * 256) {
        const int ce = min(c0 + 256, nvc);
        const u32 pre0 = cf[c0];
        const u32 pre1 = ce < nchunks ? cf[ce] : nff;
        u8* dst = o + hd.len + (int64_t)c0 * JCHUNK + pre0;        // where this pass's piece of the file starts
        const u32 mis = (u32)((uintptr_t)dst & 3);
        const u32 total = (u32)(min((int64_t)ce * JCHUNK, nbytes) - (int64_t)c0 * JCHUNK) + (pre1 - pre0);
        const int ci = c0 + threadIdx.x;
        if (ci < ce) {
            u32 ws[8];
            chunk_words(w, ci, nbytes, tb, ws);
            const int nv = (int)min((int64_t)JCHUNK, nbytes - (int64_t)ci * JCHUNK);
            u8* p = lb + mis + threadIdx.x * JCHUNK + (cf[ci] - pre0);
#pragma unroll
            for (int e = 0; e < JCHUNK; ++e) {
                if (e < nv) {
                    const u32 b = (ws[e >> 2] >> (24 - 8 * (e & 3))) & 255;
                    *p++ = (u8)b;
                    if (b == 255) *p++ = 0;
                }
            }
        }
        __syncthreads();
        u8* base = dst - mis;                                      // 4-byte aligned, LDS byte k ↔ base[k]
        const u32 end = mis + total;
        for (u32 k = threadIdx.x * 4; k < end; k += 1024) {
            if (k >= mis && k + 4 <= end) {
                *(u32*)(base + k) = *(const u32*)(lb + k);
            } else {
                for (u32 e = 0; e < 4; ++e)
                    if (k + e >= mis && k + e < end) base[k + e] = lb[k + e];
            }
        }
        __syncthreads();
    }
}

struct JpegLayout {
    int mw, mh, bw, bh, nblk, nparts_blk, nchunks, nparts_chunk;
    int64_t stream_words;                                  // per frame
    size_t off_coef, off_dcs, off_acb, off_lens, off_part, off_tot, off_stream, off_cnt, total;
};

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

static JpegLayout jpeg_layout(int n, int h, int w, size_t out_frame_stride) {
    JpegLayout L;
    L.mw = (w + 15) / 16;
    L.mh = (h + 15) / 16;
    L.bw = (w + 7) / 8;
    L.bh = (h + 7) / 8;
    L.nblk = L.mw * L.mh * 6;
    L.nparts_blk = (L.nblk + 1023) / 1024;
    L.stream_words = (int64_t)((out_frame_stride + 3) / 4 + 4) & ~(int64_t)3;
    L.nchunks = (int)((L.stream_words * 4 + JCHUNK - 1) / JCHUNK);
    L.nparts_chunk = (L.nchunks + 1023) / 1024;
    size_t o = 0;
    L.off_coef = o;   o += al256((size_t)n * (size_t)((L.nblk + 63) / 64) * 64 * 128);
    L.off_dcs = o;    o += al256((size_t)n * L.nblk * 2);
    L.off_acb = o;    o += al256((size_t)n * L.nblk * 2);
    L.off_lens = o;   o += al256((size_t)n * L.nblk * 4);
    L.off_part = o;   o += al256((size_t)n * (size_t)(L.nparts_blk > L.nparts_chunk ? L.nparts_blk : L.nparts_chunk) * 4);
    L.off_tot = o;    o += al256((size_t)n * 8);
    L.off_stream = o; o += al256((size_t)n * L.stream_words * 4);
    L.off_cnt = o;    o += al256((size_t)n * L.nchunks * 4);
    L.total = o;
    return L;
}

// exact for every |c| the DCT can produce (checked over 0 .. 65535 here): floor((a + d/2) / d), d = 8q
struct QuantMagic {
    u32 m[256], sh[256];
    bool ok[256];
    QuantMagic() {
        for (u32 qv = 1; qv < 256; ++qv) {
            const u32 d = qv * 8, half = d >> 1;
            u32 s = 0;
            while ((d << s) <= 256) ++s;
            const u32 dd = d << s;
            m[qv] = (u32)(((1ull << 32) + dd - 1) / dd);
            sh[qv] = s;
            ok[qv] = m[qv] < (1u << 24);
            for (u32 a = 0; a < 65536 && ok[qv]; ++a) {
                const u32 x = (a + half) << s;
                ok[qv] = x < (1u << 24) && (a + half) / d == (u32)(((unsigned long long)x * m[qv]) >> 32);
            }
        }
        m[0] = sh[0] = 0;
        ok[0] = false;
    }
};
static bool quant_entry(u32 qv, u32* m, u32* halfp) {
    static const QuantMagic magic;                          // built (and checked) once per process
    if (qv > 255 || !magic.ok[qv]) return false;
    *m = magic.m[qv];
    *halfp = (qv * 4) | (magic.sh[qv] << 16);
    return true;
}

} // namespace imgxf

using namespace imgxf;

IMGXF_API int imgxf_jpeg_workspace_bytes(int n, int h, int w, size_t out_frame_stride, size_t* bytes) {
    if (!bytes) return IMGXF_ERR_NULL;
    if (n < 0 || h < 1 || w < 1 || h > 32767 || w > 32767) return IMGXF_ERR_SHAPE;
    *bytes = jpeg_layout(n, h, w, out_frame_stride).total;
    return IMGXF_OK;
}

IMGXF_API int imgxf_jpeg_encode_u8(const imgxf_view* src, const imgxf_jpeg_tables* tables, const uint8_t* header,
                                   int header_bytes, uint8_t* out, size_t out_frame_stride, uint32_t* sizes,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    IMGXF_CHECK(check_view(src));
    if (!tables || !header || !out || !sizes) return IMGXF_ERR_NULL;
    if (src->c != 3) return IMGXF_ERR_UNSUPPORTED;
    if (header_bytes < 2 || header_bytes > 1024) return IMGXF_ERR_ARG;
    if (src->n == 0) return IMGXF_OK;
    if (empty_view(src)) return IMGXF_ERR_SHAPE;
    if (src->n > 65535) return IMGXF_ERR_SHAPE;
    if (out_frame_stride < (size_t)header_bytes + 2 || out_frame_stride > ((size_t)1 << 31)) return IMGXF_ERR_ARG;
    const JpegLayout L = jpeg_layout(src->n, src->h, src->w, out_frame_stride);
    if (!workspace || workspace_bytes < L.total || (((uintptr_t)workspace) & 15)) return IMGXF_ERR_WORKSPACE;
    if ((int64_t)L.nblk * 2048 > 0xfffffff0ll) return IMGXF_ERR_SHAPE;        // bit offsets are 32-bit
    JpegQuant q;
    for (int t = 0; t < 2; ++t)
        for (int i = 0; i < 64; ++i) {
            const u32 qv = tables->quant[t][i];
            if (qv < 1 || qv > 255 || !quant_entry(qv, &q.m[t][i], &q.half[t][i])) return IMGXF_ERR_ARG;
        }
    JpegHuff hf;
    for (int t = 0; t < 2; ++t) {
        for (int i = 0; i < 16; ++i) hf.dc[t][i] = (u32)tables->dc_code[t][i] | ((u32)tables->dc_len[t][i] << 16);
        for (int i = 0; i < 256; ++i) {
            hf.ac[t][i] = (u32)tables->ac_code[t][i] | ((u32)tables->ac_len[t][i] << 16);
            q.aclen[t][i] = (u8)(tables->ac_len[t][i] + (i & 15));      // code + magnitude bits of the symbol
        }
    }
    JpegHeader hd;
    memset(&hd, 0, sizeof(hd));
    memcpy(hd.b, header, (size_t)header_bytes);
    hd.len = header_bytes;
    const View s = make_view(src);
    hipStream_t st = (hipStream_t)stream;
    u8* ws = (u8*)workspace;
    int16_t* coef = (int16_t*)(ws + L.off_coef);
    int16_t* dcs = (int16_t*)(ws + L.off_dcs);
    uint16_t* acb = (uint16_t*)(ws + L.off_acb);
    u32* lens = (u32*)(ws + L.off_lens);
    u32* part = (u32*)(ws + L.off_part);
    u32* tot_bits = (u32*)(ws + L.off_tot);
    u32* tot_ff = tot_bits + s.n;
    u32* ustream = (u32*)(ws + L.off_stream);
    u32* cnt = (u32*)(ws + L.off_cnt);
    const JpegGeom g = {L.mw, L.mh, L.bw, L.bh, L.nblk};
    const int64_t coef_fs = (int64_t)((L.nblk + 63) / 64) * 64 * 64;          // int16 elements per frame, whole groups of 64 blocks
    hipLaunchKernelGGL(jpeg_transform_kernel, dim3((unsigned)((L.mw + JM - 1) / JM), (unsigned)L.mh, (unsigned)s.n), dim3(JT), 0, st,
                       s, coef, coef_fs, dcs, acb, L.nblk, L.mw, L.bw, L.bh, q);
    const dim3 bgrid((unsigned)((L.nblk + 255) / 256), (unsigned)s.n);
    hipLaunchKernelGGL(jpeg_lens_kernel, bgrid, dim3(256), 0, st, (const int16_t*)dcs, (const uint16_t*)acb, lens, g, hf);
    IMGXF_CHECK(scan_rows(lens, L.nblk, L.nblk, s.n, part, tot_bits, st));
    hipLaunchKernelGGL(jpeg_zero_kernel, bgrid, dim3(256), 0, st, ustream, L.stream_words, (const u32*)lens, (const u32*)tot_bits, L.nblk);
    hipLaunchKernelGGL(jpeg_emit_kernel, bgrid, dim3(256), 0, st, (const int16_t*)coef, coef_fs, (const int16_t*)dcs, (const u32*)lens,
                       ustream, L.stream_words, (const u32*)tot_bits, g, hf);
    const unsigned cwg = (unsigned)((L.nchunks + 255) / 256);
    const dim3 cgrid(cwg < 256u ? cwg : 256u, (unsigned)s.n);          // grid-stride over the capacity
    hipLaunchKernelGGL(jpeg_ffcount_kernel, cgrid, dim3(256), 0, st, (const u32*)ustream, L.stream_words, (const u32*)tot_bits, cnt,
                       (int64_t)L.nchunks, L.nchunks);
    IMGXF_CHECK(scan_rows(cnt, L.nchunks, L.nchunks, s.n, part, tot_ff, st));
    hipLaunchKernelGGL(jpeg_stuff_kernel, cgrid, dim3(256), 0, st, (const u32*)ustream, L.stream_words, (const u32*)tot_bits,
                       (const u32*)cnt, (int64_t)L.nchunks, L.nchunks, (const u32*)tot_ff, out, (int64_t)out_frame_stride, sizes, hd);
    return launch_status();
}
