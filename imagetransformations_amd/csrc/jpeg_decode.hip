// Baseline JPEG reader on the device: the decode half of the reference's load step `Image.open(path).convert("RGB")`
// (/root/reference/transformation.py:83), bit-identical to Pillow / libjpeg-turbo with its defaults.
//
// The host (imagetransformations_amd/jpeg.py) parses the markers, removes the byte stuffing, splits the scan at RSTn
// markers and derives the decoding tables; three kernels do the rest:
//
//   jpeg_huff_kernel    entropy decoding (jdhuff.c decode_mcu): one THREAD per restart segment, one workgroup per image.
//                       Huffman decoding is the serial direction: a file without restart markers (every file Pillow
//                       writes by default, every ImageNet file) is ONE segment, so the parallelism of this stage is the
//                       number of files in the batch.  8-bit lookahead tables in LDS, canonical maxcode / valoff walk for
//                       longer codes, a 64-bit bit buffer refilled four bytes at a time, coefficients scattered to their
//                       natural-order slots (the buffer is zero on entry).
//   jpeg_idct_kernel    dequantisation + jidctint.c jpeg_idct_islow: 8 threads per block (a column each, then a row each,
//                       through LDS), the masked range-limit table as arithmetic.
//   jpeg_color_kernel   jdsample.c fullsize / h2v1_fancy / h2v2_fancy upsampling (edge replication as jdmainct.c does)
//                       + jdcolor.c ycc_rgb_convert in its 16-bit fixed point, or gray -> RGB; 4 pixels per thread.
#include "imgxf_common.h"
#include <string.h>

namespace imgxf {

__constant__ u8 kDecNatToZig[64] = {0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};      // natural position -> zigzag index
typedef uint32_t u32_una __attribute__((aligned(1)));

struct BitReader {
    const u8* p;            // segment bytes (stuffing removed; the host pads every segment with >= 16 zero bytes)
    int len, pos;           // bytes; position of `nextw`
    uint64_t acc;           // next bits, left aligned
    int nb;                 // valid bits in acc
    u32 nextw;              // the four bytes at `pos`, already loaded: the load of the following word is issued when this
                            // one is consumed, so its latency overlaps the decoding of ~6 symbols instead of stalling them
    __device__ __forceinline__ u32 load(int at) const {
        return __builtin_bswap32(*(const u32_una*)(p + min(at, len + 8)));      // (beyond the data: the zero padding)
    }
    __device__ __forceinline__ void start(const u8* ptr, int n) {
        p = ptr; len = n; pos = 0; acc = 0; nb = 0; nextw = load(0);
    }
    // start at bit `bit` of the segment (the parallel decoder's subsequences); consumed() = the bit the next symbol starts at
    __device__ __forceinline__ void start_at(const u8* ptr, int n, u32 bit) {
        p = ptr; len = n;
        const int q = (int)(bit >> 5) * 4, r = (int)(bit & 31u);
        acc = (uint64_t)load(q) << (32 + r);
        nb = 32 - r; pos = q + 4; nextw = load(pos);
    }
    __device__ __forceinline__ u32 consumed() const { return (u32)pos * 8u - (u32)nb; }
    __device__ __forceinline__ void refill() {
        while (nb <= 32) {
            acc |= (uint64_t)nextw << (32 - nb);
            nb += 32; pos += 4;
            nextw = load(pos);
        }
    }
    __device__ __forceinline__ u32 peek(int n) const { return (u32)(acc >> (64 - n)); }
    __device__ __forceinline__ void skip(int n) { acc <<= n; nb -= n; }
};

// the part of a table the canonical walk needs (codes longer than 8 bits), kept in LDS: in global memory every step of
// the walk was two dependent memory round trips on the one busy lane
struct HuffWalk { int32_t maxcode[18]; int32_t valoff[17]; uint8_t huffval[256]; };

// the parallel decoder's parameters (jpeg_huff_par_kernel, further down)
constexpr int PAR_BITS = 1024;                 // bits per subsequence
constexpr int PAR_MIN_BYTES = 2048;            // images with shorter segments on average keep one lane per segment

struct ParState { u32 p; u32 bk; };            // next symbol at bit p; bk = block-in-MCU * 64 + coefficient index (0: DC next)

// Which kernel decodes an image (wave-uniform; every kernel evaluates it and leaves the other classes alone).  The image's
// first and last segment say how long its segments are (L bytes):
//   0  a LANE per segment (jpeg_huff_kernel): short segments.  Costs ceil(segments / 64) x L x 0.45 us (measured: 8.5 ms for
//      a 19 KB scan);
//   1  a WAVE per segment (jpeg_huff_par_kernel<64, true>, PERSEG_SLOTS workgroups per image taking its segments in turn):
//      several segments of 2 .. 16 KB — a 4K file with a restart marker per MCU row is 135 x 6 KB;
//   2  a WORKGROUP of 256 per image, its segments one after the other, 256 subsequences per chunk (~0.85 ms per chunk however
//      few of its threads have work): one or a few segments up to 64 KB — every ImageNet-size file without restart markers;
//   3  a workgroup of 1024 per image: longer segments (a 4K scan of 810 KB is 7 chunks instead of 26).
constexpr int PERSEG_SLOTS = 32;
enum { HUFF_LANES = 0, HUFF_WAVE_PER_SEGMENT = 1, HUFF_WG256 = 2, HUFF_WG1024 = 3 };

__device__ __forceinline__ int huff_class(const imgxf_jpeg_dec_image& im, const int32_t* seg_len) {
    const int64_t a = seg_len[im.seg_first], b = seg_len[im.seg_first + im.seg_count - 1];
    const int64_t L = (a + b) / 2;
    if (L < PAR_MIN_BYTES) return HUFF_LANES;
    if (L > 2 * 256 * (PAR_BITS / 8)) return HUFF_WG1024;
    if (im.seg_count >= 2 && L <= 16384) return HUFF_WAVE_PER_SEGMENT;
    const int64_t chunks = (L * 8 / PAR_BITS + 255) / 256 + 1;
    return (int64_t)im.seg_count * chunks * 1900 < (int64_t)((im.seg_count + 63) / 64) * L ? HUFF_WG256 : HUFF_LANES;
}

// one Huffman symbol: 8-bit lookahead, then the canonical walk of jdhuff.c (jpeg_huff_decode), both in LDS
__device__ __forceinline__ int huff_symbol(BitReader& br, const uint16_t* look, const HuffWalk* lut, bool& bad) {
    const u32 e = look[br.peek(8)];
    if (e) { br.skip((int)(e >> 8)); return (int)(e & 0xffu); }
    for (int l = 9; l <= 16; ++l) {
        const int code = (int)br.peek(l);
        if (code <= lut->maxcode[l]) { br.skip(l); return lut->huffval[(code + lut->valoff[l]) & 0xff]; }
    }
    bad = true;
    br.skip(16);
    return 0;
}

__global__ __launch_bounds__(64) void jpeg_huff_kernel(const u8* __restrict__ scan, const int64_t* __restrict__ seg_off,
                                                       const int32_t* __restrict__ seg_len, const imgxf_jpeg_dec_image* __restrict__ images,
                                                       const imgxf_jpeg_dec_lut* __restrict__ luts, int16_t* __restrict__ coefs,
                                                       int32_t* __restrict__ status, int serial_only) {
    __shared__ uint16_t look[6][256];
    __shared__ HuffWalk walk[6];
    // The image descriptor lives in LDS: a __constant__ / global read per coefficient is a memory round trip on the one busy
    // lane.  Coefficients are stored in ZIGZAG order (the order of the stream): mapping k to its natural position here was a
    // second LDS round trip per symbol on that lane's serial path; jpeg_idct_kernel, which has a thread per coefficient
    // column, undoes the order when it reads.
    __shared__ imgxf_jpeg_dec_image im_s;
    for (int i = threadIdx.x; i < (int)(sizeof(imgxf_jpeg_dec_image) / 4); i += 64) ((u32*)&im_s)[i] = ((const u32*)(images + blockIdx.x))[i];
    __syncthreads();
    const imgxf_jpeg_dec_image& im = im_s;
    for (int i = threadIdx.x; i < 6 * 256; i += 64) {
        const int slot = i >> 8, c = slot >> 1;
        if (c < im.ncomp) look[slot][i & 255] = luts[(slot & 1) ? im.comp[c].ac_tab : im.comp[c].dc_tab].look[i & 255];
    }
    for (int i = threadIdx.x; i < 6 * 256; i += 64) {
        const int slot = i >> 8, c = slot >> 1, j = i & 255;
        if (c >= im.ncomp) continue;
        const imgxf_jpeg_dec_lut& L = luts[(slot & 1) ? im.comp[c].ac_tab : im.comp[c].dc_tab];
        walk[slot].huffval[j] = L.huffval[j];
        if (j < 18) walk[slot].maxcode[j] = L.maxcode[j];
        if (j < 17) walk[slot].valoff[j] = L.valoff[j];
    }
    __syncthreads();
    if (!serial_only && huff_class(im, seg_len) != HUFF_LANES) return;   // (uniform) jpeg_huff_par_kernel takes this image
    const int total = im.mcux * im.mcuy;
    bool bad = false;
    for (int s = threadIdx.x; s < im.seg_count; s += 64) {
        BitReader br;
        br.start(scan + seg_off[im.seg_first + s], seg_len[im.seg_first + s]);
        int pred[3] = {0, 0, 0};
        const int m0 = s * im.restart_interval, m1 = min(total, m0 + im.restart_interval);
        int my = m0 / im.mcux, mx = m0 - my * im.mcux;
        for (int m = m0; m < m1; ++m) {
            for (int c = 0; c < im.ncomp; ++c) {
                const imgxf_jpeg_dec_comp& cp = im.comp[c];
                const HuffWalk* ldc = &walk[2 * c];
                const HuffWalk* lac = &walk[2 * c + 1];
                for (int by = 0; by < cp.v; ++by)
                    for (int bx = 0; bx < cp.h; ++bx) {
                        int16_t* blk = coefs + cp.coef_off + ((int64_t)(my * cp.v + by) * cp.blocks_x + (mx * cp.h + bx)) * 64;
                        br.refill();
                        int sz = huff_symbol(br, look[2 * c], ldc, bad) & 15;
                        if (sz) {
                            br.refill();
                            int v = (int)br.peek(sz); br.skip(sz);
                            if (v < (1 << (sz - 1))) v -= (1 << sz) - 1;
                            pred[c] += v;
                        }
                        if (pred[c]) blk[0] = (int16_t)pred[c];
                        for (int k = 1; k < 64;) {
                            br.refill();
                            const int rs = huff_symbol(br, look[2 * c + 1], lac, bad);
                            const int r = rs >> 4, sz2 = rs & 15;
                            if (sz2 == 0) {
                                if (r == 15) { k += 16; continue; }
                                break;                                      // EOB
                            }
                            k += r;
                            int v = (int)br.peek(sz2); br.skip(sz2);
                            if (v < (1 << (sz2 - 1))) v -= (1 << sz2) - 1;
                            blk[k & 63] = (int16_t)v;               // zigzag position (k & 63: a corrupt run cannot leave the block)
                            ++k;
                        }
                        if (br.pos > br.len + 16) bad = true;               // ran past the data: stop believing it
                    }
            }
            if (bad) break;                                                 // (every loop above is bounded; a bad stream ends early)
            if (++mx == im.mcux) { mx = 0; ++my; }
        }
    }
    if (bad && status) atomicOr(status + blockIdx.x, 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// Entropy decoding INSIDE a segment in parallel (round 3, late): images whose restart segments are long — every file
// without restart markers — are decoded by a whole workgroup each.  A Huffman stream synchronises itself: decoding from
// a wrong bit with a wrong guess of the position inside the MCU falls in step with the true decoding after a few symbols
// (Klein & Wiseman; Weissenberger & Schmidt, "Accelerating JPEG decompression on GPUs").  The segment is cut into
// subsequences of PAR_BITS bits, one per thread, 256 at a time:
//   round 0   every thread decodes its subsequence from its first bit, guessing "first block of an MCU, DC next"; the
//             first thread of the chunk starts from the KNOWN state.  Each leaves its exit state (bit, block-in-MCU,
//             coefficient index) for its right neighbour and the number of blocks it completed;
//   rounds    a thread whose left neighbour's exit differs from the state it started from decodes again from there;
//             until nobody changes.  The known prefix grows by at least one subsequence per round, so at most 256
//             rounds; in practice the guess is wrong for one or two subsequences and 2 - 4 rounds suffice;
//   output    an exclusive scan of the block counts gives every thread the number of the block it starts in; a last
//             decoding writes the coefficients (zigzag order, the DC DIFFERENCE in [0]);
//   DC        when the segment is done, a scan over its blocks per component turns the differences into values.
// Every decoding loop is bounded by its subsequence; garbage decoded past the end of the data never reaches memory
// (blocks beyond the segment's count are dropped) and invalid codes only count in the output pass.
// ---------------------------------------------------------------------------------------------------------------------
struct ParTables {
    const uint16_t (*look)[256];
    const HuffWalk* walk;
    const u8* comp_of_b;                        // block-in-MCU -> component
    int bpm;
};

// MODE 0: states and block count only.  MODE 1: write coefficients; `g` = number of the block the subsequence starts in,
// blocks >= G are dropped.
template <int MODE>
__device__ __forceinline__ ParState par_run(const u8* seg, int len, ParState st, u32 p_end, const ParTables& T, int& nblk,
                                            int g, int G, int m0, const imgxf_jpeg_dec_image& im, const u8* bx_of_b, const u8* by_of_b,
                                            int16_t* coefs, bool& bad) {
    BitReader br;
    br.start_at(seg, len, st.p);
    int b = (int)(st.bk >> 6), k = (int)(st.bk & 63u);
    nblk = 0;
    int16_t* blk = nullptr;
    auto locate = [&]() {                       // MODE 1: address of block g
        if (g >= G) { blk = nullptr; return; }
        const int m = m0 + g / T.bpm, my = m / im.mcux, mx = m - my * im.mcux;
        const imgxf_jpeg_dec_comp& cp = im.comp[T.comp_of_b[b]];
        blk = coefs + cp.coef_off + ((int64_t)(my * cp.v + by_of_b[b]) * cp.blocks_x + (mx * cp.h + bx_of_b[b])) * 64;
    };
    if (MODE == 1) locate();
    while (br.consumed() < p_end) {
        br.refill();
        const int c = T.comp_of_b[b];
        bool lbad = false;
        if (k == 0) {
            const int sz = huff_symbol(br, T.look[2 * c], &T.walk[2 * c], lbad) & 15;
            int v = 0;
            if (sz) {
                br.refill();
                v = (int)br.peek(sz); br.skip(sz);
                if (v < (1 << (sz - 1))) v -= (1 << sz) - 1;
            }
            if (MODE == 1 && blk && v) blk[0] = (int16_t)v;          // the difference; jpeg_huff_par_kernel's DC pass integrates
            k = 1;
        } else {
            const int rs = huff_symbol(br, T.look[2 * c + 1], &T.walk[2 * c + 1], lbad);
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
                k = (r == 15) ? k + 16 : 64;                         // ZRL / EOB
            } else {
                k += r;
                int v = (int)br.peek(sz); br.skip(sz);
                if (v < (1 << (sz - 1))) v -= (1 << sz) - 1;
                if (MODE == 1 && blk) blk[k & 63] = (int16_t)v;
                ++k;
            }
        }
        if (MODE == 1 && lbad && g < G) bad = true;
        if (k >= 64) {                                               // block complete
            k = 0; ++nblk; ++g;
            if (++b == T.bpm) b = 0;
            if (MODE == 1) locate();
        }
    }
    ParState ex; ex.p = br.consumed(); ex.bk = (u32)(b * 64 + k);
    return ex;
}

// NT = 256 threads for images whose segments fit two chunks of 256 subsequences (64 KB), 1024 threads for longer ones: a chunk
// costs the same ~0.85 ms whatever its width (it is rounds x one subsequence), so a 4K scan of 810 KB is 7 chunks instead of 26.
// PERSEG: the workgroup (one wave, NT = 64) is slot blockIdx.y of PERSEG_SLOTS for its image and takes segments blockIdx.y,
// blockIdx.y + PERSEG_SLOTS, ...
template <int NT, bool PERSEG>
__global__ __launch_bounds__(NT) void jpeg_huff_par_kernel(const u8* __restrict__ scan, const int64_t* __restrict__ seg_off,
                                                            const int32_t* __restrict__ seg_len, const imgxf_jpeg_dec_image* __restrict__ images,
                                                            const imgxf_jpeg_dec_lut* __restrict__ luts, int16_t* __restrict__ coefs,
                                                            int32_t* __restrict__ status) {
    __shared__ uint16_t look[6][256];
    __shared__ HuffWalk walk[6];
    __shared__ imgxf_jpeg_dec_image im_s;
    __shared__ u8 comp_of_b[12], bx_of_b[12], by_of_b[12];
    __shared__ u32 cand_p[NT + 1], cand_bk[NT + 1];
    __shared__ int cnt[NT];
    constexpr int NW = NT / 64;
    __shared__ int wsum[3][NW];
    const int tid = threadIdx.x;
    for (int i = tid; i < (int)(sizeof(imgxf_jpeg_dec_image) / 4); i += NT) ((u32*)&im_s)[i] = ((const u32*)(images + blockIdx.x))[i];
    __syncthreads();
    const imgxf_jpeg_dec_image& im = im_s;
    if (huff_class(im, seg_len) != (PERSEG ? HUFF_WAVE_PER_SEGMENT : (NT == 1024 ? HUFF_WG1024 : HUFF_WG256))) return;    // (uniform) another kernel takes this image
    for (int i = tid; i < 6 * 256; i += NT) {
        const int slot = i >> 8, c = slot >> 1, j = i & 255;
        if (c >= im.ncomp) continue;
        const imgxf_jpeg_dec_lut& L = luts[(slot & 1) ? im.comp[c].ac_tab : im.comp[c].dc_tab];
        look[slot][j] = L.look[j];
        walk[slot].huffval[j] = L.huffval[j];
        if (j < 18) walk[slot].maxcode[j] = L.maxcode[j];
        if (j < 17) walk[slot].valoff[j] = L.valoff[j];
    }
    int bpm = 0;
    for (int c = 0; c < im.ncomp; ++c) bpm += im.comp[c].h * im.comp[c].v;
    if (tid == 0) {
        int b = 0;
        for (int c = 0; c < im.ncomp; ++c)
            for (int by = 0; by < im.comp[c].v; ++by)
                for (int bx = 0; bx < im.comp[c].h; ++bx) { comp_of_b[b] = (u8)c; bx_of_b[b] = (u8)bx; by_of_b[b] = (u8)by; ++b; }
    }
    __syncthreads();
    ParTables T; T.look = look; T.walk = walk; T.comp_of_b = comp_of_b; T.bpm = bpm;
    const int total = im.mcux * im.mcuy;
    bool bad = false;
    for (int sgi = PERSEG ? (int)blockIdx.y : 0; sgi < im.seg_count; sgi += PERSEG ? PERSEG_SLOTS : 1) {    // (uniform) the image's restart segments, one after the other
        const u8* seg = scan + seg_off[im.seg_first + sgi];
        const int len = seg_len[im.seg_first + sgi];
        const u32 total_bits = (u32)len * 8u;
        const int m0 = sgi * im.restart_interval, m1 = min(total, m0 + im.restart_interval);
        const int G = (m1 - m0) * bpm;
        const int nsub = (int)((total_bits + PAR_BITS - 1) / PAR_BITS);
        ParState carry; carry.p = 0; carry.bk = 0;
        int gbase = 0;
        for (int c0 = 0; c0 < nsub; c0 += NT) {                      // (uniform) NT subsequences at a time
            const int i = c0 + tid;
            const bool active = i < nsub;
            const u32 p_end = min((u32)(i + 1) * PAR_BITS, total_bits);
            ParState used; used.p = (u32)i * PAR_BITS; used.bk = 0;
            if (tid == 0) used = carry;
            ParState ex = used; int nb = 0;
            if (active) ex = par_run<0>(seg, len, used, p_end, T, nb, 0, 0, 0, im, bx_of_b, by_of_b, coefs, bad);
            cand_p[tid + 1] = ex.p; cand_bk[tid + 1] = ex.bk; cnt[tid] = active ? nb : 0;
            __syncthreads();
            for (int round = 0; round < NT; ++round) {              // (uniform) until every thread started from its left neighbour's exit
                bool changed = false;
                if (active && tid > 0) {
                    ParState c; c.p = cand_p[tid]; c.bk = cand_bk[tid];
                    if (c.p != used.p || c.bk != used.bk) {
                        used = c;
                        ex = par_run<0>(seg, len, used, p_end, T, nb, 0, 0, 0, im, bx_of_b, by_of_b, coefs, bad);
                        changed = true;
                    }
                }
                __syncthreads();                                     // every candidate has been read
                if (changed) { cand_p[tid + 1] = ex.p; cand_bk[tid + 1] = ex.bk; cnt[tid] = nb; }
                if (!__syncthreads_or(changed ? 1 : 0)) break;
            }
            // exclusive scan of the block counts: the block each subsequence starts in
            int v = cnt[tid], incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if ((tid & 63) >= d) incl += o; }
            if ((tid & 63) == 63) wsum[0][tid >> 6] = incl;
            __syncthreads();
            int wbase = 0;
            for (int w = 0; w < (tid >> 6); ++w) wbase += wsum[0][w];
            int chunk_blocks = 0;
            for (int w = 0; w < NW; ++w) chunk_blocks += wsum[0][w];
            const int gstart = gbase + wbase + incl - v;
            if (active) { int nb2; par_run<1>(seg, len, used, p_end, T, nb2, gstart, G, m0, im, bx_of_b, by_of_b, coefs, bad); }
            const int last = min(NT, nsub - c0);
            carry.p = cand_p[last]; carry.bk = cand_bk[last];
            gbase += chunk_blocks;
            __syncthreads();                                         // carry / wsum / cand are re-used by the next chunk
        }
        if (gbase < G) bad = true;                                   // the data ended before the segment's last block
        // DC: differences -> values, per component, in the segment's block order (the output pass's stores are this
        // workgroup's own: a barrier makes them visible)
        __threadfence_block();
        __syncthreads();
        int pred[3] = {0, 0, 0};
        for (int g0 = 0; g0 < G; g0 += NT) {                         // (uniform)
            const int g = g0 + tid;
            int16_t* blk = nullptr; int c = 0, d = 0;
            if (g < G) {
                const int b = g % bpm, m = m0 + g / bpm, my = m / im.mcux, mx = m - my * im.mcux;
                c = comp_of_b[b];
                const imgxf_jpeg_dec_comp& cp = im.comp[c];
                blk = coefs + cp.coef_off + ((int64_t)(my * cp.v + by_of_b[b]) * cp.blocks_x + (mx * cp.h + bx_of_b[b])) * 64;
                d = blk[0];
            }
            int inc[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                int x = (g < G && c == q) ? d : 0;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) { const int o = __shfl_up(x, dd, 64); if ((tid & 63) >= dd) x += o; }
                inc[q] = x;
                if ((tid & 63) == 63) wsum[q][tid >> 6] = x;
            }
            __syncthreads();
            int mine = 0;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                int base = pred[q];
                for (int w = 0; w < (tid >> 6); ++w) base += wsum[q][w];
                if (c == q) mine = base + inc[q];
                for (int w = 0; w < NW; ++w) pred[q] += wsum[q][w];
            }
            if (blk && mine != d) blk[0] = (int16_t)mine;
            __syncthreads();
        }
    }
    if (bad && status) atomicOr(status + blockIdx.x, 1);
}

// ---- jidctint.c jpeg_idct_islow, one dimension (CONST_BITS = 13) ---------------------------------------------------
__device__ __forceinline__ void idct8(const int (&x)[8], int (&o)[8], int shift) {
    constexpr int F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633,
                  F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
    int z2 = x[2], z3 = x[6];
    int z1 = (z2 + z3) * F_0_541;
    int tmp2 = z1 + z3 * (-F_1_847);
    int tmp3 = z1 + z2 * F_0_765;
    int tmp0 = (x[0] + x[4]) * 8192, tmp1 = (x[0] - x[4]) * 8192;            // << CONST_BITS (written as a multiply: no UB on negatives)
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = x[7]; tmp1 = x[5]; tmp2 = x[3]; tmp3 = x[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * F_1_175;
    tmp0 *= F_0_298; tmp1 *= F_2_053; tmp2 *= F_3_072; tmp3 *= F_1_501;
    z1 *= -F_0_899; z2 *= -F_2_562; z3 = z3 * (-F_1_961) + z5; z4 = z4 * (-F_0_390) + z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    const int rnd = 1 << (shift - 1);
    o[0] = (tmp10 + tmp3 + rnd) >> shift; o[7] = (tmp10 - tmp3 + rnd) >> shift;
    o[1] = (tmp11 + tmp2 + rnd) >> shift; o[6] = (tmp11 - tmp2 + rnd) >> shift;
    o[2] = (tmp12 + tmp1 + rnd) >> shift; o[5] = (tmp12 - tmp1 + rnd) >> shift;
    o[3] = (tmp13 + tmp0 + rnd) >> shift; o[4] = (tmp13 - tmp0 + rnd) >> shift;
}

// sample_range_limit + CENTERJSAMPLE indexed with (x & RANGE_MASK) (jdmaster.c prepare_range_limit_table)
__device__ __forceinline__ u32 range_limit_centered(int x) {
    const int i = x & 1023;
    return (u32)(i < 128 ? i + 128 : (i < 512 ? 255 : (i < 896 ? 0 : i - 896)));
}

__global__ __launch_bounds__(256) void jpeg_idct_kernel(const int16_t* __restrict__ coefs, const imgxf_jpeg_dec_image* __restrict__ images,
                                                        const uint16_t* __restrict__ quants, u8* __restrict__ planes) {
    __shared__ int ws[32][8][9];
    const imgxf_jpeg_dec_image& im = images[blockIdx.y];
    const int lb = threadIdx.x >> 3, t = threadIdx.x & 7;
    int g = blockIdx.x * 32 + lb, c = 0;
    bool live = false;
    for (; c < im.ncomp; ++c) {
        const int nb = im.comp[c].blocks_x * im.comp[c].blocks_y;
        if (g < nb) { live = true; break; }
        g -= nb;
    }
    const imgxf_jpeg_dec_comp& cp = im.comp[live ? c : 0];
    if (live) {                                                             // pass 1: column t of block g
        const int16_t* blk = coefs + cp.coef_off + (int64_t)g * 64;
        const uint16_t* q = quants + cp.quant * 64;
        int x[8], o[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) x[r] = (int)blk[kDecNatToZig[r * 8 + t]] * (int)q[r * 8 + t];      // coefficients arrive in zigzag order
        idct8(x, o, 13 - 2);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[lb][r][t] = o[r];
    }
    __syncthreads();
    if (live) {                                                             // pass 2: row t
        int x[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = ws[lb][t][k];
        idct8(x, o, 13 + 2 + 3);
        const int by = g / cp.blocks_x, bx = g - by * cp.blocks_x;
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { lo |= range_limit_centered(o[k]) << (8 * k); hi |= range_limit_centered(o[k + 4]) << (8 * k); }
        uint2* dst = (uint2*)(planes + cp.plane_off + (int64_t)(by * 8 + t) * (cp.blocks_x * 8) + bx * 8);
        *dst = make_uint2(lo, hi);                                          // (plane_off and the pitch are multiples of 8)
    }
}

// chroma sample at full-resolution position (x, y): jdsample.c
__device__ __forceinline__ int chroma_at(const u8* pl, const imgxf_jpeg_dec_comp& cp, int pitch, int hmax, int vmax, int x, int y) {
    if (cp.h == hmax && cp.v == vmax) return pl[(int64_t)y * pitch + x];
    const int i = x >> 1;
    // jinit_upsampler: the fancy (triangle) filters only for downsampled_width > 2; narrower components are replicated
    if (cp.dw <= 2) return pl[(int64_t)(cp.v == vmax ? y : y >> 1) * pitch + i];
    if (cp.v == vmax) {                                                     // h2v1_fancy_upsample
        const u8* row = pl + (int64_t)y * pitch;
        const int cur = row[i];
        if (x & 1) return i == cp.dw - 1 ? cur : (3 * cur + row[i + 1] + 2) >> 2;
        return i == 0 ? cur : (3 * cur + row[i - 1] + 1) >> 2;
    }
    // h2v2_fancy_upsample: the nearer row counts 3, the farther 1; rows beyond the component are its edge rows
    const int r = y >> 1;
    const int nr = (y & 1) ? min(r + 1, cp.dh - 1) : max(r - 1, 0);
    const u8* r0 = pl + (int64_t)r * pitch;
    const u8* r1 = pl + (int64_t)nr * pitch;
    const int cs = 3 * r0[i] + r1[i];
    if (x & 1) return i == cp.dw - 1 ? (cs * 4 + 7) >> 4 : (3 * cs + (3 * r0[i + 1] + r1[i + 1]) + 7) >> 4;
    return i == 0 ? (cs * 4 + 8) >> 4 : (3 * cs + (3 * r0[i - 1] + r1[i - 1]) + 8) >> 4;
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(const u8* __restrict__ planes, const imgxf_jpeg_dec_image* __restrict__ images,
                                                         u8* __restrict__ out) {
    const imgxf_jpeg_dec_image& im = images[blockIdx.y];
    const int gw = (im.width + 3) >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)gw * im.height) return;
    const int y = (int)(idx / gw), x0 = (int)(idx - (int64_t)y * gw) * 4;
    const int npx = min(4, im.width - x0);
    const imgxf_jpeg_dec_comp& c0 = im.comp[0];
    const u8* py = planes + c0.plane_off + (int64_t)y * (c0.blocks_x * 8) + x0;
    u8 px[12];
    if (im.ncomp == 1) {
        for (int j = 0; j < npx; ++j) { px[3 * j] = py[j]; px[3 * j + 1] = py[j]; px[3 * j + 2] = py[j]; }
    } else {
        const imgxf_jpeg_dec_comp& c1 = im.comp[1];
        const imgxf_jpeg_dec_comp& c2 = im.comp[2];
        const u8* pb = planes + c1.plane_off;
        const u8* pr = planes + c2.plane_off;
        for (int j = 0; j < npx; ++j) {
            const int yy = py[j];
            const int cb = chroma_at(pb, c1, c1.blocks_x * 8, im.hmax, im.vmax, x0 + j, y) - 128;
            const int cr = chroma_at(pr, c2, c2.blocks_x * 8, im.hmax, im.vmax, x0 + j, y) - 128;
            // jdcolor.c build_ycc_rgb_table: FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554
            const int r = yy + ((91881 * cr + 32768) >> 16);
            const int g = yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
            const int b = yy + ((116130 * cb + 32768) >> 16);
            px[3 * j] = (u8)min(max(r, 0), 255); px[3 * j + 1] = (u8)min(max(g, 0), 255); px[3 * j + 2] = (u8)min(max(b, 0), 255);
        }
    }
    u8* dst = out + im.out_off + (int64_t)y * im.out_pitch + (int64_t)x0 * 3;
    if (npx == 4 && (((uintptr_t)dst) & 3) == 0) {
        u32* d4 = (u32*)dst;
        d4[0] = px[0] | (px[1] << 8) | (px[2] << 16) | ((u32)px[3] << 24);
        d4[1] = px[4] | (px[5] << 8) | (px[6] << 16) | ((u32)px[7] << 24);
        d4[2] = px[8] | (px[9] << 8) | (px[10] << 16) | ((u32)px[11] << 24);
    } else {
        for (int j = 0; j < 3 * npx; ++j) dst[j] = px[j];
    }
}

static int dec_check_host(const imgxf_jpeg_dec_image* host, int n, int64_t* max_blocks, int64_t* max_quads) {
    *max_blocks = 0; *max_quads = 0;
    for (int i = 0; i < n; ++i) {
        const imgxf_jpeg_dec_image& im = host[i];
        if (im.ncomp != 1 && im.ncomp != 3) return IMGXF_ERR_UNSUPPORTED;
        if (im.width < 1 || im.height < 1 || im.width > 65535 || im.height > 65535) return IMGXF_ERR_SHAPE;
        int64_t nb = 0;
        for (int c = 0; c < im.ncomp; ++c) {
            const imgxf_jpeg_dec_comp& cp = im.comp[c];
            if (cp.h < 1 || cp.h > 2 || cp.v < 1 || cp.v > 2 || cp.blocks_x < 1 || cp.blocks_y < 1) return IMGXF_ERR_UNSUPPORTED;
            if ((cp.plane_off & 7) != 0) return IMGXF_ERR_ARG;
            nb += (int64_t)cp.blocks_x * cp.blocks_y;
        }
        if (im.ncomp == 3) {
            const imgxf_jpeg_dec_comp& a = im.comp[0];
            if (a.h != im.hmax || a.v != im.vmax) return IMGXF_ERR_UNSUPPORTED;
            for (int c = 1; c < 3; ++c) {
                const imgxf_jpeg_dec_comp& cp = im.comp[c];
                const bool full = cp.h == im.hmax && cp.v == im.vmax, h2v1 = cp.h * 2 == im.hmax && cp.v == im.vmax,
                           h2v2 = cp.h * 2 == im.hmax && cp.v * 2 == im.vmax;
                if (!(full || h2v1 || h2v2)) return IMGXF_ERR_UNSUPPORTED;
            }
        }
        if (nb > *max_blocks) *max_blocks = nb;
        const int64_t quads = (int64_t)((im.width + 3) >> 2) * im.height;
        if (quads > *max_quads) *max_quads = quads;
    }
    return IMGXF_OK;
}

} // namespace imgxf

using namespace imgxf;

IMGXF_API int imgxf_jpeg_decode_huffman(const uint8_t* scan, const int64_t* seg_off, const int32_t* seg_len,
                                        const imgxf_jpeg_dec_image* images, int n, const imgxf_jpeg_dec_lut* luts,
                                        int16_t* coefs, int32_t* status, void* stream) {
    if (n < 0) return IMGXF_ERR_ARG;
    if (n == 0) return IMGXF_OK;
    if (!scan || !seg_off || !seg_len || !images || !luts || !coefs) return IMGXF_ERR_NULL;
    // images with long segments (no restart markers): a workgroup per image decodes inside the segment in parallel; the others
    // keep a lane per segment.  Both kernels look at every image and leave the other class alone.
    const int serial_only = knob_set(K_JPEG_SERIAL_HUFFMAN) ? 1 : 0;
    hipLaunchKernelGGL(jpeg_huff_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, scan, seg_off, seg_len, images, luts, coefs, status, serial_only);
    if (!serial_only)
    {
        hipLaunchKernelGGL((jpeg_huff_par_kernel<256, false>), dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, scan, seg_off, seg_len, images, luts, coefs, status);
        hipLaunchKernelGGL((jpeg_huff_par_kernel<1024, false>), dim3((unsigned)n), dim3(1024), 0, (hipStream_t)stream, scan, seg_off, seg_len, images, luts, coefs, status);
        hipLaunchKernelGGL((jpeg_huff_par_kernel<64, true>), dim3((unsigned)n, PERSEG_SLOTS), dim3(64), 0, (hipStream_t)stream, scan, seg_off, seg_len, images, luts, coefs, status);
    }
    return launch_status();
}

IMGXF_API int imgxf_jpeg_decode_idct(const int16_t* coefs, const imgxf_jpeg_dec_image* images, const imgxf_jpeg_dec_image* images_host,
                                     int n, const uint16_t* quants, uint8_t* planes, void* stream) {
    if (n < 0) return IMGXF_ERR_ARG;
    if (n == 0) return IMGXF_OK;
    if (!coefs || !images || !images_host || !quants || !planes) return IMGXF_ERR_NULL;
    if (n > 65535) return IMGXF_ERR_SHAPE;
    int64_t mb, mq;
    IMGXF_CHECK(dec_check_host(images_host, n, &mb, &mq));
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((mb + 31) / 32), (unsigned)n), dim3(256), 0, (hipStream_t)stream, coefs, images, quants, planes);
    return launch_status();
}

IMGXF_API int imgxf_jpeg_decode_color(const uint8_t* planes, const imgxf_jpeg_dec_image* images, const imgxf_jpeg_dec_image* images_host,
                                      int n, uint8_t* out, void* stream) {
    if (n < 0) return IMGXF_ERR_ARG;
    if (n == 0) return IMGXF_OK;
    if (!planes || !images || !images_host || !out) return IMGXF_ERR_NULL;
    if (n > 65535) return IMGXF_ERR_SHAPE;
    int64_t mb, mq;
    IMGXF_CHECK(dec_check_host(images_host, n, &mb, &mq));
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)((mq + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, planes, images, out);
    return launch_status();
}
