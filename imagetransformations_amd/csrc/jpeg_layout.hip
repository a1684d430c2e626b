// HOST half of the device JPEG reader (no device work in this file): what jdmarker.c / jdhuff.c's table set-up do for
// libjpeg, for a whole batch of files at once — the marker segments up to the scan, the image and component descriptors
// the kernels read (imgxf_jpeg_dec_image), the quantisation tables in natural order, the derived Huffman tables
// (jpeg_make_d_derived_tbl; equal tables shared), and the entropy-coded bytes with the stuffing removed, split at the
// restart markers and laid out for the upload (imgxf_jpeg_unstuff_host's walk).
//
// The load step being replaced is `Image.open(path).convert("RGB")`, /root/reference/transformation.py:83.  The same logic
// lives in imagetransformations_amd/jpeg_decode.py (`parse`, `derive_lut`, `_segments`), which the tests hold this file
// against; in Python it costs 25 us per file, more than the device spends decoding a 375 x 500 file (7 us at 256 per batch).
#include "imgxf_common.h"
#include <string.h>
#include <vector>

namespace {

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,
                             7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                             39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffSpec { bool present = false; uint8_t bits[16]; uint8_t vals[256]; int nvals = 0; };

struct Parsed {
    int width = 0, height = 0, ncomp = 0;
    int cid[3], ch[3], cv[3], tq[3], td[3], ta[3];
    bool have_qt[4] = {false, false, false, false};
    uint16_t qt[4][64];
    HuffSpec huff[2][4];
    int dri = 0;
    size_t ecs_start = 0;
};

// jdmarker.c: the marker segments up to and including SOS.  Returns 0 or an IMGXF_JPEG_E_* code.
int parse_header(const uint8_t* d, size_t n, Parsed& P) {
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return IMGXF_JPEG_E_NOT_JPEG;
    size_t pos = 2;
    bool have_frame = false;
    for (;;) {
        if (pos + 4 > n || d[pos] != 0xFF) return IMGXF_JPEG_E_MARKERS;
        while (d[pos + 1] == 0xFF && pos + 2 < n) ++pos;                 // fill bytes
        if (pos + 4 > n) return IMGXF_JPEG_E_MARKERS;
        const int marker = d[pos + 1];
        const size_t seglen = ((size_t)d[pos + 2] << 8) | d[pos + 3];
        if (seglen < 2 || pos + 2 + seglen > n) return IMGXF_JPEG_E_MARKERS;
        const uint8_t* seg = d + pos + 4;
        const size_t sl = seglen - 2;
        if (marker == 0xDB) {
            size_t i = 0;
            while (i < sl) {
                const int pq = seg[i] >> 4, tq = seg[i] & 15;
                ++i;
                if (tq > 3 || i + (pq ? 128 : 64) > sl) return IMGXF_JPEG_E_MARKERS;
                for (int k = 0; k < 64; ++k)
                    P.qt[tq][kZigzag[k]] = pq ? (uint16_t)((seg[i + 2 * k] << 8) | seg[i + 2 * k + 1]) : seg[i + k];
                P.have_qt[tq] = true;
                i += pq ? 128 : 64;
            }
        } else if (marker == 0xC0 || marker == 0xC1) {
            if (sl < 6) return IMGXF_JPEG_E_MARKERS;
            if (seg[0] != 8) return IMGXF_JPEG_E_PRECISION;
            P.height = (seg[1] << 8) | seg[2]; P.width = (seg[3] << 8) | seg[4]; P.ncomp = seg[5];
            if (P.ncomp != 1 && P.ncomp != 3) return IMGXF_JPEG_E_COMPONENTS;
            if (sl < 6 + 3 * (size_t)P.ncomp) return IMGXF_JPEG_E_MARKERS;
            for (int k = 0; k < P.ncomp; ++k) {
                P.cid[k] = seg[6 + 3 * k]; P.ch[k] = seg[7 + 3 * k] >> 4; P.cv[k] = seg[7 + 3 * k] & 15; P.tq[k] = seg[8 + 3 * k];
            }
            have_frame = true;
        } else if (marker >= 0xC2 && marker <= 0xCF && marker != 0xC4 && marker != 0xC8 && marker != 0xCC) {
            return IMGXF_JPEG_E_PROCESS;                                 // progressive, lossless or arithmetic coding
        } else if (marker == 0xC4) {
            size_t i = 0;
            while (i < sl) {
                if (i + 17 > sl) return IMGXF_JPEG_E_MARKERS;
                const int tc = seg[i] >> 4, th = seg[i] & 15;
                int cnt = 0;
                for (int k = 0; k < 16; ++k) cnt += seg[i + 1 + k];
                if (tc > 1 || th > 3 || cnt > 256 || i + 17 + (size_t)cnt > sl) return IMGXF_JPEG_E_MARKERS;
                HuffSpec& H = P.huff[tc][th];
                H.present = true; H.nvals = cnt;
                memcpy(H.bits, seg + i + 1, 16);
                memset(H.vals, 0, sizeof(H.vals));
                memcpy(H.vals, seg + i + 17, (size_t)cnt);
                i += 17 + (size_t)cnt;
            }
        } else if (marker == 0xDD) {
            if (sl < 2) return IMGXF_JPEG_E_MARKERS;
            P.dri = (seg[0] << 8) | seg[1];
        } else if (marker == 0xDA) {
            if (!have_frame) return IMGXF_JPEG_E_MARKERS;
            if (sl < 1) return IMGXF_JPEG_E_MARKERS;
            const int ns = seg[0];
            if (ns != P.ncomp) return IMGXF_JPEG_E_COMPONENTS;           // non-interleaved scans are not read
            if (sl < 1 + 2 * (size_t)ns) return IMGXF_JPEG_E_MARKERS;
            for (int k = 0; k < ns; ++k) {
                if (seg[1 + 2 * k] != P.cid[k]) return IMGXF_JPEG_E_SCAN_ORDER;      // unknown component, or not in frame order
                P.td[k] = seg[2 + 2 * k] >> 4; P.ta[k] = seg[2 + 2 * k] & 15;
                if (P.td[k] > 3 || P.ta[k] > 3) return IMGXF_JPEG_E_MARKERS;
            }
            P.ecs_start = pos + 2 + seglen;
            return 0;
        }
        pos += 2 + seglen;
    }
}

// jdhuff.c jpeg_make_d_derived_tbl: 8-bit lookahead + maxcode / valoff for the longer codes
void derive_lut(const HuffSpec& H, imgxf_jpeg_dec_lut& L) {
    memset(&L, 0, sizeof(L));
    for (int i = 0; i < 18; ++i) L.maxcode[i] = -1;
    L.maxcode[17] = 0xFFFFF;
    int code = 0, k = 0;
    for (int length = 1; length <= 16; ++length) {
        const int cnt = H.bits[length - 1];
        if (cnt) {
            L.valoff[length] = k - code;
            if (length <= 8)
                for (int j = 0; j < cnt; ++j) {
                    const int first = (code + j) << (8 - length);
                    const uint16_t entry = (uint16_t)((length << 8) | H.vals[(k + j) & 255]);
                    for (int e = first; e < first + (1 << (8 - length)) && e < 256; ++e) L.look[e] = entry;
                }
            k += cnt;
            code += cnt;
            L.maxcode[length] = code - 1;
        }
        code <<= 1;
    }
    memcpy(L.huffval, H.vals, 256);
}

struct Geometry { int hmax, vmax, mcux, mcuy, ri, want; int ch[3], cv[3]; };

int geometry(const Parsed& P, Geometry& g) {
    for (int c = 0; c < P.ncomp; ++c) { g.ch[c] = P.ch[c]; g.cv[c] = P.cv[c]; }
    if (P.ncomp == 1) { g.ch[0] = 1; g.cv[0] = 1; }                      // a one-component scan is never interleaved
    g.hmax = 1; g.vmax = 1;
    for (int c = 0; c < P.ncomp; ++c) {
        if (g.ch[c] < 1 || g.ch[c] > 2 || g.cv[c] < 1 || g.cv[c] > 2) return IMGXF_JPEG_E_SAMPLING;
        if (g.ch[c] > g.hmax) g.hmax = g.ch[c];
        if (g.cv[c] > g.vmax) g.vmax = g.cv[c];
    }
    if (P.width < 1 || P.height < 1) return IMGXF_JPEG_E_MARKERS;
    g.mcux = (P.width + 8 * g.hmax - 1) / (8 * g.hmax);
    g.mcuy = (P.height + 8 * g.vmax - 1) / (8 * g.vmax);
    const int total = g.mcux * g.mcuy;
    g.ri = P.dri ? P.dri : total;
    g.want = (total + g.ri - 1) / g.ri;
    if (P.ncomp == 3) {
        const bool ok = g.ch[0] == g.hmax && g.cv[0] == g.vmax && g.ch[1] == g.ch[2] && g.cv[1] == g.cv[2] &&
                        (g.ch[1] * 2 == g.hmax || g.ch[1] == g.hmax) && (g.cv[1] * 2 == g.vmax || g.cv[1] == g.vmax) &&
                        !(g.ch[1] == g.hmax && g.cv[1] != g.vmax);
        if (!ok) return IMGXF_JPEG_E_CHROMA;                             // other than 4:4:4, 4:2:2 (h2v1), 4:2:0
    }
    return 0;
}

} // namespace

// One file's scan: stuffing removed, split at RSTn, padded segments appended to scan[] (include/imgxf.h).
IMGXF_API int imgxf_jpeg_unstuff_host(const uint8_t* data, size_t n, size_t start, uint8_t* scan, size_t scan_cap, size_t* scan_pos,
                                      int64_t* seg_off, int32_t* seg_len, int max_segs, int* nsegs, size_t* ecs_end) {
    if (!data || !scan || !scan_pos || !seg_off || !seg_len || !nsegs || !ecs_end) return IMGXF_ERR_NULL;
    if (start > n || max_segs < 1) return IMGXF_ERR_ARG;
    size_t pos = start, out = *scan_pos;
    int seg = 0;
    size_t seg_begin = out;
    bool keep = true;                                        // segments past max_segs are walked (for ecs_end) but not stored
    auto close_segment = [&]() -> int {
        if (!keep) return IMGXF_OK;
        const size_t len = out - seg_begin;
        const size_t pad = ((16 - (len & 15)) & 15) + 16;    // a refill may look a few bytes past a segment
        if (out + pad > scan_cap || len > 0x7fffffffu) return IMGXF_ERR_WORKSPACE;
        memset(scan + out, 0, pad);
        seg_off[seg] = (int64_t)seg_begin;
        seg_len[seg] = (int32_t)len;
        out += pad;
        ++seg;
        seg_begin = out;
        if (seg >= max_segs) keep = false;
        return IMGXF_OK;
    };
    for (;;) {
        const uint8_t* ff = pos < n ? (const uint8_t*)memchr(data + pos, 0xFF, n - pos) : nullptr;
        const size_t upto = ff ? (size_t)(ff - data) : n;    // plain bytes [pos, upto)
        const bool lone = ff && upto + 1 >= n;               // a 0xFF as the very last byte belongs to the scan
        const size_t take = upto - pos + (lone ? 1 : 0);
        if (keep && take) {
            if (out + take > scan_cap) return IMGXF_ERR_WORKSPACE;
            memcpy(scan + out, data + pos, take);
            out += take;
        }
        if (!ff || lone) { pos = n; break; }
        const uint8_t nxt = data[upto + 1];
        if (nxt == 0x00) {                                   // stuffed zero: keep the FF
            if (keep) { if (out + 1 > scan_cap) return IMGXF_ERR_WORKSPACE; scan[out++] = 0xFF; }
            pos = upto + 2;
        } else if (nxt >= 0xD0 && nxt <= 0xD7) {             // RSTn: next segment
            const int rc = close_segment();
            if (rc != IMGXF_OK) return rc;
            pos = upto + 2;
        } else { pos = upto; break; }                        // any other marker ends the scan
    }
    const int rc = close_segment();
    if (rc != IMGXF_OK) return rc;
    *scan_pos = out;
    *nsegs = seg;
    *ecs_end = pos;
    return IMGXF_OK;
}

// Pass 1 (scan == NULL): every file's header is parsed; *n_segs, *scan_cap (a bound), *n_quants, *n_luts (bounds) say what
// pass 2 needs.  Pass 2: everything is filled.  status[i]: 0 or the IMGXF_JPEG_E_* code of file i (the call itself returns
// IMGXF_OK; the caller raises for the first refused file).
IMGXF_API int imgxf_jpeg_layout_host(const uint8_t* const* files, const size_t* sizes, int n, imgxf_jpeg_dec_image* images,
                                     imgxf_jpeg_dec_lut* luts, int lut_cap, int* n_luts, uint16_t* quants, int quant_cap, int* n_quants,
                                     uint8_t* scan, size_t scan_cap, size_t* scan_bytes, int64_t* seg_off, int32_t* seg_len, int seg_cap,
                                     int* n_segs, int64_t* coef_total, int64_t* plane_total, int32_t* status) {
    if (n < 0) return IMGXF_ERR_ARG;
    if (!files || !sizes || !n_luts || !n_quants || !scan_bytes || !n_segs || !status) return IMGXF_ERR_NULL;
    const bool fill = scan != nullptr;
    if (fill && (!images || !luts || !quants || !seg_off || !seg_len || !coef_total || !plane_total)) return IMGXF_ERR_NULL;
    std::vector<HuffSpec> uniq;                                          // derived tables are shared between equal specifications
    int nq = 0, nseg = 0;
    size_t spos = 0, cap_bound = 0;
    int64_t coef_pos = 0, plane_pos = 0;
    Parsed P;
    for (int i = 0; i < n; ++i) {
        status[i] = 0;
        P = Parsed();
        int rc = files[i] ? parse_header(files[i], sizes[i], P) : IMGXF_JPEG_E_NOT_JPEG;
        Geometry g;
        if (!rc) rc = geometry(P, g);
        if (!rc)
            for (int c = 0; c < P.ncomp && !rc; ++c) {
                if (P.tq[c] > 3 || !P.have_qt[P.tq[c]]) rc = IMGXF_JPEG_E_NO_QUANT;
                else if (!P.huff[0][P.td[c]].present || !P.huff[1][P.ta[c]].present) rc = IMGXF_JPEG_E_NO_HUFF;
            }
        if (rc) { status[i] = rc; continue; }
        cap_bound += sizes[i] - P.ecs_start + 32 * ((size_t)g.want + 1);
        if (!fill) { nseg += g.want; nq += P.ncomp; continue; }
        imgxf_jpeg_dec_image& im = images[i];
        memset(&im, 0, sizeof(im));
        im.width = P.width; im.height = P.height; im.ncomp = P.ncomp; im.hmax = g.hmax; im.vmax = g.vmax; im.mcux = g.mcux; im.mcuy = g.mcuy;
        if (nseg + g.want > seg_cap) return IMGXF_ERR_WORKSPACE;
        int got = 0; size_t ecs_end = 0;
        rc = imgxf_jpeg_unstuff_host(files[i], sizes[i], P.ecs_start, scan, scan_cap, &spos, seg_off + nseg, seg_len + nseg, g.want, &got, &ecs_end);
        if (rc != IMGXF_OK) return rc;
        if (got < g.want) { status[i] = IMGXF_JPEG_E_TRUNCATED; continue; }
        im.restart_interval = g.ri; im.seg_first = nseg; im.seg_count = g.want;
        nseg += g.want;
        for (int c = 0; c < P.ncomp; ++c) {
            imgxf_jpeg_dec_comp& cp = im.comp[c];
            cp.h = g.ch[c]; cp.v = g.cv[c];
            int tabs[2];
            for (int cls = 0; cls < 2; ++cls) {
                const HuffSpec& H = P.huff[cls][cls ? P.ta[c] : P.td[c]];
                int idx = -1;
                for (size_t u = 0; u < uniq.size(); ++u)
                    if (!memcmp(uniq[u].bits, H.bits, 16) && uniq[u].nvals == H.nvals && !memcmp(uniq[u].vals, H.vals, (size_t)H.nvals)) { idx = (int)u; break; }
                if (idx < 0) {
                    if ((int)uniq.size() >= lut_cap) return IMGXF_ERR_WORKSPACE;
                    idx = (int)uniq.size();
                    uniq.push_back(H);
                    derive_lut(H, luts[idx]);
                }
                tabs[cls] = idx;
            }
            cp.dc_tab = tabs[0]; cp.ac_tab = tabs[1];
            if (nq >= quant_cap) return IMGXF_ERR_WORKSPACE;
            cp.quant = nq;
            memcpy(quants + (size_t)nq * 64, P.qt[P.tq[c]], 64 * sizeof(uint16_t));
            ++nq;
            cp.blocks_x = g.mcux * cp.h; cp.blocks_y = g.mcuy * cp.v;
            cp.dw = (P.width * cp.h + g.hmax - 1) / g.hmax; cp.dh = (P.height * cp.v + g.vmax - 1) / g.vmax;
            cp.coef_off = coef_pos; cp.plane_off = plane_pos;
            coef_pos += (int64_t)cp.blocks_x * cp.blocks_y * 64;
            plane_pos += (int64_t)cp.blocks_x * cp.blocks_y * 64;
        }
    }
    *n_segs = nseg; *n_quants = nq; *n_luts = fill ? (int)uniq.size() : 6 * n;
    *scan_bytes = fill ? spos : cap_bound;
    if (fill) { *coef_total = coef_pos; *plane_total = plane_pos; }
    return IMGXF_OK;
}
