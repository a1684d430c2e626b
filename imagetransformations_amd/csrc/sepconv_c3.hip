// sepconv kernels for C=3 interleaved channels: register-marching fast path
// (sepconv_march.inc) when rows are 16-byte aligned and the halo fits one block,
// LDS-tiled general path (sepconv_tile.inc) otherwise.
#include "sepconv_march4.inc"
#include "sepconv_mfma.inc"
#include <stdlib.h>
namespace imgxf {
int sepconv_c3(int R, const View& s, const View& d, const View& df, const Taps& taps,
               int border, hipStream_t st) {
    const int rpw_env = knob_int(K_MARCH_RPW, 0);
    const bool no_march = knob_set(K_NO_MARCH);
    if (!no_march && march_eligible(s, d, df, 3, R, border)) {
        switch (R) {
#define IMGXF_M(r) case r: return launch_sepconv_march<3, r>(s, d, df, taps, st, rpw_env);
            IMGXF_M(1) IMGXF_M(2) IMGXF_M(3) IMGXF_M(4)
#undef IMGXF_M
            default: break;
        }
    }
    // large radii: both passes on the matrix cores (sepconv_mfma.inc, sepconv_mfma2_rgb_kernel).  Its time hardly depends
    // on the radius (1.08 - 1.14 ms per 64 4K frames at k = 13 ... 21, 1.29 - 1.32 ms at k = 25 ... 31) while the vector
    // kernel grows with it (1.30 / 1.46 / 1.84 / 3.30 ms at k = 13 / 15 / 19 / 31; 0.85 ms at k = 9): R >= 6 goes to the
    // matrix cores (IMGXF_MFMA_MIN_R moves the threshold)
    const int mfma_min_r = knob_int(K_MFMA_MIN_R, 6);
    if (!no_march && R >= mfma_min_r && mfma_eligible(s, d, df, 3, R, border, taps)) {
        switch (R) {
#define IMGXF_MM(r) case r: return launch_sepconv_mfma<r>(s, d, df, taps, st);
            IMGXF_MM(2) IMGXF_MM(3) IMGXF_MM(4) IMGXF_MM(5) IMGXF_MM(6) IMGXF_MM(7) IMGXF_MM(8) IMGXF_MM(9) IMGXF_MM(10) IMGXF_MM(11) IMGXF_MM(12) IMGXF_MM(13) IMGXF_MM(14) IMGXF_MM(15)
#undef IMGXF_MM
            default: break;
        }
    }
    if (!no_march && march4_eligible(s, d, df, 3, R, border)) {
        switch (R) {
#define IMGXF_M4(r) case r: return launch_sepconv_march4<3, r>(s, d, df, taps, st);
            IMGXF_M4(5) IMGXF_M4(6) IMGXF_M4(7) IMGXF_M4(8) IMGXF_M4(9) IMGXF_M4(10) IMGXF_M4(11) IMGXF_M4(12) IMGXF_M4(13) IMGXF_M4(14) IMGXF_M4(15)
#undef IMGXF_M4
            default: break;
        }
    }
    return dispatch_sepconv_tile<3>(R, s, d, df, taps, border, st);
}
} // namespace imgxf
