// FIXED (8-bit fixed-point taps, round half up) instances of the sepconv kernels for C=4: register-marching fast path
// (sepconv_march.inc) when rows are 16-byte aligned and the halo fits one block,
// LDS-tiled general path (sepconv_tile.inc) otherwise.
#include "sepconv_march4.inc"
#include <stdlib.h>
namespace imgxf {
int sepconv_fx_c4(int R, const View& s, const View& d, const View& df, const Taps& taps,
               int border, hipStream_t st) {
    const int rpw_env = knob_int(K_MARCH_RPW, 0);
    const bool no_march = knob_set(K_NO_MARCH);
    if (!no_march && march_eligible(s, d, df, 4, R, border)) {
        switch (R) {
#define IMGXF_M(r) case r: return launch_sepconv_march<4, r, true>(s, d, df, taps, st, rpw_env);
            IMGXF_M(1) IMGXF_M(2) IMGXF_M(3)
#undef IMGXF_M
            default: break;
        }
    }
    bool sym = true;
    for (int i = 0; i < 2 * R + 1; ++i) sym = sym && taps.x[i] == taps.x[2 * R - i] && taps.x[i] == taps.y[i];
    if (!no_march && sym && march4_eligible(s, d, df, 4, R, border)) {
        switch (R) {
#define IMGXF_M4(r) case r: return launch_sepconv_march4<4, r, true>(s, d, df, taps, st);
            IMGXF_M4(4) IMGXF_M4(5) IMGXF_M4(6) IMGXF_M4(7) IMGXF_M4(8) IMGXF_M4(9) IMGXF_M4(10) IMGXF_M4(11) IMGXF_M4(12) IMGXF_M4(13) IMGXF_M4(14) IMGXF_M4(15)
#undef IMGXF_M4
            default: break;
        }
    }
    return dispatch_sepconv_tile<4, true>(R, s, d, df, taps, border, st);
}
} // namespace imgxf
