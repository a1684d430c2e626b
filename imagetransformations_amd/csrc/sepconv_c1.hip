// sepconv tile kernels for C=1 interleaved channels (see sepconv_tile.inc).
#include "sepconv_tile.inc"
namespace imgxf {
int sepconv_tile_c1(int R, const View& s, const View& d, const View& df, const Taps& taps,
                    int border, hipStream_t st) {
    return dispatch_sepconv_tile<1>(R, s, d, df, taps, border, st);
}
} // namespace imgxf
