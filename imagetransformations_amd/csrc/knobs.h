// Tuning / routing knobs of libimgxf, read from the environment ONCE (first use) into a table;
// `imgxf_reload_knobs()` re-reads them (tests and the A/B tools change knobs inside one process).
// A launch path reads a cached int or flag instead of scanning the environment.
#pragma once

#define IMGXF_KNOB_LIST(X)                                                                          \
    X(AFFINE_FPB) X(AFFINE_NO_DMA) X(AFFINE_NO_LDS) X(AFFINE_NO_SHEAR_FAST) X(AFFINE_NO_STRIPS)      \
    X(AFFINE_NO_TALL) X(AFFINE_PK3) X(AFFINE_TILE) X(AFFINE_MF_DBG) X(AFFINE_MF_NARROW) X(AFFINE_NO_WQ) X(AFFINE_MF_WIDE) X(BOX_BYTES) X(CONV2D_NO_SEPARABLE) \
    X(FILTER3X3_BYTES) X(FX_MFMA_MIN_R) X(LANCZOS_NO_LDS) X(LANCZOS_NO_V4) X(LANCZOS_SLOW)           \
    X(MARCH4_NO_PX) X(MARCH_GROUP) X(MARCH_NO_MIXED) X(MARCH_RPW) X(MARCH_SPB) X(MARCH_TAIL)         \
    X(MARCH_U2) X(MARCH_ORDER) X(MFMA2_BPC) X(MFMA_MIN_R) X(MFMA_NO_HREG) X(MFMA_SHAPE) X(MFMA_V1) X(MFMA_V3)   \
    X(NO_MARCH) X(RESAMPLE_MFMA_OC) X(RESAMPLE_MFMA_WAVES) X(RESAMPLE_NO_MFMA) X(NOISE_RNG)           \
    X(NO_FAST_LEFTOVERS) X(JPEG_SERIAL_HUFFMAN)

namespace imgxf {

enum Knob {
#define IMGXF_KNOB_ENUM(n) K_##n,
    IMGXF_KNOB_LIST(IMGXF_KNOB_ENUM)
#undef IMGXF_KNOB_ENUM
    K_COUNT
};

struct KnobTable {
    bool set[K_COUNT];
    int ival[K_COUNT];          // atoi of the value (0 when unset)
    char str[K_COUNT][32];      // the value itself, truncated ("" when unset)
};

const KnobTable& knob_table();                                   // api.hip
inline bool knob_set(Knob k) { return knob_table().set[k]; }
inline int knob_int(Knob k, int dflt) { const KnobTable& t = knob_table(); return t.set[k] ? t.ival[k] : dflt; }
inline const char* knob_str(Knob k) { const KnobTable& t = knob_table(); return t.set[k] ? t.str[k] : nullptr; }

} // namespace imgxf
