// Library-level entry points: version, error strings, device probe.
#include "imgxf_common.h"
#include <string.h>

IMGXF_API int imgxf_version(void) { return IMGXF_VERSION; }

IMGXF_API const char* imgxf_strerror(int code) {
    switch (code) {
        case IMGXF_OK: return "ok";
        case IMGXF_ERR_NULL: return "imgxf: required pointer is NULL";
        case IMGXF_ERR_SHAPE: return "imgxf: view shapes/strides are inconsistent";
        case IMGXF_ERR_ARG: return "imgxf: argument out of range";
        case IMGXF_ERR_UNSUPPORTED: return "imgxf: no kernel for this request";
        case IMGXF_ERR_WORKSPACE: return "imgxf: workspace too small";
        case IMGXF_ERR_NO_DEVICE: return "imgxf: no gfx950 device";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "imgxf: unknown error";
}

IMGXF_API int imgxf_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

// ---- knobs (knobs.h): the environment is scanned once, not on every launch ------------------------
#include <stdlib.h>
#include <mutex>
namespace imgxf {
static KnobTable g_knobs;
static std::once_flag g_knobs_once;
static const char* const kKnobEnv[K_COUNT] = {
#define IMGXF_KNOB_NAME(n) "IMGXF_" #n,
    IMGXF_KNOB_LIST(IMGXF_KNOB_NAME)
#undef IMGXF_KNOB_NAME
};
static void load_knobs() {
    for (int k = 0; k < K_COUNT; ++k) {
        const char* v = getenv(kKnobEnv[k]);
        g_knobs.set[k] = v != nullptr;
        g_knobs.ival[k] = v ? atoi(v) : 0;
        g_knobs.str[k][0] = 0;
        if (v) { strncpy(g_knobs.str[k], v, sizeof(g_knobs.str[k]) - 1); g_knobs.str[k][sizeof(g_knobs.str[k]) - 1] = 0; }
    }
}
const KnobTable& knob_table() {
    std::call_once(g_knobs_once, load_knobs);
    return g_knobs;
}
} // namespace imgxf

IMGXF_API int imgxf_reload_knobs(void) {
    imgxf::knob_table();
    imgxf::load_knobs();
    return IMGXF_OK;
}

// ---- in-kernel shader clock (MI355X_MICROARCH.md: d s_memtime / d s_memrealtime x 100 MHz) -------
// One wave spins until `ticks` of the constant 100 MHz counter have passed and stores both deltas.
__global__ void __launch_bounds__(64) sclk_probe_kernel(unsigned long long* out, unsigned int ticks) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    // bounded: at most ~2^22 polls whatever the counters do, so the wave always finishes
    for (unsigned int i = 0; i < (1u << 22) && r1 - r0 < ticks; ++i) {
        __builtin_amdgcn_s_sleep(8);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

IMGXF_API int imgxf_probe_sclk(void* out2_u64, unsigned int ticks_100mhz, void* stream) {
    if (!out2_u64) return IMGXF_ERR_NULL;
    if (ticks_100mhz == 0 || ticks_100mhz > 100000000u) return IMGXF_ERR_ARG;      // at most 1 s
    hipLaunchKernelGGL(sclk_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out2_u64, ticks_100mhz);
    return imgxf::launch_status();
}
