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
