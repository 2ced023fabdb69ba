// Library-level entry points: error strings, version, device probe.
#include <string.h>
#include "common.h"

extern "C" {

const char* y4_strerror(int code) {
    switch (code) {
        case Y4_OK: return "ok";
        case Y4_ERR_SHAPE: return "unsupported or inconsistent shape / pitch / alignment";
        case Y4_ERR_NULL: return "required pointer is NULL";
        case Y4_ERR_LAUNCH: return "HIP launch failed";
        case Y4_ERR_WORKSPACE: return "workspace too small";
        case Y4_ERR_NODEVICE: return "no gfx950 device";
        default: return "unknown error";
    }
}

int y4_version(void) { return 100; }

int y4_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

}  // extern "C"
