// api.hip -- library identification entry points of include/evc_hip.h.
#include <hip/hip_runtime.h>
#include <string.h>
#include "../../include/evc_hip.h"

extern "C" const char* evc_version(void) { return "evc-hip 0.1"; }
extern "C" const char* evc_arch(void) { return "gfx950"; }

extern "C" int evc_device_ok(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
