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

// One wave that idles for at most `spin_us` microseconds -- or until *stop becomes non-zero, which the caller sets with a
// stream-ordered write behind the work it wants characterised -- and reports how many shader-clock ticks (s_memtime) and
// 100 MHz reference ticks (s_memrealtime) went by: their ratio x 0.1 is the shader clock in GHz the chip held meanwhile
// (agrees with sysfs freq1_input; under the convolution kernels the package sits at its power cap and the clock
// settles near 1.8 GHz, profiles/NOTES.md).  s_sleep keeps the wave off the issue ports, so the probe does not
// disturb what it measures.
__global__ void clock_probe_kernel(unsigned long long* out, unsigned long long ref_ticks, const unsigned* stop) {
    const unsigned long long r0 = wall_clock64(), c0 = clock64();
    unsigned long long r1 = r0;
    while (r1 - r0 < ref_ticks) {
        __builtin_amdgcn_s_sleep(64);
        r1 = wall_clock64();
        if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
    }
    const unsigned long long c1 = clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

extern "C" int evc_clock_probe(unsigned long long* out2, int spin_us, const unsigned* stop, void* stream) {
    if (!out2 || spin_us <= 0 || spin_us > 10000000) return EVC_EINVAL;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out2, (unsigned long long)spin_us * 100ull, stop);
    return hipGetLastError() == hipSuccess ? EVC_OK : EVC_ELAUNCH;
}
