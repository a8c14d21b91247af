// Error reporting and version of libsan_hip.so (host only).
#include <stdarg.h>
#include <stdio.h>

#include <hip/hip_runtime.h>

#include "../../include/san_hip.h"

static thread_local char g_err[512] = "";

void san_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* san_last_error_string(void) { return g_err; }
extern "C" int san_version(void) { return 1; }

// Events of a recorded step's stream hand-offs, made by the runtime this library links (not a second copy found by name) and on
// the device that will record them (ctypes.CDLL("libamdhip64.so") + the current device could be wrong on both counts).
extern "C" int san_event_create(int device, int light, void** event) {
    if (!event) {
        san_set_error("san_event_create: null pointer");
        return SAN_E_ARG;
    }
    *event = nullptr;
    int prev = -1;
    hipError_t e = hipGetDevice(&prev);
    if (e == hipSuccess && prev != device) e = hipSetDevice(device);
    if (e != hipSuccess) {
        san_set_error("san_event_create: cannot select device %d: %s", device, hipGetErrorString(e));
        return (int)e;
    }
    hipEvent_t ev = nullptr;
    unsigned flags = hipEventDisableTiming;
#ifdef hipEventDisableSystemFence
    if (light) flags |= hipEventDisableSystemFence;
#else
    if (light) flags |= 0x20000000u;
#endif
    e = hipEventCreateWithFlags(&ev, flags);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess || !ev) {
        san_set_error("san_event_create: hipEventCreateWithFlags(0x%x): %s", flags, hipGetErrorString(e));
        return e != hipSuccess ? (int)e : SAN_E_UNSUPPORTED;
    }
    *event = (void*)ev;
    return SAN_OK;
}

extern "C" int san_event_destroy(void* event) {
    if (!event) return SAN_OK;
    const hipError_t e = hipEventDestroy((hipEvent_t)event);
    if (e != hipSuccess) {
        san_set_error("san_event_destroy: %s", hipGetErrorString(e));
        return (int)e;
    }
    return SAN_OK;
}
