// Error reporting and version of libsan_hip.so (host only).
#include <stdarg.h>
#include <stdio.h>

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
