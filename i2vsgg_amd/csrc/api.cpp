// Error reporting and version of libi2vsgg_hip.so (host-only translation unit).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/i2vsgg_hip.h"

static thread_local char g_err[512] = "";

void i2v_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int32_t i2v_version(void) { return 100; }   // 0.1.0
extern "C" const char* i2v_last_error(void) { return g_err; }
