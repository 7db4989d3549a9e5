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

// Tuning table (i2v_set_tuning): the library reads no environment variable; a host that wants the knobs sets them
// explicitly (i2vsgg_amd/_lib.py forwards the documented I2V_* variables once at import).
int g_i2v_tuning[I2V_TUNE_COUNT] = {
    /* CONV_SPEC */ -1, /* SPLIT_TARGET */ 2, /* SPLIT_TARGET_SKINNY */ -1, /* SPLIT_BELOW */ 256, /* SPLIT_ATOMICS */ 0,
    /* BIG_FC_TILE */ 1, /* WGRAD_V2 */ 1, /* WGRAD_FUSED_TILE */ 128, /* WINO_ROWS */ 0, /* ROIPOOL_C128 */ 1,
    /* CONV_GEMM */ 1, /* STAGGER */ 0, /* ROIALIGN_COLS */ 1, /* WGRAD_PER_CU */ 4, /* WGRAD_XCD */ 1,
    /* FC_FOLD */ 0, /* GEMM_X3 */ 0, /* GEMM_PERSIST */ 0, /* WGRAD_PRIO */ 0, /* STREAM_TILE */ 1,
};

extern "C" int32_t i2v_set_tuning(int32_t key, int32_t value) {
    if (key < 0 || key >= I2V_TUNE_COUNT) {
        i2v_set_error("set_tuning: unknown key %d", key);
        return I2V_ERR_ARG;
    }
    g_i2v_tuning[key] = value;
    return I2V_OK;
}

extern "C" int32_t i2v_get_tuning(int32_t key) {
    return (key < 0 || key >= I2V_TUNE_COUNT) ? I2V_ERR_ARG : g_i2v_tuning[key];
}
