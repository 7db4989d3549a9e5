// Shared helpers for the gfx950 kernels of libi2vsgg_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/i2vsgg_hip.h"

#define I2V_WAVE 64

void i2v_set_error(const char* fmt, ...);
extern int g_i2v_tuning[];       // api.cpp; indexed by I2V_TUNE_*

#define I2V_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            i2v_set_error(__VA_ARGS__);          \
            return I2V_ERR_ARG;                  \
        }                                        \
    } while (0)

#define I2V_CHECK_LAUNCH(name)                                              \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            i2v_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return I2V_ERR_LAUNCH;                                          \
        }                                                                   \
    } while (0)

static inline int i2v_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t i2v_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
