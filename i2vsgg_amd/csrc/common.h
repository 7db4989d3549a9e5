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

// internal (conv.hip <-> winograd.hip): the ordered 36-plane filter gradient of the Winograd domain
int i2v_internal_wgrad_plane_splits(long long T, int Cout, int Cin);
int32_t i2v_internal_gemm_tn_batched_parts(const float* x, const float* gy, float* parts, int32_t M, int32_t N, int32_t K,
                                           int32_t nbatch, long long stride_x, long long stride_gy, int cap, int* splits, void* stream);
int32_t i2v_internal_reduce_parts(float* parts, int nparts, int planes, long long nk, void* stream);
