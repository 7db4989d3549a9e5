// Error reporting and version of libi2vsgg_hip.so (host-only translation unit).
#include <hip/hip_runtime.h>
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
    /* CONV_SPEC */ -1, /* SPLIT_TARGET */ 2, /* SPLIT_TARGET_SKINNY */ -1, /* SPLIT_BELOW */ 256, /* SPLIT_ATOMICS */ 2,
    /* BIG_FC_TILE */ 1, /* WGRAD_V2 */ 1, /* WGRAD_FUSED_TILE */ 128, /* WINO_ROWS */ 0, /* ROIPOOL_C128 */ 1,
    /* CONV_GEMM */ 1, /* STAGGER */ 0, /* ROIALIGN_COLS */ 2, /* WGRAD_PER_CU */ 4, /* WGRAD_XCD */ 1,
    /* FC_FOLD */ 0, /* GEMM_X3 */ 0, /* GEMM_PERSIST */ 0, /* WGRAD_PRIO */ 0, /* STREAM_TILE */ 1,
    /* KGROUPS */ 0, /* WGRAD_ORDERED_GFLOP */ 1000000, /* GEMM_DMA */ 1,
};

extern "C" int32_t i2v_build_flags(void) {
#ifdef I2V_EXPERIMENTS
    return I2V_BUILD_EXPERIMENTS;
#else
    return 0;
#endif
}

extern "C" int32_t i2v_set_tuning(int32_t key, int32_t value) {
    if (key < 0 || key >= I2V_TUNE_COUNT) {
        i2v_set_error("set_tuning: unknown key %d", key);
        return I2V_ERR_ARG;
    }
#ifndef I2V_EXPERIMENTS
    // The losing experiments of rounds 1-3 (measurements under profiles/) are compiled only with -DI2V_EXPERIMENTS
    // (I2V_EXPERIMENTS=1 python -m i2vsgg_amd.build): the default library does not carry their kernels, so their knobs
    // accept nothing but "off".
    const bool experiment = key == I2V_TUNE_CONV_SPEC ? value > 0
                          : (key == I2V_TUNE_FC_FOLD || key == I2V_TUNE_GEMM_X3 || key == I2V_TUNE_GEMM_PERSIST ||
                             key == I2V_TUNE_WGRAD_PRIO || key == I2V_TUNE_STAGGER) ? value != 0 : false;
    if (experiment) {
        i2v_set_error("set_tuning: key %d is an experiment; this library was built without -DI2V_EXPERIMENTS", key);
        return I2V_ERR_UNSUPPORTED;
    }
#endif
    g_i2v_tuning[key] = value;
    return I2V_OK;
}

#ifndef I2V_EXPERIMENTS
// csrc/fcfold.hip (a linear layer's forward that applies the previous step's pending SGD update on its pass over the filter:
// built, parity-tested, measured slower than forward + fused update as two kernels -- DESIGN.md 5.6) is not part of the
// default build; its entry points answer "unsupported" and the host keeps the two-kernel path.
extern "C" int32_t i2v_fc_fold_supported(int32_t, int32_t, int32_t, int32_t) { return 0; }
extern "C" int32_t i2v_fc_fold_fwd(const float*, const float*, const float*, const int32_t*, float*, float*, const float*, float*,
                                   int32_t, int32_t, int32_t, int32_t, float, float, float, void*) {
    i2v_set_error("fc_fold_fwd: this library was built without -DI2V_EXPERIMENTS");
    return I2V_ERR_UNSUPPORTED;
}
#endif

extern "C" int32_t i2v_get_tuning(int32_t key) {
    return (key < 0 || key >= I2V_TUNE_COUNT) ? I2V_ERR_ARG : g_i2v_tuning[key];
}

// Streams of the host's own (i2v_stream_create): a step object forks its graph branches onto streams that must not alias
// each other, the capturing stream or a stream some other component drew from a framework's pool (torch deals 32 pooled
// streams per device round robin).  A stream created here belongs to the caller alone.
extern "C" int32_t i2v_stream_create(int32_t device, int32_t priority, void** stream) {
    if (!stream) {
        i2v_set_error("stream_create: null output");
        return I2V_ERR_ARG;
    }
    int prev = -1, n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
        i2v_set_error("stream_create: no device %d (%d visible)", device, n);
        return I2V_ERR_ARG;
    }
    hipStream_t s = nullptr;
    hipError_t e = hipGetDevice(&prev);
    if (e == hipSuccess && prev != device) e = hipSetDevice(device);
    if (e == hipSuccess) {
        int lo = 0, hi = 0;                                   // numerically: hi <= lo, a lower value is a higher priority
        e = hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (e == hipSuccess) {
            int pr = priority < hi ? hi : (priority > lo ? lo : priority);
            e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, pr);
        }
    }
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        i2v_set_error("stream_create: %s", hipGetErrorString(e));
        return I2V_ERR_LAUNCH;
    }
    *stream = (void*)s;
    return I2V_OK;
}

extern "C" int32_t i2v_stream_destroy(void* stream) {
    if (!stream) return I2V_OK;
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) {
        i2v_set_error("stream_destroy: %s", hipGetErrorString(e));
        return I2V_ERR_LAUNCH;
    }
    return I2V_OK;
}
