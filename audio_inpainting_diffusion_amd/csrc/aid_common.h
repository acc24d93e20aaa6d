// Shared helpers for the libaid_hip.so translation units (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "aid_kernels.h"

#define AID_WAVE 64

void aid_set_error(const char* msg);
void aid_note_kernel(const char* name);   // aid_capi.hip: records which device kernel an entry point dispatched to

// Checks the launch that just happened; returns AID_E_LAUNCH from the enclosing function on failure.
#define AID_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) {                             \
            aid_set_error(hipGetErrorString(e__));           \
            return AID_E_LAUNCH;                             \
        }                                                    \
    } while (0)

#define AID_REQUIRE(cond, msg)                               \
    do {                                                     \
        if (!(cond)) {                                       \
            aid_set_error(msg);                              \
            return AID_E_BADARG;                             \
        }                                                    \
    } while (0)

static inline int aid_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline int aid_pow2ceil(int v) { return 1 << aid_ilog2(v); }
static inline int aid_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float aid_gelu(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
// d/dx gelu(x) = Phi(x) + x*phi(x)
__device__ __forceinline__ float aid_dgelu(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
