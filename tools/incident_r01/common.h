// Shared helpers for libm3slam_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "m3slam.h"

#define M3_WAVE 64

// thread-local text of the last HIP failure (m3_last_hip_error)
void m3_set_hip_error(hipError_t e, const char *where);

#define M3_REQUIRE(cond) do { if (!(cond)) return M3_ERR_INVALID_ARG; } while (0)

#define M3_CHECK_LAUNCH(where) do {                                   \
        hipError_t e__ = hipGetLastError();                            \
        if (e__ != hipSuccess) { m3_set_hip_error(e__, where); return M3_ERR_LAUNCH; } \
    } while (0)

#define M3_CHECK_HIP(call, where) do {                                 \
        hipError_t e__ = (call);                                       \
        if (e__ != hipSuccess) { m3_set_hip_error(e__, where); return M3_ERR_LAUNCH; } \
    } while (0)

static inline int m3_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- wave / block reductions (wave = 64 lanes) --------------------------------
__device__ __forceinline__ double m3_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ float m3_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ unsigned m3_wave_max(unsigned v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned o = __shfl_down(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}
