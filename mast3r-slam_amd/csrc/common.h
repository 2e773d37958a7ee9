// Shared helpers for libm3slam_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/m3slam.h"

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

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE opt-in: a process that drives several GPUs (one host
// thread per device) has to make it on each of them, and two threads may arrive together.  One bit per device in an
// atomic mask per kernel instantiation; setting the attribute twice is harmless, so the only requirement is that a
// launch on device d never precedes the opt-in on device d.
//   static M3AttrOnce once; int dev;
//   if (m3_attr_need(once, &dev)) { hipFuncSetAttribute(...); m3_attr_done(once, dev); }
#include <atomic>
struct M3AttrOnce { std::atomic<unsigned long long> mask{0}; };
static inline bool m3_attr_need(M3AttrOnce &o, int *dev) {
    int d = 0;
    *dev = -1;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) return true;         // unknown device: opt in every time
    *dev = d;
    return !((o.mask.load(std::memory_order_acquire) >> d) & 1ull);
}
static inline void m3_attr_done(M3AttrOnce &o, int dev) {
    if (dev >= 0) o.mask.fetch_or(1ull << dev, std::memory_order_release);
}

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
__device__ __forceinline__ int m3_wave_sum(int v) {
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

// ---- 36 float64 sums per thread -> one row per 256-thread workgroup, fixed order, through LDS --------------
// A float64 shuffle tree costs 36 x 6 x 2 ds_bpermute + 216 adds per wave (a third of the Gauss-Newton
// accumulation kernels at 8 points per thread).  Here every lane stores its 36 values (row = sum, column = lane;
// rows padded to 65 doubles so the column reads of lanes 0..35 fall on distinct banks), lane i < 36 adds row i
// with four independent chains, threads 0..35 add the four wave totals: 36 stores + 64 loads + 64 adds per wave.
__device__ __forceinline__ void m3_block_reduce36(const double *acc, double *__restrict__ out) {
    constexpr int kS = 36, kW = 4, kStride = 65;
    __shared__ double red[kW][kS][kStride];
    __shared__ double tot[kW][kS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < kS; ++i) red[wv][i][lane] = acc[i];
    __syncthreads();
    if (lane < kS) {
        const double *row = red[wv][lane];
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int k = 0; k < 64; k += 4) { s0 += row[k]; s1 += row[k + 1]; s2 += row[k + 2]; s3 += row[k + 3]; }
        tot[wv][lane] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (threadIdx.x < kS) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kW; ++w) s += tot[w][threadIdx.x];
        out[threadIdx.x] = s;
    }
}
