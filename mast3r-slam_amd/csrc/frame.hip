// Keyframe / frame state on the device: pointmap fusion and the new-keyframe statistics.
//
// Replaces Frame.update_pointmap (frame.py:75-133 of /root/reference/src/mlx_mast3r_slam) with the
// preceding Sim3.act of the tracker (tracker.py:146-147: Xkk = T_CkCf.act(Xkf); keyframe.update_pointmap(Xkk, Ckf))
// fused in, and the unique-match count of the keyframe test (tracker.py:153-155: mx.unique(idx[valid])).
// The reference runs 6-10 MLX elementwise kernels per fusion and a sort for the unique count; here the
// fusion is one streaming kernel (28 B read + 16 B written per point, HBM-bound) and the unique count is a
// bitmap: atomicOr of one bit per valid match, then a popcount reduction - an exact integer.
#include "common.h"
#include "sim3_dev.h"
#include "../../include/m3slam.h"

namespace {

constexpr int kThreads = 256;

// geometry.py:318-351
__device__ __forceinline__ void to_spherical(float x, float y, float z, float &r, float &phi, float &theta) {
    r = sqrtf((x * x + y * y + z * z) + 1e-10f);
    phi = atan2f(y, x);
    theta = acosf(fminf(fmaxf(z / r, -1.0f), 1.0f));
}

__global__ void __launch_bounds__(kThreads)
k_fuse_pointmap(float *__restrict__ Xc, float *__restrict__ Cc, const float *__restrict__ Xn,
                const float *__restrict__ Cn, const float *__restrict__ Tp, int N, int mode) {
    const int n = blockIdx.x * kThreads + threadIdx.x;
    if (n >= N) return;
    V3<float> x{Xn[3 * n], Xn[3 * n + 1], Xn[3 * n + 2]};
    if (Tp) x = act(load_pose<float>(Tp), x);              // Sim3.act, liegroups/sim3.py:222-231
    const float c = Cn[n];
    if (mode == M3_FUSE_REPLACE) {                         // first / recent / best_score winner
        Xc[3 * n] = x.x; Xc[3 * n + 1] = x.y; Xc[3 * n + 2] = x.z; Cc[n] = c;
        return;
    }
    const float c0 = Cc[n];
    const V3<float> x0{Xc[3 * n], Xc[3 * n + 1], Xc[3 * n + 2]};
    if (mode == M3_FUSE_INDEP_CONF) {                      // frame.py:108-114
        if (c > c0) { Xc[3 * n] = x.x; Xc[3 * n + 1] = x.y; Xc[3 * n + 2] = x.z; Cc[n] = c; }
    } else if (mode == M3_FUSE_WEIGHTED_POINTMAP) {        // frame.py:115-120
        const float tot = c0 + c;
        Xc[3 * n] = (c0 * x0.x + c * x.x) / tot;
        Xc[3 * n + 1] = (c0 * x0.y + c * x.y) / tot;
        Xc[3 * n + 2] = (c0 * x0.z + c * x.z) / tot;
        Cc[n] = tot;
    } else {                                               // weighted_spherical, frame.py:121-129
        float r0, p0, t0, r1, p1, t1;
        to_spherical(x0.x, x0.y, x0.z, r0, p0, t0);
        to_spherical(x.x, x.y, x.z, r1, p1, t1);
        const float tot = c0 + c;
        const float r = (c0 * r0 + c * r1) / tot, phi = (c0 * p0 + c * p1) / tot, th = (c0 * t0 + c * t1) / tot;
        const float st = sinf(th);
        Xc[3 * n] = r * st * cosf(phi); Xc[3 * n + 1] = r * st * sinf(phi); Xc[3 * n + 2] = r * cosf(th);
        Cc[n] = tot;
    }
}

__global__ void __launch_bounds__(kThreads)
k_mark_unique(const int64_t *__restrict__ idx, const uint8_t *__restrict__ valid, unsigned *__restrict__ bitmap,
              int N, int range) {
    const int n = blockIdx.x * kThreads + threadIdx.x;
    if (n >= N || !valid[n]) return;
    int64_t i = idx[n];
    if (i < 0) i += range;                                 // numpy / mlx negative indexing
    if (i < 0 || i >= range) return;
    atomicOr(bitmap + (i >> 5), 1u << (i & 31));
}

__global__ void __launch_bounds__(kThreads)
k_popcount(const unsigned *__restrict__ bitmap, int words, int32_t *__restrict__ count) {
    int c = 0;
    for (int w = blockIdx.x * kThreads + threadIdx.x; w < words; w += gridDim.x * kThreads) c += __popc(bitmap[w]);
    c = m3_wave_sum(c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);   // integer: order-independent
}

}  // namespace

extern "C" {

int m3_fuse_pointmap(float *X_canon, float *C, const float *X_new, const float *C_new, const float *T, int N,
                     int mode, void *stream) {
    M3_REQUIRE(X_canon && C && X_new && C_new && N > 0);
    M3_REQUIRE(mode >= M3_FUSE_REPLACE && mode <= M3_FUSE_WEIGHTED_SPHERICAL);
    hipLaunchKernelGGL(k_fuse_pointmap, dim3(m3_cdiv(N, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, X_canon, C,
                       X_new, C_new, T, N, mode);
    M3_CHECK_LAUNCH("m3_fuse_pointmap");
    return M3_OK;
}

int64_t m3_count_unique_ws_words(int range) { return range > 0 ? ((int64_t)range + 31) / 32 : 0; }

int m3_count_unique(const int64_t *idx, const uint8_t *valid, int N, int range, uint32_t *bitmap_ws,
                    int32_t *count_out, void *stream) {
    M3_REQUIRE(idx && valid && bitmap_ws && count_out && N > 0 && range > 0);
    hipStream_t st = (hipStream_t)stream;
    const int words = (int)m3_count_unique_ws_words(range);
    M3_CHECK_HIP(hipMemsetAsync(bitmap_ws, 0, (size_t)words * 4, st), "m3_count_unique/memset");
    M3_CHECK_HIP(hipMemsetAsync(count_out, 0, 4, st), "m3_count_unique/memset");
    hipLaunchKernelGGL(k_mark_unique, dim3(m3_cdiv(N, kThreads)), dim3(kThreads), 0, st, idx, valid, bitmap_ws, N, range);
    const int blocks = words < 256 * kThreads ? m3_cdiv(words, kThreads) : 256;
    hipLaunchKernelGGL(k_popcount, dim3(blocks), dim3(kThreads), 0, st, (const unsigned *)bitmap_ws, words, count_out);
    M3_CHECK_LAUNCH("m3_count_unique");
    return M3_OK;
}

}  // extern "C"
