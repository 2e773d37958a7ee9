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

// ---- "best_score" filtering (frame.py:59-73, :103-107): the score of a pointmap is the MEDIAN (or mean) of its
// confidences and the frame keeps the pointmap with the best score.  The reference sorts on the host (np.median of the
// pulled array).  Here the median is an exact radix select on the order-preserving integer image of the float bits - four
// histogram passes over one byte each, most significant first, for BOTH middle ranks at once - and the decision stays on
// the device: no sort, no host synchronisation.  As in the tracking solve, the serial bit between two passes (scan 256
// bins, pick the byte, narrow the rank) runs in the PROLOGUE of the next pass's launch, redundantly per workgroup.
// ws (uint32): hist [4 passes][2 ranks][256] | state [4 passes][2 ranks][2] = (prefix, remaining rank)
constexpr int kSelHist = 4 * 2 * 256, kSelWords = kSelHist + 4 * 2 * 2;

__device__ __forceinline__ unsigned sort_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_value(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// Byte of pass `pass` (0 = most significant) for the two ranks from the histogram of that pass, given the state before it.
// One wave-sized scan per rank by thread 0 of each rank's half would do; 256 bins are scanned serially by two threads
// (512 additions) - negligible beside the pass over N elements.
__device__ __forceinline__ void select_byte(const unsigned *__restrict__ ws, int pass, unsigned rank_lo, unsigned rank_hi,
                                            unsigned (&prefix)[2], unsigned (&rem)[2], unsigned *lds /*[4]*/) {
    const int t = threadIdx.x;
    if (t < 2) {
        unsigned pre = 0, r = t == 0 ? rank_lo : rank_hi;
        if (pass > 0) { pre = ws[kSelHist + ((pass - 1) * 2 + t) * 2]; r = ws[kSelHist + ((pass - 1) * 2 + t) * 2 + 1]; }
        const unsigned *h = ws + (pass * 2 + t) * 256;
        unsigned cum = 0;
        int b = 0;
        for (; b < 255; ++b) {
            const unsigned c = h[b];
            if (r < cum + c) break;
            cum += c;
        }
        lds[2 * t] = pre | ((unsigned)b << (24 - 8 * pass));
        lds[2 * t + 1] = r - cum;
    }
    __syncthreads();
    prefix[0] = lds[0]; rem[0] = lds[1]; prefix[1] = lds[2]; rem[1] = lds[3];
}

// pass = 0..3: histogram of byte `pass` over the elements whose more significant bytes equal the selected prefix
__global__ void __launch_bounds__(kThreads)
k_select_pass(const float *__restrict__ v, int N, unsigned *__restrict__ ws, int pass, unsigned rank_lo, unsigned rank_hi) {
    __shared__ unsigned hist[2][256];
    __shared__ unsigned sel[4];
    const int t = threadIdx.x;
    hist[0][t] = 0; hist[1][t] = 0;
    unsigned prefix[2] = {0, 0}, rem[2] = {rank_lo, rank_hi};
    if (pass > 0) {
        select_byte(ws, pass - 1, rank_lo, rank_hi, prefix, rem, sel);
        if (blockIdx.x == 0 && t < 4) ws[kSelHist + (pass - 1) * 4 + t] = sel[t];      // (prefix, rem) x 2 ranks
    } else {
        __syncthreads();
    }
    const unsigned mask = pass == 0 ? 0u : 0xffffffffu << (32 - 8 * pass);
    const int shift = 24 - 8 * pass;
    for (int n = blockIdx.x * kThreads + t; n < N; n += gridDim.x * kThreads) {
        const unsigned k = sort_key(v[n]);
        if ((k & mask) == prefix[0]) atomicAdd(&hist[0][(k >> shift) & 255], 1u);
        if ((k & mask) == prefix[1]) atomicAdd(&hist[1][(k >> shift) & 255], 1u);
    }
    __syncthreads();
    if (hist[0][t]) atomicAdd(ws + (pass * 2 + 0) * 256 + t, hist[0][t]);
    if (hist[1][t]) atomicAdd(ws + (pass * 2 + 1) * 256 + t, hist[1][t]);
}

// median = mean of the two middle order statistics (np.median / mx.median), in float32
__global__ void __launch_bounds__(kThreads)
k_select_final(const unsigned *__restrict__ ws, unsigned rank_lo, unsigned rank_hi, float *__restrict__ out) {
    __shared__ unsigned sel[4];
    unsigned prefix[2], rem[2];
    select_byte(ws, 3, rank_lo, rank_hi, prefix, rem, sel);
    if (threadIdx.x == 0) out[0] = 0.5f * (key_value(prefix[0]) + key_value(prefix[1]));
}

// state [2] = (best score so far, 1.0 if the last call replaced the pointmap): winner takes all
__global__ void k_best_score_gate(const float *__restrict__ score_new, float *__restrict__ state) {
    if (threadIdx.x != 0) return;
    const bool better = score_new[0] > state[0];
    state[1] = better ? 1.0f : 0.0f;
    if (better) state[0] = score_new[0];
}

__global__ void __launch_bounds__(kThreads)
k_fuse_replace_if(float *__restrict__ Xc, float *__restrict__ Cc, const float *__restrict__ Xn, const float *__restrict__ Cn,
                  const float *__restrict__ Tp, int N, const float *__restrict__ state) {
    if (state[1] == 0.0f) return;
    const int n = blockIdx.x * kThreads + threadIdx.x;
    if (n >= N) return;
    V3<float> x{Xn[3 * n], Xn[3 * n + 1], Xn[3 * n + 2]};
    if (Tp) x = act(load_pose<float>(Tp), x);
    Xc[3 * n] = x.x; Xc[3 * n + 1] = x.y; Xc[3 * n + 2] = x.z; Cc[n] = Cn[n];
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

int64_t m3_median_ws_words(void) { return kSelWords; }

int m3_median_f32(const float *v, int N, uint32_t *ws, float *out, void *stream) {
    M3_REQUIRE(v && ws && out && N > 0);
    hipStream_t st = (hipStream_t)stream;
    M3_CHECK_HIP(hipMemsetAsync(ws, 0, (size_t)kSelWords * 4, st), "m3_median_f32/memset");
    const unsigned lo = (unsigned)((N - 1) / 2), hi = (unsigned)(N / 2);
    const int blocks = m3_cdiv(N, kThreads) < 1024 ? m3_cdiv(N, kThreads) : 1024;
    for (int pass = 0; pass < 4; ++pass)
        hipLaunchKernelGGL(k_select_pass, dim3(blocks), dim3(kThreads), 0, st, v, N, ws, pass, lo, hi);
    hipLaunchKernelGGL(k_select_final, dim3(1), dim3(kThreads), 0, st, (const unsigned *)ws, lo, hi, out);
    M3_CHECK_LAUNCH("m3_median_f32");
    return M3_OK;
}

int m3_fuse_pointmap_if_better(float *X_canon, float *C, const float *X_new, const float *C_new, const float *T, int N,
                               const float *score_new, float *best_state, void *stream) {
    M3_REQUIRE(X_canon && C && X_new && C_new && score_new && best_state && N > 0);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_best_score_gate, dim3(1), dim3(64), 0, st, score_new, best_state);
    hipLaunchKernelGGL(k_fuse_replace_if, dim3(m3_cdiv(N, kThreads)), dim3(kThreads), 0, st, X_canon, C, X_new, C_new, T, N,
                       (const float *)best_state);
    M3_CHECK_LAUNCH("m3_fuse_pointmap_if_better");
    return M3_OK;
}

}  // extern "C"
