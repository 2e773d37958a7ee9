// Gauss-Newton frame-to-keyframe tracking solve for gfx950.
//
// Replaces FrameTracker._opt_pose_ray_dist_sim3 / _solve (tracker.py:216-324),
// act_Sim3 + point_to_ray_dist (geometry.py:46-137) and the Sim3 algebra
// (liegroups/sim3.py:107-262) of /root/reference/src/mlx_mast3r_slam.
//
// The reference materialises [N,4,7] Jacobians in MLX and pulls a 7x7 system to
// numpy every iteration (one host sync per iteration).  Here one streaming kernel
// per iteration fuses act -> ray/dist -> residual -> Huber -> J^T W J / J^T W r
// (28+7+1 sums per point, float32 per point, float64 accumulation), reduces with
// wave shuffles + LDS to one partial row per workgroup, and a single-workgroup
// kernel finishes the reduction in a FIXED order (bitwise reproducible), solves
// the 7x7 system, retracts the pose and evaluates the stop test - the loop never
// returns to the host.  HBM/L2 bound: 29 B per point per iteration.
#include "common.h"
#include "sim3_dev.h"

namespace {

constexpr int kThreads = 256;
constexpr int kBlocks = 256;          // one partial row per CU
constexpr int kSums = 36;             // 28 (H upper) + 7 (g) + 1 (cost)
// workspace layout (doubles)
constexpr int WS_T = 0;               // T_CkCf [8]
constexpr int WS_OLD = 8;             // old cost
constexpr int WS_DONE = 9;            // != 0 once converged / failed
constexpr int WS_ITERS = 10;
constexpr int WS_TAUN = 11;
constexpr int WS_COST = 12;
constexpr int WS_CONV = 13;
constexpr int WS_PART = 16;           // partials [kBlocks][kSums]

__global__ void k_track_init(const float *__restrict__ T_WCf, const float *__restrict__ T_WCk,
                             const float *__restrict__ T_rel, double *__restrict__ ws) {
    if (threadIdx.x != 0) return;
    Pose<double> T;
    if (T_rel) T = load_pose<double>(T_rel);
    else T = mul(inv_mlx(load_pose<double>(T_WCk)), load_pose<double>(T_WCf));
    store_pose(ws + WS_T, T);
    ws[WS_OLD] = INFINITY;
    ws[WS_DONE] = 0.0; ws[WS_ITERS] = 0.0; ws[WS_TAUN] = 0.0; ws[WS_COST] = 0.0; ws[WS_CONV] = 0.0;
}

__global__ void __launch_bounds__(kThreads)
k_track_accum(const float *__restrict__ Xf, const float *__restrict__ Xk, const float *__restrict__ Qk,
              const uint8_t *__restrict__ valid, double *__restrict__ ws, int N, float huber_k,
              float inv_sigma_ray, float inv_sigma_dist) {
    if (ws[WS_DONE] != 0.0) return;
    const Pose<float> T = load_pose<float>(ws + WS_T);
    double acc[kSums];
#pragma unroll
    for (int i = 0; i < kSums; ++i) acc[i] = 0.0;

    for (int n = blockIdx.x * kThreads + threadIdx.x; n < N; n += kBlocks * kThreads) {
        if (!valid[n]) continue;
        const float sq = sqrtf(Qk[n]);
        const float si_ray = inv_sigma_ray * sq, si_dist = inv_sigma_dist * sq;
        const V3<float> xf{Xf[3 * n], Xf[3 * n + 1], Xf[3 * n + 2]};
        const V3<float> xk{Xk[3 * n], Xk[3 * n + 1], Xk[3 * n + 2]};
        const V3<float> p = act(T, xf);
        const float d = sqrtf(dot(p, p) + 1e-10f), di = 1.0f / d;
        const V3<float> r = di * p;
        const float dk = sqrtf(dot(xk, xk) + 1e-10f), dki = 1.0f / dk;
        const V3<float> rk = dki * xk;
        const float res[4] = {rk.x - r.x, rk.y - r.y, rk.z - r.z, dk - d};
        const float di2 = di * di;
        // a_i = -(row i of d rd / d X); J_row = [a, p x a, a . p]
        const V3<float> a[4] = {
            {-di * (1.0f - di2 * p.x * p.x), di * di2 * p.x * p.y, di * di2 * p.x * p.z},
            {di * di2 * p.y * p.x, -di * (1.0f - di2 * p.y * p.y), di * di2 * p.y * p.z},
            {di * di2 * p.z * p.x, di * di2 * p.z * p.y, -di * (1.0f - di2 * p.z * p.z)},
            {-r.x, -r.y, -r.z}};
        float h[kSums];
#pragma unroll
        for (int i = 0; i < kSums; ++i) h[i] = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float si = (c < 3) ? si_ray : si_dist;
            const float wr = fabsf(si * res[c]);
            const float hub = (wr < huber_k) ? 1.0f : huber_k / wr;
            const float rsi = si * sqrtf(hub);
            const V3<float> pa = cross(p, a[c]);
            const float J[7] = {rsi * a[c].x, rsi * a[c].y, rsi * a[c].z, rsi * pa.x, rsi * pa.y,
                                rsi * pa.z, rsi * dot(a[c], p)};
            const float bb = rsi * res[c];
            int k = 0;
#pragma unroll
            for (int i = 0; i < 7; ++i)
#pragma unroll
                for (int j = i; j < 7; ++j) h[k++] += J[i] * J[j];
#pragma unroll
            for (int i = 0; i < 7; ++i) h[28 + i] -= J[i] * bb;
            h[35] += 0.5f * bb * bb;
        }
#pragma unroll
        for (int i = 0; i < kSums; ++i) acc[i] += (double)h[i];
    }

    __shared__ double red[kThreads / 64][kSums];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < kSums; ++i) {
        const double s = m3_wave_sum(acc[i]);
        if (lane == 0) red[wv][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < kSums) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) s += red[w][threadIdx.x];
        ws[WS_PART + blockIdx.x * kSums + threadIdx.x] = s;
    }
}

// Fixed-order final reduction of the kBlocks partial rows: wave w owns sums 9w..9w+8,
// every lane adds rows lane, lane+64, lane+128, lane+192, then a shuffle tree.
__device__ __forceinline__ void reduce_partials(const double *__restrict__ part, double *sums /*LDS[36]*/) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c = 0; c < 9; ++c) {
        const int col = wv * 9 + c;
        double s = 0.0;
        for (int r = lane; r < kBlocks; r += 64) s += part[r * kSums + col];
        s = m3_wave_sum(s);
        if (lane == 0) sums[col] = s;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(kThreads)
k_track_solve(double *__restrict__ ws, float rel_error, float delta_norm, int fixed_iters) {
    if (ws[WS_DONE] != 0.0) return;
    __shared__ double sums[kSums];
    reduce_partials(ws + WS_PART, sums);
    if (threadIdx.x != 0) return;
    double H[7][7], g[7];
    int k = 0;
    for (int i = 0; i < 7; ++i)
        for (int j = i; j < 7; ++j) { H[i][j] = sums[k]; H[j][i] = sums[k]; ++k; }
    for (int i = 0; i < 7; ++i) { g[i] = sums[28 + i]; H[i][i] += 1e-6; }
    const double cost = sums[35];
    if (!solve_small<7>(H, g, 7)) {            // singular system: stop, keep the pose (reference would raise)
        ws[WS_DONE] = 2.0;
        return;
    }
    double tn = 0.0;
    for (int i = 0; i < 7; ++i) tn += g[i] * g[i];
    tn = sqrt(tn);
    Pose<double> T = mul(load_pose<double>(ws + WS_T), exp_mlx(g));
    store_pose(ws + WS_T, T);
    const double old = ws[WS_OLD];
    const double rel_dec = fabs((old - cost) / (old + 1e-10));       // NaN on the first step, as in the reference
    const bool conv = (rel_dec < (double)rel_error) || (tn < (double)delta_norm);
    ws[WS_ITERS] += 1.0;
    ws[WS_TAUN] = tn;
    ws[WS_COST] = cost;
    ws[WS_OLD] = cost;
    if (conv && !fixed_iters) { ws[WS_DONE] = 1.0; ws[WS_CONV] = 1.0; }
}

__global__ void k_track_final(const double *__restrict__ ws, const float *__restrict__ T_WCk,
                              float *__restrict__ T_WCf_out, float *__restrict__ T_rel_out,
                              double *__restrict__ info) {
    if (threadIdx.x != 0) return;
    Pose<double> T = load_pose<double>(ws + WS_T);
    store_pose(T_rel_out, T);
    store_pose(T_WCf_out, mul(load_pose<double>(T_WCk), T));
    info[0] = ws[WS_ITERS]; info[1] = ws[WS_COST]; info[2] = ws[WS_TAUN]; info[3] = ws[WS_CONV];
}

__global__ void __launch_bounds__(kThreads)
k_track_export(const double *__restrict__ ws, double *__restrict__ out) {
    __shared__ double sums[kSums];
    reduce_partials(ws + WS_PART, sums);
    if (threadIdx.x < kSums) out[threadIdx.x] = sums[threadIdx.x];
}

__global__ void __launch_bounds__(kThreads)
k_track_gather(const float *__restrict__ Xf_canon, const float *__restrict__ Cf_avg,
               const float *__restrict__ Ck_avg, const float *__restrict__ Qff, const float *__restrict__ Qkf,
               const int64_t *__restrict__ idx, const uint8_t *__restrict__ valid_match,
               float *__restrict__ Xf_g, float *__restrict__ Qk, uint8_t *__restrict__ valid_opt,
               uint8_t *__restrict__ valid_kf, int32_t *__restrict__ counts, int N, float C_conf, float Q_conf) {
    const int n = blockIdx.x * kThreads + threadIdx.x;
    int vo = 0, vk = 0;
    if (n < N) {
        int64_t id = idx[n];
        if (id < 0) id += N;
        id = id < 0 ? 0 : (id >= N ? N - 1 : id);
        Xf_g[3 * n + 0] = Xf_canon[3 * id + 0];
        Xf_g[3 * n + 1] = Xf_canon[3 * id + 1];
        Xf_g[3 * n + 2] = Xf_canon[3 * id + 2];
        const float q = sqrtf(Qff[id] * Qkf[n]);
        Qk[n] = q;
        const bool vm = valid_match[n] != 0, vq = q > Q_conf;
        vk = vm && vq;
        vo = vk && (Cf_avg[id] > C_conf) && (Ck_avg[n] > C_conf);
        valid_opt[n] = (uint8_t)vo;
        valid_kf[n] = (uint8_t)vk;
    }
    const unsigned long long bo = __ballot(vo), bk = __ballot(vk);
    if ((threadIdx.x & 63) == 0) {
        if (bo) atomicAdd(&counts[0], __popcll(bo));
        if (bk) atomicAdd(&counts[1], __popcll(bk));
    }
}

__global__ void __launch_bounds__(kThreads)
k_sim3_act(const float *__restrict__ Tp, const float *__restrict__ X, float *__restrict__ out, int N) {
    const int n = blockIdx.x * kThreads + threadIdx.x;
    if (n >= N) return;
    const Pose<float> T = load_pose<float>(Tp);
    const V3<float> p = act(T, V3<float>{X[3 * n], X[3 * n + 1], X[3 * n + 2]});
    out[3 * n] = p.x; out[3 * n + 1] = p.y; out[3 * n + 2] = p.z;
}

}  // namespace

extern "C" {

int64_t m3_track_ws_doubles(void) { return WS_PART + (int64_t)kBlocks * kSums; }

int m3_track_gather(const float *Xf_canon, const float *Cf_avg, const float *Ck_avg, const float *Qff,
                    const float *Qkf, const int64_t *idx, const uint8_t *valid_match, float *Xf_g,
                    float *Qk, uint8_t *valid_opt, uint8_t *valid_kf, int32_t *counts, int N,
                    float C_conf, float Q_conf, void *stream) {
    M3_REQUIRE(Xf_canon && Cf_avg && Ck_avg && Qff && Qkf && idx && valid_match);
    M3_REQUIRE(Xf_g && Qk && valid_opt && valid_kf && counts && N > 0);
    hipStream_t st = (hipStream_t)stream;
    M3_CHECK_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), st), "m3_track_gather/memset");
    hipLaunchKernelGGL(k_track_gather, dim3(m3_cdiv(N, kThreads)), dim3(kThreads), 0, st, Xf_canon, Cf_avg,
                       Ck_avg, Qff, Qkf, idx, valid_match, Xf_g, Qk, valid_opt, valid_kf, counts, N, C_conf, Q_conf);
    M3_CHECK_LAUNCH("m3_track_gather");
    return M3_OK;
}

int m3_track_gn_ray_dist(const float *Xf, const float *Xk, const float *Qk, const uint8_t *valid,
                         const float *T_WCf, const float *T_WCk, float *T_WCf_out, float *T_CkCf_out,
                         double *info, double *ws, int N, int max_iters, float huber_k, float sigma_ray,
                         float sigma_dist, float rel_error, float delta_norm, int fixed_iters, void *stream) {
    M3_REQUIRE(Xf && Xk && Qk && valid && T_WCf && T_WCk && T_WCf_out && T_CkCf_out && info && ws);
    M3_REQUIRE(N > 0 && max_iters >= 0 && sigma_ray > 0.f && sigma_dist > 0.f && huber_k > 0.f);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_track_init, dim3(1), dim3(64), 0, st, T_WCf, T_WCk, (const float *)nullptr, ws);
    M3_CHECK_LAUNCH("m3_track_gn/init");
    const float isr = (float)(1.0 / (double)sigma_ray), isd = (float)(1.0 / (double)sigma_dist);
    for (int it = 0; it < max_iters; ++it) {
        hipLaunchKernelGGL(k_track_accum, dim3(kBlocks), dim3(kThreads), 0, st, Xf, Xk, Qk, valid, ws, N,
                           huber_k, isr, isd);
        hipLaunchKernelGGL(k_track_solve, dim3(1), dim3(kThreads), 0, st, ws, rel_error, delta_norm, fixed_iters);
    }
    M3_CHECK_LAUNCH("m3_track_gn/loop");
    hipLaunchKernelGGL(k_track_final, dim3(1), dim3(64), 0, st, (const double *)ws, T_WCk, T_WCf_out,
                       T_CkCf_out, info);
    M3_CHECK_LAUNCH("m3_track_gn/final");
    return M3_OK;
}

int m3_track_normal_eq(const float *Xf, const float *Xk, const float *Qk, const uint8_t *valid,
                       const float *T_CkCf, double *out, double *ws, int N, float huber_k,
                       float sigma_ray, float sigma_dist, void *stream) {
    M3_REQUIRE(Xf && Xk && Qk && valid && T_CkCf && out && ws && N > 0);
    M3_REQUIRE(sigma_ray > 0.f && sigma_dist > 0.f && huber_k > 0.f);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_track_init, dim3(1), dim3(64), 0, st, (const float *)nullptr, (const float *)nullptr,
                       T_CkCf, ws);
    const float isr = (float)(1.0 / (double)sigma_ray), isd = (float)(1.0 / (double)sigma_dist);
    hipLaunchKernelGGL(k_track_accum, dim3(kBlocks), dim3(kThreads), 0, st, Xf, Xk, Qk, valid, ws, N, huber_k,
                       isr, isd);
    hipLaunchKernelGGL(k_track_export, dim3(1), dim3(kThreads), 0, st, (const double *)ws, out);
    M3_CHECK_LAUNCH("m3_track_normal_eq");
    return M3_OK;
}

int m3_sim3_act(const float *T, const float *X, float *out, int N, void *stream) {
    M3_REQUIRE(T && X && out && N > 0);
    hipLaunchKernelGGL(k_sim3_act, dim3(m3_cdiv(N, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, T, X, out, N);
    M3_CHECK_LAUNCH("m3_sim3_act");
    return M3_OK;
}

}  // extern "C"
