// Gauss-Newton frame-to-keyframe tracking solve for gfx950.
//
// Replaces FrameTracker._opt_pose_ray_dist_sim3 / _solve (tracker.py:216-324),
// act_Sim3 + point_to_ray_dist (geometry.py:46-137) and the Sim3 algebra
// (liegroups/sim3.py:107-262) of /root/reference/src/mlx_mast3r_slam.
//
// The reference materialises [N,4,7] Jacobians in MLX and pulls a 7x7 system to
// numpy every iteration (one host sync per iteration).  Here one streaming kernel
// per iteration fuses act -> ray/dist -> residual -> Huber -> J^T W J / J^T W r
// (28+7+1 sums per point, float32 per point, float64 accumulation) and reduces with
// wave shuffles + LDS to one partial row per workgroup.  The step that follows - finish
// the reduction in a FIXED order (bitwise reproducible), solve the 7x7 system, retract
// the pose, evaluate the stop test - runs in the PROLOGUE of the next iteration's kernel,
// redundantly in every workgroup (64 x 36 partials out of L2, ~3 us) instead of as a
// launch of its own (6 us): ONE launch per iteration; the loop never returns to the host.
// State and partials are double-buffered by iteration parity so that no workgroup reads
// what another one of the same launch writes.  HBM/L2 bound: 29 B per point per iteration.
#include "common.h"
#include <cstdlib>
#include "sim3_dev.h"

namespace {

constexpr int kThreads = 256;
constexpr int kBlocks = 256;          // most partial rows a problem can have (workspace size)
constexpr int kSums = 36;             // 28 (H upper) + 7 (g) + 1 (cost)
// workspace layout (doubles): two state slots and two partial buffers, indexed by the parity of the number of solved steps
constexpr int WS_T = 0;               // T_CkCf [8]
constexpr int WS_OLD = 8;             // old cost
constexpr int WS_DONE = 9;            // != 0 once converged (1) / failed (2)
constexpr int WS_ITERS = 10;
constexpr int WS_TAUN = 11;
constexpr int WS_COST = 12;
constexpr int WS_CONV = 13;
constexpr int WS_STATE = 16;          // doubles per state slot
constexpr int WS_PART = 2 * WS_STATE; // partials [2][kBlocks][kSums]
constexpr int WS_STRIDE = WS_PART + 2 * kBlocks * kSums;   // doubles per problem

__device__ __forceinline__ double *ws_state(double *ws, int steps) { return ws + (steps & 1) * WS_STATE; }
__device__ __forceinline__ double *ws_part(double *ws, int steps) { return ws + WS_PART + (size_t)(steps & 1) * kBlocks * kSums; }

// Partial rows (= workgroups) per problem: ~2 workgroups per CU over the whole batch, so a thread sees
// enough points to amortise the 36-value block reduction (at 256 rows x 8 problems it saw 4 points
// and the reduction dominated: 62 us per iteration for 8 x 262144 points, HBM time 10 us).
static inline int track_blocks(int P) {
    static const int total = [] { const char *e = getenv("M3_TRACK_BLOCKS"); return e ? atoi(e) : 512; }();
    int b = total / (P > 0 ? P : 1);
    return b < 16 ? 16 : (b > kBlocks ? kBlocks : b);
}

__global__ void k_track_init(const float *__restrict__ T_WCf, const float *__restrict__ T_WCk,
                             const float *__restrict__ T_rel, double *__restrict__ ws) {
    if (threadIdx.x != 0) return;
    const int pb = blockIdx.x;
    ws += (size_t)pb * WS_STRIDE;
    if (T_WCf) T_WCf += 8 * pb;
    if (T_WCk) T_WCk += 8 * pb;
    if (T_rel) T_rel += 8 * pb;
    Pose<double> T;
    if (T_rel) T = load_pose<double>(T_rel);
    else T = mul(inv_mlx(load_pose<double>(T_WCk)), load_pose<double>(T_WCf));
    store_pose(ws + WS_T, T);
    ws[WS_OLD] = INFINITY;
    ws[WS_DONE] = 0.0; ws[WS_ITERS] = 0.0; ws[WS_TAUN] = 0.0; ws[WS_COST] = 0.0; ws[WS_CONV] = 0.0;
}

// Fixed-order final reduction of the kBlocks partial rows: wave w owns sums 9w..9w+8,
// every lane adds rows lane, lane+64, lane+128, lane+192, then a shuffle tree.
__device__ __forceinline__ void reduce_partials(const double *__restrict__ part, double *sums /*LDS[36]*/, int nblk) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c = 0; c < 9; ++c) {
        const int col = wv * 9 + c;
        double s = 0.0;
        for (int r = lane; r < nblk; r += 64) s += part[r * kSums + col];
        s = m3_wave_sum(s);
        if (lane == 0) sums[col] = s;
    }
    __syncthreads();
}

// One solved Gauss-Newton step = the per-iteration tail of the tracking solve, executed by a whole workgroup of 256:
//   1. the nblk x 36 partial rows are reduced by 252 threads (thread = (row group of 7, column); rows in a fixed order),
//      then 36 threads add the 7 group sums - one wave of independent, coalesced loads instead of nine dependent
//      load -> shuffle-tree rounds;
//   2. the 7 x 7 system [H + 1e-6 I | g] is solved by Gauss-Jordan elimination IN ONE WAVE, lane 8 i + j holding element
//      (i, j) (column 7 = right-hand side): 7 steps of three broadcasts and one FMA instead of ~150 dependent float64
//      operations on one lane.  No pivoting: the matrix is a Gram matrix plus 1e-6 I (symmetric positive definite); a
//      pivot that is not positive and finite reports the failure the old pivoted elimination reported for a zero column;
//   3. the retraction T <- T exp(tau) needs sin(th), cos(th), sin(th/2), cos(th/2) and exp(sigma) in float64: lanes 0-3
//      evaluate sin at (th, th + pi/2, th/2, th/2 + pi/2) in ONE library call, lane 4 the exponential - five serial
//      library calls before.
// Step number `steps` (1, 2, ...) reads state and partials of parity steps - 1 and produces the state of parity
// steps: every workgroup of the next accumulation launch runs it redundantly (identical inputs, identical operation
// order -> identical bits) and ONE of them (`writer`) stores the new state; nobody reads what this launch writes.
// Returns true when the problem is finished (converged, failed, or was already); T_out = the pose to linearise at.
__device__ __forceinline__ double bcast(double v, int src) { return __shfl(v, src, 64); }

struct StepLds { double red[7][kSums]; double sums[kSums]; double st[WS_STATE]; };

__device__ __forceinline__ bool track_step(double *__restrict__ ws, int steps, bool writer, float rel_error, float delta_norm,
                                           int fixed_iters, int nblk, StepLds &L, Pose<float> &T_out) {
    const double *sin_ = ws_state(ws, steps - 1);
    const double *part_in = ws_part(ws, steps - 1);
    double *sout = ws_state(ws, steps);
    const int t = threadIdx.x;
    if (sin_[WS_DONE] != 0.0) {                            // workgroup-uniform: carry the final state forward
        if (t < WS_STATE) {
            L.st[t] = sin_[t];
            if (writer) sout[t] = sin_[t];
        }
        __syncthreads();
        T_out = load_pose<float>(L.st + WS_T);
        return true;
    }
    if (t < 7 * kSums) {
        const int rg = t / kSums, col = t - rg * kSums;
        const double *part = part_in + col;
        double s0 = 0.0, s1 = 0.0;
        int r = rg;
        for (; r + 7 < nblk; r += 14) { s0 += part[(size_t)r * kSums]; s1 += part[(size_t)(r + 7) * kSums]; }
        if (r < nblk) s0 += part[(size_t)r * kSums];
        L.red[rg][col] = s0 + s1;
    }
    __syncthreads();
    if (t < kSums) L.sums[t] = ((L.red[0][t] + L.red[1][t]) + (L.red[2][t] + L.red[3][t])) + ((L.red[4][t] + L.red[5][t]) + L.red[6][t]);
    __syncthreads();
    if (t < 64) {                                          // one wave; every lane of it stays active to the end
        const int i = (t >> 3) < 7 ? (t >> 3) : 6, j = t & 7;
        // upper-triangular packing of the 28 sums: (a, b), a <= b -> a * 7 - a (a - 1) / 2 + (b - a)
        const int a_ = i < j ? i : j, b_ = i < j ? j : i;
        double e = (j < 7) ? L.sums[a_ * 7 - (a_ * (a_ - 1)) / 2 + (b_ - a_)] + (i == j ? 1e-6 : 0.0) : L.sums[28 + i];
        const double cost = L.sums[35];
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const double piv = bcast(e, 8 * k + k), rk = bcast(e, 8 * k + j), ik = bcast(e, 8 * i + k);
            ok = ok && (piv > 0.0) && isfinite(piv);
            // 1 / pivot from v_rcp_f64 + two Newton steps (within 1 ulp) - a correctly rounded division is ~35 dependent
            // instructions, seven of them in a row on this chain
            double ip = __builtin_amdgcn_rcp(piv);
            ip = fma(fma(-piv, ip, 1.0), ip, ip);
            ip = fma(fma(-piv, ip, 1.0), ip, ip);
            const double nrk = rk * ip;
            e = (i == k) ? nrk : e - ik * nrk;
        }
        double tau[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) tau[k] = bcast(e, 8 * k + 7);
        double tn = 0.0;
#pragma unroll
        for (int k = 0; k < 7; ++k) tn += tau[k] * tau[k];
        tn = sqrt(tn);
        // a step whose scale factor e^sigma leaves the float range (degenerate geometry) is refused like a singular
        // system: stop and keep the pose (the reference raises inside _opt_pose_*, tracker.py:121-141)
        const bool failed = !ok || !isfinite(tn) || fabs(tau[6]) > 30.0;
        // exp(tau), liegroups/sim3.py:107-154 (exp_mlx of sim3_dev.h with the transcendentals shared across lanes)
        const V3<double> v{tau[0], tau[1], tau[2]}, w{tau[3], tau[4], tau[5]};
        const double th2 = dot(w, w), th = sqrt(th2 + 1e-10);
        const bool small = th2 < 1e-8;
        const double arg = (t & 2 ? 0.5 * th : th) + (t & 1 ? 1.5707963267948966 : 0.0);
        const double tr = (t == 4) ? exp(failed ? 0.0 : tau[6]) : sin(failed ? 0.0 : arg);
        const double sin_th = bcast(tr, 0), cos_th = bcast(tr, 1), sin_h = bcast(tr, 2), cos_h = bcast(tr, 3), es = bcast(tr, 4);
        if (t == 0) {
#pragma unroll
            for (int k = 0; k < WS_STATE; ++k) L.st[k] = sin_[k];
            if (failed) {
                L.st[WS_DONE] = 2.0;
            } else {
                const double A = small ? 1.0 - th2 / 6.0 : sin_th / th;
                const double B = small ? 0.5 - th2 / 24.0 : (1.0 - cos_th) / th2;
                const double C = small ? 1.0 / 6.0 - th2 / 120.0 : (1.0 - A) / th2;
                const V3<double> wv = cross(w, v);
                Pose<double> E;
                E.t = v + B * wv + C * cross(w, wv);
                const double sinc_half = small ? 0.5 - th2 / 48.0 : sin_h / th;
                const double cos_half = small ? 1.0 - th2 / 8.0 : cos_h;
                E.q = {sinc_half * w.x, sinc_half * w.y, sinc_half * w.z, cos_half};
                E.s = es;
                store_pose(L.st + WS_T, mul(load_pose<double>(sin_ + WS_T), E));
                const double old = sin_[WS_OLD];
                const double rel_dec = fabs((old - cost) / (old + 1e-10));       // NaN on the first step, as in the reference
                const bool conv = (rel_dec < (double)rel_error) || (tn < (double)delta_norm);
                L.st[WS_ITERS] = sin_[WS_ITERS] + 1.0;
                L.st[WS_TAUN] = tn;
                L.st[WS_COST] = cost;
                L.st[WS_OLD] = cost;
                if (conv && !fixed_iters) { L.st[WS_DONE] = 1.0; L.st[WS_CONV] = 1.0; }
            }
        }
    }
    __syncthreads();
    if (writer && t < WS_STATE) sout[t] = L.st[t];
    T_out = load_pose<float>(L.st + WS_T);
    return L.st[WS_DONE] != 0.0;
}

// What an accumulation launch needs to run the previous step in its prologue.  steps = number of steps solved BEFORE this
// launch's linearisation (0 for the first launch of a solve: the state is the initial one, nothing to solve).
struct StepArgs {
    int steps; float rel_error, delta_norm; int fixed_iters, nblk;
    const float *T_WCf, *T_WCk;       // [P][8] (steps == 0 only): the solve's initial T_CkCf = T_WCk^-1 T_WCf is formed here;
};                                    // null = the state slot was initialised by k_track_init (m3_track_normal_eq)

// Returns true when this workgroup has nothing to accumulate (problem finished); otherwise T = the pose to linearise at.
__device__ __forceinline__ bool begin_iteration(double *__restrict__ ws, const StepArgs &sa, StepLds &L, Pose<float> &T) {
    if (sa.steps == 0) {
        double *st = ws_state(ws, 0);
        if (sa.T_WCf) {                                    // first launch of a solve: every thread forms the initial pose
            const size_t pb = blockIdx.y;                  // (uniform loads, ~150 flops), workgroup 0 records the state
            const Pose<double> T0 = mul(inv_mlx(load_pose<double>(sa.T_WCk + 8 * pb)), load_pose<double>(sa.T_WCf + 8 * pb));
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                store_pose(st + WS_T, T0);
                st[WS_OLD] = INFINITY;
                st[WS_DONE] = 0.0; st[WS_ITERS] = 0.0; st[WS_TAUN] = 0.0; st[WS_COST] = 0.0; st[WS_CONV] = 0.0;
            }
            double tmp[8];
            store_pose(tmp, T0);
            T = load_pose<float>(tmp);
            return false;
        }
        if (st[WS_DONE] != 0.0) return true;
        T = load_pose<float>(st + WS_T);
        return false;
    }
    return track_step(ws, sa.steps, blockIdx.x == 0, sa.rel_error, sa.delta_norm, sa.fixed_iters, sa.nblk, L, T);
}

__device__ __forceinline__ void block_reduce_store(const double *acc, double *__restrict__ out) { m3_block_reduce36(acc, out); }

// One point's contribution to the 36 sums, ADDED into h (fp32).  Every Jacobian row is J_c = rsi_c * M(p) a_c
// with M(p) = [I; [p]x; p^T] (7x3: J = [a, p x a, a.p], geometry.py:118-137) and a_c = -(row c of d(ray,dist)/dP),
// so   H = M A M^T,  g = -M b,  A = sum_c w_c a_c a_c^T (3x3 symmetric),  b = sum_c w_c res_c a_c,  w_c = rsi_c^2:
// the 4 x (28 + 7) products of the row-by-row form become 36 FMAs for (A, b) and ~60 for the congruence.
__device__ __forceinline__ void track_point(const Pose<float> &T, const V3<float> &xf, const V3<float> &xk, float q,
                                            float huber_k, float inv_sigma_ray, float inv_sigma_dist, float (&h)[kSums]) {
    // v_sqrt_f32 / v_rcp_f32 (1 ulp) instead of the correctly rounded expansions of this build's sqrtf and '/'
    // (~10 instructions each, 9 of them per point): the sums are compared with the float64 oracle at 2e-5
    const float sq = __builtin_amdgcn_sqrtf(q);
    const float si_ray = inv_sigma_ray * sq, si_dist = inv_sigma_dist * sq;
    const V3<float> p = act(T, xf);
    const float d = __builtin_amdgcn_sqrtf(dot(p, p) + 1e-10f), di = __builtin_amdgcn_rcpf(d);
    const V3<float> r = di * p;
    const float dk = __builtin_amdgcn_sqrtf(dot(xk, xk) + 1e-10f), dki = __builtin_amdgcn_rcpf(dk);
    const V3<float> rk = dki * xk;
    const float res[4] = {rk.x - r.x, rk.y - r.y, rk.z - r.z, dk - d};
    const float di2 = di * di;
    const V3<float> a[4] = {
        {-di * (1.0f - di2 * p.x * p.x), di * di2 * p.x * p.y, di * di2 * p.x * p.z},
        {di * di2 * p.y * p.x, -di * (1.0f - di2 * p.y * p.y), di * di2 * p.y * p.z},
        {di * di2 * p.z * p.x, di * di2 * p.z * p.y, -di * (1.0f - di2 * p.z * p.z)},
        {-r.x, -r.y, -r.z}};
    float Axx = 0.f, Axy = 0.f, Axz = 0.f, Ayy = 0.f, Ayz = 0.f, Azz = 0.f, cost = 0.f;
    V3<float> b{0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float si = (c < 3) ? si_ray : si_dist;
        const float wr = fabsf(si * res[c]);
        const float hub = (wr < huber_k) ? 1.0f : huber_k * __builtin_amdgcn_rcpf(wr);
        const float w = si * si * hub;                         // (si * sqrt(hub))^2
        const V3<float> wa = w * a[c];
        Axx += wa.x * a[c].x; Axy += wa.x * a[c].y; Axz += wa.x * a[c].z;
        Ayy += wa.y * a[c].y; Ayz += wa.y * a[c].z; Azz += wa.z * a[c].z;
        const float wres = w * res[c];
        b = b + wres * a[c];
        cost += 0.5f * wres * res[c];
    }
    accum_congruence(p, Axx, Axy, Axz, Ayy, Ayz, Azz, b, 1.0f, -1.0f, h);
    h[35] += cost;
}

// 4 consecutive points per lane and trip: 3 + 3 + 1 dwordx4 loads (+ 4 validity bytes) instead of 28 dword loads,
// the next group's loads issued before this group's arithmetic; the <= 4 points' sums are added in fp32 (four
// terms, 1.2e-7 relative) and folded into the 36 float64 accumulators once per group - 36 conversions + adds per
// FOUR points where the first version spent 144 per point (it ran at the float64 VALU rate, 1.3 TB/s of 29 B points).
__global__ void __launch_bounds__(kThreads)
k_track_accum(const float *__restrict__ Xf, const float *__restrict__ Xk, const float *__restrict__ Qk,
              const uint8_t *__restrict__ valid, double *__restrict__ ws, int N, float huber_k,
              float inv_sigma_ray, float inv_sigma_dist, const StepArgs sa) {
    {
        const size_t pb = blockIdx.y;
        Xf += pb * N * 3; Xk += pb * N * 3; Qk += pb * N; valid += pb * N; ws += pb * WS_STRIDE;
    }
    __shared__ StepLds step_lds;
    Pose<float> T;
    if (begin_iteration(ws, sa, step_lds, T)) return;
    double acc[kSums];
#pragma unroll
    for (int i = 0; i < kSums; ++i) acc[i] = 0.0;
    const bool vec = (N % 4 == 0) && ((reinterpret_cast<size_t>(Xf) | reinterpret_cast<size_t>(Xk) | reinterpret_cast<size_t>(Qk)) % 16 == 0) &&
                     (reinterpret_cast<size_t>(valid) % 4 == 0);
    if (vec) {
        const int groups = N / 4, stride = gridDim.x * kThreads;
        int gi = blockIdx.x * kThreads + threadIdx.x;
        float4 f[3], k[3], q;
        unsigned v = 0;
        auto load = [&](int g) {
            const float4 *pf = reinterpret_cast<const float4 *>(Xf) + 3 * (size_t)g, *pk = reinterpret_cast<const float4 *>(Xk) + 3 * (size_t)g;
            f[0] = pf[0]; f[1] = pf[1]; f[2] = pf[2];
            k[0] = pk[0]; k[1] = pk[1]; k[2] = pk[2];
            q = reinterpret_cast<const float4 *>(Qk)[g];
            v = reinterpret_cast<const unsigned *>(valid)[g];
        };
        if (gi < groups) load(gi);
        while (gi < groups) {
            const float4 f0 = f[0], f1 = f[1], f2 = f[2], k0 = k[0], k1 = k[1], k2 = k[2], qq = q;
            const unsigned vv = v;
            const int nx = gi + stride;
            if (nx < groups) load(nx);                        // in flight under the arithmetic below
            float h[kSums];
#pragma unroll
            for (int i = 0; i < kSums; ++i) h[i] = 0.f;
            if (vv & 0x000000ffu) track_point(T, V3<float>{f0.x, f0.y, f0.z}, V3<float>{k0.x, k0.y, k0.z}, qq.x, huber_k, inv_sigma_ray, inv_sigma_dist, h);
            if (vv & 0x0000ff00u) track_point(T, V3<float>{f0.w, f1.x, f1.y}, V3<float>{k0.w, k1.x, k1.y}, qq.y, huber_k, inv_sigma_ray, inv_sigma_dist, h);
            if (vv & 0x00ff0000u) track_point(T, V3<float>{f1.z, f1.w, f2.x}, V3<float>{k1.z, k1.w, k2.x}, qq.z, huber_k, inv_sigma_ray, inv_sigma_dist, h);
            if (vv & 0xff000000u) track_point(T, V3<float>{f2.y, f2.z, f2.w}, V3<float>{k2.y, k2.z, k2.w}, qq.w, huber_k, inv_sigma_ray, inv_sigma_dist, h);
#pragma unroll
            for (int i = 0; i < kSums; ++i) acc[i] += (double)h[i];
            gi = nx;
        }
    } else {
        for (int n = blockIdx.x * kThreads + threadIdx.x; n < N; n += gridDim.x * kThreads) {
            if (!valid[n]) continue;
            float h[kSums];
#pragma unroll
            for (int i = 0; i < kSums; ++i) h[i] = 0.f;
            track_point(T, V3<float>{Xf[3 * n], Xf[3 * n + 1], Xf[3 * n + 2]}, V3<float>{Xk[3 * n], Xk[3 * n + 1], Xk[3 * n + 2]},
                        Qk[n], huber_k, inv_sigma_ray, inv_sigma_dist, h);
#pragma unroll
            for (int i = 0; i < kSums; ++i) acc[i] += (double)h[i];
        }
    }

    block_reduce_store(acc, ws_part(ws, sa.steps) + blockIdx.x * kSums);
}

// Calibrated variant (tracker.py:326-406, project_calib geometry.py:156-227): residual
// (u, v, log z)_keyframe-pixel - project(T . Xf), rows weighted 1/sigma_pixel (x2), 1/sigma_depth.
struct TrackCalib { float fx, fy, cx, cy; int W, H; float border, z_eps; };

// One point of the calibrated residual, same accumulation form as track_point: the rows are J_c = rsi_c M(p) a_c with
// a_c = -(row c of d(u, v, log z)/dP), so H = M A M^T with A = sum_c w_c a_c a_c^T and g = -M b.  (px, py) = the point's
// own pixel in the keyframe.  Returns without contributing when the measurement or the projection is invalid
// (tracker.py:207, geometry.py:186-190).
__device__ __forceinline__ void track_point_calib(const Pose<float> &T, const V3<float> &xf, float zk, float q, float px, float py,
                                                  float huber_k, float inv_sigma_pixel, float inv_sigma_depth,
                                                  const TrackCalib &cal, float (&h)[kSums]) {
    if (!(zk > cal.z_eps)) return;
    const V3<float> p = act(T, xf);
    const float zi = __builtin_amdgcn_rcpf(p.z + 1e-10f);
    const float u = cal.fx * p.x * zi + cal.cx, v = cal.fy * p.y * zi + cal.cy;          // K p / (z + 1e-10)
    const bool vp = (u > cal.border) && (u < (float)(cal.W - 1) - cal.border) && (v > cal.border) &&
                    (v < (float)(cal.H - 1) - cal.border) && (p.z > cal.z_eps);
    if (!vp) return;
    const float sq = __builtin_amdgcn_sqrtf(q);
    const float si_px = inv_sigma_pixel * sq, si_d = inv_sigma_depth * sq;
    const float res[3] = {px - u, py - v, __logf(zk + 1e-10f) - __logf(p.z + 1e-10f)};
    const V3<float> a[3] = {{-cal.fx * zi, 0.f, cal.fx * p.x * zi * zi},
                            {0.f, -cal.fy * zi, cal.fy * p.y * zi * zi},
                            {0.f, 0.f, -zi}};
    float Axx = 0.f, Axy = 0.f, Axz = 0.f, Ayy = 0.f, Ayz = 0.f, Azz = 0.f, cost = 0.f;
    V3<float> b{0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float si = (c < 2) ? si_px : si_d;
        const float wr = fabsf(si * res[c]);
        const float hub = (wr < huber_k) ? 1.0f : huber_k * __builtin_amdgcn_rcpf(wr);
        const float w = si * si * hub;
        const V3<float> wa = w * a[c];
        Axx += wa.x * a[c].x; Axy += wa.x * a[c].y; Axz += wa.x * a[c].z;
        Ayy += wa.y * a[c].y; Ayz += wa.y * a[c].z; Azz += wa.z * a[c].z;
        const float wres = w * res[c];
        b = b + wres * a[c];
        cost += 0.5f * wres * res[c];
    }
    accum_congruence(p, Axx, Axy, Axz, Ayy, Ayz, Azz, b, 1.0f, -1.0f, h);
    h[35] += cost;
}

// Same streaming form as k_track_accum: 4 consecutive points per lane (3 dwordx4 of Xf, the z of Xk out of 3 more, Qk, 4
// validity bytes; the next group in flight under the arithmetic), fp32 sums over the group, one float64 fold per group.
// (The first version walked one point per lane with dword loads, an integer division per point and 108 float64
// conversions + adds per point: 49.6 us per iteration against 25.5 us for the ray-distance solve on the same streams.)
__global__ void __launch_bounds__(kThreads)
k_track_accum_calib(const float *__restrict__ Xf, const float *__restrict__ Xk, const float *__restrict__ Qk,
                    const uint8_t *__restrict__ valid, double *__restrict__ ws, int N, float huber_k,
                    float inv_sigma_pixel, float inv_sigma_depth, const TrackCalib cal, const StepArgs sa) {
    {
        const size_t pb = blockIdx.y;
        Xf += pb * N * 3; Xk += pb * N * 3; Qk += pb * N; valid += pb * N; ws += pb * WS_STRIDE;
    }
    __shared__ StepLds step_lds;
    Pose<float> T;
    if (begin_iteration(ws, sa, step_lds, T)) return;
    double acc[kSums];
#pragma unroll
    for (int i = 0; i < kSums; ++i) acc[i] = 0.0;
    const bool vec = (N % 4 == 0) && (cal.W % 4 == 0) &&
                     ((reinterpret_cast<size_t>(Xf) | reinterpret_cast<size_t>(Xk) | reinterpret_cast<size_t>(Qk)) % 16 == 0) &&
                     (reinterpret_cast<size_t>(valid) % 4 == 0);
    if (vec) {
        const int groups = N / 4, stride = gridDim.x * kThreads;
        int gi = blockIdx.x * kThreads + threadIdx.x;
        float4 f[3], k[3], q;
        unsigned v = 0;
        auto load = [&](int g) {
            const float4 *pf = reinterpret_cast<const float4 *>(Xf) + 3 * (size_t)g, *pk = reinterpret_cast<const float4 *>(Xk) + 3 * (size_t)g;
            f[0] = pf[0]; f[1] = pf[1]; f[2] = pf[2];
            k[0] = pk[0]; k[1] = pk[1]; k[2] = pk[2];
            q = reinterpret_cast<const float4 *>(Qk)[g];
            v = reinterpret_cast<const unsigned *>(valid)[g];
        };
        if (gi < groups) load(gi);
        while (gi < groups) {
            const float4 f0 = f[0], f1 = f[1], f2 = f[2], qq = q;
            const float z0 = k[0].z, z1 = k[1].y, z2 = k[2].x, z3 = k[2].w;          // z of the four keyframe points
            const unsigned vv = v;
            const int n0 = 4 * gi;
            const int nx = gi + stride;
            if (nx < groups) load(nx);                        // in flight under the arithmetic below
            const int row = n0 / cal.W;                       // W % 4 == 0: the four points share a row
            const float py = (float)row, px = (float)(n0 - row * cal.W);
            float h[kSums];
#pragma unroll
            for (int i = 0; i < kSums; ++i) h[i] = 0.f;
            if (vv & 0x000000ffu) track_point_calib(T, V3<float>{f0.x, f0.y, f0.z}, z0, qq.x, px, py, huber_k, inv_sigma_pixel, inv_sigma_depth, cal, h);
            if (vv & 0x0000ff00u) track_point_calib(T, V3<float>{f0.w, f1.x, f1.y}, z1, qq.y, px + 1.0f, py, huber_k, inv_sigma_pixel, inv_sigma_depth, cal, h);
            if (vv & 0x00ff0000u) track_point_calib(T, V3<float>{f1.z, f1.w, f2.x}, z2, qq.z, px + 2.0f, py, huber_k, inv_sigma_pixel, inv_sigma_depth, cal, h);
            if (vv & 0xff000000u) track_point_calib(T, V3<float>{f2.y, f2.z, f2.w}, z3, qq.w, px + 3.0f, py, huber_k, inv_sigma_pixel, inv_sigma_depth, cal, h);
#pragma unroll
            for (int i = 0; i < kSums; ++i) acc[i] += (double)h[i];
            gi = nx;
        }
    } else {
        for (int n = blockIdx.x * kThreads + threadIdx.x; n < N; n += gridDim.x * kThreads) {
            if (!valid[n]) continue;
            const int row = n / cal.W;
            float h[kSums];
#pragma unroll
            for (int i = 0; i < kSums; ++i) h[i] = 0.f;
            track_point_calib(T, V3<float>{Xf[3 * n], Xf[3 * n + 1], Xf[3 * n + 2]}, Xk[3 * n + 2], Qk[n], (float)(n - row * cal.W),
                              (float)row, huber_k, inv_sigma_pixel, inv_sigma_depth, cal, h);
#pragma unroll
            for (int i = 0; i < kSums; ++i) acc[i] += (double)h[i];
        }
    }
    block_reduce_store(acc, ws_part(ws, sa.steps) + blockIdx.x * kSums);
}

// constrain_points_to_ray (geometry.py:273-302): keep z, move the point onto its pixel's ray
__global__ void __launch_bounds__(kThreads)
k_constrain_to_ray(const float *__restrict__ X, float *__restrict__ out, int N, int W, float fx, float fy, float cx,
                   float cy) {
    const size_t pb = blockIdx.y;
    const int n = blockIdx.x * kThreads + threadIdx.x;
    if (n >= N) return;
    const float z = X[(pb * N + n) * 3 + 2];
    const int py = n / W, px = n - py * W;
    float *o = out + (pb * N + n) * 3;
    o[0] = ((float)px - cx) / fx * z; o[1] = ((float)py - cy) / fy * z; o[2] = z;
}

// Results of a solve from its final state (one thread).
// status: 0 = iteration budget used up, 1 = converged (optimizer.py:11-46), 2 = the solve FAILED (singular normal
// matrix or a divergent step: the pose is the last good one) - what the reference reports by raising inside
// _opt_pose_* (tracker.py:121-141), the caller then relocalises
__device__ __forceinline__ void export_state(const double *st, const float *__restrict__ T_WCk, float *__restrict__ T_WCf_out,
                                             float *__restrict__ T_rel_out, double *__restrict__ info) {
    const Pose<double> T = load_pose<double>(st + WS_T);
    store_pose(T_rel_out, T);
    store_pose(T_WCf_out, mul(load_pose<double>(T_WCk), T));
    info[0] = st[WS_ITERS]; info[1] = st[WS_COST]; info[2] = st[WS_TAUN];
    info[3] = st[WS_DONE] == 2.0 ? 2.0 : st[WS_CONV];
}

// The LAST step of a solve has no accumulation behind it to ride in: one workgroup per problem, which also exports the
// results (T_WCf = T_WCk T_CkCf, T_CkCf, info).
__global__ void __launch_bounds__(kThreads)
k_track_solve(double *__restrict__ ws, int steps, float rel_error, float delta_norm, int fixed_iters, int nblk,
              const float *__restrict__ T_WCk, float *__restrict__ T_WCf_out, float *__restrict__ T_rel_out,
              double *__restrict__ info) {
    const size_t pb = blockIdx.x;
    ws += pb * WS_STRIDE;
    __shared__ StepLds L;
    Pose<float> T;
    track_step(ws, steps, true, rel_error, delta_norm, fixed_iters, nblk, L, T);
    if (threadIdx.x == 0) export_state(L.st, T_WCk + 8 * pb, T_WCf_out + 8 * pb, T_rel_out + 8 * pb, info + 4 * pb);
}

// max_iters == 0: nothing was solved, the results are the initial state (k_track_init)
__global__ void k_track_final(const double *__restrict__ ws, const float *__restrict__ T_WCk,
                              float *__restrict__ T_WCf_out, float *__restrict__ T_rel_out,
                              double *__restrict__ info) {
    if (threadIdx.x != 0) return;
    const size_t pb = blockIdx.x;
    export_state(ws + pb * WS_STRIDE, T_WCk + 8 * pb, T_WCf_out + 8 * pb, T_rel_out + 8 * pb, info + 4 * pb);
}

__global__ void __launch_bounds__(kThreads)
k_track_export(const double *__restrict__ ws, double *__restrict__ out, int nblk) {
    ws += (size_t)blockIdx.x * WS_STRIDE; out += (size_t)blockIdx.x * kSums;
    __shared__ double sums[kSums];
    reduce_partials(ws + WS_PART, sums, nblk);
    if (threadIdx.x < kSums) out[threadIdx.x] = sums[threadIdx.x];
}

// One gathered point: Xf = Xf_canon[idx], Qk = sqrt(Qff[idx] * Qkf), the two validity masks (tracker.py:88-113, :177-214).
struct Gathered { float x, y, z, q; int vo, vk; };
__device__ __forceinline__ Gathered gather_point(const float *__restrict__ Xf_canon, const float *__restrict__ Cf_avg,
                                                 const float *__restrict__ Qff, int64_t id, float qkf, float ck, int vm,
                                                 int N, float C_conf, float Q_conf) {
    if (id < 0) id += N;
    id = id < 0 ? 0 : (id >= N ? N - 1 : id);
    Gathered g;
    // one 12-byte load per gathered row (global_load_dwordx3 needs dword alignment only) instead of three dword loads
    struct __attribute__((packed, aligned(4))) Row3 { float x, y, z; };
    const Row3 row = reinterpret_cast<const Row3 *>(Xf_canon)[id];
    g.x = row.x; g.y = row.y; g.z = row.z;
    g.q = sqrtf(Qff[id] * qkf);
    g.vk = (vm != 0) && (g.q > Q_conf);
    g.vo = g.vk && (Cf_avg[id] > C_conf) && (ck > C_conf);
    return g;
}

// VEC: four consecutive points per lane - the per-point streams (idx, Qkf, Ck, valid_match in; Xf, Qk, the two masks out)
// move as 16-byte (4-byte for the masks) accesses; only the three gathers per point stay scalar.  The first version
// handled one point per lane: 12-byte-strided dword stores and ONE-byte mask stores, 0.13 of the HBM roof.
template <bool VEC>
__global__ void __launch_bounds__(kThreads)
k_track_gather(const float *__restrict__ Xf_canon, const float *__restrict__ Cf_avg,
               const float *__restrict__ Ck_avg, const float *__restrict__ Qff, const float *__restrict__ Qkf,
               const int64_t *__restrict__ idx, const uint8_t *__restrict__ valid_match,
               float *__restrict__ Xf_g, float *__restrict__ Qk, uint8_t *__restrict__ valid_opt,
               uint8_t *__restrict__ valid_kf, int32_t *__restrict__ counts, int N, float C_conf, float Q_conf) {
    {
        const size_t pb = blockIdx.y;
        Xf_canon += pb * N * 3; Cf_avg += pb * N; Ck_avg += pb * N; Qff += pb * N; Qkf += pb * N; idx += pb * N;
        valid_match += pb * N; Xf_g += pb * N * 3; Qk += pb * N; valid_opt += pb * N; valid_kf += pb * N; counts += 2 * pb;
    }
    int vo = 0, vk = 0;                                   // number of valid points of this lane
    if constexpr (VEC) {
        const int g4 = blockIdx.x * kThreads + threadIdx.x;
        if (g4 < N / 4) {
            const longlong2 i01 = reinterpret_cast<const longlong2 *>(idx)[2 * g4], i23 = reinterpret_cast<const longlong2 *>(idx)[2 * g4 + 1];
            const float4 qk4 = reinterpret_cast<const float4 *>(Qkf)[g4], ck4 = reinterpret_cast<const float4 *>(Ck_avg)[g4];
            const unsigned vm4 = reinterpret_cast<const unsigned *>(valid_match)[g4];
            const Gathered a = gather_point(Xf_canon, Cf_avg, Qff, i01.x, qk4.x, ck4.x, vm4 & 0xffu, N, C_conf, Q_conf);
            const Gathered b = gather_point(Xf_canon, Cf_avg, Qff, i01.y, qk4.y, ck4.y, vm4 & 0xff00u, N, C_conf, Q_conf);
            const Gathered c = gather_point(Xf_canon, Cf_avg, Qff, i23.x, qk4.z, ck4.z, vm4 & 0xff0000u, N, C_conf, Q_conf);
            const Gathered d = gather_point(Xf_canon, Cf_avg, Qff, i23.y, qk4.w, ck4.w, vm4 & 0xff000000u, N, C_conf, Q_conf);
            float4 *xo = reinterpret_cast<float4 *>(Xf_g) + 3 * (size_t)g4;
            xo[0] = make_float4(a.x, a.y, a.z, b.x);
            xo[1] = make_float4(b.y, b.z, c.x, c.y);
            xo[2] = make_float4(c.z, d.x, d.y, d.z);
            reinterpret_cast<float4 *>(Qk)[g4] = make_float4(a.q, b.q, c.q, d.q);
            reinterpret_cast<unsigned *>(valid_opt)[g4] = (unsigned)a.vo | ((unsigned)b.vo << 8) | ((unsigned)c.vo << 16) | ((unsigned)d.vo << 24);
            reinterpret_cast<unsigned *>(valid_kf)[g4] = (unsigned)a.vk | ((unsigned)b.vk << 8) | ((unsigned)c.vk << 16) | ((unsigned)d.vk << 24);
            vo = a.vo + b.vo + c.vo + d.vo;
            vk = a.vk + b.vk + c.vk + d.vk;
        }
    } else {
        const int n = blockIdx.x * kThreads + threadIdx.x;
        if (n < N) {
            const Gathered a = gather_point(Xf_canon, Cf_avg, Qff, idx[n], Qkf[n], Ck_avg[n], valid_match[n], N, C_conf, Q_conf);
            Xf_g[3 * n + 0] = a.x; Xf_g[3 * n + 1] = a.y; Xf_g[3 * n + 2] = a.z;
            Qk[n] = a.q;
            valid_opt[n] = (uint8_t)a.vo;
            valid_kf[n] = (uint8_t)a.vk;
            vo = a.vo; vk = a.vk;
        }
    }
    // one atomic pair per BLOCK (same-address atomics serialise at ~12 ns each)
    __shared__ int cnt[2];
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { vo += __shfl_down(vo, off, 64); vk += __shfl_down(vk, off, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (vo) atomicAdd(&cnt[0], vo);
        if (vk) atomicAdd(&cnt[1], vk);
    }
    __syncthreads();
    if (threadIdx.x < 2 && cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], cnt[threadIdx.x]);
}

// Tiled form for spatially coherent matches (the tracker's regime: neighbouring keyframe pixels match neighbouring frame
// pixels).  A workgroup owns 1024 consecutive keyframe points; when the frame indices they point at span at most kGatherCap
// points, that contiguous range of Xf_canon / Cf / Qff is staged in LDS with coalesced 16-byte loads and the three
// data-dependent gathers per point read LDS instead of issuing 12 scattered global loads per lane (one 32-byte sector per
// 4-12 useful bytes: the round-3 kernel moved 1.52 x its algorithmic bytes and sat at 0.27-0.30 of the HBM roof).
// Workgroups whose matches are scattered take the global-gather path.  Same values, same arithmetic: same bits.
constexpr int kGatherCap = 3072;                      // staged frame points: 20 B each = 60 KiB of LDS
__global__ void __launch_bounds__(kThreads)
k_track_gather_lds(const float *__restrict__ Xf_canon, const float *__restrict__ Cf_avg,
                   const float *__restrict__ Ck_avg, const float *__restrict__ Qff, const float *__restrict__ Qkf,
                   const int64_t *__restrict__ idx, const uint8_t *__restrict__ valid_match,
                   float *__restrict__ Xf_g, float *__restrict__ Qk, uint8_t *__restrict__ valid_opt,
                   uint8_t *__restrict__ valid_kf, int32_t *__restrict__ counts, int N, float C_conf, float Q_conf) {
    extern __shared__ __attribute__((aligned(16))) float gl[];
    float *Xs = gl, *Cs = gl + 3 * kGatherCap, *Qs = Cs + kGatherCap;
    __shared__ int rng[2], cnt[2];
    {
        const size_t pb = blockIdx.y;
        Xf_canon += pb * N * 3; Cf_avg += pb * N; Ck_avg += pb * N; Qff += pb * N; Qkf += pb * N; idx += pb * N;
        valid_match += pb * N; Xf_g += pb * N * 3; Qk += pb * N; valid_opt += pb * N; valid_kf += pb * N; counts += 2 * pb;
    }
    const int tid = threadIdx.x, g4 = blockIdx.x * kThreads + tid;
    const bool live = g4 < N / 4;
    if (tid == 0) { rng[0] = INT_MAX; rng[1] = INT_MIN; cnt[0] = 0; cnt[1] = 0; }
    int id[4] = {0, 0, 0, 0};
    float4 qk4 = make_float4(0.f, 0.f, 0.f, 0.f), ck4 = qk4;
    unsigned vm4 = 0;
    if (live) {
        const longlong2 i01 = reinterpret_cast<const longlong2 *>(idx)[2 * g4], i23 = reinterpret_cast<const longlong2 *>(idx)[2 * g4 + 1];
        qk4 = reinterpret_cast<const float4 *>(Qkf)[g4]; ck4 = reinterpret_cast<const float4 *>(Ck_avg)[g4];
        vm4 = reinterpret_cast<const unsigned *>(valid_match)[g4];
        const long long raw[4] = {i01.x, i01.y, i23.x, i23.y};
#pragma unroll
        for (int k = 0; k < 4; ++k) {                        // the index normalisation of gather_point
            long long v = raw[k];
            if (v < 0) v += N;
            id[k] = (int)(v < 0 ? 0 : (v >= N ? N - 1 : v));
        }
    }
    int lo = live ? min(min(id[0], id[1]), min(id[2], id[3])) : INT_MAX;
    int hi = live ? max(max(id[0], id[1]), max(id[2], id[3])) : INT_MIN;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_xor(lo, off, 64)); hi = max(hi, __shfl_xor(hi, off, 64)); }
    __syncthreads();
    if ((tid & 63) == 0) { atomicMin(&rng[0], lo); atomicMax(&rng[1], hi); }
    __syncthreads();
    const int base = rng[0] & ~3, span = rng[1] - base + 1;   // base: a multiple of 4 points = 16-byte aligned in all three arrays
    const bool staged = rng[0] <= rng[1] && span <= kGatherCap;   // workgroup-uniform
    if (staged) {
        const int n4 = (span + 3) / 4;                          // groups of 4 points; the tail group may reach past N - 1:
        const int last4 = (N - base) / 4;                       //   whole groups that exist (N % 4 == 0, base % 4 == 0)
        for (int i = tid; i < n4; i += kThreads) {
            const int j = i < last4 ? i : last4 - 1;
            reinterpret_cast<float4 *>(Cs)[i] = reinterpret_cast<const float4 *>(Cf_avg + base)[j];
            reinterpret_cast<float4 *>(Qs)[i] = reinterpret_cast<const float4 *>(Qff + base)[j];
        }
        for (int i = tid; i < 3 * n4; i += kThreads) {
            const int j = i < 3 * last4 ? i : 3 * last4 - 1;
            reinterpret_cast<float4 *>(Xs)[i] = reinterpret_cast<const float4 *>(Xf_canon + (size_t)base * 3)[j];
        }
    }
    __syncthreads();
    int vo = 0, vk = 0;
    if (live) {
        Gathered r[4];
        const float qkf[4] = {qk4.x, qk4.y, qk4.z, qk4.w}, ck[4] = {ck4.x, ck4.y, ck4.z, ck4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int vm = (int)((vm4 >> (8 * k)) & 0xffu);
            float x, y, z, qf, cf;
            if (staged) {
                const int o = id[k] - base;
                x = Xs[3 * o]; y = Xs[3 * o + 1]; z = Xs[3 * o + 2]; qf = Qs[o]; cf = Cs[o];
            } else {
                x = Xf_canon[3 * (size_t)id[k]]; y = Xf_canon[3 * (size_t)id[k] + 1]; z = Xf_canon[3 * (size_t)id[k] + 2];
                qf = Qff[id[k]]; cf = Cf_avg[id[k]];
            }
            r[k].x = x; r[k].y = y; r[k].z = z;
            r[k].q = sqrtf(qf * qkf[k]);
            r[k].vk = (vm != 0) && (r[k].q > Q_conf);
            r[k].vo = r[k].vk && (cf > C_conf) && (ck[k] > C_conf);
        }
        float4 *xo = reinterpret_cast<float4 *>(Xf_g) + 3 * (size_t)g4;
        xo[0] = make_float4(r[0].x, r[0].y, r[0].z, r[1].x);
        xo[1] = make_float4(r[1].y, r[1].z, r[2].x, r[2].y);
        xo[2] = make_float4(r[2].z, r[3].x, r[3].y, r[3].z);
        reinterpret_cast<float4 *>(Qk)[g4] = make_float4(r[0].q, r[1].q, r[2].q, r[3].q);
        reinterpret_cast<unsigned *>(valid_opt)[g4] = (unsigned)r[0].vo | ((unsigned)r[1].vo << 8) | ((unsigned)r[2].vo << 16) | ((unsigned)r[3].vo << 24);
        reinterpret_cast<unsigned *>(valid_kf)[g4] = (unsigned)r[0].vk | ((unsigned)r[1].vk << 8) | ((unsigned)r[2].vk << 16) | ((unsigned)r[3].vk << 24);
        vo = r[0].vo + r[1].vo + r[2].vo + r[3].vo;
        vk = r[0].vk + r[1].vk + r[2].vk + r[3].vk;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { vo += __shfl_down(vo, off, 64); vk += __shfl_down(vk, off, 64); }
    if ((tid & 63) == 0) {
        if (vo) atomicAdd(&cnt[0], vo);
        if (vk) atomicAdd(&cnt[1], vk);
    }
    __syncthreads();
    if (tid < 2 && cnt[tid]) atomicAdd(&counts[tid], cnt[tid]);
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

int64_t m3_track_ws_doubles(void) { return WS_STRIDE; }

int m3_track_gather_batch(const float *Xf_canon, const float *Cf_avg, const float *Ck_avg, const float *Qff,
                          const float *Qkf, const int64_t *idx, const uint8_t *valid_match, float *Xf_g,
                          float *Qk, uint8_t *valid_opt, uint8_t *valid_kf, int32_t *counts, int P, int N,
                          float C_conf, float Q_conf, void *stream) {
    M3_REQUIRE(Xf_canon && Cf_avg && Ck_avg && Qff && Qkf && idx && valid_match);
    M3_REQUIRE(Xf_g && Qk && valid_opt && valid_kf && counts && N > 0 && P > 0 && P <= 65535);
    hipStream_t st = (hipStream_t)stream;
    M3_CHECK_HIP(hipMemsetAsync(counts, 0, 2 * P * sizeof(int32_t), st), "m3_track_gather/memset");
    const uintptr_t al = (uintptr_t)idx | (uintptr_t)Qkf | (uintptr_t)Ck_avg | (uintptr_t)Xf_g | (uintptr_t)Qk;
    const uintptr_t al4 = (uintptr_t)valid_match | (uintptr_t)valid_opt | (uintptr_t)valid_kf;
    const uintptr_t alg = (uintptr_t)Xf_canon | (uintptr_t)Cf_avg | (uintptr_t)Qff;       // staged arrays: 16-byte loads from a 4-point base
    static const bool tiled = [] { const char *e = getenv("M3_GATHER_LDS"); return !(e && atoi(e) == 0); }();
    if (tiled && N % 4 == 0 && N >= 4 && ((al | alg) & 15) == 0 && (al4 & 3) == 0) {
        constexpr int kLds = kGatherCap * 20;
        static M3AttrOnce once;
        int dev__;
        if (m3_attr_need(once, &dev__)) {
            M3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_track_gather_lds),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, kLds), "m3_track_gather/attr");
            m3_attr_done(once, dev__);
        }
        hipLaunchKernelGGL(k_track_gather_lds, dim3(m3_cdiv(N / 4, kThreads), P), dim3(kThreads), kLds, st, Xf_canon, Cf_avg,
                           Ck_avg, Qff, Qkf, idx, valid_match, Xf_g, Qk, valid_opt, valid_kf, counts, N, C_conf, Q_conf);
    } else if (N % 4 == 0 && (al & 15) == 0 && (al4 & 3) == 0)       // per-problem strides are then multiples of 16 / 4 bytes too
        hipLaunchKernelGGL(k_track_gather<true>, dim3(m3_cdiv(N / 4, kThreads), P), dim3(kThreads), 0, st, Xf_canon, Cf_avg,
                           Ck_avg, Qff, Qkf, idx, valid_match, Xf_g, Qk, valid_opt, valid_kf, counts, N, C_conf, Q_conf);
    else
        hipLaunchKernelGGL(k_track_gather<false>, dim3(m3_cdiv(N, kThreads), P), dim3(kThreads), 0, st, Xf_canon, Cf_avg,
                           Ck_avg, Qff, Qkf, idx, valid_match, Xf_g, Qk, valid_opt, valid_kf, counts, N, C_conf, Q_conf);
    M3_CHECK_LAUNCH("m3_track_gather");
    return M3_OK;
}

int m3_track_gather(const float *Xf_canon, const float *Cf_avg, const float *Ck_avg, const float *Qff,
                    const float *Qkf, const int64_t *idx, const uint8_t *valid_match, float *Xf_g,
                    float *Qk, uint8_t *valid_opt, uint8_t *valid_kf, int32_t *counts, int N,
                    float C_conf, float Q_conf, void *stream) {
    return m3_track_gather_batch(Xf_canon, Cf_avg, Ck_avg, Qff, Qkf, idx, valid_match, Xf_g, Qk, valid_opt, valid_kf,
                                 counts, 1, N, C_conf, Q_conf, stream);
}

int m3_track_gn_ray_dist_batch(const float *Xf, const float *Xk, const float *Qk, const uint8_t *valid,
                               const float *T_WCf, const float *T_WCk, float *T_WCf_out, float *T_CkCf_out,
                               double *info, double *ws, int P, int N, int max_iters, float huber_k,
                               float sigma_ray, float sigma_dist, float rel_error, float delta_norm,
                               int fixed_iters, void *stream) {
    M3_REQUIRE(Xf && Xk && Qk && valid && T_WCf && T_WCk && T_WCf_out && T_CkCf_out && info && ws);
    M3_REQUIRE(N > 0 && P > 0 && P <= 65535 && max_iters >= 0 && sigma_ray > 0.f && sigma_dist > 0.f && huber_k > 0.f);
    hipStream_t st = (hipStream_t)stream;
    const float isr = (float)(1.0 / (double)sigma_ray), isd = (float)(1.0 / (double)sigma_dist);
    const int nblk = track_blocks(P);
    if (max_iters == 0) {
        hipLaunchKernelGGL(k_track_init, dim3(P), dim3(64), 0, st, T_WCf, T_WCk, (const float *)nullptr, ws);
        hipLaunchKernelGGL(k_track_final, dim3(P), dim3(64), 0, st, (const double *)ws, T_WCk, T_WCf_out, T_CkCf_out, info);
        M3_CHECK_LAUNCH("m3_track_gn/no iterations");
        return M3_OK;
    }
    // max_iters + 1 launches: launch 0 forms the initial pose and linearises at it; launch `it` solves step `it` (from the
    // partials of launch it - 1) in its prologue and linearises at the result; the last step and the export share a launch
    for (int it = 0; it < max_iters; ++it)
        hipLaunchKernelGGL(k_track_accum, dim3(nblk, P), dim3(kThreads), 0, st, Xf, Xk, Qk, valid, ws, N,
                           huber_k, isr, isd, StepArgs{it, rel_error, delta_norm, fixed_iters, nblk, T_WCf, T_WCk});
    hipLaunchKernelGGL(k_track_solve, dim3(P), dim3(kThreads), 0, st, ws, max_iters, rel_error, delta_norm, fixed_iters, nblk,
                       T_WCk, T_WCf_out, T_CkCf_out, info);
    M3_CHECK_LAUNCH("m3_track_gn/loop");
    return M3_OK;
}

int m3_track_gn_ray_dist(const float *Xf, const float *Xk, const float *Qk, const uint8_t *valid,
                         const float *T_WCf, const float *T_WCk, float *T_WCf_out, float *T_CkCf_out,
                         double *info, double *ws, int N, int max_iters, float huber_k, float sigma_ray,
                         float sigma_dist, float rel_error, float delta_norm, int fixed_iters, void *stream) {
    return m3_track_gn_ray_dist_batch(Xf, Xk, Qk, valid, T_WCf, T_WCk, T_WCf_out, T_CkCf_out, info, ws, 1, N,
                                      max_iters, huber_k, sigma_ray, sigma_dist, rel_error, delta_norm, fixed_iters,
                                      stream);
}

int m3_track_gn_calib_batch(const float *Xf, const float *Xk, const float *Qk, const uint8_t *valid,
                            const float *T_WCf, const float *T_WCk, float *T_WCf_out, float *T_CkCf_out,
                            double *info, double *ws, int P, int N, int H, int W, const float *K4, int max_iters,
                            float huber_k, float sigma_pixel, float sigma_depth, float pixel_border, float depth_eps,
                            float rel_error, float delta_norm, int fixed_iters, void *stream) {
    M3_REQUIRE(Xf && Xk && Qk && valid && T_WCf && T_WCk && T_WCf_out && T_CkCf_out && info && ws && K4);
    M3_REQUIRE(N > 0 && P > 0 && P <= 65535 && H > 0 && W > 0 && (int64_t)H * W == N && max_iters >= 0);
    M3_REQUIRE(sigma_pixel > 0.f && sigma_depth > 0.f && huber_k > 0.f);
    hipStream_t st = (hipStream_t)stream;
    TrackCalib cal{K4[0], K4[1], K4[2], K4[3], W, H, pixel_border, depth_eps};
    const float isp = (float)(1.0 / (double)sigma_pixel), isd = (float)(1.0 / (double)sigma_depth);
    const int nblk = track_blocks(P);
    if (max_iters == 0) {
        hipLaunchKernelGGL(k_track_init, dim3(P), dim3(64), 0, st, T_WCf, T_WCk, (const float *)nullptr, ws);
        hipLaunchKernelGGL(k_track_final, dim3(P), dim3(64), 0, st, (const double *)ws, T_WCk, T_WCf_out, T_CkCf_out, info);
        M3_CHECK_LAUNCH("m3_track_gn_calib/no iterations");
        return M3_OK;
    }
    for (int it = 0; it < max_iters; ++it)
        hipLaunchKernelGGL(k_track_accum_calib, dim3(nblk, P), dim3(kThreads), 0, st, Xf, Xk, Qk, valid, ws, N,
                           huber_k, isp, isd, cal, StepArgs{it, rel_error, delta_norm, fixed_iters, nblk, T_WCf, T_WCk});
    hipLaunchKernelGGL(k_track_solve, dim3(P), dim3(kThreads), 0, st, ws, max_iters, rel_error, delta_norm, fixed_iters, nblk,
                       T_WCk, T_WCf_out, T_CkCf_out, info);
    M3_CHECK_LAUNCH("m3_track_gn_calib/loop");
    return M3_OK;
}

int m3_constrain_points_to_ray(const float *X, float *out, int P, int H, int W, const float *K4, void *stream) {
    M3_REQUIRE(X && out && K4 && P > 0 && P <= 65535 && H > 0 && W > 0 && K4[0] != 0.f && K4[1] != 0.f);
    const int N = H * W;
    hipLaunchKernelGGL(k_constrain_to_ray, dim3(m3_cdiv(N, kThreads), P), dim3(kThreads), 0, (hipStream_t)stream, X, out,
                       N, W, K4[0], K4[1], K4[2], K4[3]);
    M3_CHECK_LAUNCH("m3_constrain_points_to_ray");
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
                       isr, isd, StepArgs{0, 0.f, 0.f, 1, kBlocks, nullptr, nullptr});
    hipLaunchKernelGGL(k_track_export, dim3(1), dim3(kThreads), 0, st, (const double *)ws, out, kBlocks);
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
