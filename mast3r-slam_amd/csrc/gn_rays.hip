// Backend Gauss-Newton ("rays" = 3-D point alignment) for gfx950.
//
// Replaces kernels.gauss_newton_rays (backends/mpsgraph/kernels.py:262-322), numpy twin
// gauss_newton.py:23-280, Metal gn_jacobian_kernel (gauss_newton.metal:66-252) and its HOST
// reduction (gn_metal_runner.py:221-292) of /root/reference/src/mlx_mast3r_slam.
//
// The Metal path writes 119 floats PER POINT (476 B) and reduces them on the host; here
// the per-point 7x7 contributions never leave registers: float32 per point, float64
// accumulation, wave-shuffle + LDS reduction to 36 doubles per (edge, chunk), a second
// fixed-order pass to 36 doubles per edge.  HBM bound: ~41 B per (edge, point).
// With the reference's "simplified adjoint" Ji = -Jj (gauss_newton.py:189-213) the five
// blocks of an edge are +-Hjj and +-gj, so only Hjj (28) and gj (7) are accumulated.
#include "common.h"
#include "sim3_dev.h"

// blocked float64 Cholesky solve (gn_chol.hip)
int m3_chol_solve_launch(double *H, double *b, double *y, double *x, double *Ld, double *fail, const double *off, int dim,
                         double shift, hipStream_t st);
int64_t m3_chol_diag_doubles(int dim);

namespace {

constexpr int kThreads = 256;
constexpr int kSums = 36;         // 28 Hjj upper + 7 gj + 1 count
constexpr int kMaxChunks = 128;
constexpr int kPtsPerChunk = 2048;
constexpr int kSolveThreads = 1024;
constexpr int kMaxDim = 63;       // 9 free keyframes: up to here ONE workgroup factors in place (63 serial steps); beyond, the blocked
                                  // Cholesky of gn_chol.hip (round 3: the 231-unknown system of a 34-keyframe graph took 2.1 ms in
                                  // the single-workgroup kernel - dim serial steps with three barriers each over a matrix in global memory)

// residual_mode: 0 = "rays" (3-D point error), 1 = "points" (+ 1/(|Xi|+1e-6) weight), 2 = "calib"
// (pixel + log-depth residual, gauss_newton_calib.py:17-274)
struct CalibParams {
    float fx, fy, cx, cy, width, height, border, z_eps, inv_sigma_pixel, inv_sigma_depth;
};

__host__ __device__ inline int gn_chunks(int P) {
    int c = (P + kPtsPerChunk - 1) / kPtsPerChunk;
    return c < 1 ? 1 : (c > kMaxChunks ? kMaxChunks : c);
}

// One point of edge (i, j): adds its 36 sums to h (fp32) in the UNROTATED tangent frame.  The pose Jacobian of
// the transformed point Y = Tij . Xj is JX = [R_i / s_i | B(Y) R_i | Y] = M'(Y) G with M' = [I / s_i | B | Y] and
// G = blockdiag(R_i, R_i, 1) constant per edge (gauss_newton.py:164-213), so the kernel accumulates M'^T A M' and
// M'^T b - the same congruence as the tracking solve - and k_gn_reduce applies G once per edge: no quaternion
// rotation per point.  A = sum_c w_c D_c D_c^T with D = d(residual)/dY (identity for the 3-D residuals).
// Returns false when the point is gated out.
template <int MODE>
__device__ __forceinline__ bool gn_point(const Pose<float> &Tij, float s_inv, const V3<float> &Xi, const V3<float> &Xj,
                                         float qc, float inv_sigma, const CalibParams &cal, float *h) {
    const V3<float> Y = act(Tij, Xj);
    float err[3] = {Y.x - Xi.x, Y.y - Xi.y, Y.z - Xi.z};
    float sqrt_w = inv_sigma * __builtin_amdgcn_sqrtf(qc);
    if (MODE == 1)                                      // gauss_newton_points.py:103-107: 1/(|Xi| + 1e-6)
        sqrt_w *= __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(dot(Xi, Xi)) + 1e-6f);
    float d0x = 1.f, d0z = 0.f, d1y = 1.f, d1z = 0.f, d2z = 1.f;
    if (MODE == 2) {                                    // gauss_newton_calib.py:118-176
        if (!(Y.z > cal.z_eps) || !(Xi.z > cal.z_eps)) return false;
        const float zj = 1.0f / Y.z, zi = 1.0f / Xi.z;
        const float pju = cal.fx * Y.x * zj + cal.cx, pjv = cal.fy * Y.y * zj + cal.cy;
        const float piu = cal.fx * Xi.x * zi + cal.cx, piv = cal.fy * Xi.y * zi + cal.cy;
        if (!(pju >= cal.border && pju < cal.width - cal.border && pjv >= cal.border && pjv < cal.height - cal.border))
            return false;
        err[0] = (pju - piu) * cal.inv_sigma_pixel;
        err[1] = (pjv - piv) * cal.inv_sigma_pixel;
        err[2] = (logf(Y.z) - logf(Xi.z)) * cal.inv_sigma_depth;
        sqrt_w = sqrtf(qc);
        d0x = cal.fx * zj * cal.inv_sigma_pixel; d0z = -cal.fx * Y.x * zj * zj * cal.inv_sigma_pixel;
        d1y = cal.fy * zj * cal.inv_sigma_pixel; d1z = -cal.fy * Y.y * zj * zj * cal.inv_sigma_pixel;
        d2z = zj * cal.inv_sigma_depth;
    }
    const float w2 = sqrt_w * sqrt_w;
    float w[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float we = fabsf(sqrt_w * err[c]);
        w[c] = ((we < 1.345f) ? 1.0f : (MODE == 2 ? 1.345f / we : 1.345f * __builtin_amdgcn_rcpf(we))) * w2;
    }
    // D rows: (d0x, 0, d0z), (0, d1y, d1z), (0, 0, d2z)
    const float Axx = w[0] * d0x * d0x, Axz = w[0] * d0x * d0z, Ayy = w[1] * d1y * d1y, Ayz = w[1] * d1y * d1z;
    const float Azz = w[0] * d0z * d0z + w[1] * d1z * d1z + w[2] * d2z * d2z;
    const float e0 = w[0] * err[0], e1 = w[1] * err[1], e2 = w[2] * err[2];
    const V3<float> b{e0 * d0x, e1 * d1y, e0 * d0z + e1 * d1z + e2 * d2z};
    accum_congruence(Y, Axx, 0.f, Axz, Ayy, Ayz, Azz, b, s_inv, 1.0f, h);
    h[35] += 1.0f;
    return true;
}

// 4 consecutive points of the edge per lane and trip: validity (4 bytes), Q, idx, C_j and X_j come in as dwordx4
// loads with the next trip's loads issued before this trip's arithmetic; only X_i / C_i at the match index are
// gathers.  fp32 sums over the <= 4 points, one float64 fold per trip (see k_track_accum).
template <int MODE>
__global__ void __launch_bounds__(kThreads)
k_gn_blocks(const float *__restrict__ Twc, const float *__restrict__ Xs, const float *__restrict__ Cs,
            const int32_t *__restrict__ ii, const int32_t *__restrict__ jj, const int32_t *__restrict__ idx,
            const uint8_t *__restrict__ valid, const float *__restrict__ Q, double *__restrict__ part,
            const double *__restrict__ done, int K, int P, int chunks, float inv_sigma, float C_thresh,
            float Q_thresh, const CalibParams cal) {
    if (done && done[0] != 0.0) return;
    const int e = blockIdx.y, chunk = blockIdx.x;
    const int ix = ii[e], jx = jj[e];
    double acc[kSums];
#pragma unroll
    for (int i = 0; i < kSums; ++i) acc[i] = 0.0;
    if (ix >= 0 && ix < K && jx >= 0 && jx < K) {
        const Pose<double> Ti = load_pose<double>(Twc + 8 * ix), Tj = load_pose<double>(Twc + 8 * jx);
        const Pose<double> Tij_d = rel_ops(Ti, Tj);
        Pose<float> Tij;
        Tij.t = {(float)Tij_d.t.x, (float)Tij_d.t.y, (float)Tij_d.t.z};
        Tij.q = {(float)Tij_d.q.x, (float)Tij_d.q.y, (float)Tij_d.q.z, (float)Tij_d.q.w};
        Tij.s = (float)Tij_d.s;
        const float s_inv = (float)(1.0 / Ti.s);
        const float *Xi_base = Xs + (size_t)ix * P * 3, *Xj_base = Xs + (size_t)jx * P * 3;
        const float *Ci = Cs + (size_t)ix * P, *Cj = Cs + (size_t)jx * P;
        const size_t eo = (size_t)e * P;
        auto one = [&](int k, bool vld, float qc, int id, float cj, const V3<float> &Xj, float *h) {
            if (!vld) return;
            if (id < 0) id += P;
            id = id < 0 ? 0 : (id >= P ? P - 1 : id);
            if (!(qc > Q_thresh) || !(Ci[id] > C_thresh) || !(cj > C_thresh)) return;
            gn_point<MODE>(Tij, s_inv, V3<float>{Xi_base[3 * id], Xi_base[3 * id + 1], Xi_base[3 * id + 2]}, Xj, qc,
                           inv_sigma, cal, h);
        };
        const bool vec = (P % 4 == 0) && ((reinterpret_cast<size_t>(Xs) | reinterpret_cast<size_t>(Cs) | reinterpret_cast<size_t>(Q) |
                                           reinterpret_cast<size_t>(idx)) % 16 == 0) && (reinterpret_cast<size_t>(valid) % 4 == 0);
        if (vec) {
            const int groups = P / 4, stride = chunks * kThreads;
            int gi = chunk * kThreads + threadIdx.x;
            float4 xj[3], q4, c4;
            int4 i4;
            unsigned v = 0;
            auto load = [&](int g) {
                const float4 *px = reinterpret_cast<const float4 *>(Xj_base) + 3 * (size_t)g;
                xj[0] = px[0]; xj[1] = px[1]; xj[2] = px[2];
                q4 = reinterpret_cast<const float4 *>(Q + eo)[g];
                c4 = reinterpret_cast<const float4 *>(Cj)[g];
                i4 = reinterpret_cast<const int4 *>(idx + eo)[g];
                v = reinterpret_cast<const unsigned *>(valid + eo)[g];
            };
            if (gi < groups) load(gi);
            while (gi < groups) {
                const float4 x0 = xj[0], x1 = xj[1], x2 = xj[2], qq = q4, cc = c4;
                const int4 id4 = i4;
                const unsigned vv = v;
                const int nx = gi + stride;
                if (nx < groups) load(nx);
                float h[kSums];
#pragma unroll
                for (int i = 0; i < kSums; ++i) h[i] = 0.f;
                one(4 * gi + 0, vv & 0x000000ffu, qq.x, id4.x, cc.x, V3<float>{x0.x, x0.y, x0.z}, h);
                one(4 * gi + 1, vv & 0x0000ff00u, qq.y, id4.y, cc.y, V3<float>{x0.w, x1.x, x1.y}, h);
                one(4 * gi + 2, vv & 0x00ff0000u, qq.z, id4.z, cc.z, V3<float>{x1.z, x1.w, x2.x}, h);
                one(4 * gi + 3, vv & 0xff000000u, qq.w, id4.w, cc.w, V3<float>{x2.y, x2.z, x2.w}, h);
#pragma unroll
                for (int i = 0; i < kSums; ++i) acc[i] += (double)h[i];
                gi = nx;
            }
        } else {
            for (int k = chunk * kThreads + threadIdx.x; k < P; k += chunks * kThreads) {
                float h[kSums];
#pragma unroll
                for (int i = 0; i < kSums; ++i) h[i] = 0.f;
                one(k, valid[eo + k] != 0, Q[eo + k], idx[eo + k], Cj[k], V3<float>{Xj_base[3 * k], Xj_base[3 * k + 1], Xj_base[3 * k + 2]}, h);
#pragma unroll
                for (int i = 0; i < kSums; ++i) acc[i] += (double)h[i];
            }
        }
    }
    m3_block_reduce36(acc, part + ((size_t)e * chunks + chunk) * kSums);
}

// Fixed-order reduction over the chunks of one edge, then the per-edge constant frame change
// H = G^T H' G, g = G^T g' with G = blockdiag(R_i, R_i, 1) (see gn_point), in float64.
__global__ void __launch_bounds__(64)
k_gn_reduce(const double *__restrict__ part, double *__restrict__ blocks, const double *__restrict__ done,
            const float *__restrict__ Twc, const int32_t *__restrict__ ii, int K, int chunks) {
    if (done && done[0] != 0.0) return;
    const int e = blockIdx.x, t = threadIdx.x;
    __shared__ double Hp[7][7], gp[7], G[7][7];
    double s = 0.0;
    if (t < kSums)
        for (int c = 0; c < chunks; ++c) s += part[((size_t)e * chunks + c) * kSums + t];
    if (t < 28) {
        int r = 0, k = t;
        while (k >= 7 - r) { k -= 7 - r; ++r; }
        Hp[r][r + k] = s; Hp[r + k][r] = s;
    } else if (t < 35) gp[t - 28] = s;
    else if (t == 35) blocks[(size_t)e * kSums + 35] = s;
    if (t < 49) G[t / 7][t % 7] = (t / 7 == t % 7 && t == 48) ? 1.0 : 0.0;
    __syncthreads();
    const int ix = ii[e];
    if (t == 0 && ix >= 0 && ix < K) {
        const Pose<double> Ti = load_pose<double>(Twc + 8 * ix);
        const V3<double> r0 = qrot(Ti.q, V3<double>{1.0, 0.0, 0.0}), r1 = qrot(Ti.q, V3<double>{0.0, 1.0, 0.0}),
                         r2 = qrot(Ti.q, V3<double>{0.0, 0.0, 1.0});          // columns of R_i
        const double R[3][3] = {{r0.x, r1.x, r2.x}, {r0.y, r1.y, r2.y}, {r0.z, r1.z, r2.z}};
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) { G[a][b] = R[a][b]; G[3 + a][3 + b] = R[a][b]; }
    }
    __syncthreads();
    if (t < 49) {
        const int r = t / 7, c = t % 7;
        if (r <= c) {
            double h = 0.0;
            for (int a = 0; a < 7; ++a) {
                double u = 0.0;
                for (int b = 0; b < 7; ++b) u += Hp[a][b] * G[b][c];
                h += G[a][r] * u;
            }
            blocks[(size_t)e * kSums + r * 7 - (r * (r - 1)) / 2 + (c - r)] = h;
        }
    } else if (t < 56) {
        const int r = t - 49;
        double v = 0.0;
        for (int a = 0; a < 7; ++a) v += G[a][r] * gp[a];
        blocks[(size_t)e * kSums + 28 + r] = v;
    }
}

// Dense system from the per-edge blocks (gauss_newton.py:220-251): an edge (i, j) adds +Hjj to the diagonal
// blocks (i,i), (j,j), -Hjj to (i,j), (j,i), -gj to g_i and +gj to g_j.  GATHER form: workgroup (a, b) owns the
// 7x7 block (a, b) of H (and g_a when b == a) and walks the edges in index order, so every entry is summed in
// a FIXED order - bitwise reproducible, unlike an atomicAdd scatter whose order changes from run to run (on
// degenerate geometry that difference alone decided between a finite and an overflowing step).
// Round 3: one workgroup (one wave) per block ROW a instead of one per 7 x 7 block.  The per-block grid was
// num_free^2 workgroups that each walked all E edges - O(F^2 E): 65 025 x 1 524 edge visits for BASELINE configs[4]
// (most of the 6.7 ms the 1 785-unknown step took in the first backend bench of this round).  Here 64 lanes test 64 edges
// at a time (coalesced loads, one ballot), the few edges that touch keyframe a are then taken in ascending edge order:
// the diagonal block and g_a accumulate in registers, an off-diagonal block (a, other) is a read-modify-write of the
// row this workgroup alone owns.  Every entry is still summed in edge order - the same bits as before.  H must be
// zero on entry (the launcher clears it on the stream).
__global__ void __launch_bounds__(64)
k_gn_assemble(const double *__restrict__ blocks, const int32_t *__restrict__ ii, const int32_t *__restrict__ jj,
              const int32_t *__restrict__ local, double *__restrict__ H, double *__restrict__ g,
              const double *__restrict__ done, int K, int dim, int E) {
    if (done && done[0] != 0.0) return;
    const int a = blockIdx.x, t = threadIdx.x;
    const int r = t / 7, c = t % 7;
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    const int hidx = lo * 7 - lo * (lo - 1) / 2 + (hi - lo);
    double h = 0.0, gv = 0.0;
    for (int e0 = 0; e0 < E; e0 += 64) {
        const int e = e0 + t;
        int il = -2, jl = -2;
        bool touch = false;
        if (e < E) {
            const int ix = ii[e], jx = jj[e];
            if (ix >= 0 && ix < K && jx >= 0 && jx < K && blocks[(size_t)e * kSums + 35] != 0.0) {
                il = local[ix]; jl = local[jx];
                touch = (il == a) || (jl == a);
            }
        }
        unsigned long long m = __ballot(touch);
        while (m) {                                          // ascending edge index: the summation order of every entry
            const int bit = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int eil = __shfl(il, bit, 64), ejl = __shfl(jl, bit, 64);
            const double *blk = blocks + (size_t)(e0 + bit) * kSums;
            const bool diag_i = eil == a, diag_j = ejl == a, both = eil >= 0 && ejl >= 0;
            if (t < 49) {
                const double v = blk[hidx];
                if (diag_i) h += v;                          // (i,i)
                if (diag_j) h += v;                          // (j,j)
                if (both && diag_i && ejl != a) H[(size_t)(a * 7 + r) * dim + ejl * 7 + c] -= v;   // (i,j)
                if (both && diag_j && eil != a) H[(size_t)(a * 7 + r) * dim + eil * 7 + c] -= v;   // (j,i)
                if (both && diag_i && diag_j) h -= 2.0 * v;  // a self edge (never built by the factor graph): (i,j) + (j,i) on the diagonal
            } else if (t < 56) {
                if (diag_i) gv -= blk[28 + (t - 49)];
                if (diag_j) gv += blk[28 + (t - 49)];
            }
        }
    }
    if (t < 49) H[(size_t)(a * 7 + r) * dim + a * 7 + c] = h;
    else if (t < 56) g[a * 7 + (t - 49)] = gv;
}

// After the solve (dx in x[0..dim)): |dx|, largest scale step, stop test, retraction T <- exp(dx) T of the free
// keyframes.  One workgroup; shared by the single-workgroup solver and the blocked one (gn_chol.hip).
__device__ void gn_step_tail(double *__restrict__ x, float *__restrict__ Twc, const int32_t *__restrict__ local,
                             double *__restrict__ info, int K, int dim, float delta_thresh, int apply) {
    const int t = threadIdx.x;
    // |dx|^2 and the largest log-scale component: per-thread partials over whole 7-vectors, then a fixed-order tree
    // (one thread walking all of x took 77-157 us at 832-1785 unknowns: dependent loads and an integer modulo per element)
    __shared__ double ps[kSolveThreads], pm[kSolveThreads];
    {
        double s = 0.0, m = 0.0;
        for (int v = t; v < dim / 7; v += kSolveThreads) {
            const double *xv = x + 7 * v;
#pragma unroll
            for (int c = 0; c < 7; ++c) s += xv[c] * xv[c];
            m = fmax(m, fabs(xv[6]));
        }
        ps[t] = s; pm[t] = m;
    }
    __syncthreads();
    for (int w = kSolveThreads / 2; w > 0; w >>= 1) {
        if (t < w) { ps[t] += ps[t + w]; pm[t] = fmax(pm[t], pm[t + w]); }
        __syncthreads();
    }
    const double nrm2 = ps[0], smax = pm[0];
    const double dn = sqrt(nrm2);
    if (t == 0) info[1] = dn;
    if (!isfinite(dn) || smax > 30.0) {                     // a scale step e^sigma beyond float range (degenerate
        if (t == 0) { info[2] = 1.0; info[3] = 1.0; }       // geometry): report failure, keep the poses
        return;
    }
    if (dn < (double)delta_thresh) {                        // stop BEFORE the update (gauss_newton.py:262-265)
        if (t == 0) info[2] = 1.0;
        return;
    }
    if (!apply) return;
    for (int kf = t; kf < K; kf += (int)blockDim.x) {
        const int l = local[kf];
        if (l < 0) continue;
        store_pose(Twc + 8 * kf, retract_ops(x + 7 * l, load_pose<double>(Twc + 8 * kf)));
    }
    if (t == 0) info[0] += 1.0;
}


// One workgroup: (H + 1e-6 I) dx = -g by in-place Cholesky (lower) + two triangular solves,
// stop test, then T <- exp(dx) T for the free keyframes.  dx is left in x[0..dim).
__global__ void __launch_bounds__(kSolveThreads)
k_gn_step(double *__restrict__ H, double *__restrict__ g, double *__restrict__ x, float *__restrict__ Twc,
          const int32_t *__restrict__ local, double *__restrict__ info, int K, int dim, float delta_thresh,
          int apply) {
    if (info[2] != 0.0) return;
    const int t = threadIdx.x;
    __shared__ double piv;
    __shared__ int fail;
    if (t == 0) fail = 0;
    for (int i = t; i < dim; i += kSolveThreads) { H[(size_t)i * dim + i] += 1e-6; x[i] = -g[i]; }
    __syncthreads();
    for (int k = 0; k < dim; ++k) {
        if (t == 0) {
            const double d = H[(size_t)k * dim + k];
            if (!(d > 0.0) || !isfinite(d)) fail = 1;
            piv = sqrt(d);
            H[(size_t)k * dim + k] = piv;
        }
        __syncthreads();
        if (fail) break;
        const double inv = 1.0 / piv;
        for (int i = k + 1 + t; i < dim; i += kSolveThreads) H[(size_t)i * dim + k] *= inv;
        __syncthreads();
        // trailing update of the lower triangle, rows i > k, cols k < j <= i
        const int m = dim - k - 1;
        for (int idx = t; idx < m * m; idx += kSolveThreads) {
            const int i = k + 1 + idx / m, j = k + 1 + idx % m;
            if (j <= i) H[(size_t)i * dim + j] -= H[(size_t)i * dim + k] * H[(size_t)j * dim + k];
        }
        __syncthreads();
    }
    if (fail) {
        if (t == 0) { info[2] = 1.0; info[3] = 1.0; }
        return;
    }
    for (int k = 0; k < dim; ++k) {                         // L y = b
        if (t == 0) x[k] = x[k] / H[(size_t)k * dim + k];
        __syncthreads();
        const double yk = x[k];
        for (int i = k + 1 + t; i < dim; i += kSolveThreads) x[i] -= H[(size_t)i * dim + k] * yk;
        __syncthreads();
    }
    for (int k = dim - 1; k >= 0; --k) {                    // L^T dx = y
        if (t == 0) x[k] = x[k] / H[(size_t)k * dim + k];
        __syncthreads();
        const double xk = x[k];
        for (int i = t; i < k; i += kSolveThreads) x[i] -= H[(size_t)k * dim + i] * xk;
        __syncthreads();
    }
    gn_step_tail(x, Twc, local, info, K, dim, delta_thresh, apply);
}

// tail of a blocked solve: fail flag of the factorisation -> info, else the common tail
__global__ void __launch_bounds__(kSolveThreads)
k_gn_tail(double *__restrict__ x, float *__restrict__ Twc, const int32_t *__restrict__ local, double *__restrict__ info,
          const double *__restrict__ fail, int K, int dim, float delta_thresh, int apply) {
    if (info[2] != 0.0) return;
    if (fail[0] != 0.0) {
        if (threadIdx.x == 0) { info[2] = 1.0; info[3] = 1.0; }
        return;
    }
    gn_step_tail(x, Twc, local, info, K, dim, delta_thresh, apply);
}

__global__ void k_gn_neg_rhs(const double *__restrict__ g, double *__restrict__ x, double *__restrict__ fail, int dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < dim) x[i] = -g[i];
    if (i == 0) fail[0] = 0.0;
}

__global__ void __launch_bounds__(kThreads)
k_gn_retract(float *__restrict__ Twc, const double *__restrict__ dx, const int32_t *__restrict__ local, int K) {
    const int kf = blockIdx.x * kThreads + threadIdx.x;
    if (kf >= K) return;
    const int l = local[kf];
    if (l < 0) return;
    store_pose(Twc + 8 * kf, retract_ops(dx + 7 * l, load_pose<double>(Twc + 8 * kf)));
}

__global__ void k_gn_info_init(double *info) {
    if (threadIdx.x < 4) info[threadIdx.x] = 0.0;
}

// host: calib = 10 floats (fx, fy, cx, cy, width, height, border, z_eps, sigma_pixel, sigma_depth) or null
CalibParams make_calib(const float *c) {
    CalibParams p{};
    if (c) {
        p.fx = c[0]; p.fy = c[1]; p.cx = c[2]; p.cy = c[3]; p.width = c[4]; p.height = c[5]; p.border = c[6];
        p.z_eps = c[7]; p.inv_sigma_pixel = 1.0f / c[8]; p.inv_sigma_depth = 1.0f / c[9];
    }
    return p;
}


// One Gauss-Newton step from an assembled system, any size, stream-ordered: dim <= kMaxDim -> the single-workgroup
// kernel; larger -> blocked Cholesky (gn_chol.hip) + k_gn_tail.  Hbuf = H [dim*dim] | g | x | b | y [dim each] | fail [8]
// | Ld [m3_chol_diag_doubles(dim)] = m3_gn_rays_hbuf_doubles(dim).
int launch_step(double *Hbuf, float *Twc, const int32_t *local, double *info, int K, int dim, float delta_thresh,
                int apply, hipStream_t st) {
    double *H = Hbuf, *g = Hbuf + (size_t)dim * dim, *x = g + dim, *b = x + dim, *y = b + dim, *fail = y + dim;
    if (dim <= kMaxDim) {
        hipLaunchKernelGGL(k_gn_step, dim3(1), dim3(kSolveThreads), 0, st, H, g, x, Twc, local, info, K, dim, delta_thresh, apply);
        M3_CHECK_LAUNCH("m3_gn_rays/step");
        return M3_OK;
    }
    hipLaunchKernelGGL(k_gn_neg_rhs, dim3(m3_cdiv(dim, 256)), dim3(256), 0, st, (const double *)g, b, fail, dim);
    const int rc = m3_chol_solve_launch(H, b, y, x, fail + 8, fail, info + 2, dim, 1e-6, st);
    if (rc != M3_OK) return rc;
    hipLaunchKernelGGL(k_gn_tail, dim3(1), dim3(kSolveThreads), 0, st, x, Twc, local, info, (const double *)fail, K, dim,
                       delta_thresh, apply);
    M3_CHECK_LAUNCH("m3_gn_rays/tail");
    return M3_OK;
}

int launch_blocks(const float *Twc, const float *Xs, const float *Cs, const int32_t *ii, const int32_t *jj,
                  const int32_t *idx, const uint8_t *valid, const float *Q, double *blocks, double *ws,
                  const double *done, int K, int P, int E, float sigma_ray, float C_thresh, float Q_thresh,
                  int point_mode, const CalibParams &cal, hipStream_t st) {
    // chunks per edge: enough workgroups to fill the chip (~2048 over all edges) but no more - every chunk
    // ends in a 36-value float64 block reduction.  Never more than m3_gn_rays_chunks(P), the count the
    // caller sized the workspace for.  (config 5, 180 edges x 262144 points: 1.69 -> 1.47 ms)
    int chunks = m3_cdiv(2048, E);
    if (chunks > gn_chunks(P)) chunks = gn_chunks(P);
    if (chunks < 1) chunks = 1;
    const float inv_sigma = (float)(1.0 / (double)sigma_ray);
    const dim3 grid(chunks, E), blk(kThreads);
    if (point_mode == 0)
        hipLaunchKernelGGL(k_gn_blocks<0>, grid, blk, 0, st, Twc, Xs, Cs, ii, jj, idx, valid, Q, ws, done, K, P, chunks, inv_sigma, C_thresh, Q_thresh, cal);
    else if (point_mode == 1)
        hipLaunchKernelGGL(k_gn_blocks<1>, grid, blk, 0, st, Twc, Xs, Cs, ii, jj, idx, valid, Q, ws, done, K, P, chunks, inv_sigma, C_thresh, Q_thresh, cal);
    else
        hipLaunchKernelGGL(k_gn_blocks<2>, grid, blk, 0, st, Twc, Xs, Cs, ii, jj, idx, valid, Q, ws, done, K, P, chunks, inv_sigma, C_thresh, Q_thresh, cal);
    hipLaunchKernelGGL(k_gn_reduce, dim3(E), dim3(64), 0, st, (const double *)ws, blocks, done, Twc, ii, K, chunks);
    M3_CHECK_LAUNCH("m3_gn_rays_blocks");
    return M3_OK;
}

}  // namespace

extern "C" {

int m3_gn_rays_chunks(int P) { return gn_chunks(P); }
int m3_gn_rays_max_dim(void) { return kMaxDim; }
int64_t m3_gn_rays_hbuf_doubles(int dim) {
    return (int64_t)dim * dim + 4 * (int64_t)dim + 8 + (dim > kMaxDim ? m3_chol_diag_doubles(dim) : 0);
}

int m3_gn_rays_blocks(const float *Twc, const float *Xs, const float *Cs, const int32_t *ii, const int32_t *jj,
                      const int32_t *idx, const uint8_t *valid, const float *Q, double *blocks, double *ws,
                      int K, int P, int E, float sigma_ray, float C_thresh, float Q_thresh, int point_mode, const float *calib,
                      void *stream) {
    M3_REQUIRE(Twc && Xs && Cs && ii && jj && idx && valid && Q && blocks && ws);
    M3_REQUIRE(K > 0 && P > 0 && E > 0 && E <= 65535 && sigma_ray > 0.f && point_mode >= 0 && point_mode <= 2);
    M3_REQUIRE(point_mode != 2 || calib);
    return launch_blocks(Twc, Xs, Cs, ii, jj, idx, valid, Q, blocks, ws, nullptr, K, P, E, sigma_ray, C_thresh,
                         Q_thresh, point_mode, make_calib(calib), (hipStream_t)stream);
}

int m3_gn_rays_assemble(const double *blocks, const int32_t *ii, const int32_t *jj, const int32_t *local,
                        double *H, double *g, int K, int E, int num_free, void *stream) {
    M3_REQUIRE(blocks && ii && jj && local && H && g && K > 0 && E > 0 && num_free > 0);
    hipStream_t st = (hipStream_t)stream;
    const int dim = 7 * num_free;
    M3_CHECK_HIP(hipMemsetAsync(H, 0, (size_t)dim * dim * sizeof(double), st), "m3_gn_rays_assemble/memset");
    hipLaunchKernelGGL(k_gn_assemble, dim3(num_free), dim3(64), 0, st, blocks, ii, jj, local, H, g,
                       (const double *)nullptr, K, dim, E);
    M3_CHECK_LAUNCH("m3_gn_rays_assemble");
    return M3_OK;
}

int m3_gn_rays_retract(float *Twc, const double *dx, const int32_t *local, int K, void *stream) {
    M3_REQUIRE(Twc && dx && local && K > 0);
    hipLaunchKernelGGL(k_gn_retract, dim3(m3_cdiv(K, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, Twc, dx,
                       local, K);
    M3_CHECK_LAUNCH("m3_gn_rays_retract");
    return M3_OK;
}

int m3_gn_rays_solve(float *Twc, const float *Xs, const float *Cs, const int32_t *ii, const int32_t *jj,
                     const int32_t *idx, const uint8_t *valid, const float *Q, const int32_t *local,
                     double *blocks, double *ws, double *Hbuf, double *info, int K, int P, int E, int num_free,
                     float sigma_ray, float C_thresh, float Q_thresh, int max_iter, float delta_thresh,
                     int point_mode, const float *calib, void *stream) {
    M3_REQUIRE(Twc && Xs && Cs && ii && jj && idx && valid && Q && local && blocks && ws && Hbuf && info);
    M3_REQUIRE(K > 0 && P > 0 && E > 0 && E <= 65535 && num_free > 0 && max_iter >= 0 && sigma_ray > 0.f);
    M3_REQUIRE(point_mode >= 0 && point_mode <= 2 && (point_mode != 2 || calib));
    const CalibParams cal = make_calib(calib);
    const int dim = 7 * num_free;
    hipStream_t st = (hipStream_t)stream;
    double *H = Hbuf, *g = Hbuf + (size_t)dim * dim;          // Hbuf: m3_gn_rays_hbuf_doubles(dim)
    const double *done = info + 2;
    hipLaunchKernelGGL(k_gn_info_init, dim3(1), dim3(64), 0, st, info);
    for (int it = 0; it < max_iter; ++it) {
        int rc = launch_blocks(Twc, Xs, Cs, ii, jj, idx, valid, Q, blocks, ws, done, K, P, E, sigma_ray, C_thresh,
                               Q_thresh, point_mode, cal, st);
        if (rc != M3_OK) return rc;
        M3_CHECK_HIP(hipMemsetAsync(H, 0, (size_t)dim * dim * sizeof(double), st), "m3_gn_rays_solve/memset");
        hipLaunchKernelGGL(k_gn_assemble, dim3(num_free), dim3(64), 0, st, (const double *)blocks, ii, jj,
                           local, H, g, done, K, dim, E);
        M3_CHECK_LAUNCH("m3_gn_rays_solve/iter");
        rc = launch_step(Hbuf, Twc, local, info, K, dim, delta_thresh, 1, st);
        if (rc != M3_OK) return rc;
    }
    return M3_OK;
}

// One step from blocks that the caller assembled elsewhere (the edge-sharded solve: blocks all-gathered over the
// ranks): assemble -> factor -> solve -> stop test -> retract, all on the stream; info as in m3_gn_rays_solve
// (zero it once with m3_gn_rays_info_init before the first step).  Hbuf: m3_gn_rays_hbuf_doubles(7 * num_free).
int m3_gn_rays_step(float *Twc, const double *blocks, const int32_t *ii, const int32_t *jj, const int32_t *local,
                    double *Hbuf, double *info, int K, int E, int num_free, float delta_thresh, void *stream) {
    M3_REQUIRE(Twc && blocks && ii && jj && local && Hbuf && info && K > 0 && E > 0 && num_free > 0);
    hipStream_t st = (hipStream_t)stream;
    const int dim = 7 * num_free;
    double *H = Hbuf, *g = Hbuf + (size_t)dim * dim;
    M3_CHECK_HIP(hipMemsetAsync(H, 0, (size_t)dim * dim * sizeof(double), st), "m3_gn_rays_step/memset");
    hipLaunchKernelGGL(k_gn_assemble, dim3(num_free), dim3(64), 0, st, blocks, ii, jj, local, H, g,
                       (const double *)(info + 2), K, dim, E);
    M3_CHECK_LAUNCH("m3_gn_rays_step/assemble");
    return launch_step(Hbuf, Twc, local, info, K, dim, delta_thresh, 1, st);
}

int m3_gn_rays_info_init(double *info, void *stream) {
    M3_REQUIRE(info);
    hipLaunchKernelGGL(k_gn_info_init, dim3(1), dim3(64), 0, (hipStream_t)stream, info);
    M3_CHECK_LAUNCH("m3_gn_rays_info_init");
    return M3_OK;
}

}  // extern "C"
