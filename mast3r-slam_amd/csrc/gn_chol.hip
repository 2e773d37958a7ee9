// Blocked dense Cholesky solve in float64 for the backend Gauss-Newton systems that do not fit the single-workgroup
// kernel of gn_rays.hip (7F > 448: e.g. BASELINE configs[4], 256 keyframes -> 1785 unknowns).  Replaces the
// reference's host solve  dx = np.linalg.solve(H + 1e-6 I, -g)  (gauss_newton.py:253-260, linalg.py:17-50) - and the
// torch.linalg (hipSOLVER) call plus three host synchronisations per iteration that round 1 used for large graphs.
//
// Right-looking blocked factorisation, block size 64, entirely stream-ordered (no host round trip; a failed pivot
// sets a device flag that turns every later kernel of the solve into a no-op):
//   step k:  k_chol_panel   every row block i >= k:  factor the diagonal block (64 x 64, in LDS - recomputed by each
//                           workgroup instead of a separate launch + boundary), then L_ik = A_ik L_kk^-T.  The factor
//                           L_kk goes to a SEPARATE buffer (Ld [nblk][64][64]): A_kk in H is read by every workgroup of
//                           the launch, at times the dispatcher chooses (a workgroup that starts late - CUs busy with
//                           another stream's kernels - must still find A_kk, not L_kk: round-2 advisor finding), so
//                           nothing overwrites it; the substitution kernels read L_kk from Ld.
//            k_chol_update  every lower block pair i >= j > k:  A_ij -= L_ik L_jk^T
//   then forward substitution (one launch per block column: y_k = L_kk^-1 b_k, b_i -= L_ik y_k for i > k) and
//   backward substitution (x_k = L_kk^-T y_k, y_j -= L_kj^T x_k for j < k).
// 2 * 28 + 28 + 28 = 112 launches for 1785 unknowns.  H is row-major [dim, dim]; only the lower triangle is read.
#include "common.h"

namespace {

constexpr int NB = 64;
constexpr int kThreads = 256;

__device__ __forceinline__ bool solve_off(const double *flag) { return flag && flag[0] != 0.0; }

// Factor the nb x nb block held in LDS `d` (lower triangle, leading dimension NB) in place.  All 256 threads.
// Returns false (uniformly) on a non-positive / non-finite pivot.
__device__ bool chol_diag_lds(double (*d)[NB + 1], int nb, int *bad) {
    const int t = threadIdx.x;
    for (int k = 0; k < nb; ++k) {
        if (t == 0) {
            const double p = d[k][k];
            if (!(p > 0.0) || !isfinite(p)) *bad = 1;
            d[k][k] = sqrt(p > 0.0 ? p : 1.0);
        }
        __syncthreads();
        if (*bad) return false;
        const double inv = 1.0 / d[k][k];
        for (int i = k + 1 + t; i < nb; i += kThreads) d[i][k] *= inv;
        __syncthreads();
        const int m = nb - k - 1;
        for (int idx = t; idx < m * m; idx += kThreads) {
            const int i = k + 1 + idx / m, j = k + 1 + idx % m;
            if (j <= i) d[i][j] -= d[i][k] * d[j][k];
        }
        __syncthreads();
    }
    return true;
}

// grid.x = row block i - k (0 = the diagonal block itself); adds `shift` to the diagonal of block k first
__global__ void __launch_bounds__(kThreads)
k_chol_panel(double *__restrict__ H, double *__restrict__ Ld, int dim, int k, double shift, double *__restrict__ fail,
             const double *__restrict__ off) {
    if (solve_off(off) || fail[0] != 0.0) return;
    __shared__ double d[NB][NB + 1];
    __shared__ double a[NB][NB + 1];
    __shared__ int bad;
    const int t = threadIdx.x, k0 = k * NB, nb = min(NB, dim - k0);
    if (t == 0) bad = 0;
    for (int idx = t; idx < nb * nb; idx += kThreads) {
        const int r = idx / nb, c = idx % nb;
        d[r][c] = (c <= r) ? H[(size_t)(k0 + r) * dim + k0 + c] + (r == c ? shift : 0.0) : 0.0;
    }
    __syncthreads();
    if (!chol_diag_lds(d, nb, &bad)) {
        if (t == 0 && blockIdx.x == 0) fail[0] = 1.0;
        return;
    }
    const int ib = k + blockIdx.x;
    if (blockIdx.x == 0) {                                   // L_kk -> Ld[k] (never over A_kk: the other workgroups read it)
        double *L = Ld + (size_t)k * NB * NB;
        for (int idx = t; idx < NB * NB; idx += kThreads) {
            const int r = idx / NB, c = idx % NB;
            L[idx] = (r < nb && c <= r) ? d[r][c] : 0.0;
        }
        return;
    }
    // L_ik = A_ik L_kk^-T : row r of the block solves  x L_kk^T = a_r  by forward substitution over the columns
    const int i0 = ib * NB, mb = min(NB, dim - i0);
    for (int idx = t; idx < mb * nb; idx += kThreads) a[idx / nb][idx % nb] = H[(size_t)(i0 + idx / nb) * dim + k0 + idx % nb];
    __syncthreads();
    // 4 threads per row would need a reduction per column; one thread per row is 64 x 64 / 2 FMAs on 64 lanes
    if (t < mb) {
        for (int c = 0; c < nb; ++c) {
            double s = a[t][c];
            for (int j = 0; j < c; ++j) s -= a[t][j] * d[c][j];
            a[t][c] = s / d[c][c];
        }
    }
    __syncthreads();
    for (int idx = t; idx < mb * nb; idx += kThreads) H[(size_t)(i0 + idx / nb) * dim + k0 + idx % nb] = a[idx / nb][idx % nb];
}

// grid = (bj, bi) offsets over the trailing lower triangle: block (i, j) with i >= j > k:  A_ij -= L_ik L_jk^T
__global__ void __launch_bounds__(kThreads)
k_chol_update(double *__restrict__ H, int dim, int k, const double *__restrict__ fail, const double *__restrict__ off) {
    if (solve_off(off) || fail[0] != 0.0) return;
    const int ib = k + 1 + blockIdx.y, jb = k + 1 + blockIdx.x;
    if (jb > ib) return;
    __shared__ double li[NB][NB + 1], lj[NB][NB + 1];
    const int t = threadIdx.x, k0 = k * NB, i0 = ib * NB, j0 = jb * NB;
    const int kb = min(NB, dim - k0), mi = min(NB, dim - i0), mj = min(NB, dim - j0);
    for (int idx = t; idx < NB * NB; idx += kThreads) {
        const int r = idx / NB, c = idx % NB;
        li[r][c] = (r < mi && c < kb) ? H[(size_t)(i0 + r) * dim + k0 + c] : 0.0;
        lj[r][c] = (r < mj && c < kb) ? H[(size_t)(j0 + r) * dim + k0 + c] : 0.0;
    }
    __syncthreads();
    const int tr = (t / 16) * 4, tc = (t % 16) * 4;          // 4 x 4 outputs per thread
    double acc[4][4] = {};
    for (int c = 0; c < kb; ++c) {
        double x[4], y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = li[tr + e][c]; y[e] = lj[tc + e][c]; }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[e][f] += x[e] * y[f];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int r = tr + e, c = tc + f;
            if (r < mi && c < mj && (ib > jb || c <= r)) H[(size_t)(i0 + r) * dim + j0 + c] -= acc[e][f];
        }
}

// forward: grid.x = row block i - k; every workgroup solves y_k = L_kk^-1 b_k from an LDS copy of L_kk (64 sequential
// steps), block 0 stores it, block i > 0 then does b_i -= L_ik y_k.
__global__ void __launch_bounds__(kThreads)
k_chol_fwd(const double *__restrict__ H, const double *__restrict__ Ld, double *__restrict__ b, double *__restrict__ yout,
           int dim, int k, const double *__restrict__ fail, const double *__restrict__ off) {
    if (solve_off(off) || fail[0] != 0.0) return;
    __shared__ double d[NB][NB + 1];
    __shared__ double y[NB];
    const int t = threadIdx.x, k0 = k * NB, nb = min(NB, dim - k0);
    for (int idx = t; idx < NB * NB; idx += kThreads) d[idx / NB][idx % NB] = Ld[(size_t)k * NB * NB + idx];
    if (t < nb) y[t] = b[k0 + t];                             // b_k is final (earlier launches); nobody writes it here:
    __syncthreads();                                          // the solved block goes to a SEPARATE vector, the other
    for (int c = 0; c < nb; ++c) {                            // workgroups of this launch are reading b_k right now
        if (t == c) y[c] = y[c] / d[c][c];
        __syncthreads();
        if (t > c && t < nb) y[t] -= d[t][c] * y[c];
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        if (t < nb) yout[k0 + t] = y[t];
        return;
    }
    const int i0 = (k + blockIdx.x) * NB, mb = min(NB, dim - i0);
    const int r = t / 4, q = t % 4;                          // 4 threads per row
    double s = 0.0;
    if (r < mb)
        for (int c = q; c < nb; c += 4) s += H[(size_t)(i0 + r) * dim + k0 + c] * y[c];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (r < mb && q == 0) b[i0 + r] -= s;
}

// backward: grid.x = k - j (0 = the diagonal block): x_k = L_kk^-T y_k, then y_j -= L_kj^T x_k for the blocks j < k
__global__ void __launch_bounds__(kThreads)
k_chol_bwd(const double *__restrict__ H, const double *__restrict__ Ld, double *__restrict__ yv, double *__restrict__ xout,
           int dim, int k, const double *__restrict__ fail, const double *__restrict__ off) {
    if (solve_off(off) || fail[0] != 0.0) return;
    __shared__ double d[NB][NB + 1];
    __shared__ double y[NB];
    __shared__ double red[kThreads / NB][NB];
    const int t = threadIdx.x, k0 = k * NB, nb = min(NB, dim - k0);
    for (int idx = t; idx < NB * NB; idx += kThreads) d[idx / NB][idx % NB] = Ld[(size_t)k * NB * NB + idx];
    if (t < nb) y[t] = yv[k0 + t];
    __syncthreads();
    for (int c = nb - 1; c >= 0; --c) {
        if (t == c) y[c] = y[c] / d[c][c];
        __syncthreads();
        if (t < c) y[t] -= d[c][t] * y[c];                    // L^T: element (t, c) of L^T = L[c][t]
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        if (t < nb) xout[k0 + t] = y[t];
        return;
    }
    const int j0 = (k - blockIdx.x) * NB;                    // a full block (j < k)
    // y_j[c] -= sum_r L_kj[r][c] x_k[r]: thread (c, part) sums a quarter of the rows, coalesced over c
    const int c = t % NB, part = t / NB;
    double s = 0.0;
    for (int r = part; r < nb; r += kThreads / NB) s += H[(size_t)(k0 + r) * dim + j0 + c] * y[r];
    red[part][c] = s;
    __syncthreads();
    if (part == 0) yv[j0 + c] -= (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

}  // namespace

// (H + shift I) x = b: H [dim, dim] row-major float64 (lower triangle read; the off-diagonal blocks are overwritten by
// L, the diagonal blocks keep A_kk of the running factorisation), b [dim] (destroyed), y [dim] scratch, Ld
// [m3_chol_diag_doubles(dim)] scratch for the diagonal factors, x [dim] out.  fail[0] is set to 1 on a non-positive
// pivot (then x is garbage); `off` (may be null): when off[0] != 0 the whole sequence is a no-op (the solver's
// device-side stop flag).
int64_t m3_chol_diag_doubles(int dim) { return (int64_t)((dim + NB - 1) / NB) * NB * NB; }

int m3_chol_solve_launch(double *H, double *b, double *y, double *x, double *Ld, double *fail, const double *off, int dim,
                         double shift, hipStream_t st) {
    const int nblk = (dim + NB - 1) / NB;
    for (int k = 0; k < nblk; ++k) {
        hipLaunchKernelGGL(k_chol_panel, dim3(nblk - k), dim3(kThreads), 0, st, H, Ld, dim, k, shift, fail, off);
        if (k + 1 < nblk)
            hipLaunchKernelGGL(k_chol_update, dim3(nblk - k - 1, nblk - k - 1), dim3(kThreads), 0, st, H, dim, k,
                               (const double *)fail, off);
    }
    for (int k = 0; k < nblk; ++k)
        hipLaunchKernelGGL(k_chol_fwd, dim3(nblk - k), dim3(kThreads), 0, st, (const double *)H, (const double *)Ld, b, y,
                           dim, k, (const double *)fail, off);
    for (int k = nblk - 1; k >= 0; --k)
        hipLaunchKernelGGL(k_chol_bwd, dim3(k + 1), dim3(kThreads), 0, st, (const double *)H, (const double *)Ld, y, x, dim,
                           k, (const double *)fail, off);
    M3_CHECK_LAUNCH("m3_chol_solve");
    return M3_OK;
}

extern "C" {

// Level-1 entry (linalg.cholesky_solve, linalg.py:17-50, for systems of any size): solves (H + shift I) x = b.
// H [dim,dim] float64 row-major (destroyed), b [dim] (destroyed), x [dim] out, ws: m3_chol_ws_doubles(dim) doubles
// (ws[0] = status on return: 0 ok, 1 not positive definite).
int64_t m3_chol_ws_doubles(int dim) { return dim > 0 ? 1 + (int64_t)dim + m3_chol_diag_doubles(dim) : 0; }

int m3_chol_solve(double *H, double *b, double *x, double *ws, int dim, double shift, void *stream) {
    M3_REQUIRE(H && b && x && ws && dim > 0 && x != b);
    hipStream_t st = (hipStream_t)stream;
    M3_CHECK_HIP(hipMemsetAsync(ws, 0, sizeof(double), st), "m3_chol_solve/memset");
    return m3_chol_solve_launch(H, b, ws + 1, x, ws + 1 + dim, ws, nullptr, dim, shift, st);
}

}  // extern "C"
