// Blocked dense Cholesky solve in float64 for the backend Gauss-Newton systems (BASELINE configs[4]: 256 keyframes ->
// 1785 unknowns).  Replaces the reference's host solve  dx = np.linalg.solve(H + 1e-6 I, -g)  (gauss_newton.py:253-260,
// linalg.py:17-50) - and the torch.linalg (hipSOLVER) call plus three host synchronisations per iteration that round 1 used.
//
// Right-looking blocked factorisation, block size 64, entirely stream-ordered (no host round trip; a failed pivot sets a
// device flag that turns every later kernel of the solve into a no-op):
//   step k:  k_chol_diag    ONE workgroup: factor the diagonal block A_kk (+ shift) in LDS and invert the factor; L_kk and
//                           L_kk^-1 go to SEPARATE buffers (Ld, Li [nblk][64][64]) - A_kk in H is never overwritten, so no
//                           kernel ever reads a diagonal block another workgroup is rewriting (round-2 advisor finding)
//            k_chol_trsm    every row block i > k:  L_ik = A_ik L_kk^-T  as a 64 x 64 x 64 product against the inverse
//            k_chol_update  every lower block pair i >= j > k:  A_ij -= L_ik L_jk^T
//   then forward substitution (one launch per block column: y_k = L_kk^-1 b_k as a matrix-vector product, b_i -= L_ik y_k
//   for i > k) and backward substitution (x_k = L_kk^-T y_k, y_j -= L_kj^T x_k for j < k).
// Round 3: the first form factored the diagonal block redundantly in EVERY workgroup of a panel launch, with an integer
// division per updated element, and solved the 64 triangular systems one row per thread: k_chol_panel 131 us per launch,
// 5.6 ms for 1785 unknowns.  Now the serial work exists once per block column, no index divisions (a thread owns fixed
// rows / columns), and every triangular solve is a product with the inverted 64 x 64 factor.
// 28 + 27 + 27 + 28 + 28 = 138 launches for 1785 unknowns.  H is row-major [dim, dim]; only the lower triangle is read.
#include "common.h"

namespace {

constexpr int NB = 64;
constexpr int kThreads = 256;

__device__ __forceinline__ bool solve_off(const double *flag) { return flag && flag[0] != 0.0; }

// One workgroup.  A_kk + shift I -> L (lower triangle), then li = L^-1.  The block lives in REGISTERS during the
// factorisation: thread (ti, tj) owns the 4 x 4 elements (ti + 16 a, tj + 16 b); per elimination step only the scaled
// pivot column travels through LDS (64 doubles), two barriers per step.  (A first version updated the block in LDS, 16
// dependent read-modify-writes per thread and step: 87 us per launch, 61 % of the 1785-unknown solve.)
// Rows / columns >= nb (last block) are the identity.
__global__ void __launch_bounds__(kThreads)
k_chol_diag(const double *__restrict__ H, double *__restrict__ Ld, double *__restrict__ Li, int dim, int k, double shift,
            double *__restrict__ fail, const double *__restrict__ off) {
    if (solve_off(off) || fail[0] != 0.0) return;
    __shared__ double d[NB][NB + 1];
    __shared__ double li[NB][NB + 1];
    __shared__ double colbuf[NB];
    __shared__ double pivs;
    __shared__ int bad;
    const int t = threadIdx.x, ti = t >> 4, tj = t & 15;
    const int k0 = k * NB, nb = min(NB, dim - k0);
    if (t == 0) bad = 0;
    double e[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int r = ti + 16 * a, c = tj + 16 * b;
            double v = (r == c) ? 1.0 : 0.0;
            if (r < nb && c <= r) v = H[(size_t)(k0 + r) * dim + k0 + c] + (r == c ? shift : 0.0);
            e[a][b] = v;
        }
    __syncthreads();
    for (int p = 0; p < nb; ++p) {
        // the owner of (p, p) publishes the pivot
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (ti + 16 * a == p && tj + 16 * b == p) {
                    const double piv = e[a][b];
                    if (!(piv > 0.0) || !isfinite(piv)) bad = 1;
                    const double sq = sqrt(piv > 0.0 ? piv : 1.0);
                    e[a][b] = sq;
                    pivs = sq;
                }
        __syncthreads();
        if (bad) break;
        const double inv = 1.0 / pivs;
        // owners of column p scale it and publish it (rows > p; row p publishes the pivot itself, rows < p zero)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            if (tj + 16 * b == p) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int r = ti + 16 * a;
                    if (r > p) e[a][b] *= inv;
                    colbuf[r] = r >= p ? e[a][b] : 0.0;
                }
            }
        __syncthreads();
        double lr[4], lc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) lr[a] = colbuf[ti + 16 * a];
#pragma unroll
        for (int b = 0; b < 4; ++b) lc[b] = colbuf[tj + 16 * b];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int r = ti + 16 * a, c = tj + 16 * b;
                if (c > p && c <= r) e[a][b] -= lr[a] * lc[b];
            }
        // (the next step's first barrier orders these reads of colbuf before its next writes)
    }
    if (bad) {
        if (t == 0) fail[0] = 1.0;
        return;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) d[ti + 16 * a][tj + 16 * b] = e[a][b];
    __syncthreads();
    // inverse of the lower-triangular factor, row by row: X[r][c] = -(sum_{j = c .. r-1} L[r][j] X[j][c]) / L[r][r]
    // (X[c][c] = 1 / L[c][c]); 4 threads per column split the sum.
    const int col = t >> 2, q = t & 3;
    for (int r = 0; r < NB; ++r) {
        double sum = 0.0;
        if (col < r)
            for (int j = col + q; j < r; j += 4) sum += d[r][j] * li[j][col];
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        if (q == 0) li[r][col] = (col == r) ? 1.0 / d[r][r] : (col < r ? -sum / d[r][r] : 0.0);
        __syncthreads();
    }
    double *L = Ld + (size_t)k * NB * NB, *I = Li + (size_t)k * NB * NB;
    for (int idx = t; idx < NB * NB; idx += kThreads) {
        const int r = idx >> 6, c = idx & 63;
        L[idx] = c <= r ? d[r][c] : 0.0;
        I[idx] = c <= r ? li[r][c] : 0.0;
    }
}

// grid.x = row block i - k - 1:  L_ik = A_ik L_kk^-T, i.e. out[r][c] = sum_{j <= c} A_ik[r][j] Linv[c][j]
__global__ void __launch_bounds__(kThreads)
k_chol_trsm(double *__restrict__ H, const double *__restrict__ Li, int dim, int k, const double *__restrict__ fail,
            const double *__restrict__ off) {
    if (solve_off(off) || fail[0] != 0.0) return;
    __shared__ double a[NB][NB + 1], w[NB][NB + 1];
    const int t = threadIdx.x, k0 = k * NB, kb = min(NB, dim - k0);
    const int i0 = (k + 1 + blockIdx.x) * NB, mb = min(NB, dim - i0);
    const double *I = Li + (size_t)k * NB * NB;
    for (int idx = t; idx < NB * NB; idx += kThreads) {
        const int r = idx >> 6, c = idx & 63;
        a[r][c] = (r < mb && c < kb) ? H[(size_t)(i0 + r) * dim + k0 + c] : 0.0;
        w[r][c] = I[idx];
    }
    __syncthreads();
    const int tr = (t >> 4) * 4, tc = (t & 15) * 4;          // 4 x 4 outputs per thread
    double acc[4][4] = {};
    for (int j = 0; j < kb; ++j) {
        double x[4], y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[e] = a[tr + e][j]; y[e] = w[tc + e][j]; }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[e][f] += x[e] * y[f];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int f = 0; f < 4; ++f)
            if (tr + e < mb && tc + f < kb) H[(size_t)(i0 + tr + e) * dim + k0 + tc + f] = acc[e][f];
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
        const int r = idx >> 6, c = idx & 63;
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

// y_k = Linv_kk b_k (64 x 64 lower-triangular matrix-vector product, 4 threads per row) - every workgroup of the launch
// computes it for itself from b_k, which nobody writes in this launch (block 0 stores y_k to a SEPARATE vector)
__device__ __forceinline__ void tri_matvec(const double *__restrict__ I, const double *__restrict__ v, double *y_lds, bool transposed) {
    const int t = threadIdx.x, r = t >> 2, q = t & 3;
    double s = 0.0;
    if (!transposed) { for (int c = q; c <= r; c += 4) s += I[r * NB + c] * v[c]; }
    else { for (int c = r + q; c < NB; c += 4) s += I[c * NB + r] * v[c]; }               // (L^-T)[r][c] = Linv[c][r], c >= r
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (q == 0) y_lds[r] = s;
}

// forward: grid.x = row block i - k; block 0 stores y_k, block i > 0 does b_i -= L_ik y_k
__global__ void __launch_bounds__(kThreads)
k_chol_fwd(const double *__restrict__ H, const double *__restrict__ Li, double *__restrict__ b, double *__restrict__ yout,
           int dim, int k, const double *__restrict__ fail, const double *__restrict__ off) {
    if (solve_off(off) || fail[0] != 0.0) return;
    __shared__ double v[NB], y[NB];
    const int t = threadIdx.x, k0 = k * NB, nb = min(NB, dim - k0);
    if (t < NB) v[t] = t < nb ? b[k0 + t] : 0.0;
    __syncthreads();
    tri_matvec(Li + (size_t)k * NB * NB, v, y, false);
    __syncthreads();
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
k_chol_bwd(const double *__restrict__ H, const double *__restrict__ Li, double *__restrict__ yv, double *__restrict__ xout,
           int dim, int k, const double *__restrict__ fail, const double *__restrict__ off) {
    if (solve_off(off) || fail[0] != 0.0) return;
    __shared__ double v[NB], y[NB];
    __shared__ double red[kThreads / NB][NB];
    const int t = threadIdx.x, k0 = k * NB, nb = min(NB, dim - k0);
    if (t < NB) v[t] = t < nb ? yv[k0 + t] : 0.0;
    __syncthreads();
    tri_matvec(Li + (size_t)k * NB * NB, v, y, true);
    __syncthreads();
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
// [m3_chol_diag_doubles(dim)] scratch for the diagonal factors AND their inverses, x [dim] out.  fail[0] is set to 1 on a
// non-positive pivot (then x is garbage); `off` (may be null): when off[0] != 0 the whole sequence is a no-op (the
// solver's device-side stop flag).
int64_t m3_chol_diag_doubles(int dim) { return 2 * (int64_t)((dim + NB - 1) / NB) * NB * NB; }

int m3_chol_solve_launch(double *H, double *b, double *y, double *x, double *Ld, double *fail, const double *off, int dim,
                         double shift, hipStream_t st) {
    const int nblk = (dim + NB - 1) / NB;
    double *Li = Ld + (size_t)nblk * NB * NB;
    for (int k = 0; k < nblk; ++k) {
        hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(kThreads), 0, st, (const double *)H, Ld, Li, dim, k, shift, fail, off);
        if (k + 1 < nblk) {
            hipLaunchKernelGGL(k_chol_trsm, dim3(nblk - k - 1), dim3(kThreads), 0, st, H, (const double *)Li, dim, k,
                               (const double *)fail, off);
            hipLaunchKernelGGL(k_chol_update, dim3(nblk - k - 1, nblk - k - 1), dim3(kThreads), 0, st, H, dim, k,
                               (const double *)fail, off);
        }
    }
    for (int k = 0; k < nblk; ++k)
        hipLaunchKernelGGL(k_chol_fwd, dim3(nblk - k), dim3(kThreads), 0, st, (const double *)H, (const double *)Li, b, y,
                           dim, k, (const double *)fail, off);
    for (int k = nblk - 1; k >= 0; --k)
        hipLaunchKernelGGL(k_chol_bwd, dim3(k + 1), dim3(kThreads), 0, st, (const double *)H, (const double *)Li, y, x, dim,
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
