// Blocked dense Cholesky solve in float64 for the backend Gauss-Newton systems (BASELINE configs[4]: 256 keyframes ->
// 1785 unknowns).  Replaces the reference's host solve  dx = np.linalg.solve(H + 1e-6 I, -g)  (gauss_newton.py:253-260,
// linalg.py:17-50) - and the torch.linalg (hipSOLVER) call plus three host synchronisations per iteration that round 1 used.
//
// Right-looking blocked factorisation, block size 64, entirely stream-ordered (no host round trip; a failed pivot sets a
// device flag that turns every later kernel of the solve into a no-op):
//   step k:  k_chol_diag    ONE workgroup: factor the diagonal block A_kk (+ shift) and invert the factor in one fused
//                           64-step register-resident loop; L_kk and L_kk^-1 go to SEPARATE buffers (Ld, Li
//                           [nblk][64][64]) - A_kk in H is never overwritten, so no kernel ever reads a diagonal block
//                           another workgroup is rewriting (round-2 advisor finding); also y_k = L_kk^-1 b_k
//            k_chol_trsm    every row block i > k:  L_ik = A_ik L_kk^-T  as a 64 x 64 x 64 float64-MFMA product against
//                           the inverse, and the forward substitution's b_i -= L_ik y_k
//            k_chol_update  every lower block pair i >= j > k:  A_ij -= L_ik L_jk^T  (float64 MFMA)
//   then k_chol_bwd_chain   the whole backward substitution in one launch (flag-chained workgroups).
// 28 + 27 + 27 + 1 = 83 launches for 1785 unknowns.  H is row-major [dim, dim]; only the lower triangle is read.
// Round-3 history at 1785 unknowns (device time of one solve): diagonal block factored redundantly in every workgroup of
// a panel launch, row-per-thread triangular solves: 5.6 ms; one diagonal workgroup + products with the inverse: 4.1 ms;
// register-resident diagonal kernel, forward substitution fused: 2.3 ms; see the kernels below for the later steps.
#include "common.h"

namespace {

constexpr int NB = 64;
constexpr int kThreads = 256;

__device__ __forceinline__ bool solve_off(const double *flag) { return flag && flag[0] != 0.0; }

// sqrt(x) and 1 / sqrt(x) from v_rsq_f64 + two coupled Newton (Goldschmidt) steps and one residual correction of the
// root - a dozen dependent instructions on the critical path of an elimination step instead of the ~60 of sqrt() and a
// division.  Both within 2 ulp.
__device__ __forceinline__ void sqrt_rsqrt(double x, double &sq, double &inv) {
    const double y0 = __builtin_amdgcn_rsq(x);
    double g = x * y0, h = 0.5 * y0;
    double r = fma(-g, h, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-g, h, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    const double dlt = fma(-g, g, x);
    sq = fma(dlt, h, g);
    inv = h + h;
}

// The diagonal kernel runs 8 waves in two roles of 256 threads; thread t' = t & 255 of a role: tr = t' & 15, tc = t' >> 4
// owns the 4 x 4 elements (tr + 16 a, tc + 16 b) of ITS matrix IN REGISTERS - role 0 the block being factored, role 1
// the running inverse.  Per elimination step only the scaled pivot column travels through LDS (double-buffered): ONE
// barrier per step.  PA = 16-column panel of the step, a template parameter so that the blocks a step touches are fixed
// at compile time (factor: a >= b >= PA; inverse: a >= PA >= b).
//   role 0:  column p owners (tc == p % 16) take the pivot from dg (below), scale the column, publish it; after the
//            barrier every thread eliminates the column from its elements
//   role 1:  right-looking forward substitution on X = I: X[p][:] *= 1 / L[p][p], X[r][:] -= L[r][p] X[p][:] (r > p).
//            X[p][c] lives in thread (tr = p % 16, same tc) - the same 16-lane row - and arrives by a lane shuffle;
//            L[r][p] is the published column (zero for r <= p).  The inverse is a pure consumer of what the factor
//            publishes: it costs no barrier of its own and, on its own waves, no issue slot of the
//            update -> pivot -> rsqrt -> scale -> publish chain that bounds a step.
// (One fused loop on 4 waves: 26 us per launch, the shuffles and the X updates sat in the chain.)
template <int PA>
__device__ __forceinline__ void chol_diag_panel(int role, double (&e)[4][4], double (&dg)[4], double (*colbuf)[NB],
                                                double *invd, int *bad, int tr, int tc) {
    for (int pp = 0; pp < 16; ++pp) {
        const int p = PA * 16 + pp;
        double *cb = colbuf[pp & 1];
        if (role == 0 && tc == pp) {
            // dg[b] = A[c][c] - sum_{q < p} L[c][q]^2 of the thread's own columns c = tc + 16 b, kept by EVERY thread
            // from the column values it reads anyway: the pivot needs no exchange
            const double piv = dg[PA];
            const bool ok = piv > 0.0 && piv < __builtin_huge_val();
            double sq, inv;
            sqrt_rsqrt(ok ? piv : 1.0, sq, inv);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int r = tr + 16 * a;
                if (a >= PA) {
                    double v = e[a][PA];
                    v = r > p ? v * inv : (r == p ? sq : v);
                    e[a][PA] = v;
                    cb[r] = r > p ? v : 0.0;
                } else {
                    cb[r] = 0.0;
                }
            }
            if (tr == 0) {
                invd[p] = inv;
                if (!ok) *bad = 1;
            }
        }
        __syncthreads();
        double lr[4];
#pragma unroll
        for (int a = PA; a < 4; ++a) lr[a] = cb[tr + 16 * a];
        if (role == 0) {
            double lc[4];
#pragma unroll
            for (int a = PA; a < 4; ++a) lc[a] = cb[tc + 16 * a];
#pragma unroll
            for (int a = PA; a < 4; ++a) {
                dg[a] = fma(-lc[a], lc[a], dg[a]);
#pragma unroll
                for (int b = PA; b <= a; ++b) e[a][b] = fma(-lr[a], lc[b], e[a][b]);
            }
        } else {
            const double s = invd[p];
#pragma unroll
            for (int b = 0; b <= PA; ++b) {
                const double xv = __shfl(e[PA][b], pp, 16) * s;
#pragma unroll
                for (int a = PA; a < 4; ++a) e[a][b] = fma(-lr[a], xv, e[a][b]);
                if (tr == pp) e[PA][b] = xv;
            }
        }
    }
}

// One workgroup.  A_kk + shift I -> L_kk (Ld[k]), its inverse (Li[k]) and the forward-substitution block
// y_k = L_kk^-1 b_k (b_k is final here: the trsm launches of the earlier block columns have subtracted L_kj y_j).
// History of this kernel at 1785 unknowns: factorisation in LDS with 16 dependent read-modify-writes per thread and
// step, sqrt + division, 3 barriers, then a 64-step inversion: 87 us per launch, 61 % of the solve; register-resident
// with a separate inversion loop: 30 us; one fused 64-step loop: 26 us; factor and inverse on separate waves: see above.
// Rows / columns >= nb (last block) are the identity.
constexpr int kDiagThreads = 512;

__global__ void __launch_bounds__(kDiagThreads)
k_chol_diag(const double *__restrict__ H, double *__restrict__ Ld, double *__restrict__ Li, const double *__restrict__ bvec,
            double *__restrict__ yout, unsigned *__restrict__ flags, int dim, int k, double shift, double *__restrict__ fail,
            const double *__restrict__ off) {
    if (solve_off(off) || fail[0] != 0.0) return;
    if (k == 0)
        for (int i = threadIdx.x; i < (dim + NB - 1) / NB; i += kDiagThreads) flags[i] = 0u;   // the backward chain's hand-over flags
    __shared__ double d[NB][NB + 1];
    __shared__ double xi[NB][NB + 1];
    __shared__ double colbuf[2][NB];
    __shared__ double invd[NB], vb[NB];
    __shared__ int bad;
    const int t = threadIdx.x, role = t >> 8, tr = t & 15, tc = (t & 255) >> 4;
    const int k0 = k * NB, nb = min(NB, dim - k0);
    if (t == 0) bad = 0;
    if (t < NB) vb[t] = t < nb ? bvec[k0 + t] : 0.0;
    for (int idx = t; idx < NB * NB; idx += kDiagThreads) {
        const int r = idx >> 6, c = idx & 63;
        double v = (r == c) ? 1.0 : 0.0;
        if (r < nb && c <= r) v = H[(size_t)(k0 + r) * dim + k0 + c] + (r == c ? shift : 0.0);
        d[r][c] = v;
    }
    __syncthreads();
    double e[4][4], dg[4];                                   // role 0: the block and its running diagonal; role 1: X = I
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) e[a][b] = role == 0 ? d[tr + 16 * a][tc + 16 * b] : ((a == b && tr == tc) ? 1.0 : 0.0);
#pragma unroll
    for (int b = 0; b < 4; ++b) dg[b] = d[tc + 16 * b][tc + 16 * b];
    __syncthreads();
    chol_diag_panel<0>(role, e, dg, colbuf, invd, &bad, tr, tc);
    chol_diag_panel<1>(role, e, dg, colbuf, invd, &bad, tr, tc);
    chol_diag_panel<2>(role, e, dg, colbuf, invd, &bad, tr, tc);
    chol_diag_panel<3>(role, e, dg, colbuf, invd, &bad, tr, tc);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int r = tr + 16 * a, c = tc + 16 * b;
            (role == 0 ? d : xi)[r][c] = c <= r ? e[a][b] : 0.0;
        }
    __syncthreads();
    if (bad) {
        if (t == 0) fail[0] = 1.0;
        return;
    }
    double *L = Ld + (size_t)k * NB * NB, *I = Li + (size_t)k * NB * NB;
    for (int idx = t; idx < NB * NB; idx += kDiagThreads) {
        const int r = idx >> 6, c = idx & 63;
        L[idx] = d[r][c];
        I[idx] = c <= r ? xi[r][c] : 0.0;
    }
    if (t < kThreads) {   // y_k = L^-1 b_k: 4 threads per row
        const int r = t >> 2, q = t & 3;
        double s = 0.0;
        for (int c = q; c <= r; c += 4) s += xi[r][c] * vb[c];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (q == 0 && r < nb) yout[k0 + r] = s;
    }
}

// C[64 x 64] = A B^T of two row-major 64-row x 64-k tiles on the float64 matrix pipe (v_mfma_f64_16x16x4_f64), operands
// straight from global memory into the instruction's register layout - no LDS, no barrier.  Wave w of the 4 owns the
// 32 x 32 quadrant (w >> 1, w & 1) as 2 x 2 MFMA blocks.  Lane l = (i = l % 16, g = l / 16) feeds row i of a block at
// k-slot g; the k index a slot stands for is free (any permutation of the summation): slot g of steps 2 q, 2 q + 1 is
// k = 8 q + 2 g, 8 q + 2 g + 1, so one 16-byte load per lane feeds two steps and the four lanes of a row read 64
// contiguous bytes (with k = 16 g + s, every lane of a load touched its own cache line: 64 lines per instruction, the
// kernel was bound by the L1 address path, 15 us for ONE tile).  Result layout (probed on gfx950,
// tools/experiments/mfma_f64_probe.hip): acc[v] = C[g + 4 v][i] of the block.
// (The first version staged both tiles in LDS and did 16 FMAs per 8 LDS reads and thread: LDS-bound, 23.5 us per
// update launch and 15.3 us per trsm launch at 1785 unknowns.)
typedef double d4 __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(8))) dpair { double x, y; };   // rows are only 8-byte aligned (odd dim)

__device__ __forceinline__ void tile_nt_mfma(const double *__restrict__ A, int lda, int ma, const double *__restrict__ B,
                                             int ldb, int mb, d4 (&acc)[2][2]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    double af[2][16], bf[2][16];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ra = (w >> 1) * 32 + h * 16 + i, rb = (w & 1) * 32 + h * 16 + i;
        const double *pa = A + (size_t)ra * lda + 2 * g, *pb = B + (size_t)rb * ldb + 2 * g;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const dpair va = ra < ma ? *(const dpair *)(pa + 8 * q) : dpair{0.0, 0.0};
            const dpair vb = rb < mb ? *(const dpair *)(pb + 8 * q) : dpair{0.0, 0.0};
            af[h][2 * q] = va.x; af[h][2 * q + 1] = va.y;
            bf[h][2 * q] = vb.x; bf[h][2 * q + 1] = vb.y;
        }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[x][s], bf[y][s], acc[x][y], 0, 0, 0);
}

// grid.x = row block i - k - 1:  L_ik = A_ik L_kk^-T, i.e. out[r][c] = sum_j A_ik[r][j] Linv[c][j], in place; the forward
// substitution rides along: b_i -= L_ik y_k.  (Launched only for k < nblk - 1: the k extent is a full 64.)
__global__ void __launch_bounds__(kThreads)
k_chol_trsm(double *__restrict__ H, const double *__restrict__ Li, double *__restrict__ bvec, const double *__restrict__ yv,
            int dim, int k, const double *__restrict__ fail, const double *__restrict__ off) {
    if (solve_off(off) || fail[0] != 0.0) return;
    __shared__ double red[2][NB];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, i = lane & 15, g = lane >> 4, k0 = k * NB;
    const int i0 = (k + 1 + blockIdx.x) * NB, mb = min(NB, dim - i0);
    double *Aik = H + (size_t)i0 * dim + k0;
    d4 acc[2][2];
    tile_nt_mfma(Aik, dim, mb, Li + (size_t)k * NB * NB, NB, NB, acc);
    __syncthreads();                                         // both column halves have read the rows they overwrite
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = (w >> 1) * 32 + x * 16 + g + 4 * v;
            double sum = 0.0;
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int c = (w & 1) * 32 + y * 16 + i;
                if (r < mb) Aik[(size_t)r * dim + c] = acc[x][y][v];
                sum += acc[x][y][v] * yv[k0 + c];
            }
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            sum += __shfl_xor(sum, 4, 64);
            sum += __shfl_xor(sum, 8, 64);
            if (i == 0) red[w & 1][r] = sum;
        }
    __syncthreads();
    if (t < mb) bvec[i0 + t] -= red[0][t] + red[1][t];
}

// grid = (bj, bi) offsets over the trailing lower triangle: block (i, j) with i >= j > k:  A_ij -= L_ik L_jk^T
__global__ void __launch_bounds__(kThreads)
k_chol_update(double *__restrict__ H, int dim, int k, const double *__restrict__ fail, const double *__restrict__ off) {
    if (solve_off(off) || fail[0] != 0.0) return;
    const int ib = k + 1 + blockIdx.y, jb = k + 1 + blockIdx.x;
    if (jb > ib) return;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, i = lane & 15, g = lane >> 4;
    if (ib == jb && w == 1) return;                          // the strictly upper quadrant of a diagonal block
    const int k0 = k * NB, i0 = ib * NB, j0 = jb * NB;
    const int mi = min(NB, dim - i0), mj = min(NB, dim - j0);
    d4 acc[2][2];
    tile_nt_mfma(H + (size_t)i0 * dim + k0, dim, mi, H + (size_t)j0 * dim + k0, dim, mj, acc);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = (w >> 1) * 32 + x * 16 + g + 4 * v, c = (w & 1) * 32 + y * 16 + i;
                if (r < mi && c < mj && (ib > jb || c <= r)) H[(size_t)(i0 + r) * dim + j0 + c] -= acc[x][y][v];
            }
}

// Backward substitution x = L^-T y as ONE launch: workgroup w owns block column k = nblk - 1 - w,
//   x_k = L_kk^-T (y_k - sum_{i > k} L_ik^T x_i),
// accumulates the terms in the order the x_i appear (i = nblk - 1 downwards: a fixed order) and publishes x_k behind a
// flag.  Producers have lower workgroup indices than their consumers, so they are dispatched no later and a spinning
// consumer can never keep its producer off the machine.  The tile of the next term is fetched BEFORE its flag is
// awaited: a stage of the chain costs one flag hand-over + 64 x 64 multiply-adds + the triangular product from LDS.
// (28 dependent launches of ~10 us each before.)
constexpr unsigned kChainPolls = 1u << 23;
__global__ void __launch_bounds__(kThreads)
k_chol_bwd_chain(const double *__restrict__ H, const double *__restrict__ Li, const double *__restrict__ yv, double *xout,
                 unsigned *flags, int dim, int nblk, double *fail, const double *__restrict__ off) {
    if (solve_off(off) || fail[0] != 0.0) return;
    __shared__ double li[NB][NB + 1];
    __shared__ double xs[NB], v[NB];
    __shared__ double red[kThreads / NB][NB];
    __shared__ int bail;
    if (threadIdx.x == 0) bail = 0;
    const int t = threadIdx.x, k = nblk - 1 - (int)blockIdx.x, k0 = k * NB, nb = min(NB, dim - k0);
    const double *I = Li + (size_t)k * NB * NB;
    for (int idx = t; idx < NB * NB; idx += kThreads) li[idx >> 6][idx & 63] = I[idx];
    const int c = t & 63, part = t >> 6;                     // thread (c, part): column c of the tile, every 4th row
    double tile[16], acc = 0.0;
    auto load_tile = [&](int i) {
        const int i0 = i * NB, mi = min(NB, dim - i0);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = part + 4 * e;
            tile[e] = r < mi ? H[(size_t)(i0 + r) * dim + k0 + c] : 0.0;      // k < nblk - 1 here: column k0 + c < dim
        }
    };
    if (k < nblk - 1) load_tile(nblk - 1);
    for (int i = nblk - 1; i > k; --i) {
        if (t == 0) {
            // The wait is BOUNDED (kChainPolls polls of >= 64 cycles each, ~0.3 s): the chain relies on workgroups being
            // dispatched in index order (see above and include/m3slam.h), which HIP does not promise; if a producer never
            // shows up the solve reports failure through fail[0] instead of hanging the stream.
            unsigned polls = 0;
            while (__hip_atomic_load(&flags[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                if (++polls > kChainPolls) { bail = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (bail) {                                          // workgroup-uniform: give up, tell the host, release consumers
            if (t == 0) {
                __hip_atomic_store(&fail[0], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&flags[k], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (t < NB) xs[t] = i * NB + t < dim ? __hip_atomic_load(&xout[i * NB + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) acc = fma(tile[e], xs[part + 4 * e], acc);
        if (i - 1 > k) load_tile(i - 1);
    }
    red[part][c] = acc;
    __syncthreads();
    if (t < NB) v[t] = (t < nb ? yv[k0 + t] : 0.0) - ((red[0][t] + red[1][t]) + (red[2][t] + red[3][t]));
    __syncthreads();
    {   // x_k[r] = sum_{c >= r} Linv[c][r] v[c], 4 threads per r
        const int r = t >> 2, q = t & 3;
        double s = 0.0;
        for (int cc = r + q; cc < NB; cc += 4) s = fma(li[cc][r], v[cc], s);
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (q == 0 && r < nb) __hip_atomic_store(&xout[k0 + r], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
    __syncthreads();
    if (t == 0) __hip_atomic_store(&flags[k], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

// (H + shift I) x = b: H [dim, dim] row-major float64 (lower triangle read; the off-diagonal blocks are overwritten by
// L, the diagonal blocks keep A_kk of the running factorisation), b [dim] (destroyed), y [dim] scratch, Ld
// [m3_chol_diag_doubles(dim)] scratch for the diagonal factors AND their inverses, x [dim] out.  fail[0] is set to 1 on a
// non-positive pivot (then x is garbage); `off` (may be null): when off[0] != 0 the whole sequence is a no-op (the
// solver's device-side stop flag).
int64_t m3_chol_diag_doubles(int dim) { return (int64_t)((dim + NB - 1) / NB) * (2 * NB * NB + 1); }

int m3_chol_solve_launch(double *H, double *b, double *y, double *x, double *Ld, double *fail, const double *off, int dim,
                         double shift, hipStream_t st) {
    const int nblk = (dim + NB - 1) / NB;
    double *Li = Ld + (size_t)nblk * NB * NB;
    unsigned *flags = (unsigned *)(Li + (size_t)nblk * NB * NB);     // one hand-over flag per block column (reset by k_chol_diag, k = 0)
    for (int k = 0; k < nblk; ++k) {
        hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(kDiagThreads), 0, st, (const double *)H, Ld, Li, (const double *)b, y, flags,
                           dim, k, shift, fail, off);
        if (k + 1 < nblk) {
            hipLaunchKernelGGL(k_chol_trsm, dim3(nblk - k - 1), dim3(kThreads), 0, st, H, (const double *)Li, b, (const double *)y,
                               dim, k, (const double *)fail, off);
            hipLaunchKernelGGL(k_chol_update, dim3(nblk - k - 1, nblk - k - 1), dim3(kThreads), 0, st, H, dim, k,
                               (const double *)fail, off);
        }
    }
    hipLaunchKernelGGL(k_chol_bwd_chain, dim3(nblk), dim3(kThreads), 0, st, (const double *)H, (const double *)Li,
                       (const double *)y, x, flags, dim, nblk, fail, off);
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
