// Device-side quaternion / Sim(3) algebra, templated on the scalar (float for the
// per-point work, double for the per-iteration pose bookkeeping).
// Pose layout: [tx,ty,tz, qx,qy,qz,qw, s].
//
// Two families, as in the reference (see oracle/sim3.py for the citations):
//   *_mlx   : liegroups/so3.py + sim3.py (tracker; right-multiplied retraction)
//   *_ops   : backends/mpsgraph/sim3_ops.py (backend GN; left-multiplied, full W matrix)
#pragma once
#include <hip/hip_runtime.h>

template <typename T> struct V3 { T x, y, z; };
template <typename T> struct Q4 { T x, y, z, w; };
template <typename T> struct Pose { V3<T> t; Q4<T> q; T s; };

template <typename T> __device__ __forceinline__ V3<T> v3(T x, T y, T z) { return V3<T>{x, y, z}; }
template <typename T> __device__ __forceinline__ V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> __device__ __forceinline__ V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> __device__ __forceinline__ V3<T> operator*(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename T> __device__ __forceinline__ T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> __device__ __forceinline__ V3<T> cross(V3<T> a, V3<T> b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

template <typename T> __device__ __forceinline__ Q4<T> qmul(Q4<T> a, Q4<T> b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
            a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w,
            a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
template <typename T> __device__ __forceinline__ Q4<T> qinv(Q4<T> q) { return {-q.x, -q.y, -q.z, q.w}; }
// v + qw*(2 q x v) + q x (2 q x v)
template <typename T> __device__ __forceinline__ V3<T> qrot(Q4<T> q, V3<T> v) {
    V3<T> qv{q.x, q.y, q.z};
    V3<T> u = T(2) * cross(qv, v);
    return v + q.w * u + cross(qv, u);
}

template <typename T, typename S> __device__ __forceinline__ Pose<T> load_pose(const S *p) {
    Pose<T> P;
    P.t = {T(p[0]), T(p[1]), T(p[2])};
    P.q = {T(p[3]), T(p[4]), T(p[5]), T(p[6])};
    P.s = T(p[7]);
    return P;
}
template <typename T, typename S> __device__ __forceinline__ void store_pose(S *p, Pose<T> P) {
    p[0] = S(P.t.x); p[1] = S(P.t.y); p[2] = S(P.t.z);
    p[3] = S(P.q.x); p[4] = S(P.q.y); p[5] = S(P.q.z); p[6] = S(P.q.w);
    p[7] = S(P.s);
}

// ---- tracker family (liegroups/sim3.py) ------------------------------------
template <typename T> __device__ __forceinline__ V3<T> act(Pose<T> P, V3<T> p) { return P.s * qrot(P.q, p) + P.t; }
template <typename T> __device__ __forceinline__ Pose<T> inv_mlx(Pose<T> P) {
    Pose<T> R;
    R.q = qinv(P.q);
    R.s = T(1) / (P.s + T(1e-10));
    R.t = (-R.s) * qrot(R.q, P.t);
    return R;
}
template <typename T> __device__ __forceinline__ Pose<T> mul(Pose<T> A, Pose<T> B) {
    Pose<T> R;
    R.q = qmul(A.q, B.q);
    R.s = A.s * B.s;
    R.t = A.t + A.s * qrot(A.q, B.t);
    return R;
}
__device__ inline Pose<double> exp_mlx(const double *tau) {
    V3<double> v{tau[0], tau[1], tau[2]}, w{tau[3], tau[4], tau[5]};
    const double th2 = dot(w, w), th = sqrt(th2 + 1e-10);
    const bool small = th2 < 1e-8;
    const double A = small ? 1.0 - th2 / 6.0 : sin(th) / th;
    const double B = small ? 0.5 - th2 / 24.0 : (1.0 - cos(th)) / th2;
    const double C = small ? 1.0 / 6.0 - th2 / 120.0 : (1.0 - A) / th2;
    V3<double> wv = cross(w, v);
    Pose<double> P;
    P.t = v + B * wv + C * cross(w, wv);
    const double sinc_half = small ? 0.5 - th2 / 48.0 : sin(0.5 * th) / th;
    const double cos_half = small ? 1.0 - th2 / 8.0 : cos(0.5 * th);
    P.q = {sinc_half * w.x, sinc_half * w.y, sinc_half * w.z, cos_half};
    P.s = exp(tau[6]);
    return P;
}

// ---- backend family (sim3_ops.py) ---------------------------------------------
__device__ inline Pose<double> rel_ops(Pose<double> Ti, Pose<double> Tj) {     // Ti^-1 * Tj
    Pose<double> R;
    const double si_inv = 1.0 / Ti.s;
    R.s = si_inv * Tj.s;
    Q4<double> qi = qinv(Ti.q);
    R.q = qmul(qi, Tj.q);
    R.t = si_inv * qrot(qi, Tj.t - Ti.t);
    return R;
}
__device__ inline Pose<double> exp_ops(const double *xi) {
    constexpr double EPS = 1e-6;
    V3<double> tau{xi[0], xi[1], xi[2]}, w{xi[3], xi[4], xi[5]};
    const double sigma = xi[6];
    const double th2 = dot(w, w), th = sqrt(th2 + EPS);
    const bool small_t = th2 < EPS, small_s = fabs(sigma) < EPS;
    const double s = exp(sigma);
    Pose<double> P;
    const double imag = small_t ? 0.5 - th2 / 48.0 : sin(0.5 * th) / th;
    const double real = small_t ? 1.0 - th2 / 8.0 : cos(0.5 * th);
    P.q = {imag * w.x, imag * w.y, imag * w.z, real};
    P.s = s;
    const double C = small_s ? 1.0 : (s - 1.0) / sigma;
    double A, B;
    if (small_s) {
        A = small_t ? 0.5 : (1.0 - cos(th)) / th2;
        B = small_t ? 1.0 / 6.0 : (th - sin(th)) / (th2 * th);
    } else if (small_t) {
        A = ((sigma - 1.0) * s + 1.0) / (sigma * sigma);
        B = (s * 0.5 * sigma * sigma + s - 1.0 - sigma * s) / (sigma * sigma * sigma);
    } else {
        A = (s * sin(th) * sigma + (1.0 - s * cos(th)) * th) / (th * (th2 + sigma * sigma));
        B = (C - ((s * cos(th) - 1.0) * sigma + s * sin(th) * th) / (th2 + sigma * sigma)) / th2;
    }
    V3<double> c1 = cross(w, tau), c2 = cross(w, c1);
    P.t = C * tau + A * c1 + B * c2;
    return P;
}
__device__ inline Pose<double> retract_ops(const double *xi, Pose<double> T) {  // exp(xi) * T
    Pose<double> d = exp_ops(xi), R;
    R.q = qmul(d.q, T.q);
    R.t = d.s * qrot(d.q, T.t) + d.t;
    R.s = d.s * T.s;
    return R;
}

// Dense n x n solve by Gaussian elimination with partial pivoting (n <= 7), in place.
// Returns false when a pivot is exactly zero / non-finite.
template <int NN> __device__ inline bool solve_small(double (*A)[NN], double *b, int n) {
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = fabs(A[k][k]);
        for (int r = k + 1; r < n; ++r) { double v = fabs(A[r][k]); if (v > best) { best = v; piv = r; } }
        if (!(best > 0.0) || !isfinite(best)) return false;
        if (piv != k) {
            for (int c = 0; c < n; ++c) { double t = A[k][c]; A[k][c] = A[piv][c]; A[piv][c] = t; }
            double t = b[k]; b[k] = b[piv]; b[piv] = t;
        }
        const double inv = 1.0 / A[k][k];
        for (int r = k + 1; r < n; ++r) {
            const double f = A[r][k] * inv;
            for (int c = k; c < n; ++c) A[r][c] -= f * A[k][c];
            b[r] -= f * b[k];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        double v = b[k];
        for (int c = k + 1; c < n; ++c) v -= A[k][c] * b[c];
        b[k] = v / A[k][k];
    }
    return true;
}

// ---- normal-equation contribution of one point whose Jacobian rows are J_c = M(p) a_c --------------------------
// M(p) = [st * I ; [p]x ; p^T]  (7 x 3: J = [st * a, p x a, a . p] - Sim(3) tangent order translation, rotation,
// scale).  With A = sum_c w_c a_c a_c^T (3 x 3 symmetric) and b = sum_c w_c r_c a_c:
//     H = M A M^T  (28 upper-triangular entries, row-major: h[0..27]),   g = gs * M b  (h[28..34])
// i.e. 36 FMAs for (A, b) + ~60 for the congruence instead of (28 + 7) products per residual row.
__device__ __forceinline__ void accum_congruence(const V3<float> &p, float Axx, float Axy, float Axz, float Ayy,
                                                 float Ayz, float Azz, const V3<float> &b, float st, float gs,
                                                 float *h) {
    const V3<float> Ap{Axx * p.x + Axy * p.y + Axz * p.z, Axy * p.x + Ayy * p.y + Ayz * p.z, Axz * p.x + Ayz * p.y + Azz * p.z};
    const V3<float> c0 = cross(p, V3<float>{Axx, Axy, Axz}), c1 = cross(p, V3<float>{Axy, Ayy, Ayz}),
                    c2 = cross(p, V3<float>{Axz, Ayz, Azz});                       // H_t,omega[i][j] = st * c_i[j]
    const V3<float> w0 = cross(p, V3<float>{c0.x, c1.x, c2.x}), w1 = cross(p, V3<float>{c0.y, c1.y, c2.y}),
                    w2 = cross(p, V3<float>{c0.z, c1.z, c2.z});                    // H_omega,omega[i][j] = w_i[j]
    const V3<float> pAp = cross(p, Ap), pb = cross(p, b);
    const float st2 = st * st;
    h[0] += st2 * Axx; h[1] += st2 * Axy; h[2] += st2 * Axz; h[3] += st * c0.x; h[4] += st * c0.y; h[5] += st * c0.z; h[6] += st * Ap.x;
    h[7] += st2 * Ayy; h[8] += st2 * Ayz; h[9] += st * c1.x; h[10] += st * c1.y; h[11] += st * c1.z; h[12] += st * Ap.y;
    h[13] += st2 * Azz; h[14] += st * c2.x; h[15] += st * c2.y; h[16] += st * c2.z; h[17] += st * Ap.z;
    h[18] += w0.x; h[19] += w0.y; h[20] += w0.z; h[21] += pAp.x;
    h[22] += w1.y; h[23] += w1.z; h[24] += pAp.y;
    h[25] += w2.z; h[26] += pAp.z;
    h[27] += dot(p, Ap);
    h[28] += gs * st * b.x; h[29] += gs * st * b.y; h[30] += gs * st * b.z;
    h[31] += gs * pb.x; h[32] += gs * pb.y; h[33] += gs * pb.z; h[34] += gs * dot(p, b);
}
