/*
 * m3slam.h - C ABI of libm3slam_hip.so: the MI355X (gfx950) hot path of MASt3R-SLAM.
 *
 * Drop-in boundary.  Every entry point replaces one array-in /
 * array-out operator of the reference's kernel-dispatch layer
 * (/root/reference/src/mlx_mast3r_slam/backends/mpsgraph/kernels.py) or one
 * fused span of its MLX host code; the replaced interface is cited per function.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (hipMalloc / torch ROCm storage) unless
 *     the parameter is documented as "host"; arrays are C-contiguous;
 *   - the caller owns and allocates every buffer, including workspaces; the
 *     library allocates nothing and keeps no global state;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work
 *     is stream-ordered and asynchronous, nothing synchronises the host;
 *   - return value: M3_OK (0) or a negative m3_status; no silent fallback exists;
 *   - float = IEEE binary32, poses are 8 floats [tx,ty,tz,qx,qy,qz,qw,s].
 */
#ifndef M3SLAM_H
#define M3SLAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    M3_OK = 0,
    M3_ERR_INVALID_ARG = -1,   /* null pointer, non-positive size, unsupported value */
    M3_ERR_LAUNCH = -2,        /* hipGetLastError() != hipSuccess after a launch */
    M3_ERR_UNSUPPORTED = -3    /* size outside what the kernel was built for */
} m3_status;

/* ABI version (major*1000 + minor) and human-readable status text. */
int m3_abi_version(void);
const char *m3_status_string(int status);
/* Text of the last HIP error seen by this library on the calling thread ("" if none). */
const char *m3_last_hip_error(void);

/* ------------------------------------------------------------------ matching */

/* prep_for_iter_proj (matching.py:134-175 + normalize_rays :121 + img_gradient
 * image.py:9-34).  X11,X21 [B,H,W,3]; idx_init int64 [B,H*W] or NULL (identity).
 * Out: rays_with_grad [B,H,W,9] = (ray, d ray/dx, d ray/dy), pts3d_norm [B,H*W,3],
 * p_init [B,H*W,2] = (idx % W, idx / W) as float. */
int m3_prep_iter_proj(const float *X11, const float *X21, const int64_t *idx_init,
                      float *rays_with_grad, float *pts3d_norm, float *p_init,
                      int B, int H, int W, void *stream);

/* kernels.iter_proj (kernels.py:107-148, numpy twin :151-254; Metal iter_proj.metal:82).
 * rays_with_grad [B,H,W,9], pts3d_norm [B,N,3], p_init [B,N,2] -> p_out [B,N,2],
 * valid_out uint8 [B,N].  ws: uint32 [B*max_iter + B] workspace (zeroed by the call).
 * stop_scope: 0 = reference behaviour, all points stop at the first LM iteration whose
 * max step norm over the WHOLE call is < convergence_thresh; 1 = per batch item. */
int m3_iter_proj(const float *rays_with_grad, const float *pts3d_norm, const float *p_init,
                 float *p_out, uint8_t *valid_out, uint32_t *ws,
                 int B, int H, int W, int N, int max_iter, float lambda_init,
                 float convergence_thresh, int stop_scope, void *stream);

/* kernels.refine_matches (kernels.py:463-493, numpy twin :496-537; Metal
 * refine_matches.metal:160).  D11 [B,H,W,D], D21 [B,N,D], p_in int32 [B,N,2] ->
 * p_out int32 [B,N,2].  chained = 0: numpy-twin semantics (every dilation pass
 * re-centres on p_in, i.e. the dilation-1 pass decides); 1: Metal semantics (passes
 * chain).  p_out must not alias p_in. */
int m3_refine_matches(const float *D11, const float *D21, const int32_t *p_in, int32_t *p_out,
                      int B, int H, int W, int D, int N, int radius, int dilation_max,
                      int chained, void *stream);

/* Tail of match_iterative_proj (matching.py:436-461): gather X11 at clip(p), 3-D distance
 * test, AND with valid_proj, idx = u + W*v.  p_f32 (iter_proj output, truncated like
 * .astype(int32), matching.py:410) is used when p_i32 is NULL. */
int m3_match_epilogue(const float *X11, const float *X21, const int32_t *p_i32, const float *p_f32,
                      const uint8_t *valid_proj, int64_t *idx_out, uint8_t *valid_out,
                      int B, int H, int W, float dist_thresh, void *stream);

/* match_simple (matching.py:41-90): idx = idx_init or arange; valid = |X11[idx]-X21| < thresh.
 * idx_out may alias idx_init. */
int m3_match_simple(const float *X11, const float *X21, const int64_t *idx_init,
                    int64_t *idx_out, uint8_t *valid_out, int B, int H, int W,
                    float dist_thresh, void *stream);

/* float [.,2] -> int32 [.,2] truncation (p.astype(int32), matching.py:410). */
int m3_trunc_i32(const float *p, int32_t *out, int64_t count, void *stream);

/* ------------------------------------------------------------------ tracking */

/* FrameTracker.track glue (tracker.py:88-113, _get_points_poses :177-214): for each
 * keyframe pixel n:  Xf_g = Xf_canon[idx[n]], Cf = Cf_avg[idx[n]],
 * Qk = sqrt(Qff[idx[n]] * Qkf[n]), valid_opt = valid_match & Cf>C_conf & Ck>C_conf & Qk>Q_conf,
 * valid_kf = valid_match & Qk>Q_conf.  counts int32[2] = (sum valid_opt, sum valid_kf)
 * (zeroed by the call). */
int m3_track_gather(const float *Xf_canon, const float *Cf_avg, const float *Ck_avg,
                    const float *Qff, const float *Qkf, const int64_t *idx,
                    const uint8_t *valid_match, float *Xf_g, float *Qk,
                    uint8_t *valid_opt, uint8_t *valid_kf, int32_t *counts,
                    int N, float C_conf, float Q_conf, void *stream);

/* Number of doubles the tracking workspace needs. */
int64_t m3_track_ws_doubles(void);

/* FrameTracker._opt_pose_ray_dist_sim3 (tracker.py:258-324) with _solve (:216-256),
 * act_Sim3 / point_to_ray_dist (geometry.py:46-137), Sim3 inv/mul/exp/retr
 * (liegroups/sim3.py:107-262), check_convergence (optimizer.py:11-46) - the whole
 * <= max_iters Gauss-Newton loop runs on the device with no host round trip.
 * Xf [N,3] (already gathered), Xk [N,3], Qk [N], valid uint8 [N], T_WCf/T_WCk [8].
 * Out: T_WCf_out [8], T_CkCf_out [8], info double[4] = (iterations run, last cost,
 * last |tau|, converged flag).  ws: double[m3_track_ws_doubles()].
 * fixed_iters != 0 disables the convergence test (exactly max_iters iterations). */
int m3_track_gn_ray_dist(const float *Xf, const float *Xk, const float *Qk, const uint8_t *valid,
                         const float *T_WCf, const float *T_WCk,
                         float *T_WCf_out, float *T_CkCf_out, double *info, double *ws,
                         int N, int max_iters, float huber_k, float sigma_ray, float sigma_dist,
                         float rel_error, float delta_norm, int fixed_iters, void *stream);

/* One Gauss-Newton normal-equation build at a given relative pose (the JTJ/JTr
 * reduction of tracker.py:239-244): out double[36] = H upper triangle (28, row-major),
 * g (7), cost (1).  ws as above. */
int m3_track_normal_eq(const float *Xf, const float *Xk, const float *Qk, const uint8_t *valid,
                       const float *T_CkCf, double *out, double *ws, int N, float huber_k,
                       float sigma_ray, float sigma_dist, void *stream);

/* Sim3.act over a point map (tracker.py:146 Xkk = T_CkCf.act(Xkf)): out = s R X + t. */
int m3_sim3_act(const float *T, const float *X, float *out, int N, void *stream);

/* ------------------------------------------------------------------ backend GN ("rays") */

/* Per-edge normal-equation blocks of kernels.gauss_newton_rays (kernels.py:262-322; numpy
 * twin gauss_newton.py:100-251; Metal gn_jacobian_kernel gauss_newton.metal:66-252 + host
 * reduction gn_metal_runner.py:221-292).  Twc [K,8], Xs [K,P,3], Cs [K,P], ii,jj int32 [E],
 * idx int32 [E,P], valid uint8 [E,P], Q [E,P] -> blocks double [E,36] = (Hjj upper triangle
 * 28, gj 7, valid count 1).  With the reference's Ji = -Jj: Hii = Hjj, Hij = -Hjj, gi = -gj.
 * ws: double [E * m3_gn_rays_chunks(P) * 36]. */
int m3_gn_rays_chunks(int P);
int m3_gn_rays_blocks(const float *Twc, const float *Xs, const float *Cs, const int32_t *ii,
                      const int32_t *jj, const int32_t *idx, const uint8_t *valid, const float *Q,
                      double *blocks, double *ws, int K, int P, int E, float sigma_ray,
                      float C_thresh, float Q_thresh, void *stream);

/* Dense normal equations from the per-edge blocks (gauss_newton.py:220-251): H double
 * [dim,dim], g double [dim], dim = 7*num_free (both zeroed by the call; the 1e-6 I
 * regulariser of gauss_newton.py:254 is NOT added here). */
int m3_gn_rays_assemble(const double *blocks, const int32_t *ii, const int32_t *jj,
                        const int32_t *local, double *H, double *g, int K, int E, int num_free,
                        void *stream);

/* T[kf] <- exp(dx[7*local[kf] ..]) * T[kf] for every free keyframe (retract_sim3,
 * sim3_ops.py:229-251; Metal pose_update_kernel gauss_newton.metal:255). */
int m3_gn_rays_retract(float *Twc, const double *dx, const int32_t *local, int K, void *stream);

/* Whole gauss_newton_rays loop on the device (gauss_newton.py:95-280): per iteration
 * blocks -> dense H (7F x 7F, + 1e-6 I) and g -> Cholesky solve -> |dx| < delta_thresh stop
 * -> T <- exp(dx) T (retract_sim3, sim3_ops.py:229) for the free keyframes.
 * local int32 [K]: keyframe -> free-block index, < 0 = pinned or unused (host builds it
 * from unique(ii,jj) and pin, gauss_newton.py:73-81).  Twc is updated IN PLACE.
 * Hbuf double [dim*dim + 2*dim], dim = 7*num_free <= m3_gn_rays_max_dim().
 * info double[4] = (iterations applied, last |dx|, converged/stopped flag, solver failure flag). */
int m3_gn_rays_max_dim(void);
int m3_gn_rays_solve(float *Twc, const float *Xs, const float *Cs, const int32_t *ii,
                     const int32_t *jj, const int32_t *idx, const uint8_t *valid, const float *Q,
                     const int32_t *local, double *blocks, double *ws, double *Hbuf, double *info,
                     int K, int P, int E, int num_free, float sigma_ray, float C_thresh,
                     float Q_thresh, int max_iter, float delta_thresh, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* M3SLAM_H */
