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
 *     library allocates nothing and keeps no mutable state between calls except
 *     per-kernel, per-device "large-LDS opt-in done" bits (atomic; a process may
 *     drive several devices from several threads) and the thread-local text of
 *     the last HIP error;
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
/* Compute units of the current device (cached per device): the "fills the chip" grid thresholds of the convolution
 * dispatch (ops.conv3x3_direct_ok, the sliced single-pass convolution) derive from it instead of a constant 256. */
int m3_device_cu_count(void);

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
 * valid_out uint8 [B,N].  ws: uint32 [m3_iter_proj_ws_words(B, N, max_iter)] workspace (per-wave step
 * maxima, plain stores - no atomics - plus the per-item iteration limit).
 * stop_scope: 0 = reference behaviour, all points stop at the first LM iteration whose
 * max step norm over the WHOLE call is < convergence_thresh; 1 = per batch item. */
int64_t m3_iter_proj_ws_words(int B, int N, int max_iter);
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
/* Same with both descriptor arrays stored as IEEE half ("fp16 features", BASELINE configs[4]; halves the
 * kernel's HBM bytes).  Values are widened to fp32 exactly and scored with the same fp32 sequence,
 * so p_out equals m3_refine_matches on the half-rounded descriptors bit for bit. */
int m3_refine_matches_f16(const void *D11, const void *D21, const int32_t *p_in, int32_t *p_out,
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

/* Nearest neighbour in descriptor space, the search primitive of "fast reciprocal NN" matching (named by
 * BASELINE.json; absent from the reference tree - SURVEY 8a row K8 - so the semantics are this library's):
 * idx_out[b][s] = argmax_n <Q[b][s], DB[b][n]> in fp32 (two fused-multiply-add chains over the even and the
 * odd dimensions, added at the end), ties to the lowest n; score_out (may be NULL) = the maximum.  Q [B,S,D], DB [B,N,D] f32, D in {16, 24, 32},
 * 16-byte aligned; keys_ws: uint64 [B*S] scratch. */
int m3_nn_search(const float *Q, const float *DB, int32_t *idx_out, float *score_out, uint64_t *keys_ws,
                 int B, int S, int N, int D, void *stream);

/* The same search on the matrix cores (v_mfma_f32_16x16x32_f16, one k-step covers D <= 32): operands are packed to
 * K-padded fp16 first - fp16 descriptors (in_f16 = 1, BASELINE configs[4] "fp16 features") take 1 MFMA per 16 x 16
 * scores with exact products; fp32 descriptors (in_f16 = 0) are split hi + lo and take 3 (score error <= 2^-24 for
 * unit vectors).  Ties to the lowest n as above.  pack_ws: m3_nn_pack_bytes(...) bytes, 16-byte aligned. */
int64_t m3_nn_pack_bytes(int B, int S, int N, int in_f16);
int m3_nn_search_mfma(const void *Q, const void *DB, int32_t *idx_out, float *score_out, uint64_t *keys_ws,
                      void *pack_ws, int B, int S, int N, int D, int in_f16, void *stream);

/* Fast reciprocal nearest-neighbour matching, P pairs at once, loop on the device (MASt3R sec. 3.3; BASELINE.json names
 * it, the reference tree has no code for it: semantics in mast3r_slam/matching.py, oracle/matching.py).  Each
 * descriptor map [P,N,D] (fp32, or IEEE fp16 with in_f16 = 1; D in {16, 24, 32}) is packed ONCE
 * (m3_frnn_pack -> m3_frnn_pack_bytes(P, N, in_f16) bytes, 16-byte aligned) and serves as the database of one search
 * direction and as the query source of the other.  m3_frnn_round runs view 1 -> view 2 -> view 1 for every seed:
 * cur int32 [P,S] = the view-1 pixel a seed sits on (in / out), active uint8 [P,S] (in / out), got1 / got2 int32
 * [P,S] = this round's reciprocal pairs (-1 where none), xy2_ws int32 [P,S], keys_ws uint64 [P,S] scratch that
 * must be zero on entry and is left zero. */
int64_t m3_frnn_pack_bytes(int P, int N, int in_f16);
int m3_frnn_pack(const void *Dmap, void *packed, int P, int N, int D, int in_f16, void *stream);
int m3_frnn_round(const void *packed1, const void *packed2, int32_t *cur, uint8_t *active, int32_t *got1,
                  int32_t *got2, int32_t *xy2_ws, uint64_t *keys_ws, int P, int S, int N1, int N2, int in_f16,
                  void *stream);
/* The same round restricted to the seeds that are still active (rounds >= 2): act_ws int32 [P * (S + 1)] scratch receives
 * the ascending list of active seed slots per pair and their count; search workgroups past a pair's count exit at once.
 * Same results as m3_frnn_round. */
int m3_frnn_round_active(const void *packed1, const void *packed2, int32_t *cur, uint8_t *active, int32_t *got1,
                         int32_t *got2, int32_t *xy2_ws, uint64_t *keys_ws, int32_t *act_ws, int P, int S, int N1,
                         int N2, int in_f16, void *stream);
/* The same rounds with an EXACT pruned search (round 4).  The maps are H x W pixels in raster order (N = H * W).  Per
 * packed map, once per call, m3_frnn_blockstats writes for every 8 x 8 pixel tile a reference point (its centroid, fp16),
 * its radius and a norm bound (stats: m3_frnn_stats_bytes(P, H, W) bytes, 16-byte aligned).  A search then scores the
 * queries against the tile centroids on the matrix core (1/64 of the full search), drops every (16-query tile, block)
 * whose Cauchy-Schwarz bound <q, c> + |q| r lies below a lower bound of the query's final maximum, and scores the
 * surviving blocks with the MFMA sequence of the brute-force kernel - index and score equal m3_frnn_round's bit for bit
 * (a block holding the maximum or a tie always survives; ties go to the lowest pixel index).  When more than a quarter
 * of the pairs survive (descriptor maps without spatial coherence) the brute-force kernel runs instead; both test one
 * device counter, no host decision.  act_ws as in m3_frnn_round_active, or NULL for a round on every seed; seed_order
 * int32 [S] or NULL: the order in which the active slots are listed (with act_ws) - the searches work on groups of
 * consecutive entries, so an order that walks the seed grid in small patches prunes more; no result depends on it.
 * prune_ws: m3_frnn_prune_ws_bytes(...) bytes, 16-byte aligned.  Finite descriptors are assumed. */
int64_t m3_frnn_stats_bytes(int P, int H, int W);
int m3_frnn_blockstats(const void *packed, void *stats, int P, int H, int W, int in_f16, void *stream);
int64_t m3_frnn_prune_ws_bytes(int P, int S, int H1, int W1, int H2, int W2);
int m3_frnn_round_pruned(const void *packed1, const void *packed2, const void *stats1, const void *stats2, int32_t *cur,
                         uint8_t *active, int32_t *got1, int32_t *got2, int32_t *xy2_ws, uint64_t *keys_ws,
                         int32_t *act_ws, const int32_t *seed_order, void *prune_ws, int P, int S, int H1, int W1, int H2,
                         int W2, int in_f16, void *stream);
/* The reciprocal pairs of `rounds` rounds (got1 / got2 int32 [rounds,P,S]) as fixed-shape device outputs - no sort, no
 * host synchronisation (the matcher can be captured into a hipGraph): map1 int32 [P,N1] = view-1 pixel -> its partner
 * in view 2 (-1 = none); optionally the tracker's maps idx2 int64 [P,N2] / valid2 uint8 [P,N2] (view-2 pixel -> view-1
 * pixel; both or neither); pairs int32 [P,S,2] = the distinct (p1, p2) of every image pair sorted by p1, count int32 [P]
 * (rows >= count[pair] are -1; a seed converges at most once, so S bounds the number of pairs); chunk_ws int32
 * [P * m3_frnn_chunks(N1)] scratch. */
int m3_frnn_chunks(int N1);
int m3_frnn_collect(const int32_t *got1, const int32_t *got2, int rounds, int P, int S, int N1, int N2, int32_t *map1,
                    int64_t *idx2, uint8_t *valid2, int32_t *pairs, int32_t *count, int32_t *chunk_ws, void *stream);

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

/* Batched forms: P independent problems laid out back to back ([P,N,...] arrays, [P,8] poses,
 * counts int32 [P,2], info double [P,4], ws double [P * m3_track_ws_doubles()]); one launch
 * sequence serves all P (the per-GPU shard of a keyframe-pair batch). */
int m3_track_gather_batch(const float *Xf_canon, const float *Cf_avg, const float *Ck_avg,
                          const float *Qff, const float *Qkf, const int64_t *idx,
                          const uint8_t *valid_match, float *Xf_g, float *Qk, uint8_t *valid_opt,
                          uint8_t *valid_kf, int32_t *counts, int P, int N, float C_conf,
                          float Q_conf, void *stream);
int m3_track_gn_ray_dist_batch(const float *Xf, const float *Xk, const float *Qk,
                               const uint8_t *valid, const float *T_WCf, const float *T_WCk,
                               float *T_WCf_out, float *T_CkCf_out, double *info, double *ws,
                               int P, int N, int max_iters, float huber_k, float sigma_ray,
                               float sigma_dist, float rel_error, float delta_norm,
                               int fixed_iters, void *stream);

/* Number of doubles the tracking workspace needs (per problem). */
int64_t m3_track_ws_doubles(void);

/* FrameTracker._opt_pose_ray_dist_sim3 (tracker.py:258-324) with _solve (:216-256),
 * act_Sim3 / point_to_ray_dist (geometry.py:46-137), Sim3 inv/mul/exp/retr
 * (liegroups/sim3.py:107-262), check_convergence (optimizer.py:11-46) - the whole
 * <= max_iters Gauss-Newton loop runs on the device with no host round trip.
 * Xf [N,3] (already gathered), Xk [N,3], Qk [N], valid uint8 [N], T_WCf/T_WCk [8].
 * Out: T_WCf_out [8], T_CkCf_out [8], info double[4] = (iterations run, last cost,
 * last |tau|, status: 0 = iteration budget used up, 1 = converged, 2 = solve failed - singular
 * normal matrix or divergent step, the pose is the last good one; the reference raises there and
 * FrameTracker.track returns try_reloc, tracker.py:121-141).  ws: double[m3_track_ws_doubles()].
 * fixed_iters != 0 disables the convergence test (exactly max_iters iterations). */
int m3_track_gn_ray_dist(const float *Xf, const float *Xk, const float *Qk, const uint8_t *valid,
                         const float *T_WCf, const float *T_WCk,
                         float *T_WCf_out, float *T_CkCf_out, double *info, double *ws,
                         int N, int max_iters, float huber_k, float sigma_ray, float sigma_dist,
                         float rel_error, float delta_norm, int fixed_iters, void *stream);

/* FrameTracker._opt_pose_calib_sim3 (tracker.py:326-406) with project_calib (geometry.py:156-227):
 * residual (u, v, log z) of keyframe pixel n = (n % W, n / W) minus the projection of T . Xf[n], gated by
 * valid & Xk.z > depth_eps & projection inside the image.  Xf/Xk must already be ray-constrained
 * (m3_constrain_points_to_ray).  K4 = HOST array (fx, fy, cx, cy).  Batched over P problems like
 * m3_track_gn_ray_dist_batch; same outputs and workspace. */
int m3_track_gn_calib_batch(const float *Xf, const float *Xk, const float *Qk, const uint8_t *valid,
                            const float *T_WCf, const float *T_WCk, float *T_WCf_out,
                            float *T_CkCf_out, double *info, double *ws, int P, int N, int H, int W,
                            const float *K4, int max_iters, float huber_k, float sigma_pixel,
                            float sigma_depth, float pixel_border, float depth_eps, float rel_error,
                            float delta_norm, int fixed_iters, void *stream);

/* constrain_points_to_ray (geometry.py:273-302): out[n] = ((u-cx)/fx z, (v-cy)/fy z, z) with z = X[n].z
 * and (u, v) the pixel of n; X, out [P,H*W,3]; K4 = HOST (fx, fy, cx, cy). */
int m3_constrain_points_to_ray(const float *X, float *out, int P, int H, int W, const float *K4,
                               void *stream);

/* One Gauss-Newton normal-equation build at a given relative pose (the JTJ/JTr
 * reduction of tracker.py:239-244): out double[36] = H upper triangle (28, row-major),
 * g (7), cost (1).  ws as above. */
int m3_track_normal_eq(const float *Xf, const float *Xk, const float *Qk, const uint8_t *valid,
                       const float *T_CkCf, double *out, double *ws, int N, float huber_k,
                       float sigma_ray, float sigma_dist, void *stream);

/* Sim3.act over a point map (tracker.py:146 Xkk = T_CkCf.act(Xkf)): out = s R X + t. */
int m3_sim3_act(const float *T, const float *X, float *out, int N, void *stream);

/* ------------------------------------------------------------------ frame / keyframe state */

/* Frame.update_pointmap (frame.py:75-133), in place on the frame's X_canon [N,3] / C [N] (the summed
 * confidence; get_average_conf = C / N_frames stays on the host side).  If T (device, [8]) is not NULL
 * the new points are first moved by Sim3.act(T, .) - the keyframe update of tracker.py:146-147
 * (Xkk = T_CkCf.act(Xkf); keyframe.update_pointmap(Xkk, Ckf)) in one pass.  Modes:
 *   REPLACE             X, C <- new                      ("first" on its first update, "recent", a winning "best_score")
 *   INDEP_CONF          per point: take new where C_new > C           (frame.py:108-114)
 *   WEIGHTED_POINTMAP   X <- (C X + C_new X_new) / (C + C_new), C <- C + C_new   (:115-120, the default)
 *   WEIGHTED_SPHERICAL  the same average on (r, phi, theta)           (:121-129, geometry.py:318-351) */
enum { M3_FUSE_REPLACE = 0, M3_FUSE_INDEP_CONF = 1, M3_FUSE_WEIGHTED_POINTMAP = 2, M3_FUSE_WEIGHTED_SPHERICAL = 3 };
int m3_fuse_pointmap(float *X_canon, float *C, const float *X_new, const float *C_new, const float *T,
                     int N, int mode, void *stream);

/* "best_score" filtering (frame.py:59-73, :103-107) without the host: m3_median_f32 = the median of v[0..N) as
 * np.median / mx.median define it (mean of the two middle order statistics, float32) by an exact radix select on the
 * float bits - out float [1] on the device, ws uint32 [m3_median_ws_words()] scratch; any finite values, -0 < +0.
 * m3_fuse_pointmap_if_better: best_state float [2] on the device = (best score so far, flag); if *score_new >
 * best_state[0] the frame's pointmap is REPLACED (as M3_FUSE_REPLACE, T as above), best_state[0] <- *score_new and
 * best_state[1] <- 1, else nothing changes and best_state[1] <- 0.  Initialise best_state[0] with the first pointmap's
 * score (the first update always replaces, frame.py:88-92). */
int64_t m3_median_ws_words(void);
int m3_median_f32(const float *v, int N, uint32_t *ws, float *out, void *stream);
int m3_fuse_pointmap_if_better(float *X_canon, float *C, const float *X_new, const float *C_new, const float *T, int N,
                               const float *score_new, float *best_state, void *stream);

/* Number of distinct values among idx[n] with valid[n] != 0 (tracker.py:153-155, mx.unique(idx[valid])),
 * values in [0, range): bitmap (atomicOr) + popcount, an exact integer.  bitmap_ws: uint32
 * [m3_count_unique_ws_words(range)] scratch; count_out: int32 [1] on the device. */
int64_t m3_count_unique_ws_words(int range);
int m3_count_unique(const int64_t *idx, const uint8_t *valid, int N, int range, uint32_t *bitmap_ws,
                    int32_t *count_out, void *stream);

/* ------------------------------------------------------------------ backend GN ("rays") */

/* Per-edge normal-equation blocks of kernels.gauss_newton_rays (kernels.py:262-322; numpy
 * twin gauss_newton.py:100-251; Metal gn_jacobian_kernel gauss_newton.metal:66-252 + host
 * reduction gn_metal_runner.py:221-292).  Twc [K,8], Xs [K,P,3], Cs [K,P], ii,jj int32 [E],
 * idx int32 [E,P], valid uint8 [E,P], Q [E,P] -> blocks double [E,36] = (Hjj upper triangle
 * 28, gj 7, valid count 1).  With the reference's Ji = -Jj: Hii = Hjj, Hij = -Hjj, gi = -gj.
 * ws: double [E * m3_gn_rays_chunks(P) * 36].
 * point_mode = 1 selects kernels.gauss_newton_points (kernels.py:396-460, numpy twin
 * gauss_newton_points.py:17-207; Metal gn_points_jacobian_kernel gauss_newton_points.metal:65):
 * the same residual with the extra scale-invariant weight 1/(|Xi| + 1e-6) and sigma = sigma_point.
 * point_mode = 2 selects kernels.gauss_newton_calib (kernels.py:325-393, numpy twin
 * gauss_newton_calib.py:17-274; Metal gn_calib_jacobian_kernel gauss_newton_calib.metal:74): residual
 * ((du, dv)/sigma_pixel, dlog z/sigma_depth) with depth and image-border gates; `calib` is a HOST
 * array of 10 floats (fx, fy, cx, cy, width, height, border, z_eps, sigma_pixel, sigma_depth),
 * NULL otherwise; sigma_ray is ignored in that mode (pass any positive value). */
int m3_gn_rays_chunks(int P);
int m3_gn_rays_blocks(const float *Twc, const float *Xs, const float *Cs, const int32_t *ii,
                      const int32_t *jj, const int32_t *idx, const uint8_t *valid, const float *Q,
                      double *blocks, double *ws, int K, int P, int E, float sigma_ray,
                      float C_thresh, float Q_thresh, int point_mode, const float *calib,
                      void *stream);

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
 * Hbuf double [m3_gn_rays_hbuf_doubles(dim)], dim = 7*num_free: ANY size - systems up to
 * m3_gn_rays_max_dim() (63) are factored in place by one workgroup, larger ones (BASELINE configs[4]: 256
 * keyframes -> 1785 unknowns) by the blocked Cholesky of gn_chol.hip; either way the loop never
 * leaves the stream (the reference solves on the host, gauss_newton.py:253-260).
 * info double[4] = (iterations applied, last |dx|, converged/stopped flag, solver failure flag). */
int m3_gn_rays_max_dim(void);
int64_t m3_gn_rays_hbuf_doubles(int dim);
int m3_gn_rays_solve(float *Twc, const float *Xs, const float *Cs, const int32_t *ii,
                     const int32_t *jj, const int32_t *idx, const uint8_t *valid, const float *Q,
                     const int32_t *local, double *blocks, double *ws, double *Hbuf, double *info,
                     int K, int P, int E, int num_free, float sigma_ray, float C_thresh,
                     float Q_thresh, int max_iter, float delta_thresh, int point_mode,
                     const float *calib, void *stream);

/* One Gauss-Newton step from per-edge blocks the caller already has (the edge-sharded solve: every rank
 * evaluated its own edges, the 36-double blocks were all-gathered): assemble -> factor -> solve -> stop test
 * -> retract, stream-ordered, stop / failure flags on the device in info (m3_gn_rays_info_init once first). */
int m3_gn_rays_info_init(double *info, void *stream);
int m3_gn_rays_step(float *Twc, const double *blocks, const int32_t *ii, const int32_t *jj,
                    const int32_t *local, double *Hbuf, double *info, int K, int E, int num_free,
                    float delta_thresh, void *stream);

/* linalg.cholesky_solve (linalg.py:17-50) for a system of any size: (H + shift I) x = b in float64 by blocked
 * Cholesky (block 64), stream-ordered.  H [dim,dim] row-major (lower triangle read; destroyed), b [dim]
 * (destroyed), x [dim], ws double[m3_chol_ws_doubles(dim)]: ws[0] = 0 ok / 1 not positive definite (the rest is
 * scratch: the substitution vector and the 64 x 64 diagonal factors, which are kept OUT of H so that no
 * workgroup of a panel launch ever reads a diagonal block another one has already overwritten).
 * The backward substitution is ONE launch whose workgroups hand x_k to each other through device flags; a consumer
 * always has a higher workgroup index than its producer, i.e. the chain assumes workgroups are dispatched in index
 * order (true on gfx950, not promised by HIP).  The wait is bounded: if a producer has not published after ~0.3 s
 * of polling the launch gives up and ws[0] = 1 (status "failed"), it never hangs the stream. */
int64_t m3_chol_ws_doubles(int dim);
int m3_chol_solve(double *H, double *b, double *x, double *ws, int dim, double shift, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* M3SLAM_H */
