/*
 * m3slam_model.h - C ABI of the network operators in libm3slam_hip.so (gfx950).
 *
 * These are the device ops behind model.encode / model.reconstruct, the two calls the
 * reference makes into its (absent) third-party network package `mlx_mast3r`
 * (/root/reference/src/mlx_mast3r_slam/mast3r_utils.py:278,281,347-355,418-421).
 * The reference tree holds no source for that arithmetic; the architecture follows the
 * public MASt3R / DUSt3R / CroCo-v2 definition (DESIGN.md "Model").  Conventions as in
 * m3slam.h: device pointers, caller-owned buffers, stream-ordered, int status.
 * 16-bit tensors are passed as void* (round-to-nearest-even).  Every operator that converts to or
 * from the 16-bit storage type exists as NAME_dt(..., int dtype, void *stream) with dtype =
 * M3_DT_BF16 (v_mfma_f32_16x16x32_bf16) or M3_DT_F16 (IEEE half, v_mfma_f32_16x16x32_f16 - same
 * MFMA rate, 3 more mantissa bits, range 65504); the *_bf16 names are the dtype = M3_DT_BF16 forms.
 * This is the `precision` argument of load_mast3r (mast3r_utils.py:47-52: "fp16" | "fp32" | "bf16").
 * Pure data movement (m3_relu_bf16 by sign bit, m3_concat2_bf16, m3_unshuffle_bf16) serves both types.
 */
#ifndef M3SLAM_MODEL_H
#define M3SLAM_MODEL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 16-bit storage type of a launch: operands, 16-bit outputs and 16-bit residuals share it.
 * M3_DT_F16_PVBF16 is the attention form of the fp16 trunk and is accepted by the RoPE projections
 * (m3_gemm_rope*_dt, m3_gemm_grouped2*_dt with M3_EPI_BF16_ROPE) and by m3_attention_prescaled_dt only: an fp16 launch
 * whose v columns (those >= rope_cols) are STORED as bf16, and an attention whose S = Q K^T product runs on fp16
 * operands while V and the probabilities are bf16 (the operand type of the deferred-maximum loop; q / k keep fp16's
 * 11-bit mantissa, which is what peaked softmax rows need - DESIGN.md section 4). */
enum { M3_DT_BF16 = 0, M3_DT_F16 = 1, M3_DT_F16_PVBF16 = 2 };

/* epilogue selectors of m3_gemm_* / m3_conv3x3_* ("BF16" = the launch's 16-bit type) */
enum {
    M3_EPI_BF16 = 0,       /* C(bf16) = acc + bias */
    M3_EPI_BF16_GELU = 1,  /* C(bf16) = gelu_erf(acc + bias) */
    M3_EPI_F32 = 2,        /* C(f32)  = acc + bias */
    M3_EPI_F32_ACCUM = 3,  /* C(f32)  = R(f32) + acc + bias   (residual stream; C may alias R) */
    M3_EPI_BF16_RELU = 4,  /* C(bf16) = relu(acc + bias) */
    M3_EPI_BF16_ADD = 5,   /* C(bf16) = R(bf16) + acc + bias  (C may alias R) */
    M3_EPI_BF16_ROPE = 6,  /* C(bf16) = rope2d(acc + bias) on the leading rope_cols columns */
    /* flag, OR-ed into the epilogue of m3_conv3x3_dt / m3_conv3x3_grouped2_dt: the convolution reads relu(X) - the
     * ReLU is applied to the operand fragments in registers, relu(X) is never written (DPT residual unit: relu -> conv1) */
    M3_EPI_INPUT_RELU = 0x100
};

/* C[M,N] = epi(A[M,K] . W[N,K]^T + bias): A, W bf16 K-major (torch nn.Linear layout), fp32
 * accumulation on v_mfma_f32_16x16x32_bf16.  K % 64 == 0, N % 4 == 0, ldc >= N. */
int m3_gemm_bf16(const void *A, const void *W, const float *bias, void *C, const void *R,
                 int M, int N, int K, int ldc, int epilogue, void *stream);
int m3_gemm_dt(const void *A, const void *W, const float *bias, void *C, const void *R,
               int M, int N, int K, int ldc, int epilogue, int dtype, void *stream);

/* Which kernel m3_gemm_bf16 / _rope / _grouped2 dispatch a dense [M,N] problem to: 256 or 192 = the
 * 256-row ping-pong kernel with 256- / 192-wide tiles, 128 or 64 = the small-problem kernel. */
int m3_gemm_pick_tile(int M, int N, int groups);
/* Diagnostic hook (tests, tools/gemm_shapes.py): force the tile shape of every later dense launch of the process -
 * 64, 128, 192 or 256; 0 = automatic choice.  Every shape accumulates
 * K in the same order, so the choice never changes a result bit.  Returns the previous setting (initially the value of
 * the environment variable M3_GEMM_TILE, or 0). */
int m3_gemm_set_tile(int tile);

/* Projection GEMM with RoPE-2D fused into the epilogue: C(16-bit) = rope(A . W^T + bias) on the
 * 64-wide heads in columns [0, rope_cols) (q|k of a q|k|v projection), plain bias add beyond.
 * Row m is token m % tokens_per_image of its image; rope_tok f32 [tokens_per_image][2][2][16] holds, per
 * token and axis (0: its y position, 1: its x position), 16 cosines then 16 sines for the frequencies
 * base^(-i/16) - i.e. rope_tok[t][a][0|1][i] = cos_sin[pos_yx[t][a]][i][0|1] of m3_rope2d_bf16's tables,
 * gathered once per image size; 16-byte aligned.  N % 64 == 0. */
int m3_gemm_bf16_rope(const void *A, const void *W, const float *bias, void *C, int M, int N, int K,
                      int ldc, const float *rope_tok, int tokens_per_image, int rope_cols, void *stream);
/* _dt forms: columns [0, q_cols) (the q heads; q_cols <= rope_cols) are additionally multiplied by q_scale after the
 * rotation and before the 16-bit rounding - softmax scale * log2(e) folded into q, which m3_attention_prescaled_dt
 * expects (q_cols = 0: nothing is scaled). */
int m3_gemm_rope_dt(const void *A, const void *W, const float *bias, void *C, int M, int N, int K,
                    int ldc, const float *rope_tok, int tokens_per_image, int rope_cols, int q_cols,
                    float q_scale, int dtype, void *stream);

/* Position mode of the fused RoPE epilogue: instead of the per-token cos/sin table (64 B read per 64 B written) the
 * kernel takes the tokens' grid positions pos_yx int32 [tokens_per_image][2] (y, x) and computes cos/sin of
 * pos * base^(-i/16), i = 0..15, itself (hardware sin/cos, absolute error ~1e-6; base = 100 for CroCo's RoPE100).
 * Same results as m3_gemm_rope_dt with the table built from the same positions, to that accuracy. */
int m3_gemm_rope_pos_dt(const void *A, const void *W, const float *bias, void *C, int M, int N, int K, int ldc,
                        const int32_t *pos_yx, int tokens_per_image, float base, int rope_cols, int q_cols, float q_scale,
                        int dtype, void *stream);
int m3_gemm_grouped2_rope_pos_dt(const void *A, const void *W0, const void *W1, const float *bias0, const float *bias1,
                                 void *C, int M, int N, int K, int ldc, int64_t a_gstride, int64_t c_gstride,
                                 const int32_t *pos_yx, int tokens_per_image, float base, int rope_cols, int q_cols,
                                 float q_scale, int dtype, void *stream);
/* Two same-shape GEMMs in one launch (the two decoder branches have different weights): group g
 * (0/1) computes C + g*c_gstride = epi((A + g*a_gstride) . W[g]^T + bias[g]); strides in elements.
 * epilogue may be any M3_EPI_* including M3_EPI_BF16_ROPE (= 6; then the RoPE tables are required). */
int m3_gemm_bf16_grouped2(const void *A, const void *W0, const void *W1, const float *bias0,
                          const float *bias1, void *C, const void *R, int M, int N, int K, int ldc,
                          int64_t a_gstride, int64_t c_gstride, int epilogue, const float *rope_tok,
                          int tokens_per_image, int rope_cols, void *stream);
int m3_gemm_grouped2_dt(const void *A, const void *W0, const void *W1, const float *bias0,
                        const float *bias1, void *C, const void *R, int M, int N, int K, int ldc,
                        int64_t a_gstride, int64_t c_gstride, int epilogue, const float *rope_tok,
                        int tokens_per_image, int rope_cols, int q_cols, float q_scale, int dtype, void *stream);

/* General dense GEMM entry (one or two groups, any epilogue, position-mode RoPE) with the LayerNorm FOLD: the reference's
 * network applies LayerNorm between every residual update and the projection that follows (public MASt3R / CroCo block:
 * x + proj(attn(norm1(x))), x + fc2(gelu(fc1(norm2(x))))).  Instead of a LayerNorm pass over the fp32 stream (read 4 B,
 * write 2 B per element, one launch per norm) the two GEMMs around it share the work:
 *   producer (epilogue M3_EPI_F32 / M3_EPI_F32_ACCUM, c16 / stats_out set): besides the fp32 stream x' it writes c16 = x'
 *     rounded to the launch's 16-bit type (same ldc) and, per row and 32-column slot, (sum x', sum x'^2) of the fp32 values
 *     into stats_out [N/32][M][2] (slot-major: a tile's rows are contiguous within a slot; N % 32 == 0, M even);
 *   consumer (16-bit epilogues, ln_stats set): A is such a copy of the RAW stream, W = gamma-scaled weights, ln_colsum[n] =
 *     sum_k W[n][k] of the ROUNDED 16-bit weights, bias = b + W . beta; with mean / rstd of row m from ln_stats
 *     [ln_slots][M][2] (ln_slots = K / 32, a multiple of 4) the epilogue computes rstd[m] * (acc - mean[m] * ln_colsum[n])
 *     + bias[n] = LayerNorm(x')[m] . W_orig[n]^T + b[n] up to the rounding of x' (instead of LayerNorm(x')) to 16 bits.
 * Every kernel adds a row's slots in the same order, so the statistics - like the products - do not depend on the tile
 * shape a launch is dispatched to.  Group 1 of a 2-group launch reads ln_stats + ln_gstride (floats; may be negative: the
 * decoder's cross-attention memory is the OTHER branch's stream) and writes stats_out + stats_gstride, c16 + c_gstride.
 * Fields not used by a launch are 0 / NULL.  rope_pos != NULL selects M3_EPI_BF16_ROPE's position mode. */
typedef struct m3_gemm_desc {
    const void *A, *W, *W1;            /* W1: group 1's weights (groups == 2) */
    const float *bias, *bias1;
    void *C;
    const void *R;
    const int32_t *rope_pos;
    void *c16;
    float *stats_out;
    const float *ln_stats, *ln_colsum, *ln_colsum1;
    const void *r_lo;                  /* hi / lo stream (fp16 launches, producer): the residual stream lives in two 16-bit planes, */
    void *c_lo;                        /* x = hi + lo (hi = x rounded to fp16, lo = the rounded remainder: 22 bits).  R = hi in,     */
                                       /* r_lo = lo in, c16 = hi out, c_lo = lo out (in place allowed); C is NULL and not written.   */
                                       /* The hi plane IS the consumer's operand: the residual launch moves 4 + 4 bytes per element  */
                                       /* as with an fp32 stream, and neither a LayerNorm pass nor a 16-bit copy exists.             */
    int64_t a_gstride, c_gstride, ln_gstride, stats_gstride;
    int32_t M, N, K, ldc, epilogue, dtype, groups;
    int32_t tokens_per_image, rope_cols, q_cols, ln_slots;
    float rope_base, q_scale, ln_eps;
    int32_t stats_slots;               /* producer: slots of the stats_out buffer = m3_ln_slot_count(M, N, groups) (checked)               */
    int32_t rope_max_pos;              /* RoPE: > 0 promises that every entry of rope_pos is in [0, rope_max_pos) and rope_max_pos <= 64  */
                                       /* (a 1024-pixel side): each workgroup then builds the rope_max_pos x 16 cos / sin table once in   */
                                       /* LDS instead of eight v_sin / v_cos per lane, row and 32-column block (same values bit for bit). */
                                       /* 0 (or > 64): computed per element.  A position outside the promise takes the last table row.    */
} m3_gemm_desc;
int m3_gemm_ex(const m3_gemm_desc *desc, void *stream);
/* Statistics slots a producer launch of a [M, N] stream (groups 1 or 2) writes per row: stats_out is [slots][M][2] floats per
 * group.  A slot is a node of the rows' canonical sum tree (32-column leaves -> 64-column pairs -> 128-column halves -> top
 * nodes of 192 columns for N % 192 == 0, else of 256 for N % 256 == 0; at most 4 top nodes) - the widest one the launch's tile
 * width is a multiple of: N / 64, N / 128 or one slot per 256- / 192-wide tile.  A consumer is given the same count in ln_slots
 * (its K = N) and finishes the tree; every combination gives a row the same bits.  0: no statistics for this width. */
int m3_ln_slot_count(int M, int N, int groups);

/* 3x3 convolution, padding 1, stride 1 or 2, as an implicit GEMM: X bf16 NHWC [B,H,W,Cin],
 * W bf16 [Cout,3,3,Cin], Y NHWC [B,OH,OW,Cout].  Cin % 64 == 0, Cout % 4 == 0.  zero16: 16
 * zero bytes in device memory (source of the padding taps). */
/* Two same-shape convolutions in one launch (the two DPT heads): X [2,B,H,W,Cin], Y / R [2,B,OH,OW,Cout], group g
 * uses (W_g, bias_g).  Split-K scratch: 2 x m3_conv3x3_splitk_bytes(B, ...). */
int m3_conv3x3_grouped2_dt(const void *X, const void *W0, const void *W1, const float *bias0, const float *bias1,
                           void *Y, const void *R, const void *zero16, int B, int H, int Wd, int Cin, int Cout,
                           int stride, int epilogue, void *splitk_ws, int64_t splitk_ws_bytes, int dtype,
                           void *stream);

/* Split-K scratch of m3_conv3x3_bf16: small feature maps with a long K (the 16x16 / 32x32 DPT maps,
 * K = 9*Cin up to 6912) cannot fill the chip with output tiles, so they are multiplied in K-slices
 * into fp32 partial planes which a second kernel sums in a fixed order before applying the epilogue.
 * The slice count depends on the per-image geometry only (not on B): a pair's result does not depend
 * on the batch it is computed in.  Returns the scratch bytes the call needs (0: direct path). */
int64_t m3_conv3x3_splitk_bytes(int B, int H, int Wd, int Cin, int Cout, int stride);

int m3_conv3x3_bf16(const void *X, const void *W, const float *bias, void *Y, const void *R,
                    const void *zero16, int B, int H, int Wd, int Cin, int Cout, int stride,
                    int epilogue, void *splitk_ws, int64_t splitk_ws_bytes, void *stream);
int m3_conv3x3_dt(const void *X, const void *W, const float *bias, void *Y, const void *R,
                  const void *zero16, int B, int H, int Wd, int Cin, int Cout, int stride,
                  int epilogue, void *splitk_ws, int64_t splitk_ws_bytes, int dtype, void *stream);

/* Tail of the DPT head in one launch (head.2 conv3x3 128->128 + ReLU, head.4 1x1 128->4, pointmap
 * post-processing): pts [B,H,W,3] = xyz/|xyz| * expm1(|xyz|), conf [B,H,W] = 1 + exp(c) with
 * (xyz, c) = W4 . relu(conv3x3(X, W) + bias) + b4.  W [128,3,3,Cin], W4 [4,128] 16-bit; the
 * 128-channel full-resolution map is never written and never rounded to 16 bits (the unfused chain
 * m3_conv3x3(RELU) -> m3_gemm(F32) -> m3_pts_post rounds it once: that rounding was the largest
 * single term of the pointmap error against the fp32 oracle). */
int m3_conv3x3_relu_head4(const void *X, const void *W, const float *bias, const void *W4, const float *b4,
                          float *pts, float *conf, const void *zero16, int B, int H, int Wd, int Cin,
                          void *stream);
int m3_conv3x3_relu_head4_dt(const void *X, const void *W, const float *bias, const void *W4, const float *b4,
                             float *pts, float *conf, const void *zero16, int B, int H, int Wd, int Cin,
                             int dtype, void *stream);

/* The same tail as a DIRECT convolution with the x2 bilinear (align_corners) upsampling of its input fused in
 * (public DPT head: head.0 -> Upsample(x2) -> head.2 conv3x3 + ReLU -> head.4 1x1; oracle/model.py dpt_head):
 * X NHWC [B,H/2,W/2,128] when upsample != 0, else [B,H,W,128]; W [128,3,3,128], W4 [4,128] in the 16-bit dtype.
 * A workgroup stages the 18x18x128 halo of its 16x16 output tile once in LDS (interpolating it on the way in),
 * so the full-resolution 128-channel map is neither written nor re-read.  H, W multiples of 16. */
int m3_dpt_tail_dt(const void *X, const void *W, const float *bias, const void *W4, const float *b4,
                   float *pts, float *conf, const void *zero16, int B, int H, int Wd, int upsample,
                   int dtype, void *stream);

/* m3_dpt_tail_dt for both heads in one launch: X [2,B,h,w,128], pts [2,B,H,W,3], conf [2,B,H,W]. */
int m3_dpt_tail_grouped2_dt(const void *X, const void *Wc0, const void *Wc1, const float *bias0, const float *bias1,
                            const void *W40, const void *W41, const float *b40, const float *b41, float *pts,
                            float *conf, const void *zero16, int B, int H, int Wd, int upsample, int dtype,
                            void *stream);

/* head.0 of the public DPT head with ITS x2 upsample fused in, as the same direct convolution (oracle/model.py
 * dpt_head: refinenet1's trailing x2 interpolation -> head.0 conv3x3 256 -> 128): X NHWC [B,H/2,W/2,Cin] when upsample
 * != 0, else [B,H,W,Cin]; W [128,3,3,Cin] (Cin = 256 or 128); Y NHWC [B,H,W,128] = conv(up(X)) + bias in the 16-bit
 * dtype.  The upsampled Cin-channel map is neither written nor re-read.  H, W multiples of 16. */
int m3_conv3x3_up_direct_dt(const void *X, const void *W, const float *bias, void *Y, const void *zero16, int B, int H,
                            int Wd, int Cin, int upsample, int dtype, void *stream);
/* ... for both heads in one launch: X [2,B,h,w,Cin], Y [2,B,H,W,128]; head g uses (W_g, bias_g). */
int m3_conv3x3_up_direct_grouped2_dt(const void *X, const void *W0, const void *W1, const float *bias0,
                                     const float *bias1, void *Y, const void *zero16, int B, int H, int Wd, int Cin,
                                     int upsample, int dtype, void *stream);

/* The direct convolution as a general 3x3 / pad 1 / stride 1 operator for the wide DPT maps (the residual units of the
 * fusion blocks, 256 -> 256 channels at 128 x 128 / 64 x 64): X NHWC [(2,)B,H,W,Cin], W_g [Cout,3,3,Cin], Y (and R) NHWC
 * [(2,)B,H,W,Cout], Cin, Cout in {128, 256}, H, W multiples of 16; epilogue M3_EPI_BF16 | M3_EPI_BF16_RELU |
 * M3_EPI_BF16_ADD, optionally | M3_EPI_INPUT_RELU.  W1 == NULL: a single group.  Returns THE SAME BITS as
 * m3_conv3x3_dt / m3_conv3x3_grouped2_dt on the same operands (both walk K as (64-channel slice, tap, k-step) and apply
 * the epilogue in the same order): the caller picks by problem size - the direct form pays once the grid
 * (H/16 * ceil(W/32) * B * Cout/128 * groups workgroups) fills the chip. */
int m3_conv3x3_direct_grouped2_dt(const void *X, const void *W0, const void *W1, const float *bias0, const float *bias1,
                                  void *Y, const void *R, const void *zero16, int B, int H, int Wd, int Cin, int Cout,
                                  int epilogue, int dtype, void *stream);

/* Fused multi-head attention, head dim 64: O = softmax(scale * Q K^T) V, bf16 in/out, fp32
 * softmax.  Q/K/V/O are addressed as base + batch*batch_stride + token*row_stride + head*64
 * (element units), so q, k, v may live interleaved in one [tokens, 3C] projection buffer.
 * Keys/values of batch item b are read from item (b + kv_batch_shift) % nbatch (decoder
 * cross-attention to the other view).  Any Tq, Tk >= 1 (key tail masked with -inf, query tail
 * rows not stored): resize_img emits every multiple of 16, e.g. 512x336 -> 672 tokens. */
int m3_attention_bf16(const void *Q, const void *K, const void *V, void *O, int q_row_stride,
                      int kv_row_stride, int o_row_stride, int64_t q_batch_stride,
                      int64_t kv_batch_stride, int64_t o_batch_stride, int nbatch, int heads,
                      int Tq, int Tk, int kv_batch_shift, float scale, void *stream);
int m3_attention_dt(const void *Q, const void *K, const void *V, void *O, int q_row_stride,
                    int kv_row_stride, int o_row_stride, int64_t q_batch_stride,
                    int64_t kv_batch_stride, int64_t o_batch_stride, int nbatch, int heads,
                    int Tq, int Tk, int kv_batch_shift, float scale, int dtype, void *stream);

/* CroCo RoPE-2D ("RoPE100") in place on the 64-wide heads of X [tokens,row_stride] bf16:
 * dims 0..31 rotate with the token's y, 32..63 with its x; pos_yx int32 [tokens_per_image,2],
 * cos_sin f32 [max_pos,16,2]. */
int m3_rope2d_bf16(void *X, const int32_t *pos_yx, const float *cos_sin, int row_stride, int tokens,
                   int heads, int tokens_per_image, void *stream);
/* The same with q PRE-SCALED by softmax scale * log2(e) (m3_gemm_rope_dt's q_scale): a score is an exp2 argument as
 * it leaves the matrix core and the reference maximum enters the MFMA as its accumulator initialiser.  fp16 operands:
 * the tile maximum is tracked and the output rescaled only when it exceeds the reference by 2^8.  bf16 operands: the
 * reference is the first key tile's maximum, row sums are accumulated by the matrix core, the range is kept by a 2^-64
 * rescale when a row sum passes 2^60, and a workgroup whose exp2 overflowed (a score more than 127 above that
 * reference) recomputes its query block with the tracking loop - same result to the rounding of P. */
int m3_attention_prescaled_dt(const void *Q, const void *K, const void *V, void *O, int q_row_stride,
                              int kv_row_stride, int o_row_stride, int64_t q_batch_stride,
                              int64_t kv_batch_stride, int64_t o_batch_stride, int nbatch, int heads,
                              int Tq, int Tk, int kv_batch_shift, int dtype, void *stream);
int m3_rope2d_dt(void *X, const int32_t *pos_yx, const float *cos_sin, int row_stride, int tokens,
                 int heads, int tokens_per_image, int dtype, void *stream);

/* y(bf16)[M,C] = LayerNorm(x(f32)[M,C]) * gamma + beta; C % 256 == 0, C <= 2048. */
int m3_layernorm_bf16(const float *x, const float *gamma, const float *beta, void *y, int M, int C,
                      float eps, void *stream);
int m3_layernorm_dt(const float *x, const float *gamma, const float *beta, void *y, int M, int C,
                    float eps, int dtype, void *stream);

/* Two-group LayerNorm in one launch: rows [0,M) use (gamma0,beta0), rows [M,2M) use (gamma1,beta1);
 * output row r normalises input row (r + in_row_shift) % (2M) (in_row_shift = M swaps the halves:
 * the decoder's norm_y of the OTHER view). */
int m3_layernorm_bf16_grouped2(const float *x, const float *gamma0, const float *beta0, const float *gamma1,
                               const float *beta1, void *y, int M, int C, int in_row_shift, float eps,
                               void *stream);
int m3_layernorm_grouped2_dt(const float *x, const float *gamma0, const float *beta0, const float *gamma1,
                             const float *beta1, void *y, int M, int C, int in_row_shift, float eps,
                             int dtype, void *stream);
/* LayerNorm of a residual stream kept as two fp16 planes (x = float(hi) + float(lo); m3_gemm_desc.c16 / c_lo): rows
 * [0, split) use (gamma0, beta0), rows [split, rows) use (gamma1, beta1).  Equals m3_layernorm_dt on hi + lo bit for bit.
 * Replaces the reference's enc_norm / dec_norm module calls on the fp16 trunk (the encoder / decoder forward the reference
 * calls at mast3r_utils.py:278-294). */
int m3_layernorm_hl_dt(const void *hi, const void *lo, const float *gamma0, const float *beta0, const float *gamma1,
                       const float *beta1, void *y, int rows, int C, int split, float eps, int dtype, void *stream);
/* Decoder block entry: two LayerNorms of the SAME rows in one pass.  x f32 [2,M,C] (two branches);
 * y_own[g][r] = LN(x[g][r]; ga_g, ba_g) (norm1) and y_cross[1-g][r] = LN(x[g][r]; gb_(1-g), bb_(1-g)) (norm_y: the
 * tokens of branch g as the other branch's cross-attention memory).  Bit-identical to m3_layernorm_grouped2_dt called
 * twice (in_row_shift 0 and M). */
int m3_layernorm_dual2_dt(const float *x, const float *ga0, const float *ba0, const float *ga1, const float *ba1,
                          const float *gb0, const float *bb0, const float *gb1, const float *bb1, void *y_own,
                          void *y_cross, int M, int C, float eps, int dtype, void *stream);

/* uint8 image [B,H,W,3] -> bf16 patch matrix [B*(H/16)*(W/16), 768] (column c*256+py*16+px),
 * normalised (v/255-0.5)/0.5 (resize_img, mast3r_utils.py:186-188). */
int m3_patchify16(const uint8_t *img, void *A, int B, int H, int W, void *stream);
int m3_patchify16_dt(const uint8_t *img, void *A, int B, int H, int W, int dtype, void *stream);

int m3_f32_to_bf16(const float *x, void *y, int64_t n, void *stream);               /* n % 4 == 0 */
int m3_cast_f32_dt(const float *x, void *y, int64_t n, int dtype, void *stream);    /* f32 -> 16-bit, n % 4 == 0 */
int m3_cast16(const void *x, void *y, int64_t n, int from_dtype, int to_dtype, void *stream);   /* bf16 <-> f16, n % 8 == 0 */
int m3_relu_bf16(const void *x, void *y, int64_t n, void *stream);                  /* n % 8 == 0; bf16 or f16 */
int m3_add_bf16(const void *a, const void *b, void *y, int64_t n, void *stream);     /* n % 8 == 0 */
int m3_add_dt(const void *a, const void *b, void *y, int64_t n, int dtype, void *stream);
int m3_concat2_bf16(const void *a, const void *b, void *out, int64_t M, int Ca, int Cb, void *stream);
/* Packed snapshot of a step's result tensors into ONE send buffer (the all-gather's operand): segment i copies
 * nbytes from src to dst + dst_off (16-byte aligned; mode 0), or writes nbytes / 4 int32 values narrowed from int64 (mode 1).
 * segs is a HOST array of nseg <= 8 entries; one launch. */
typedef struct m3_pack_seg { const void *src; int64_t dst_off; int64_t nbytes; int32_t mode; int32_t reserved; } m3_pack_seg;
int m3_pack_fields(void *dst, const m3_pack_seg *segs, int nseg, void *stream);
/* k = s transposed-conv GEMM output [B*h*w, s*s*C] -> NHWC [B,h*s,w*s,Cpad] (first C channels). */
int m3_unshuffle_bf16(const void *in, void *out, int B, int h, int w, int s, int C, int Cpad, void *stream);
/* bilinear x2, align_corners = True, NHWC bf16. */
int m3_upsample2x_bf16(const void *in, void *out, int B, int H, int W, int C, void *stream);
int m3_upsample2x_dt(const void *in, void *out, int B, int H, int W, int C, int dtype, void *stream);
/* DPT fusion block (public DPT FeatureFusionBlock: upsample the coarser path, add the refined skip connection):
 * out [B,OH,OW,C] = bilinear_x2(low [B,H,W,C], align_corners)[:, :OH, :OW] + y [B,OH,OW,C], OH <= 2H, OW <= 2W,
 * summed in fp32 and rounded once.  out may alias y. */
int m3_add_upsample2x_dt(const void *low, const void *y, void *out, int B, int H, int W, int OH, int OW, int C,
                         int dtype, void *stream);
/* DPT output [P,4] f32 -> pts3d [P,3] = xyz/|xyz| * expm1(|xyz|), conf [P] = 1 + exp(c). */
int m3_pts_post(const float *in, float *pts, float *conf, int64_t P, void *stream);
/* feature-head output [B*(H/16)*(W/16), 6400] bf16 -> pixel shuffle 16 -> desc [B,H,W,24] f32
 * (L2-normalised), desc_conf [B,H,W] = exp(channel 24). */
int m3_desc_post(const void *in, float *desc, float *dconf, int B, int H, int W, void *stream);
int m3_desc_post_dt(const void *in, float *desc, float *dconf, int B, int H, int W, int dtype, void *stream);
/* Same with the descriptors stored as IEEE half [B,H,W,24] ("fp16 features", BASELINE configs[4]): the fp32 value
 * rounded once to nearest-even.  desc_conf stays fp32. */
int m3_desc_post_f16(const void *in, void *desc_f16, float *dconf, int B, int H, int W, int dtype, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* M3SLAM_MODEL_H */
