// Shared pieces of the bf16 MFMA GEMM kernels (gemm.hip: 128x128 tile, gemm256.hip: 256x256 ping-pong).
#pragma once
#include "common.h"
#include "../../include/m3slam_model.h"

namespace m3gemm {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;

constexpr int BK = 64;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {        // round-to-nearest-even, NaN preserved
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
// GELU (erf form), two elements per lane at a time so the chain maps to v_pk_mul/v_pk_fma_f32.
// erf(z) = z * P(z^2) on |z| <= 3 (clamped; 1 - erf(3) = 2.2e-5), P = degree-8 minimax fit, max abs
// error 2.5e-5 in erf, 6e-5 in gelu(x) (at |x| ~ 4.2 where one bf16 ulp is 1.6e-2) - far below the
// 2^-9 rounding of the bf16 output.  No transcendental: ~8 VALU issues per element instead of the
// 16 + rcp + exp of the Abramowitz-Stegun 7.1.26 form used first (~4 us per 256x256 tile in the fc1 epilogue).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
    f32x2 z = x * 0.70710678118654752f;
    z.x = __builtin_amdgcn_fmed3f(z.x, -3.0f, 3.0f);
    z.y = __builtin_amdgcn_fmed3f(z.y, -3.0f, 3.0f);
    const f32x2 u = z * z;
    f32x2 p = (f32x2)(4.074053805e-08f);
    p = __builtin_elementwise_fma(p, u, (f32x2)(-1.944763426e-06f));
    p = __builtin_elementwise_fma(p, u, (f32x2)(4.105960033e-05f));
    p = __builtin_elementwise_fma(p, u, (f32x2)(-5.110292695e-04f));
    p = __builtin_elementwise_fma(p, u, (f32x2)(4.235391971e-03f));
    p = __builtin_elementwise_fma(p, u, (f32x2)(-2.510276809e-02f));
    p = __builtin_elementwise_fma(p, u, (f32x2)(1.110792086e-01f));
    p = __builtin_elementwise_fma(p, u, (f32x2)(-3.753148019e-01f));
    p = __builtin_elementwise_fma(p, u, (f32x2)(1.128268361e+00f));
    const f32x2 hx = x * 0.5f;
    return __builtin_elementwise_fma(hx, z * p, hx);
}

// two fp32 -> packed bf16x2 (round to nearest even), lo in bits 0..15: one v_cvt_pk_bf16_f32.  Written as a vector
// conversion, NOT as inline asm: the result of a transcendental (v_exp_f32, v_rcp_f32, v_rsq_f32 ...) may not be
// read by the next VALU instruction without a wait state, and the hazard recogniser cannot see into an asm
// statement - an asm conversion placed right behind the softmax's v_exp_f32 read stale registers in some lanes,
// depending on timing (round 2, found when the row-sum adds that used to separate the two were removed).
typedef __bf16 m3_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 m3_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, m3_bf16x2));
}

// ---- 16-bit storage type of a launch: DT = 0 bf16 (v_mfma_f32_16x16x32_bf16), DT = 1 IEEE fp16
// (v_mfma_f32_16x16x32_f16, same issue rate).  Operands, 16-bit outputs and 16-bit residuals of one
// launch share the type; accumulation is fp32 either way.
enum { DT_BF16 = 0, DT_F16 = 1 };
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <int DT> __device__ __forceinline__ unsigned pack16(float lo, float hi) {
    if constexpr (DT == DT_BF16) return pack_bf16(lo, hi);
    else return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, m3_f16x2));   // round to nearest even
}
template <int DT> __device__ __forceinline__ float lo16(unsigned q) {   // low half of a packed pair -> fp32
    if constexpr (DT == DT_BF16) return __uint_as_float(q << 16);
    else return (float)__builtin_bit_cast(_Float16, (unsigned short)(q & 0xffffu));
}
template <int DT> __device__ __forceinline__ float hi16(unsigned q) {
    if constexpr (DT == DT_BF16) return __uint_as_float(q & 0xffff0000u);
    else return (float)__builtin_bit_cast(_Float16, (unsigned short)(q >> 16));
}
template <int DT> __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (DT == DT_BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// ReLU on a fragment of eight 16-bit floats (bf16 or fp16): a negative value has its sign bit set, i.e. is negative as a
// signed 16-bit integer too, and every non-negative value keeps its bits under max(x, 0)
typedef __attribute__((ext_vector_type(8))) short s16x8;
__device__ __forceinline__ bf16x8 relu_frag(bf16x8 v) {
    return __builtin_elementwise_max(v, (bf16x8)(short)0);
}

__device__ __forceinline__ void glds16(const void *g, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned *)g,
                                     (__attribute__((address_space(3))) unsigned *)lds_wave_base, 16, 0, 0);
}

// Workgroup barrier that LDS traffic cannot cross.  For LLVM `__builtin_amdgcn_s_barrier()` touches no memory, so
// the scheduler is free to move ds_reads - and the s_waitcnt that retires them - across it: in the round-1 build the
// end-of-K-tile barrier of k_gemm<.., 64> and of k_attn was issued BEFORE the wave's last fragment reads had
// returned, so a faster wave could re-stage the buffer (LDS-DMA of tile t+2) under them.  Seen as 1-3 wrong 64x64
// tiles per 1000 (one K-tile's rows of one wave stale) at 4-5 workgroups per CU, never at 1-2.  This form pins the
// order: everything before (incl. the MFMAs that consume the reads) stays before, own LDS ops are retired, then
// the barrier, and nothing after moves up.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

struct GemmArgs {
    const bf16_t *A;        // [M,K] bf16 (dense) or NHWC image (conv)
    const bf16_t *W;        // [N,K] bf16
    const float *bias;      // [N] or null
    void *C;                // bf16 [M,ldc] or f32 [M,ldc]
    const void *R;          // residual (same dtype/shape as C) or null
    const bf16_t *zero16;   // 16 zero bytes (conv padding source)
    int M, N, K, ldc;
    // conv geometry
    int H, Wd, Cin, OH, OW, stride;
    // fused RoPE-2D epilogue (EPI_BF16_ROPE): columns < rope_cols are 64-wide heads to rotate
    const float *rope_tok;  // [tokens_per_image, 2 (y|x), 2 (cos|sin), 16]; or, position mode:
    const int *rope_pos;    // [tokens_per_image, 2 (y|x)] grid positions - cos/sin are then computed in the epilogue
    float rope_log2_base;   //   from frequencies base^(-i/16), i = 0..15 (v_sin_f32 / v_cos_f32, arguments in revolutions)
    int tokens_per_image, rope_cols;
    int ln_gsz, ln_tops;    // consumer: stored slots per top node (1 .. 4) and top nodes (<= 4)
    int rope_pmax;          //   position mode: > 0 = every position is in [0, rope_pmax) (<= kRopeTableRows): the kernels then build
                            //   the cos / sin of all rope_pmax x 16 (position, frequency) pairs ONCE per workgroup in LDS
    int q_cols;             // columns < q_cols (the q heads) are multiplied by q_scale after the rotation, before the
    float q_scale;          // 16-bit rounding: softmax scale * log2(e) folded into q (attention then needs no per-score FMA)
    // grouped launch (blockIdx.y = group): group 1 uses W2/bias2 and A/C/R advanced by the strides
    const bf16_t *W2;
    const float *bias2;
    long long a_gstride, c_gstride;   // elements of A, elements of C/R
    int groups;
    // split-K (128x128 kernel, EPI_F32 only): blockIdx.z = split, C = fp32 partials [splits][M][ldc]
    int splits;
    int slices;             // > 1 (conv, 128 x 128 kernel): ONE workgroup walks all K slices of the split-K rule and adds the slice
                            //   sums in the finishing kernel's order (same bits as `splits` partial planes + k_splitk_finish)
    // EPI_RELU_HEAD4: W2 = bf16 [4][N] projection, bias2 = its 4 biases, C = pts f32 [M,3], C2 = conf f32 [M]
    float *C2;
    int dt;                 // DT_BF16 / DT_F16: 16-bit storage type of A, W and of 16-bit C / R
    int relu_a;             // conv3x3: ReLU applied to the INPUT fragments (max with 0 on the 16-bit lanes, read as signed
                            //   integers: one v_pk_max_i16 per dword for bf16 and fp16 alike) - the DPT residual unit's
                            //   relu(x) -> conv1 without materialising relu(x)  (M3_EPI_INPUT_RELU)
    int v_bf16;             // EPI_BF16_ROPE in a DT_F16 launch: the columns >= rope_cols (v of a q|k|v or k|v projection) are
                            //   stored as bf16 - the operand type of the fast attention loop's P.V product (M3_DT_F16_PVBF16)
    // ---- LayerNorm folded into the GEMMs on both sides of it (m3_gemm_ex; fp16 trunk) ----
    // producer (EPI_F32 / EPI_F32_ACCUM): besides the fp32 stream x' the launch writes C16 = x' rounded to the launch's 16-bit
    // type (same ldc) and, per row and statistics slot, (sum x', sum x'^2) into stats_out [N / stats_w][M][2] - from the fp32 values
    void *C16;
    float *stats_out;
    int stats_w;            // producer: columns per stored statistics slot - 64, 128, 192 or 256: a node of the rows' canonical sum
                            // tree that the launch's tile width is a multiple of (the workgroup adds its waves' 32-column leaves:
                            // stats_tile_finalize).  One slot per tile where the tile is a top node (256-row kernel, BN = 256 / 192)
    // hi / lo form of the stream (C_lo set; fp16 launches): the stream is kept as TWO 16-bit planes, x = hi + lo with hi = x
    // rounded to fp16 and lo = the rounded remainder (22 significant bits; |x| < 65504).  hi IS the consumer's operand, so the
    // residual launch moves 4 + 4 bytes per element as with an fp32 stream and no separate copy exists: R = hi in, R_lo = lo in,
    // C16 = hi out, C_lo = lo out (in place allowed), C is not written.
    const void *R_lo;
    void *C_lo;
    // consumer (16-bit epilogues): A is such a copy of the RAW stream and W carries gamma; the epilogue turns the product into
    // the product with the normalised row: rstd[m] * (acc - mean[m] * ln_colsum[n]) (+ bias, which carries beta . W^T);
    // mean / rstd of row m come from ln_stats [M][ln_slots][2], ln_slots = K / 32 (summed in ONE order by every kernel)
    const float *ln_stats;
    const float *ln_colsum, *ln_colsum2;       // [N] sums over k of the 16-bit weights (group 0 / group 1)
    int ln_slots;
    float ln_eps;
    long long ln_gstride, stats_gstride;       // floats between group 0's and group 1's statistics (may be negative)
};

// Per-group view of the arguments (group 1 of a 2-group launch).
template <int EPI>
__device__ __forceinline__ GemmArgs select_group(const GemmArgs &in, int grp) {
    GemmArgs g = in;
    if (grp == 1) {
        constexpr long long esz = (EPI == 2 /*EPI_F32*/ || EPI == 3 /*EPI_F32_ACCUM*/) ? 4 : 2;
        g.A = in.A + in.a_gstride;
        g.W = in.W2;
        g.bias = in.bias2;
        g.C = reinterpret_cast<unsigned char *>(in.C) + in.c_gstride * esz;
        if (in.R) g.R = reinterpret_cast<const unsigned char *>(in.R) + in.c_gstride * esz;
        if (in.C16) g.C16 = reinterpret_cast<unsigned char *>(in.C16) + in.c_gstride * 2;
        if (in.C_lo) {                                          // hi / lo stream: 16-bit planes, C / R are not fp32 tensors
            g.C_lo = reinterpret_cast<unsigned char *>(in.C_lo) + in.c_gstride * 2;
            if (in.R) g.R = reinterpret_cast<const unsigned char *>(in.R) + in.c_gstride * 2;
            if (in.R_lo) g.R_lo = reinterpret_cast<const unsigned char *>(in.R_lo) + in.c_gstride * 2;
        }
        if (in.stats_out) g.stats_out = in.stats_out + in.stats_gstride;
        if (in.ln_stats) { g.ln_stats = in.ln_stats + in.ln_gstride; g.ln_colsum = in.ln_colsum2; }
    }
    return g;
}

enum { EPI_BF16 = 0, EPI_BF16_GELU = 1, EPI_F32 = 2, EPI_F32_ACCUM = 3, EPI_BF16_RELU = 4, EPI_BF16_ADD = 5,
       EPI_BF16_ROPE = 6,
       EPI_RELU_HEAD4 = 7 /* internal: relu -> bf16 -> 1x1 projection to 4 channels -> pointmap post-processing */ };


// Epilogue for one 16x16 accumulator tile: the lane holds C[m][n..n+3] (operands were swapped).
template <int EPI, int DT = DT_BF16>
__device__ __forceinline__ void store_tile(const GemmArgs &g, f32x4 v, int m, int n) {
    if (m >= g.M || n >= g.N) return;                       // N is a multiple of 4 (checked on the host)
    if (EPI != EPI_BF16_ROPE && g.bias) {
        const float4 b = *reinterpret_cast<const float4 *>(g.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    const size_t off = (size_t)m * g.ldc + n;
    if (EPI == EPI_F32 || EPI == EPI_F32_ACCUM) {
        float *C = reinterpret_cast<float *>(g.C) + off;
        if (EPI == EPI_F32_ACCUM) {
            const float4 r = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(g.R) + off);
            v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        }
        *reinterpret_cast<float4 *>(C) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        if (EPI == EPI_BF16_GELU) {
            const f32x2 lo = gelu_erf2(f32x2{v[0], v[1]}), hi = gelu_erf2(f32x2{v[2], v[3]});
            v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
        }
        if (EPI == EPI_BF16_ADD) {
            const uint2 r = *reinterpret_cast<const uint2 *>(reinterpret_cast<const bf16_t *>(g.R) + off);
            v[0] += lo16<DT>(r.x); v[1] += hi16<DT>(r.x); v[2] += lo16<DT>(r.y); v[3] += hi16<DT>(r.y);
        }
        if (EPI == EPI_BF16_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        uint2 o;
        if (EPI == EPI_BF16_ROPE && DT == DT_F16 && g.v_bf16 && n >= g.rope_cols) {
            o.x = pack16<DT_BF16>(v[0], v[1]); o.y = pack16<DT_BF16>(v[2], v[3]);
        } else {
            o.x = pack16<DT>(v[0], v[1]); o.y = pack16<DT>(v[2], v[3]);
        }
        *reinterpret_cast<uint2 *>(reinterpret_cast<bf16_t *>(g.C) + off) = o;
    }
}

// Fused RoPE-2D on 16-row x (16*NJ)-column accumulator strips that start on a 32-column boundary:
// t[0..NJ-1] are the 16-column tiles, lane holds columns 16*j + 4*(lane>>4) + e.  Every 32-column
// block is half a head: even blocks rotate with the token's y, odd blocks with x; element i pairs
// with i+16, i.e. tile 2b with tile 2b+1 in the SAME lane and register.  Bias is added first.
// Coefficients come from a PER-TOKEN table rope_tok[tokens_per_image][2 (y|x)][2 (cos|sin)][16 freq] f32
// (256 B per token, L2-resident): two independent 16-byte reads per (row, block), no position lookup in
// front of it and no branch around it - the first version (position table -> cos/sin table, both behind
// per-row branches) made every row a dependent chain of two L2 round trips: 128 us instead of 93 us for
// the 16384 x 3072 x 1024 projection.
struct RopeCoef { float4 c, s; };               // cos / sin of frequencies fi .. fi+3
// Position mode (g.rope_pos): the table read is 64 B of cos/sin per 64 B of output, fetched as 16 rows x 64 B pieces by
// every wave that shares a row - by ablation ~10 us of a 105 us projection, while the rotation arithmetic itself is free.
// One 4-byte position per (row, axis) and eight transcendentals per 32-column block replace it: angle in revolutions =
// pos * base^(-i/16) / 2pi (at most ~10 for a 64 x 64 token grid; v_sin_f32's domain is +-256), absolute error ~1e-6.
struct RopeFreq { float rev[4]; };              // this lane's four frequencies (columns fi..fi+3 of a 16-wide group) / 2pi
__device__ __forceinline__ RopeFreq rope_freqs(const GemmArgs &g, int lane) {
    RopeFreq f;
    const int fi = (lane >> 4) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) f.rev[k] = exp2f(-(float)(fi + k) * (g.rope_log2_base * (1.0f / 16.0f))) * 0.15915494309189535f;
    return f;
}
// LDS table of the rotation coefficients, position mode: rows of 32 floats, row p = cos(p f_i), i = 0..15 | sin(p f_i), i = 0..15
// - the SAME expressions as the per-lane computation in rope_load (same bits).  Per 32-column block and row a lane then reads
// two 16-byte LDS words instead of issuing eight transcendentals: in the 256 x 256 tile a lane owns 32 rows x 2 blocks = 512
// v_sin / v_cos (quarter rate: ~16 cycles each per wave), 6.8 us per tile with two waves per SIMD - all of what the fused
// rotation cost per launch (12.6 us of the 16384 x 3072 x 1024 projection); and every wave column / N-tile recomputed the same
// 64 values per token.  Built after the K loop in stage memory the epilogue scratch does not reach; the caller synchronises.
constexpr int kRopeTableRows = 64;                  // positions 0..63: images up to 1024 x 1024 pixels (8 KiB)
__device__ __forceinline__ void rope_table_build(const GemmArgs &g, float *tab, int tid, int nthreads) {
    for (int e = tid; e < g.rope_pmax * 16; e += nthreads) {
        const int p = e >> 4, f = e & 15;
        const float rev = exp2f(-(float)f * (g.rope_log2_base * (1.0f / 16.0f))) * 0.15915494309189535f;
        const float ang = (float)p * rev;
        tab[p * 32 + f] = __builtin_amdgcn_cosf(ang);
        tab[p * 32 + 16 + f] = __builtin_amdgcn_sinf(ang);
    }
}
// token grid position (y, x) of row m: ONE 8-byte load, issued by the caller for all rows of a pass before any of them is used
// (read inside rope_load, per (row, block), each 4-byte load was waited for where it stood: eight dependent round trips per
// pass - most of what the fused rotation cost per tile, found in round 5 by reading the ISA)
__device__ __forceinline__ int2 rope_pos_of(const GemmArgs &g, int m) {
    const int mm = m < g.M ? m : g.M - 1;
    return *reinterpret_cast<const int2 *>(g.rope_pos + (size_t)(mm % g.tokens_per_image) * 2);
}
template <int NJ>
__device__ __forceinline__ void rope_load(const GemmArgs &g, RopeCoef (&cf)[NJ / 2], int m, int n_base, int lane,
                                          const RopeFreq &fr, int2 pyx = int2{0, 0}, const float *ropet = nullptr) {
    const int mm = m < g.M ? m : g.M - 1;
    const int tok = mm % g.tokens_per_image;
    const int fi = (lane >> 4) * 4;
#pragma unroll
    for (int blk = 0; blk < NJ / 2; ++blk) {
        if (n_base + 32 * blk >= g.rope_cols) continue;                 // wave-uniform: v columns are not rotated
        const int axis = ((n_base + 32 * blk) >> 5) & 1;                 // 0: y, 1: x
        if (ropet) {                                                     // kernel-uniform: the workgroup's LDS table
            int p = axis ? pyx.y : pyx.x;
            p = p < g.rope_pmax ? p : g.rope_pmax - 1;                   // (a position outside the promised range: no stray read)
            const float *row = ropet + p * 32 + fi;
            cf[blk].c = *reinterpret_cast<const float4 *>(row);
            cf[blk].s = *reinterpret_cast<const float4 *>(row + 16);
        } else if (g.rope_pos) {                                         // kernel-uniform
            const float pos = (float)(axis ? pyx.y : pyx.x);
            cf[blk].c = make_float4(__builtin_amdgcn_cosf(pos * fr.rev[0]), __builtin_amdgcn_cosf(pos * fr.rev[1]),
                                    __builtin_amdgcn_cosf(pos * fr.rev[2]), __builtin_amdgcn_cosf(pos * fr.rev[3]));
            cf[blk].s = make_float4(__builtin_amdgcn_sinf(pos * fr.rev[0]), __builtin_amdgcn_sinf(pos * fr.rev[1]),
                                    __builtin_amdgcn_sinf(pos * fr.rev[2]), __builtin_amdgcn_sinf(pos * fr.rev[3]));
        } else {
            const float *row = g.rope_tok + (size_t)(tok * 2 + axis) * 32 + fi;
            cf[blk].c = *reinterpret_cast<const float4 *>(row);
            cf[blk].s = *reinterpret_cast<const float4 *>(row + 16);
        }
    }
}
// (bias add +) rotation on packed pairs: 8 v_pk_* per 32-column block instead of 16 + 16 scalar ops
template <int NJ>
__device__ __forceinline__ void rope_apply(const GemmArgs &g, f32x4 *t, const RopeCoef (&cf)[NJ / 2], const float4 *bj,
                                           int n_base) {
#pragma unroll
    for (int blk = 0; blk < NJ / 2; ++blk) {
        f32x4 &a = t[2 * blk], &b = t[2 * blk + 1];
        const float4 ba = bj[2 * blk], bb = bj[2 * blk + 1];
        const f32x2 a0 = f32x2{a[0], a[1]} + f32x2{ba.x, ba.y}, a1 = f32x2{a[2], a[3]} + f32x2{ba.z, ba.w};
        const f32x2 b0 = f32x2{b[0], b[1]} + f32x2{bb.x, bb.y}, b1 = f32x2{b[2], b[3]} + f32x2{bb.z, bb.w};
        if (n_base + 32 * blk >= g.rope_cols) {                         // wave-uniform
            a = f32x4{a0.x, a0.y, a1.x, a1.y}; b = f32x4{b0.x, b0.y, b1.x, b1.y};
            continue;
        }
        const f32x2 c0 = {cf[blk].c.x, cf[blk].c.y}, c1 = {cf[blk].c.z, cf[blk].c.w};
        const f32x2 s0 = {cf[blk].s.x, cf[blk].s.y}, s1 = {cf[blk].s.z, cf[blk].s.w};
        f32x2 ra0 = __builtin_elementwise_fma(a0, c0, -(b0 * s0)), ra1 = __builtin_elementwise_fma(a1, c1, -(b1 * s1));
        f32x2 rb0 = __builtin_elementwise_fma(b0, c0, a0 * s0), rb1 = __builtin_elementwise_fma(b1, c1, a1 * s1);
        if (n_base + 32 * blk < g.q_cols) {                             // wave-uniform
            const f32x2 qs = {g.q_scale, g.q_scale};
            ra0 = ra0 * qs; ra1 = ra1 * qs; rb0 = rb0 * qs; rb1 = rb1 * qs;
        }
        a = f32x4{ra0.x, ra0.y, ra1.x, ra1.y}; b = f32x4{rb0.x, rb0.y, rb1.x, rb1.y};
    }
}
// unaligned fallback: bias + rotation of one strip in place
template <int NJ>
__device__ __forceinline__ void rope_strip(const GemmArgs &g, f32x4 *t, int m, int n_base, int lane) {
    if (m >= g.M) return;
    const int fi = (lane >> 4) * 4;
    float4 bj[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        bj[j] = (g.bias && n_base + j * 16 + fi < g.N) ? *reinterpret_cast<const float4 *>(g.bias + n_base + j * 16 + fi)
                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
    RopeCoef cf[NJ / 2];
    rope_load<NJ>(g, cf, m, n_base, lane, rope_freqs(g, lane), g.rope_pos ? rope_pos_of(g, m) : int2{0, 0});
    rope_apply<NJ>(g, t, cf, bj, n_base);
}

// Sum over the aligned group of 8 lanes a lane belongs to, result in every lane, on the DPP path (three v_add_f32 with a lane
// permutation; __shfl_xor compiles to ds_bpermute - an LDS round trip per step).  xor 1, xor 2 inside the quad, then the
// half-row mirror (lane i <-> 7 - i) pairs the two quads: every lane adds the same two quad sums, so all 8 hold the same bits.
__device__ __forceinline__ float dpp_sum8(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    return v;
}
// value of the lane's quad neighbour lane ^ 1
__device__ __forceinline__ unsigned dpp_xor1(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}

// The statistics' canonical sum.  A row's (sum, sum of squares) over its C columns is defined as ONE expression tree, whatever
// kernel and tile shape produce or consume the pieces - so a row's statistics (and with them the network's output) do not depend
// on the batch a pair is computed in:
//   leaves      32 columns (8 lanes x 4 columns, dpp_sum8 in the producer's epilogue)
//   top nodes   192 columns (C % 192 == 0, e.g. 768) or else 256 (C % 256 == 0): ((l0 + l1) + (l2 + l3)) + ((l4 + l5) + (l6 + l7)), the last
//               pair absent (zero) for 192; its left-aligned subtrees are the 64-column pairs p_j = l_2j + l_2j+1 and (for 256)
//               the 128-column halves (p0 + p1), (p2 + p3)
//   total       ((t0 + t1) + t2) + t3 over the <= 4 top nodes (absent ones zero; x + 0 = x keeps the bits)
// A producer stores the widest node its tile width is a multiple of - pairs (64-column tiles, and 128-wide tiles of a 192-family
// stream), halves (128-wide tiles), whole top nodes (256-row kernel with BN = 256 / 192: one slot per tile) - slot-major
// [C / w][M][2]; a consumer builds each top node as (u0 + u1) + (u2 + u3) from its stored slots (absent ones zero) - the same
// expression whatever the level.  A consumer pays for statistics bytes what it pays for operand bytes
// (profiles/r05_fold_probe.md): 2 KiB per 256-row tile and slot.
__device__ __forceinline__ float ln_tree8(const float (&x)[8]) {
    return ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
}
// LayerNorm fold, consumer side: mean and 1 / sqrt(var + eps) of the tile's rows.  FOUR threads per PAIR of rows (the (sum, sum
// of squares) of two neighbouring rows of a slot are one 16-byte load): thread (pair p, q) builds top node q of rows 2p, 2p + 1
// from its ln_gsz stored slots, the quad then folds the nodes in index order on the DPP path, every lane the same expression.
// M even (host check); rows past M read the last pair.  Returns (mean, rstd) of row 2p in .x .y and of row 2p + 1 in .z .w, in
// all four lanes of the quad; threads >= 2 ROWS repeat pair 0's work (ln_table_store ignores them).
struct LnLoads { float4 v[4]; };
// (mean, 1 / sqrt(var + eps)) of a row from its sum and sum of squares over `cols` columns, in ONE spelled-out operation
// sequence, so each kernel that builds a row table gets the same bits
__device__ __forceinline__ float2 ln_mean_rstd(float s, float q, int cols, float eps) {
    const float inv = 1.0f / (float)cols;
    const float mean = s * inv;
    const float var = fmaxf(__builtin_fmaf(-mean, mean, q * inv), 0.f);      // the fma spelled out: hipcc contracts a * b - c * d
    return make_float2(mean, rsqrtf(var + eps));                             // as it likes per call site (__fmul_rn does not stop it)
}
// first half: the thread's loads go out (and stay in flight: a kernel puts its first K-tile's loads between the two halves).
// UNCONDITIONAL loads from clamped addresses, the bound applied to the value in the second half (a bound on the load makes hipcc
// branch around each one and wait for it where it stands)
template <int ROWS>
__device__ __forceinline__ LnLoads ln_row_issue(const GemmArgs &g, int m0, int tid) {
    LnLoads L;
    const int pair = (tid < 2 * ROWS ? tid : 0) >> 2, q = tid & 3;
    const int node = q < g.ln_tops ? q : g.ln_tops - 1;
    int m = m0 + 2 * pair;
    m = m < g.M - 1 ? m : g.M - 2;
    const float4 *p = reinterpret_cast<const float4 *>(g.ln_stats + ((size_t)node * g.ln_gsz * g.M + m) * 2);
    const size_t slot_stride = (size_t)g.M / 2;              // float4 units between consecutive slots
#pragma unroll
    for (int i = 0; i < 4; ++i) L.v[i] = p[(size_t)(i < g.ln_gsz ? i : g.ln_gsz - 1) * slot_stride];
    return L;
}
template <int ROWS>
__device__ __forceinline__ float4 ln_row_finish(const GemmArgs &g, const LnLoads &L, int m0, int tid) {
    const bool live = (tid & 3) < g.ln_tops;
    float4 u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool on = live && i < g.ln_gsz;
        u[i] = make_float4(on ? L.v[i].x : 0.f, on ? L.v[i].y : 0.f, on ? L.v[i].z : 0.f, on ? L.v[i].w : 0.f);
    }
    const float node[4] = {(u[0].x + u[1].x) + (u[2].x + u[3].x), (u[0].y + u[1].y) + (u[2].y + u[3].y),
                           (u[0].z + u[1].z) + (u[2].z + u[3].z), (u[0].w + u[1].w) + (u[2].w + u[3].w)};
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int b = __builtin_bit_cast(int, node[k]);      // the quad's four top nodes, folded in index order by every lane
        const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, b, 0x00, 0xF, 0xF, true));   // quad_perm [0,0,0,0]
        const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, b, 0x55, 0xF, 0xF, true));   // [1,1,1,1]
        const float t2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, b, 0xAA, 0xF, 0xF, true));   // [2,2,2,2]
        const float t3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, b, 0xFF, 0xF, 0xF, true));   // [3,3,3,3]
        t[k] = ((t0 + t1) + t2) + t3;
    }
    const float2 a = ln_mean_rstd(t[0], t[1], g.K, g.ln_eps), b2 = ln_mean_rstd(t[2], t[3], g.K, g.ln_eps);
    return make_float4(a.x, a.y, b2.x, b2.y);
}
template <int ROWS>
__device__ __forceinline__ float4 ln_row_stats(const GemmArgs &g, int m0, int tid) {
    return ln_row_finish<ROWS>(g, ln_row_issue<ROWS>(g, m0, tid), m0, tid);
}
// byte offset of the statistics staging inside a wave's epilogue scratch (behind the 16-row fp32 transpose image)
template <int NJ> constexpr int stats_stage_offset() { return (16 * ((NJ == 4) ? 256 : NJ * 64 + 16) + 15) & ~15; }
// Producer: after the waves' epilogues (and a workgroup barrier) the 32-column leaves of the whole tile sit in the waves' scratch,
// [leaf of the wave][row of the wave] float2 behind each transpose image.  Thread t < TROWS takes row t: for every slot of
// g.stats_w columns inside the tile it adds the slot's leaves in the canonical tree (left-aligned, absent leaves zero) and stores
// it; consecutive threads store consecutive rows of a slot (8 B each).
template <int TROWS, int NI, int NJ, int WN>
__device__ __forceinline__ void stats_tile_finalize(const GemmArgs &g, const unsigned char *lds, int wave_stride, int wst_off,
                                                    int m0, int n0, int tid) {
    constexpr int WROWS = 16 * NI, LPW = NJ / 2, LEAVES = WN * LPW;     // rows per wave, leaves per wave, leaves per tile
    if (tid >= TROWS) return;
    const int wr = tid / WROWS, rr = tid - wr * WROWS;
    const int lpu = g.stats_w >> 5;                          // leaves per slot: 2, 4, 6 or 8
    const int m = m0 + tid;
    for (int k = 0; k * lpu < LEAVES; ++k) {
        float s[8], q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool on = i < lpu;
            const int l = k * lpu + (on ? i : 0);
            const int wc = l / LPW, sl = l - wc * LPW;
            const float2 v = *reinterpret_cast<const float2 *>(lds + (wr * WN + wc) * wave_stride + wst_off + (sl * WROWS + rr) * 8);
            s[i] = on ? v.x : 0.f; q[i] = on ? v.y : 0.f;
        }
        const int n = n0 + k * g.stats_w;
        if (m < g.M && n < g.N)
            *reinterpret_cast<float2 *>(g.stats_out + ((size_t)(n / g.stats_w) * g.M + m) * 2) = make_float2(ln_tree8(s), ln_tree8(q));
    }
}
template <int ROWS>
__device__ __forceinline__ void ln_table_store(float2 *tab, const float4 &mr, int tid) {
    if (tid < 2 * ROWS && (tid & 3) == 0) {
        tab[2 * (tid >> 2)] = make_float2(mr.x, mr.y);
        tab[2 * (tid >> 2) + 1] = make_float2(mr.z, mr.w);
    }
}
template <int ROWS>
__device__ __forceinline__ void ln_row_table(const GemmArgs &g, float2 *tab, int m0, int tid) {
    ln_table_store<ROWS>(tab, ln_row_stats<ROWS>(g, m0, tid), tid);
}

// ---------------------------------------------------------------------------------------------
// Row-contiguous epilogue through LDS.
//
// In the accumulator layout a store instruction of one 16x16 tile touches 16 rows x 32 bytes: the
// write path is then limited by the number of partial-line requests, not by bytes (measured with
// s_memtime on 16384x3072x1024: 13.8 us of store issue per 256x256 tile against a 22 us K loop,
// 24 us with the fp32 residual read-modify-write).  After the K loop the operand stages are dead,
// so every wave transposes its own sub-tile through a private LDS scratch (padded rows, no
// workgroup barrier needed) and writes/reads global memory 16 bytes per lane with 8 (bf16) or 16
// (fp32) consecutive lanes per row: full 128-byte lines, 2x fewer instructions.
//   wave sub-tile = (16*NI) rows x (16*NJ) columns at (m_base, n_base); wlds = this wave's scratch,
//   64 * (32*NJ + 16) bytes (9216 for NJ = 4, 13312 for NJ = 6).
// Bias / activation / RoPE are applied in the accumulator layout before the transpose, residuals
// (fp32 accumulate, bf16 add) on the row-contiguous side.  Falls back to store_tile when the
// output is not 16-byte aligned.
// TPMAX / RESID_AHEAD trade registers for latency hiding (k_gemm_duo lives on 128 registers and has a second
// workgroup on the CU to cover the epilogue's round trips): TPMAX caps the accumulator row tiles per pass of the
// 16-bit path (RoPE keeps TP x NJ/2 coefficient sets live), RESID_AHEAD = false loads a pass's residual tile in the
// pass itself instead of one pass ahead (one register set instead of two).
// ALIGNED = true: the host has checked the alignment conditions below, the element-wise fallback is not compiled in.
template <int EPI, int NI, int NJ = 4, int DT = DT_BF16, int TPMAX = 4, bool RESID_AHEAD = true, bool ALIGNED = false>
__device__ __forceinline__ void epilogue_rows(const GemmArgs &g, f32x4 (&acc)[NI][NJ], unsigned char *wlds,
                                              int m_base, int n_base, int lane, const float2 *lnt = nullptr, int lrow = 0,
                                              const float *ropet = nullptr) {
    constexpr bool F32OUT = (EPI == EPI_F32 || EPI == EPI_F32_ACCUM);
    constexpr bool F32LDS = F32OUT || EPI == EPI_BF16_ADD;        // keep one rounding for the bf16 residual add
    const int r = lane & 15, gq = lane >> 4;
    const bool aligned = ((reinterpret_cast<size_t>(g.C) & 15) == 0) && (g.ldc % (F32OUT ? 4 : 8) == 0) &&
                         (F32OUT || g.N % 8 == 0) &&
                         (!(EPI == EPI_F32_ACCUM || EPI == EPI_BF16_ADD) || (reinterpret_cast<size_t>(g.R) & 15) == 0);
    if (!ALIGNED && !aligned) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (EPI == EPI_BF16_ROPE) rope_strip<NJ>(g, acc[i], m_base + i * 16 + r, n_base, lane);
#pragma unroll
            for (int j = 0; j < NJ; ++j) store_tile<EPI, DT>(g, acc[i][j], m_base + i * 16 + r, n_base + j * 16 + gq * 4);
        }
        return;
    }
    // Bias of this lane's columns: UNCONDITIONAL loads from clamped addresses under one kernel-uniform test, the column bound
    // applied to the values.  Written as `(bias && n < N) ? load : 0` hipcc branched around every load and waited for each in
    // turn (s_and_saveexec / global_load / s_waitcnt vmcnt(0), four times): four dependent L2 round trips in front of every
    // tile's epilogue since round 2 (found in round 5 by reading the ISA for the LayerNorm-fold column sums, same pattern).
    float4 bj[NJ];
    if (g.bias) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n_base + j * 16 + gq * 4;
            bj[j] = *reinterpret_cast<const float4 *>(g.bias + (n < g.N ? n : g.N - 4));
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (n_base + j * 16 + gq * 4 >= g.N) bj[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) bj[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (F32LDS) {
        // fp32 output of 64-column sub-tiles: unpadded 256-byte rows, 16-byte chunk c of row r at chunk c ^ (r & 7) - the
        // 8 consecutive lanes (= 8 rows at one column quad) a ds_write_b128 is served in hit 8 different chunks (32 banks),
        // and the 16-lane groups of the ds_read_b128 (half a row + the complementary half of the next) every bank once.
        // (272-byte padded rows before: conflict-free writes, 2-way reads.)
        constexpr bool SWZ32 = F32OUT && (NJ == 4);
        constexpr int RS = SWZ32 ? 256 : NJ * 64 + 16;            // 16*NJ fp32 (+ 16 bytes of padding)
        constexpr bool RESID = (EPI == EPI_F32_ACCUM || EPI == EPI_BF16_ADD);
        constexpr int PT = 1;      // ONE accumulator row tile per pass: the statistics staging below needs the room behind a 16-row
                                   // transpose image inside the wave's scratch (two-tile passes of EPI_F32 overran it), and residual tiles are
        constexpr int PR = 16 * PT;                               // double-buffered in registers: keep a pass small)
        constexpr int CPR = F32OUT ? NJ * 4 : NJ * 2;             // 16-byte output chunks per row (4 fp32 / 8 x 16-bit columns)
        constexpr int NIT = PR * CPR / 64;
        // Residual tile of a pass: read row-contiguous, 16 B per lane, from CLAMPED addresses with no branch
        // around the loads, and one pass AHEAD of its use - the first version loaded each chunk inside the
        // bounds test, i.e. one exposed L2/HBM round trip per chunk (32 per wave): 18.6 us of a 52 us launch.
        // LayerNorm-fold statistics of the sub-tile, staged behind the transpose scratch: [NJ / 2 slots][16 NI rows] float2
        static_assert(!F32OUT || ((PR * RS + 15) & ~15) == stats_stage_offset<NJ>(), "stats_tile_finalize reads the staging where the epilogue wrote it");
        float2 *wst = reinterpret_cast<float2 *>(wlds + ((PR * RS + 15) & ~15));
        uint4 q[RESID_AHEAD ? 2 : 1][RESID ? NIT : 1];
        // (the kernel-uniform stream-form test stays OUTSIDE the load loop: inside it, hipcc branched around every load and waited
        //  for each where it stood - 64 dependent round trips per wave in the residual epilogue)
        auto load_resid = [&](int pass, uint4 (&dst)[RESID ? NIT : 1]) {
            if constexpr (RESID) {
                size_t off[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int c = it * 64 + lane, rl = c / CPR, ch = c - rl * CPR;
                    int m = m_base + pass * PR + rl, n = n_base + ch * (F32OUT ? 4 : 8);
                    m = m < g.M ? m : g.M - 1;
                    n = n < g.N ? n : 0;
                    off[it] = (size_t)m * g.ldc + n;
                }
                if (F32OUT && g.C_lo) {                             // hi / lo stream: 4 + 4 values of 16 bits
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const uint2 hi = *reinterpret_cast<const uint2 *>(reinterpret_cast<const bf16_t *>(g.R) + off[it]);
                        const uint2 lo = *reinterpret_cast<const uint2 *>(reinterpret_cast<const bf16_t *>(g.R_lo) + off[it]);
                        dst[it] = uint4{hi.x, hi.y, lo.x, lo.y};
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < NIT; ++it)
                        dst[it] = F32OUT ? *reinterpret_cast<const uint4 *>(reinterpret_cast<const float *>(g.R) + off[it])
                                         : *reinterpret_cast<const uint4 *>(reinterpret_cast<const bf16_t *>(g.R) + off[it]);
                }
            }
        };
        if (RESID_AHEAD) load_resid(0, q[0]);
#pragma unroll
        for (int pass = 0; pass < NI / PT; ++pass) {
            if (RESID_AHEAD) { if (pass + 1 < NI / PT) load_resid(pass + 1, q[(pass + 1) & (RESID_AHEAD ? 1 : 0)]); }
            else load_resid(pass, q[0]);
#pragma unroll
            for (int ii = 0; ii < PT; ++ii)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const f32x4 v = acc[pass * PT + ii][j];
                    *reinterpret_cast<float4 *>(wlds + (ii * 16 + r) * RS + ((SWZ32 ? ((j * 4 + gq) ^ (r & 7)) : (j * 4 + gq)) << 4)) =
                        make_float4(v[0] + bj[j].x, v[1] + bj[j].y, v[2] + bj[j].z, v[3] + bj[j].w);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (F32OUT) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int c = it * 64 + lane, rl = c / CPR, ch = c - rl * CPR;
                    float4 v = *reinterpret_cast<const float4 *>(wlds + rl * RS + ((SWZ32 ? (ch ^ (rl & 7)) : ch) << 4));
                    const int m = m_base + pass * PR + rl, n = n_base + ch * 4;
                    if (EPI == EPI_F32_ACCUM) {
                        const uint4 u = q[RESID_AHEAD ? (pass & 1) : 0][it];
                        if (g.C_lo) {                               // x = hi + lo is exact in fp32 (11 + 11 bits), then ONE rounding
                            v.x += lo16<DT>(u.x) + lo16<DT>(u.z); v.y += hi16<DT>(u.x) + hi16<DT>(u.z);
                            v.z += lo16<DT>(u.y) + lo16<DT>(u.w); v.w += hi16<DT>(u.y) + hi16<DT>(u.w);
                        } else {
                            v.x += __uint_as_float(u.x); v.y += __uint_as_float(u.y); v.z += __uint_as_float(u.z); v.w += __uint_as_float(u.w);
                        }
                    }
                    if (g.C_lo) {                                   // kernel-uniform: hi / lo planes instead of the fp32 tensor
                        const unsigned h0 = pack16<DT>(v.x, v.y), h1 = pack16<DT>(v.z, v.w);
                        const unsigned l0 = pack16<DT>(v.x - lo16<DT>(h0), v.y - hi16<DT>(h0));
                        const unsigned l1 = pack16<DT>(v.z - lo16<DT>(h1), v.w - hi16<DT>(h1));
                        const unsigned g0 = dpp_xor1(h0), g1 = dpp_xor1(h1), k0 = dpp_xor1(l0), k1 = dpp_xor1(l1);
                        if (!(lane & 1) && m < g.M && n < g.N) {    // the even lane of a pair stores 8 columns of both planes
                            *reinterpret_cast<uint4 *>(reinterpret_cast<bf16_t *>(g.C16) + (size_t)m * g.ldc + n) = uint4{h0, h1, g0, g1};
                            *reinterpret_cast<uint4 *>(reinterpret_cast<bf16_t *>(g.C_lo) + (size_t)m * g.ldc + n) = uint4{l0, l1, k0, k1};
                        }
                    } else {
                    if (m < g.M && n < g.N)
                        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(g.C) + (size_t)m * g.ldc + n) = v;
                    if (g.C16) {                                    // kernel-uniform: 16-bit copy of the stream (LayerNorm fold)
                        // two neighbouring lanes hold 8 consecutive columns of a row: the even one stores all 16 bytes
                        const unsigned p0 = pack16<DT>(v.x, v.y), p1 = pack16<DT>(v.z, v.w);
                        const unsigned q0 = dpp_xor1(p0), q1 = dpp_xor1(p1);
                        if (!(lane & 1) && m < g.M && n < g.N)
                            *reinterpret_cast<uint4 *>(reinterpret_cast<bf16_t *>(g.C16) + (size_t)m * g.ldc + n) = uint4{p0, p1, q0, q1};
                    }
                    }
                    if (g.stats_out) {                              // kernel-uniform: (sum, sum of squares) per 32-column slot
                        const bool live = m < g.M && n < g.N;
                        const float s1 = dpp_sum8(live ? (v.x + v.y) + (v.z + v.w) : 0.f);
                        const float s2 = dpp_sum8(live ? (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w) : 0.f);
                        if ((ch & 7) == 0)                          // staged in the wave's scratch: [slot][row of the sub-tile]
                            wst[(ch >> 3) * (16 * NI) + pass * PR + rl] = make_float2(s1, s2);
                    }
                }
            } else {                                              // EPI_BF16_ADD: 8 columns per lane
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int c = it * 64 + lane, rl = c / CPR, ch = c - rl * CPR;
                    const float4 a = *reinterpret_cast<const float4 *>(wlds + rl * RS + ch * 32);
                    const float4 b = *reinterpret_cast<const float4 *>(wlds + rl * RS + ch * 32 + 16);
                    const int m = m_base + pass * PR + rl, n = n_base + ch * 8;
                    const uint4 u = q[RESID_AHEAD ? (pass & 1) : 0][it];
                    uint4 o;
                    o.x = pack16<DT>(a.x + lo16<DT>(u.x), a.y + hi16<DT>(u.x));
                    o.y = pack16<DT>(a.z + lo16<DT>(u.y), a.w + hi16<DT>(u.y));
                    o.z = pack16<DT>(b.x + lo16<DT>(u.z), b.y + hi16<DT>(u.z));
                    o.w = pack16<DT>(b.z + lo16<DT>(u.w), b.w + hi16<DT>(u.w));
                    if (m < g.M && n < g.N)
                        *reinterpret_cast<uint4 *>(reinterpret_cast<bf16_t *>(g.C) + (size_t)m * g.ldc + n) = o;
                }
            }
            asm volatile("" ::: "memory");
        }
        // (the sub-tile's statistics leaves stay in the scratch: the kernel's stats_tile_finalize turns the tile's leaves into slots)
    } else {
        // 64-column sub-tiles (NJ = 4: 128-byte rows) use an UNPADDED scratch: the 16-byte chunk c of row r is stored at
        // chunk c ^ ((r >> 1) & 7), and odd rows store the two 8-byte halves of a chunk swapped.
        //   reads  (ds_read_b128: four groups of 16 lanes, {0-3, 12-15, 20-27} ..., 64 banks): a group covers two half
        //          rows and one row pair; consecutive rows sit 32 banks apart and the XOR moves row pairs to different
        //          chunks - every bank once;
        //   writes (ds_write_b64: four groups of 16 CONSECUTIVE lanes = 16 rows at one column quad, 32 banks): the row
        //          offset (128 B) vanishes mod 32 banks, the XOR gives the 8 row pairs 8 different chunks and the half
        //          swap separates the two rows of a pair - every bank once.
        // The first form padded rows to 144 bytes: 2-way conflicts on both sides, SQ_LDS_BANK_CONFLICT ~1500 cycles per
        // 256 x 256 tile (profiles/r03_gemm_pmc.md).  Other widths keep the padded rows.
        constexpr bool SWZ = (NJ == 4);
        constexpr int RS = SWZ ? 128 : NJ * 32 + 16;              // 16*NJ 16-bit values (+ 16 bytes of padding)
        constexpr int TP = NI < TPMAX ? NI : TPMAX;               // accumulator row tiles per pass (TPMAX: register budget)
        constexpr int PR = 16 * TP, CPR = NJ * 2;                 // rows per pass, 16-byte chunks (8 columns) per row
        RopeFreq fr{};
        if constexpr (EPI == EPI_BF16_ROPE) { if (g.rope_pos && !ropet) fr = rope_freqs(g, lane); }
        if (lnt) {
            // LayerNorm fold (kernel-uniform), before any bias: acc <- rstd[m] * (acc - mean[m] * colsum[n]).  Done for the whole
            // sub-tile up front so that the column sums are dead before the RoPE coefficients of the passes become live (with
            // both live the 128 x 128 kernel's RoPE instantiation lost its second workgroup per CU)
            float4 csj[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {                        // unconditional, clamped (columns past N are never stored)
                const int n = n_base + j * 16 + gq * 4;
                csj[j] = *reinterpret_cast<const float4 *>(g.ln_colsum + (n < g.N ? n : g.N - 4));
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float2 mr = lnt[lrow + i * 16 + r];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x4 &v = acc[i][j];
                    v[0] = mr.y * (v[0] - mr.x * csj[j].x); v[1] = mr.y * (v[1] - mr.x * csj[j].y);
                    v[2] = mr.y * (v[2] - mr.x * csj[j].z); v[3] = mr.y * (v[3] - mr.x * csj[j].w);
                }
            }
        }
#pragma unroll
        for (int pass = 0; pass < NI / TP; ++pass) {
            RopeCoef cf[EPI == EPI_BF16_ROPE ? TP : 1][NJ / 2];
            if constexpr (EPI == EPI_BF16_ROPE) {                 // all coefficient reads of the pass in flight at once
                int2 pyx[TP];
#pragma unroll
                for (int ii = 0; ii < TP; ++ii) pyx[ii] = int2{0, 0};
                if (g.rope_pos) {                                 // kernel-uniform; the loads below are unconditional
#pragma unroll
                    for (int ii = 0; ii < TP; ++ii) pyx[ii] = rope_pos_of(g, m_base + (pass * TP + ii) * 16 + r);
                }
#pragma unroll
                for (int ii = 0; ii < TP; ++ii)
                    rope_load<NJ>(g, cf[ii], m_base + (pass * TP + ii) * 16 + r, n_base, lane, fr, pyx[ii], ropet);
            }
#pragma unroll
            for (int ii = 0; ii < TP; ++ii) {
                const int i = pass * TP + ii;
                if constexpr (EPI == EPI_BF16_ROPE) rope_apply<NJ>(g, acc[i], cf[ii], bj, n_base);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x4 v = acc[i][j];
                    if constexpr (EPI != EPI_BF16_ROPE) { v[0] += bj[j].x; v[1] += bj[j].y; v[2] += bj[j].z; v[3] += bj[j].w; }
                    if (EPI == EPI_BF16_GELU) {
                        const f32x2 lo = gelu_erf2(f32x2{v[0], v[1]}), hi = gelu_erf2(f32x2{v[2], v[3]});
                        v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
                    }
                    if (EPI == EPI_BF16_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    }
                    uint2 pk;
                    if (EPI == EPI_BF16_ROPE && DT == DT_F16 && g.v_bf16 && n_base + (j >> 1) * 32 >= g.rope_cols) {   // wave-uniform
                        pk.x = pack16<DT_BF16>(v[0], v[1]); pk.y = pack16<DT_BF16>(v[2], v[3]);
                    } else {
                        pk.x = pack16<DT>(v[0], v[1]); pk.y = pack16<DT>(v[2], v[3]);
                    }
                    if constexpr (SWZ)
                        *reinterpret_cast<uint2 *>(wlds + (ii * 16 + r) * RS + (((j * 2 + (gq >> 1)) ^ ((r >> 1) & 7)) << 4) + (((gq ^ r) & 1) << 3)) = pk;
                    else
                        *reinterpret_cast<uint2 *>(wlds + (ii * 16 + r) * RS + (j * 16 + gq * 4) * 2) = pk;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < PR * CPR / 64; ++it) {
                const int c = it * 64 + lane, rl = c / CPR, ch = c - rl * CPR;
                uint4 v = *reinterpret_cast<const uint4 *>(wlds + rl * RS + ((SWZ ? (ch ^ ((rl >> 1) & 7)) : ch) << 4));
                if (SWZ && (rl & 1)) v = uint4{v.z, v.w, v.x, v.y};
                const int m = m_base + pass * PR + rl, n = n_base + ch * 8;
                if (m < g.M && n < g.N)
                    *reinterpret_cast<uint4 *>(reinterpret_cast<bf16_t *>(g.C) + (size_t)m * g.ldc + n) = v;
            }
            asm volatile("" ::: "memory");
        }
    }
}

// Implicit-GEMM convolution: K-tile kt covers input channels [64 q, 64 q + 64) of tap `tap` with (q, tap) = (kt / 9, kt % 9)
// - slice-major, the order in which the direct convolution (conv_tail.hip) walks K, so both forms add the same products
// in the same order and return the same bits.  Offset of that K-tile in a weight row [3][3][Cin]:
__device__ __forceinline__ size_t conv_k_offset(int kt, int Cin) {
    const int q = kt / 9, tap = kt - q * 9;
    return (size_t)tap * Cin + q * BK;
}

// XCD-aware bijective tile remap: consecutive tiles (sharing an A panel) land on one XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
}

}  // namespace m3gemm
