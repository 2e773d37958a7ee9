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
// GELU (erf form).  erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the bf16 output
// rounding of 2^-9): one v_rcp, one v_exp and a 5-term Horner chain instead of libm's erff.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float erf_abs = 1.0f - p * t * e;
    const float erf = x < 0.f ? -erf_abs : erf_abs;
    return 0.5f * x * (1.0f + erf);
}

__device__ __forceinline__ void glds16(const void *g, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned *)g,
                                     (__attribute__((address_space(3))) unsigned *)lds_wave_base, 16, 0, 0);
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
    const int *pos_yx;      // [tokens_per_image, 2]
    const float *cos_sin;   // [max_pos, 16, 2]
    int tokens_per_image, rope_cols;
    // grouped launch (blockIdx.y = group): group 1 uses W2/bias2 and A/C/R advanced by the strides
    const bf16_t *W2;
    const float *bias2;
    long long a_gstride, c_gstride;   // elements of A, elements of C/R
    int groups;
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
    }
    return g;
}

enum { EPI_BF16 = 0, EPI_BF16_GELU = 1, EPI_F32 = 2, EPI_F32_ACCUM = 3, EPI_BF16_RELU = 4, EPI_BF16_ADD = 5,
       EPI_BF16_ROPE = 6 };


// Epilogue for one 16x16 accumulator tile: the lane holds C[m][n..n+3] (operands were swapped).
template <int EPI>
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
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
        if (EPI == EPI_BF16_ADD) {
            const ushort4 r = *reinterpret_cast<const ushort4 *>(reinterpret_cast<const bf16_t *>(g.R) + off);
            v[0] += bf2f(r.x); v[1] += bf2f(r.y); v[2] += bf2f(r.z); v[3] += bf2f(r.w);
        }
        if (EPI == EPI_BF16_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        ushort4 o;
        o.x = f2bf(v[0]); o.y = f2bf(v[1]); o.z = f2bf(v[2]); o.w = f2bf(v[3]);
        *reinterpret_cast<ushort4 *>(reinterpret_cast<bf16_t *>(g.C) + off) = o;
    }
}

// Fused RoPE-2D on one 16-row x 64-column accumulator strip (= one attention head of one token
// per lane): t[0..3] are the four 16-column tiles, lane holds columns 16*j + 4*(lane>>4) + e.
// Columns 0..31 rotate with the token's y, 32..63 with x; element i pairs with i+16, i.e. tile
// 0 with tile 1 and tile 2 with tile 3 in the SAME lane and register.  Bias is added first.
__device__ __forceinline__ void rope_strip(const GemmArgs &g, f32x4 *t, int m, int n_base, int lane) {
    if (m >= g.M) return;
    const int fi = (lane >> 4) * 4;
    if (g.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 b = *reinterpret_cast<const float4 *>(g.bias + n_base + j * 16 + fi);
            t[j][0] += b.x; t[j][1] += b.y; t[j][2] += b.z; t[j][3] += b.w;
        }
    }
    if (n_base >= g.rope_cols) return;
    const int2 pos = *reinterpret_cast<const int2 *>(g.pos_yx + 2 * (m % g.tokens_per_image));
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const float4 *cs = reinterpret_cast<const float4 *>(g.cos_sin + ((size_t)(blk ? pos.y : pos.x) * 16 + fi) * 2);
        const float4 c01 = cs[0], c23 = cs[1];               // (cos,sin) of freq fi, fi+1 | fi+2, fi+3
        const float cc[4] = {c01.x, c01.z, c23.x, c23.z}, ss[4] = {c01.y, c01.w, c23.y, c23.w};
        f32x4 &a = t[2 * blk], &b = t[2 * blk + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x1 = a[e], x2 = b[e];
            a[e] = x1 * cc[e] - x2 * ss[e];
            b[e] = x2 * cc[e] + x1 * ss[e];
        }
    }
}

// XCD-aware bijective tile remap: consecutive tiles (sharing an A panel) land on one XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
}

}  // namespace m3gemm
