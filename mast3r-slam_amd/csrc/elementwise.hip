// HBM-bound helper kernels of the network path (gfx950): LayerNorm, patch extraction,
// bilinear 2x upsampling, pixel shuffles and the output heads' post-processing.  All are
// streaming kernels with 8-16 byte vector accesses per lane; none is on the MFMA roofline.
// Kernels that convert to / from the 16-bit storage type are templated on it (DT_BF16 / DT_F16, gemm_common.h);
// pure data movement (ReLU by sign bit, concat, un-shuffle) is the same for both.
#include "gemm_common.h"

namespace {

using m3gemm::bf16_t;
using m3gemm::DT_BF16;
using m3gemm::DT_F16;
using m3gemm::pack16;
using m3gemm::lo16;
using m3gemm::hi16;
constexpr int kThreads = 256;

template <int DT> __device__ __forceinline__ float to_f32(bf16_t v) { return lo16<DT>((unsigned)v); }
template <int DT> __device__ __forceinline__ bf16_t from_f32(float f) { return (bf16_t)(pack16<DT>(f, 0.f) & 0xffffu); }

#define M3_DT_LAUNCH(dtype, KERNEL, ...)                                                        \
    do {                                                                                        \
        if ((dtype) == DT_F16) hipLaunchKernelGGL((KERNEL<DT_F16>), __VA_ARGS__);               \
        else hipLaunchKernelGGL((KERNEL<DT_BF16>), __VA_ARGS__);                                \
    } while (0)
#define M3_DT_OK(dtype) M3_REQUIRE((dtype) == DT_BF16 || (dtype) == DT_F16)

// ---------------------------------------------------------------- LayerNorm: one wave per row
// x f32 [M,C] -> y bf16 [M,C]; two-pass statistics in fp32 from registers (row read once).
// HL: the input is a residual stream kept as two fp16 planes, x = float(hi) + float(lo) (LayerNorm fold, gemm_common.h); `x` then
// points at the hi plane and `lo` at the lo plane.  Same arithmetic behind the loads, so the result equals the fp32-input kernel on
// hi.float() + lo.float() bit for bit.
template <int VPL /* float4 per lane */, int DT, bool HL = false>
__global__ void __launch_bounds__(kThreads)
k_layernorm(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
            const float *__restrict__ gamma2, const float *__restrict__ beta2, bf16_t *__restrict__ y, int M, int C,
            int split, int in_shift, float eps, const bf16_t *__restrict__ lo = nullptr) {
    const int row = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    if (row >= split) { gamma = gamma2; beta = beta2; }
    int in_row = row + in_shift;
    in_row = in_row >= M ? in_row - M : in_row;
    float4 v[VPL];
    float s = 0.f;
    if constexpr (HL) {
        const uint2 *hr = reinterpret_cast<const uint2 *>(reinterpret_cast<const bf16_t *>(x) + (size_t)in_row * C);
        const uint2 *lr = reinterpret_cast<const uint2 *>(lo + (size_t)in_row * C);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const uint2 h = hr[lane + 64 * i], l = lr[lane + 64 * i];
            v[i] = make_float4(m3gemm::lo16<m3gemm::DT_F16>(h.x) + m3gemm::lo16<m3gemm::DT_F16>(l.x),
                               m3gemm::hi16<m3gemm::DT_F16>(h.x) + m3gemm::hi16<m3gemm::DT_F16>(l.x),
                               m3gemm::lo16<m3gemm::DT_F16>(h.y) + m3gemm::lo16<m3gemm::DT_F16>(l.y),
                               m3gemm::hi16<m3gemm::DT_F16>(h.y) + m3gemm::hi16<m3gemm::DT_F16>(l.y));
        }
    } else {
        const float4 *xr = reinterpret_cast<const float4 *>(x + (size_t)in_row * C);
#pragma unroll
        for (int i = 0; i < VPL; ++i) v[i] = xr[lane + 64 * i];
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = rsqrtf(q / (float)C + eps);
    uint2 *yr = reinterpret_cast<uint2 *>(y + (size_t)row * C);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const float4 gm = reinterpret_cast<const float4 *>(gamma)[lane + 64 * i];
        const float4 bt = reinterpret_cast<const float4 *>(beta)[lane + 64 * i];
        uint2 o;
        o.x = pack16<DT>((v[i].x - mean) * rstd * gm.x + bt.x, (v[i].y - mean) * rstd * gm.y + bt.y);
        o.y = pack16<DT>((v[i].z - mean) * rstd * gm.z + bt.z, (v[i].w - mean) * rstd * gm.w + bt.w);
        yr[lane + 64 * i] = o;
    }
}

// Two LayerNorms of the same rows in one pass (decoder block entry: norm1 of a branch's own tokens and norm_y of
// the same tokens as the OTHER branch's cross-attention memory): x f32 [2M,C] (two branches) is read once, the
// statistics are shared, y_own[row] uses the row's branch parameters (ga*, ba*), y_cross[(row + M) mod 2M] the
// destination branch's (gb*, bb*).  Bit-identical to the two separate launches.
template <int VPL, int DT>
__global__ void __launch_bounds__(kThreads)
k_layernorm_dual(const float *__restrict__ x, const float *__restrict__ ga0, const float *__restrict__ ba0,
                 const float *__restrict__ ga1, const float *__restrict__ ba1, const float *__restrict__ gb0,
                 const float *__restrict__ bb0, const float *__restrict__ gb1, const float *__restrict__ bb1,
                 bf16_t *__restrict__ y_own, bf16_t *__restrict__ y_cross, int M, int C, float eps) {
    const int row = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (row >= 2 * M) return;
    const int lane = threadIdx.x & 63;
    const bool g1 = row >= M;
    const float *ga = g1 ? ga1 : ga0, *ba = g1 ? ba1 : ba0;           // own branch
    const float *gb = g1 ? gb0 : gb1, *bb = g1 ? bb0 : bb1;           // destination (other) branch
    const int crow = g1 ? row - M : row + M;
    const float4 *xr = reinterpret_cast<const float4 *>(x + (size_t)row * C);
    float4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { v[i] = xr[lane + 64 * i]; s += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = rsqrtf(q / (float)C + eps);
    uint2 *yo = reinterpret_cast<uint2 *>(y_own + (size_t)row * C), *yc = reinterpret_cast<uint2 *>(y_cross + (size_t)crow * C);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const float4 gm = reinterpret_cast<const float4 *>(ga)[lane + 64 * i], bt = reinterpret_cast<const float4 *>(ba)[lane + 64 * i];
        const float4 gn = reinterpret_cast<const float4 *>(gb)[lane + 64 * i], bn = reinterpret_cast<const float4 *>(bb)[lane + 64 * i];
        uint2 o, p;
        o.x = pack16<DT>((v[i].x - mean) * rstd * gm.x + bt.x, (v[i].y - mean) * rstd * gm.y + bt.y);
        o.y = pack16<DT>((v[i].z - mean) * rstd * gm.z + bt.z, (v[i].w - mean) * rstd * gm.w + bt.w);
        p.x = pack16<DT>((v[i].x - mean) * rstd * gn.x + bn.x, (v[i].y - mean) * rstd * gn.y + bn.y);
        p.y = pack16<DT>((v[i].z - mean) * rstd * gn.z + bn.z, (v[i].w - mean) * rstd * gn.w + bn.w);
        yo[lane + 64 * i] = o;
        yc[lane + 64 * i] = p;
    }
}

// ---------------------------------------------------------------- patch extraction (im2col of the 16x16/16 conv)
// img uint8 [B,H,W,3] -> A 16-bit [B*(H/16)*(W/16), 768], column = c*256 + py*16 + px, value (v/255-0.5)/0.5
// A thread owns 8 horizontally consecutive pixels of a patch row, ALL three channels: 24 contiguous image bytes in (three
// 8-byte loads; the offset is a multiple of 24), three 16-byte stores out.  (One element per thread with a modulo chain
// and 2-byte stores before: 36.5 us for 16 images of 512 x 512; same values.)
template <int DT>
__global__ void __launch_bounds__(kThreads)
k_patchify(const uint8_t *__restrict__ img, bf16_t *__restrict__ A, int B, int H, int W) {
    const int gw = W / 16, gh = H / 16;
    const int64_t total = (int64_t)B * gh * gw * 32;                          // (token, py, half row)
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    const int h8 = (int)(i & 1), py = (int)((i >> 1) & 15);
    const int64_t tok = i >> 5;
    const int tx = (int)(tok % gw), ty = (int)((tok / gw) % gh), b = (int)(tok / ((int64_t)gw * gh));
    const uint8_t *src = img + (((size_t)b * H + ty * 16 + py) * W + tx * 16 + h8 * 8) * 3;
    union { uint2 q[3]; uint8_t u[24]; } in;
    in.q[0] = reinterpret_cast<const uint2 *>(src)[0];
    in.q[1] = reinterpret_cast<const uint2 *>(src)[1];
    in.q[2] = reinterpret_cast<const uint2 *>(src)[2];
    bf16_t *dst = A + (size_t)tok * 768 + py * 16 + h8 * 8;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        unsigned w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            w[k] = pack16<DT>(((float)in.u[(2 * k) * 3 + c] / 255.0f - 0.5f) / 0.5f, ((float)in.u[(2 * k + 1) * 3 + c] / 255.0f - 0.5f) / 0.5f);
        *reinterpret_cast<uint4 *>(dst + c * 256) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---------------------------------------------------------------- generic small elementwise ops
template <int DT>
__global__ void __launch_bounds__(kThreads)
k_f32_to_16(const float *__restrict__ x, bf16_t *__restrict__ y, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4 *>(x)[i];
    uint2 o; o.x = pack16<DT>(v.x, v.y); o.y = pack16<DT>(v.z, v.w);
    reinterpret_cast<uint2 *>(y)[i] = o;
}

template <int DTI, int DTO>
__global__ void __launch_bounds__(kThreads)
k_cast16(const bf16_t *__restrict__ x, bf16_t *__restrict__ y, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n8) return;
    union U { uint4 q; unsigned w[4]; } u, o;
    u.q = reinterpret_cast<const uint4 *>(x)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) o.w[k] = pack16<DTO>(lo16<DTI>(u.w[k]), hi16<DTI>(u.w[k]));
    reinterpret_cast<uint4 *>(y)[i] = o.q;
}

__global__ void __launch_bounds__(kThreads)
k_relu_bf16(const bf16_t *__restrict__ x, bf16_t *__restrict__ y, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n8) return;
    uint4 v = reinterpret_cast<const uint4 *>(x)[i];
    unsigned *w = reinterpret_cast<unsigned *>(&v);
#pragma unroll
    for (int k = 0; k < 4; ++k) {             // bf16 relu: clear halves whose sign bit is set
        unsigned lo = w[k] & 0xffffu, hi = w[k] >> 16;
        lo = (lo & 0x8000u) ? 0u : lo;
        hi = (hi & 0x8000u) ? 0u : hi;
        w[k] = lo | (hi << 16);
    }
    reinterpret_cast<uint4 *>(y)[i] = v;
}

template <int DT>
__global__ void __launch_bounds__(kThreads)
k_add16(const bf16_t *__restrict__ a, const bf16_t *__restrict__ b, bf16_t *__restrict__ y, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n8) return;
    union U { uint4 q; unsigned w[4]; } u, v, o;
    u.q = reinterpret_cast<const uint4 *>(a)[i];
    v.q = reinterpret_cast<const uint4 *>(b)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) o.w[k] = pack16<DT>(lo16<DT>(u.w[k]) + lo16<DT>(v.w[k]), hi16<DT>(u.w[k]) + hi16<DT>(v.w[k]));
    reinterpret_cast<uint4 *>(y)[i] = o.q;
}

// rows of `a` [M,Ca] and `b` [M,Cb] (bf16) side by side into out [M,Ca+Cb]; Ca, Cb multiples of 8
__global__ void __launch_bounds__(kThreads)
k_concat2(const bf16_t *__restrict__ a, const bf16_t *__restrict__ b, bf16_t *__restrict__ out, int64_t M,
          int Ca, int Cb) {
    const int cw = (Ca + Cb) / 8;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= M * cw) return;
    const int64_t m = i / cw;
    const int c = (int)(i % cw) * 8;
    const uint4 v = (c < Ca) ? *reinterpret_cast<const uint4 *>(a + m * Ca + c)
                             : *reinterpret_cast<const uint4 *>(b + m * Cb + (c - Ca));
    *reinterpret_cast<uint4 *>(out + m * (Ca + Cb) + c) = v;
}

// GEMM output of a k=s transposed conv, [B*h*w, s*s*C] with column (dy*s+dx)*C + c  ->  NHWC [B,h*s,w*s,C]
__global__ void __launch_bounds__(kThreads)
k_unshuffle(const bf16_t *__restrict__ in, bf16_t *__restrict__ out, int B, int h, int w, int s, int C, int Cpad) {
    const int c8 = C / 8;
    const int64_t total = (int64_t)B * h * s * w * s * c8;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % c8) * 8;
    int64_t p = i / c8;
    const int X = (int)(p % (w * s)); p /= (w * s);
    const int Y = (int)(p % (h * s));
    const int b = (int)(p / (h * s));
    const int y = Y / s, dy = Y % s, x = X / s, dx = X % s;
    const uint4 v = *reinterpret_cast<const uint4 *>(in + (((size_t)b * h + y) * w + x) * (size_t)(s * s * C) +
                                                     (dy * s + dx) * C + c);
    *reinterpret_cast<uint4 *>(out + (((size_t)b * h * s + Y) * (w * s) + X) * Cpad + c) = v;
}

// bilinear x2, align_corners=True, NHWC 16-bit, C multiple of 8
template <int DT>
__global__ void __launch_bounds__(kThreads)
k_upsample2x(const bf16_t *__restrict__ in, bf16_t *__restrict__ out, int B, int H, int W, int C) {
    const int OH = 2 * H, OW = 2 * W, c8 = C / 8;
    const int64_t total = (int64_t)B * OH * OW * c8;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % c8) * 8;
    int64_t p = i / c8;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    const float sy = (OH > 1) ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    const float sx = (OW > 1) ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    const float fy = oy * sy, fx = ox * sx;
    int y0 = (int)fy, x0 = (int)fx;
    y0 = min(y0, H - 1); x0 = min(x0, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const bf16_t *base = in + (size_t)b * H * W * C + c;
    union U { uint4 q; unsigned w[4]; } a, bq, cq, d, o;
    a.q = *reinterpret_cast<const uint4 *>(base + ((size_t)y0 * W + x0) * C);
    bq.q = *reinterpret_cast<const uint4 *>(base + ((size_t)y0 * W + x1) * C);
    cq.q = *reinterpret_cast<const uint4 *>(base + ((size_t)y1 * W + x0) * C);
    d.q = *reinterpret_cast<const uint4 *>(base + ((size_t)y1 * W + x1) * C);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float tl = lo16<DT>(a.w[k]) * (1.f - wx) + lo16<DT>(bq.w[k]) * wx;
        const float bl = lo16<DT>(cq.w[k]) * (1.f - wx) + lo16<DT>(d.w[k]) * wx;
        const float th = hi16<DT>(a.w[k]) * (1.f - wx) + hi16<DT>(bq.w[k]) * wx;
        const float bh = hi16<DT>(cq.w[k]) * (1.f - wx) + hi16<DT>(d.w[k]) * wx;
        o.w[k] = pack16<DT>(tl * (1.f - wy) + bl * wy, th * (1.f - wy) + bh * wy);
    }
    // streamed once by the next convolution's gather, 4x the bytes of the input: keep it out of the way of the taps in L2
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(u32x4{o.w[0], o.w[1], o.w[2], o.w[3]},
                                reinterpret_cast<u32x4 *>(out + (((size_t)b * OH + oy) * OW + ox) * C + c));
}

// out[b,oy,ox,:] = bilinear_x2(low)[b,oy,ox,:] + y[b,oy,ox,:] for oy < OH <= 2H, ox < OW <= 2W (the DPT fusion block's
// "upsample the coarser path, add the refined skip connection", with the crop the odd token grids need).  The
// interpolation weights are those of the FULL 2H x 2W align_corners map; the sum is formed in fp32 and rounded once -
// the upsampled map (537 MB per head at the last level) is neither written nor read back.
template <int DT>
__global__ void __launch_bounds__(kThreads)
k_add_upsample2x(const bf16_t *__restrict__ low, const bf16_t *__restrict__ y, bf16_t *__restrict__ out, int B, int H,
                 int W, int OH, int OW, int C) {
    const int c8 = C / 8;
    const int64_t total = (int64_t)B * OH * OW * c8;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % c8) * 8;
    int64_t p = i / c8;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int b = (int)(p / OH);
    const float sy = (H > 1) ? (float)(H - 1) / (float)(2 * H - 1) : 0.f;
    const float sx = (W > 1) ? (float)(W - 1) / (float)(2 * W - 1) : 0.f;
    const float fy = oy * sy, fx = ox * sx;
    int y0 = (int)fy, x0 = (int)fx;
    y0 = min(y0, H - 1); x0 = min(x0, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const bf16_t *base = low + (size_t)b * H * W * C + c;
    union U { uint4 q; unsigned w[4]; } a, bq, cq, d, r, o;
    const size_t oi = (((size_t)b * OH + oy) * OW + ox) * C + c;
    r.q = *reinterpret_cast<const uint4 *>(y + oi);
    a.q = *reinterpret_cast<const uint4 *>(base + ((size_t)y0 * W + x0) * C);
    bq.q = *reinterpret_cast<const uint4 *>(base + ((size_t)y0 * W + x1) * C);
    cq.q = *reinterpret_cast<const uint4 *>(base + ((size_t)y1 * W + x0) * C);
    d.q = *reinterpret_cast<const uint4 *>(base + ((size_t)y1 * W + x1) * C);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float tl = lo16<DT>(a.w[k]) * (1.f - wx) + lo16<DT>(bq.w[k]) * wx;
        const float bl = lo16<DT>(cq.w[k]) * (1.f - wx) + lo16<DT>(d.w[k]) * wx;
        const float th = hi16<DT>(a.w[k]) * (1.f - wx) + hi16<DT>(bq.w[k]) * wx;
        const float bh = hi16<DT>(cq.w[k]) * (1.f - wx) + hi16<DT>(d.w[k]) * wx;
        o.w[k] = pack16<DT>((tl * (1.f - wy) + bl * wy) + lo16<DT>(r.w[k]), (th * (1.f - wy) + bh * wy) + hi16<DT>(r.w[k]));
    }
    *reinterpret_cast<uint4 *>(out + oi) = o.q;
}

// DPT head output [P,4] f32 (xyz, conf logit) -> pts3d [P,3] = xyz/|xyz| * expm1(|xyz|), conf [P] = 1 + exp(c)
__global__ void __launch_bounds__(kThreads)
k_pts_post(const float *__restrict__ in, float *__restrict__ pts, float *__restrict__ conf, int64_t P) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= P) return;
    const float4 v = reinterpret_cast<const float4 *>(in)[i];
    const float d = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    const float dc = fmaxf(d, 1e-8f);
    const float sc = expm1f(d) / dc;
    pts[3 * i + 0] = v.x * sc; pts[3 * i + 1] = v.y * sc; pts[3 * i + 2] = v.z * sc;
    conf[i] = 1.0f + expf(v.w);
}

// feature-MLP output [B*gh*gw, 25*256] 16-bit (column c*256 + py*16 + px) -> pixel shuffle(16) ->
// desc [B,H,W,24] L2-normalised, desc_conf [B,H,W] f32 = exp(channel 24).
// F16OUT: the descriptors are stored as IEEE half ("fp16 features", BASELINE configs[4]) - the fp32 value rounded once.
// One 16 x 16 patch per workgroup: its 25 x 256 input values (12.8 KB, contiguous) are copied to LDS with 16-byte loads,
// thread t normalises pixel (t / 16, t % 16) (sum of squares over channels 0..23 ascending), the 256 x 24 results go to
// an LDS tile and leave as 16 rows of 1 536 contiguous bytes, 16 bytes per lane: 630 MB at 5.3 TB/s (118 us for 16
// maps of 512 x 512).  History: one pixel per thread with 25 two-byte loads 226 us; eight consecutive pixels per thread
// (16-byte loads, but a store instruction's 64 lanes 768 bytes apart: 64 cache lines each) 178-192 us; same bits all.
template <int DT, bool F16OUT>
__global__ void __launch_bounds__(kThreads)
k_desc_post(const bf16_t *__restrict__ in, void *__restrict__ desc_out, float *__restrict__ dconf, int B, int H, int W) {
    constexpr int OB = F16OUT ? 48 : 96;                       // output bytes per pixel
    __shared__ __attribute__((aligned(16))) unsigned short vin[25 * 256];
    __shared__ __attribute__((aligned(16))) unsigned char vout[256 * 96];
    const int t = threadIdx.x;
    const int gw = W / 16, gh = H / 16;
    const int patch = blockIdx.x;                              // (b, patch row, patch column)
    const int pxc = patch % gw, pyr = (patch / gw) % gh, b = patch / (gw * gh);
    const uint4 *src = reinterpret_cast<const uint4 *>(in + (size_t)patch * 6400);
    for (int i = t; i < 800; i += kThreads) reinterpret_cast<uint4 *>(vin)[i] = src[i];
    __syncthreads();
    float v[25];
#pragma unroll
    for (int c = 0; c < 25; ++c) v[c] = lo16<DT>((unsigned)vin[c * 256 + t]);
    float n2 = 0.f;
#pragma unroll
    for (int c = 0; c < 24; ++c) n2 += v[c] * v[c];
    const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
    if constexpr (F16OUT) {
        uint4 *o = reinterpret_cast<uint4 *>(vout + t * OB);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            o[c] = make_uint4(pack16<DT_F16>(v[8 * c] * inv, v[8 * c + 1] * inv), pack16<DT_F16>(v[8 * c + 2] * inv, v[8 * c + 3] * inv),
                              pack16<DT_F16>(v[8 * c + 4] * inv, v[8 * c + 5] * inv), pack16<DT_F16>(v[8 * c + 6] * inv, v[8 * c + 7] * inv));
    } else {
        float4 *o = reinterpret_cast<float4 *>(vout + t * OB);
#pragma unroll
        for (int c = 0; c < 6; ++c) o[c] = make_float4(v[4 * c] * inv, v[4 * c + 1] * inv, v[4 * c + 2] * inv, v[4 * c + 3] * inv);
    }
    const int py = t >> 4, px = t & 15;
    dconf[((size_t)b * H + pyr * 16 + py) * W + pxc * 16 + px] = expf(v[24]);
    __syncthreads();
    // 16 rows x (16 * OB) bytes, each row contiguous in the output image
    constexpr int RB = 16 * OB, PIECES = 16 * RB / 16;         // 16-byte pieces of the tile
    unsigned char *base = reinterpret_cast<unsigned char *>(desc_out);
    for (int i = t; i < PIECES; i += kThreads) {
        const int row = i / (RB / 16), off = (i - row * (RB / 16)) * 16;
        const size_t pix0 = ((size_t)b * H + pyr * 16 + row) * W + pxc * 16;
        *reinterpret_cast<uint4 *>(base + pix0 * OB + off) = *reinterpret_cast<const uint4 *>(vout + row * RB + off);
    }
}

}  // namespace

// Packed snapshot of up to 8 result tensors into one send buffer (dist.PackedGather): segment s = blockIdx.y copies nbytes
// from src to dst + dst_off, 16 bytes per lane per trip (mode 0), or narrows int64 values to int32 (mode 1: the match
// index travels as int32).  One launch instead of seven device-to-device copies (the runtime's copy kernel moved the 77 MB
// of an 8-pair step at ~0.15 TB/s: 0.6 ms of copy-kernel time per step).
struct PackSegs { m3_pack_seg seg[8]; };
__global__ void __launch_bounds__(256)
k_pack_fields(unsigned char *__restrict__ dst, const PackSegs segs) {
    const m3_pack_seg sg = segs.seg[blockIdx.y];
    const int64_t stride = (int64_t)gridDim.x * 256;
    unsigned char *d = dst + sg.dst_off;
    if (sg.mode == 0) {
        const int64_t n16 = sg.nbytes >> 4;
        const uint4 *s4 = reinterpret_cast<const uint4 *>(sg.src);
        uint4 *d4 = reinterpret_cast<uint4 *>(d);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) d4[i] = s4[i];
        const int64_t tail = sg.nbytes & 15;                     // < 16 trailing bytes
        if (blockIdx.x == 0 && threadIdx.x < tail)
            d[(n16 << 4) + threadIdx.x] = reinterpret_cast<const unsigned char *>(sg.src)[(n16 << 4) + threadIdx.x];
    } else {
        const int64_t n = sg.nbytes >> 2;                        // int32 values out
        const long long *s8 = reinterpret_cast<const long long *>(sg.src);
        int *d4 = reinterpret_cast<int *>(d);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) d4[i] = (int)s8[i];
    }
}

extern "C" {

#define M3_LN_CASES(DTV)                                                                                                \
    switch (C / 256) {                                                                                                  \
        M3_LN(1, DTV); M3_LN(2, DTV); M3_LN(3, DTV); M3_LN(4, DTV); M3_LN(5, DTV); M3_LN(6, DTV); M3_LN(7, DTV); M3_LN(8, DTV); \
        default: return M3_ERR_UNSUPPORTED;                                                                             \
    }

int m3_layernorm_dt(const float *x, const float *gamma, const float *beta, void *y, int M, int C, float eps, int dtype,
                    void *stream) {
    M3_REQUIRE(x && gamma && beta && y && M > 0 && C > 0 && C % 256 == 0 && C <= 2048);
    M3_DT_OK(dtype);
    dim3 grid(m3_cdiv(M, kThreads / 64)), blk(kThreads);
    hipStream_t st = (hipStream_t)stream;
#define M3_LN(V, DTV) case V: hipLaunchKernelGGL((k_layernorm<V, DTV>), grid, blk, 0, st, x, gamma, beta, gamma, beta, (bf16_t *)y, M, C, M, 0, eps); break
    if (dtype == DT_F16) { M3_LN_CASES(DT_F16) } else { M3_LN_CASES(DT_BF16) }
#undef M3_LN
    M3_CHECK_LAUNCH("m3_layernorm");
    return M3_OK;
}
int m3_layernorm_bf16(const float *x, const float *gamma, const float *beta, void *y, int M, int C, float eps,
                      void *stream) {
    return m3_layernorm_dt(x, gamma, beta, y, M, C, eps, DT_BF16, stream);
}

int m3_layernorm_grouped2_dt(const float *x, const float *gamma0, const float *beta0, const float *gamma1,
                             const float *beta1, void *y, int M, int C, int in_row_shift, float eps, int dtype,
                             void *stream) {
    M3_REQUIRE(x && gamma0 && beta0 && gamma1 && beta1 && y && M > 0 && C > 0 && C % 256 == 0 && C <= 2048);
    M3_REQUIRE(in_row_shift == 0 || in_row_shift == M);
    M3_DT_OK(dtype);
    dim3 grid(m3_cdiv(2 * M, kThreads / 64)), blk(kThreads);
    hipStream_t st = (hipStream_t)stream;
#define M3_LN(V, DTV) case V: hipLaunchKernelGGL((k_layernorm<V, DTV>), grid, blk, 0, st, x, gamma0, beta0, gamma1, beta1, (bf16_t *)y, 2 * M, C, M, in_row_shift, eps); break
    if (dtype == DT_F16) { M3_LN_CASES(DT_F16) } else { M3_LN_CASES(DT_BF16) }
#undef M3_LN
    M3_CHECK_LAUNCH("m3_layernorm_grouped2");
    return M3_OK;
}
int m3_layernorm_hl_dt(const void *hi, const void *lo, const float *gamma0, const float *beta0, const float *gamma1,
                       const float *beta1, void *y, int rows, int C, int split, float eps, int dtype, void *stream) {
    M3_REQUIRE(hi && lo && gamma0 && beta0 && gamma1 && beta1 && y && rows > 0 && split >= 0 && C > 0 && C % 256 == 0 && C <= 2048);
    M3_REQUIRE(((reinterpret_cast<size_t>(hi) | reinterpret_cast<size_t>(lo)) & 7) == 0);
    M3_DT_OK(dtype);
    dim3 grid(m3_cdiv(rows, kThreads / 64)), blk(kThreads);
    hipStream_t st = (hipStream_t)stream;
#define M3_LN(V, DTV) case V: hipLaunchKernelGGL((k_layernorm<V, DTV, true>), grid, blk, 0, st, (const float *)hi, gamma0, beta0, gamma1, beta1, (bf16_t *)y, rows, C, split, 0, eps, (const bf16_t *)lo); break
    if (dtype == DT_F16) { M3_LN_CASES(DT_F16) } else { M3_LN_CASES(DT_BF16) }
#undef M3_LN
    M3_CHECK_LAUNCH("m3_layernorm_hl");
    return M3_OK;
}
int m3_layernorm_bf16_grouped2(const float *x, const float *gamma0, const float *beta0, const float *gamma1,
                               const float *beta1, void *y, int M, int C, int in_row_shift, float eps, void *stream) {
    return m3_layernorm_grouped2_dt(x, gamma0, beta0, gamma1, beta1, y, M, C, in_row_shift, eps, DT_BF16, stream);
}

int m3_layernorm_dual2_dt(const float *x, const float *ga0, const float *ba0, const float *ga1, const float *ba1,
                          const float *gb0, const float *bb0, const float *gb1, const float *bb1, void *y_own,
                          void *y_cross, int M, int C, float eps, int dtype, void *stream) {
    M3_REQUIRE(x && ga0 && ba0 && ga1 && ba1 && gb0 && bb0 && gb1 && bb1 && y_own && y_cross && y_own != y_cross);
    M3_REQUIRE(M > 0 && C > 0 && C % 256 == 0 && C <= 2048);
    M3_DT_OK(dtype);
    dim3 grid(m3_cdiv(2 * M, kThreads / 64)), blk(kThreads);
    hipStream_t st = (hipStream_t)stream;
#define M3_LN(V, DTV) case V: hipLaunchKernelGGL((k_layernorm_dual<V, DTV>), grid, blk, 0, st, x, ga0, ba0, ga1, ba1, gb0, bb0, gb1, bb1, (bf16_t *)y_own, (bf16_t *)y_cross, M, C, eps); break
    if (dtype == DT_F16) { M3_LN_CASES(DT_F16) } else { M3_LN_CASES(DT_BF16) }
#undef M3_LN
    M3_CHECK_LAUNCH("m3_layernorm_dual2");
    return M3_OK;
}

int m3_patchify16_dt(const uint8_t *img, void *A, int B, int H, int W, int dtype, void *stream) {
    M3_REQUIRE(img && A && B > 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0);
    // k_patchify reads the image with 8-byte loads and stores 16-byte pieces: a uint8 view with an odd storage offset is
    // a status, not a misaligned vector access
    M3_REQUIRE(reinterpret_cast<uintptr_t>(img) % 8 == 0 && reinterpret_cast<uintptr_t>(A) % 16 == 0);
    M3_DT_OK(dtype);
    const int64_t total = (int64_t)B * (H / 16) * (W / 16) * 32;      // 8 pixels x 3 channels per thread
    M3_DT_LAUNCH(dtype, k_patchify, dim3(m3_cdiv(total, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, img,
                 (bf16_t *)A, B, H, W);
    M3_CHECK_LAUNCH("m3_patchify16");
    return M3_OK;
}
int m3_patchify16(const uint8_t *img, void *A, int B, int H, int W, void *stream) {
    return m3_patchify16_dt(img, A, B, H, W, DT_BF16, stream);
}

int m3_cast_f32_dt(const float *x, void *y, int64_t n, int dtype, void *stream) {
    M3_REQUIRE(x && y && n > 0 && n % 4 == 0);
    M3_DT_OK(dtype);
    M3_DT_LAUNCH(dtype, k_f32_to_16, dim3(m3_cdiv(n / 4, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, x,
                 (bf16_t *)y, n / 4);
    M3_CHECK_LAUNCH("m3_cast_f32");
    return M3_OK;
}
int m3_f32_to_bf16(const float *x, void *y, int64_t n, void *stream) { return m3_cast_f32_dt(x, y, n, DT_BF16, stream); }

int m3_cast16(const void *x, void *y, int64_t n, int from_dtype, int to_dtype, void *stream) {
    M3_REQUIRE(x && y && n > 0 && n % 8 == 0);
    M3_DT_OK(from_dtype); M3_DT_OK(to_dtype);
    M3_REQUIRE(from_dtype != to_dtype);
    const dim3 grid(m3_cdiv(n / 8, kThreads)), blk(kThreads);
    if (from_dtype == DT_BF16)
        hipLaunchKernelGGL((k_cast16<DT_BF16, DT_F16>), grid, blk, 0, (hipStream_t)stream, (const bf16_t *)x, (bf16_t *)y, n / 8);
    else
        hipLaunchKernelGGL((k_cast16<DT_F16, DT_BF16>), grid, blk, 0, (hipStream_t)stream, (const bf16_t *)x, (bf16_t *)y, n / 8);
    M3_CHECK_LAUNCH("m3_cast16");
    return M3_OK;
}

int m3_relu_bf16(const void *x, void *y, int64_t n, void *stream) {
    M3_REQUIRE(x && y && n > 0 && n % 8 == 0);
    hipLaunchKernelGGL(k_relu_bf16, dim3(m3_cdiv(n / 8, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const bf16_t *)x, (bf16_t *)y, n / 8);
    M3_CHECK_LAUNCH("m3_relu_bf16");
    return M3_OK;
}

int m3_add_dt(const void *a, const void *b, void *y, int64_t n, int dtype, void *stream) {
    M3_REQUIRE(a && b && y && n > 0 && n % 8 == 0);
    M3_DT_OK(dtype);
    M3_DT_LAUNCH(dtype, k_add16, dim3(m3_cdiv(n / 8, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                 (const bf16_t *)a, (const bf16_t *)b, (bf16_t *)y, n / 8);
    M3_CHECK_LAUNCH("m3_add");
    return M3_OK;
}
int m3_add_bf16(const void *a, const void *b, void *y, int64_t n, void *stream) { return m3_add_dt(a, b, y, n, DT_BF16, stream); }

int m3_concat2_bf16(const void *a, const void *b, void *out, int64_t M, int Ca, int Cb, void *stream) {
    M3_REQUIRE(a && b && out && M > 0 && Ca > 0 && Cb > 0 && Ca % 8 == 0 && Cb % 8 == 0);
    const int64_t total = M * ((Ca + Cb) / 8);
    hipLaunchKernelGGL(k_concat2, dim3(m3_cdiv(total, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const bf16_t *)a, (const bf16_t *)b, (bf16_t *)out, M, Ca, Cb);
    M3_CHECK_LAUNCH("m3_concat2_bf16");
    return M3_OK;
}

int m3_unshuffle_bf16(const void *in, void *out, int B, int h, int w, int s, int C, int Cpad, void *stream) {
    M3_REQUIRE(in && out && B > 0 && h > 0 && w > 0 && s > 0 && C > 0 && C % 8 == 0 && Cpad >= C && Cpad % 8 == 0);
    const int64_t total = (int64_t)B * h * s * w * s * (C / 8);
    hipLaunchKernelGGL(k_unshuffle, dim3(m3_cdiv(total, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const bf16_t *)in, (bf16_t *)out, B, h, w, s, C, Cpad);
    M3_CHECK_LAUNCH("m3_unshuffle_bf16");
    return M3_OK;
}

int m3_upsample2x_dt(const void *in, void *out, int B, int H, int W, int C, int dtype, void *stream) {
    M3_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0);
    M3_DT_OK(dtype);
    const int64_t total = (int64_t)B * 2 * H * 2 * W * (C / 8);
    M3_DT_LAUNCH(dtype, k_upsample2x, dim3(m3_cdiv(total, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                 (const bf16_t *)in, (bf16_t *)out, B, H, W, C);
    M3_CHECK_LAUNCH("m3_upsample2x");
    return M3_OK;
}
int m3_upsample2x_bf16(const void *in, void *out, int B, int H, int W, int C, void *stream) {
    return m3_upsample2x_dt(in, out, B, H, W, C, DT_BF16, stream);
}

int m3_add_upsample2x_dt(const void *low, const void *y, void *out, int B, int H, int W, int OH, int OW, int C, int dtype,
                         void *stream) {
    M3_REQUIRE(low && y && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0);
    M3_REQUIRE(OH > 0 && OW > 0 && OH <= 2 * H && OW <= 2 * W);
    M3_DT_OK(dtype);
    const int64_t total = (int64_t)B * OH * OW * (C / 8);
    M3_DT_LAUNCH(dtype, k_add_upsample2x, dim3(m3_cdiv(total, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                 (const bf16_t *)low, (const bf16_t *)y, (bf16_t *)out, B, H, W, OH, OW, C);
    M3_CHECK_LAUNCH("m3_add_upsample2x");
    return M3_OK;
}

int m3_pts_post(const float *in, float *pts, float *conf, int64_t P, void *stream) {
    M3_REQUIRE(in && pts && conf && P > 0);
    hipLaunchKernelGGL(k_pts_post, dim3(m3_cdiv(P, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, in, pts, conf, P);
    M3_CHECK_LAUNCH("m3_pts_post");
    return M3_OK;
}

static int desc_post_launch(const void *in, void *desc, float *dconf, int B, int H, int W, int dtype, bool f16out, void *stream) {
    M3_REQUIRE(in && desc && dconf && B > 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0);
    M3_DT_OK(dtype);
    dim3 grid((unsigned)((int64_t)B * (H / 16) * (W / 16))), blk(kThreads);      // one 16 x 16 patch per workgroup
    hipStream_t st = (hipStream_t)stream;
#define M3_DP(DTV, F) hipLaunchKernelGGL((k_desc_post<DTV, F>), grid, blk, 0, st, (const bf16_t *)in, desc, dconf, B, H, W)
    if (dtype == DT_F16) { if (f16out) M3_DP(DT_F16, true); else M3_DP(DT_F16, false); }
    else { if (f16out) M3_DP(DT_BF16, true); else M3_DP(DT_BF16, false); }
#undef M3_DP
    M3_CHECK_LAUNCH("m3_desc_post");
    return M3_OK;
}
int m3_desc_post_dt(const void *in, float *desc, float *dconf, int B, int H, int W, int dtype, void *stream) {
    return desc_post_launch(in, desc, dconf, B, H, W, dtype, false, stream);
}
int m3_desc_post_f16(const void *in, void *desc_f16, float *dconf, int B, int H, int W, int dtype, void *stream) {
    return desc_post_launch(in, desc_f16, dconf, B, H, W, dtype, true, stream);
}
int m3_desc_post(const void *in, float *desc, float *dconf, int B, int H, int W, void *stream) {
    return m3_desc_post_dt(in, desc, dconf, B, H, W, DT_BF16, stream);
}


int m3_pack_fields(void *dst, const m3_pack_seg *segs, int nseg, void *stream) {
    M3_REQUIRE(dst && segs && nseg > 0 && nseg <= 8 && (reinterpret_cast<size_t>(dst) & 15) == 0);
    PackSegs p{};
    int64_t most = 0;
    for (int i = 0; i < nseg; ++i) {
        M3_REQUIRE(segs[i].src && segs[i].nbytes >= 0 && segs[i].dst_off >= 0 && segs[i].dst_off % 16 == 0 &&
                   (segs[i].mode == 0 || segs[i].mode == 1));
        M3_REQUIRE(segs[i].mode == 1 ? (segs[i].nbytes % 4 == 0 && (reinterpret_cast<size_t>(segs[i].src) & 7) == 0)
                                     : (reinterpret_cast<size_t>(segs[i].src) & 15) == 0);
        p.seg[i] = segs[i];
        const int64_t units = segs[i].mode == 0 ? (segs[i].nbytes >> 4) : (segs[i].nbytes >> 2);
        most = units > most ? units : most;
    }
    int64_t blocks = m3_cdiv(most, (int64_t)256 * 4);             // ~4 trips per thread on the largest segment
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(k_pack_fields, dim3((unsigned)blocks, (unsigned)nseg), dim3(256), 0, (hipStream_t)stream,
                       (unsigned char *)dst, p);
    M3_CHECK_LAUNCH("m3_pack_fields");
    return M3_OK;
}

}  // extern "C"
