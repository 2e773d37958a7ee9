// Dense matching stage for gfx950: prep_for_iter_proj, iter_proj, refine_matches,
// match epilogue, match_simple.   HBM/L2-gather bound integer+fp32 work: one
// thread per point, coalesced streaming of the per-point arrays, gathers served
// from L2 / Infinity Cache (the 9-channel ray image is 9.4 MB, the descriptor
// image 25 MB at 512x512 - both far below the 256 MiB Infinity Cache).
//
// BUILD WITH -ffp-contract=off: the arithmetic below is written so that every
// float32/float64 operation is individually rounded in exactly the order the
// CPU oracle (oracle/matching.py) uses; parity on p / valid / idx is bit-exact.
//
// Reference semantics (paths under /root/reference/src/mlx_mast3r_slam):
//   prep        matching.py:121-175, image.py:9-34
//   iter_proj   backends/mpsgraph/kernels.py:151-254 (numpy twin)
//   refine      backends/mpsgraph/kernels.py:496-537 (numpy twin)
//   epilogue    matching.py:436-461 ; match_simple matching.py:41-90
#include "common.h"
#include <hip/hip_fp16.h>
#include <limits.h>

namespace {

constexpr int kThreads = 256;

// ---------------------------------------------------------------- prep
__device__ __forceinline__ void unit_ray(const float *__restrict__ X, float &rx, float &ry, float &rz) {
    float x = X[0], y = X[1], z = X[2];
    float n2 = (x * x + y * y) + z * z;
    float nrm = sqrtf(n2 + 1e-10f);
    rx = x / nrm; ry = y / nrm; rz = z / nrm;
}

__global__ void __launch_bounds__(kThreads)
k_prep(const float *__restrict__ X11, const float *__restrict__ X21, const int64_t *__restrict__ idx_init,
       float *__restrict__ rwg, float *__restrict__ tgt, float *__restrict__ p_init, int H, int W) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * kThreads + threadIdx.x;
    const int N = H * W;
    if (n >= N) return;
    const int y = n / W, x = n - y * W;
    const size_t pix = (size_t)b * N + n;
    const float *base = X11 + (size_t)b * N * 3;
    float r0, r1, r2;
    unit_ray(base + (size_t)n * 3, r0, r1, r2);
    float gx0 = 0.f, gx1 = 0.f, gx2 = 0.f, gy0 = 0.f, gy1 = 0.f, gy2 = 0.f;
    if (x >= 1 && x <= W - 2) {
        float a0, a1, a2, c0, c1, c2;
        unit_ray(base + (size_t)(n + 1) * 3, a0, a1, a2);
        unit_ray(base + (size_t)(n - 1) * 3, c0, c1, c2);
        gx0 = (a0 - c0) / 2.0f; gx1 = (a1 - c1) / 2.0f; gx2 = (a2 - c2) / 2.0f;
    }
    if (y >= 1 && y <= H - 2) {
        float a0, a1, a2, c0, c1, c2;
        unit_ray(base + (size_t)(n + W) * 3, a0, a1, a2);
        unit_ray(base + (size_t)(n - W) * 3, c0, c1, c2);
        gy0 = (a0 - c0) / 2.0f; gy1 = (a1 - c1) / 2.0f; gy2 = (a2 - c2) / 2.0f;
    }
    float *o = rwg + pix * 9;
    o[0] = r0; o[1] = r1; o[2] = r2;
    o[3] = gx0; o[4] = gx1; o[5] = gx2;
    o[6] = gy0; o[7] = gy1; o[8] = gy2;
    float t0, t1, t2;
    unit_ray(X21 + pix * 3, t0, t1, t2);
    float *t = tgt + pix * 3;
    t[0] = t0; t[1] = t1; t[2] = t2;
    int64_t id = idx_init ? idx_init[pix] : (int64_t)n;
    // python semantics of % and // for (possibly negative) indices
    int64_t v = id / W, u = id - v * W;
    if (u < 0) { u += W; v -= 1; }
    p_init[pix * 2 + 0] = (float)u;
    p_init[pix * 2 + 1] = (float)v;
}

// ---------------------------------------------------------------- iter_proj
// 4-byte aligned vector loads: pixels are 36 B apart, so a pixel pair (72 B) is fetched with
// four dwordx4 + one dwordx2 instead of 18 dword requests (global loads only need dword alignment)
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) F2u { float x, y; };

__device__ __forceinline__ void load_pair(const float *__restrict__ p, float *a /*9*/, float *b /*9*/) {
    const F4u v0 = *reinterpret_cast<const F4u *>(p);
    const F4u v1 = *reinterpret_cast<const F4u *>(p + 4);
    const F4u v2 = *reinterpret_cast<const F4u *>(p + 8);
    const F4u v3 = *reinterpret_cast<const F4u *>(p + 12);
    const F2u v4 = *reinterpret_cast<const F2u *>(p + 16);
    a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w; a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w; a[8] = v2.x;
    b[0] = v2.y; b[1] = v2.z; b[2] = v2.w; b[3] = v3.x; b[4] = v3.y; b[5] = v3.z; b[6] = v3.w; b[7] = v4.x; b[8] = v4.y;
}

__device__ __forceinline__ float clipf(float v, float hi) {
    v = (v < 0.0f) ? 0.0f : v;     // NaN stays NaN, like np.clip
    v = (v > hi) ? hi : v;
    return v;
}
__device__ __forceinline__ float dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
    return (a0 * b0 + a1 * b1) + a2 * b2;
}

// FIRST = true : run max_iter LM steps, record per-iteration max step norm (as uint bits).
// FIRST = false: re-run with the per-batch iteration limit found by k_iter_limit; blocks
//                whose limit == max_iter have nothing to redo and exit immediately.
// Thread -> point mapping.  When the points are the pixels of the H x W image (N == H*W, H and W
// multiples of 16) a workgroup takes a 16x16 pixel tile instead of 256 consecutive pixels of a row:
// matches move smoothly over the image, so the gathers of a tile touch ~(16+r)^2 pixels of the other
// image instead of two or more 256-pixel strips - fewer L2 lines per workgroup and 2x less HBM-side
// fetch for iter_proj (PMC: 1.27 GB per 8-pair launch before).  Results are per point: unchanged bits.
__device__ __forceinline__ int point_of_thread(int blk, int t, int W, int tiled) {
    if (!tiled) return blk * kThreads + t;
    const int tiles_x = W >> 4, ty = blk / tiles_x, tx = blk - ty * tiles_x;
    return ((ty << 4) + (t >> 4)) * W + (tx << 4) + (t & 15);
}

template <bool FIRST>
__global__ void __launch_bounds__(kThreads)
k_iter_proj(const float *__restrict__ rwg, const float *__restrict__ tgt, const float *__restrict__ p_init,
            float *__restrict__ p_out, uint8_t *__restrict__ valid_out, uint32_t *__restrict__ stepmax,
            const uint32_t *__restrict__ limit, int H, int W, int N, int max_iter, float lam,
            float xhi, float yhi, int tiled) {
    const int b = blockIdx.y;
    int n_iter = max_iter;
    if (!FIRST) {
        n_iter = (int)limit[b];
        if (n_iter >= max_iter) return;
    }
    const int n = point_of_thread(blockIdx.x, threadIdx.x, W, tiled);
    const bool live = n < N;
    const size_t pt = (size_t)b * N + (live ? n : 0);
    const float *img = rwg + (size_t)b * H * W * 9;
    float px = p_init[pt * 2 + 0], py = p_init[pt * 2 + 1];
    const float t0 = tgt[pt * 3 + 0], t1 = tgt[pt * 3 + 1], t2 = tgt[pt * 3 + 2];

    for (int it = 0; it < n_iter; ++it) {
        const float cx = clipf(px, xhi), cy = clipf(py, yhi);
        int x0 = (int)floorf(cx), y0 = (int)floorf(cy);
        x0 = min(max(x0, 0), W - 1); y0 = min(max(y0, 0), H - 1);   // only matters for NaN input
        const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
        const double fx = (double)cx - (double)x0, fy = (double)cy - (double)y0;
        const double w00 = (1.0 - fx) * (1.0 - fy), w01 = (1.0 - fx) * fy;
        const double w10 = fx * (1.0 - fy), w11 = fx * fy;
        float c00[9], c10[9], c01[9], c11[9];             // corner (x0,y0), (x1,y0), (x0,y1), (x1,y1)
        const float *row0 = img + ((size_t)y0 * W + x0) * 9, *row1 = img + ((size_t)y1 * W + x0) * 9;
        if (x1 == x0 + 1) {                                // the two corners of a row are 72 contiguous bytes
            load_pair(row0, c00, c10);
            load_pair(row1, c01, c11);
        } else {                                           // x0 clamped at the last column: both corners coincide
#pragma unroll
            for (int c = 0; c < 9; ++c) { c00[c] = row0[c]; c10[c] = row0[c]; c01[c] = row1[c]; c11[c] = row1[c]; }
        }
        float s[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            double v = ((w00 * (double)c00[c] + w01 * (double)c01[c]) + w10 * (double)c10[c])
                       + w11 * (double)c11[c];
            s[c] = (float)v;
        }
        const float r0 = s[0] - t0, r1 = s[1] - t1, r2 = s[2] - t2;
        const float a = dot3(s[3], s[4], s[5], s[3], s[4], s[5]) + lam;
        const float bb = dot3(s[3], s[4], s[5], s[6], s[7], s[8]);
        const float c = dot3(s[6], s[7], s[8], s[3], s[4], s[5]);
        const float d = dot3(s[6], s[7], s[8], s[6], s[7], s[8]) + lam;
        const float j0 = dot3(s[3], s[4], s[5], r0, r1, r2);
        const float j1 = dot3(s[6], s[7], s[8], r0, r1, r2);
        float det = a * d - bb * c;
        det = (det < 1e-10f) ? 1e-10f : det;
        const float inv_det = 1.0f / det;
        const float dx = -(d * j0 - bb * j1) * inv_det;
        const float dy = -((-c) * j0 + a * j1) * inv_det;
        px = px + dx;
        py = py + dy;
        if (FIRST) {
            // per-block maximum of the step norm (as uint bits), one plain store per block and iteration:
            // thousands of atomics on ONE address per iteration serialise (~12 ns each) and used to cost 10x
            // the arithmetic of this kernel
            // (round 4: per WAVE - the workgroup-level maximum cost two barriers per LM iteration, i.e. the four waves of
            // a workgroup walked their data-dependent gathers in lock step)
            const float dn = sqrtf(dx * dx + dy * dy);
            unsigned bits = live ? (__float_as_uint(dn) & 0x7fffffffu) : 0u;
            bits = m3_wave_max(bits);
            if ((threadIdx.x & 63) == 0)
                stepmax[((size_t)b * max_iter + it) * (gridDim.x * (kThreads / 64)) + blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)] = bits;
        }
    }
    if (live) {
        p_out[pt * 2 + 0] = clipf(px, (float)(W - 1));
        p_out[pt * 2 + 1] = clipf(py, (float)(H - 1));
        valid_out[pt] = (px >= 0.0f) && (px < (float)W) && (py >= 0.0f) && (py < (float)H);
    }
}

// Early-stop limit in two small launches (one workgroup doing all B*max_iter reductions took 61 us for 8
// maps): k_iter_reduce - one wave per (batch item, iteration) reduces the per-block maxima; k_iter_limit -
// one wave finds the first LM iteration whose max step norm is < thresh -> number of iterations to run.
__global__ void __launch_bounds__(kThreads)
k_iter_reduce(const uint32_t *__restrict__ stepmax, uint32_t *__restrict__ red, int total, int nblk) {
    const int i = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (i >= total) return;
    unsigned m = 0u;
    for (int k = threadIdx.x & 63; k < nblk; k += 64) m = max(m, stepmax[(size_t)i * nblk + k]);
    m = m3_wave_max(m);
    if ((threadIdx.x & 63) == 0) red[i] = m;
}

__global__ void __launch_bounds__(64)
k_iter_limit(const uint32_t *__restrict__ red, uint32_t *__restrict__ limit, int B, int max_iter, float thresh,
             int per_batch) {
    if (threadIdx.x != 0) return;
    if (per_batch) {
        for (int b = 0; b < B; ++b) {
            int lim = max_iter;
            for (int it = 0; it < max_iter; ++it)
                if (__uint_as_float(red[b * max_iter + it]) < thresh) { lim = it + 1; break; }
            limit[b] = (uint32_t)lim;
        }
    } else {
        int lim = max_iter;
        for (int it = 0; it < max_iter; ++it) {
            unsigned m = 0;
            for (int b = 0; b < B; ++b) m = max(m, red[b * max_iter + it]);
            if (__uint_as_float(m) < thresh) { lim = it + 1; break; }
        }
        for (int b = 0; b < B; ++b) limit[b] = (uint32_t)lim;
    }
}

// ---------------------------------------------------------------- refine_matches
// score = sequential sum_d (q[d]*r[d]) (mul, then add; contraction is off), strict '>' in
// (dy outer, dx inner) raster order, out-of-bounds candidates skipped.
//
// Descriptor storage TD = float or __half ("fp16 features", BASELINE configs[4]): half descriptors are widened to
// fp32 exactly (every fp16 value is an fp32 value) and the arithmetic is the same fp32 sequence, so the result is
// bit-identical to the fp32 path - and to the oracle - run on the half-rounded descriptors; HBM traffic halves
// (29.3 instead of 54.5 MB per 512x512 map).
template <typename TD> struct DescIO;
template <> struct DescIO<float> {
    static constexpr int kVec = 4;                             // elements per 16-byte load
    __device__ static __forceinline__ void load16(const float *p, float *o) {
        const float4 v = *reinterpret_cast<const float4 *>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    __device__ static __forceinline__ float load1(const float *p) { return *p; }
};
template <> struct DescIO<__half> {
    static constexpr int kVec = 8;
    __device__ static __forceinline__ void load16(const __half *p, float *o) {
        const uint4 u = *reinterpret_cast<const uint4 *>(p);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __half2float(__ushort_as_half((unsigned short)(w[i] & 0xffffu)));
            o[2 * i + 1] = __half2float(__ushort_as_half((unsigned short)(w[i] >> 16)));
        }
    }
    __device__ static __forceinline__ float load1(const __half *p) { return __half2float(*p); }
};

template <int D, typename TD>
__device__ __forceinline__ void load_desc(const TD *__restrict__ p, float *q) {
    constexpr int V = DescIO<TD>::kVec;
#pragma unroll
    for (int k = 0; k < D / V; ++k) DescIO<TD>::load16(p + k * V, q + k * V);
}

template <int D, typename TD>
__device__ __forceinline__ void refine_pass(const TD *__restrict__ img, const float *q, int H, int W,
                                            int radius, int dil, int cx, int cy, int &bx, int &by) {
    float best = -INFINITY;
    bx = cx; by = cy;
    for (int dy = -radius; dy <= radius; ++dy) {
        const int ny = cy + dy * dil;
        if (ny < 0 || ny >= H) continue;
        for (int dx = -radius; dx <= radius; ++dx) {
            const int nx = cx + dx * dil;
            if (nx < 0 || nx >= W) continue;
            float r[D];
            load_desc<D, TD>(img + ((size_t)ny * W + nx) * D, r);
            float score = 0.0f;
#pragma unroll
            for (int k = 0; k < D; ++k) score = score + q[k] * r[k];
            if (score > best) { best = score; bx = nx; by = ny; }
        }
    }
}

template <int D, typename TD>
__global__ void __launch_bounds__(kThreads)
k_refine(const TD *__restrict__ D11, const TD *__restrict__ D21, const int32_t *__restrict__ p_in,
         int32_t *__restrict__ p_out, int H, int W, int N, int radius, int dil_max, int chained, int tiled) {
    const int b = blockIdx.y;
    const int n = point_of_thread(blockIdx.x, threadIdx.x, W, tiled);
    if (n >= N) return;
    const size_t pt = (size_t)b * N + n;
    const TD *img = D11 + (size_t)b * H * W * D;
    float q[D];
    load_desc<D, TD>(D21 + pt * D, q);
    int cx = p_in[pt * 2 + 0], cy = p_in[pt * 2 + 1];
    int bx = cx, by = cy;
    if (chained) {
        for (int dil = dil_max; dil >= 1; --dil) {
            refine_pass<D, TD>(img, q, H, W, radius, dil, cx, cy, bx, by);
            cx = bx; cy = by;
        }
    } else {
        refine_pass<D, TD>(img, q, H, W, radius, 1, cx, cy, bx, by);
    }
    p_out[pt * 2 + 0] = bx;
    p_out[pt * 2 + 1] = by;
}

// Single-pass (dilation 1) refinement with the candidate descriptors staged in LDS.  A workgroup owns a
// 16x16 pixel tile of view 2; on real (smooth) geometry its matches fall into a compact region of view 1,
// so the region [min - r, max + r] (clamped to the image) is copied once into LDS and every thread reads
// its 49 candidates from there instead of issuing 49 x 6 global float4 gathers (the L1-bound part: 9.9 GB
// of gather traffic for 8 maps).  Same candidates, same order, same arithmetic -> same bits.  Tiles whose
// region does not fit (scattered matches) take the global path.  LDS pixel stride D + 4 floats (112 B for
// D = 24) keeps neighbouring pixels on different banks.  Half descriptors are widened while staging: the
// LDS image and the inner loop are the fp32 ones.
constexpr int kRefineLdsBytes = 64 * 1024;

// R > 0: the radius is a compile-time constant and the 2R+1 candidates of a window row are scored TOGETHER: 2R+1
// independent accumulation chains instead of one 2*D-deep dependent mul/add chain per candidate (with 2 waves per SIMD
// that chain's latency was the kernel's time), no per-candidate bounds branches (out-of-image candidates read a clamped
// address and are masked at selection), no index multiply per candidate.  Each score is still the same sequential
// fp32 sum and the selection still walks the candidates in raster order with a strict '>': identical bits.
// DMA (round 4, fp32 descriptors only - half ones are widened on the way in): the candidate region goes global -> LDS by
// LDS-DMA (global_load_lds, 16 bytes per lane, no register round trip, every piece of the workgroup in flight at once and
// ONE wait) instead of load -> wait -> ds_write through VGPRs.  The LDS image is unchanged (7 sixteen-byte slots per
// pixel, the seventh is padding): a wave-instruction fills 64 consecutive slots, slot s = (pixel s / 7, chunk s % 7), the
// padding slots re-read chunk 5.  Same image, same inner loop: same bits.
template <int D, typename TD, int R, bool DMA = false>
__global__ void __launch_bounds__(kThreads, 2)
k_refine_lds(const TD *__restrict__ D11, const TD *__restrict__ D21, const int32_t *__restrict__ p_in,
             int32_t *__restrict__ p_out, int H, int W, int N, int radius) {
    extern __shared__ float4 tile[];
    __shared__ int bb[4];
    constexpr int PS4 = D / 4 + 1;                          // float4 per staged pixel
    constexpr int kMaxPix = kRefineLdsBytes / (PS4 * 16);
    constexpr int V = DescIO<TD>::kVec, CH = D / V;         // 16-byte global chunks per pixel
    const int b = blockIdx.y;
    // Lane -> pixel inside the wave's 4 x 16 pixel strip.  A ds_read_b128 is served in four groups of 16 lanes,
    // {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32 (MI355X_MICROARCH.md, LDS table); with the staged pixel
    // stride of 7 sixteen-byte slots a group is conflict-free when its 16 pixels are distinct mod 16.  The natural map
    // (lane = 16 row + x) puts half a group in one image row and half in the next: their candidates are rw + 8 pixels
    // apart and collide whenever the staged width rw is not a multiple of 16 - PMC: SQ_LDS_BANK_CONFLICT = 3.2 x
    // SQ_ACTIVE_INST_LDS, 58 % of the wave cycles waiting.  Here every group owns ONE row of 16 consecutive pixels, so
    // for locally translational matches (the compact-region case this kernel exists for) its candidates are 16
    // consecutive staged pixels.  Same points, same arithmetic per point: same bits.
    const int lane = threadIdx.x & 63, m = lane & 31;
    const bool grpB = (m >= 4 && m < 12) || (m >= 16 && m < 20) || m >= 28;
    const int xin = m < 4 ? m : m < 12 ? m - 4 : m < 16 ? m - 8 : m < 20 ? m - 8 : m < 28 ? m - 12 : m - 16;
    const int t_perm = (threadIdx.x & ~63) + ((((lane >> 5) << 1) + (grpB ? 1 : 0)) << 4) + xin;
    const int n = point_of_thread(blockIdx.x, t_perm, W, 1);           // N == H*W: every thread has a point
    const size_t pt = (size_t)b * N + n;
    const TD *img = D11 + (size_t)b * H * W * D;
    float q[D];
    load_desc<D, TD>(D21 + pt * D, q);
    const int cx = p_in[pt * 2 + 0], cy = p_in[pt * 2 + 1];
    if (threadIdx.x == 0) { bb[0] = INT_MAX; bb[1] = INT_MIN; bb[2] = INT_MAX; bb[3] = INT_MIN; }
    __syncthreads();
    int mnx = cx, mxx = cx, mny = cy, mxy = cy;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mnx = min(mnx, __shfl_down(mnx, off, 64)); mxx = max(mxx, __shfl_down(mxx, off, 64));
        mny = min(mny, __shfl_down(mny, off, 64)); mxy = max(mxy, __shfl_down(mxy, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&bb[0], mnx); atomicMax(&bb[1], mxx); atomicMin(&bb[2], mny); atomicMax(&bb[3], mxy);
    }
    __syncthreads();
    const long long lx0 = (long long)bb[0] - radius, lx1 = (long long)bb[1] + radius;
    const long long ly0 = (long long)bb[2] - radius, ly1 = (long long)bb[3] + radius;
    const int x0 = (int)(lx0 < 0 ? 0 : lx0), x1 = (int)(lx1 > W - 1 ? W - 1 : lx1);
    const int y0 = (int)(ly0 < 0 ? 0 : ly0), y1 = (int)(ly1 > H - 1 ? H - 1 : ly1);
    const int rw = x1 - x0 + 1, rh = y1 - y0 + 1;
    int bx = cx, by = cy;
    if (rw > 0 && rh > 0 && (long long)rw * rh <= kMaxPix) {            // uniform over the workgroup
        // Staging, four 16-byte pieces per thread IN FLIGHT: the first form loaded, waited and stored one piece per loop
        // trip - ~14 dependent L2 round trips per workgroup, ~40 % of the kernel (profiles/r03_matcher_pmc.md: the fp16
        // variant, which stages half the pieces, was 50-65 us faster for that reason alone).
        const int per_row = rw * CH, total = rh * per_row;
        constexpr int U = 4;                                   // (8 in flight measured the same)
        if constexpr (DMA) {
            static_assert(DescIO<TD>::kVec == 4, "LDS-DMA staging copies bytes: fp32 descriptors only");
            const int slots = rw * rh * PS4, nslots = (slots + 63) & ~63;      // <= 4096 = the 64 KiB of the tile
            for (int i0 = threadIdx.x; i0 < nslots; i0 += kThreads) {
                const int sl = i0 < slots ? i0 : slots - 1;
                const int pix = sl / PS4, k = sl - pix * PS4;
                const int ry = pix / rw, rx = pix - ry * rw;
                const TD *src = img + ((size_t)(y0 + ry) * W + x0 + rx) * D + (k < CH ? k : CH - 1) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned *)src,
                                                 (__attribute__((address_space(3))) unsigned *)(tile + (i0 - lane)), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else
        for (int i0 = threadIdx.x; i0 < total; i0 += kThreads * U) {
            float v[U][V];
            int dofs[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * kThreads;
                const int ic = i < total ? i : total - 1;             // clamped: the load is harmless, the store is skipped
                const int ry = ic / per_row, rem = ic - ry * per_row;
                const int rx = rem / CH, k = rem - rx * CH;
                DescIO<TD>::load16(img + ((size_t)(y0 + ry) * W + x0 + rx) * D + k * V, v[u]);
                dofs[u] = (ry * rw + rx) * PS4 + k * (V / 4);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (i0 + u * kThreads >= total) continue;
                float4 *dst = tile + dofs[u];
#pragma unroll
                for (int j = 0; j < V / 4; ++j) dst[j] = make_float4(v[u][4 * j], v[u][4 * j + 1], v[u][4 * j + 2], v[u][4 * j + 3]);
            }
        }
        __syncthreads();
        float best = -INFINITY;
        if constexpr (R > 0) {
            // Software pipeline over HALF window rows (3 of the 6 sixteen-byte chunks of each of the 2R+1 candidates): the
            // reads of the next half are issued before the arithmetic of the current one and pinned there.  Left to the
            // scheduler (which keeps the register count low for an occupancy the 64 KiB of LDS per workgroup forbids
            // anyway: 2 waves per SIMD), chunks 4 and 5 of every candidate were read one at a time with a full
            // s_waitcnt lgkmcnt(0) in front of their 8 instructions - 14 exposed LDS round trips per window row, 60 % of
            // the wave cycles in s_waitcnt (PMC, profiles/r03_matcher_pmc.md).  168 registers of staging instead.
            constexpr int NC = 2 * R + 1, HK = D / 8;             // candidates per row, chunks per half (D = 24: 3)
            static_assert(D % 8 == 0, "half rows need an even chunk count");
            float4 ra[HK][NC], rb[HK][NC];
            int off[NC];                                            // float4 offsets of the row's candidates in the tile
            auto row_offsets = [&](int dy) {
                const int ny = cy + dy;
                const int nyc = ny < y0 ? y0 : (ny > y1 ? y1 : ny);
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    const int nx = cx + j - R;
                    const int nxc = nx < x0 ? x0 : (nx > x1 ? x1 : nx);
                    off[j] = ((nyc - y0) * rw + (nxc - x0)) * PS4;
                }
            };
            auto issue = [&](float4 (&dst)[HK][NC], int half) {
#pragma unroll
                for (int kk = 0; kk < HK; ++kk)
#pragma unroll
                    for (int j = 0; j < NC; ++j) dst[kk][j] = tile[off[j] + half * HK + kk];
                __builtin_amdgcn_sched_barrier(0);
            };
            auto accumulate = [&](const float4 (&src)[HK][NC], int half, float (&sc)[NC]) {
#pragma unroll
                for (int kk = 0; kk < HK; ++kk) {
                    const int k = half * HK + kk;
#pragma unroll
                    for (int j = 0; j < NC; ++j) {
                        const float4 v = src[kk][j];
                        sc[j] = sc[j] + q[4 * k + 0] * v.x;
                        sc[j] = sc[j] + q[4 * k + 1] * v.y;
                        sc[j] = sc[j] + q[4 * k + 2] * v.z;
                        sc[j] = sc[j] + q[4 * k + 3] * v.w;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            row_offsets(-R);
            issue(ra, 0);
#pragma unroll
            for (int dy = -R; dy <= R; ++dy) {
                const int ny = cy + dy;
                const bool row_ok = (ny >= 0) && (ny < H);
                float sc[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j) sc[j] = 0.0f;
                issue(rb, 1);                                      // second half of this row: in flight under the first half's sums
                accumulate(ra, 0, sc);
                if (dy < R) { row_offsets(dy + 1); issue(ra, 0); } // first half of the next row: in flight under this row's second half
                accumulate(rb, 1, sc);
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    const int nx = cx + j - R;
                    if (row_ok && nx >= 0 && nx < W && sc[j] > best) { best = sc[j]; bx = nx; by = ny; }
                }
            }
        } else {
        for (int dy = -radius; dy <= radius; ++dy) {
            const int ny = cy + dy;
            if (ny < 0 || ny >= H) continue;
            for (int dx = -radius; dx <= radius; ++dx) {
                const int nx = cx + dx;
                if (nx < 0 || nx >= W) continue;
                const float4 *r4 = tile + ((ny - y0) * rw + (nx - x0)) * PS4;
                float score = 0.0f;
#pragma unroll
                for (int k = 0; k < D / 4; ++k) {
                    const float4 v = r4[k];
                    score = score + q[4 * k + 0] * v.x;
                    score = score + q[4 * k + 1] * v.y;
                    score = score + q[4 * k + 2] * v.z;
                    score = score + q[4 * k + 3] * v.w;
                }
                if (score > best) { best = score; bx = nx; by = ny; }
            }
        }
        }
    } else {
        refine_pass<D, TD>(img, q, H, W, radius, 1, cx, cy, bx, by);
    }
    p_out[pt * 2 + 0] = bx;
    p_out[pt * 2 + 1] = by;
}

// generic descriptor length (any D >= 1): query re-read from global (L1-resident)
template <typename TD>
__global__ void __launch_bounds__(kThreads)
k_refine_generic(const TD *__restrict__ D11, const TD *__restrict__ D21,
                 const int32_t *__restrict__ p_in, int32_t *__restrict__ p_out, int H, int W, int D, int N,
                 int radius, int dil_max, int chained) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * kThreads + threadIdx.x;
    if (n >= N) return;
    const size_t pt = (size_t)b * N + n;
    const TD *img = D11 + (size_t)b * H * W * D;
    const TD *q = D21 + pt * D;
    int cx = p_in[pt * 2 + 0], cy = p_in[pt * 2 + 1];
    int bx = cx, by = cy;
    const int first = chained ? dil_max : 1;
    for (int dil = first; dil >= 1; --dil) {
        float best = -INFINITY;
        bx = cx; by = cy;
        for (int dy = -radius; dy <= radius; ++dy) {
            const int ny = cy + dy * dil;
            if (ny < 0 || ny >= H) continue;
            for (int dx = -radius; dx <= radius; ++dx) {
                const int nx = cx + dx * dil;
                if (nx < 0 || nx >= W) continue;
                const TD *r = img + ((size_t)ny * W + nx) * D;
                float score = 0.0f;
                for (int k = 0; k < D; ++k) score = score + DescIO<TD>::load1(q + k) * DescIO<TD>::load1(r + k);
                if (score > best) { best = score; bx = nx; by = ny; }
            }
        }
        cx = bx; cy = by;
    }
    p_out[pt * 2 + 0] = bx;
    p_out[pt * 2 + 1] = by;
}

// ---------------------------------------------------------------- epilogue / simple
__global__ void __launch_bounds__(kThreads)
k_epilogue(const float *__restrict__ X11, const float *__restrict__ X21, const int32_t *__restrict__ p_i32,
           const float *__restrict__ p_f32, const uint8_t *__restrict__ valid_proj,
           int64_t *__restrict__ idx_out, uint8_t *__restrict__ valid_out, int H, int W, float thresh) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * kThreads + threadIdx.x;
    const int N = H * W;
    if (n >= N) return;
    const size_t pt = (size_t)b * N + n;
    int u, v;
    if (p_i32) { u = p_i32[pt * 2]; v = p_i32[pt * 2 + 1]; }
    else { u = (int)p_f32[pt * 2]; v = (int)p_f32[pt * 2 + 1]; }
    const int xc = min(max(u, 0), W - 1), yc = min(max(v, 0), H - 1);
    const float *a = X11 + ((size_t)b * N + (size_t)yc * W + xc) * 3;
    const float *c = X21 + pt * 3;
    const float d0 = a[0] - c[0], d1 = a[1] - c[1], d2 = a[2] - c[2];
    const float dist = sqrtf((d0 * d0 + d1 * d1) + d2 * d2);
    valid_out[pt] = (valid_proj[pt] != 0) && (dist < thresh);
    idx_out[pt] = (int64_t)u + (int64_t)W * (int64_t)v;
}

__global__ void __launch_bounds__(kThreads)
k_match_simple(const float *__restrict__ X11, const float *__restrict__ X21,
               const int64_t *__restrict__ idx_init, int64_t *__restrict__ idx_out,
               uint8_t *__restrict__ valid_out, int N, float thresh) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * kThreads + threadIdx.x;
    if (n >= N) return;
    const size_t pt = (size_t)b * N + n;
    int64_t id = idx_init ? idx_init[pt] : (int64_t)n;
    idx_out[pt] = id;
    if (id < 0) id += N;                       // numpy/mlx negative indexing
    id = id < 0 ? 0 : (id >= N ? N - 1 : id);  // never fault on bad input
    const float *a = X11 + ((size_t)b * N + (size_t)id) * 3;
    const float *c = X21 + pt * 3;
    const float d0 = a[0] - c[0], d1 = a[1] - c[1], d2 = a[2] - c[2];
    valid_out[pt] = sqrtf((d0 * d0 + d1 * d1) + d2 * d2) < thresh;
}

__global__ void __launch_bounds__(kThreads)
k_trunc(const float *__restrict__ p, int32_t *__restrict__ out, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < count) out[i] = (int32_t)p[i];
}

}  // namespace

// ================================================================== C ABI
extern "C" {

int m3_prep_iter_proj(const float *X11, const float *X21, const int64_t *idx_init, float *rwg,
                      float *tgt, float *p_init, int B, int H, int W, void *stream) {
    M3_REQUIRE(X11 && X21 && rwg && tgt && p_init);
    M3_REQUIRE(B > 0 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 31) && B <= 65535);
    dim3 grid(m3_cdiv((int64_t)H * W, kThreads), B);
    hipLaunchKernelGGL(k_prep, grid, dim3(kThreads), 0, (hipStream_t)stream, X11, X21, idx_init, rwg, tgt,
                       p_init, H, W);
    M3_CHECK_LAUNCH("m3_prep_iter_proj");
    return M3_OK;
}

int64_t m3_iter_proj_ws_words(int B, int N, int max_iter) {
    if (B <= 0 || N <= 0 || max_iter < 0) return 0;
    return (int64_t)B * max_iter * m3_cdiv(N, kThreads) * (kThreads / 64) + B + (int64_t)B * max_iter;   // per-wave step maxima
}

int m3_iter_proj(const float *rwg, const float *tgt, const float *p_init, float *p_out,
                 uint8_t *valid_out, uint32_t *ws, int B, int H, int W, int N, int max_iter,
                 float lambda_init, float convergence_thresh, int stop_scope, void *stream) {
    M3_REQUIRE(rwg && tgt && p_init && p_out && valid_out && ws);
    M3_REQUIRE(B > 0 && H > 0 && W > 0 && N > 0 && max_iter >= 0 && B <= 65535);
    M3_REQUIRE((int64_t)H * W < (1ll << 31) && (stop_scope == 0 || stop_scope == 1));
    hipStream_t st = (hipStream_t)stream;
    const int nblk = m3_cdiv(N, kThreads), nwav = nblk * (kThreads / 64);
    uint32_t *stepmax = ws, *limit = ws + (size_t)B * max_iter * nwav, *red = limit + B;
    const float xhi = (float)((double)W - 1.001), yhi = (float)((double)H - 1.001);
    dim3 grid(nblk, B);
    const int tiled = (N == H * W && H % 16 == 0 && W % 16 == 0) ? 1 : 0;
    hipLaunchKernelGGL(k_iter_proj<true>, grid, dim3(kThreads), 0, st, rwg, tgt, p_init, p_out, valid_out,
                       stepmax, (const uint32_t *)limit, H, W, N, max_iter, lambda_init, xhi, yhi, tiled);
    M3_CHECK_LAUNCH("m3_iter_proj/pass1");
    if (max_iter > 1) {
        hipLaunchKernelGGL(k_iter_reduce, dim3(m3_cdiv(B * max_iter, kThreads / 64)), dim3(kThreads), 0, st,
                           (const uint32_t *)stepmax, red, B * max_iter, nwav);
        hipLaunchKernelGGL(k_iter_limit, dim3(1), dim3(64), 0, st, (const uint32_t *)red, limit, B, max_iter,
                           convergence_thresh, stop_scope);
        M3_CHECK_LAUNCH("m3_iter_proj/limit");
        hipLaunchKernelGGL(k_iter_proj<false>, grid, dim3(kThreads), 0, st, rwg, tgt, p_init, p_out,
                           valid_out, stepmax, (const uint32_t *)limit, H, W, N, max_iter, lambda_init, xhi, yhi, tiled);
        M3_CHECK_LAUNCH("m3_iter_proj/pass2");
    }
    return M3_OK;
}

}  // extern "C"

namespace {
template <typename TD>
int refine_launch(const TD *D11, const TD *D21, const int32_t *p_in, int32_t *p_out, int B, int H, int W, int D,
                  int N, int radius, int dilation_max, int chained, hipStream_t st) {
    M3_REQUIRE(D11 && D21 && p_in && p_out && p_in != p_out);
    M3_REQUIRE(B > 0 && H > 0 && W > 0 && D > 0 && N > 0 && radius >= 0 && B <= 65535);
    M3_REQUIRE((int64_t)H * W < (1ll << 31));
    const int dmax = dilation_max < 1 ? 1 : dilation_max;
    dim3 grid(m3_cdiv(N, kThreads), B), blk(kThreads);
    // vector path: 16-byte loads need aligned bases and a pixel stride that is a multiple of 16 bytes
    const bool aligned = (((uintptr_t)D11 | (uintptr_t)D21) & 15) == 0;
    const int tiled = (N == H * W && H % 16 == 0 && W % 16 == 0) ? 1 : 0;
#define M3_REFINE(DD) hipLaunchKernelGGL((k_refine<DD, TD>), grid, blk, 0, st, D11, D21, p_in, p_out, H, W, N, radius, dmax, chained, tiled)
    const bool single_pass = !chained || dmax == 1;
    if (aligned && D == 24 && tiled && single_pass && radius <= 4) {
        static const bool dma = [] { const char *e = getenv("M3_REFINE_DMA"); return !(e && atoi(e) == 0); }();
        if constexpr (DescIO<TD>::kVec == 4) {
            if (dma && radius == 3) {
                hipLaunchKernelGGL((k_refine_lds<24, TD, 3, true>), grid, blk, kRefineLdsBytes, st, D11, D21, p_in, p_out, H, W, N, radius);
                M3_CHECK_LAUNCH("m3_refine_matches");
                return M3_OK;
            }
        }
        if (radius == 3) hipLaunchKernelGGL((k_refine_lds<24, TD, 3>), grid, blk, kRefineLdsBytes, st, D11, D21, p_in, p_out, H, W, N, radius);
        else hipLaunchKernelGGL((k_refine_lds<24, TD, 0>), grid, blk, kRefineLdsBytes, st, D11, D21, p_in, p_out, H, W, N, radius);
    } else if (aligned && D == 24) M3_REFINE(24);
    else if (aligned && D == 16) M3_REFINE(16);
    else if (aligned && D == 32) M3_REFINE(32);
    else if (aligned && D == 64) M3_REFINE(64);
    else hipLaunchKernelGGL((k_refine_generic<TD>), grid, blk, 0, st, D11, D21, p_in, p_out, H, W, D, N, radius, dmax, chained);
#undef M3_REFINE
    M3_CHECK_LAUNCH("m3_refine_matches");
    return M3_OK;
}
}  // namespace

extern "C" {

int m3_refine_matches(const float *D11, const float *D21, const int32_t *p_in, int32_t *p_out, int B,
                      int H, int W, int D, int N, int radius, int dilation_max, int chained, void *stream) {
    return refine_launch<float>(D11, D21, p_in, p_out, B, H, W, D, N, radius, dilation_max, chained, (hipStream_t)stream);
}
int m3_refine_matches_f16(const void *D11, const void *D21, const int32_t *p_in, int32_t *p_out, int B,
                          int H, int W, int D, int N, int radius, int dilation_max, int chained, void *stream) {
    return refine_launch<__half>((const __half *)D11, (const __half *)D21, p_in, p_out, B, H, W, D, N, radius,
                                 dilation_max, chained, (hipStream_t)stream);
}

int m3_match_epilogue(const float *X11, const float *X21, const int32_t *p_i32, const float *p_f32,
                      const uint8_t *valid_proj, int64_t *idx_out, uint8_t *valid_out, int B, int H, int W,
                      float dist_thresh, void *stream) {
    M3_REQUIRE(X11 && X21 && (p_i32 || p_f32) && valid_proj && idx_out && valid_out);
    M3_REQUIRE(B > 0 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 31) && B <= 65535);
    dim3 grid(m3_cdiv((int64_t)H * W, kThreads), B);
    hipLaunchKernelGGL(k_epilogue, grid, dim3(kThreads), 0, (hipStream_t)stream, X11, X21, p_i32, p_f32,
                       valid_proj, idx_out, valid_out, H, W, dist_thresh);
    M3_CHECK_LAUNCH("m3_match_epilogue");
    return M3_OK;
}

int m3_match_simple(const float *X11, const float *X21, const int64_t *idx_init, int64_t *idx_out,
                    uint8_t *valid_out, int B, int H, int W, float dist_thresh, void *stream) {
    M3_REQUIRE(X11 && X21 && idx_out && valid_out);
    M3_REQUIRE(B > 0 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 31) && B <= 65535);
    dim3 grid(m3_cdiv((int64_t)H * W, kThreads), B);
    hipLaunchKernelGGL(k_match_simple, grid, dim3(kThreads), 0, (hipStream_t)stream, X11, X21, idx_init,
                       idx_out, valid_out, H * W, dist_thresh);
    M3_CHECK_LAUNCH("m3_match_simple");
    return M3_OK;
}

int m3_trunc_i32(const float *p, int32_t *out, int64_t count, void *stream) {
    M3_REQUIRE(p && out && count > 0);
    hipLaunchKernelGGL(k_trunc, dim3(m3_cdiv(count, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, p, out, count);
    M3_CHECK_LAUNCH("m3_trunc_i32");
    return M3_OK;
}

}  // extern "C"
