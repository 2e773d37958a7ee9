// Brute-force nearest neighbour in descriptor space (maximum dot product) - the search primitive of
// "fast reciprocal NN" matching (MASt3R, Leroy et al. 2024, section 3.3; mast3r/fast_nn.py of the public
// implementation).  BASELINE.json's north_star names this matcher; the reference tree has no implementation
// of it, so the semantics below are this repo's and are pinned by its own oracle:
//
//   score(s, n) = E + O,  E = fma chain over even k (ascending), O = fma chain over odd k (ascending), fp32
//   idx[s] = argmax_n score(s, n), ties -> the LOWEST n; score_out[s] = that maximum.
// (two interleaved chains = one v_pk_fma_f32 per pair of dimensions: twice the plain-FMA rate)
//
// Work layout: a lane owns one query (its D floats live in registers); database entries are wave-uniform,
// so they are fetched with scalar loads and feed the FMAs as SGPR operands - no LDS, no cross-lane
// reduction.  The database is split over blockIdx.y (so that a 4096-query search still fills the chip); the
// per-split winners are merged with ONE 64-bit atomicMax per query on a key = (order-preserving score bits,
// inverted index), which also implements the lowest-index tie-break deterministically.
#include "common.h"
#include "../../include/m3slam.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ unsigned ordered_bits(float f) {      // monotone float -> uint map
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_ordered(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

constexpr int kQPL = 2;                                       // queries per lane: halves the scalar row traffic per FMA

template <int D>
__global__ void __launch_bounds__(kThreads)
k_nn_search(const float *__restrict__ Q, const float *__restrict__ DB, unsigned long long *__restrict__ keys,
            int S, int N, int per_split) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int b = blockIdx.z;
    f32x2 q[kQPL][D / 2];
    int sidx[kQPL];
#pragma unroll
    for (int u = 0; u < kQPL; ++u) {
        sidx[u] = (blockIdx.x * kQPL + u) * kThreads + threadIdx.x;
        const int sq = sidx[u] < S ? sidx[u] : S - 1;        // idle lanes shadow the last query (no divergence)
        const float *qp = Q + ((size_t)b * S + sq) * D;
#pragma unroll
        for (int k = 0; k < D; k += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(qp + k);
            q[u][k / 2] = f32x2{v.x, v.y}; q[u][k / 2 + 1] = f32x2{v.z, v.w};
        }
    }
    const int n0 = blockIdx.y * per_split;
    const int n1 = n0 + per_split < N ? n0 + per_split : N;
    const float *db = DB + (size_t)b * N * D;
    float best[kQPL];
    int best_n[kQPL];
#pragma unroll
    for (int u = 0; u < kQPL; ++u) { best[u] = -INFINITY; best_n[u] = n0; }
#pragma unroll 2
    for (int n = n0; n < n1; ++n) {                          // n is wave-uniform: the row loads are scalar loads
        const f32x2 *r = reinterpret_cast<const f32x2 *>(db + (size_t)n * D);
#pragma unroll
        for (int u = 0; u < kQPL; ++u) {
            f32x2 acc2 = {0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < D / 2; ++k) acc2 = __builtin_elementwise_fma(q[u][k], r[k], acc2);
            const float acc = acc2.x + acc2.y;
            if (acc > best[u]) { best[u] = acc; best_n[u] = n; }   // strict: the first (lowest) n wins inside a split
        }
    }
    if (n0 < n1) {
#pragma unroll
        for (int u = 0; u < kQPL; ++u) {
            if (sidx[u] >= S) continue;
            const unsigned long long key = ((unsigned long long)ordered_bits(best[u]) << 32) |
                                           (unsigned long long)(0xffffffffu - (unsigned)best_n[u]);
            atomicMax(keys + (size_t)b * S + sidx[u], key);
        }
    }
}

__global__ void __launch_bounds__(kThreads)
k_nn_unpack(const unsigned long long *__restrict__ keys, int32_t *__restrict__ idx, float *__restrict__ score,
            long long total) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    const unsigned long long k = keys[i];
    idx[i] = (int32_t)(0xffffffffu - (unsigned)(k & 0xffffffffull));
    if (score) score[i] = from_ordered((unsigned)(k >> 32));
}

// ------------------------------------------------------------------------------------------------------------
// The same search on the matrix cores.  scores = Q . DB^T is a [S x D] x [D x N] GEMM with a tiny K (D <= 32 = ONE
// v_mfma_f32_16x16x32_f16 k-step) and an arg-max epilogue; the fp32 kernel above runs at the plain-FMA rate
// (66-72 TFLOP/s).  Operands are packed once per call to [rows][32] fp16 (K zero-padded):
//   fp16 descriptors (BASELINE configs[4] "fp16 features")  -> 1 MFMA per 16 x 16 scores, exact products, fp32 sums
//   fp32 descriptors -> hi = fp16(x), lo = fp16(x - hi); score = hi.hi + hi.lo + lo.hi (3 MFMAs), dropped lo.lo term
//                       <= 2^-24 for unit vectors: the result is as close to the float64 oracle as the fp32 FMA chain
// Workgroup = 4 waves x 128 queries (8 query tiles in registers per wave); 64 database rows per step are staged once
// in LDS (LDS-DMA, double-buffered, swizzled so the ds_read_b128 fragment reads are conflict-free) and shared by the
// four waves.  Arg-max: two v_max3 per tile find whether ANY lane improved; only then (O(log N) times per query) the
// exact sequential update runs - rows ascend, strict '>': the lowest index wins ties as in the fp32 kernel.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <bool F16IN, int D>
__global__ void __launch_bounds__(kThreads)
k_nn_pack(const void *__restrict__ X, unsigned short *__restrict__ hi, unsigned short *__restrict__ lo, long long rows) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;      // one thread per (row, 8-element chunk)
    if (i >= rows * 4) return;
    const long long r = i >> 2;
    const int c = (int)(i & 3) * 8;
    union { unsigned short h[8]; uint4 q; } H, L;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float x = 0.f;
        if (c + k < D) x = F16IN ? (float)reinterpret_cast<const _Float16 *>(X)[r * D + c + k] : reinterpret_cast<const float *>(X)[r * D + c + k];
        const _Float16 h = (_Float16)x;
        H.h[k] = __builtin_bit_cast(unsigned short, h);
        const _Float16 l = (_Float16)(x - (float)h);
        L.h[k] = __builtin_bit_cast(unsigned short, l);
    }
    reinterpret_cast<uint4 *>(hi)[i] = H.q;
    if (lo) reinterpret_cast<uint4 *>(lo)[i] = L.q;
}

constexpr int kQT = 8;                      // query tiles (16 queries each) per wave
constexpr int kQPW = kQT * 16;              // queries per wave
constexpr int kQPB = kQPW * (kThreads / 64);   // queries per workgroup: 512
constexpr int kRows = 64;                   // database rows per LDS step

template <int PASSES>
__global__ void __launch_bounds__(kThreads)
k_nn_mfma(const unsigned short *__restrict__ Qhi, const unsigned short *__restrict__ Qlo,
          const unsigned short *__restrict__ Dhi, const unsigned short *__restrict__ Dlo,
          unsigned long long *__restrict__ keys, int S, int N, int per_split) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][kRows * 64];   // [stage][hi|lo][64 rows x 64 B]
    const int b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, g = lane >> 4;
    const size_t qb = (size_t)b * S, db = (size_t)b * N;
    // query fragments (MFMA B operand): lane -> query col, k = 8 g .. 8 g + 7
    f16x8 qh[kQT], ql[PASSES == 3 ? kQT : 1];
    const int q0 = blockIdx.x * kQPB + wave * kQPW;
#pragma unroll
    for (int t = 0; t < kQT; ++t) {
        int q = q0 + t * 16 + col;
        q = q < S ? q : S - 1;
        qh[t] = *reinterpret_cast<const f16x8 *>(Qhi + (qb + q) * 32 + g * 8);
        if (PASSES == 3) ql[t] = *reinterpret_cast<const f16x8 *>(Qlo + (qb + q) * 32 + g * 8);
    }
    const int n0 = blockIdx.y * per_split, n1 = min(n0 + per_split, N);
    float best[kQT];
    int bestn[kQT];
#pragma unroll
    for (int t = 0; t < kQT; ++t) { best[t] = -INFINITY; bestn[t] = n0; }
    // staging: thread -> (row = tid / 4, chunk' = tid % 4); source chunk = chunk' ^ ((-(row >> 2)) & 3)
    const int srow = tid >> 2, sc = (tid & 3) ^ ((0 - (srow >> 2)) & 3);
    auto stage = [&](int blk, int buf) {
        int n = n0 + blk * kRows + srow;
        n = n < N ? n : N - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned *)(Dhi + (db + n) * 32 + sc * 8),
                                         (__attribute__((address_space(3))) unsigned *)(&lds[buf][0][wave * 1024]), 16, 0, 0);
        if (PASSES == 3)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned *)(Dlo + (db + n) * 32 + sc * 8),
                                             (__attribute__((address_space(3))) unsigned *)(&lds[buf][1][wave * 1024]), 16, 0, 0);
    };
    const int nblk = (n1 - n0 + kRows - 1) / kRows;
    if (nblk > 0) stage(0, 0);
    for (int blk = 0; blk < nblk; ++blk) {
        const int buf = blk & 1;
        if (blk + 1 < nblk) {
            stage(blk + 1, buf ^ 1);
            if (PASSES == 3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const bool ragged = n0 + (blk + 1) * kRows > n1;                      // only the last step of a split
#pragma unroll
        for (int rt = 0; rt < kRows / 16; ++rt) {
            const int row = rt * 16 + col;                                    // A operand: lane -> row (lane & 15), k chunk g
            const int off = row * 64 + ((g ^ ((0 - (row >> 2)) & 3)) << 4);
            const f16x8 ah = *reinterpret_cast<const f16x8 *>(&lds[buf][0][off]);
            f16x8 al = ah;
            if (PASSES == 3) al = *reinterpret_cast<const f16x8 *>(&lds[buf][1][off]);
            const int nrow = n0 + blk * kRows + rt * 16 + g * 4;              // first of this lane's 4 output rows
#pragma unroll
            for (int t = 0; t < kQT; ++t) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, qh[t], acc, 0, 0, 0);
                if (PASSES == 3) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ql[t], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, qh[t], acc, 0, 0, 0);
                }
                if (ragged) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (nrow + j >= n1) acc[j] = -INFINITY;
                }
                const float mx = fmaxf(__builtin_fmaxf(__builtin_fmaxf(acc[0], acc[1]), acc[2]), acc[3]);
                if (__any(mx > best[t])) {                                    // rare after the first few steps
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (acc[j] > best[t]) { best[t] = acc[j]; bestn[t] = nrow + j; }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                         // reads of `buf` done before it is restaged
        __builtin_amdgcn_sched_barrier(0);
    }
    if (n0 >= n1) return;
    // the 4 lane groups hold disjoint row subsets of the same query: keep the larger score, ties -> lower index
#pragma unroll
    for (int t = 0; t < kQT; ++t) {
#pragma unroll
        for (int sh = 16; sh <= 32; sh <<= 1) {
            const float os = __shfl_xor(best[t], sh, 64);
            const int on = __shfl_xor(bestn[t], sh, 64);
            if (os > best[t] || (os == best[t] && on < bestn[t])) { best[t] = os; bestn[t] = on; }
        }
        const int q = q0 + t * 16 + col;
        if (g == 0 && q < S) {
            const unsigned long long key = ((unsigned long long)ordered_bits(best[t]) << 32) |
                                           (unsigned long long)(0xffffffffu - (unsigned)bestn[t]);
            atomicMax(keys + qb + q, key);
        }
    }
}

}  // namespace

extern "C" {

int m3_nn_search(const float *Q, const float *DB, int32_t *idx_out, float *score_out, uint64_t *keys_ws,
                 int B, int S, int N, int D, void *stream) {
    M3_REQUIRE(Q && DB && idx_out && keys_ws && B > 0 && S > 0 && N > 0 && B <= 65535);
    M3_REQUIRE(D == 16 || D == 24 || D == 32);
    M3_REQUIRE(((reinterpret_cast<size_t>(Q) | reinterpret_cast<size_t>(DB)) & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    M3_CHECK_HIP(hipMemsetAsync(keys_ws, 0, (size_t)B * S * 8, st), "m3_nn_search/memset");
    const int qblocks = m3_cdiv(S, kThreads * kQPL);
    int splits = m3_cdiv(2048, qblocks * B);                 // ~8 workgroups per CU over the whole call
    if (splits < 1) splits = 1;
    if (splits > 1024) splits = 1024;
    if (splits > N) splits = N;
    const int per_split = m3_cdiv(N, splits);
    splits = m3_cdiv(N, per_split);
    dim3 grid(qblocks, splits, B);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(keys_ws);
    if (D == 24) hipLaunchKernelGGL(k_nn_search<24>, grid, dim3(kThreads), 0, st, Q, DB, keys, S, N, per_split);
    else if (D == 16) hipLaunchKernelGGL(k_nn_search<16>, grid, dim3(kThreads), 0, st, Q, DB, keys, S, N, per_split);
    else hipLaunchKernelGGL(k_nn_search<32>, grid, dim3(kThreads), 0, st, Q, DB, keys, S, N, per_split);
    M3_CHECK_LAUNCH("m3_nn_search");
    const long long total = (long long)B * S;
    hipLaunchKernelGGL(k_nn_unpack, dim3((unsigned)m3_cdiv(total, (long long)kThreads)), dim3(kThreads), 0, st,
                       (const unsigned long long *)keys, idx_out, score_out, total);
    M3_CHECK_LAUNCH("m3_nn_search/unpack");
    return M3_OK;
}

// MFMA search.  Q [B,S,D], DB [B,N,D] in fp32 (in_f16 = 0) or IEEE fp16 (in_f16 = 1), D <= 32 and D % 4 == 0.
// pack_ws: m3_nn_pack_bytes(B, S, N, in_f16) bytes of scratch for the K-padded fp16 operands.
int64_t m3_nn_pack_bytes(int B, int S, int N, int in_f16) {
    if (B <= 0 || S <= 0 || N <= 0) return 0;
    return (int64_t)B * ((int64_t)S + N) * 64 * (in_f16 ? 1 : 2);
}

int m3_nn_search_mfma(const void *Q, const void *DB, int32_t *idx_out, float *score_out, uint64_t *keys_ws,
                      void *pack_ws, int B, int S, int N, int D, int in_f16, void *stream) {
    M3_REQUIRE(Q && DB && idx_out && keys_ws && pack_ws && B > 0 && S > 0 && N > 0 && B <= 65535);
    M3_REQUIRE((D == 16 || D == 24 || D == 32) && (reinterpret_cast<size_t>(pack_ws) & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    unsigned short *qhi = (unsigned short *)pack_ws, *dhi = qhi + (size_t)B * S * 32;
    unsigned short *qlo = in_f16 ? nullptr : dhi + (size_t)B * N * 32, *dlo = in_f16 ? nullptr : qlo + (size_t)B * S * 32;
    const long long qr = (long long)B * S, dr = (long long)B * N;
    const dim3 blk(kThreads), gq((unsigned)m3_cdiv(qr * 4, (long long)kThreads)), gd((unsigned)m3_cdiv(dr * 4, (long long)kThreads));
#define M3_PACK(DD)                                                                                     \
    do {                                                                                                \
        if (in_f16) { hipLaunchKernelGGL((k_nn_pack<true, DD>), gq, blk, 0, st, Q, qhi, qlo, qr);        \
                      hipLaunchKernelGGL((k_nn_pack<true, DD>), gd, blk, 0, st, DB, dhi, dlo, dr); }     \
        else { hipLaunchKernelGGL((k_nn_pack<false, DD>), gq, blk, 0, st, Q, qhi, qlo, qr);              \
               hipLaunchKernelGGL((k_nn_pack<false, DD>), gd, blk, 0, st, DB, dhi, dlo, dr); }           \
    } while (0)
    if (D == 24) M3_PACK(24); else if (D == 16) M3_PACK(16); else M3_PACK(32);
#undef M3_PACK
    M3_CHECK_LAUNCH("m3_nn_search_mfma/pack");
    M3_CHECK_HIP(hipMemsetAsync(keys_ws, 0, (size_t)B * S * 8, st), "m3_nn_search_mfma/memset");
    const int qblocks = m3_cdiv(S, kQPB);
    int splits = m3_cdiv(1024, qblocks * B);                  // ~4 workgroups per CU over the whole call
    if (splits < 1) splits = 1;
    if (splits > m3_cdiv(N, kRows)) splits = m3_cdiv(N, kRows);
    int per_split = m3_cdiv(N, splits);
    per_split = m3_cdiv(per_split, kRows) * kRows;            // whole LDS steps except in the last split
    splits = m3_cdiv(N, per_split);
    const dim3 grid(qblocks, splits, B);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(keys_ws);
    if (in_f16) hipLaunchKernelGGL(k_nn_mfma<1>, grid, blk, 0, st, qhi, qlo, dhi, dlo, keys, S, N, per_split);
    else hipLaunchKernelGGL(k_nn_mfma<3>, grid, blk, 0, st, qhi, qlo, dhi, dlo, keys, S, N, per_split);
    M3_CHECK_LAUNCH("m3_nn_search_mfma");
    const long long total = (long long)B * S;
    hipLaunchKernelGGL(k_nn_unpack, dim3((unsigned)m3_cdiv(total, (long long)kThreads)), dim3(kThreads), 0, st,
                       (const unsigned long long *)keys, idx_out, score_out, total);
    M3_CHECK_LAUNCH("m3_nn_search_mfma/unpack");
    return M3_OK;
}

}  // extern "C"
