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

}  // extern "C"
