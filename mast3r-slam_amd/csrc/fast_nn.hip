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
// Workgroup = 4 waves x 128 queries (8 query tiles in registers per wave); 128 database rows per step are staged once
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
constexpr int kRows = 128;                  // database rows per LDS step (two workgroup barriers per step; 64 rows: 1564 us per round, 128: 1509)

// Round 3.  The round-2 loop issued ONE MFMA at a time: its result went to an AGPR quad that the next instructions
// read back (v_accvgpr_read x 4), so the following MFMA waited out the matrix core's full latency, and the ragged-tail
// compares ran on every tile: 358 TFLOP/s (K padded to 32) = 0.14 of the peak.  Now (a) the file is built with
// -amdgpu-mfma-vgpr-form (results land in VGPRs), (b) the 2 x 4 MFMAs of two row tiles x four query tiles are issued
// back to back into eight accumulators and only then reduced (max over a lane's 8 scores of a query: 3 x v_max3 + v_max,
// one compare + wave vote per PAIR of tiles), (c) the ragged tail is a separate instantiation of the step (RAGGED) taken
// for the last step of a split only, (d) queries are gathered BY INDEX from a packed map (qidx), so a descriptor map is
// packed once per matcher call and serves as database of one search direction and query source of the other.
template <int PASSES, bool RAGGED>
__device__ __forceinline__ void nn_step(const unsigned char *__restrict__ hi_s, const unsigned char *__restrict__ lo_s,
                                        const f16x8 (&qh)[kQT], const f16x8 (&ql)[PASSES == 3 ? kQT : 1],
                                        float (&best)[kQT], int (&bestn)[kQT], int nbase, int n1, int col, int g) {
#pragma unroll
    for (int rp = 0; rp < kRows / 32; ++rp) {                                 // pairs of 16-row tiles
        f16x8 ah[2], al[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = (2 * rp + h) * 16 + col;                          // A operand: lane -> row (lane & 15), k chunk g
            const int off = row * 64 + ((g ^ ((0 - (row >> 2)) & 3)) << 4);
            ah[h] = *reinterpret_cast<const f16x8 *>(hi_s + off);
            al[h] = ah[h];
            if (PASSES == 3) al[h] = *reinterpret_cast<const f16x8 *>(lo_s + off);
        }
        const int nrow = nbase + rp * 32 + g * 4;                             // first of this lane's 2 x 4 output rows
#pragma unroll
        for (int t0 = 0; t0 < kQT; t0 += 4) {
            f32x4 acc[4][2];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[h], qh[t0 + tt], c, 0, 0, 0);
                    if (PASSES == 3) {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[h], ql[t0 + tt], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[h], qh[t0 + tt], c, 0, 0, 0);
                    }
                    acc[tt][h] = c;
                }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int t = t0 + tt;
                f32x4 &a0 = acc[tt][0], &a1 = acc[tt][1];
                if (RAGGED) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (nrow + j >= n1) a0[j] = -INFINITY;
                        if (nrow + 16 + j >= n1) a1[j] = -INFINITY;
                    }
                }
                float mx = __builtin_fmaxf(__builtin_fmaxf(a0[0], a0[1]), a0[2]);
                mx = __builtin_fmaxf(__builtin_fmaxf(mx, a0[3]), a1[0]);
                mx = __builtin_fmaxf(__builtin_fmaxf(mx, a1[1]), a1[2]);
                mx = __builtin_fmaxf(mx, a1[3]);
                // the running maximum is updated on the straight-line path (one v_max); only the INDEX lives in the rare
                // branch - with both updated there, every loop-carried best[] register became a phi and was copied
                // around the branch on every tile (PMC: 8.4 VALU instructions per MFMA, most of them v_mov)
                const bool imp = mx > best[t];
                best[t] = __builtin_fmaxf(best[t], mx);
                if (__any(imp)) {                                             // rare after the first few steps
                    int bn = bestn[t];
#pragma unroll
                    for (int j = 3; j >= 0; --j) bn = (imp && a1[j] == best[t]) ? nrow + 16 + j : bn;   // descending: the lowest row
#pragma unroll
                    for (int j = 3; j >= 0; --j) bn = (imp && a0[j] == best[t]) ? nrow + j : bn;        // that holds the maximum wins
                    bestn[t] = bn;
                }
            }
        }
    }
}

// Qhi / Qlo: packed rows [P][NQ][32]; qidx int32 [P][S] selects the query rows (null: query s = row s, NQ = S).
template <int PASSES>
__global__ void __launch_bounds__(kThreads, PASSES == 1 ? 4 : 2)       // fp16 descriptors: 4 waves per SIMD (<= 128 registers)
k_nn_mfma(const unsigned short *__restrict__ Qhi, const unsigned short *__restrict__ Qlo, const int32_t *__restrict__ qidx,
          const unsigned short *__restrict__ Dhi, const unsigned short *__restrict__ Dlo,
          unsigned long long *__restrict__ keys, int S_all, int NQ, int N, int per_split,
          const int32_t *__restrict__ qlist, const int32_t *__restrict__ qcount) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][kRows * 64];   // [stage][hi|lo][64 rows x 64 B]
    const int b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, g = lane >> 4;
    const size_t qb = (size_t)b * NQ, db = (size_t)b * N, kb = (size_t)b * S_all;
    // active-set form (rounds >= 2 of the reciprocal matcher): only the first qcount[b] entries of qlist[b] are queries -
    // the seed slots that have not converged yet, in ascending order; a workgroup past the end has nothing to do
    // (workgroup-uniform exit in front of every barrier).  keys / qidx stay indexed by the seed SLOT.
    const int S = qcount ? qcount[b] : S_all;
    int bx = blockIdx.x, by = blockIdx.y;
    if (qcount) {
        // The grid was sized for S_all queries (gridDim.x query blocks x gridDim.y database splits = one round of the
        // chip).  With only nb = ceil(S / 512) query blocks left, the same workgroups are re-dealt as nb query blocks x
        // (slots / nb) database splits: every CU still works, each on a shorter database range - the round's time shrinks
        // with the active set instead of staying one full-length workgroup long.  (The per-split winners merge by atomicMax
        // whatever the number of splits.)
        if (S <= 0) return;
        const int nb = (S + kQPB - 1) / kQPB, slots = (int)(gridDim.x * gridDim.y);
        const int w = by * (int)gridDim.x + bx, nsplit = slots / nb;
        bx = w % nb; by = w / nb;
        if (by >= nsplit) return;
        per_split = (((N + nsplit - 1) / nsplit) + kRows - 1) / kRows * kRows;
    }
    // query fragments (MFMA B operand): lane -> query col, k = 8 g .. 8 g + 7
    f16x8 qh[kQT], ql[PASSES == 3 ? kQT : 1];
    const int q0 = bx * kQPB + wave * kQPW;
#pragma unroll
    for (int t = 0; t < kQT; ++t) {
        int q = q0 + t * 16 + col;
        q = q < S ? q : S - 1;
        const int slot = qlist ? qlist[kb + q] : q;
        int qr = qidx ? qidx[kb + slot] : slot;
        qr = qr < 0 ? 0 : (qr >= NQ ? NQ - 1 : qr);
        qh[t] = *reinterpret_cast<const f16x8 *>(Qhi + (qb + qr) * 32 + g * 8);
        if (PASSES == 3) ql[t] = *reinterpret_cast<const f16x8 *>(Qlo + (qb + qr) * 32 + g * 8);
    }
    const int n0 = by * per_split, n1 = min(n0 + per_split, N);
    float best[kQT];
    int bestn[kQT];
#pragma unroll
    for (int t = 0; t < kQT; ++t) { best[t] = -INFINITY; bestn[t] = n0; }
    // staging: thread -> (row = tid / 4, chunk' = tid % 4); source chunk = chunk' ^ ((-(row >> 2)) & 3)
    const int srow = tid >> 2, sc = (tid & 3) ^ ((0 - (srow >> 2)) & 3);
    auto stage = [&](int blk, int buf) {
#pragma unroll
        for (int i = 0; i < kRows / 64; ++i) {
            int n = n0 + blk * kRows + i * 64 + srow;
            n = n < N ? n : N - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned *)(Dhi + (db + n) * 32 + sc * 8),
                                             (__attribute__((address_space(3))) unsigned *)(&lds[buf][0][i * 4096 + wave * 1024]), 16, 0, 0);
            if (PASSES == 3)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) unsigned *)(Dlo + (db + n) * 32 + sc * 8),
                                                 (__attribute__((address_space(3))) unsigned *)(&lds[buf][1][i * 4096 + wave * 1024]), 16, 0, 0);
        }
    };
    const int nblk = (n1 - n0 + kRows - 1) / kRows;
    if (nblk > 0) stage(0, 0);
    for (int blk = 0; blk < nblk; ++blk) {
        const int buf = blk & 1;
        if (blk + 1 < nblk) {
            stage(blk + 1, buf ^ 1);
            static_assert(kRows == 128, "the counted waits below assume two 64-row issues per plane and stage");
            if (PASSES == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int nbase = n0 + blk * kRows;
        if (nbase + kRows > n1) nn_step<PASSES, true>(lds[buf][0], lds[buf][1], qh, ql, best, bestn, nbase, n1, col, g);   // last step of a split only
        else nn_step<PASSES, false>(lds[buf][0], lds[buf][1], qh, ql, best, bestn, nbase, n1, col, g);
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
            atomicMax(keys + kb + (qlist ? qlist[kb + q] : q), key);
        }
    }
}

// ---- one round of fast reciprocal NN on the device (matching.fast_reciprocal_nn_device), batched over P pairs -----------
// mid: keys of the forward search -> xy2 (the view-2 pixel each seed landed on), keys cleared for the backward search
__global__ void __launch_bounds__(kThreads)
k_frnn_mid(unsigned long long *__restrict__ keys, int32_t *__restrict__ xy2, long long total) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    xy2[i] = (int32_t)(0xffffffffu - (unsigned)(keys[i] & 0xffffffffull));   // (inactive slots: key 0 -> a value nobody reads)
    keys[i] = 0ull;
}
// compact: the still-active seed slots of every pair, ascending, + their number - the query list of the next round's
// searches.  One workgroup per pair walks its S flags in order (ballot + prefix counts): deterministic.
__global__ void __launch_bounds__(kThreads)
k_frnn_compact(const uint8_t *__restrict__ active, int32_t *__restrict__ list, int32_t *__restrict__ count, int S) {
    __shared__ int base, wsum[kThreads / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int s0 = 0; s0 < S; s0 += kThreads) {
        const int sl = s0 + tid;
        const bool a = sl < S && active[(size_t)b * S + sl] != 0;
        const unsigned long long m = __ballot(a);
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (a) list[(size_t)b * S + off + __popcll(m & ((1ull << lane) - 1ull))] = sl;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < kThreads / 64; ++w) t += wsum[w]; base += t; }
        __syncthreads();
    }
    if (tid == 0) count[b] = base;
}
// collect: every reciprocal pair of every round (got1 / got2 [rounds][P][S], -1 = none) into dense maps - map1[pair][p1]
// = p2 (int32, -1 = none) and, for the tracker, idx2[pair][p2] = p1 / valid2[pair][p2] = 1.  A reciprocal pair is mutual,
// so p1 <-> p2 is one-to-one per image pair and seeds that found the same pair write the same values.
__global__ void __launch_bounds__(kThreads)
k_frnn_scatter(const int32_t *__restrict__ got1, const int32_t *__restrict__ got2, int32_t *__restrict__ map1,
               long long *__restrict__ idx2, uint8_t *__restrict__ valid2, int P, int S, int N1, int N2, long long total) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    const int p1 = got1[i], p2 = got2[i];
    if (p1 < 0 || p1 >= N1 || p2 < 0 || p2 >= N2) return;
    const int b = (int)((i / S) % P);
    map1[(size_t)b * N1 + p1] = p2;
    if (idx2) { idx2[(size_t)b * N2 + p2] = p1; valid2[(size_t)b * N2 + p2] = 1; }
}
// the distinct pairs as a list sorted by (pair, p1): ordered compaction of map1 in two launches (per-chunk counts, then
// offsets = sum of the counts in front + an in-chunk prefix) - what torch.unique (a sort and a host synchronisation) did
constexpr int kChunk = kThreads * 16;
__global__ void __launch_bounds__(kThreads)
k_frnn_count(const int32_t *__restrict__ map1, int32_t *__restrict__ chunk_cnt, int N1, int nchunk) {
    __shared__ int wsum[kThreads / 64];
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const int32_t *m = map1 + (size_t)b * N1;
    int n = 0;
    for (int k = 0; k < 16; ++k) { const int i = c * kChunk + tid * 16 + k; n += (i < N1 && m[i] >= 0) ? 1 : 0; }
    for (int sh = 32; sh >= 1; sh >>= 1) n += __shfl_xor(n, sh, 64);
    if ((tid & 63) == 0) wsum[tid >> 6] = n;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < kThreads / 64; ++w) t += wsum[w]; chunk_cnt[b * nchunk + c] = t; }
}
__global__ void __launch_bounds__(kThreads)
k_frnn_emit(const int32_t *__restrict__ map1, const int32_t *__restrict__ chunk_cnt, int32_t *__restrict__ pairs,
            int32_t *__restrict__ count, int N1, int nchunk, int cap) {
    __shared__ int pre[kThreads];
    __shared__ int off0;
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const int32_t *m = map1 + (size_t)b * N1;
    if (tid == 0) {
        int o = 0, tot = 0;
        for (int k = 0; k < nchunk; ++k) { const int v = chunk_cnt[b * nchunk + k]; if (k < c) o += v; tot += v; }
        off0 = o;
        if (c == 0) count[b] = tot < cap ? tot : cap;
    }
    int n = 0;
    for (int k = 0; k < 16; ++k) { const int i = c * kChunk + tid * 16 + k; n += (i < N1 && m[i] >= 0) ? 1 : 0; }
    pre[tid] = n;
    __syncthreads();
    for (int d = 1; d < kThreads; d <<= 1) {                       // inclusive scan of the 256 per-thread counts
        const int v = tid >= d ? pre[tid - d] : 0;
        __syncthreads();
        pre[tid] += v;
        __syncthreads();
    }
    int o = off0 + pre[tid] - n;
    for (int k = 0; k < 16; ++k) {
        const int i = c * kChunk + tid * 16 + k;
        if (i < N1 && m[i] >= 0) {
            if (o < cap) { pairs[((size_t)b * cap + o) * 2] = i; pairs[((size_t)b * cap + o) * 2 + 1] = m[i]; }
            ++o;
        }
    }
}
// end: keys of the backward search -> back; a seed that returned to where it started is a reciprocal pair (recorded in
// row `round` of got1 / got2, -1 elsewhere) and leaves the active set, the others continue from where they landed
__global__ void __launch_bounds__(kThreads)
k_frnn_end(unsigned long long *__restrict__ keys, const int32_t *__restrict__ xy2, int32_t *__restrict__ cur,
           uint8_t *__restrict__ active, int32_t *__restrict__ got1, int32_t *__restrict__ got2, long long total) {
    const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    const int32_t back = (int32_t)(0xffffffffu - (unsigned)(keys[i] & 0xffffffffull));
    keys[i] = 0ull;
    const int32_t c = cur[i];
    const bool act = active[i] != 0, conv = act && back == c;
    got1[i] = conv ? c : -1;
    got2[i] = conv ? xy2[i] : -1;
    active[i] = (uint8_t)(act && !conv);
    if (act && !conv) cur[i] = back;
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
    if (in_f16) hipLaunchKernelGGL(k_nn_mfma<1>, grid, blk, 0, st, qhi, qlo, (const int32_t *)nullptr, dhi, dlo, keys, S, S, N, per_split,
                                   (const int32_t *)nullptr, (const int32_t *)nullptr);
    else hipLaunchKernelGGL(k_nn_mfma<3>, grid, blk, 0, st, qhi, qlo, (const int32_t *)nullptr, dhi, dlo, keys, S, S, N, per_split,
                            (const int32_t *)nullptr, (const int32_t *)nullptr);
    M3_CHECK_LAUNCH("m3_nn_search_mfma");
    const long long total = (long long)B * S;
    hipLaunchKernelGGL(k_nn_unpack, dim3((unsigned)m3_cdiv(total, (long long)kThreads)), dim3(kThreads), 0, st,
                       (const unsigned long long *)keys, idx_out, score_out, total);
    M3_CHECK_LAUNCH("m3_nn_search_mfma/unpack");
    return M3_OK;
}


// ---- batched fast reciprocal NN: pack each descriptor map ONCE, then rounds of (forward search, backward search) ------
// A packed map is [P][N][32] fp16 (K zero-padded), followed by the lo plane for fp32 descriptors.
int64_t m3_frnn_pack_bytes(int P, int N, int in_f16) {
    if (P <= 0 || N <= 0) return 0;
    return (int64_t)P * N * 64 * (in_f16 ? 1 : 2);
}
int m3_frnn_pack(const void *Dmap, void *packed, int P, int N, int D, int in_f16, void *stream) {
    M3_REQUIRE(Dmap && packed && P > 0 && N > 0 && (D == 16 || D == 24 || D == 32));
    M3_REQUIRE((reinterpret_cast<size_t>(packed) & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    unsigned short *hi = (unsigned short *)packed, *lo = in_f16 ? nullptr : hi + (size_t)P * N * 32;
    const long long rows = (long long)P * N;
    const dim3 blk(kThreads), grid((unsigned)m3_cdiv(rows * 4, (long long)kThreads));
#define M3_PACK1(DD)                                                                                         \
    do { if (in_f16) hipLaunchKernelGGL((k_nn_pack<true, DD>), grid, blk, 0, st, Dmap, hi, lo, rows);          \
         else hipLaunchKernelGGL((k_nn_pack<false, DD>), grid, blk, 0, st, Dmap, hi, lo, rows); } while (0)
    if (D == 24) M3_PACK1(24); else if (D == 16) M3_PACK1(16); else M3_PACK1(32);
#undef M3_PACK1
    M3_CHECK_LAUNCH("m3_frnn_pack");
    return M3_OK;
}

static int frnn_search(const void *qpacked, int NQ, const int32_t *qidx, const void *dpacked, int N, unsigned long long *keys,
                       int P, int S, int in_f16, hipStream_t st, const int32_t *qlist = nullptr, const int32_t *qcount = nullptr) {
    const unsigned short *qhi = (const unsigned short *)qpacked, *qlo = in_f16 ? nullptr : qhi + (size_t)P * NQ * 32;
    const unsigned short *dhi = (const unsigned short *)dpacked, *dlo = in_f16 ? nullptr : dhi + (size_t)P * N * 32;
    const int qblocks = m3_cdiv(S, kQPB);
    int splits = m3_cdiv(1024, qblocks * P);                  // ~4 workgroups per CU over the whole call
    if (splits < 1) splits = 1;
    if (splits > m3_cdiv(N, kRows)) splits = m3_cdiv(N, kRows);
    int per_split = m3_cdiv(N, splits);
    per_split = m3_cdiv(per_split, kRows) * kRows;
    splits = m3_cdiv(N, per_split);
    const dim3 grid(qblocks, splits, P), blk(kThreads);
    if (in_f16) hipLaunchKernelGGL(k_nn_mfma<1>, grid, blk, 0, st, qhi, qlo, qidx, dhi, dlo, keys, S, NQ, N, per_split, qlist, qcount);
    else hipLaunchKernelGGL(k_nn_mfma<3>, grid, blk, 0, st, qhi, qlo, qidx, dhi, dlo, keys, S, NQ, N, per_split, qlist, qcount);
    return M3_OK;
}

// One round for P pairs at once (no reference counterpart: SURVEY 8a row K8).  packed1 / packed2: the two views'
// packed maps (m3_frnn_pack; N1 / N2 rows per pair).  cur int32 [P,S]: the view-1 pixel every seed currently sits on
// (in / out), active uint8 [P,S] (in / out), got1 / got2 int32 [P,S]: this round's reciprocal pairs (-1 = none),
// xy2_ws int32 [P,S] and keys_ws uint64 [P,S] scratch; keys_ws must be ZERO on entry (it is left zero).
int m3_frnn_round(const void *packed1, const void *packed2, int32_t *cur, uint8_t *active, int32_t *got1, int32_t *got2,
                  int32_t *xy2_ws, uint64_t *keys_ws, int P, int S, int N1, int N2, int in_f16, void *stream) {
    M3_REQUIRE(packed1 && packed2 && cur && active && got1 && got2 && xy2_ws && keys_ws);
    M3_REQUIRE(P > 0 && P <= 65535 && S > 0 && N1 > 0 && N2 > 0);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(keys_ws);
    const long long total = (long long)P * S;
    const dim3 eb(kThreads), eg((unsigned)m3_cdiv(total, (long long)kThreads));
    frnn_search(packed1, N1, cur, packed2, N2, keys, P, S, in_f16, st);               // view 1 -> view 2
    hipLaunchKernelGGL(k_frnn_mid, eg, eb, 0, st, keys, xy2_ws, total);
    frnn_search(packed2, N2, xy2_ws, packed1, N1, keys, P, S, in_f16, st);            // and back
    hipLaunchKernelGGL(k_frnn_end, eg, eb, 0, st, keys, (const int32_t *)xy2_ws, cur, active, got1, got2, total);
    M3_CHECK_LAUNCH("m3_frnn_round");
    return M3_OK;
}

// m3_frnn_round restricted to the seeds that are still active: act_ws int32 [P * (S + 1)] scratch receives the ascending
// list of active seed slots of every pair ([P][S]) and their count ([P]); the two searches then run on those queries only
// (workgroups past a pair's count exit at once).  After the first round most seeds of a well-textured pair have found
// their mutual nearest neighbour, so rounds >= 2 cost a fraction of a full round.  Same results as m3_frnn_round.
int m3_frnn_round_active(const void *packed1, const void *packed2, int32_t *cur, uint8_t *active, int32_t *got1,
                         int32_t *got2, int32_t *xy2_ws, uint64_t *keys_ws, int32_t *act_ws, int P, int S, int N1, int N2,
                         int in_f16, void *stream) {
    M3_REQUIRE(packed1 && packed2 && cur && active && got1 && got2 && xy2_ws && keys_ws && act_ws);
    M3_REQUIRE(P > 0 && P <= 65535 && S > 0 && N1 > 0 && N2 > 0);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(keys_ws);
    int32_t *list = act_ws, *count = act_ws + (size_t)P * S;
    const long long total = (long long)P * S;
    const dim3 eb(kThreads), eg((unsigned)m3_cdiv(total, (long long)kThreads));
    hipLaunchKernelGGL(k_frnn_compact, dim3(P), eb, 0, st, (const uint8_t *)active, list, count, S);
    frnn_search(packed1, N1, cur, packed2, N2, keys, P, S, in_f16, st, list, count);
    hipLaunchKernelGGL(k_frnn_mid, eg, eb, 0, st, keys, xy2_ws, total);
    frnn_search(packed2, N2, xy2_ws, packed1, N1, keys, P, S, in_f16, st, list, count);
    hipLaunchKernelGGL(k_frnn_end, eg, eb, 0, st, keys, (const int32_t *)xy2_ws, cur, active, got1, got2, total);
    M3_CHECK_LAUNCH("m3_frnn_round_active");
    return M3_OK;
}

// The reciprocal pairs of `rounds` rounds (got1 / got2 int32 [rounds,P,S]) as FIXED-SHAPE device outputs - no sort, no
// host synchronisation, so the whole matcher can be captured into a hipGraph:
//   map1  int32 [P,N1]   view-1 pixel -> its reciprocal partner in view 2, -1 = none            (always written)
//   idx2  int64 [P,N2], valid2 uint8 [P,N2]   the tracker's maps: view-2 pixel -> view-1 pixel  (optional: both or none)
//   pairs int32 [P,S,2], count int32 [P]   the distinct (p1, p2) of every image pair sorted by p1, rows >= count[pair] = -1
//                                          (a seed converges at most once: S bounds the number of pairs)
//   chunk_ws int32 [P * m3_frnn_chunks(N1)] scratch.
int m3_frnn_chunks(int N1) { return N1 > 0 ? (N1 + kChunk - 1) / kChunk : 0; }
int m3_frnn_collect(const int32_t *got1, const int32_t *got2, int rounds, int P, int S, int N1, int N2, int32_t *map1,
                    int64_t *idx2, uint8_t *valid2, int32_t *pairs, int32_t *count, int32_t *chunk_ws, void *stream) {
    M3_REQUIRE(got1 && got2 && map1 && pairs && count && chunk_ws && rounds > 0 && P > 0 && P <= 65535 && S > 0 && N1 > 0 && N2 > 0);
    M3_REQUIRE((idx2 == nullptr) == (valid2 == nullptr));
    hipStream_t st = (hipStream_t)stream;
    M3_CHECK_HIP(hipMemsetAsync(map1, 0xff, (size_t)P * N1 * 4, st), "m3_frnn_collect/memset");
    M3_CHECK_HIP(hipMemsetAsync(pairs, 0xff, (size_t)P * S * 8, st), "m3_frnn_collect/memset");
    if (idx2) {
        M3_CHECK_HIP(hipMemsetAsync(idx2, 0, (size_t)P * N2 * 8, st), "m3_frnn_collect/memset");
        M3_CHECK_HIP(hipMemsetAsync(valid2, 0, (size_t)P * N2, st), "m3_frnn_collect/memset");
    }
    const long long total = (long long)rounds * P * S;
    hipLaunchKernelGGL(k_frnn_scatter, dim3((unsigned)m3_cdiv(total, (long long)kThreads)), dim3(kThreads), 0, st, got1, got2, map1,
                       (long long *)idx2, valid2, P, S, N1, N2, total);
    const int nchunk = m3_frnn_chunks(N1);
    hipLaunchKernelGGL(k_frnn_count, dim3(nchunk, P), dim3(kThreads), 0, st, (const int32_t *)map1, chunk_ws, N1, nchunk);
    hipLaunchKernelGGL(k_frnn_emit, dim3(nchunk, P), dim3(kThreads), 0, st, (const int32_t *)map1, (const int32_t *)chunk_ws, pairs,
                       count, N1, nchunk, S);
    M3_CHECK_LAUNCH("m3_frnn_collect");
    return M3_OK;
}

}  // extern "C"
