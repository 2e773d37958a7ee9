// Nearest neighbour in descriptor space (maximum dot product) - the search primitive of "fast reciprocal NN" matching
// (MASt3R, Leroy et al. 2024, section 3.3; mast3r/fast_nn.py of the public implementation).  Three forms, same result:
// an fp32 FMA-chain kernel (k_nn_search), the brute-force search on the matrix cores (k_nn_mfma) and, since round 4, an
// exact search that bounds most of the database away first (k_frnn_blockstats / _seed_lb / _survivors / _eval, further
// down) with k_nn_mfma as its device-side fallback.  BASELINE.json's north_star names this matcher; the reference tree has no implementation
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

// ---- exact search with block bounds: shared definitions (kernels further down) ----
constexpr float kSlack = 1.52587890625e-05f;   // 2^-16: covers the fp32 rounding of the bound and of the scores it is compared with
// more than a quarter of all (query tile, block) pairs survived: the brute-force kernel is the faster one
__device__ __forceinline__ bool prune_overflow(int total, int S, int NB) {
    return (long long)total * 4 > (long long)((S + 15) / 16) * NB;
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
          const int32_t *__restrict__ qlist, const int32_t *__restrict__ qcount,
          const int32_t *__restrict__ gate_total, int gate_nb) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][kRows * 64];   // [stage][hi|lo][64 rows x 64 B]
    const int b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // fallback of the block-bound search (k_frnn_eval below): runs only when the bounds pruned too little
    if (gate_total && !prune_overflow(gate_total[b], qcount ? qcount[b] : S_all, gate_nb)) return;
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

// ------------------------------------------------------------------------------------------------------------
// Round 4: the SAME arg-max with most of the database never touched.  The brute-force search above scores every seed
// against every pixel of the other view (4096 x 262 144 x 2 directions per pair and round); a descriptor map is smooth,
// so whole patches of pixels can be excluded by a bound.  Per map, once per matcher call (k_frnn_blockstats): for every
// block = 8 x 8 pixel tile (a first version used 64 consecutive pixels of an image row: radius 0.63 on the synthetic
// scene's unit descriptors, a quarter of all blocks survived; a square tile is four times more compact) a reference point
// c (the centroid, rounded to fp16 so that it is an exact MFMA operand), the radius r = max ||x - c|| and ||c|| + r.
// Blocks are numbered by * ceil(W / 8) + bx; block-local row 8 ry + rx is pixel (8 by + ry, 8 bx + rx); the packed map
// itself stays in raster order (a packed row is 64 bytes: the tile's rows are gathered by address).
// For a query q and ANY row x of the block (Cauchy-Schwarz)
//        <q, x> = <q, c> + <q, x - c>  <=  <q, c> + ||q|| r.
// Per search:
//   1. k_nn_mfma on the CENTROIDS (N / 64 rows) picks a promising block per query; k_frnn_seed_lb scores its 64 rows ->
//      a lower bound m(q) of the final maximum (minus a slack for the rounding differences to the MFMA scores);
//   2. k_frnn_survivors: <q, c> for all (query, block) pairs on the matrix core (1/64 of the full search), one bit per
//      (16-query tile, block): set unless  <q, c> + ||q|| (r + slack)  <  m(q)  for all 16 queries;
//   3. k_frnn_eval scores the surviving blocks with the SAME MFMA sequence as nn_step (bit-identical scores) and keeps
//      (score, lowest PIXEL index) - tiles are not visited in raster order, so ties compare indices explicitly - then the
//      same 64-bit key merge; a block that holds the maximum, or a tie with it, always survives (its bound is >= its
//      best score >= m), so index AND score equal the brute-force result bit for bit;
//   4. if more than a quarter of the pairs survived (descriptors without spatial coherence, e.g. a random-weight
//      network), k_frnn_eval stands down and k_nn_mfma runs instead (both test the same device counter).
// Finite descriptors are assumed (this file is built with -fno-honor-nans, as the brute-force kernel always was).
__device__ __forceinline__ void load_row32(const unsigned short *__restrict__ hi, const unsigned short *__restrict__ lo, size_t row,
                                           float (&x)[32]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        union { uint4 q; _Float16 h[8]; } H, L;
        H.q = reinterpret_cast<const uint4 *>(hi + row * 32)[c];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[c * 8 + k] = (float)H.h[k];
        if (lo) {
            L.q = reinterpret_cast<const uint4 *>(lo + row * 32)[c];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[c * 8 + k] += (float)L.h[k];
        }
    }
}

// cen [P][NBp][32] fp16, rad / bn [P][NBp] fp32; NBp = blocks rounded up to a multiple of 64 (padding: rad = -inf).
// One wave per tile: lane = (tile row ry = lane >> 3, dimension quad dq = lane & 7) owns the 8 pixels of its row x 4
// dimensions (8-byte loads; the 8 lanes of a row cover the pixels' 64-byte packed rows), so the centroid needs 3 shuffle
// steps per dimension quad and the squared distances 3 per pixel (a first version - lane = pixel, 32 wave reductions - took
// 78 us per 8 maps, more than the searches it serves).
__global__ void __launch_bounds__(kThreads)
k_frnn_blockstats(const unsigned short *__restrict__ hi, const unsigned short *__restrict__ lo, unsigned short *__restrict__ cen,
                  float *__restrict__ rad, float *__restrict__ bn, int H, int W, int NB, int NBp) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, blk = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    if (blk >= NBp) return;
    const size_t sb = (size_t)b * NBp + blk;
    if (blk >= NB) {
        if (lane < 4) reinterpret_cast<uint4 *>(cen + sb * 32)[lane] = make_uint4(0u, 0u, 0u, 0u);
        if (lane == 0) { rad[sb] = -INFINITY; bn[sb] = 0.f; }
        return;
    }
    const int BW = (W + 7) >> 3, by = blk / BW, bx = blk - by * BW;
    const int ry = lane >> 3, dq = lane & 7;
    const int y = by * 8 + ry, ch = H - by * 8 < 8 ? H - by * 8 : 8, cw = W - bx * 8 < 8 ? W - bx * 8 : 8;
    const bool rowok = y < H;
    float v[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool ok = rowok && i < cw;
        const size_t row = (size_t)b * H * W + (ok ? (size_t)y * W + bx * 8 + i : (size_t)by * 8 * W + bx * 8);
        union { uint2 q; _Float16 h[4]; } A;
        A.q = *reinterpret_cast<const uint2 *>(hi + row * 32 + dq * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[i][k] = (float)A.h[k];
        if (lo) {
            A.q = *reinterpret_cast<const uint2 *>(lo + row * 32 + dq * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[i][k] += (float)A.h[k];
        }
        if (!ok) { v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f; }
    }
    const float inv = 1.0f / (float)(ch * cw);
    float cf[4], c2 = 0.f;
    union { unsigned short h[4]; uint2 q; } C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float sum = ((v[0][k] + v[1][k]) + (v[2][k] + v[3][k])) + ((v[4][k] + v[5][k]) + (v[6][k] + v[7][k]));
#pragma unroll
        for (int sh = 8; sh <= 32; sh <<= 1) sum += __shfl_xor(sum, sh, 64);        // over the 8 tile rows
        const _Float16 c16 = (_Float16)(sum * inv);
        C.h[k] = __builtin_bit_cast(unsigned short, c16);
        cf[k] = (float)c16;
        c2 = fmaf(cf[k], cf[k], c2);
    }
    float dmax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float d2 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float e = v[i][k] - cf[k]; d2 = fmaf(e, e, d2); }
#pragma unroll
        for (int sh = 1; sh <= 4; sh <<= 1) d2 += __shfl_xor(d2, sh, 64);           // over the 8 dimension quads
        dmax = (rowok && i < cw) ? fmaxf(dmax, d2) : dmax;
    }
#pragma unroll
    for (int sh = 1; sh <= 4; sh <<= 1) c2 += __shfl_xor(c2, sh, 64);
#pragma unroll
    for (int sh = 8; sh <= 32; sh <<= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, sh, 64));
    if (ry == 0) reinterpret_cast<uint2 *>(cen + sb * 32)[dq] = C.q;
    if (lane == 0) {
        const float r = sqrtf(dmax) * 1.000244140625f;                          // (1 + 2^-12): the fp32 rounding of the distance
        rad[sb] = r;
        bn[sb] = sqrtf(c2) * 1.000244140625f + r;                               // >= ||x|| for every row of the block
    }
}

// one wave per query: the exact-enough maximum over the rows of the block the centroid search chose -> mlb, ||q||
template <bool SPLIT>
__global__ void __launch_bounds__(kThreads)
k_frnn_seed_lb(const unsigned short *__restrict__ Qhi, const unsigned short *__restrict__ Qlo, const int32_t *__restrict__ qidx,
               const unsigned short *__restrict__ Dhi, const unsigned short *__restrict__ Dlo, const float *__restrict__ bn,
               unsigned long long *__restrict__ keysC, float *__restrict__ mlb, float *__restrict__ qnorm, int S_all, int NQ,
               int H, int W, int NB, int NBp, const int32_t *__restrict__ qlist, const int32_t *__restrict__ qcount) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = qcount ? qcount[b] : S_all;
    const int q = blockIdx.x * (kThreads / 64) + wave;
    if (q >= S) return;
    const size_t kb = (size_t)b * S_all;
    const int slot = qlist ? qlist[kb + q] : q;
    int qr = qidx ? qidx[kb + slot] : slot;
    qr = qr < 0 ? 0 : (qr >= NQ ? NQ - 1 : qr);
    const unsigned long long key = keysC[kb + slot];
    int bs = key == 0ull ? 0 : (int)(0xffffffffu - (unsigned)(key & 0xffffffffull));
    bs = bs < 0 ? 0 : (bs >= NB ? NB - 1 : bs);
    float qv[32], x[32];
    load_row32(Qhi, SPLIT ? Qlo : nullptr, (size_t)b * NQ + qr, qv);
    const int BW = (W + 7) >> 3, by = bs / BW, bx = bs - by * BW;
    const int py = by * 8 + (lane >> 3), px = bx * 8 + (lane & 7);
    const bool ok = py < H && px < W;
    load_row32(Dhi, SPLIT ? Dlo : nullptr, (size_t)b * H * W + (ok ? (size_t)py * W + px : (size_t)by * 8 * W + bx * 8), x);
    float sc = 0.f, q2 = 0.f;
#pragma unroll
    for (int d = 0; d < 32; ++d) { sc = fmaf(qv[d], x[d], sc); q2 = fmaf(qv[d], qv[d], q2); }
    sc = ok ? sc : -INFINITY;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) sc = fmaxf(sc, __shfl_xor(sc, sh, 64));
    if (lane == 0) {
        const float qn = sqrtf(q2) * 1.000244140625f;
        mlb[kb + slot] = sc - kSlack * qn * bn[(size_t)b * NBp + bs];
        qnorm[kb + slot] = qn;
        keysC[kb + slot] = 0ull;                                                 // the scratch keys are left zero
    }
}

// surv [P][nqt_all][NBp / 64] words: bit i of word w = block 64 w + i may hold the maximum of one of the tile's 16 queries.
// <q, c> on the matrix core with the QUERIES as the A operand: a lane then holds one block (column lane & 15) against four
// queries (rows 4 g + j), ORs its four tests, two shuffles OR the four lane groups and ONE ballot yields the tile's 16
// block bits.  (With the blocks as rows every (block, lane group) bit had to be cut out of four ballots on the scalar
// unit - 120 instructions per MFMA, 82 us per search: more than the scoring of the survivors.)
constexpr int kQG = 4;                      // query tiles per wave in k_frnn_survivors
constexpr int kQE = 1;                      // query tiles per workgroup in k_frnn_eval (4: 135 instead of 55 us per search - the kernel is
                                            // bound by the serial work of a wave, not by the gathers the tiles would share)
constexpr int kEvalY = 4;                   // workgroups per query tile in k_frnn_eval (1 / 2 / 4 / 8: 1292 / 1148 / 1065 / 1084 us per matcher call)
template <bool SPLIT>
__global__ void __launch_bounds__(kThreads)
k_frnn_survivors(const unsigned short *__restrict__ Qhi, const unsigned short *__restrict__ Qlo, const int32_t *__restrict__ qidx,
                 const unsigned short *__restrict__ cen, const float *__restrict__ rad, const float *__restrict__ bn,
                 const float *__restrict__ mlb, const float *__restrict__ qnorm, unsigned long long *__restrict__ surv,
                 int32_t *__restrict__ total, int S_all, int NQ, int NBp, int nqt_all, int wpw,
                 const int32_t *__restrict__ qlist, const int32_t *__restrict__ qcount) {
    const int b = blockIdx.z, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, g = lane >> 4;
    const int S = qcount ? qcount[b] : S_all;
    const int qt0 = (blockIdx.x * (kThreads / 64) + wave) * kQG;                 // kQG consecutive query tiles share every centroid fragment
    if (qt0 * 16 >= S) return;
    const int nt = (S - qt0 * 16 + 15) / 16 < kQG ? (S - qt0 * 16 + 15) / 16 : kQG;   // wave-uniform
    const size_t kb = (size_t)b * S_all, sbase = (size_t)b * NBp;
    f16x8 qh[kQG], ql[SPLIT ? kQG : 1];
    float mq[kQG][4], nq[kQG][4];                                                // the four queries per tile this lane gets scores of
#pragma unroll
    for (int u = 0; u < kQG; ++u) {
        int q = (qt0 + u) * 16 + col;
        q = q < S ? q : S - 1;
        const int slot = qlist ? qlist[kb + q] : q;
        int qr = qidx ? qidx[kb + slot] : slot;
        qr = qr < 0 ? 0 : (qr >= NQ ? NQ - 1 : qr);
        qh[u] = *reinterpret_cast<const f16x8 *>(Qhi + ((size_t)b * NQ + qr) * 32 + g * 8);   // A: row = query col, k chunk g
        if (SPLIT) ql[u] = *reinterpret_cast<const f16x8 *>(Qlo + ((size_t)b * NQ + qr) * 32 + g * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int qq = (qt0 + u) * 16 + 4 * g + j;
            qq = qq < S ? qq : S - 1;
            const int sj = qlist ? qlist[kb + qq] : qq;
            mq[u][j] = mlb[kb + sj];
            nq[u][j] = qnorm[kb + sj];
        }
    }
    const int NBW = NBp / 64;
    const int w0 = blockIdx.y * wpw, w1 = w0 + wpw < NBW ? w0 + wpw : NBW;
    int cnt = 0;
    for (int w = w0; w < w1; ++w) {
        f16x8 c[4];
        float e[4], r[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {                                            // B: column = block col, k chunk g
            const size_t blk = sbase + w * 64 + t * 16 + col;
            c[t] = *reinterpret_cast<const f16x8 *>(cen + blk * 32 + g * 8);
            r[t] = rad[blk];
            e[t] = r[t] + kSlack * bn[blk];
        }
#pragma unroll
        for (int u = 0; u < kQG; ++u) {
            if (u >= nt) break;
            unsigned long long bits = 0ull;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(qh[u], c[t], acc, 0, 0, 0);
                if (SPLIT) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ql[u], c[t], acc, 0, 0, 0);
                bool keep = false;
#pragma unroll
                for (int j = 0; j < 4; ++j) keep = keep || !(acc[j] + nq[u][j] * e[t] < mq[u][j]);
                int k = (keep && !(r[t] < 0.f)) ? 1 : 0;                          // padding blocks have r = -inf
                k |= __shfl_xor(k, 16, 64);
                k |= __shfl_xor(k, 32, 64);
                bits |= (__ballot(k != 0) & 0xffffull) << (16 * t);
            }
            if (lane == 0) surv[((size_t)b * nqt_all + qt0 + u) * NBW + w] = bits;
            cnt += __popcll(bits);
        }
    }
    if (lane == 0 && cnt) atomicAdd(total + b, cnt);
}

// grid (groups of QG query tiles, Y, P): a group's 4 Y waves take the words of the UNION of its tiles' survivor rows
// round-robin (interleaved by 64-block word so that a cluster of survivors spreads).  Every gathered block is scored against
// all QG tiles of the group - consecutive tiles are neighbours on the seed grid (seed_order) and need nearly the same
// blocks, so the gathers, which bound this kernel (a block is 4 - 8 KB for 4 MFMAs per tile), are shared; scoring a block a
// tile did not ask for only adds real candidates.  The rows of the NEXT surviving block are requested before the current
// block is scored (two register sets, alternating: with one set and a copy the wait for the copy exposed every round trip).
template <int PASSES, int QG>
__global__ void __launch_bounds__(kThreads)
k_frnn_eval(const unsigned short *__restrict__ Qhi, const unsigned short *__restrict__ Qlo, const int32_t *__restrict__ qidx,
            const unsigned short *__restrict__ Dhi, const unsigned short *__restrict__ Dlo,
            const unsigned long long *__restrict__ surv, const int32_t *__restrict__ total, unsigned long long *__restrict__ keys,
            int S_all, int NQ, int H, int W, int NB, int NBp, int nqt_all, const int32_t *__restrict__ qlist,
            const int32_t *__restrict__ qcount) {
    const int b = blockIdx.z, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, g = lane >> 4;
    const int S = qcount ? qcount[b] : S_all;
    const int qt0 = blockIdx.x * QG;
    if (qt0 * 16 >= S || prune_overflow(total[b], S, NB)) return;
    const int nt = (S - qt0 * 16 + 15) / 16 < QG ? (S - qt0 * 16 + 15) / 16 : QG;   // workgroup-uniform
    const size_t kb = (size_t)b * S_all, db = (size_t)b * H * W;
    const int BW = (W + 7) >> 3;
    f16x8 qh[QG], ql[PASSES == 3 ? QG : 1];
    int slot[QG];
    float best[QG];
    int bestn[QG];
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        const int q = (qt0 + u) * 16 + col, qc = q < S ? q : S - 1;
        slot[u] = qlist ? qlist[kb + qc] : qc;
        int qr = qidx ? qidx[kb + slot[u]] : slot[u];
        qr = qr < 0 ? 0 : (qr >= NQ ? NQ - 1 : qr);
        qh[u] = *reinterpret_cast<const f16x8 *>(Qhi + ((size_t)b * NQ + qr) * 32 + g * 8);
        if (PASSES == 3) ql[u] = *reinterpret_cast<const f16x8 *>(Qlo + ((size_t)b * NQ + qr) * 32 + g * 8);
        best[u] = -INFINITY;
        bestn[u] = 0;
    }
    const int NBW = NBp / 64;
    const unsigned long long *srow = surv + ((size_t)b * nqt_all + qt0) * NBW;
    constexpr int NA = PASSES == 3 ? 8 : 4;
    const int stride = (kThreads / 64) * (int)gridDim.y, first = wave + (kThreads / 64) * (int)blockIdx.y;
    for (int wbase = first; wbase < NBW; wbase += 64 * stride) {                // 64 words of this wave at a time (one per lane)
        const int wmine = wbase + lane * stride;
        unsigned long long mine = 0ull;
        if (wmine < NBW)
            for (int u = 0; u < nt; ++u) mine |= srow[(size_t)u * NBW + wmine];
        int k = -1;
        unsigned long long bits = 0ull;
        auto next_block = [&]() -> int {                                         // wave-uniform; -1 = no more
            while (bits == 0ull) {
                if (++k >= 64 || wbase + k * stride >= NBW) return -1;
                const unsigned lo32 = __builtin_amdgcn_readlane((unsigned)mine, k), hi32 = __builtin_amdgcn_readlane((unsigned)(mine >> 32), k);
                bits = ((unsigned long long)hi32 << 32) | lo32;
            }
            const int i = __builtin_ctzll(bits);
            bits &= bits - 1ull;
            return (wbase + k * stride) * 64 + i;
        };
        auto load_rows = [&](int blk, f16x8 (&a)[NA]) {                          // A rows 16 h + col = pixel (y0 + 2 h + (col >> 3), x0 + (col & 7))
            const int by = blk / BW, bx = blk - by * BW;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                int py = by * 8 + 2 * h + (col >> 3), px = bx * 8 + (col & 7);
                py = py < H ? py : H - 1;
                px = px < W ? px : W - 1;
                const size_t row = db + (size_t)py * W + px;
                a[h] = *reinterpret_cast<const f16x8 *>(Dhi + row * 32 + g * 8);
                if (PASSES == 3) a[4 + h] = *reinterpret_cast<const f16x8 *>(Dlo + row * 32 + g * 8);
            }
        };
        auto score = [&](int blk, const f16x8 (&a)[NA]) {
            const int by = blk / BW, bx = blk - by * BW;
            const int y0 = by * 8, x0 = bx * 8;
            // this lane's outputs: tile h, j -> block row 16 h + 4 g + j = pixel (y0 + 2 h + (g >> 1), x0 + 4 (g & 1) + j)
            const int ox = x0 + 4 * (g & 1);
            const bool edge = y0 + 8 > H || x0 + 8 > W;                         // tiles on the bottom / right edge
#pragma unroll
            for (int u = 0; u < QG; ++u) {
                if (u >= nt) break;
                f32x4 acc[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h], qh[u], c, 0, 0, 0);   // the MFMA sequence of nn_step: same bits
                    if (PASSES == 3) {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[h], ql[u], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[4 + h], qh[u], c, 0, 0, 0);
                    }
                    acc[h] = c;
                }
                if (edge) {
#pragma unroll
                    for (int h = 0; h < 4; ++h)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (y0 + 2 * h + (g >> 1) >= H || ox + j >= W) acc[h][j] = -INFINITY;
                }
                float mx = acc[0][0];
#pragma unroll
                for (int h = 0; h < 4; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mx = __builtin_fmaxf(mx, acc[h][j]);
                const bool ge = mx >= best[u] && mx > -INFINITY;                // ties included: the LOWEST pixel index must win
                if (__any(ge)) {                                                // rare after the first few blocks
                    int nn = 0x7fffffff;                                        // lowest pixel index of this lane that holds mx
#pragma unroll
                    for (int h = 3; h >= 0; --h)
#pragma unroll
                        for (int j = 3; j >= 0; --j) nn = (acc[h][j] == mx) ? (y0 + 2 * h + (g >> 1)) * W + ox + j : nn;
                    if (ge && (mx > best[u] || nn < bestn[u])) { best[u] = mx; bestn[u] = nn; }
                }
            }
        };
        f16x8 ra[NA], rb[NA];
        int ba = next_block();
        if (ba >= 0) load_rows(ba, ra);
        while (ba >= 0) {
            const int bb = next_block();
            if (bb >= 0) load_rows(bb, rb);
            score(ba, ra);
            if (bb < 0) break;
            ba = next_block();
            if (ba >= 0) load_rows(ba, ra);
            score(bb, rb);
        }
    }
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        if (u >= nt) break;
        float bs = best[u];
        int bn_ = bs == -INFINITY ? 0 : bestn[u];
#pragma unroll
        for (int sh = 16; sh <= 32; sh <<= 1) {
            const float os = __shfl_xor(bs, sh, 64);
            const int on = __shfl_xor(bn_, sh, 64);
            if (os > bs || (os == bs && on < bn_)) { bs = os; bn_ = on; }
        }
        if (g == 0 && (qt0 + u) * 16 + col < S) {
            const unsigned long long key = ((unsigned long long)ordered_bits(bs) << 32) | (unsigned long long)(0xffffffffu - (unsigned)bn_);
            atomicMax(keys + kb + slot[u], key);
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
k_frnn_compact(const uint8_t *__restrict__ active, const int32_t *__restrict__ order, int32_t *__restrict__ list,
               int32_t *__restrict__ count, int S) {
    __shared__ int base, wsum[kThreads / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int s0 = 0; s0 < S; s0 += kThreads) {
        const int pos = s0 + tid;
        const int sl = (order && pos < S) ? order[pos] : pos;                   // walk the slots in the caller's order (NULL: ascending)
        const bool a = pos < S && sl >= 0 && sl < S && active[(size_t)b * S + sl] != 0;
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
                                   (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, 0);
    else hipLaunchKernelGGL(k_nn_mfma<3>, grid, blk, 0, st, qhi, qlo, (const int32_t *)nullptr, dhi, dlo, keys, S, S, N, per_split,
                            (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, 0);
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
    if (in_f16) hipLaunchKernelGGL(k_nn_mfma<1>, grid, blk, 0, st, qhi, qlo, qidx, dhi, dlo, keys, S, NQ, N, per_split, qlist, qcount,
                                   (const int32_t *)nullptr, 0);
    else hipLaunchKernelGGL(k_nn_mfma<3>, grid, blk, 0, st, qhi, qlo, qidx, dhi, dlo, keys, S, NQ, N, per_split, qlist, qcount,
                            (const int32_t *)nullptr, 0);
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
    hipLaunchKernelGGL(k_frnn_compact, dim3(P), eb, 0, st, (const uint8_t *)active, (const int32_t *)nullptr, list, count, S);
    frnn_search(packed1, N1, cur, packed2, N2, keys, P, S, in_f16, st, list, count);
    hipLaunchKernelGGL(k_frnn_mid, eg, eb, 0, st, keys, xy2_ws, total);
    frnn_search(packed2, N2, xy2_ws, packed1, N1, keys, P, S, in_f16, st, list, count);
    hipLaunchKernelGGL(k_frnn_end, eg, eb, 0, st, keys, (const int32_t *)xy2_ws, cur, active, got1, got2, total);
    M3_CHECK_LAUNCH("m3_frnn_round_active");
    return M3_OK;
}

// ---- block-bound search: statistics of a packed map, scratch, and the round built on it ---------------------------------
static inline int prune_nb(int H, int W) { return m3_cdiv(H, 8) * m3_cdiv(W, 8); }
static inline int prune_nbp(int H, int W) { return m3_cdiv(prune_nb(H, W), 64) * 64; }
// stats of a packed map [P][H * W]: centroids fp16 [P][NBp][32], then rad fp32 [P][NBp], then bn fp32 [P][NBp]
int64_t m3_frnn_stats_bytes(int P, int H, int W) {
    if (P <= 0 || H <= 0 || W <= 0) return 0;
    return (int64_t)P * prune_nbp(H, W) * (64 + 4 + 4);
}
int m3_frnn_blockstats(const void *packed, void *stats, int P, int H, int W, int in_f16, void *stream) {
    M3_REQUIRE(packed && stats && P > 0 && P <= 65535 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 31));
    M3_REQUIRE((reinterpret_cast<size_t>(stats) & 15) == 0);
    const unsigned short *hi = (const unsigned short *)packed, *lo = in_f16 ? nullptr : hi + (size_t)P * H * W * 32;
    const int NB = prune_nb(H, W), NBp = prune_nbp(H, W);
    unsigned short *cen = (unsigned short *)stats;
    float *rad = reinterpret_cast<float *>(cen + (size_t)P * NBp * 32), *bn = rad + (size_t)P * NBp;
    hipLaunchKernelGGL(k_frnn_blockstats, dim3(m3_cdiv(NBp, kThreads / 64), P), dim3(kThreads), 0, (hipStream_t)stream, hi, lo, cen, rad, bn,
                       H, W, NB, NBp);
    M3_CHECK_LAUNCH("m3_frnn_blockstats");
    return M3_OK;
}
// scratch of m3_frnn_round_pruned for S seeds
int64_t m3_frnn_prune_ws_bytes(int P, int S, int H1, int W1, int H2, int W2) {
    if (P <= 0 || S <= 0 || H1 <= 0 || W1 <= 0 || H2 <= 0 || W2 <= 0) return 0;
    const int64_t head = ((int64_t)P * S * 8 + (int64_t)P * 4 + 15) / 16 * 16;        // centroid-search keys + survivor totals (zeroed per search)
    const int nbp = prune_nbp(H1, W1) > prune_nbp(H2, W2) ? prune_nbp(H1, W1) : prune_nbp(H2, W2);
    return head + (int64_t)P * S * 8 + (int64_t)P * m3_cdiv(S, 16) * (nbp / 64) * 8;
}

static int frnn_search_pruned(const void *qpacked, int NQ, const int32_t *qidx, const void *dpacked, int H, int W, const void *dstats,
                              unsigned long long *keys, int P, int S, int in_f16, hipStream_t st, const int32_t *qlist,
                              const int32_t *qcount, void *ws) {
    const int N = H * W;
    const unsigned short *qhi = (const unsigned short *)qpacked, *qlo = in_f16 ? nullptr : qhi + (size_t)P * NQ * 32;
    const unsigned short *dhi = (const unsigned short *)dpacked, *dlo = in_f16 ? nullptr : dhi + (size_t)P * N * 32;
    const int NB = prune_nb(H, W), NBp = prune_nbp(H, W), NBW = NBp / 64, nqt = m3_cdiv(S, 16);
    const unsigned short *cen = (const unsigned short *)dstats;
    const float *rad = reinterpret_cast<const float *>(cen + (size_t)P * NBp * 32), *bn = rad + (size_t)P * NBp;
    const int64_t head = ((int64_t)P * S * 8 + (int64_t)P * 4 + 15) / 16 * 16;
    unsigned long long *keysC = (unsigned long long *)ws;
    int32_t *total = reinterpret_cast<int32_t *>(keysC + (size_t)P * S);
    float *mlb = reinterpret_cast<float *>((unsigned char *)ws + head), *qn = mlb + (size_t)P * S;
    unsigned long long *surv = reinterpret_cast<unsigned long long *>(qn + (size_t)P * S);
    M3_CHECK_HIP(hipMemsetAsync(ws, 0, (size_t)head, st), "m3_frnn_round_pruned/memset");
    const dim3 blk(kThreads);
    {   // 1. the most promising block per query: the brute-force kernel on the NB centroids (hi plane of the queries)
        const int qblocks = m3_cdiv(S, kQPB);
        int splits = m3_cdiv(1024, qblocks * P);
        if (splits < 1) splits = 1;
        // the centroid table is [P][NBp][32] (NBp = NB rounded up to 64): the search runs over NBp rows so that pair b's
        // rows start at b * NBp as k_frnn_blockstats laid them out (with N = NB every pair b > 0 scored the wrong rows
        // whenever NB % 64 != 0, e.g. 224 x 224: results stayed exact, the lower bound - and with it the pruning - did not);
        // the padding rows are zero vectors and k_frnn_seed_lb clamps the chosen block to < NB
        if (splits > m3_cdiv(NBp, kRows)) splits = m3_cdiv(NBp, kRows);
        int per_split = m3_cdiv(m3_cdiv(NBp, splits), kRows) * kRows;
        splits = m3_cdiv(NBp, per_split);
        hipLaunchKernelGGL(k_nn_mfma<1>, dim3(qblocks, splits, P), blk, 0, st, qhi, (const unsigned short *)nullptr, qidx, cen,
                           (const unsigned short *)nullptr, keysC, S, NQ, NBp, per_split, qlist, qcount, (const int32_t *)nullptr, 0);
    }
    const dim3 gl(m3_cdiv(S, kThreads / 64), P);
    const int ngrp = m3_cdiv(nqt, kQG);
    int ysplit = m3_cdiv(8192, ngrp * P);
    ysplit = ysplit < 1 ? 1 : (ysplit > NBW ? NBW : ysplit);
    const int wpw = m3_cdiv(NBW, ysplit);
    const dim3 gs(m3_cdiv(ngrp, kThreads / 64), m3_cdiv(NBW, wpw), P), ge(m3_cdiv(nqt, kQE), kEvalY, P);
    if (in_f16) {
        hipLaunchKernelGGL(k_frnn_seed_lb<false>, gl, blk, 0, st, qhi, qlo, qidx, dhi, dlo, bn, keysC, mlb, qn, S, NQ, H, W, NB, NBp, qlist, qcount);
        hipLaunchKernelGGL(k_frnn_survivors<false>, gs, blk, 0, st, qhi, qlo, qidx, cen, rad, bn, (const float *)mlb, (const float *)qn,
                           surv, total, S, NQ, NBp, nqt, wpw, qlist, qcount);
        hipLaunchKernelGGL((k_frnn_eval<1, kQE>), ge, blk, 0, st, qhi, qlo, qidx, dhi, dlo, (const unsigned long long *)surv,
                           (const int32_t *)total, keys, S, NQ, H, W, NB, NBp, nqt, qlist, qcount);
    } else {
        hipLaunchKernelGGL(k_frnn_seed_lb<true>, gl, blk, 0, st, qhi, qlo, qidx, dhi, dlo, bn, keysC, mlb, qn, S, NQ, H, W, NB, NBp, qlist, qcount);
        hipLaunchKernelGGL(k_frnn_survivors<true>, gs, blk, 0, st, qhi, qlo, qidx, cen, rad, bn, (const float *)mlb, (const float *)qn,
                           surv, total, S, NQ, NBp, nqt, wpw, qlist, qcount);
        hipLaunchKernelGGL((k_frnn_eval<3, kQE>), ge, blk, 0, st, qhi, qlo, qidx, dhi, dlo, (const unsigned long long *)surv,
                           (const int32_t *)total, keys, S, NQ, H, W, NB, NBp, nqt, qlist, qcount);
    }
    {   // 4. the brute-force search, gated on the survivor count (its workgroups leave at once when the bounds worked)
        const int qblocks = m3_cdiv(S, kQPB);
        int splits = m3_cdiv(1024, qblocks * P);
        if (splits < 1) splits = 1;
        if (splits > m3_cdiv(N, kRows)) splits = m3_cdiv(N, kRows);
        int per_split = m3_cdiv(m3_cdiv(N, splits), kRows) * kRows;
        splits = m3_cdiv(N, per_split);
        const dim3 grid(qblocks, splits, P);
        if (in_f16) hipLaunchKernelGGL(k_nn_mfma<1>, grid, blk, 0, st, qhi, qlo, qidx, dhi, dlo, keys, S, NQ, N, per_split, qlist, qcount,
                                       (const int32_t *)total, NB);
        else hipLaunchKernelGGL(k_nn_mfma<3>, grid, blk, 0, st, qhi, qlo, qidx, dhi, dlo, keys, S, NQ, N, per_split, qlist, qcount,
                                (const int32_t *)total, NB);
    }
    return M3_OK;
}

// m3_frnn_round / m3_frnn_round_active with the block-bound search: the maps are H1 x W1 / H2 x W2 pixels in raster order,
// stats1 / stats2 = m3_frnn_blockstats of packed1 / packed2, prune_ws = m3_frnn_prune_ws_bytes(...) bytes of scratch
// (16-byte aligned), act_ws as in m3_frnn_round_active or NULL (round on every seed slot).  seed_order int32 [S] (or NULL):
// the order in which the active slots are listed - the searches work on groups of 16 / 64 CONSECUTIVE list entries, and
// the fewer blocks the queries of a group need between them, the less is scored: an order that walks the seed grid in
// 4 x 4 patches (matching.py) halves the surviving work of a row-major one.  It changes no result (requires act_ws).
// Same cur / active / got1 / got2 as the brute-force rounds, bit for bit; the time follows how well the descriptor maps
// cluster.
int m3_frnn_round_pruned(const void *packed1, const void *packed2, const void *stats1, const void *stats2, int32_t *cur,
                         uint8_t *active, int32_t *got1, int32_t *got2, int32_t *xy2_ws, uint64_t *keys_ws, int32_t *act_ws,
                         const int32_t *seed_order, void *prune_ws, int P, int S, int H1, int W1, int H2, int W2, int in_f16,
                         void *stream) {
    M3_REQUIRE(packed1 && packed2 && stats1 && stats2 && cur && active && got1 && got2 && xy2_ws && keys_ws && prune_ws);
    M3_REQUIRE(P > 0 && P <= 65535 && S > 0 && H1 > 0 && W1 > 0 && H2 > 0 && W2 > 0 && (reinterpret_cast<size_t>(prune_ws) & 15) == 0);
    M3_REQUIRE((int64_t)H1 * W1 < (1ll << 31) && (int64_t)H2 * W2 < (1ll << 31) && (!seed_order || act_ws));
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(keys_ws);
    const int32_t *list = act_ws, *count = act_ws ? act_ws + (size_t)P * S : nullptr;
    const long long total = (long long)P * S;
    const int N1 = H1 * W1, N2 = H2 * W2;
    const dim3 eb(kThreads), eg((unsigned)m3_cdiv(total, (long long)kThreads));
    if (act_ws) hipLaunchKernelGGL(k_frnn_compact, dim3(P), eb, 0, st, (const uint8_t *)active, seed_order, act_ws, act_ws + (size_t)P * S, S);
    int rc = frnn_search_pruned(packed1, N1, cur, packed2, H2, W2, stats2, keys, P, S, in_f16, st, list, count, prune_ws);
    if (rc != M3_OK) return rc;
    hipLaunchKernelGGL(k_frnn_mid, eg, eb, 0, st, keys, xy2_ws, total);
    rc = frnn_search_pruned(packed2, N2, xy2_ws, packed1, H1, W1, stats1, keys, P, S, in_f16, st, list, count, prune_ws);
    if (rc != M3_OK) return rc;
    hipLaunchKernelGGL(k_frnn_end, eg, eb, 0, st, keys, (const int32_t *)xy2_ws, cur, active, got1, got2, total);
    M3_CHECK_LAUNCH("m3_frnn_round_pruned");
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
