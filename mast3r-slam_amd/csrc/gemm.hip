// bf16 MFMA GEMM / implicit-GEMM convolution for gfx950 (CDNA4).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T + bias[N] )          (torch Linear layout, both K-major)
//
// Workgroup = 256 threads = 4 waves (2 x 2), tile 128(M) x 128(N) x 64(K); each wave owns a
// 64 x 64 sub-tile = 4 x 4 v_mfma_f32_16x16x32_bf16 accumulators (fp32).
// Staging: global -> LDS directly with global_load_lds_dwordx4 (no VGPR round trip), two LDS
// stages (64 KiB) so the loads of K-tile t+1 fly while tile t is multiplied.  LDS rows are 128 B
// (64 bf16); the 16-byte chunk index is XOR-swizzled with (row>>1)&7 - applied on the global
// SOURCE address (the LDS destination of a direct load is lane-linear) and again on the
// ds_read_b128 fragment reads - which makes every 16-lane read group conflict-free.
// The MFMA is issued "swapped" (W fragment as the A operand, activation fragment as B) so each
// lane ends up with 4 CONSECUTIVE output columns of one row: bias / GELU / residual epilogues
// and the stores are 8- or 16-byte vector ops.
//
// MODE_CONV3 turns the A operand into an on-the-fly im2col gather (implicit GEMM) of a 3x3,
// pad 1 convolution over an NHWC bf16 image: K = 9*Cin, K-tile kt -> tap (kt*64)/Cin; taps that
// fall outside the image read a 16-byte zero page instead.
#include <atomic>
#include "gemm_common.h"
#include <cmath>
#include <stdlib.h>

using namespace m3gemm;

namespace {

constexpr int BM = 128, BN = 128;                         // default tile; T = 64 gives 64x64 tiles (latency regime)
constexpr int kThreads = 256;

#ifndef M3_GEMM64_STAGES
#define M3_GEMM64_STAGES 2        // 4: three K-tiles in flight - measured at one pair per step: 21.9 vs 21.7 us per launch, no gain (the 64 x 64 tile is bound by LDS-DMA ISSUE, 4 pieces per 8 MFMAs and wave, not by load latency)
#endif
constexpr int k_gemm_stages(int T) { return T == 64 ? M3_GEMM64_STAGES : 2; }

template <int MODE /*0 dense, 1 conv3x3*/, int EPI, int T, int DT, bool SLICED = false>
__global__ void __launch_bounds__(kThreads)
k_gemm(const GemmArgs gin) {
    constexpr int BM = T, BN = T;
    constexpr int NT = T / 32;                               // MFMA tiles per wave and direction; 32-row load issues
    constexpr int kStageBytes = (BM + BN) * BK * 2;          // 32 KiB (16 KiB)
    // LDS stages (the loop below takes any power of two): two.  Four for T = 64 (three K-tiles in flight) were measured in
    // round 5 at one pair per step - no change, see M3_GEMM64_STAGES
    constexpr int STAGES = k_gemm_stages(T);
    static_assert(4 * (NT == 4 ? 9216 : 4608) + BM * 8 + kRopeTableRows * 128 <= STAGES * kStageBytes, "row table + RoPE table behind the scratch");
    GemmArgs g = select_group<EPI>(gin, blockIdx.y);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;               // wave position in the 2x2 grid (M, N)
    // split-K: this workgroup multiplies K-tiles [kt0, kt0 + nk) into its own fp32 partial plane
    int kt0 = 0, nk = g.K / BK;
    if (EPI == EPI_F32 && g.splits > 1) {
        const int all = nk, s = blockIdx.z;
        kt0 = (int)((long long)all * s / g.splits);
        nk = (int)((long long)all * (s + 1) / g.splits) - kt0;
        g.C = reinterpret_cast<float *>(g.C) + (size_t)s * g.M * g.ldc;
    }

    // XCD-aware tile order: consecutive tiles of one M-panel land on one XCD (shared A panel in L2)
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    const int bid = xcd_remap(blockIdx.x, nwg);
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-thread staging geometry: thread t moves 16-byte slot t of every 4 KiB issue ----
    // slot s -> LDS (row = s/8, chunk' = s%8); global chunk = chunk' ^ ((row>>1)&7)
    const int srow = tid >> 3, sch = (tid & 7) ^ ((srow >> 1) & 7);
    const bf16_t *a_src[NT];
    const bf16_t *w_src[NT];
    int a_oy[NT], a_ox[NT];                                 // conv: output pixel of the row
    const bf16_t *a_img[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        int m = m0 + i * 32 + srow;
        m = m < g.M ? m : g.M - 1;                          // clamp (stores are predicated)
        int n = n0 + i * 32 + srow;
        n = n < g.N ? n : g.N - 1;
        w_src[i] = g.W + (size_t)n * g.K + sch * 8;
        if (MODE == 0) {
            a_src[i] = g.A + (size_t)m * g.K + sch * 8;
        } else {
            const int pix = g.OH * g.OW;
            const int b = m / pix, rem = m - b * pix;
            a_oy[i] = (rem / g.OW) * g.stride - 1;
            a_ox[i] = (rem % g.OW) * g.stride - 1;
            a_img[i] = g.A + (size_t)b * g.H * g.Wd * g.Cin + sch * 8;
        }
    }
    auto stage = [&](int kt, int buf) {
        kt += kt0;
        unsigned char *base = lds + buf * kStageBytes;
        int ky = 0, kx = 0, c0 = 0;
        if (MODE == 1) {                                     // K-tile kt = (64-channel slice q, tap): see conv_k_offset
            const int q = kt / 9, tap = kt - q * 9;
            c0 = q * BK;
            ky = tap / 3; kx = tap - ky * 3;
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const void *src;
            if (MODE == 0) {
                src = a_src[i] + (size_t)kt * BK;
            } else {
                const int iy = a_oy[i] + ky, ix = a_ox[i] + kx;
                const bool in = (iy >= 0) && (iy < g.H) && (ix >= 0) && (ix < g.Wd);
                src = in ? (const void *)(a_img[i] + ((size_t)iy * g.Wd + ix) * g.Cin + c0) : (const void *)g.zero16;
            }
            glds16(src, base + i * 4096 + wave * 1024);
        }
        const size_t koff = MODE == 1 ? conv_k_offset(kt, g.Cin) : (size_t)kt * BK;
#pragma unroll
        for (int i = 0; i < NT; ++i)
            glds16(w_src[i] + koff, base + BM * BK * 2 + i * 4096 + wave * 1024);
    };

    f32x4 acc[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: lane -> row (lane&15), k-chunk (lane>>4) (+4 for the second k-step)
    const int frow = lane & 15, fch = lane >> 4;

    // Sliced accumulation (convolutions on 128 x 128 tiles only): when the split-K rule (pick_splits: a function of the
    // per-image geometry) asks for S slices but the BATCH already fills the chip with output tiles, one workgroup walks all S
    // slices itself and adds the slice sums in the finishing kernel's order - tot = ((p0 + p1) + p2) + ... - so the result
    // is bit-identical to S partial planes + k_splitk_finish without writing and re-reading the planes.
    // (its own instantiation: the second accumulator set costs 64 registers and the 2-waves-per-SIMD occupancy of the plain one)
    constexpr bool CAN_SLICE = SLICED;
    static_assert(!SLICED || (MODE == 1 && T == 128), "sliced accumulation: convolutions on 128 x 128 tiles");
    f32x4 tot[CAN_SLICE ? NT : 1][CAN_SLICE ? NT : 1];
    const int slices = CAN_SLICE ? g.slices : 1;
    int slice = 0, slice_end = slices > 1 ? (int)((long long)nk * 1 / slices) : nk;

    // LayerNorm fold (kernel-uniform): four threads per row pair fetch the rows' mean / rstd FIRST - the tile's loads go out
    // behind them - and carry them through the K loop in four registers; the table is written behind the waves' epilogue
    // scratch after the loop
    float4 ln_mr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 0 && g.ln_stats) {
        const LnLoads L = ln_row_issue<BM>(g, m0, tid);
#pragma unroll
        for (int t0 = 0; t0 < STAGES - 1; ++t0)
            if (t0 < nk) stage(t0, t0);
        ln_mr = ln_row_finish<BM>(g, L, m0, tid);
    } else {
#pragma unroll
        for (int t0 = 0; t0 < STAGES - 1; ++t0)
            if (t0 < nk) stage(t0, t0);
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & (STAGES - 1);
        // refill the stage the previous iteration finished with (its trailing barrier orders the fragment reads before this)
        if (kt + STAGES - 1 < nk) stage(kt + STAGES - 1, (kt + STAGES - 1) & (STAGES - 1));
        // wait for tile kt only: the 2*NT loads of each of the up to STAGES - 1 younger tiles may stay in flight
        const int ahead = nk - 1 - kt < STAGES - 1 ? nk - 1 - kt : STAGES - 1;
        if constexpr (STAGES == 2) {
            if (ahead) {
                if constexpr (NT == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            static_assert(STAGES == 2 || (STAGES == 4 && NT == 2), "vmcnt immediates below are for 4 loads per tile");
            if (ahead == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (ahead == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        lds_barrier();                          // every wave's loads of tile kt have landed
        const unsigned char *As = lds + buf * kStageBytes;
        const unsigned char *Ws = As + BM * BK * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[NT], wf[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int ra = wr * (T / 2) + i * 16 + frow;
                const int ca = (ks * 4 + fch) ^ ((ra >> 1) & 7);
                af[i] = *reinterpret_cast<const bf16x8 *>(As + ra * 128 + ca * 16);
                const int rw = wc * (T / 2) + i * 16 + frow;
                const int cw = (ks * 4 + fch) ^ ((rw >> 1) & 7);
                wf[i] = *reinterpret_cast<const bf16x8 *>(Ws + rw * 128 + cw * 16);
            }
            if (MODE == 1 && g.relu_a) {                    // kernel-uniform (M3_EPI_INPUT_RELU)
#pragma unroll
                for (int i = 0; i < NT; ++i) af[i] = relu_frag(af[i]);
            }
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = mfma16<DT>(wf[j], af[i], acc[i][j]);
        }
        lds_barrier();                          // all waves done reading `buf` (reads RETURNED) before it is restaged
        if constexpr (CAN_SLICE) {
            if (slices > 1 && kt + 1 == slice_end) {             // workgroup-uniform: a slice is complete
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        tot[i][j] = slice == 0 ? acc[i][j] : tot[i][j] + acc[i][j];
                        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                ++slice;
                slice_end = (int)((long long)nk * (slice + 1) / slices);
            }
        }
    }
    if constexpr (CAN_SLICE) {
        if (slices > 1) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = tot[i][j];
        }
    }

    if constexpr (EPI == EPI_RELU_HEAD4) {
        // DPT head tail fused into the last 3x3 convolution (N = 128 = one tile column): h = relu(acc + b) kept in
        // fp32 (the 16-bit rounding of this map was the largest single contribution to the pointmap error),
        // raw[o] = sum_n h[n] W4[o][n] + b4[o] (o < 4), then the pointmap post-processing of k_pts_post.  Saves
        // writing and re-reading the full-resolution 128-channel map (537 MB each way for 8 pairs) and two launches.
        static_assert(T == 128, "head fusion needs the 128-wide tile");
        const int r = lane & 15, gq = lane >> 4;
        float part[NT][4];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int o = 0; o < 4; ++o) part[i][o] = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = wc * 64 + j * 16 + gq * 4;
            const float4 b = g.bias ? *reinterpret_cast<const float4 *>(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            float w4[4][4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const uint2 q = *reinterpret_cast<const uint2 *>(g.W2 + (size_t)o * g.N + n);
                w4[o][0] = lo16<DT>(q.x); w4[o][1] = hi16<DT>(q.x); w4[o][2] = lo16<DT>(q.y); w4[o][3] = hi16<DT>(q.y);
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const f32x4 a = acc[i][j];
                const float h[4] = {fmaxf(a[0] + b.x, 0.f), fmaxf(a[1] + b.y, 0.f), fmaxf(a[2] + b.z, 0.f), fmaxf(a[3] + b.w, 0.f)};
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    part[i][o] += (h[0] * w4[o][0] + h[1] * w4[o][1]) + (h[2] * w4[o][2] + h[3] * w4[o][3]);
            }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float v = part[i][o];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                part[i][o] = v;
            }
        float *red = reinterpret_cast<float *>(lds);             // operand stages are dead after the last barrier
        if (wc == 1 && gq == 0) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
                *reinterpret_cast<float4 *>(red + (wr * 64 + i * 16 + r) * 4) =
                    make_float4(part[i][0], part[i][1], part[i][2], part[i][3]);
        }
        __syncthreads();
        if (wc == 0 && gq == 0) {
            float *pts = reinterpret_cast<float *>(g.C);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int m = m0 + wr * 64 + i * 16 + r;
                if (m >= g.M) continue;
                const float4 q = *reinterpret_cast<const float4 *>(red + (wr * 64 + i * 16 + r) * 4);
                const float x = part[i][0] + q.x + g.bias2[0], y = part[i][1] + q.y + g.bias2[1];
                const float z = part[i][2] + q.z + g.bias2[2], c = part[i][3] + q.w + g.bias2[3];
                const float d = sqrtf(x * x + y * y + z * z);
                const float sc = expm1f(d) / fmaxf(d, 1e-8f);
                pts[3 * (size_t)m + 0] = x * sc; pts[3 * (size_t)m + 1] = y * sc; pts[3 * (size_t)m + 2] = z * sc;
                g.C2[m] = 1.0f + expf(c);
            }
        }
    } else {
        float2 *lnt = nullptr;
        if (MODE == 0 && g.ln_stats) {                       // the stages are dead after the loop's last barrier
            lnt = reinterpret_cast<float2 *>(lds + 4 * (NT == 4 ? 9216 : 4608));
            ln_table_store<BM>(lnt, ln_mr, tid);
            __syncthreads();
        }
        // ---- epilogue: row-contiguous stores through a per-wave LDS scratch (gemm_common.h) --------
        const float *ropet = nullptr;
        if constexpr (EPI == EPI_BF16_ROPE && MODE == 0) {
            if (g.rope_pos && g.rope_pmax > 0) {             // kernel-uniform: cos / sin of every (position, frequency), once per
                float *tab = reinterpret_cast<float *>(lds + 4 * (NT == 4 ? 9216 : 4608) + BM * 8);   // workgroup, behind scratch + row table
                rope_table_build(g, tab, tid, kThreads);
                __syncthreads();
                ropet = tab;
            }
        }
        // (the RoPE instantiation of the 128 x 128 tile: two row tiles per pass instead of four - half the coefficient sets live -
        // keeps it at two workgroups per CU: 260 -> 2xx registers)
        epilogue_rows<EPI, NT, NT, DT, (EPI == EPI_BF16_ROPE && T == 128) ? 2 : 4>(g, acc, lds + wave * (NT == 4 ? 9216 : 4608), m0 + wr * (T / 2),
                                                                                  n0 + wc * (T / 2), lane, lnt, wr * (T / 2), ropet);
        if constexpr (MODE == 0 && (EPI == EPI_F32 || EPI == EPI_F32_ACCUM)) {
            if (g.stats_out) {                               // kernel-uniform: the tile's statistics leaves -> slots of the sum tree
                __syncthreads();
                stats_tile_finalize<BM, NT, NT, 2>(g, lds, NT == 4 ? 9216 : 4608, stats_stage_offset<NT>(), m0, n0, tid);
            }
        }
    }
}

template <int MODE, int T, int DT>
int launch_dt(const GemmArgs &a, int epi, hipStream_t st) {
    constexpr int kLdsBytes = k_gemm_stages(T) * (2 * T) * BK * 2;      // 64 KiB (T = 128: 2 stages; T = 64: 4 stages)
    const int tiles = m3_cdiv(a.M, T) * m3_cdiv(a.N, T);
    dim3 grid(tiles, a.groups > 1 ? a.groups : 1, a.splits > 1 ? a.splits : 1), blk(kThreads);
#define M3_L(E) case E: hipLaunchKernelGGL((k_gemm<MODE, E, T, DT>), grid, blk, kLdsBytes, st, a); break
    if (a.slices > 1) {
        if constexpr (MODE == 1 && T == 128) {
            switch (epi) {
                case EPI_BF16: hipLaunchKernelGGL((k_gemm<1, EPI_BF16, 128, DT, true>), grid, blk, kLdsBytes, st, a); break;
                case EPI_BF16_RELU: hipLaunchKernelGGL((k_gemm<1, EPI_BF16_RELU, 128, DT, true>), grid, blk, kLdsBytes, st, a); break;
                case EPI_BF16_ADD: hipLaunchKernelGGL((k_gemm<1, EPI_BF16_ADD, 128, DT, true>), grid, blk, kLdsBytes, st, a); break;
                default: return M3_ERR_INVALID_ARG;
            }
        } else return M3_ERR_INVALID_ARG;
    } else
    if (epi == EPI_RELU_HEAD4) {
        if constexpr (MODE == 1 && T == 128) hipLaunchKernelGGL((k_gemm<1, EPI_RELU_HEAD4, 128, DT>), grid, blk, kLdsBytes, st, a);
        else return M3_ERR_INVALID_ARG;
    } else
    switch (epi) {
        M3_L(EPI_BF16); M3_L(EPI_BF16_GELU); M3_L(EPI_F32); M3_L(EPI_F32_ACCUM); M3_L(EPI_BF16_RELU); M3_L(EPI_BF16_ADD); M3_L(EPI_BF16_ROPE);
        default: return M3_ERR_INVALID_ARG;
    }
#undef M3_L
    M3_CHECK_LAUNCH("m3_gemm");
    return M3_OK;
}

template <int MODE, int T = 128>
int launch(const GemmArgs &a, int epi, hipStream_t st) {
    return a.dt == DT_F16 ? launch_dt<MODE, T, DT_F16>(a, epi, st) : launch_dt<MODE, T, DT_BF16>(a, epi, st);
}

// Tile choice: the 256-row ping-pong kernel runs one workgroup per CU, the 128x128 kernel two.
// Estimated cost = rounds over the 256 CUs x work per tile (the ping-pong kernel is ~1.4x more
// efficient per FLOP once the grid fills the chip).  Its 192-wide variant is taken when N is a
// multiple of 192 and the tile count then divides into full rounds better (the decoder's N = 768,
// 1536, 2304: 256-wide tiles would leave a quarter of every round idle).  All three kernels
// accumulate K in the same order, so the choice never changes a result bit.
// M3_GEMM_TILE=128|192|256 forces a path (experiments).  Returns 128, 192 or 256.
static const double kCost64 = [] { const char *e = getenv("M3_GEMM_COST64"); return e ? atof(e) : 3.2; }();
static std::atomic<int> g_forced_tile{[] { const char *e = getenv("M3_GEMM_TILE"); return e ? atoi(e) : 0; }()};
int pick_tile(int M, int N, int groups = 1, bool dense = true) {
    const int forced = g_forced_tile.load(std::memory_order_relaxed);
    const bool can192 = dense && N % 192 == 0;
    if (forced == 128 || forced == 256) return forced;
    if (forced == 192) return can192 ? 192 : 256;
    const long t256 = (long)m3_cdiv(M, 256) * m3_cdiv(N, 256) * groups, t128 = (long)m3_cdiv(M, 128) * m3_cdiv(N, 128) * groups;
    const double c256 = (double)((t256 + 255) / 256) * 4.0 / 1.4, c128 = (double)((t128 + 511) / 512) * 2.0;
    double c192 = 1e30;
    if (can192) {
        const long t192 = (long)m3_cdiv(M, 256) * (N / 192) * groups;
        c192 = (double)((t192 + 255) / 256) * 3.0 / 1.4 * 1.02;        // 3/4 of the work per tile; ties go to 256
    }
    // 64x64 tiles (up to 4 workgroups per CU) for the latency regime: a quarter of the work per tile at
    // a round of 1024 of them costing ~0.8 of a round of 512 128x128 tiles (measured); only when the 128-tile grid leaves most of the chip idle
    double c64 = 1e30;
    if (dense) {
        const long t64 = (long)m3_cdiv(M, 64) * m3_cdiv(N, 64) * groups;
        c64 = (double)((t64 + 1023) / 1024) * 2.0 * 0.25 * kCost64;
    }
    if (forced == 64) return dense ? 64 : 128;
    if (c64 < c128 && c64 < c256 && c64 < c192) return 64;
    if (c192 < c256 && c192 < c128) return 192;
    return c256 < c128 ? 256 : 128;
}
bool use_256(int M, int N, int groups = 1) { return pick_tile(M, N, groups, false) == 256; }

// ---- split-K for problems that cannot fill the chip with output tiles (small feature maps, long K:
// the stride-2 / 768-channel DPT convolutions have 96 tiles and 108 K-tiles).  Each split writes an
// fp32 partial plane, a second kernel adds the planes in a FIXED order and applies the real epilogue.
// The split count is a function of the PER-IMAGE geometry only (never of the batch size), so a pair
// gives bitwise the same result alone, in a batch of 8 or in another rank's shard.  64 / tiles aims
// at ~512 workgroups for the 8-image batches of the benchmark shard.
int pick_splits(int pix_per_image, int N, int K) {
    const long tiles = (long)m3_cdiv(pix_per_image, BM) * m3_cdiv(N, BN);
    const int nk = K / BK;
    if (nk < 8) return 1;
    long s = 64 / tiles;
    if (s > nk / 4) s = nk / 4;
    if (s > 16) s = 16;
    return s < 2 ? 1 : (int)s;
}

template <int EPI, int DT>
__global__ void __launch_bounds__(256)
k_splitk_finish(const GemmArgs gin, const float *__restrict__ part, int S) {
    const GemmArgs g = select_group<EPI>(gin, blockIdx.y);
    part += (size_t)blockIdx.y * S * gin.M * gin.N;            // partial planes are laid out [group][slice][M][N]
    const int nq = g.N / 4;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    const int m = (int)(q / nq), n = (int)(q - (long)m * nq) * 4;
    if (m >= g.M) return;
    const size_t plane = (size_t)g.M * g.N;
    const float *p = part + (size_t)m * g.N + n;
    float4 a = *reinterpret_cast<const float4 *>(p);
    for (int s = 1; s < S; ++s) {
        const float4 b = *reinterpret_cast<const float4 *>(p + s * plane);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    store_tile<EPI, DT>(g, f32x4{a.x, a.y, a.z, a.w}, m, n);
}

template <int MODE>
int run_split(const GemmArgs &a, int S, int epi, void *ws, hipStream_t st) {
    if (MODE == 1 && (epi == EPI_BF16 || epi == EPI_BF16_RELU || epi == EPI_BF16_ADD)) {
        // enough output tiles to give every CU a workgroup: one pass, the S slice sums added in the finishing order (same bits)
        const long tiles = (long)m3_cdiv(a.M, BM) * m3_cdiv(a.N, BN) * (a.groups > 1 ? a.groups : 1);
        static const bool sliced_ok = [] { const char *e = getenv("M3_CONV_SLICED"); return !(e && atoi(e) == 0); }();
        if (sliced_ok && tiles >= m3_device_cu_count()) {
            GemmArgs q = a;
            q.slices = S; q.splits = 1;
            return launch<MODE>(q, epi, st);
        }
    }
    GemmArgs p = a;
    p.C = ws; p.ldc = a.N; p.bias = nullptr; p.bias2 = nullptr; p.R = nullptr; p.splits = S;
    p.c_gstride = (long long)S * a.M * a.N;                     // group 1's planes follow group 0's S planes
    const int rc = launch<MODE>(p, EPI_F32, st);
    if (rc != M3_OK) return rc;
    const dim3 grid((unsigned)m3_cdiv((long)a.M * (a.N / 4), 256L), a.groups > 1 ? a.groups : 1);
#define M3_F(E) case E:                                                                                        \
        if (a.dt == DT_F16) hipLaunchKernelGGL((k_splitk_finish<E, DT_F16>), grid, dim3(256), 0, st, a, (const float *)ws, S); \
        else hipLaunchKernelGGL((k_splitk_finish<E, DT_BF16>), grid, dim3(256), 0, st, a, (const float *)ws, S);      \
        break
    switch (epi) {
        M3_F(EPI_BF16); M3_F(EPI_BF16_GELU); M3_F(EPI_F32); M3_F(EPI_F32_ACCUM); M3_F(EPI_BF16_RELU); M3_F(EPI_BF16_ADD);
        default: return M3_ERR_INVALID_ARG;
    }
#undef M3_F
    M3_CHECK_LAUNCH("m3_gemm/splitk_finish");
    return M3_OK;
}

}  // namespace

int m3_launch_gemm256_dense(const GemmArgs &a, int epi, int bn, hipStream_t st);
int m3_launch_gemm256_conv(const GemmArgs &a, int epi, hipStream_t st);

// Large dense tiles go to the ping-pong kernel.  (A one-wave-per-SIMD 256x256 variant with AGPR accumulators was built
// and measured in round 2 - bit-identical, 20-80 % slower: tools/experiments/gemm4w.hip, DESIGN.md section 3.  Round 3: a
// 256x128x32 tile with TWO independent workgroups per CU, meant to run one tile's epilogue under the other's K loop -
// bit-identical, 13-34 % slower: tools/experiments/gemm_dual.hip, DESIGN.md section 10.  Round 5: the same idea with the
// ping-pong alternation kept INSIDE each of two 8-wave workgroups (256x128x64, 128 registers, 80 KiB of LDS each) -
// bit-identical, within -4 ... +9 % of k_gemm256, and every forced stagger of the two workgroups slower: the K loop is bound
// by the L2 -> LDS operand feed, which a 256x128 tile loads 1.5x harder: tools/experiments/gemm_duo.hip,
// profiles/r05_gemm_duo_experiment.md.)
static int launch_dense_big(const GemmArgs &a, int epi, int tile, hipStream_t st) {
    return m3_launch_gemm256_dense(a, epi, tile, st);
}

extern "C" {

int m3_gemm_set_tile(int tile) { return g_forced_tile.exchange(tile, std::memory_order_relaxed); }

int m3_gemm_pick_tile(int M, int N, int groups) {
    if (M <= 0 || N <= 0) return 0;
    return pick_tile(M, N, groups > 1 ? groups : 1);
}

static bool dt_ok(int dtype) { return dtype == DT_BF16 || dtype == DT_F16; }
// the RoPE projections also take M3_DT_F16_PVBF16 (2): an fp16 launch whose v columns (>= rope_cols) are stored as bf16
static bool dt_ok_rope(int dtype) { return dt_ok(dtype) || dtype == 2; }
static void set_dt(GemmArgs &a, int dtype) { a.dt = dtype == 2 ? DT_F16 : dtype; a.v_bf16 = dtype == 2; }

int m3_gemm_dt(const void *A, const void *W, const float *bias, void *C, const void *R, int M, int N, int K,
               int ldc, int epilogue, int dtype, void *stream) {
    M3_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0 && dt_ok(dtype));
    M3_REQUIRE(K % BK == 0 && N % 4 == 0 && ldc >= N && ldc % 4 == 0 && epilogue != EPI_BF16_ROPE);
    M3_REQUIRE(!((epilogue == EPI_F32_ACCUM || epilogue == EPI_BF16_ADD) && !R));
    GemmArgs a{};
    a.A = (const bf16_t *)A; a.W = (const bf16_t *)W; a.bias = bias; a.C = C; a.R = R;
    a.M = M; a.N = N; a.K = K; a.ldc = ldc; a.dt = dtype;
    const int tile = pick_tile(M, N);
    if (tile >= 192) return launch_dense_big(a, epilogue, tile, (hipStream_t)stream);
    if (tile == 64) return launch<0, 64>(a, epilogue, (hipStream_t)stream);
    return launch<0>(a, epilogue, (hipStream_t)stream);
}
int m3_gemm_bf16(const void *A, const void *W, const float *bias, void *C, const void *R, int M, int N, int K,
                 int ldc, int epilogue, void *stream) {
    return m3_gemm_dt(A, W, bias, C, R, M, N, K, ldc, epilogue, DT_BF16, stream);
}

int m3_gemm_rope_dt(const void *A, const void *W, const float *bias, void *C, int M, int N, int K, int ldc,
                    const float *rope_tok, int tokens_per_image, int rope_cols, int q_cols, float q_scale, int dtype,
                    void *stream) {
    M3_REQUIRE(A && W && C && rope_tok && M > 0 && N > 0 && K > 0 && tokens_per_image > 0 && dt_ok_rope(dtype));
    M3_REQUIRE((reinterpret_cast<size_t>(rope_tok) & 15) == 0 && q_cols >= 0 && q_cols <= rope_cols && q_cols % 64 == 0);
    M3_REQUIRE(K % BK == 0 && N % 64 == 0 && ldc >= N && ldc % 4 == 0 && rope_cols % 64 == 0 && rope_cols <= N);
    GemmArgs a{};
    a.A = (const bf16_t *)A; a.W = (const bf16_t *)W; a.bias = bias; a.C = C;
    a.M = M; a.N = N; a.K = K; a.ldc = ldc; set_dt(a, dtype);
    a.rope_tok = rope_tok; a.tokens_per_image = tokens_per_image; a.rope_cols = rope_cols;
    a.q_cols = q_cols; a.q_scale = q_scale;
    const int tile = pick_tile(M, N);
    if (tile >= 192) return launch_dense_big(a, EPI_BF16_ROPE, tile, (hipStream_t)stream);
    if (tile == 64) return launch<0, 64>(a, EPI_BF16_ROPE, (hipStream_t)stream);
    return launch<0>(a, EPI_BF16_ROPE, (hipStream_t)stream);
}
int m3_gemm_rope_pos_dt(const void *A, const void *W, const float *bias, void *C, int M, int N, int K, int ldc,
                        const int32_t *pos_yx, int tokens_per_image, float base, int rope_cols, int q_cols, float q_scale,
                        int dtype, void *stream) {
    M3_REQUIRE(A && W && C && pos_yx && M > 0 && N > 0 && K > 0 && tokens_per_image > 0 && base > 1.0f && dt_ok_rope(dtype));
    M3_REQUIRE(q_cols >= 0 && q_cols <= rope_cols && q_cols % 64 == 0);
    M3_REQUIRE(K % BK == 0 && N % 64 == 0 && ldc >= N && ldc % 4 == 0 && rope_cols % 64 == 0 && rope_cols <= N);
    GemmArgs a{};
    a.A = (const bf16_t *)A; a.W = (const bf16_t *)W; a.bias = bias; a.C = C;
    a.M = M; a.N = N; a.K = K; a.ldc = ldc; set_dt(a, dtype);
    a.rope_pos = pos_yx; a.rope_log2_base = log2f(base); a.tokens_per_image = tokens_per_image; a.rope_cols = rope_cols;
    a.q_cols = q_cols; a.q_scale = q_scale;
    const int tile = pick_tile(M, N);
    if (tile >= 192) return launch_dense_big(a, EPI_BF16_ROPE, tile, (hipStream_t)stream);
    if (tile == 64) return launch<0, 64>(a, EPI_BF16_ROPE, (hipStream_t)stream);
    return launch<0>(a, EPI_BF16_ROPE, (hipStream_t)stream);
}
int m3_gemm_bf16_rope(const void *A, const void *W, const float *bias, void *C, int M, int N, int K, int ldc,
                      const float *rope_tok, int tokens_per_image, int rope_cols, void *stream) {
    return m3_gemm_rope_dt(A, W, bias, C, M, N, K, ldc, rope_tok, tokens_per_image, rope_cols, 0, 1.0f, DT_BF16, stream);
}

// Two GEMMs of identical shape in one launch (the two decoder branches / the two heads):
// group g reads A + g*a_gstride, weights W[g], bias[g] and writes C + g*c_gstride.
int m3_gemm_grouped2_dt(const void *A, const void *W0, const void *W1, const float *bias0, const float *bias1,
                        void *C, const void *R, int M, int N, int K, int ldc, int64_t a_gstride,
                        int64_t c_gstride, int epilogue, const float *rope_tok,
                        int tokens_per_image, int rope_cols, int q_cols, float q_scale, int dtype, void *stream) {
    M3_REQUIRE(A && W0 && W1 && C && M > 0 && N > 0 && K > 0 && (dt_ok(dtype) || (dtype == 2 && epilogue == EPI_BF16_ROPE)));
    M3_REQUIRE(K % BK == 0 && N % 4 == 0 && ldc >= N && ldc % 4 == 0);
    M3_REQUIRE(!((epilogue == EPI_F32_ACCUM || epilogue == EPI_BF16_ADD) && !R));
    M3_REQUIRE((bias0 == nullptr) == (bias1 == nullptr));
    if (epilogue == EPI_BF16_ROPE)
        M3_REQUIRE(rope_tok && (reinterpret_cast<size_t>(rope_tok) & 15) == 0 && tokens_per_image > 0 && N % 64 == 0 &&
                   rope_cols % 64 == 0 && rope_cols <= N && q_cols >= 0 && q_cols <= rope_cols && q_cols % 64 == 0);
    GemmArgs a{};
    a.A = (const bf16_t *)A; a.W = (const bf16_t *)W0; a.W2 = (const bf16_t *)W1; a.bias = bias0; a.bias2 = bias1;
    a.C = C; a.R = R; a.M = M; a.N = N; a.K = K; a.ldc = ldc; set_dt(a, dtype);
    a.a_gstride = a_gstride; a.c_gstride = c_gstride; a.groups = 2;
    a.rope_tok = rope_tok; a.tokens_per_image = tokens_per_image; a.rope_cols = rope_cols;
    a.q_cols = q_cols; a.q_scale = q_scale;
    const int tile = pick_tile(M, N, 2);
    if (tile >= 192) return launch_dense_big(a, epilogue, tile, (hipStream_t)stream);
    if (tile == 64) return launch<0, 64>(a, epilogue, (hipStream_t)stream);
    return launch<0>(a, epilogue, (hipStream_t)stream);
}
int m3_gemm_grouped2_rope_pos_dt(const void *A, const void *W0, const void *W1, const float *bias0, const float *bias1,
                                 void *C, int M, int N, int K, int ldc, int64_t a_gstride, int64_t c_gstride,
                                 const int32_t *pos_yx, int tokens_per_image, float base, int rope_cols, int q_cols,
                                 float q_scale, int dtype, void *stream) {
    M3_REQUIRE(A && W0 && W1 && C && pos_yx && M > 0 && N > 0 && K > 0 && tokens_per_image > 0 && base > 1.0f && dt_ok_rope(dtype));
    M3_REQUIRE(K % BK == 0 && N % 64 == 0 && ldc >= N && ldc % 4 == 0 && (bias0 == nullptr) == (bias1 == nullptr));
    M3_REQUIRE(rope_cols % 64 == 0 && rope_cols <= N && q_cols >= 0 && q_cols <= rope_cols && q_cols % 64 == 0);
    GemmArgs a{};
    a.A = (const bf16_t *)A; a.W = (const bf16_t *)W0; a.W2 = (const bf16_t *)W1; a.bias = bias0; a.bias2 = bias1;
    a.C = C; a.M = M; a.N = N; a.K = K; a.ldc = ldc; set_dt(a, dtype);
    a.a_gstride = a_gstride; a.c_gstride = c_gstride; a.groups = 2;
    a.rope_pos = pos_yx; a.rope_log2_base = log2f(base); a.tokens_per_image = tokens_per_image; a.rope_cols = rope_cols;
    a.q_cols = q_cols; a.q_scale = q_scale;
    const int tile = pick_tile(M, N, 2);
    if (tile >= 192) return launch_dense_big(a, EPI_BF16_ROPE, tile, (hipStream_t)stream);
    if (tile == 64) return launch<0, 64>(a, EPI_BF16_ROPE, (hipStream_t)stream);
    return launch<0>(a, EPI_BF16_ROPE, (hipStream_t)stream);
}
int m3_gemm_bf16_grouped2(const void *A, const void *W0, const void *W1, const float *bias0, const float *bias1,
                          void *C, const void *R, int M, int N, int K, int ldc, int64_t a_gstride,
                          int64_t c_gstride, int epilogue, const float *rope_tok,
                          int tokens_per_image, int rope_cols, void *stream) {
    return m3_gemm_grouped2_dt(A, W0, W1, bias0, bias1, C, R, M, N, K, ldc, a_gstride, c_gstride, epilogue, rope_tok,
                               tokens_per_image, rope_cols, 0, 1.0f, DT_BF16, stream);
}

// widest node of the statistics' canonical tree for a stream of C columns (gemm_common.h): 256, 192 or 0 (no fold for this width)
static int ln_top_width(int C) {
    if (C % 192 == 0 && C / 192 <= 4) return 192;            // 768 = the decoder width: the 256-row kernel runs it in 192-wide tiles
    if (C % 256 == 0 && C / 256 <= 4) return 256;
    return 0;
}
// columns per statistics slot a [M, N] producer launch stores: the widest node of the tree its tile width is a multiple of.  Same
// decision in m3_ln_slot_count (the host sizes the buffer with it) and in m3_gemm_ex.
static int ln_store_width(int M, int N, int groups) {
    const int tile = pick_tile(M, N, groups), top = ln_top_width(N);
    if (top == 0) return 0;
    if (tile == top) return top;                             // 256-row kernel, one slot per tile
    if (top == 256 && tile == 128) return 128;
    return 64;                                               // every tile width is a multiple of 64
}
int m3_ln_slot_count(int M, int N, int groups) {
    if (M <= 0 || N <= 0) return 0;
    const int w = ln_store_width(M, N, groups > 1 ? 2 : 1);
    return w ? N / w : 0;
}

int m3_gemm_ex(const m3_gemm_desc *d, void *stream) {
    M3_REQUIRE(d && d->A && d->W && (d->C || d->c_lo) && d->M > 0 && d->N > 0 && d->K > 0);
    const int epi = d->epilogue, groups = d->groups > 1 ? 2 : 1;
    const bool rope = epi == EPI_BF16_ROPE;
    M3_REQUIRE(rope ? dt_ok_rope(d->dtype) : dt_ok(d->dtype));
    M3_REQUIRE(d->K % BK == 0 && d->N % 4 == 0 && d->ldc >= d->N && d->ldc % 4 == 0);
    M3_REQUIRE(!((epi == EPI_F32_ACCUM || epi == EPI_BF16_ADD) && !d->R));
    M3_REQUIRE(epi >= EPI_BF16 && epi <= EPI_BF16_ROPE);
    if (d->c_lo)                            // hi / lo stream: fp16 planes through the row-contiguous fp32 epilogue
        M3_REQUIRE((epi == EPI_F32 || epi == EPI_F32_ACCUM) && d->dtype == DT_F16 && d->c16 && !d->C && d->N % 8 == 0 && d->ldc % 8 == 0 &&
                   d->M % 2 == 0 && (reinterpret_cast<size_t>(d->c_lo) & 15) == 0 && (reinterpret_cast<size_t>(d->c16) & 15) == 0 &&
                   (epi == EPI_F32 || (d->r_lo && (reinterpret_cast<size_t>(d->R) & 15) == 0 && (reinterpret_cast<size_t>(d->r_lo) & 15) == 0)));
    if (groups == 2) M3_REQUIRE(d->W1 && (d->bias == nullptr) == (d->bias1 == nullptr));
    if (rope)
        M3_REQUIRE(d->rope_pos && d->tokens_per_image > 0 && d->rope_base > 1.0f && d->N % 64 == 0 && d->rope_cols % 64 == 0 &&
                   d->rope_cols <= d->N && d->q_cols >= 0 && d->q_cols <= d->rope_cols && d->q_cols % 64 == 0);
    const bool f32out = epi == EPI_F32 || epi == EPI_F32_ACCUM;
    if (d->c16 || d->stats_out) {          // LayerNorm fold, producer: the row-contiguous epilogue must be the one that runs
        M3_REQUIRE(f32out && (d->N % 64 == 0 || !d->stats_out) && d->M % 2 == 0 && d->ldc % 8 == 0 && (reinterpret_cast<size_t>(d->C) & 15) == 0 &&
                   (!d->R || (reinterpret_cast<size_t>(d->R) & 15) == 0) &&
                   (!d->c16 || (reinterpret_cast<size_t>(d->c16) & 15) == 0) &&
                   (!d->stats_out || (reinterpret_cast<size_t>(d->stats_out) & 15) == 0));
        if (groups == 2) M3_REQUIRE(d->c_gstride % 8 == 0 && d->stats_gstride % 4 == 0);
        if (d->stats_out) M3_REQUIRE(d->stats_slots > 0 && d->stats_slots == m3_ln_slot_count(d->M, d->N, groups));   // the buffer the host sized with it
    }
    if (d->ln_stats) {                     // LayerNorm fold, consumer
        const int top = ln_top_width(d->K);                 // the statistics arrive as pairs, halves or finished top nodes of the tree
        M3_REQUIRE(top != 0 && (d->ln_slots == d->K / 64 || (top == 256 && d->ln_slots == d->K / 128) || d->ln_slots == d->K / top));
        M3_REQUIRE(!f32out && epi != EPI_BF16_ADD && d->ln_colsum &&
                   d->ln_eps > 0.0f && d->M % 2 == 0 && (reinterpret_cast<size_t>(d->ln_stats) & 15) == 0 &&
                   (reinterpret_cast<size_t>(d->ln_colsum) & 15) == 0 && (reinterpret_cast<size_t>(d->C) & 15) == 0 &&
                   d->ldc % 8 == 0 && d->N % 8 == 0);
        if (groups == 2) M3_REQUIRE(d->ln_colsum1 && (reinterpret_cast<size_t>(d->ln_colsum1) & 15) == 0 && d->ln_gstride % 4 == 0);
    }
    GemmArgs a{};
    a.A = (const bf16_t *)d->A; a.W = (const bf16_t *)d->W; a.W2 = (const bf16_t *)d->W1; a.bias = d->bias; a.bias2 = d->bias1;
    a.C = d->C; a.R = d->R; a.M = d->M; a.N = d->N; a.K = d->K; a.ldc = d->ldc;
    if (rope) set_dt(a, d->dtype); else a.dt = d->dtype;
    a.a_gstride = d->a_gstride; a.c_gstride = d->c_gstride; a.groups = groups == 2 ? 2 : 0;
    if (rope) {
        a.rope_pos = d->rope_pos; a.rope_log2_base = log2f(d->rope_base); a.tokens_per_image = d->tokens_per_image;
        a.rope_cols = d->rope_cols; a.q_cols = d->q_cols; a.q_scale = d->q_scale;
        a.rope_pmax = (d->rope_max_pos > 0 && d->rope_max_pos <= kRopeTableRows) ? d->rope_max_pos : 0;
    }
    a.C16 = d->c16; a.stats_out = d->stats_out; a.stats_gstride = d->stats_gstride;
    a.R_lo = d->r_lo; a.C_lo = d->c_lo;
    a.ln_stats = d->ln_stats; a.ln_colsum = d->ln_colsum; a.ln_colsum2 = d->ln_colsum1; a.ln_slots = d->ln_slots;
    a.ln_eps = d->ln_eps; a.ln_gstride = d->ln_gstride;
    if (d->ln_stats) {
        const int top = ln_top_width(d->K);
        a.ln_tops = d->K / top;
        a.ln_gsz = d->ln_slots / a.ln_tops;                  // 1 .. 4 stored slots per top node
    }
    a.stats_w = d->stats_out ? ln_store_width(d->M, d->N, groups) : 0;
    const int tile = pick_tile(d->M, d->N, groups);
    if (tile >= 192) return launch_dense_big(a, epi, tile, (hipStream_t)stream);
    if (tile == 64) return launch<0, 64>(a, epi, (hipStream_t)stream);
    return launch<0>(a, epi, (hipStream_t)stream);
}

int64_t m3_conv3x3_splitk_bytes(int B, int H, int Wd, int Cin, int Cout, int stride) {
    if (B <= 0 || H <= 0 || Wd <= 0 || Cin <= 0 || Cout <= 0 || (stride != 1 && stride != 2)) return 0;
    const int OH = (H + 2 - 3) / stride + 1, OW = (Wd + 2 - 3) / stride + 1;
    const int S = pick_splits(OH * OW, Cout, 9 * Cin);
    return S > 1 ? (int64_t)S * B * OH * OW * Cout * 4 : 0;
}

int m3_conv3x3_dt(const void *X, const void *W, const float *bias, void *Y, const void *R, const void *zero16,
                  int B, int H, int Wd, int Cin, int Cout, int stride, int epilogue, void *splitk_ws,
                  int64_t splitk_ws_bytes, int dtype, void *stream) {
    const int relu_in = (epilogue & 0x100) ? 1 : 0;           // M3_EPI_INPUT_RELU
    epilogue &= 0xff;
    M3_REQUIRE(X && W && Y && zero16 && B > 0 && H > 0 && Wd > 0 && Cin > 0 && Cout > 0 && dt_ok(dtype));
    M3_REQUIRE(Cin % BK == 0 && Cout % 4 == 0 && (stride == 1 || stride == 2) && epilogue != EPI_BF16_ROPE);
    M3_REQUIRE(!((epilogue == EPI_F32_ACCUM || epilogue == EPI_BF16_ADD) && !R));
    GemmArgs a{};
    a.relu_a = relu_in;
    a.A = (const bf16_t *)X; a.W = (const bf16_t *)W; a.bias = bias; a.C = Y; a.R = R;
    a.zero16 = (const bf16_t *)zero16;
    a.H = H; a.Wd = Wd; a.Cin = Cin; a.stride = stride;
    a.OH = (H + 2 - 3) / stride + 1; a.OW = (Wd + 2 - 3) / stride + 1;
    a.M = B * a.OH * a.OW; a.N = Cout; a.K = 9 * Cin; a.ldc = Cout; a.dt = dtype;
    // split-K is decided by the per-image geometry alone; a shape that wants it must be given its scratch
    const int S = pick_splits(a.OH * a.OW, Cout, a.K);
    if (S > 1) {
        M3_REQUIRE(splitk_ws && (reinterpret_cast<size_t>(splitk_ws) & 15) == 0 &&
                   splitk_ws_bytes >= (int64_t)S * a.M * a.N * 4);
        return run_split<1>(a, S, epilogue, splitk_ws, (hipStream_t)stream);
    }
    if (use_256(a.M, a.N)) return m3_launch_gemm256_conv(a, epilogue, (hipStream_t)stream);
    return launch<1>(a, epilogue, (hipStream_t)stream);
}

// Two convolutions of identical shape in one launch (the two DPT heads: same maps, different weights):
// X [2,B,H,W,Cin], W0 / W1, bias0 / bias1, Y (and R) [2,B,OH,OW,Cout].  Split-K scratch: twice m3_conv3x3_splitk_bytes.
int m3_conv3x3_grouped2_dt(const void *X, const void *W0, const void *W1, const float *bias0, const float *bias1,
                           void *Y, const void *R, const void *zero16, int B, int H, int Wd, int Cin, int Cout,
                           int stride, int epilogue, void *splitk_ws, int64_t splitk_ws_bytes, int dtype, void *stream) {
    const int relu_in = (epilogue & 0x100) ? 1 : 0;           // M3_EPI_INPUT_RELU
    epilogue &= 0xff;
    M3_REQUIRE(X && W0 && W1 && Y && zero16 && B > 0 && H > 0 && Wd > 0 && Cin > 0 && Cout > 0 && dt_ok(dtype));
    M3_REQUIRE(Cin % BK == 0 && Cout % 4 == 0 && (stride == 1 || stride == 2) && epilogue != EPI_BF16_ROPE);
    M3_REQUIRE(!((epilogue == EPI_F32_ACCUM || epilogue == EPI_BF16_ADD) && !R));
    M3_REQUIRE((bias0 == nullptr) == (bias1 == nullptr));
    GemmArgs a{};
    a.relu_a = relu_in;
    a.A = (const bf16_t *)X; a.W = (const bf16_t *)W0; a.W2 = (const bf16_t *)W1; a.bias = bias0; a.bias2 = bias1;
    a.C = Y; a.R = R; a.zero16 = (const bf16_t *)zero16;
    a.H = H; a.Wd = Wd; a.Cin = Cin; a.stride = stride;
    a.OH = (H + 2 - 3) / stride + 1; a.OW = (Wd + 2 - 3) / stride + 1;
    a.M = B * a.OH * a.OW; a.N = Cout; a.K = 9 * Cin; a.ldc = Cout; a.dt = dtype;
    a.groups = 2; a.a_gstride = (long long)B * H * Wd * Cin; a.c_gstride = (long long)a.M * Cout;
    const int S = pick_splits(a.OH * a.OW, Cout, a.K);
    if (S > 1) {
        M3_REQUIRE(splitk_ws && (reinterpret_cast<size_t>(splitk_ws) & 15) == 0 &&
                   splitk_ws_bytes >= 2 * (int64_t)S * a.M * a.N * 4);
        return run_split<1>(a, S, epilogue, splitk_ws, (hipStream_t)stream);
    }
    if (use_256(a.M, a.N, 2)) return m3_launch_gemm256_conv(a, epilogue, (hipStream_t)stream);
    return launch<1>(a, epilogue, (hipStream_t)stream);
}

int m3_conv3x3_bf16(const void *X, const void *W, const float *bias, void *Y, const void *R, const void *zero16,
                    int B, int H, int Wd, int Cin, int Cout, int stride, int epilogue, void *splitk_ws,
                    int64_t splitk_ws_bytes, void *stream) {
    return m3_conv3x3_dt(X, W, bias, Y, R, zero16, B, H, Wd, Cin, Cout, stride, epilogue, splitk_ws, splitk_ws_bytes,
                         DT_BF16, stream);
}

// Last stage of the DPT head in one launch: Y = relu(conv3x3(X) + bias) (Cout = 128, never written),
// raw = Y . W4^T + b4 (4 channels), pts = xyz / |xyz| * expm1(|xyz|), conf = 1 + exp(raw[3]).
int m3_conv3x3_relu_head4_dt(const void *X, const void *W, const float *bias, const void *W4, const float *b4,
                             float *pts, float *conf, const void *zero16, int B, int H, int Wd, int Cin, int dtype,
                             void *stream) {
    M3_REQUIRE(X && W && W4 && b4 && pts && conf && zero16 && B > 0 && H > 0 && Wd > 0 && Cin > 0 && Cin % BK == 0);
    M3_REQUIRE(dt_ok(dtype));
    M3_REQUIRE((int64_t)B * H * Wd < (1ll << 31));
    GemmArgs a{};
    a.A = (const bf16_t *)X; a.W = (const bf16_t *)W; a.bias = bias; a.C = pts; a.C2 = conf;
    a.W2 = (const bf16_t *)W4; a.bias2 = b4;
    a.zero16 = (const bf16_t *)zero16;
    a.H = H; a.Wd = Wd; a.Cin = Cin; a.stride = 1; a.OH = H; a.OW = Wd;
    a.M = B * H * Wd; a.N = 128; a.K = 9 * Cin; a.ldc = 3; a.dt = dtype;
    return launch<1>(a, EPI_RELU_HEAD4, (hipStream_t)stream);
}
int m3_conv3x3_relu_head4(const void *X, const void *W, const float *bias, const void *W4, const float *b4,
                          float *pts, float *conf, const void *zero16, int B, int H, int Wd, int Cin, void *stream) {
    return m3_conv3x3_relu_head4_dt(X, W, bias, W4, b4, pts, conf, zero16, B, H, Wd, Cin, DT_BF16, stream);
}

}  // extern "C"
