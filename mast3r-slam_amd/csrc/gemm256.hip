// 256x256x64 "ping-pong" bf16 MFMA GEMM / implicit-GEMM conv for gfx950 - the large-problem path.
//
// Why: on the 128x128 kernel every wave issues 8 global_load_lds and 16 ds_read_b128 per 32 MFMAs
// and all waves of a SIMD do the same thing at the same time, so the matrix pipe idles while
// loads/LDS reads issue (ablation in profiles/r01_gemm_ablation.md: MFMA-only 1480 TFLOP/s,
// +LDS reads 1110, +global loads 960, both 760).  Here:
//   * tile 256x256, 8 waves (2 x 4), wave sub-tile 128x64 = 8x4 MFMA tiles: per 32 MFMAs only
//     4 global_load_lds and 12 ds_read_b128;
//   * one workgroup per CU (128 KiB LDS, two stages), so the two waves that share a SIMD belong to
//     the same workgroup: waves 0-3 ("ping") and 4-7 ("pong") run the SAME per-K-tile sequence
//          READ(ks0)+LOADS(A half of next tile) | MFMA(ks0) | READ(ks1)+LOADS(W half) | MFMA(ks1)
//     one phase apart, a workgroup barrier between phases - while one wave of a SIMD issues its
//     32 MFMAs the other one issues its LDS reads / global loads.  Each group has its own
//     straight-line loop (register liveness stays per phase); both execute the same barrier count.
// LDS image, swizzle, swapped-operand accumulator layout and epilogues are those of gemm.hip.
// (Round 5, measured and not kept: every global_load_lds issued in an MFMA phase of its wave - ping all eight in phase 1, pong
// four each in phases 0 and 2 - so that the READ phases carry fragment reads only: bit-identical, 3-5 % slower on every encoder
// shape, profiles/r05_gemm_duo_experiment.md.)
#include "gemm_common.h"

using namespace m3gemm;

namespace {

// Two tile widths share the kernel body:
//   BN = 256: waves 2(M) x 4(N), wave sub-tile 128 x 64  (8 x 4 MFMA tiles)
//   BN = 192: waves 4(M) x 2(N), wave sub-tile  64 x 96  (4 x 6 MFMA tiles) - for N = 768 * k (decoder):
//             256-wide tiles leave a quarter of the CUs idle there (e.g. 16384 x 768: 192 tiles on 256 CUs).
constexpr int BM = 256;
constexpr int kThreads = 512;

template <int MODE /*0 dense, 1 conv3x3*/, int EPI, int BN, int DT>
__global__ void __launch_bounds__(kThreads, 2)
k_gemm256(const GemmArgs gin) {
    constexpr int WN = BN == 256 ? 4 : 2, WM = 8 / WN;      // wave grid
    constexpr int NI = BM / WM / 16, NJ = BN / WN / 16;     // MFMA tiles per wave: 8 x 4 or 4 x 6
    constexpr int WISS = BN / 64;                           // 64-row global_load_lds issues of the W tile
    constexpr int kStageBytes = (BM + BN) * BK * 2;         // 64 KiB / 56 KiB
    static_assert(8 * (64 * (32 * NJ + 16)) + kRopeTableRows * 128 <= 2 * kStageBytes, "RoPE table behind the epilogue scratch");
    const GemmArgs g = select_group<EPI>(gin, blockIdx.y);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int group = wave >> 2;                            // 0 = ping (first wave of each SIMD), 1 = pong
    const int wr = wave / WN, wc = wave % WN;               // wave sub-tile: rows wr*16*NI, cols wc*16*NJ

    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    // Within an XCD's contiguous id range walk the tiles in bands of 8 M-tiles, M fastest: the ~32 tiles an
    // XCD runs at a time then form an 8 x 4 block (12 operand panels through its L2) instead of 2 x 16
    // (18 panels) - PMC FETCH_SIZE of the 16384x4096x1024 GEMM 297 -> ~200 MB.
    constexpr int GM = 8;
    const int band = bid / (GM * tiles_n), first_m = band * GM;
    const int gsz = tiles_m - first_m < GM ? tiles_m - first_m : GM;
    const int in_band = bid - band * GM * tiles_n;
    const int tm = first_m + in_band % gsz, tn = in_band / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    // staging: thread t moves 16-byte slot t of each 8 KiB issue (64 rows x 128 B); 4 issues per operand
    const int srow = tid >> 3, sch = (tid & 7) ^ ((srow >> 1) & 7);
    const bf16_t *a_src[4];
    const bf16_t *w_src[WISS];
    int a_oy[4], a_ox[4];
    const bf16_t *a_img[4];
#pragma unroll
    for (int i = 0; i < WISS; ++i) {
        int n = n0 + i * 64 + srow;
        n = n < g.N ? n : g.N - 1;
        w_src[i] = g.W + (size_t)n * g.K + sch * 8;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + i * 64 + srow;
        m = m < g.M ? m : g.M - 1;
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
    const int nk = g.K / BK;

    // K-tile staging in two halves (4 global_load_lds each) so the load issue is spread over phases
    auto stage_a = [&](int kt, int buf) {
        unsigned char *base = lds + buf * kStageBytes;
        int ky = 0, kx = 0, c0 = 0;
        if (MODE == 1) {                                     // K-tile kt = (64-channel slice q, tap): see conv_k_offset
            const int q = kt / 9, tap = kt - q * 9;
            c0 = q * BK;
            ky = tap / 3; kx = tap - ky * 3;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const void *src;
            if (MODE == 0) {
                src = a_src[i] + (size_t)kt * BK;
            } else {
                const int iy = a_oy[i] + ky, ix = a_ox[i] + kx;
                const bool in = (iy >= 0) && (iy < g.H) && (ix >= 0) && (ix < g.Wd);
                src = in ? (const void *)(a_img[i] + ((size_t)iy * g.Wd + ix) * g.Cin + c0) : (const void *)g.zero16;
            }
            glds16(src, base + i * 8192 + wave * 1024);
        }
    };
    auto stage_w = [&](int kt, int buf) {
        unsigned char *base = lds + buf * kStageBytes;
        const size_t koff = MODE == 1 ? conv_k_offset(kt, g.Cin) : (size_t)kt * BK;
#pragma unroll
        for (int i = 0; i < WISS; ++i)
            glds16(w_src[i] + koff, base + BM * BK * 2 + i * 8192 + wave * 1024);
    };
    auto stage = [&](int kt, int buf) { stage_a(kt, buf); stage_w(kt, buf); };

    f32x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fch = lane >> 4;
    // per-lane LDS byte offsets of the fragments (k-step 0; k-step 1 flips chunk bit 2 = byte 64)
    int a_off[NI], w_off[NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = wr * (16 * NI) + i * 16 + frow;
        a_off[i] = r * 128 + ((fch ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = wc * (16 * NJ) + j * 16 + frow;
        w_off[j] = BM * BK * 2 + r * 128 + ((fch ^ ((r >> 1) & 7)) << 4);
    }

    bf16x8 af[NI], wf[NJ];
    auto read_frags = [&](int buf, int ks) {
        const unsigned char *base = lds + buf * kStageBytes;
#pragma unroll
        for (int j = 0; j < NJ; ++j) wf[j] = *reinterpret_cast<const bf16x8 *>(base + (w_off[j] ^ (ks << 6)));
#pragma unroll
        for (int i = 0; i < NI; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(base + (a_off[i] ^ (ks << 6)));
        if (MODE == 1 && g.relu_a) {                         // kernel-uniform: relu(x) -> conv without a relu(x) tensor
#pragma unroll
            for (int i = 0; i < NI; ++i) af[i] = relu_frag(af[i]);
        }
    };
    auto mfma_all = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = mfma16<DT>(wf[j], af[i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto phase_end = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto phase_end_wait = [&]() {                          // retire this wave's loads of the next tile first
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // prologue: tile 0 into stage 0, visible to everyone; LayerNorm fold: the rows' mean / rstd table goes into the 2 KiB
    // behind the stages while those first loads fly (dense launches only)
    if (MODE == 0 && g.ln_stats) {                           // kernel-uniform; the statistics loads go out FIRST, the tile's
        float2 *lnt = reinterpret_cast<float2 *>(lds + 2 * kStageBytes);   // LDS-DMA behind them: the table arithmetic then
        const LnLoads L = ln_row_issue<BM>(g, m0, tid);                    // runs while the first K-tile is still in flight
        stage(0, 0);
        ln_table_store<BM>(lnt, ln_row_finish<BM>(g, L, m0, tid), tid);
    } else {
        stage(0, 0);
    }
    float2 *lnt = (MODE == 0 && g.ln_stats) ? reinterpret_cast<float2 *>(lds + 2 * kStageBytes) : nullptr;
    phase_end_wait();

    if (group == 0) {
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            read_frags(buf, 0);                            // phase 0
            if (kt + 1 < nk) stage_a(kt + 1, buf ^ 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            phase_end();
            mfma_all();                                    // phase 1
            phase_end();
            read_frags(buf, 1);                            // phase 2
            if (kt + 1 < nk) stage_w(kt + 1, buf ^ 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            phase_end();
            mfma_all();                                    // phase 3
            phase_end_wait();
        }
        phase_end();                                       // matches the pong group's drain phase
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt > 0) mfma_all();                        // phase 0: k-step 1 of the previous tile
            phase_end();
            read_frags(buf, 0);                            // phase 1
            if (kt + 1 < nk) stage_a(kt + 1, buf ^ 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            phase_end();
            if (kt + 1 < nk) stage_w(kt + 1, buf ^ 1);     // phase 2: loads issue under this wave's own MFMAs
            mfma_all();
            phase_end();
            read_frags(buf, 1);                            // phase 3
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            phase_end_wait();
        }
        mfma_all();                                        // drain: k-step 1 of the last tile
        phase_end();
    }

    // epilogue: the operand stages are dead after the last barrier; each wave transposes its sub-tile
    // through a private LDS scratch (9 / 13 KiB) and stores full rows (gemm_common.h)
    const float *ropet = nullptr;
    if constexpr (EPI == EPI_BF16_ROPE && MODE == 0) {
        if (g.rope_pos && g.rope_pmax > 0) {                 // kernel-uniform: the rotation's cos / sin table, once per workgroup, in
            float *tab = reinterpret_cast<float *>(lds + 8 * (64 * (32 * NJ + 16)));     // the 56 / 8 KiB of dead stages behind the
            rope_table_build(g, tab, tid, kThreads);                                      // eight waves' scratch
            __syncthreads();
            ropet = tab;
        }
    }
    epilogue_rows<EPI, NI, NJ, DT>(g, acc, lds + wave * (64 * (32 * NJ + 16)), m0 + wr * (16 * NI), n0 + wc * (16 * NJ), lane, lnt,
                                   wr * (16 * NI), ropet);
    if constexpr (MODE == 0 && (EPI == EPI_F32 || EPI == EPI_F32_ACCUM)) {
        if (g.stats_out) {                                   // kernel-uniform: the tile's statistics leaves -> slots of the sum tree
            __syncthreads();
            stats_tile_finalize<BM, NI, NJ, WN>(g, lds, 64 * (32 * NJ + 16), stats_stage_offset<NJ>(), m0, n0, tid);
        }
    }
}

template <int MODE, int BN, int DT>
int launch256(const GemmArgs &a, int epi, hipStream_t st) {
    constexpr int kLdsBytes = 2 * (BM + BN) * BK * 2 + BM * 8;    // 128 KiB / 112 KiB of stages + the LayerNorm-fold row table
    const int tiles = m3_cdiv(a.M, BM) * m3_cdiv(a.N, BN);
    dim3 grid(tiles, a.groups > 1 ? a.groups : 1), blk(kThreads);
#define M3_L(E)                                                                                              \
    case E: {                                                                                                \
        static M3AttrOnce once;                                                                              \
        int dev__;                                                                                           \
        if (m3_attr_need(once, &dev__)) {                                                                    \
            M3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm256<MODE, E, BN, DT>),        \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes),         \
                         "m3_gemm256/attr");                                                                 \
            m3_attr_done(once, dev__);                                                                       \
        }                                                                                                    \
        hipLaunchKernelGGL((k_gemm256<MODE, E, BN, DT>), grid, blk, kLdsBytes, st, a);                           \
    } break
    switch (epi) {
        M3_L(EPI_BF16); M3_L(EPI_BF16_GELU); M3_L(EPI_F32); M3_L(EPI_F32_ACCUM); M3_L(EPI_BF16_RELU); M3_L(EPI_BF16_ADD); M3_L(EPI_BF16_ROPE);
        default: return M3_ERR_INVALID_ARG;
    }
#undef M3_L
    M3_CHECK_LAUNCH("m3_gemm256");
    return M3_OK;
}

}  // namespace

// entry points used by gemm.hip's dispatcher
int m3_launch_gemm256_dense(const GemmArgs &a, int epi, int bn, hipStream_t st) {
    if (a.dt == DT_F16) return bn == 192 ? launch256<0, 192, DT_F16>(a, epi, st) : launch256<0, 256, DT_F16>(a, epi, st);
    return bn == 192 ? launch256<0, 192, DT_BF16>(a, epi, st) : launch256<0, 256, DT_BF16>(a, epi, st);
}
int m3_launch_gemm256_conv(const GemmArgs &a, int epi, hipStream_t st) {
    return a.dt == DT_F16 ? launch256<1, 256, DT_F16>(a, epi, st) : launch256<1, 256, DT_BF16>(a, epi, st);
}
