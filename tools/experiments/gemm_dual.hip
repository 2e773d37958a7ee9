// EXPERIMENT (round 3), not part of the library: 256x128x32 bf16/fp16 MFMA GEMM, TWO workgroups per CU.
// To rebuild it: copy into mast3r-slam_amd/csrc/, declare m3_launch_gemm_dual_dense in gemm.hip and route a forced tile
// code (m3_gemm_set_tile) to it in launch_dense_big.
// Outcome (tools/check_gemm_paths.py digests identical to k_gemm256; tools/gemm_shapes.py, 8 pairs, same process as
// torch.mm): 13-34 % SLOWER than the ping-pong kernel on every shape - enc qkv+RoPE 105.3 -> 121.6 us, fc1+GELU 137.5 ->
// 157.7, fc2+residual 119.5 -> 149.4, proj 47.5 -> 58.1, feature fc2 (K = 7168, 6 rounds) 1351 -> 1808 (1113 -> 831
// TFLOP/s).  The epilogue does overlap, but the K loop loses more than that returns: (i) a 256x128 tile stages
// 24 KiB per 32 of K for 4 waves = 6 LDS-DMA issues per wave and k-step against 4 in the 256x256 tile (an issue occupies
// the issuing wave for 60-185 cycles) and every k-step ends in a workgroup barrier; (ii) nothing but s_setprio orders
// the two workgroups, so the read / load phases and the MFMA blocks of the waves that share a SIMD collide instead of
// alternating - the barrier-enforced alternation of the ping-pong kernel is what keeps its matrix pipe at 77-79 %.
//
// Why: k_gemm256 (gemm256.hip) runs one 8-wave workgroup per CU.  Its K loop sits at the practical ceiling of the
// matrix pipe (77-79 % duty), but prologue and epilogue of a tile run with the pipe idle: at K = 1024 a 256x256 tile
// is 22 us of K loop and 8-12 us of first-load latency, GELU / RoPE / residual arithmetic, LDS transpose and stores -
// a quarter of every round.  There is no register room for a second accumulator set in that
// kernel and its two wave groups are one barrier phase apart by construction, so nothing of its own can cover the gap.
//
// Here the SAME wave tile (128 x 64 = 8 x 4 MFMA tiles, 12 ds_read_b128 per 32 MFMAs) is kept, but the two waves
// that share a SIMD belong to DIFFERENT workgroups (4 waves, 256 x 128 tile, 72 KiB LDS each): they are not tied by
// barriers, s_setprio around the MFMA block makes them fall into the read | MFMA alternation the ping-pong kernel
// enforces, and when one workgroup leaves its K loop for the epilogue (or is replaced by the next workgroup of the
// grid) the other keeps the matrix pipe busy.  K step 32 (one MFMA k-step per barrier, as many barriers per MFMA as
// before), three LDS stages of 24 KiB filled by LDS-DMA two steps ahead.
// LDS image per stage: A rows [256][64 B], then W rows [128][64 B]; 16-byte chunk c of row r is stored at chunk
// c ^ (2 * ((r >> 3) & 1)): with 64-byte rows the sixteen lanes the hardware serves together in a ds_read_b128
// (lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32) then cover all 64 banks exactly once.
// Accumulation order over K is that of the other tile shapes (k ascending in steps of 32): identical bits.
#include "gemm_common.h"

using namespace m3gemm;

namespace {

constexpr int BM = 256, BN = 128, BKS = 32;
constexpr int kThreads = 256, kStages = 3;
constexpr int kStageBytes = (BM + BN) * BKS * 2;          // 24 KiB
constexpr int kLdsBytes = kStages * kStageBytes;          // 72 KiB: two workgroups per CU

template <int EPI, int DT>
__global__ void __launch_bounds__(kThreads, 2)
k_gemm_dual(const GemmArgs gin) {
    constexpr int NI = 8, NJ = 4;
    const GemmArgs g = select_group<EPI>(gin, blockIdx.y);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;                // wave sub-tile: rows wr*128, cols wc*64

    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    constexpr int GM = 8;                                   // bands of 8 M-tiles, M fastest (see gemm256.hip)
    const int band = bid / (GM * tiles_n), first_m = band * GM;
    const int gsz = tiles_m - first_m < GM ? tiles_m - first_m : GM;
    const int in_band = bid - band * GM * tiles_n;
    const int tm = first_m + in_band % gsz, tn = in_band / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    // staging: thread t moves 16-byte slot t of each 4 KiB issue (64 rows x 64 B): A 4 issues, W 2
    const int srow = tid >> 2, sch = (tid & 3) ^ (((srow >> 3) & 1) << 1);
    const bf16_t *a_src[4];
    const bf16_t *w_src[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + i * 64 + srow;
        m = m < g.M ? m : g.M - 1;
        a_src[i] = g.A + (size_t)m * g.K + sch * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int n = n0 + i * 64 + srow;
        n = n < g.N ? n : g.N - 1;
        w_src[i] = g.W + (size_t)n * g.K + sch * 8;
    }
    const int nk = g.K / BKS;
    auto stage = [&](int kt, int buf) {
        unsigned char *base = lds + buf * kStageBytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(a_src[i] + (size_t)kt * BKS, base + i * 4096 + wave * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(w_src[i] + (size_t)kt * BKS, base + BM * BKS * 2 + i * 4096 + wave * 1024);
    };

    f32x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fch = lane >> 4;
    const int fsw = (fch ^ (((frow >> 3) & 1) << 1)) << 4;  // tile origins are multiples of 16 rows: bit 3 of the row = bit 3 of frow
    const int a_off0 = (wr * 128 + frow) * 64 + fsw;
    const int w_off0 = BM * BKS * 2 + (wc * 64 + frow) * 64 + fsw;

    bf16x8 af[NI], wf[NJ];
    auto read_frags = [&](int buf) {
        const unsigned char *base = lds + buf * kStageBytes;
#pragma unroll
        for (int j = 0; j < NJ; ++j) wf[j] = *reinterpret_cast<const bf16x8 *>(base + w_off0 + j * 1024);
#pragma unroll
        for (int i = 0; i < NI; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(base + a_off0 + i * 1024);
    };
    auto mfma_all = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16<DT>(wf[j], af[i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };

    // prologue: steps 0 and 1 in flight, step 0 landed and visible to the workgroup
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int nbuf = buf == kStages - 1 ? 0 : buf + 1;
        read_frags(buf);
        // stage kt + 2 goes into the buffer that was read in step kt - 1: every wave has retired those reads before the
        // barrier that ended step kt - 1
        const bool more = kt + 2 < nk;
        if (more) stage(kt + 2, nbuf == kStages - 1 ? 0 : nbuf + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mfma_all();
        __builtin_amdgcn_sched_barrier(0);
        // step kt + 1 must have landed: at most the six loads of step kt + 2 may still be in flight
        if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        buf = nbuf;
    }

    // epilogue: the stages are dead after the last barrier; each wave transposes its sub-tile through a private 9 KiB
    // LDS scratch and stores full rows (gemm_common.h)
    epilogue_rows<EPI, NI, NJ, DT>(g, acc, lds + wave * (64 * (32 * NJ + 16)), m0 + wr * 128, n0 + wc * 64, lane);
}

template <int DT>
int launch_dual(const GemmArgs &a, int epi, hipStream_t st) {
    const int tiles = m3_cdiv(a.M, BM) * m3_cdiv(a.N, BN);
    dim3 grid(tiles, a.groups > 1 ? a.groups : 1), blk(kThreads);
#define M3_L(E)                                                                                              \
    case E: {                                                                                                \
        static M3AttrOnce once;                                                                              \
        int dev__;                                                                                           \
        if (m3_attr_need(once, &dev__)) {                                                                    \
            M3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm_dual<E, DT>),            \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes),         \
                         "m3_gemm_dual/attr");                                                               \
            m3_attr_done(once, dev__);                                                                       \
        }                                                                                                    \
        hipLaunchKernelGGL((k_gemm_dual<E, DT>), grid, blk, kLdsBytes, st, a);                               \
    } break
    switch (epi) {
        M3_L(EPI_BF16); M3_L(EPI_BF16_GELU); M3_L(EPI_F32); M3_L(EPI_F32_ACCUM); M3_L(EPI_BF16_RELU); M3_L(EPI_BF16_ADD); M3_L(EPI_BF16_ROPE);
        default: return M3_ERR_INVALID_ARG;
    }
#undef M3_L
    M3_CHECK_LAUNCH("m3_gemm_dual");
    return M3_OK;
}

}  // namespace

// entry point used by gemm.hip's dispatcher (dense problems, K a multiple of 32)
int m3_launch_gemm_dual_dense(const GemmArgs &a, int epi, hipStream_t st) {
    return a.dt == DT_F16 ? launch_dual<DT_F16>(a, epi, st) : launch_dual<DT_BF16>(a, epi, st);
}
