// EXPERIMENT, not part of the library build (round 2; result: slower than k_gemm256, see the end of this header).
// To try it: copy into mast3r-slam_amd/csrc/, route launch_dense_big() in gemm.hip to m3_launch_gemm4w_dense, rebuild,
// and compare with tools/check_gemm_paths.py (digests must match) and tools/gemm_shapes.py.
//
// 256x256 bf16/fp16 MFMA GEMM with ONE wave per SIMD (4 waves x 512 registers) for the large dense problems of
// the ViT trunk (K % 128 == 0).
//
// k_gemm256 (gemm256.hip) runs two waves per SIMD in ping-pong phases with a 128x64 wave tile: 12 ds_read_b128 per
// 32 MFMAs and four workgroup barriers per 64 of K.  Here a wave owns a 128x128 sub-tile: 8 x 8 MFMA tiles = 256
// accumulator registers, pinned to the AGPR half of the register file by inline-asm MFMAs with "+a" operands (hipcc
// left to itself moves them through VGPRs and spills, profiles/r01_gemm_ablation.md).  16 fragment reads feed 64
// MFMAs (a third fewer LDS bytes per FLOP) and the software pipeline lives inside the wave:
//   * K advances in steps of 32 through a ring of four 32 KiB LDS slots ([A 256 x 64 B][W 256 x 64 B]);
//   * during the MFMAs of step s the wave reads the fragments of step s+1 (second register set) and issues the
//     LDS-DMA of step s+4 into the slot step s has just vacated, so a global load has three steps (about 3000
//     cycles) to land;
//   * one workgroup barrier per step, after `s_waitcnt vmcnt(16)` (everything but the two newest steps has landed).
// 64-byte LDS rows are swizzled chunk' = chunk ^ ((row >> 2) & 3) on the global-source side (LDS-DMA destinations are
// lane-linear), which makes every 16-lane ds_read_b128 group hit 16 distinct 16-byte bank groups.
//
// Measured on MI355X (tools/gemm_shapes.py, same process, us per launch, k_gemm256 -> this kernel):
//   16384x3072x1024 plain 92.4 -> 115.5, +RoPE 105.8 -> 189.9; 16384x4096x1024 GELU 136.9 -> 174.3;
//   16384x1024x4096 fp32 accumulate 118.1 -> 144.2; 8192x6400x7168 plain 725.1 -> 730.3.  Outputs bit-identical.
// Why: (1) an LDS-DMA issue costs the issuing wave 60-185 cycles (MI355X_MICROARCH.md, cycle constants); with two
// waves per SIMD the partner's MFMAs cover it, alone on the SIMD the 8 pieces of a step add ~600 cycles to its 1024
// MFMA cycles - register staging (global_load -> ds_write_b128) would be needed, at 64 more VGPRs for two steps in
// flight; (2) with 4 waves per CU nothing overlaps the epilogue of a tile: RoPE costs +74 us here against +13 us.
#include "../../mast3r-slam_amd/csrc/gemm_common.h"

using namespace m3gemm;

namespace {

constexpr int BM = 256, BN = 256, KS = 32;
constexpr int kThreads = 256;
constexpr int kSlotBytes = (BM + BN) * KS * 2;      // 32 KiB
constexpr int kLdsBytes4w = 4 * kSlotBytes;         // 128 KiB

template <int DT>
__device__ __forceinline__ void mfma_a(f32x4 &acc, const bf16x8 &a, const bf16x8 &b) {
    if constexpr (DT == DT_BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

template <int EPI, int DT>
__global__ void __launch_bounds__(kThreads, 1)
k_gemm4w(const GemmArgs gin) {
    const GemmArgs g = select_group<EPI>(gin, blockIdx.y);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;                  // wave sub-tile: rows wr*128.., columns wc*128..
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    constexpr int GM = 8;                                     // banded tile order as in k_gemm256
    const int band = bid / (GM * tiles_n), first_m = band * GM;
    const int gsz = tiles_m - first_m < GM ? tiles_m - first_m : GM;
    const int in_band = bid - band * GM * tiles_n;
    const int tm = first_m + in_band % gsz, tn = in_band / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    // LDS-DMA sources: wave-issue i of an operand covers rows i*64 + wave*16 + (lane >> 2), 16-byte chunk lane & 3
    const int srow = lane >> 2, sch = (lane & 3) ^ ((lane >> 4) & 3);
    const bf16_t *a_src[4], *w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + i * 64 + wave * 16 + srow, n = n0 + i * 64 + wave * 16 + srow;
        m = m < g.M ? m : g.M - 1;
        n = n < g.N ? n : g.N - 1;
        a_src[i] = g.A + (size_t)m * g.K + sch * 8;
        w_src[i] = g.W + (size_t)n * g.K + sch * 8;
    }
    unsigned char *const dma_dst = lds + wave * 1024;         // + slot*32768 + operand*16384 + i*4096
    const int nsteps = g.K / KS;                              // multiple of 4, >= 4

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fch = lane >> 4;
    const int swz = (fch ^ ((frow >> 2) & 3)) << 4;
    const unsigned char *const a_rd = lds + (wr * 128 + frow) * 64 + swz;              // + slot*32768 + i*1024
    const unsigned char *const w_rd = lds + 16384 + (wc * 128 + frow) * 64 + swz;      // + slot*32768 + j*1024

    bf16x8 af[2][8], wf[2][8];
    auto read_frag = [&](int set, int slot, int idx) {        // idx 0..7: W fragments, 8..15: A fragments
        if (idx < 8) wf[set][idx] = *reinterpret_cast<const bf16x8 *>(w_rd + slot * kSlotBytes + idx * 1024);
        else af[set][idx - 8] = *reinterpret_cast<const bf16x8 *>(a_rd + slot * kSlotBytes + (idx - 8) * 1024);
    };
    // idx 0..3: A issues, 4..7: W issues.  Every call moves its source pointer one K step on (the instruction's
    // immediate offset cannot carry the K position: the hardware adds it to the LDS address as well).
    auto dma = [&](int slot, int idx, int inc) {
        if (idx < 4) { glds16(a_src[idx], dma_dst + slot * kSlotBytes + idx * 4096); a_src[idx] += inc; }
        else { glds16(w_src[idx - 4], dma_dst + slot * kSlotBytes + 16384 + (idx - 4) * 4096); w_src[idx - 4] += inc; }
    };

    // prologue: steps 0..3 in flight; fragments of step 0 in set 0; slot 0 free again
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) dma(0, idx, KS);
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) dma(1, idx, KS);
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) dma(2, idx, KS);
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) dma(3, idx, nsteps > 4 ? KS : 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int idx = 0; idx < 16; ++idx) read_frag(0, 0, idx);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // One step: 64 MFMAs on register set U&1; between them the fragment reads of the next step and the DMA of step +4.
    // The last steps of the K loop run the same code: their DMAs re-read the last K step into slots nobody reads again
    // and their fragment reads feed no MFMA - a few KiB of L2 traffic per tile instead of a second copy of the body
    // (whose differently allocated registers cost a 256-register shuffle at the loop exit).
    auto step = [&](auto u_tag, int s) {
        constexpr int U = decltype(u_tag)::value;
        constexpr int cur = U & 1, nxt = cur ^ 1;
        const int k_inc = (s + U + 5 < nsteps) ? KS : 0;      // the sources stop on the last K step
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                mfma_a<DT>(acc[i][j], wf[cur][j], af[cur][i]);
                const int q = i * 8 + j;
                if ((q & 3) == 1) {
                    read_frag(nxt, (U + 1) & 3, q >> 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if ((q & 7) == 4) {
                    dma(U, q >> 3, k_inc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s < nsteps; s += 4) {
        step(std::integral_constant<int, 0>{}, s);
        step(std::integral_constant<int, 1>{}, s);
        step(std::integral_constant<int, 2>{}, s);
        step(std::integral_constant<int, 3>{}, s);
        // Inline-asm MFMAs are invisible to the hazard recogniser: nothing may read an accumulator until the last
        // MFMA has retired (18 wait states).  The register allocator is free to place copies right behind the loop,
        // so the distance is bought inside it, on the last trip only.
        if (s + 4 >= nsteps) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                             // every wave's trailing DMAs have landed
    __builtin_amdgcn_sched_barrier(0);
    // operand slots are dead after the last barrier: per-wave epilogue scratch (17 KiB each)
    epilogue_rows<EPI, 8, 8, DT>(g, acc, lds + wave * (64 * (32 * 8 + 16)), m0 + wr * 128, n0 + wc * 128, lane);
}

template <int DT>
int launch4w(const GemmArgs &a, int epi, hipStream_t st) {
    const int tiles = m3_cdiv(a.M, BM) * m3_cdiv(a.N, BN);
    dim3 grid(tiles, a.groups > 1 ? a.groups : 1), blk(kThreads);
#define M3_L(E)                                                                                              \
    case E: {                                                                                                \
        static bool attr_set = false;                                                                        \
        if (!attr_set) {                                                                                     \
            M3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gemm4w<E, DT>),               \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes4w),       \
                         "m3_gemm4w/attr");                                                                  \
            attr_set = true;                                                                                 \
        }                                                                                                    \
        hipLaunchKernelGGL((k_gemm4w<E, DT>), grid, blk, kLdsBytes4w, st, a);                                \
    } break
    switch (epi) {
        M3_L(EPI_BF16); M3_L(EPI_BF16_GELU); M3_L(EPI_F32); M3_L(EPI_F32_ACCUM); M3_L(EPI_BF16_RELU); M3_L(EPI_BF16_ADD); M3_L(EPI_BF16_ROPE);
        default: return M3_ERR_INVALID_ARG;
    }
#undef M3_L
    M3_CHECK_LAUNCH("m3_gemm4w");
    return M3_OK;
}

}  // namespace

// Dense problems only (MODE 0); requires K % 128 == 0.
int m3_launch_gemm4w_dense(const GemmArgs &a, int epi, hipStream_t st) {
    if (a.K % 128 != 0 || a.K < 128) return M3_ERR_INVALID_ARG;
    return a.dt == DT_F16 ? launch4w<DT_F16>(a, epi, st) : launch4w<DT_BF16>(a, epi, st);
}
