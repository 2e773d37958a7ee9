// Fused (flash-style) multi-head attention for gfx950, head dim 64, bf16 in / bf16 out, fp32
// softmax and accumulation.  Serves the ViT encoder self-attention, the decoder self-attention
// and the decoder cross-attention (keys/values of the OTHER view via kv_batch_shift).
//
// Workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 query rows
// (two 16-row tiles) and walks the keys in tiles of 64.  Everything is computed TRANSPOSED so
// that a query row lives on one lane (q = lane & 15) for the whole kernel:
//     S^T = K . Q^T      (MFMA A = K fragment, B = Q fragment)  -> lane holds 16 keys of its q
//     O^T = V^T . P^T    (MFMA A = V^T fragment, B = P fragment)  -> lane holds 16 d of its q
// so the row max / row sum need only in-lane work plus two xor-shuffles, the rescale factor is
// lane-local, and P goes from the S accumulators to the next MFMA's B operand with a bf16 pack
// and no LDS round trip.  K and V tiles are staged global -> LDS by global_load_lds (double
// buffered); K fragments are ds_read_b128 from an XOR-swizzled image, V^T fragments come from
// the row-major V image through ds_read_b64_tr_b16 (hardware transpose), conflict-free with a
// chunk-pair swizzle.
#include "gemm_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

using m3gemm::bf16x8;
using m3gemm::f32x4;
using m3gemm::bf16_t;
using m3gemm::glds16;
using m3gemm::pack16;
using m3gemm::mfma16;
using m3gemm::DT_BF16;
using m3gemm::DT_F16;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;
constexpr int QROWS = 128;     // query rows per workgroup
constexpr int KT = 64;         // keys per tile
constexpr int HD = 64;         // head dim
constexpr int kTileBytes = KT * HD * 2;          // 8 KiB
constexpr int kLds = 4 * kTileBytes;             // K,V x 2 stages = 32 KiB

using m3gemm::pack_bf16;

struct AttnArgs {
    const bf16_t *Q, *K, *V;
    bf16_t *O;
    int q_row_stride, kv_row_stride, o_row_stride;       // elements between consecutive tokens
    long long q_batch_stride, kv_batch_stride, o_batch_stride;   // elements between batch items
    int Tq, Tk, heads, nbatch, kv_batch_shift;
    float scale_log2e;                                    // softmax scale * log2(e)
};

// QT = 16-row query tiles per wave: 2 -> 128 query rows per workgroup (throughput regime), 1 -> 64 rows
// per workgroup (twice the workgroups: used when the 128-row grid would leave CUs with < 2 workgroups,
// e.g. one pair = 2 images x 16 heads x 8 blocks = 256 workgroups on 256 CUs).
// Any Tq, Tk >= 1: query rows past Tq are computed on a clamped row and not stored; keys past Tk (last
// tile only) are staged from the clamped last row and their scores set to -inf before the softmax.
// MODE 0: classic online softmax; scale * log2(e) is applied per score.
// MODE 1, 2 ("prescaled"): q already carries softmax scale * log2(e) (folded in by the projection GEMM's RoPE
// epilogue), so a score is an exp2 argument as it leaves the matrix core, and the reference maximum m_ref of a query
// enters the S^T MFMA as its accumulator INITIALISER (C = -m_ref, lane-local): the MFMA returns s - m_ref.
//   MODE 1 (safe, any magnitude): the tile maximum is still computed; m_ref is raised (and o, l rescaled) only when it
//     exceeds the reference by more than kDefer (wave-uniform, rare after the first tiles).  Row sums as packed adds.
//   MODE 2 (fast, bf16 P operand): the kernel is bound by VALU ISSUE (PMC: VALU + MFMA issue = 96 % of the SIMD
//     cycles; per tile and wave 64 v_exp_f32 = 512 cycles, as many as its 32 MFMAs), so everything but the exp2 and
//     the 16-bit packing leaves the VALU: m_ref is the TRUE maximum of the first tile and never recomputed; later
//     tiles only need exp2(s - m_ref) to stay finite, and bf16 has the fp32 exponent range.  Softmax is invariant to
//     the reference, so a lagging one costs no accuracy - numerator and denominator carry the same factor.  The row
//     sums are one more MFMA per (query tile, k-step) against an all-ones A fragment - l accumulates in a matrix-core
//     register across tiles, already summed over the lane groups, and sums exactly the rounded P that multiplies V.
//     A per-lane test before the next tile (l > 2^60 -> scale o, l by 2^-64 and move m_ref) keeps the range.  Overflow
//     (a score far above the reference - not seen on any network input, but possible in principle) makes the WHOLE
//     workgroup recompute its block with the MODE 1 loop.  The test is sticky and conservative: l > 2^100 at a range
//     check or at the end (then o = sum p v may already be inf although l = sum p is finite - l alone would come back
//     into range and hide it: round-2 advisor finding), a non-finite l, or a non-finite o at the end.
#ifndef M3_ATTN_ABL
#define M3_ATTN_ABL 0       // experiments (timing-only, wrong results), bit mask on top of M3_ATTN_EXP: 1 = K fragments read once (no
#endif                      // per-tile ds_read_b128), 2 = V^T fragments read once, 4 = only tile 0 is staged (no global -> LDS traffic)
#ifndef M3_ATTN_EXP
#define M3_ATTN_EXP 0       // experiments (timing-only builds, wrong results): 4 = no exp2 (p = s), 5 = no exp2 and constant P (no packing)
#endif
constexpr float kDefer = 8.0f;
// PVDT: 16-bit type of V in memory and of the probabilities P, i.e. of the O^T = V^T . P^T product.  PVDT = DT except in
// the mixed mode of the fp16 trunk (DT = fp16, PVDT = bf16, M3_DT_F16_PVBF16): q and k - whose rounding is what an
// 8-bit mantissa costs on peaked softmax rows (logits of 30-80: DESIGN.md section 4) - keep fp16's 11 bits, while P and V
// take bf16's exponent range, which is what lets MODE 2 run without tracking the maximum.
template <int QT, int DT, int MODE, int PVDT = DT>
__global__ void __launch_bounds__(kThreads, (MODE == 1 && QT == 2) ? 3 : 4)
k_attn(const AttnArgs a) {
    static_assert(MODE != 2 || PVDT == DT_BF16, "the deferred-maximum loop needs the fp32 exponent range of bf16 for P");
    constexpr int QR = QT * 64;                                 // query rows per workgroup
    __shared__ __attribute__((aligned(16))) unsigned char lds[kLds];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 15, g = lane >> 4;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so give every XCD one contiguous
    // range of ids -- the Tq/128 query blocks of a (batch, head) then share its K/V through ONE L2
    // instead of pulling them into eight (measured 5x the compulsory HBM reads before this remap).
    const int nq = (a.Tq + QR - 1) / QR, nwg = gridDim.x;
    const int per = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
    const int id = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (blockIdx.x >> 3);
    const int qblk = id % nq, head = (id / nq) % a.heads, b = id / (nq * a.heads);
    const int kvb = (b + a.kv_batch_shift) % a.nbatch;
    const bf16_t *Qp = a.Q + (size_t)b * a.q_batch_stride + head * HD;
    const bf16_t *Kp = a.K + (size_t)kvb * a.kv_batch_stride + head * HD;
    const bf16_t *Vp = a.V + (size_t)kvb * a.kv_batch_stride + head * HD;

    // Q fragments (B operand): lane -> q row (lane&15), d = 32*ks + 8*g + j
    bf16x8 qf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int row = qblk * QR + wave * (16 * QT) + qt * 16 + lq;
        row = row < a.Tq ? row : a.Tq - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[qt][ks] = *reinterpret_cast<const bf16x8 *>(Qp + (size_t)row * a.q_row_stride + ks * 32 + g * 8);
    }

    // staging: thread t moves 16-byte slot t (row = t/8, chunk' = t%8) of each 4 KiB half tile
    const int srow = tid >> 3, sc = tid & 7;
    const int kch = sc ^ ((srow >> 1) & 7);                 // K image: chunk ^ ((row>>1)&7)
    const int vch = sc ^ (((srow >> 1) & 3) << 1);          // V image: chunk-pair ^ ((row>>1)&3)
    auto stage = [&](int t, int buf) {
        unsigned char *kb = lds + buf * 2 * kTileBytes, *vb = kb + kTileBytes;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int kr = t * KT + i * 32 + srow;
            kr = kr < a.Tk ? kr : a.Tk - 1;
            const size_t row = (size_t)kr * a.kv_row_stride;
            glds16(Kp + row + kch * 8, kb + i * 4096 + wave * 1024);
            glds16(Vp + row + vch * 8, vb + i * 4096 + wave * 1024);
        }
    };

    f32x4 o[QT][4];
    float m_run[QT], l_run[QT];
    bool ovf = false;                                             // MODE 2: sticky "a row sum left the safe range"
    const int nt = (a.Tk + KT - 1) / KT;
    const int tq = lq >> 2, tp = lq & 3;

    auto run = [&](auto mode_tag) {
        constexpr int MD = decltype(mode_tag)::value;
        f32x4 l_acc[QT];                                          // MODE 2: row sums, accumulated by the matrix core
        bf16x8 ones;
#pragma unroll
        for (int j = 0; j < 8; ++j) ones[j] = (short)(PVDT == DT_BF16 ? 0x3F80 : 0x3C00);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            m_run[qt] = -INFINITY; l_run[qt] = 0.f;
            l_acc[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        stage(0, 0);
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            // ONE barrier per tile: it publishes tile t (every wave waited for its own share) and, because a wave
            // reaches it only after its last fragment read of tile t-1, it also frees that tile's stage - which is
            // where tile t+1 is staged right behind it, with the whole of tile t's arithmetic to land.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            m3gemm::lds_barrier();
            if (t + 1 < nt && !(M3_ATTN_ABL & 4)) stage(t + 1, buf ^ 1);
            const unsigned char *Ks = lds + ((M3_ATTN_ABL & 4) ? 0 : buf) * 2 * kTileBytes, *Vs = Ks + kTileBytes;

            if constexpr (MD == 2) {
                // range keeper for the row sums of the tiles so far (tested here, a tile late, so that the test does
                // not wait for the matrix core); per query, identical in its 4 lanes
                bool big = false;
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) big |= l_acc[qt][0] > 0x1p60f;
                if (t > 0 && __any(big)) {
                    asm volatile("; rare path" ::: "memory");
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        const bool hit = l_acc[qt][0] > 0x1p60f;
                        ovf |= !(l_acc[qt][0] <= 0x1p100f);          // sticky: o may have overflowed where l has not
                        const float alpha = hit ? 0x1p-64f : 1.0f;
                        m_run[qt] += hit ? 64.0f : 0.0f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) l_acc[qt][r] *= alpha;
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[qt][dt][r] *= alpha;
                    }
                }
            }

            // ---- S^T = K . Q^T : s[qt][kt] holds keys kt*16 + g*4 + r of query lq ------------------
            f32x4 s[QT][4];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const float c0 = (MD >= 1 && t > 0) ? -m_run[qt] : 0.f;          // prescaled: m_run holds m_ref (log2 units)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) s[qt][kt] = f32x4{c0, c0, c0, c0};
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const int r = kt * 16 + lq;
                    const int c = (ks * 4 + g) ^ ((r >> 1) & 7);
                    bf16x8 kf;
                    if ((M3_ATTN_ABL & 1) && t > 0) { kf = qf[0][ks]; asm volatile("" : "+v"(kf)); }
                    else kf = *reinterpret_cast<const bf16x8 *>(Ks + r * 128 + c * 16);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        s[qt][kt] = mfma16<DT>(kf, qf[qt][ks], s[qt][kt]);
                }

            if (t == nt - 1 && (a.Tk & (KT - 1))) {           // key tail: wave-uniform branch, last tile only
                const int kbase = t * KT + g * 4;
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (kbase + kt * 16 + r >= a.Tk) s[qt][kt][r] = -INFINITY;
            }

            // ---- online softmax (row = lane-local query) -------------------------------------------
            bf16x8 pf[QT][2];
            auto tile_max = [&](int qt) {
                float mx = fmaxf(__builtin_fmaxf(s[qt][0][0], s[qt][0][1]), s[qt][0][2]);
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = (kt == 0 ? 3 : 0); r < 4; r += 2)
                        mx = (r + 1 < 4) ? fmaxf(__builtin_fmaxf(mx, s[qt][kt][r]), s[qt][kt][r + 1]) : fmaxf(mx, s[qt][kt][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                return fmaxf(mx, __shfl_xor(mx, 32, 64));
            };
            auto pack_p = [&](int qt) {
                // P fragment for k-step kk: element j<4 -> key (2kk)*16 + g*4 + j, j>=4 -> key (2kk+1)*16 + g*4 + j-4
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    union { unsigned u[4]; bf16x8 v; } pk;
                    pk.u[0] = pack16<PVDT>(s[qt][2 * kk][0], s[qt][2 * kk][1]);
                    pk.u[1] = pack16<PVDT>(s[qt][2 * kk][2], s[qt][2 * kk][3]);
                    pk.u[2] = pack16<PVDT>(s[qt][2 * kk + 1][0], s[qt][2 * kk + 1][1]);
                    pk.u[3] = pack16<PVDT>(s[qt][2 * kk + 1][2], s[qt][2 * kk + 1][3]);
                    pf[qt][kk] = pk.v;
                }
            };
            if constexpr (MD == 2) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    if (t == 0) {                                            // the first tile's true maximum is the reference
                        const float mx = tile_max(qt);
                        m_run[qt] = mx;
#pragma unroll
                        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) s[qt][kt][r] -= mx;
                    }
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[qt][kt][r] = (M3_ATTN_EXP >= 4) ? s[qt][kt][r] : __builtin_amdgcn_exp2f(s[qt][kt][r]);
#if M3_ATTN_EXP == 1
                    for (int kt = 0; kt < 4; ++kt) for (int r = 0; r < 4; ++r) l_run[qt] += s[qt][kt][r];
#endif
#if M3_ATTN_EXP == 5
                    pf[qt][0] = ones; pf[qt][1] = ones;
                    asm volatile("" :: "v"(s[qt][0]), "v"(s[qt][1]), "v"(s[qt][2]), "v"(s[qt][3]));
#else
                    pack_p(qt);
#endif
                }
            } else if constexpr (MD == 1) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const float mx = tile_max(qt);
                    if (t == 0) {                                            // first tile: the true maximum becomes the reference
                        m_run[qt] = mx;
#pragma unroll
                        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) s[qt][kt][r] -= mx;
                    } else if (__any(mx > kDefer)) {                         // some query outgrew its reference: exact update
                        asm volatile("; rare path: keep it a branch (if-converted, its 32 subtracts + selects ran on every tile)" ::: "memory");
                        const float delta = fmaxf(mx, 0.f);
                        const float alpha = __builtin_amdgcn_exp2f(-delta);
                        m_run[qt] += delta;
                        l_run[qt] *= alpha;
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[qt][dt][r] *= alpha;
#pragma unroll
                        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) s[qt][kt][r] -= delta;
                    }
                    f32x2 rs2 = {0.f, 0.f};
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[qt][kt][r] = __builtin_amdgcn_exp2f(s[qt][kt][r]);
                        rs2 += f32x2{s[qt][kt][0], s[qt][kt][1]};
                        rs2 += f32x2{s[qt][kt][2], s[qt][kt][3]};
                    }
                    l_run[qt] += rs2.x + rs2.y;
                    pack_p(qt);
                }
            } else {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    float mx = s[qt][0][0];
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[qt][kt][r]);
                    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                    const float m_new = fmaxf(m_run[qt], mx);
                    const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * a.scale_log2e);
                    const float mb = m_new * a.scale_log2e;
                    m_run[qt] = m_new;
                    float rs = 0.f;
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = __builtin_amdgcn_exp2f(s[qt][kt][r] * a.scale_log2e - mb);
                            s[qt][kt][r] = p;
                            rs += p;
                        }
                    l_run[qt] = l_run[qt] * alpha + rs;
                    if (!__all(alpha == 1.0f)) {       // exact skip: no row of this wave raised its running max
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[qt][dt][r] *= alpha;
                    }
                    pack_p(qt);
                }
            }

            // ---- O^T += V^T . P^T : A fragment = V^T[d = dt*16 + lq][same key permutation] ------------
            // transposed read: lane (4q+p) of a 16-lane group addresses row (key0 + q), cols d0 + 4p..4p+3
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    union { bf16x4 h[2]; bf16x8 v; } vf;
                    if ((M3_ATTN_ABL & 2) && t > 0) { vf.v = qf[0][kk]; asm volatile("" : "+v"(vf.v)); }
                    else
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int row = (2 * kk + half) * 16 + g * 4 + tq;
                        const int ch = (dt * 2 + (tp >> 1)) ^ (((row >> 1) & 3) << 1);
                        const unsigned char *p = Vs + row * 128 + ch * 16 + (tp & 1) * 8;
                        vf.h[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) bf16x4 *)p);
                    }
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        o[qt][dt] = mfma16<PVDT>(vf.v, pf[qt][kk], o[qt][dt]);
                }
                if constexpr (MD == 2) {
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) l_acc[qt] = mfma16<PVDT>(ones, pf[qt][kk], l_acc[qt]);
                }
            }
        }
        m3gemm::lds_barrier();                      // the stages are reused right after the loop (flag word / recomputation)
        if constexpr (MD == 2 && M3_ATTN_EXP != 1) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) l_run[qt] = l_acc[qt][0];       // complete row sum, no lane reduction left
        } else {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float l = l_run[qt];
                l += __shfl_xor(l, 16, 64);
                l_run[qt] = l + __shfl_xor(l, 32, 64);
            }
        }
    };

    if constexpr (MODE == 2) {
        run(std::integral_constant<int, 2>{});
        bool bad = ovf;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            bad |= !(l_run[qt] > 0.f && l_run[qt] <= 0x1p100f);
            float osum = 0.f;                                     // inf / NaN anywhere in the row's outputs survives the sum
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) osum += (o[qt][dt][0] + o[qt][dt][1]) + (o[qt][dt][2] + o[qt][dt][3]);
            bad |= !(fabsf(osum) < INFINITY);
        }
#if M3_ATTN_EXP == 3
        bad = true;
#endif
        // workgroup-wide OR through a word of the (now idle) tile buffers; every wave passed the loop's last barrier
        int *flag = reinterpret_cast<int *>(lds);
        if (tid == 0) *flag = 0;
        m3gemm::lds_barrier();
        if (__any(bad) && lane == 0) atomicOr(flag, 1);
        m3gemm::lds_barrier();
        const int redo = *reinterpret_cast<volatile int *>(flag);
        m3gemm::lds_barrier();                                  // everyone has read the flag before tile 0 is staged over it
        if (redo) run(std::integral_constant<int, 1>{});        // exp2 overflowed somewhere: exact recomputation
    } else {
        run(std::integral_constant<int, MODE>{});
    }

    // ---- finalize: O[q][dt*16 + g*4 + r] = o / l ---------------------------------------------------
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float inv = 1.0f / l_run[qt];
        const int row = qblk * QR + wave * (16 * QT) + qt * 16 + lq;
        if (row >= a.Tq) continue;
        bf16_t *op = a.O + (size_t)b * a.o_batch_stride + (size_t)row * a.o_row_stride + head * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            uint2 w;
            w.x = pack16<DT>(o[qt][dt][0] * inv, o[qt][dt][1] * inv);
            w.y = pack16<DT>(o[qt][dt][2] * inv, o[qt][dt][3] * inv);
            *reinterpret_cast<uint2 *>(op + dt * 16 + g * 4) = w;
        }
    }
}

// ---------------------------------------------------------------- RoPE-2D (CroCo "RoPE100")
// In place on a [tokens, row_stride] bf16 buffer: for every head, the first 32 dims rotate with
// the token's y position, the last 32 with x; within a 32-block element i pairs with i+16:
//   out[i]    = x[i] cos(p f_i) - x[i+16] sin(p f_i)
//   out[i+16] = x[i+16] cos(p f_i) + x[i] sin(p f_i),   f_i = base^(-i/16), i = 0..15
// cs: fp32 table [max_pos][16][2] = (cos, sin).  One thread handles one (token, head, 32-block).
template <int DT>
__global__ void __launch_bounds__(kThreads)
k_rope2d(bf16_t *__restrict__ X, const int *__restrict__ pos_yx, const float *__restrict__ cs, int row_stride,
         int tokens, int heads, int tokens_per_image) {
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    const int total = tokens * heads * 2;
    if (idx >= total) return;
    const int blk = idx & 1, head = (idx >> 1) % heads, tok = (idx >> 1) / heads;
    const int pos = pos_yx[(tok % tokens_per_image) * 2 + blk];
    bf16_t *p = X + (size_t)tok * row_stride + head * HD + blk * 32;
    union { uint4 q[4]; bf16_t h[32]; } v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v.q[i] = reinterpret_cast<const uint4 *>(p)[i];
    const float *t = cs + (size_t)pos * 32;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float c = t[2 * i], sn = t[2 * i + 1];
        const float x1 = m3gemm::lo16<DT>(v.h[i]), x2 = m3gemm::lo16<DT>(v.h[i + 16]);
        const unsigned pk = pack16<DT>(x1 * c - x2 * sn, x2 * c + x1 * sn);
        v.h[i] = (bf16_t)(pk & 0xffff);
        v.h[i + 16] = (bf16_t)(pk >> 16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) reinterpret_cast<uint4 *>(p)[i] = v.q[i];
}

}  // namespace

extern "C" {

static int attention_launch(const void *Q, const void *K, const void *V, void *O, int q_row_stride, int kv_row_stride,
                            int o_row_stride, int64_t q_batch_stride, int64_t kv_batch_stride, int64_t o_batch_stride,
                            int nbatch, int heads, int Tq, int Tk, int kv_batch_shift, float scale, int dtype, bool pre,
                            void *stream) {
    M3_REQUIRE(Q && K && V && O && nbatch > 0 && heads > 0 && Tq > 0 && Tk > 0);
    // dtype 2 = M3_DT_F16_PVBF16 (prescaled entry only): q, k, O fp16; V holds bf16, P is bf16
    M3_REQUIRE((dtype == DT_BF16 || dtype == DT_F16 || (dtype == 2 && pre)) && ((int64_t)Tq / 64 + 1) * heads * nbatch < (1ll << 31));
    M3_REQUIRE(q_row_stride % 8 == 0 && kv_row_stride % 8 == 0 && o_row_stride % 4 == 0);
    M3_REQUIRE(kv_batch_shift >= 0);
    AttnArgs a;
    a.Q = (const bf16_t *)Q; a.K = (const bf16_t *)K; a.V = (const bf16_t *)V; a.O = (bf16_t *)O;
    a.q_row_stride = q_row_stride; a.kv_row_stride = kv_row_stride; a.o_row_stride = o_row_stride;
    a.q_batch_stride = q_batch_stride; a.kv_batch_stride = kv_batch_stride; a.o_batch_stride = o_batch_stride;
    a.Tq = Tq; a.Tk = Tk; a.heads = heads; a.nbatch = nbatch; a.kv_batch_shift = kv_batch_shift;
    a.scale_log2e = scale * 1.4426950408889634f;
    const int64_t wg128 = (int64_t)m3_cdiv(Tq, QROWS) * heads * nbatch;
    const int64_t wg64 = (int64_t)m3_cdiv(Tq, 64) * heads * nbatch;
    hipStream_t st = (hipStream_t)stream;
    // M3_ATTN_SAFE=1: prescaled bf16 launches take the max-tracking loop (MODE 1) instead of the fast one (experiments)
    static const bool safe_bf16 = [] { const char *e = getenv("M3_ATTN_SAFE"); return e && atoi(e) != 0; }();
#define M3_ATTN(QTV, GRID)                                                                                  \
    do {                                                                                                    \
        if (dtype == 2) hipLaunchKernelGGL((k_attn<QTV, DT_F16, 2, DT_BF16>), dim3((unsigned)(GRID)), dim3(kThreads), 0, st, a); \
        else if (dtype == DT_F16) { if (pre) hipLaunchKernelGGL((k_attn<QTV, DT_F16, 1>), dim3((unsigned)(GRID)), dim3(kThreads), 0, st, a); \
                               else hipLaunchKernelGGL((k_attn<QTV, DT_F16, 0>), dim3((unsigned)(GRID)), dim3(kThreads), 0, st, a); } \
        else { if (pre) { if (safe_bf16) hipLaunchKernelGGL((k_attn<QTV, DT_BF16, 1>), dim3((unsigned)(GRID)), dim3(kThreads), 0, st, a); \
                          else hipLaunchKernelGGL((k_attn<QTV, DT_BF16, 2>), dim3((unsigned)(GRID)), dim3(kThreads), 0, st, a); }          \
               else hipLaunchKernelGGL((k_attn<QTV, DT_BF16, 0>), dim3((unsigned)(GRID)), dim3(kThreads), 0, st, a); }                    \
    } while (0)
    if (wg128 >= 512) M3_ATTN(2, wg128); else M3_ATTN(1, wg64);
#undef M3_ATTN
    M3_CHECK_LAUNCH("m3_attention");
    return M3_OK;
}

int m3_attention_dt(const void *Q, const void *K, const void *V, void *O, int q_row_stride, int kv_row_stride,
                    int o_row_stride, int64_t q_batch_stride, int64_t kv_batch_stride, int64_t o_batch_stride,
                    int nbatch, int heads, int Tq, int Tk, int kv_batch_shift, float scale, int dtype, void *stream) {
    return attention_launch(Q, K, V, O, q_row_stride, kv_row_stride, o_row_stride, q_batch_stride, kv_batch_stride,
                            o_batch_stride, nbatch, heads, Tq, Tk, kv_batch_shift, scale, dtype, false, stream);
}
// q carries softmax scale * log2(e) already (m3_gemm_rope_dt's q_scale): O = softmax2(Q K^T) V with p = 2^(s - m)
int m3_attention_prescaled_dt(const void *Q, const void *K, const void *V, void *O, int q_row_stride, int kv_row_stride,
                              int o_row_stride, int64_t q_batch_stride, int64_t kv_batch_stride, int64_t o_batch_stride,
                              int nbatch, int heads, int Tq, int Tk, int kv_batch_shift, int dtype, void *stream) {
    return attention_launch(Q, K, V, O, q_row_stride, kv_row_stride, o_row_stride, q_batch_stride, kv_batch_stride,
                            o_batch_stride, nbatch, heads, Tq, Tk, kv_batch_shift, 1.0f, dtype, true, stream);
}
int m3_attention_bf16(const void *Q, const void *K, const void *V, void *O, int q_row_stride, int kv_row_stride,
                      int o_row_stride, int64_t q_batch_stride, int64_t kv_batch_stride, int64_t o_batch_stride,
                      int nbatch, int heads, int Tq, int Tk, int kv_batch_shift, float scale, void *stream) {
    return m3_attention_dt(Q, K, V, O, q_row_stride, kv_row_stride, o_row_stride, q_batch_stride, kv_batch_stride,
                           o_batch_stride, nbatch, heads, Tq, Tk, kv_batch_shift, scale, DT_BF16, stream);
}

int m3_rope2d_dt(void *X, const int32_t *pos_yx, const float *cos_sin, int row_stride, int tokens, int heads,
                 int tokens_per_image, int dtype, void *stream) {
    M3_REQUIRE(X && pos_yx && cos_sin && tokens > 0 && heads > 0 && tokens_per_image > 0 && row_stride % 8 == 0);
    M3_REQUIRE(dtype == DT_BF16 || dtype == DT_F16);
    const int64_t total = (int64_t)tokens * heads * 2;
    if (dtype == DT_F16)
        hipLaunchKernelGGL(k_rope2d<DT_F16>, dim3(m3_cdiv(total, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                           (bf16_t *)X, pos_yx, cos_sin, row_stride, tokens, heads, tokens_per_image);
    else
        hipLaunchKernelGGL(k_rope2d<DT_BF16>, dim3(m3_cdiv(total, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                           (bf16_t *)X, pos_yx, cos_sin, row_stride, tokens, heads, tokens_per_image);
    M3_CHECK_LAUNCH("m3_rope2d");
    return M3_OK;
}
int m3_rope2d_bf16(void *X, const int32_t *pos_yx, const float *cos_sin, int row_stride, int tokens, int heads,
                   int tokens_per_image, void *stream) {
    return m3_rope2d_dt(X, pos_yx, cos_sin, row_stride, tokens, heads, tokens_per_image, DT_BF16, stream);
}

}  // extern "C"
