// ARCHIVE (round 4, negative result) - not built into libm3slam_hip.so.
//
// 8-wave "ping-pong" form of the prescaled attention kernel: the body below was attention.hip: k_attn_pp, dispatched for
// Tq % 512 == 0 launches with >= 256 workgroups.  It is CORRECT (it passed the prescaled / overflow-recomputation tests and
// the bitwise "a pair alone == the pair in a batch of 8 through the graph" test against k_attn) and NOT faster:
//
//   16 images x 16 heads x 1024^2:  k_attn<2, bf16, 2>  91.9 us (748 TFLOP/s)   k_attn_pp  91.4 - 94.0 us
//   16 images x 12 heads x 1024^2:                      71.2 us (724 TFLOP/s)              83.2 - 85.2 us (384 workgroups = 1.5 rounds)
//   step (8 pairs): 26.61 ms with k_attn, 26.93 ms with k_attn_pp (same box)
//   s_setprio 1 on the VALU / read phases: 95.3 us; on the MFMA phases: 96.1 us (no effect either way)
//
// Why it was tried: with the WHOLE softmax removed k_attn still runs at 939 TFLOP/s (profiles/r04_attention_ablation.md), so
// the phase discipline that took the GEMM from 760 to 1 000+ TFLOP/s looked like the lever.  Why it does not pay here: per
// wave and key tile (4 query tiles) the loop issues 72 MFMAs (each holds the SIMD's vector issue for 8 of its 16 cycles =
// 576), 64 v_exp_f32 (8 cycles = 512), 32 v_cvt_pk (~150) and ~140 other VALU instructions (address arithmetic the
// 256-register budget forces the compiler to rematerialise, accumulator initialisers, the range keeper) = ~1 800 issue
// cycles against 1 152 matrix cycles: at head dimension 64 the SIMD's ISSUE port, shared by the two waves, is the limiter,
// and both very different schedules land on the same ~750 TFLOP/s.  Kept for the record with the ISA phase counts
// (per barrier interval: 40 MFMA | 8 ds_read + 2 LDS-DMA + 72 VALU | 32 MFMA + 16 VALU | 64 exp + 32 cvt + 16 ds_read_tr + 70 VALU).
//
// To build it again: paste into attention.hip in front of the RoPE section and dispatch from attention_launch
// (git history: commit "8-wave ping-pong attention kernel (k_attn_pp)").

// ---------------------------------------------------------------- 8-wave ping-pong form (round 4)
// Timing-only builds of k_attn (profiles/r04_attention_ablation.md) showed where the 4-wave kernel's time goes: with the
// WHOLE softmax removed (no exp2, no packing, constant P) it still runs at 939 TFLOP/s = 0.38 of the MFMA peak - the
// skeleton itself (every wave reads its fragments from LDS and then waits for them in front of its own MFMAs, four
// free-running waves per SIMD, one barrier per tile) idles the matrix pipe more than half of the time; exp2 and packing
// add only 20 %.  This kernel gives attention the phase discipline of gemm256.hip: a 512-thread workgroup = 8 waves, two
// per SIMD, each owning FOUR 16-row query tiles (64 rows; 512 rows per workgroup, so K / V fragments are read once per 4
// query tiles instead of 2), and per key tile every wave runs
//        VA: stage tile t+1 (LDS-DMA), range keeper, READ K fragments  |  MA: 32 S^T MFMAs
//        VB: exp2 + pack (+ first-tile maximum, key-tail mask), READ V^T fragments  |  MB: 32 O^T MFMAs + 8 row-sum MFMAs
// with waves 0-3 ("ping") and 4-7 ("pong") ONE phase apart and a workgroup barrier between phases: while one wave of a
// SIMD issues MFMAs its partner issues LDS reads / transcendentals (MI355X_MICROARCH.md "Two waves per SIMD": matrix
// beside VALU / memory is the complementary pairing; no s_setprio - a prioritised MFMA wave starves its partner's
// v_exp).  Same arithmetic per query row as k_attn<.., MODE 2> - the same MFMAs on the same operands in the same order,
// the same exp2 / packing, the same range keeper - hence the same bits (test: a pair alone vs in a batch of 8), and the
// same overflow rule: a workgroup whose row sums left the safe range recomputes its block with the max-tracking loop.
constexpr int kPPThreads = 512;
constexpr int kPPQT = 4;                    // 16-row query tiles per wave
constexpr int kPPRows = 8 * kPPQT * 16;     // 512 query rows per workgroup

template <int DT, int PVDT>
__global__ void __launch_bounds__(kPPThreads, 2)
k_attn_pp(const AttnArgs a) {
    constexpr int QT = kPPQT;
    static_assert(PVDT == DT_BF16, "the deferred-maximum loop needs the fp32 exponent range of bf16 for P");
    __shared__ __attribute__((aligned(16))) unsigned char lds[kLds];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 15, g = lane >> 4, group = wave >> 2;
    const int nq = (a.Tq + kPPRows - 1) / kPPRows, nwg = gridDim.x;
    const int per = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
    const int id = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (blockIdx.x >> 3);
    const int qblk = id % nq, head = (id / nq) % a.heads, b = id / (nq * a.heads);
    const int kvb = (b + a.kv_batch_shift) % a.nbatch;
    const bf16_t *Qp = a.Q + (size_t)b * a.q_batch_stride + head * HD;
    const bf16_t *Kp = a.K + (size_t)kvb * a.kv_batch_stride + head * HD;
    const bf16_t *Vp = a.V + (size_t)kvb * a.kv_batch_stride + head * HD;
    const int row0 = qblk * kPPRows + wave * (16 * QT);

    bf16x8 qf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int row = row0 + qt * 16 + lq;
        row = row < a.Tq ? row : a.Tq - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[qt][ks] = *reinterpret_cast<const bf16x8 *>(Qp + (size_t)row * a.q_row_stride + ks * 32 + g * 8);
    }
    // staging: 512 threads move one 16-byte slot of the K tile and one of the V tile each (row = tid / 8, chunk' = tid % 8)
    const int srow = tid >> 3, sc = tid & 7;
    const int kch = sc ^ ((srow >> 1) & 7), vch = sc ^ (((srow >> 1) & 3) << 1);
    auto stage = [&](int t, int buf) {
        unsigned char *kb = lds + buf * 2 * kTileBytes, *vb = kb + kTileBytes;
        int kr = t * KT + srow;
        kr = kr < a.Tk ? kr : a.Tk - 1;
        const size_t row = (size_t)kr * a.kv_row_stride;
        glds16(Kp + row + kch * 8, kb + wave * 1024);
        glds16(Vp + row + vch * 8, vb + wave * 1024);
    };
    const int nt = (a.Tk + KT - 1) / KT;
    const int tq = lq >> 2, tp = lq & 3;

    f32x4 o[QT][4], s[QT][4], l_acc[QT];
    float m_run[QT], l_run[QT];
    bf16x8 pf[QT][2], kf[2][4], vf[2][4], ones;
    bool ovf = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (short)0x3F80;

    auto reset = [&]() {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            m_run[qt] = -INFINITY; l_run[qt] = 0.f;
            l_acc[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto read_k = [&](int buf) {
        const unsigned char *Ks = lds + buf * 2 * kTileBytes;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const int r = kt * 16 + lq;
                const int c = (ks * 4 + g) ^ ((r >> 1) & 7);
                kf[ks][kt] = *reinterpret_cast<const bf16x8 *>(Ks + r * 128 + c * 16);
            }
    };
    auto read_v = [&](int buf) {
        const unsigned char *Vs = lds + buf * 2 * kTileBytes + kTileBytes;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                union { bf16x4 h[2]; bf16x8 v; } u;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int row = (2 * kk + half) * 16 + g * 4 + tq;
                    const int ch = (dt * 2 + (tp >> 1)) ^ (((row >> 1) & 3) << 1);
                    const unsigned char *p = Vs + row * 128 + ch * 16 + (tp & 1) * 8;
                    u.h[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4 *)p);
                }
                vf[kk][dt] = u.v;
            }
    };
    auto qk = [&](int t) {                                     // S^T = K . Q^T (+ -m_ref as the accumulator initialiser)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const float c0 = t > 0 ? -m_run[qt] : 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) s[qt][kt] = f32x4{c0, c0, c0, c0};
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) s[qt][kt] = mfma16<DT>(kf[ks][kt], qf[qt][ks], s[qt][kt]);
    };
    auto pv = [&](bool rowsum) {                               // O^T += V^T . P^T, row sums against an all-ones A fragment
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) o[qt][dt] = mfma16<PVDT>(vf[kk][dt], pf[qt][kk], o[qt][dt]);
            if (rowsum) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) l_acc[qt] = mfma16<PVDT>(ones, pf[qt][kk], l_acc[qt]);
            }
        }
    };
    auto mask_tail = [&](int t) {
        if (t == nt - 1 && (a.Tk & (KT - 1))) {                // key tail: wave-uniform branch, last tile only
            const int kbase = t * KT + g * 4;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kbase + kt * 16 + r >= a.Tk) s[qt][kt][r] = -INFINITY;
        }
    };
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
    auto softmax_fast = [&](int t) {                           // MODE 2: reference = the first tile's true maximum
        mask_tail(t);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            if (t == 0) {
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
                for (int r = 0; r < 4; ++r) s[qt][kt][r] = __builtin_amdgcn_exp2f(s[qt][kt][r]);
            pack_p(qt);
        }
    };
    auto range_keep = [&](int t) {                             // as k_attn: tested a tile late, per query, sticky overflow flag
        bool big = false;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) big |= l_acc[qt][0] > 0x1p60f;
        if (t > 0 && __any(big)) {
            asm volatile("; rare path" ::: "memory");
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const bool hit = l_acc[qt][0] > 0x1p60f;
                ovf |= !(l_acc[qt][0] <= 0x1p100f);
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
    };
    auto phase_end = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto phase_end_lds = [&]() {                               // this wave's fragment reads have returned
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto phase_end_all = [&]() {                               // ... and its share of the next tile has landed
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- fast loop -------------------------------------------------------------------------------------------------
#ifndef M3_PP_PRIO
#define M3_PP_PRIO 0        // experiment: 1 = VALU / read phases at s_setprio 1, 2 = MFMA phases at s_setprio 1
#endif
#define M3_V_BEGIN() do { if (M3_PP_PRIO == 1) __builtin_amdgcn_s_setprio(1); else if (M3_PP_PRIO == 2) __builtin_amdgcn_s_setprio(0); } while (0)
#define M3_M_BEGIN() do { if (M3_PP_PRIO == 1) __builtin_amdgcn_s_setprio(0); else if (M3_PP_PRIO == 2) __builtin_amdgcn_s_setprio(1); } while (0)
    reset();
    stage(0, 0);
    phase_end_all();
    if (group == 0) {
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            M3_V_BEGIN();
            if (t + 1 < nt) stage(t + 1, buf ^ 1);             // VA: the stage of tile t-1 is free (everyone read it two phases ago)
            range_keep(t);
            read_k(buf);
            phase_end_lds();
            M3_M_BEGIN();
            qk(t);                                             // MA
            phase_end();
            M3_V_BEGIN();
            softmax_fast(t);                                   // VB
            read_v(buf);
            phase_end_lds();
            M3_M_BEGIN();
            pv(true);                                          // MB
            phase_end_all();
        }
        phase_end();                                           // the pong group's drain phase
    } else {
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            M3_M_BEGIN();
            if (t > 0) pv(true);                               // MB of tile t-1
            phase_end();
            M3_V_BEGIN();
            if (t + 1 < nt) stage(t + 1, buf ^ 1);             // VA
            range_keep(t);
            read_k(buf);
            phase_end_lds();
            M3_M_BEGIN();
            qk(t);                                             // MA
            phase_end();
            M3_V_BEGIN();
            softmax_fast(t);                                   // VB
            read_v(buf);
            phase_end_all();
        }
        M3_M_BEGIN();
        pv(true);                                              // drain: MB of the last tile
        phase_end();
    }
    __builtin_amdgcn_s_setprio(0);
#undef M3_V_BEGIN
#undef M3_M_BEGIN
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) l_run[qt] = l_acc[qt][0];

    // ---- did exp2 leave the safe range anywhere in this workgroup? ----------------------------------------------------
    bool bad = ovf;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        bad |= !(l_run[qt] > 0.f && l_run[qt] <= 0x1p100f);
        float osum = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) osum += (o[qt][dt][0] + o[qt][dt][1]) + (o[qt][dt][2] + o[qt][dt][3]);
        bad |= !(fabsf(osum) < INFINITY);
    }
    int *flag = reinterpret_cast<int *>(lds);
    if (tid == 0) *flag = 0;
    m3gemm::lds_barrier();
    if (__any(bad) && lane == 0) atomicOr(flag, 1);
    m3gemm::lds_barrier();
    const int redo = *reinterpret_cast<volatile int *>(flag);
    m3gemm::lds_barrier();
    if (redo) {
        // ---- exact recomputation (rare): the max-tracking loop of k_attn<.., MODE 1>, every wave in step, one barrier per tile
        reset();
        stage(0, 0);
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            m3gemm::lds_barrier();
            if (t + 1 < nt) stage(t + 1, buf ^ 1);
            read_k(buf);
            qk(t);
            mask_tail(t);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const float mx = tile_max(qt);
                if (t == 0) {
                    m_run[qt] = mx;
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[qt][kt][r] -= mx;
                } else if (__any(mx > kDefer)) {
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
            read_v(buf);
            pv(false);
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float l = l_run[qt];
            l += __shfl_xor(l, 16, 64);
            l_run[qt] = l + __shfl_xor(l, 32, 64);
        }
    }

    // ---- finalize: O[q][dt*16 + g*4 + r] = o / l ----------------------------------------------------------------------
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float inv = 1.0f / l_run[qt];
        const int row = row0 + qt * 16 + lq;
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

