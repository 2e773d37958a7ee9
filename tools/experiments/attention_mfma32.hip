// ARCHIVED EXPERIMENT (round 4) - not compiled into libm3slam_hip.so.
//
// k_attn32: the prescaled / deferred-maximum attention loop of csrc/attention.hip (k_attn MODE 2) on
// v_mfma_f32_32x32x16 tiles: one wave = 32 queries, a 64-key tile = 2 key blocks x 4 k-steps for S^T and 2 d blocks x
// 4 k-steps for O^T.  The idea: an MFMA holds the SIMD's vector issue for the same few cycles whatever its size, so the
// 32 x 32 form halves the MFMA issue slots per flop and leaves more of them to v_exp_f32 / v_cvt_pk.
//
// It is CORRECT (tests/test_gpu_model.py -k attention: 22 passed with the dispatch below switched on; operand layouts
// probed on the device: A lane l = row l & 31, k = 8 (l >> 5) + j; B lane l = column l & 31, same k; D lane l register
// r = row 8 (r / 4) + 4 (l >> 5) + (r % 4), column l & 31) and SLOWER.  Same box, one gpurun call, 16 images x {16, 12}
// heads x 1024 x 1024, bf16, random data (tools/bench_attn.py):
//
//   k_attn<2, bf16, MODE 2> (shipped, 128 VGPRs, 4 waves / SIMD)                       87.5 - 93.0 us   70.1 - 72.9 us
//   k_attn32, row sums by a ones-MFMA (172 VGPRs, 2 waves / SIMD)                      113.0 - 114.3    82.7 - 84.3
//   k_attn32, row sums on the VALU, amdgpu_waves_per_eu(3,3) (168 VGPRs, 2 - 9 spilled) 97.2 - 98.7     76.9
//
// Why it cannot win here: (1) it needs 11 % more matrix cycles (row sums cost a full 32-row MFMA per k-step, or 32 VALU
// adds), (2) its accumulators (2 x 16 for S^T, 2 x 16 for O^T, + fragments) cost occupancy, and (3) - the part that
// bounds BOTH kernels - the chip lowers its clock under MFMA load on random operands: /opt/skills/guides/
// MI355X_MICROARCH.md "DVFS give-back" measures ~1 250 TFLOP/s for a BARE bf16 16x16x32 MFMA loop on random data (1 480 on
// zeros, i.e. at 2.3 GHz) and the 32x32x16 shape 1.12 - 1.15 x BELOW that at equal cycles.  The shipped kernel's bare
// MFMA + barrier skeleton (profiles/r04_attention_ablation.md: 1 115 TFLOP/s) is therefore already at ~0.9 of what the
// matrix pipe sustains on this data, and the full kernel (750 - 860) at 0.60 - 0.69 of it; 1.0 PFLOP/s with a softmax in
// the loop would need 0.8.
//
// Kept for the record; to revive, paste the kernel before the RoPE section of csrc/attention.hip and the dispatch in front
// of the M3_ATTN launch of m3_attention_dt.

// ---------------------------------------------------------------- 32 x 32 x 16 MFMA form (round 4 experiment)
// Same algorithm as k_attn<.., MODE 2> (prescaled q, reference = the first tile's maximum as the accumulator
// initialiser, row sums on the matrix core, range keeper, workgroup-wide exact recomputation on overflow) on
// v_mfma_f32_32x32x16: a wave owns 32 queries (one MFMA N block), a 64-key tile is 2 key blocks x 4 k-steps for S^T and
// 2 d blocks x 4 k-steps for O^T - 20 MFMAs of 32 cycles per tile instead of 36 of 16, i.e. HALF the MFMA issue slots per
// flop (an MFMA holds the SIMD's vector issue for 8 cycles whatever its size; profiles/r04_attention_ablation.md).
// Register layouts probed on the device (scratch probe, round 4): A lane l = row l & 31, k = 8 (l >> 5) + j; B lane l =
// column l & 31, same k; D lane l, register r = row 8 (r / 4) + 4 (l >> 5) + (r % 4), column l & 31.
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int DT> __device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (DT == DT_BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(m3gemm::f16x8, a), __builtin_bit_cast(m3gemm::f16x8, b), c, 0, 0, 0);
}

template <int NW /* waves = 32-query blocks per workgroup */, int DT, int PVDT>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_attn32(const AttnArgs a) {
    static_assert(PVDT == DT_BF16, "deferred-maximum loop: bf16 P");
    constexpr int QR = 32 * NW, NTH = 64 * NW;
    __shared__ __attribute__((aligned(16))) unsigned char lds[kLds];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int nq = (a.Tq + QR - 1) / QR, nwg = gridDim.x;
    const int per = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
    const int id = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (blockIdx.x >> 3);
    const int qblk = id % nq, head = (id / nq) % a.heads, b = id / (nq * a.heads);
    const int kvb = (b + a.kv_batch_shift) % a.nbatch;
    const bf16_t *Qp = a.Q + (size_t)b * a.q_batch_stride + head * HD;
    const bf16_t *Kp = a.K + (size_t)kvb * a.kv_batch_stride + head * HD;
    const bf16_t *Vp = a.V + (size_t)kvb * a.kv_batch_stride + head * HD;
    const int qrow = qblk * QR + wave * 32 + n;

    bf16x8 qf[4];                                              // B operand: query n, d = 16 ks + 8 h + j
    {
        const int row = qrow < a.Tq ? qrow : a.Tq - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8 *>(Qp + (size_t)row * a.q_row_stride + ks * 16 + h * 8);
    }
    auto stage = [&](int t, int buf) {                         // K / V tile: 512 sixteen-byte slots each
        unsigned char *kb = lds + buf * 2 * kTileBytes, *vb = kb + kTileBytes;
#pragma unroll
        for (int i = 0; i < 512 / NTH; ++i) {
            const int slot = i * NTH + tid, srow = slot >> 3, sc = slot & 7;
            const int kch = sc ^ ((srow >> 1) & 7), vch = sc ^ (((srow >> 1) & 3) << 1);
            int kr = t * KT + srow;
            kr = kr < a.Tk ? kr : a.Tk - 1;
            const size_t row = (size_t)kr * a.kv_row_stride;
            glds16(Kp + row + kch * 8, kb + (i * NW + wave) * 1024);
            glds16(Vp + row + vch * 8, vb + (i * NW + wave) * 1024);
        }
    };
    const int nt = (a.Tk + KT - 1) / KT;
    f32x16 o[2], s[2];
    float m_run, l_run, l_acc0;
    bool ovf = false;
    bf16x8 pf[2][2];

    auto reset = [&]() {
        m_run = -INFINITY; l_run = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
        l_acc0 = 0.f;
    };
    auto qk = [&](const unsigned char *Ks, int t) {
        const float c0 = t > 0 ? -m_run : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[0][r] = c0; s[1][r] = c0; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int row = kb * 32 + n;
                const int c = (2 * ks + h) ^ ((row >> 1) & 7);
                const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(Ks + row * 128 + c * 16);
                s[kb] = mfma32<DT>(kf, qf[ks], s[kb]);
            }
    };
    auto mask_tail = [&](int t) {
        if (t == nt - 1 && (a.Tk & (KT - 1))) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * KT + kb * 32 + 8 * (r / 4) + 4 * h + (r % 4) >= a.Tk) s[kb][r] = -INFINITY;
        }
    };
    auto tile_max = [&]() {
        float mx = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        return fmaxf(mx, __shfl_xor(mx, 32, 64));
    };
    auto pack_p = [&]() {                                      // k-step s2 of key block kb = registers 8 s2 .. 8 s2 + 7
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                union { unsigned u[4]; bf16x8 v; } pk;
#pragma unroll
                for (int q = 0; q < 4; ++q) pk.u[q] = pack16<PVDT>(s[kb][8 * s2 + 2 * q], s[kb][8 * s2 + 2 * q + 1]);
                pf[kb][s2] = pk.v;
            }
    };
    auto pv = [&](const unsigned char *Vs, bool rowsum) {
        const int lq = lane & 15, g16 = (lane >> 4) & 1, tq = lq >> 2, tp = lq & 3;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    union { bf16x4 hh[2]; bf16x8 v; } vf;      // A operand: d = 32 db + (lane & 31), K index 8 h + j <-> the keys of pf[kb][s2]
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int row = kb * 32 + 16 * s2 + 8 * half + 4 * h + tq;
                        const int ch = ((db * 2 + g16) * 2 + (tp >> 1)) ^ (((row >> 1) & 3) << 1);
                        const unsigned char *p = Vs + row * 128 + ch * 16 + (tp & 1) * 8;
                        vf.hh[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4 *)p);
                    }
                    o[db] = mfma32<PVDT>(vf.v, pf[kb][s2], o[db]);
                }
            }
    };

    // ---- fast loop (MODE 2) ----
    reset();
    stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        m3gemm::lds_barrier();
        if (t + 1 < nt) stage(t + 1, buf ^ 1);
        const unsigned char *Ks = lds + buf * 2 * kTileBytes, *Vs = Ks + kTileBytes;
        if (t > 0 && __any(l_acc0 > 0x1p60f)) {                // range keeper, a tile late (l_acc0: this lane's half of the row sum)
            float lw = l_acc0 + __shfl_xor(l_acc0, 32, 64);
            const bool hit = lw > 0x1p60f;
            ovf |= !(lw <= 0x1p100f);
            const float alpha = hit ? 0x1p-64f : 1.0f;
            m_run += hit ? 64.0f : 0.0f;
            l_acc0 *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        }
        qk(Ks, t);
        mask_tail(t);
        if (t == 0) {
            const float mx = tile_max();
            m_run = mx;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[0][r] -= mx; s[1][r] -= mx; }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[0][r] = __builtin_amdgcn_exp2f(s[0][r]); s[1][r] = __builtin_amdgcn_exp2f(s[1][r]); }
        {
            float r0 = 0.f, r1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { r0 += s[0][r]; r1 += s[1][r]; }
            l_acc0 += r0 + r1;
        }
        pack_p();
        pv(Vs, true);
    }
    m3gemm::lds_barrier();
    l_run = l_acc0 + __shfl_xor(l_acc0, 32, 64);
    bool bad = ovf || !(l_run > 0.f && l_run <= 0x1p100f);
    {
        float osum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) osum += o[0][r] + o[1][r];
        bad |= !(fabsf(osum) < INFINITY);
    }
    int *flag = reinterpret_cast<int *>(lds);
    if (tid == 0) *flag = 0;
    m3gemm::lds_barrier();
    if (__any(bad) && lane == 0) atomicOr(flag, 1);
    m3gemm::lds_barrier();
    const int redo = *reinterpret_cast<volatile int *>(flag);
    m3gemm::lds_barrier();
    if (redo) {                                                // exact recomputation: track the maximum (k_attn MODE 1's rule)
        reset();
        stage(0, 0);
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            m3gemm::lds_barrier();
            if (t + 1 < nt) stage(t + 1, buf ^ 1);
            const unsigned char *Ks = lds + buf * 2 * kTileBytes, *Vs = Ks + kTileBytes;
            qk(Ks, t);
            mask_tail(t);
            const float mx = tile_max();
            if (t == 0) {
                m_run = mx;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[0][r] -= mx; s[1][r] -= mx; }
            } else if (__any(mx > kDefer)) {
                const float delta = fmaxf(mx, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m_run += delta;
                l_run *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; s[0][r] -= delta; s[1][r] -= delta; }
            }
            float rs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[0][r] = __builtin_amdgcn_exp2f(s[0][r]); s[1][r] = __builtin_amdgcn_exp2f(s[1][r]);
                rs += s[0][r] + s[1][r];
            }
            l_run += rs;
            pack_p();
            pv(Vs, false);
        }
        l_run += __shfl_xor(l_run, 32, 64);
    }
    // ---- finalize: lane (query n, h) holds d = 32 db + 8 (r / 4) + 4 h + (r % 4) ----
    if (qrow < a.Tq) {
        const float inv = 1.0f / l_run;
        bf16_t *op = a.O + (size_t)b * a.o_batch_stride + (size_t)qrow * a.o_row_stride + head * HD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint2 w;
                w.x = pack16<DT>(o[db][4 * i] * inv, o[db][4 * i + 1] * inv);
                w.y = pack16<DT>(o[db][4 * i + 2] * inv, o[db][4 * i + 3] * inv);
                *reinterpret_cast<uint2 *>(op + db * 32 + 8 * i + 4 * h) = w;
            }
    }
}


// ---- dispatch (inside m3_attention_dt, before the M3_ATTN launch) ----
#if 0
    // experiment: the 32 x 32 x 16 MFMA form for prescaled launches with bf16 P (M3_ATTN32=1)
    static const bool use32 = [] { const char *e = getenv("M3_ATTN32"); return e && atoi(e) != 0; }();
    if (use32 && pre && !safe_bf16 && (dtype == DT_BF16 || dtype == 2)) {
        if (wg128 >= 512) {
            if (dtype == 2) hipLaunchKernelGGL((k_attn32<4, DT_F16, DT_BF16>), dim3((unsigned)wg128), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_attn32<4, DT_BF16, DT_BF16>), dim3((unsigned)wg128), dim3(256), 0, st, a);
        } else {
            if (dtype == 2) hipLaunchKernelGGL((k_attn32<2, DT_F16, DT_BF16>), dim3((unsigned)wg64), dim3(128), 0, st, a);
            else hipLaunchKernelGGL((k_attn32<2, DT_BF16, DT_BF16>), dim3((unsigned)wg64), dim3(128), 0, st, a);
        }
    } else
#endif
