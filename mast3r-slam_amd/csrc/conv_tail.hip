// Tail of the DPT head as ONE direct convolution kernel for gfx950:
//
//   [x2 bilinear upsample (align_corners) of the 128-channel head.0 map]  ->  head.2 conv3x3 128->128 + bias + ReLU
//   ->  head.4 1x1 128->4  ->  pts3d = xyz/|xyz| * expm1(|xyz|), conf = 1 + exp(c)
//
// (public DPT head; oracle/model.py dpt_head :177-180, head :186-190).  The implicit-GEMM form (gemm.hip
// EPI_RELU_HEAD4) gathers every input pixel nine times - once per tap - straight from memory and needs the
// upsampled 512x512x128 map materialised first: k_upsample2x wrote 537 MB per head and the convolution read it
// back (9 x through L2).  Here a workgroup owns a 16 x 16 output tile, stages the 18 x 18 x 128 input halo ONCE in
// LDS - interpolating it on the way in when the upsample is fused, so the full-resolution map never exists - and
// runs the nine taps as 128 x 128 GEMM slices from LDS: 1.27 x the compulsory input traffic instead of 9 x, no
// upsample kernel, no 1 GB round trip.
//
// Workgroup = 512 threads = 8 waves on a 16 (rows) x 32 (pixels) output tile: wave (wp, wc) owns rows 4wp..4wp+3
// (8 MFMA row tiles of 16 pixels) x output channels 64wc..64wc+63 (4 column tiles) = 32 accumulator tiles, the
// 256 x 256 GEMM tile's shape (24 ds_read_b128 per 64 MFMAs).  The 128 input channels are processed as two halves:
// per half, the 18 x 34 x 64-channel halo (78 KiB, 128-byte pixel rows, 16-byte chunks XOR-swizzled by the halo
// column) is staged once, then nine taps x 64 channels of weights (16 KiB each, double-buffered LDS-DMA) are
// multiplied from LDS.  Weights streamed per workgroup: 288 KiB per 512 pixels (a 16 x 16 tile version streamed
// them per 256 pixels and was bound by exactly that L2 -> LDS traffic: 2.4 GB per launch, 913 us).
//
// The same kernel body, templated on the input width and the epilogue, is also head.0 with ITS upsample fused in
// (k_conv_tail<DT, true, 256, false>): [x2 upsample of refinenet1's 256-channel output] -> conv3x3 256->128 + bias ->
// 16-bit NHWC map.  Four 64-channel quarters instead of two halves, outputs transposed through LDS (the dead halo) into
// full 128-byte rows.  Before: k_upsample2x wrote the 256 x 256 x 256 map (537 MB for 16 images) and the implicit-GEMM
// convolution read 1.71 GB for it (208 + 717 us per step).
#include "gemm_common.h"

using namespace m3gemm;

namespace {

constexpr int kThreads = 512;
constexpr int TH = 16, TW = 32;                 // output tile
constexpr int HH = TH + 2, HW = TW + 2;         // halo 18 x 34
constexpr int kHaloPix = HH * HW;               // 612
constexpr int kHaloBytes = kHaloPix * 128;      // 78 336: one 64-channel half, 128 B per pixel
constexpr int kWStage = 128 * 128;              // 16 384: [128 out][64 in] of one tap
constexpr int PH = 11, PW = 19;                 // input patch behind an upsampled halo (pixels)
constexpr int kPatchBytes = (PH * PW * 128 + 1023) / 1024 * 1024;   // 27 648: whole 1 KiB LDS-DMA wave-instructions
constexpr int kLdsBytes = kHaloBytes + 2 * kWStage + kPatchBytes;   // 138 752

struct TailArgs {
    const bf16_t *X;        // NHWC [B, IH, IW, CIN]: IH = H (no upsample) or H / 2 (fused x2 upsample)
    const bf16_t *Wc;       // [128][3][3][CIN]
    const float *bias;      // [128] or null
    const bf16_t *W4;       // [4][128]
    const float *b4;        // [4]
    float *pts, *conf;      // [B,H,W,3], [B,H,W]                      (TAIL)
    bf16_t *Y;              // NHWC [B,H,W,cout] 16-bit: epi(conv + bias)   (!TAIL); a workgroup writes channels
                            //   [128 z, 128 z + 128), z = blockIdx.z (cout = 128 or 256)
    const bf16_t *R;        // EP_ADD: residual, laid out as Y
    int cout, relu_in;      // relu_in: the convolution reads relu(X) (fragments clamped in registers)
    const bf16_t *zero16;
    int B, H, W, IH, IW;
    // second head (blockIdx.y = 1): its own weights; X / pts / conf advance by one head's extent
    const bf16_t *Wc2, *W42;
    const float *bias2, *b42;
};

// EP: what happens to the 128 output channels of a workgroup
enum { EP_PLAIN = 0 /* Y = conv + bias */, EP_TAIL = 1 /* relu -> head.4 -> pointmap */, EP_RELU = 2 /* Y = relu(conv + bias) */,
       EP_ADD = 3 /* Y = R + conv + bias, one rounding */ };

template <int DT, bool UPS, int CIN, int EP>
__global__ void __launch_bounds__(kThreads, 2)
k_conv_tail(const TailArgs ain) {
    constexpr bool TAIL = EP == EP_TAIL;
    static_assert(CIN % 64 == 0 && (TAIL ? CIN == 128 : true), "64-channel steps; the fused tail is the 128-channel head.2");
    constexpr int NQ = CIN / 64;                            // 64-channel input slices ("halves" of the 128-channel tail)
    TailArgs a = ain;
    if (blockIdx.y == 1) {
        a.Wc = ain.Wc2; a.W4 = ain.W42; a.bias = ain.bias2; a.b4 = ain.b42;
        a.X = ain.X + (size_t)ain.B * ain.IH * ain.IW * CIN;
        if (TAIL) {
            a.pts = ain.pts + (size_t)ain.B * ain.H * ain.W * 3;
            a.conf = ain.conf + (size_t)ain.B * ain.H * ain.W;
        } else {
            a.Y = ain.Y + (size_t)ain.B * ain.H * ain.W * ain.cout;
            if (EP == EP_ADD) a.R = ain.R + (size_t)ain.B * ain.H * ain.W * ain.cout;
        }
    }
    if (!TAIL) {                                            // output-channel half of this workgroup: weight rows, bias, Y / R columns
        const int z = blockIdx.z;
        a.Wc += (size_t)z * 128 * 9 * CIN;
        if (a.bias) a.bias += z * 128;
        a.Y += z * 128;
        if (EP == EP_ADD) a.R += z * 128;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *halo = lds, *wst = lds + kHaloBytes, *patch = lds + kHaloBytes + 2 * kWStage;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wc = wave & 1;                 // pixel-row group (4 rows), output-channel half
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = a.H / TH;
    const int nwg = tiles_x * tiles_y * a.B;
    const int bid = xcd_remap(blockIdx.x, nwg);              // neighbouring tiles (shared halo rows) on one XCD
    const int b = bid / (tiles_x * tiles_y), trem = bid - b * tiles_x * tiles_y;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const bf16_t *img = a.X + (size_t)b * a.IH * a.IW * CIN;

    // weights of (tap, input half) -> stage buf: 128 rows x 8 chunks' = 1024 slots, 2 per thread
    auto stage_w = [&](int tap, int half, int buf) {
        unsigned char *base = wst + buf * kWStage;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int slot = i * kThreads + tid;
            const int row = slot >> 3, cp = slot & 7;
            const int c = half * 8 + (cp ^ ((row >> 1) & 7));
            glds16(a.Wc + ((size_t)row * 9 + tap) * CIN + c * 8, base + i * (kThreads * 16) + wave * 1024);
        }
    };
    // x2 bilinear (align_corners): the 18 x 34 halo of the H x W image blends an (at most) 11 x 19 input patch
    const float sy = UPS && a.H > 1 ? (float)(a.IH - 1) / (float)(a.H - 1) : 0.f;
    const float sx = UPS && a.W > 1 ? (float)(a.IW - 1) / (float)(a.W - 1) : 0.f;
    const int py0 = (int)((float)max(oy0 - 1, 0) * sy), px0 = (int)((float)max(ox0 - 1, 0) * sx);
    auto stage_patch = [&](int half) {          // 209 pixels x 8 chunks = 1672 slots -> 27 wave-instructions (padded)
        for (int wi = wave; wi < (PH * PW * 8 + 63) / 64; wi += kThreads / 64) {
            int slot = wi * 64 + lane;
            slot = slot < PH * PW * 8 ? slot : PH * PW * 8 - 1;
            const int pp = slot >> 3, c = slot & 7;
            const int y = min(py0 + pp / PW, a.IH - 1), x = min(px0 + pp % PW, a.IW - 1);
            glds16(img + ((size_t)y * a.IW + x) * CIN + (half * 8 + c) * 8, patch + wi * 1024);
        }
    };
    auto stage_halo_direct = [&](int half) {    // 612 pixels x 8 chunks' = 4896 slots = 76.5 wave-instructions
        for (int wi = wave; wi < (kHaloPix * 8 + 63) / 64; wi += kThreads / 64) {
            int slot = wi * 64 + lane;
            const bool live = slot < kHaloPix * 8;
            slot = live ? slot : kHaloPix * 8 - 1;
            const int pix = slot >> 3, cp = slot & 7;
            const int hy = pix / HW, hx = pix - hy * HW;
            const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
            const int c = half * 8 + (cp ^ ((hx >> 1) & 7));
            const bool in = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const void *src = in ? (const void *)(img + ((size_t)iy * a.IW + ix) * CIN + c * 8) : (const void *)a.zero16;
            // the last wave-instruction is half full: its upper 32 lanes would land in the weight stage - use a plain store
            if (wi * 64 + 64 <= kHaloPix * 8) glds16(src, halo + wi * 1024);
            else if (live) *reinterpret_cast<uint4 *>(halo + (size_t)slot * 16) = *reinterpret_cast<const uint4 *>(src);
        }
    };
    auto blend_halo = [&]() {                    // patch (LDS) -> halo (LDS), same arithmetic as k_upsample2x
        for (int slot = tid; slot < kHaloPix * 8; slot += kThreads) {
            const int pix = slot >> 3, cp = slot & 7;
            const int hy = pix / HW, hx = pix - hy * HW;
            const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
            const int c = cp ^ ((hx >> 1) & 7);
            uint4 o = make_uint4(0u, 0u, 0u, 0u);
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
                const float fy = iy * sy, fx = ix * sx;
                int y0 = (int)fy, x0 = (int)fx;
                y0 = min(y0, a.IH - 1); x0 = min(x0, a.IW - 1);
                const int y1 = min(y0 + 1, a.IH - 1), x1 = min(x0 + 1, a.IW - 1);
                const float wy = fy - (float)y0, wx = fx - (float)x0;
                const unsigned char *p = patch + c * 16;
                union U { uint4 q; unsigned w[4]; } q00, q01, q10, q11, r;
                q00.q = *reinterpret_cast<const uint4 *>(p + ((y0 - py0) * PW + (x0 - px0)) * 128);
                q01.q = *reinterpret_cast<const uint4 *>(p + ((y0 - py0) * PW + (x1 - px0)) * 128);
                q10.q = *reinterpret_cast<const uint4 *>(p + ((y1 - py0) * PW + (x0 - px0)) * 128);
                q11.q = *reinterpret_cast<const uint4 *>(p + ((y1 - py0) * PW + (x1 - px0)) * 128);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float tl = lo16<DT>(q00.w[k]) * (1.f - wx) + lo16<DT>(q01.w[k]) * wx;
                    const float bl = lo16<DT>(q10.w[k]) * (1.f - wx) + lo16<DT>(q11.w[k]) * wx;
                    const float th = hi16<DT>(q00.w[k]) * (1.f - wx) + hi16<DT>(q01.w[k]) * wx;
                    const float bh = hi16<DT>(q10.w[k]) * (1.f - wx) + hi16<DT>(q11.w[k]) * wx;
                    r.w[k] = pack16<DT>(tl * (1.f - wy) + bl * wy, th * (1.f - wy) + bh * wy);
                }
                o = r.q;
            }
            *reinterpret_cast<uint4 *>(halo + (size_t)slot * 16) = o;
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fch = lane >> 4;
    int w_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = wc * 64 + j * 16 + frow;
        w_off[j] = r * 128 + ((fch ^ ((r >> 1) & 7)) << 4);
    }

    // Step g = half * 9 + tap multiplies 64 input channels of one tap (64 MFMAs per wave) from weight stage g & 1;
    // the weights of step g + 1 are prefetched during step g into the other stage.  Waves 0-3 ("ping", the first wave
    // of every SIMD) and 4-7 ("pong") run   READ(ks0) | MFMA(ks0) | READ(ks1) | MFMA(ks1)   one phase apart with a
    // workgroup barrier per phase - the ping-pong schedule of gemm256.hip: while one wave of a SIMD issues its 32 MFMAs
    // the other issues its 12 ds_read_b128 (a version with every wave in the same phase took 54 us per tile against
    // 18 us of MFMA issue).
    const int group = wave >> 2;
    bf16x8 af[8], wf[4];
    // halo fragment addresses: the swizzle depends on the halo COLUMN only (chunk' = chunk ^ ((hx >> 1) & 7)), so the
    // per-lane part is one of six column offsets (kx = 0..2, left / right 16-pixel tile) and the rest is uniform:
    // 12 address instructions per read phase (a per-pixel swizzle cost ~100, as much VALU time as the MFMAs took)
    // MFMA row t of a 16-pixel tile is PIXEL kPixOf[t], not pixel t: a ds_read_b128 is served in the lane groups
    // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32) - all 16 rows at two different k-chunks - and with the natural
    // order two of the three tap shifts kx put two lanes of a group on the same (pixel parity, swizzled chunk) = the same
    // banks: SQ_LDS_BANK_CONFLICT was 24 % of the kernel's LDS cycles, 7 % of its time.  This order (found by search
    // over the 16! permutations, swizzle unchanged) is conflict-free for kx = 0, 1, 2, both tile halves and all four
    // groups; the epilogue maps accumulator rows back through the same table.
    const int pix_of_row = (int)((0x3D9F2A40E6C851B7ULL >> (4 * frow)) & 15);     // {7,11,1,5,8,12,6,14,0,4,10,2,15,9,13,3}
    // (A table colofs[kx][half] of the six column offsets, indexed by the runtime tap column, was placed in SCRATCH memory by
    // the compiler: two scratch loads - a vector-memory round trip - in front of every tap's fragment reads.  The ten VALU
    // instructions that compute the two offsets cost nothing next to that.)
    auto col_offset = [&](int hx) { return hx * 128 + ((fch ^ ((hx >> 1) & 7)) << 4); };
    auto read_frags = [&](int g, int tap, int ks) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const unsigned char *wsrc = wst + (g & 1) * kWStage;
        const int flip = ks << 6;
        const int c0 = col_offset(pix_of_row + kx) ^ flip;
        const int c1 = col_offset(16 + pix_of_row + kx) ^ flip;
        const unsigned char *rowp = halo + (wp * 4 + ky) * (HW * 128);
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8 *>(wsrc + (w_off[j] ^ flip));
#pragma unroll
        for (int i = 0; i < 8; ++i)
            af[i] = *reinterpret_cast<const bf16x8 *>(rowp + (i >> 1) * (HW * 128) + ((i & 1) ? c1 : c0));
        if (!TAIL && a.relu_in) {                           // kernel-uniform: conv(relu(x)) without a relu(x) tensor
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = relu_frag(af[i]);
        }
    };
    auto mfma_all = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<DT>(wf[j], af[i], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto phase_end = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto phase_end_wait = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto prefetch_w = [&](int g) { if (g + 1 < 9 * NQ) stage_w((g + 1) % 9, (g + 1) / 9, (g + 1) & 1); };
    // nine steps on the halo half that is in LDS; both groups execute the same number of barriers (4 * 9 + 1)
    auto run_half = [&](int half) {
        if (group == 0) {
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const int g = half * 9 + tap;
                read_frags(g, tap, 0);                         // phase 0
                prefetch_w(g);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                phase_end();
                mfma_all();                                    // phase 1
                phase_end();
                read_frags(g, tap, 1);                         // phase 2
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                phase_end();
                mfma_all();                                    // phase 3
                phase_end_wait();
            }
            phase_end();                                       // the pong group's drain phase
        } else {
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const int g = half * 9 + tap;
                if (tap > 0) mfma_all();                       // phase 0: k-step 1 of the previous tap
                phase_end();
                read_frags(g, tap, 0);                         // phase 1
                prefetch_w(g);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                phase_end();
                mfma_all();                                    // phase 2
                phase_end();
                read_frags(g, tap, 1);                         // phase 3
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                phase_end_wait();
            }
            mfma_all();                                        // drain: k-step 1 of the last tap
            phase_end();
        }
    };

    stage_w(0, 0, 0);
    if (UPS) stage_patch(0); else stage_halo_direct(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
#pragma unroll 1
    for (int q = 0; q < NQ; ++q) {
        if (UPS) {
            blend_halo();                                     // slice q: patch -> halo (q > 0: the patch streamed in under the
            lds_barrier();                                    //   previous slice's taps and was waited for by their vmcnt(0))
            if (q + 1 < NQ) stage_patch(q + 1);
        } else if (q > 0) {
            stage_halo_direct(q);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
        }
        run_half(q);                                          // ends behind a barrier: every read of this halo slice is done
    }

    if constexpr (EP == EP_ADD) {
        // ---- epilogue (residual unit's conv2): Y = R + (acc + bias), ONE rounding - the sub-tile goes through the dead halo
        // in fp32 (two 16-pixel tiles per pass: 32 rows x 272 B per wave), is read back 8 channels per lane (8 lanes = the
        // 128 contiguous bytes of a pixel's 64-channel half), meets the residual there and leaves as full 128-byte pieces
        const int r = lane & 15, gq = lane >> 4;
        constexpr int RS = 272;                              // 64 fp32 + 16 bytes of padding: conflict-free writes
        unsigned char *wl = halo + wave * (32 * RS);
        float4 bj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            bj[j] = a.bias ? *reinterpret_cast<const float4 *>(a.bias + wc * 64 + j * 16 + gq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        auto pix = [&](int pass, int rl, int &oy, int &ox) {
            const int i = pass * 2 + (rl >> 4);
            oy = oy0 + wp * 4 + (i >> 1);
            ox = ox0 + (i & 1) * 16 + (int)((0x3D9F2A40E6C851B7ULL >> (4 * (rl & 15))) & 15);
        };
        uint4 q[2][4];                                       // residual pieces of a pass, loaded one pass ahead of their use
        auto load_resid = [&](int pass, uint4 (&dst)[4]) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int c = it * 64 + lane, rl = c >> 3, ch = c & 7;
                int oy, ox;
                pix(pass, rl, oy, ox);
                ox = ox < a.W ? ox : a.W - 1;
                dst[it] = *reinterpret_cast<const uint4 *>(a.R + (((size_t)b * a.H + oy) * a.W + ox) * a.cout + wc * 64 + ch * 8);
            }
        };
        load_resid(0, q[0]);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            if (pass + 1 < 4) load_resid(pass + 1, q[(pass + 1) & 1]);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = acc[pass * 2 + ii][j];
                    *reinterpret_cast<float4 *>(wl + (ii * 16 + r) * RS + ((j * 4 + gq) << 4)) =
                        make_float4(v[0] + bj[j].x, v[1] + bj[j].y, v[2] + bj[j].z, v[3] + bj[j].w);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int c = it * 64 + lane, rl = c >> 3, ch = c & 7;
                const float4 f0 = *reinterpret_cast<const float4 *>(wl + rl * RS + ch * 32);
                const float4 f1 = *reinterpret_cast<const float4 *>(wl + rl * RS + ch * 32 + 16);
                const uint4 u = q[pass & 1][it];
                uint4 o;
                o.x = pack16<DT>(f0.x + lo16<DT>(u.x), f0.y + hi16<DT>(u.x));
                o.y = pack16<DT>(f0.z + lo16<DT>(u.y), f0.w + hi16<DT>(u.y));
                o.z = pack16<DT>(f1.x + lo16<DT>(u.z), f1.y + hi16<DT>(u.z));
                o.w = pack16<DT>(f1.z + lo16<DT>(u.w), f1.w + hi16<DT>(u.w));
                int oy, ox;
                pix(pass, rl, oy, ox);
                if (ox < a.W)
                    *reinterpret_cast<uint4 *>(a.Y + (((size_t)b * a.H + oy) * a.W + ox) * a.cout + wc * 64 + ch * 8) = o;
            }
            asm volatile("" ::: "memory");
        }
    } else if constexpr (!TAIL) {
        // ---- epilogue (head.0, residual unit's conv1): [relu](acc + bias) -> 16 bits, transposed through the dead halo (8 KiB
        // per wave) so that the 64 channels a wave owns leave as full 128-byte pieces of a pixel's row (the scratch layout and
        // its swizzle are epilogue_rows' 16-bit form, gemm_common.h: 8-byte writes / 16-byte reads, every bank once)
        const int r = lane & 15, gq = lane >> 4;
        unsigned char *wl = halo + wave * 8192;
        float4 bj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            bj[j] = a.bias ? *reinterpret_cast<const float4 *>(a.bias + wc * 64 + j * 16 + gq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = acc[pass * 4 + ii][j];
                    v[0] += bj[j].x; v[1] += bj[j].y; v[2] += bj[j].z; v[3] += bj[j].w;
                    if (EP == EP_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    }
                    uint2 pk;
                    pk.x = pack16<DT>(v[0], v[1]); pk.y = pack16<DT>(v[2], v[3]);
                    *reinterpret_cast<uint2 *>(wl + (ii * 16 + r) * 128 + (((j * 2 + (gq >> 1)) ^ ((r >> 1) & 7)) << 4) + (((gq ^ r) & 1) << 3)) = pk;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int c = it * 64 + lane, rl = c >> 3, ch = c & 7;
                uint4 v = *reinterpret_cast<const uint4 *>(wl + rl * 128 + ((ch ^ ((rl >> 1) & 7)) << 4));
                if (rl & 1) v = uint4{v.z, v.w, v.x, v.y};
                const int i = pass * 4 + (rl >> 4);
                const int oy = oy0 + wp * 4 + (i >> 1);
                const int ox = ox0 + (i & 1) * 16 + (int)((0x3D9F2A40E6C851B7ULL >> (4 * (rl & 15))) & 15);
                if (ox < a.W)
                    *reinterpret_cast<uint4 *>(a.Y + (((size_t)b * a.H + oy) * a.W + ox) * a.cout + wc * 64 + ch * 8) = v;
            }
            asm volatile("" ::: "memory");
        }
    } else {
        // ---- epilogue: h = relu(acc + bias) in fp32, raw[o] = sum_n h[n] W4[o][n] + b4[o], pointmap post-processing ----
        const int r = lane & 15, gq = lane >> 4;
        float part[8][4];
    #pragma unroll
        for (int i = 0; i < 8; ++i)
    #pragma unroll
            for (int o = 0; o < 4; ++o) part[i][o] = 0.f;
    #pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = wc * 64 + j * 16 + gq * 4;
            const float4 bq = a.bias ? *reinterpret_cast<const float4 *>(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            float w4[4][4];
    #pragma unroll
            for (int o = 0; o < 4; ++o) {
                const uint2 q = *reinterpret_cast<const uint2 *>(a.W4 + (size_t)o * 128 + n);
                w4[o][0] = lo16<DT>(q.x); w4[o][1] = hi16<DT>(q.x); w4[o][2] = lo16<DT>(q.y); w4[o][3] = hi16<DT>(q.y);
            }
    #pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 v = acc[i][j];
                const float h[4] = {fmaxf(v[0] + bq.x, 0.f), fmaxf(v[1] + bq.y, 0.f), fmaxf(v[2] + bq.z, 0.f), fmaxf(v[3] + bq.w, 0.f)};
    #pragma unroll
                for (int o = 0; o < 4; ++o)
                    part[i][o] += (h[0] * w4[o][0] + h[1] * w4[o][1]) + (h[2] * w4[o][2] + h[3] * w4[o][3]);
            }
        }
    #pragma unroll
        for (int i = 0; i < 8; ++i)
    #pragma unroll
            for (int o = 0; o < 4; ++o) {
                float v = part[i][o];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                part[i][o] = v;
            }
        // both channel halves' partial sums through LDS, then ONE pixel per lane: lane (r, gq) of wave (wp, wc) finishes
        // accumulator row tile i = 4 wc + gq (the first version left the 8 x (sqrt, expm1, exp) of a row to 16 lanes of
        // every second wave)
        float *red = reinterpret_cast<float *>(wst);             // weight stages are dead after the last barrier
        if (gq == 0) {
    #pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<float4 *>(red + (((wp * 2 + wc) * 8 + i) * 16 + r) * 4) = make_float4(part[i][0], part[i][1], part[i][2], part[i][3]);
        }
        __syncthreads();
        {
            const int i = 4 * wc + gq;
            const float4 p0 = *reinterpret_cast<const float4 *>(red + (((wp * 2 + 0) * 8 + i) * 16 + r) * 4);
            const float4 p1 = *reinterpret_cast<const float4 *>(red + (((wp * 2 + 1) * 8 + i) * 16 + r) * 4);
            const int oy = oy0 + wp * 4 + (i >> 1), ox = ox0 + (i & 1) * 16 + pix_of_row;      // r == frow
            if (ox < a.W) {                                       // W = 16 (mod 32): the tile's right half is outside
                const float x = p0.x + p1.x + a.b4[0], y = p0.y + p1.y + a.b4[1];
                const float z = p0.z + p1.z + a.b4[2], c = p0.w + p1.w + a.b4[3];
                const float d = sqrtf(x * x + y * y + z * z);
                const float sc = expm1f(d) / fmaxf(d, 1e-8f);
                const size_t m = ((size_t)b * a.H + oy) * a.W + ox;
                a.pts[3 * m + 0] = x * sc; a.pts[3 * m + 1] = y * sc; a.pts[3 * m + 2] = z * sc;
                a.conf[m] = 1.0f + expf(c);
            }
        }
    }
}

}  // namespace

extern "C" {

// X: NHWC [B, H/2, W/2, CIN] when upsample != 0 (the x2 bilinear, align_corners upsampling is done on the fly),
// else [B, H, W, CIN].  H, W multiples of 16.  ep = EP_TAIL: the fused tail (CIN = 128, pts / conf out); otherwise
// epi(conv + bias) -> Y NHWC [B,H,W,cout] 16-bit (CIN, cout in {128, 256}; R = residual for EP_ADD).
static int dpt_tail_launch(const void *X, const void *Wc, const float *bias, const void *W4, const float *b4,
                           const void *Wc2, const float *bias2, const void *W42, const float *b42, float *pts,
                           float *conf, void *Y, const void *R, const void *zero16, int B, int H, int W, int cin, int cout,
                           int ep, int relu_in, int upsample, int dtype, void *stream) {
    const int groups = Wc2 ? 2 : 1;
    const bool tail = ep == EP_TAIL;
    M3_REQUIRE(X && Wc && zero16 && B > 0 && H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0);
    M3_REQUIRE(tail ? (W4 && b4 && pts && conf && cin == 128 && !Y)
                    : (Y && (cin == 128 || cin == 256) && (cout == 128 || cout == 256) && (ep == EP_PLAIN || ep == EP_RELU || ep == EP_ADD)));
    M3_REQUIRE((ep == EP_ADD) == (R != nullptr) && (!upsample || ep == EP_PLAIN || tail));
    M3_REQUIRE(groups == 1 || ((!tail || (W42 && b42)) && (bias == nullptr) == (bias2 == nullptr)));
    M3_REQUIRE((dtype == DT_BF16 || dtype == DT_F16) && (!upsample || (H % 2 == 0 && W % 2 == 0)));
    M3_REQUIRE((int64_t)groups * B * H * W < (1ll << 31));
    M3_REQUIRE(reinterpret_cast<uintptr_t>(X) % 16 == 0 && reinterpret_cast<uintptr_t>(Wc) % 16 == 0 &&
               reinterpret_cast<uintptr_t>(Y) % 16 == 0 && reinterpret_cast<uintptr_t>(R) % 16 == 0);
    TailArgs a;
    a.Wc2 = (const bf16_t *)Wc2; a.W42 = (const bf16_t *)W42; a.bias2 = bias2; a.b42 = b42;
    a.X = (const bf16_t *)X; a.Wc = (const bf16_t *)Wc; a.bias = bias; a.W4 = (const bf16_t *)W4; a.b4 = b4;
    a.pts = pts; a.conf = conf; a.Y = (bf16_t *)Y; a.R = (const bf16_t *)R; a.cout = tail ? 128 : cout; a.relu_in = relu_in;
    a.zero16 = (const bf16_t *)zero16; a.B = B; a.H = H; a.W = W;
    a.IH = upsample ? H / 2 : H; a.IW = upsample ? W / 2 : W;
    const dim3 grid((unsigned)((H / TH) * ((W + TW - 1) / TW) * B), groups, tail ? 1 : cout / 128), blk(kThreads);
    hipStream_t st = (hipStream_t)stream;
#define M3_TAIL(DTV, UP, CI, EPV)                                                                                \
    do {                                                                                                         \
        static M3AttrOnce once;                                                                                  \
        int dev__;                                                                                               \
        if (m3_attr_need(once, &dev__)) {                                                                        \
            M3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_tail<DTV, UP, CI, EPV>),     \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes), "m3_dpt_tail/attr"); \
            m3_attr_done(once, dev__);                                                                           \
        }                                                                                                        \
        hipLaunchKernelGGL((k_conv_tail<DTV, UP, CI, EPV>), grid, blk, kLdsBytes, st, a);                        \
    } while (0)
#define M3_TAIL_DT(UP, CI, EPV) do { if (dtype == DT_F16) M3_TAIL(DT_F16, UP, CI, EPV); else M3_TAIL(DT_BF16, UP, CI, EPV); } while (0)
    if (tail) { if (upsample) M3_TAIL_DT(true, 128, EP_TAIL); else M3_TAIL_DT(false, 128, EP_TAIL); }
    else if (ep == EP_PLAIN && cin == 128) { if (upsample) M3_TAIL_DT(true, 128, EP_PLAIN); else M3_TAIL_DT(false, 128, EP_PLAIN); }
    else if (ep == EP_PLAIN) { if (upsample) M3_TAIL_DT(true, 256, EP_PLAIN); else M3_TAIL_DT(false, 256, EP_PLAIN); }
    else if (ep == EP_RELU) { if (cin == 128) M3_TAIL_DT(false, 128, EP_RELU); else M3_TAIL_DT(false, 256, EP_RELU); }
    else { if (cin == 128) M3_TAIL_DT(false, 128, EP_ADD); else M3_TAIL_DT(false, 256, EP_ADD); }
#undef M3_TAIL_DT
#undef M3_TAIL
    M3_CHECK_LAUNCH("m3_dpt_tail");
    return M3_OK;
}

int m3_dpt_tail_dt(const void *X, const void *Wc, const float *bias, const void *W4, const float *b4, float *pts,
                   float *conf, const void *zero16, int B, int H, int W, int upsample, int dtype, void *stream) {
    M3_REQUIRE(pts && conf);
    return dpt_tail_launch(X, Wc, bias, W4, b4, nullptr, nullptr, nullptr, nullptr, pts, conf, nullptr, nullptr, zero16, B, H, W,
                           128, 128, EP_TAIL, 0, upsample, dtype, stream);
}

// Both heads in one launch: X [2,B,h,w,128], pts [2,B,H,W,3], conf [2,B,H,W]; head g uses (Wc_g, bias_g, W4_g, b4_g).
int m3_dpt_tail_grouped2_dt(const void *X, const void *Wc0, const void *Wc1, const float *bias0, const float *bias1,
                            const void *W40, const void *W41, const float *b40, const float *b41, float *pts,
                            float *conf, const void *zero16, int B, int H, int W, int upsample, int dtype, void *stream) {
    M3_REQUIRE(Wc1 != nullptr && pts && conf);
    return dpt_tail_launch(X, Wc0, bias0, W40, b40, Wc1, bias1, W41, b41, pts, conf, nullptr, nullptr, zero16, B, H, W, 128, 128,
                           EP_TAIL, 0, upsample, dtype, stream);
}

// Direct 3x3 convolution to 128 output channels with the x2 upsample of its input fused in (head.0 of the DPT head:
// Cin = 256; Cin = 128 also accepted): X NHWC [B, H/2, W/2, Cin] (upsample) or [B, H, W, Cin], Wc [128][3][3][Cin],
// Y NHWC [B, H, W, 128] = conv(up(X)) + bias, 16-bit.  H, W multiples of 16 (of 2 as well with the upsample).
int m3_conv3x3_up_direct_dt(const void *X, const void *Wc, const float *bias, void *Y, const void *zero16, int B, int H,
                            int W, int Cin, int upsample, int dtype, void *stream) {
    M3_REQUIRE(Y != nullptr);
    return dpt_tail_launch(X, Wc, bias, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, Y, nullptr, zero16,
                           B, H, W, Cin, 128, EP_PLAIN, 0, upsample, dtype, stream);
}

// ... for both heads in one launch: X [2,B,h,w,Cin], Y [2,B,H,W,128]; head g uses (Wc_g, bias_g).
int m3_conv3x3_up_direct_grouped2_dt(const void *X, const void *Wc0, const void *Wc1, const float *bias0, const float *bias1,
                                     void *Y, const void *zero16, int B, int H, int W, int Cin, int upsample, int dtype,
                                     void *stream) {
    M3_REQUIRE(Y != nullptr && Wc1 != nullptr);
    return dpt_tail_launch(X, Wc0, bias0, nullptr, nullptr, Wc1, bias1, nullptr, nullptr, nullptr, nullptr, Y, nullptr, zero16, B,
                           H, W, Cin, 128, EP_PLAIN, 0, upsample, dtype, stream);
}

// The same direct convolution as a general 3x3 / padding 1 / stride 1 operator for the wide DPT maps (the residual units
// of the fusion blocks: 256 -> 256 at 128 x 128 and 64 x 64): Cin, Cout in {128, 256}; epilogue M3_EPI_BF16 | _RELU | _ADD
// (R = residual, laid out as Y), optionally OR-ed with M3_EPI_INPUT_RELU.  SAME BITS as m3_conv3x3_dt on the same
// operands: both walk K as (64-channel slice, tap, k-step) and apply the epilogue in the same order, so a caller may pick
// either by problem size.  H, W multiples of 16.  W1 == NULL: one group.
int m3_conv3x3_direct_grouped2_dt(const void *X, const void *W0, const void *W1, const float *bias0, const float *bias1,
                                  void *Y, const void *R, const void *zero16, int B, int H, int W, int Cin, int Cout,
                                  int epilogue, int dtype, void *stream) {
    const int relu_in = (epilogue & 0x100) ? 1 : 0;
    const int e = epilogue & 0xff;
    M3_REQUIRE(Y != nullptr && (e == M3_EPI_BF16 || e == M3_EPI_BF16_RELU || e == M3_EPI_BF16_ADD));
    const int ep = e == M3_EPI_BF16 ? EP_PLAIN : (e == M3_EPI_BF16_RELU ? EP_RELU : EP_ADD);
    return dpt_tail_launch(X, W0, bias0, nullptr, nullptr, W1, bias1, nullptr, nullptr, nullptr, nullptr, Y, R, zero16, B, H, W,
                           Cin, Cout, ep, relu_in, 0, dtype, stream);
}

}  // extern "C"
