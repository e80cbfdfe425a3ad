// Implicit-GEMM (dilated) convolution, second-generation kernel for gfx950 — forward (K1) and
// data-gradient (K2).  Same math and layouts as conv_igemm.hip (activations NHWC bf16, weights
// KRSC bf16, output NHWC bf16 with row stride ldy), restructured around what the rocprofv3 counters
// of the first kernel showed on the ResNet-101 layer-3 shapes (profiles/r1_pmc_l3_convs_v4_b32.txt):
// 7.4 VALU instructions per MFMA (address generation), 18 % MFMA utilisation, each K-step exposed to
// a full memory round trip because only ONE stage was ever in flight.
//
//   * operand tiles are fetched with `buffer_load_dwordx4 … lds` through buffer descriptors: the
//     per-lane byte offset is computed ONCE, every K-step only adds a wave-uniform (SGPR) tap/channel
//     offset, and padding / tails are a per-lane "offset := out-of-range" select — the hardware range
//     check writes the zeros.  Padding validity of the R·S taps is a per-row bit mask built in the
//     prologue, so a K-step costs 4 VALU per activation load and 0 per weight load (1×1, pad 0: 0).
//   * NST-deep LDS ring (2–4 stages) with counted `s_waitcnt vmcnt(N)` and raw `s_barrier`: NST-1
//     stages stay in flight across the barrier instead of draining every step.
//   * 8-wave (512-thread) block tiles up to 256×256 for the wide layers (31 B/clk/CU of operand
//     traffic at MFMA peak instead of 64 for 128²), 4-wave 128×128/128×64 for the small ones.
//   * v_mfma_f32_16x16x32_bf16 with the WEIGHT tile as the A operand: a lane's 4 accumulator
//     registers are then 4 consecutive output channels of one pixel, so the epilogue packs them to
//     one 8-byte LDS store (4× fewer LDS stores than column-per-lane), and the tile leaves LDS as
//     coalesced 16-byte row stores.
//   * fused BatchNorm statistics: per-lane Σ / Σ² over the pixel tiles in registers, 16-lane DPP
//     row reduction, cross-wave combine in LDS, one coalesced fp32 atomic pair per (block, channel).
//   * strided data-gradient needs no special loop: with sdiv = forward stride, a tap is valid only
//     where (out − pad + r·dil) is a multiple of sdiv, and there floor(base/sdiv) + ceil(tap/sdiv)
//     is exact, so the per-lane base and the uniform tap offset stay separable.
#include "common.h"
#include <stdlib.h>
#include <stdio.h>

namespace {

// K elements per step: 64 (128-B operand rows) or 32 (64-B rows).  At 256x256 two 64-wide stages fill
// the LDS, so the ring is only 2 deep and every step drains the stage it just issued (vmcnt(0)): the
// DMA latency is exposed once per step.  32-wide stages make the same 128 KiB a 4-deep ring — two
// stages stay in flight across every barrier.
constexpr unsigned kOOB = 0x80000000u;       // byte offset beyond any descriptor (tensors < 2 GiB)

typedef __attribute__((ext_vector_type(4))) float f32x4v;

struct Conv2Args {
    const bf16_raw* x;      // [N, Hi, Wi, Cin]
    const bf16_raw* w;      // [Cout, R, S, Cin]
    bf16_raw* y;            // [M, ldy]
    const float* bias;      // [Cout] or null
    float* stats;           // [stat_reps, 2, Cout] or null (never together with bias)
    const bf16_raw* res;    // [M, ldres] or null: added to the output in the store loop (the skip-path
    int ldres;              //   gradient of a residual block joins the data gradient here, no add kernel)
    const float* res_scale; const float* res_shift;   // [Cout] or null: the store loop adds res * scale + shift
                            //   (ops.sepconv: the statistics term of a folded BatchNorm's data gradient)
    // Inference epilogue (dp_conv_igemm2e): y = act(conv * oscale + bias [+ res]) — an evaluation-mode BatchNorm
    // (per-channel scale | shift of its running statistics) and the ReLU behind it leave with the convolution's own
    // store instead of in a second pass over the activation.
    const float* oscale;    // [Cout] or null (1)
    int relu;               // clamp at zero: on the accumulators without res, behind the residual add with it
    // Fused BatchNorm+ReLU backward prologue (data-gradient launches only).  The tensor this kernel
    // differentiates, a = relu(BN(bnx) [+ identity]), is the previous layer's activation; instead of
    // da the store loop writes dm = da·[a > 0] and accumulates Σdm, Σdm·x̂ per channel — the two
    // reductions BN backward needs — so the separate bn_bwd_reduce pass (dy, x, y re-read) is gone.
    const bf16_raw* bnx;    // [M, Cout] pre-BN activations (x̂ = (bnx − mean)·rstd); null with bnsums given: mask and Σdm
                            //   only (ops.bnfold: Σdm·x̂ of a BatchNorm behind a 1x1 convolution comes out of that
                            //   convolution's weight gradient, so the pre-BN tensor is not read here at all)
    const bf16_raw* bna;    // [M, Cout] a itself (mask source) or null: mask = bnx·scale + shift > 0
    const unsigned char* bnmask;   // [M, Cout/8] positivity bits of a (bn_relu.hip, ApplyArgs::mask): the mask source
                            //   of a residual block's output at 1/16 of the bytes of bna; takes precedence over bna
    const float* bn_mean; const float* bn_rstd;   // [Cout]
    const float* bn_w; const float* bn_b;         // [Cout] affine parameters or null (1, 0)
    float* bnsums;          // [bn_reps, 2, Cout]
    int bn_reps;
    float mask_scale;       // dm = da·[a > 0]·mask_scale: 1, or 1/(1-p) when a dropout sits behind the ReLU (its zeros
                            //   are in `a` already — the mask source — so only the scale of the kept elements is missing)
    int stat_reps;
    int M;
    int Hi, Wi, Cin, Ho, Wo, Cout, R, S;
    int ldx;                // elements between neighbouring input pixels: Cin, or less for OVERLAPPING
                            //   windows (a pixel's "channels" run on into its right-hand neighbours: the
                            //   small-Cin stem reads R row segments of S·4 elements instead of R·S taps)
    int smul, sdl, pad_h, pad_w, dil_h, dil_w;   // sdl = log2(sdiv)
    int ldy;
    int csteps, ksteps, ctail;
    int tiles_m, tiles_n;
    unsigned x_bytes, w_bytes;
    int ablate;             // diagnostics only (dp_conv_set_ablate): 1 = no LDS-DMA in the K loop,
                            // 4 = no fragment reads and no MFMA; results are garbage, the TIME tells the bound
    int w_img, img_px;      // per-image weights (channel attention: a 1x1 convolution whose [Cout, Cin] matrix differs per
                            //   image): elements between two images' weights and pixels per image (a multiple of the tile
                            //   height, so a tile never straddles two images); w_img = 0: one weight tensor for all
#ifdef DP_STAMP
    unsigned long long* dbg;   // diagnostic build only (tools/stamp_conv.py): 7 s_memtime + 2 s_memrealtime stamps per block (stride 16)
#endif
};

// Phase stamps of a block's life (diagnostic build -DDP_STAMP, never in libdpamd_hip.so): scalar s_memtime
// reads kept in SGPRs, written by one lane at the very end.
#ifdef DP_STAMP
#define DP_STAMP_DECL unsigned long long stamp_[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define DP_STAMP_AT(i) stamp_[i] = __builtin_amdgcn_s_memtime()
#else
#define DP_STAMP_DECL
#define DP_STAMP_AT(i)
#endif

// (Spreading a stage's LDS-DMA pieces over the MFMA groups of the K loop — what gained 4-6 % in the weight-gradient
// kernel's 3-stage ring — is 1 % SLOWER here: with 2-stage rings a piece issued late in step t has less than a step
// left to land before step t + 1 waits for it; profiles/r2_conv_dma_spread_negative.log.  Not built.)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_wave_base, unsigned voff,
                                          unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff,
                                             soff, 0, 0);
}

// sum over the 16 lanes of a DPP row (all lanes of the row end up with the total)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

// read/source swizzle of a tile row's 16-byte chunks (an involution applied on both sides; the DMA
// destination stays linear).  128-B rows: 2 rows per 256-B bank row -> XOR (row>>1)&7.  64-B rows: 4 rows
// per bank row, and a ds_read_b128 lane group pairs rows {0-3,12-15} of one chunk column with rows {4-11}
// of the neighbouring one (MI355X_MICROARCH.md, LDS), which the table {0,3,2,1}[(row>>2)&3] keeps apart.
template <int BK>
__device__ __forceinline__ int chunk_swz(int row) {
    return BK == 64 ? ((row >> 1) & 7) : ((0 - (row >> 2)) & 3);
}

template <int N>
__device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

typedef __attribute__((ext_vector_type(4))) int i32x4v;

__device__ __forceinline__ void lds_read16(i32x4v& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}

// rows per batch of the fused BN-backward store loop: three loads per row, all issued before the first is consumed.
// 8 (6) rows instead of 4 (3): -4...-7 % on the 1x1 data gradients of ResNet-101 (gpurun_out/ad: the loop moves
// 4 tensors per output and runs at the bytes in flight per CU until it reaches the HBM rate)
#ifndef DP_BN_GRP
#define DP_BN_GRP(NR) ((NR) % 8 == 0 ? 8 : ((NR) % 6 == 0 ? 6 : ((NR) % 4 == 0 ? 4 : ((NR) % 3 == 0 ? 3 : ((NR) % 2 == 0 ? 2 : 1)))))
#endif

// Fragment reads are software-pipelined against the MFMAs by hand (see the K loop).
// INFER: the evaluation epilogue (Conv2Args::oscale / relu) is compiled into its own instances — as run-time
// branches of the one kernel it cost the TRAINING step 0.9 % (same-box A/B, profiles/r3_eval_forward_fused.md)
template <int BM, int BN, int WGM, int WGN, int NST, int MINW, bool UNIT, int BK = 64, bool INFER = false>
__global__ __launch_bounds__(64 * WGM * WGN, MINW) void conv_igemm2_kernel(Conv2Args p) {
    constexpr int T = 64 * WGM * WGN;
    constexpr int CPR = BK / 8;                          // 16-byte chunks per tile row
    constexpr int ROWB = BK * 2;                         // bytes per tile row
    constexpr int RPR = T / CPR;                         // tile rows filled per load round
    constexpr int AR = BM / RPR, BR = BN / RPR;          // load rounds per stage
    constexpr int L = AR + BR;                           // LDS-DMA instructions per thread per stage
    constexpr int WTM = BM / WGM, WTN = BN / WGN;        // wave tile: pixels × channels
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int C_LD = BN * 2 + 8;                     // epilogue row stride: 16 rows -> 32 distinct banks
    static_assert(BM % RPR == 0 && BN % RPR == 0, "tile rows must be whole load rounds");
    static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile must be whole 16x16 MFMA tiles");
    static_assert((NST - 1) * L <= 63, "vmcnt field");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    DP_STAMP_DECL;
#ifdef DP_STAMP
    stamp_[7] = __builtin_amdgcn_s_memrealtime();
    stamp_[9] = __builtin_amdgcn_s_getreg(63492);    // HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]
    stamp_[10] = __builtin_amdgcn_s_getreg(63508);   // XCC_ID
#endif
    // (A start-time stagger of the first round of blocks by wave slot / CU — so that one block's store loop overlaps
    // another's K loop — was measured at +-0.5 % of the step over 12 settings: README "De-phasing".  Not built.)
    DP_STAMP_AT(0);

    // ---- XCD-aware tile mapping (bijective for any grid size) ----
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % p.tiles_n;
    const int tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const bf16_raw* wimg = p.w_img ? p.w + (size_t)(m0 / p.img_px) * p.w_img : p.w;       // uniform
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wimg, 0, (int)p.w_bytes, 0x00020000);

    // ---- per-thread load geometry: lane -> (row rr of the round, physical 16-B chunk pc) ----
    const int pc = tid & (CPR - 1);
    const int rr = tid / CPR;
    const int lc = pc ^ chunk_swz<BK>(rr);   // logical chunk fetched (source-side swizzle)
    unsigned a_voff[AR], a_mask[AR];
    {
        const int sdm = (1 << p.sdl) - 1;
        int m = m0 + rr;
        int ox = m % p.Wo;
        int t = m / p.Wo;
        int oy = t % p.Ho;
        int n = t / p.Ho;
        const int adv_x = RPR % p.Wo, adv_y = RPR / p.Wo;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int by = oy * p.smul - p.pad_h, bx = ox * p.smul - p.pad_w;
            const int fby = by >> p.sdl, fbx = bx >> p.sdl;          // floor division (arithmetic shift)
            const unsigned base = (unsigned)((((n * p.Hi + fby) * p.Wi + fbx) * p.ldx + lc * 8) * 2);
            const bool row_ok = m < p.M;
            if (UNIT) {
                a_voff[i] = row_ok ? base : kOOB;
                a_mask[i] = 0;
            } else {
                unsigned vx = 0, mk = 0;
                for (int s = 0; s < p.S; ++s) {
                    const int ix = bx + s * p.dil_w;
                    const unsigned ok = (unsigned)((ix & sdm) == 0) & (unsigned)((unsigned)(ix >> p.sdl) < (unsigned)p.Wi);
                    vx |= ok << s;
                }
                for (int r = 0; r < p.R; ++r) {
                    const int iy = by + r * p.dil_h;
                    const bool ok = ((iy & sdm) == 0) & ((unsigned)(iy >> p.sdl) < (unsigned)p.Hi);
                    mk |= ok ? (vx << (r * p.S)) : 0u;
                }
                a_voff[i] = base;
                a_mask[i] = row_ok ? mk : 0u;
            }
            // next round: RPR pixels further
            m += RPR;
            ox += adv_x; oy += adv_y;
            if (ox >= p.Wo) { ox -= p.Wo; oy += 1; }
            while (oy >= p.Ho) { oy -= p.Ho; n += 1; }
        }
    }
    unsigned b_voff[BR];
    {
        const int w_row_stride = p.R * p.S * p.Cin;
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int n = n0 + i * RPR + rr;
            b_voff[i] = n < p.Cout ? (unsigned)((n * w_row_stride + lc * 8) * 2) : kOOB;
        }
    }

    // ---- taps that are padding for EVERY pixel of this tile are not iterated at all ----
    // A dilated 3x3 on a 33x33 map (ASPP, dilation 12 / 18) has whole bands of rows for which the taps
    // one dilation above (or below) fall outside the image: a 192-pixel tile is ~6 image rows, so for
    // those tiles three of the nine taps would be K-steps of zeros (DMA of out-of-range lanes, MFMAs on
    // zeros).  tile_mask = OR of the per-row validity masks; the iterator below walks its set bits only.
    unsigned tile_mask = 0xffffffffu;
    int ksteps = p.ksteps;
    if (!UNIT) {
        unsigned mk = 0;
#pragma unroll
        for (int i = 0; i < AR; ++i) mk |= a_mask[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mk |= (unsigned)__shfl_xor((int)mk, off, 64);
        unsigned* sm = reinterpret_cast<unsigned*>(smem);
        if (lane == 0) sm[wave] = mk;
        __syncthreads();
        unsigned all = 0;
#pragma unroll
        for (int w2 = 0; w2 < T / 64; ++w2) all |= sm[w2];
        tile_mask = (unsigned)__builtin_amdgcn_readfirstlane((int)all);
        __syncthreads();                                   // smem is about to receive the first stages
        ksteps = __builtin_popcount(tile_mask) * p.csteps;
    }

    // ---- K-step iterator of the ISSUE side (uniform): filter tap (kr, ks), 64-channel slice kc ----
    int kr = 0, ks = 0, kc = 0;
    if (!UNIT) {
        while (kr < p.R && !((tile_mask >> (kr * p.S + ks)) & 1u)) { if (++ks == p.S) { ks = 0; ++kr; } }
    }
    const int sdr = (1 << p.sdl) - 1;
    // A stage is issued piece by piece (one piece = one 1-KiB wave instruction): stage_begin does the uniform address
    // math, stage_piece(j) issues operand piece j (0 .. AR-1: pixels, AR .. AR+BR-1: weights), stage_end advances the
    // K-step iterator.  The prologue issues whole stages; the K loop spreads a stage's pieces between its MFMA groups.
    struct StageCtx { unsigned char* abase; unsigned char* bbase; unsigned w_soff, a_soff, tapbit; bool cv; };
    auto stage_begin = [&](int buf) -> StageCtx {
        StageCtx c;
        c.abase = smem + buf * STAGE + wave * 1024;       // one wave-instruction fills 1 KiB
        c.bbase = c.abase + A_BYTES;
        c.w_soff = (unsigned)(((kr * p.S + ks) * p.Cin + kc * BK) * 2);
        c.cv = !p.ctail || (lc * 8 < p.Cin - kc * BK);
        if (UNIT) {
            c.a_soff = (unsigned)(kc * BK * 2);
            c.tapbit = 0;
        } else {
            const int cdy = (kr * p.dil_h + sdr) >> p.sdl, cdx = (ks * p.dil_w + sdr) >> p.sdl;   // ceil(tap / sdiv)
            c.a_soff = (unsigned)(((cdy * p.Wi + cdx) * p.ldx + kc * BK) * 2);
            c.tapbit = 1u << (kr * p.S + ks);
        }
        return c;
    };
    auto stage_piece = [&](const StageCtx& c, int j) {
        if (j < AR) {
            if (UNIT) {
                lds_dma16(rx, c.abase + j * (RPR * ROWB), c.cv ? a_voff[j] : kOOB, c.a_soff);
            } else {
                const bool ok = ((a_mask[j] & c.tapbit) != 0) & c.cv;
                // the sum is formed per lane: the base may be "negative" (padding) and only base+tap is in range
                lds_dma16(rx, c.abase + j * (RPR * ROWB), ok ? a_voff[j] + c.a_soff : kOOB, 0);
            }
        } else {
            lds_dma16(rw, c.bbase + (j - AR) * (RPR * ROWB), b_voff[j - AR], c.w_soff);
        }
    };
    auto stage_end = [&]() {
        if (++kc == p.csteps) {
            kc = 0;
            do { if (++ks == p.S) { ks = 0; ++kr; } } while (!UNIT && kr < p.R && !((tile_mask >> (kr * p.S + ks)) & 1u));
        }
    };
    auto issue_stage = [&](int buf) {
        const StageCtx c = stage_begin(buf);
#pragma unroll
        for (int j = 0; j < AR + BR; ++j) stage_piece(c, j);
        stage_end();
    };

    // ---- accumulators: acc[mi][ni][j] = out(pixel tile mi, lane&15 ; channel tile ni, (lane>>4)*4 + j) ----
    f32x4v acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4v{0.f, 0.f, 0.f, 0.f};

    const int wm = wave / WGN, wn = wave % WGN;
    const int l15 = lane & 15, q = lane >> 4;
    const int fsw = chunk_swz<BK>(l15);                         // read-side swizzle (tile bases are multiples of 16)
    const int x_row_off = (wm * WTM + l15) * ROWB;
    const int w_row_off = A_BYTES + (wn * WTN + l15) * ROWB;
    const int ch0 = ((q ^ fsw) << 4), ch1 = (((4 + q) ^ fsw) << 4);      // ch1: second half of a 64-wide step

    // ---- prologue: NST-1 stages in flight ----
    DP_STAMP_AT(1);
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < ksteps) issue_stage(s);

    int buf = 0;
    for (int t = 0; t < ksteps; ++t) {
        // stage t must have landed; up to NST-2 younger stages may stay in flight
        const int younger = min(NST - 2, ksteps - 1 - t);
        if (NST >= 4 && younger >= 2) wait_vmcnt<(NST >= 4 ? 2 : 0) * L>();
        else if (NST >= 3 && younger >= 1) wait_vmcnt<(NST >= 3 ? 1 : 0) * L>();
        else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // my reads of the buffer about to be refilled are done
        __builtin_amdgcn_s_barrier();
#ifdef DP_STAMP
        if (t == 0) DP_STAMP_AT(2);
#endif
        const bool do_issue = t + NST - 1 < ksteps && !(p.ablate & 1);      // uniform
        int ib = buf + NST - 1;
        if (ib >= NST) ib -= NST;
        if (do_issue) issue_stage(ib);
        if (p.ablate & 4) { if (++buf == NST) buf = 0; continue; }
        {
            // Left to itself hipcc sinks every fragment read to just before its first MFMA (2 reads, wait,
            // 8 MFMAs, 2 reads, wait, ...: the ISA of the first version), so each wave exposes an LDS
            // round trip per pair of pixel tiles and reads and MFMAs of a wave never overlap — the time of
            // a step was reads + MFMAs (tools/ablate_conv.py: 421 + 526 us of a 993 us loop).  Here the
            // reads are inline asm in a fixed order: the NEXT group of pixel-tile fragments (and, at the end
            // of a 32-wide half, the next half's weight fragments) is issued BEFORE the MFMAs of the
            // current group, and each group waits with a COUNTED lgkmcnt for exactly what it consumes.
            constexpr int KK = BK / 32;
            constexpr int PA = MI % 2 == 0 ? 2 : 1;            // pixel tiles per fragment group
            constexpr int G = MI / PA;                         // groups per 32-wide half
            const unsigned sb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + buf * STAGE;
            const unsigned xa0 = sb + x_row_off + ch0, xa1 = sb + x_row_off + ch1;
            const unsigned wa0 = sb + w_row_off + ch0, wa1 = sb + w_row_off + ch1;
            i32x4v wfr[2][NI], xfr[2][PA];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) lds_read16(wfr[0][ni], wa0 + ni * (16 * ROWB));
#pragma unroll
            for (int a = 0; a < PA; ++a) lds_read16(xfr[0][a], xa0 + a * (16 * ROWB));
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int cur = (kk * G + g) & 1;
                    int newer = 0;                             // reads issued after the ones this group consumes
                    if (g + 1 < G) {
#pragma unroll
                        for (int a = 0; a < PA; ++a)
                            lds_read16(xfr[cur ^ 1][a], (kk == 0 ? xa0 : xa1) + ((g + 1) * PA + a) * (16 * ROWB));
                        newer = PA;
                    } else if (kk + 1 < KK) {
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) lds_read16(wfr[(kk + 1) & 1][ni], wa1 + ni * (16 * ROWB));
#pragma unroll
                        for (int a = 0; a < PA; ++a) lds_read16(xfr[cur ^ 1][a], xa1 + a * (16 * ROWB));
                        newer = NI + PA;
                    }
                    if (newer == 0) wait_lgkm<0>();
                    else if (newer == PA) wait_lgkm<PA>();
                    else wait_lgkm<NI + PA>();
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int a = 0; a < PA; ++a)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[g * PA + a][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, wfr[kk & 1][ni]), __builtin_bit_cast(bf16x8, xfr[cur][a]),
                                acc[g * PA + a][ni], 0, 0, 0);
                }
            }
        }
        if (++buf == NST) buf = 0;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // all waves done reading operand tiles; LDS is reused for the epilogue
    DP_STAMP_AT(3);

    // ---- epilogue ----
    const bool do_stats = p.stats != nullptr;                                // uniform
    float* stat_scratch = reinterpret_cast<float*>(smem + BM * C_LD);        // [WGM][2][BN]
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int col = wn * WTN + ni * 16 + q * 4;          // first of this lane's 4 channels (tile-local)
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n0 + col + j < p.Cout) b4[j] = p.bias[n0 + col + j];
        }
        float sc4[INFER ? 4 : 1];
        bool relu_acc = false;                                                        // uniform
        if constexpr (INFER) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sc4[j] = (p.oscale && n0 + col + j < p.Cout) ? p.oscale[n0 + col + j] : 1.f;
            relu_acc = p.relu && p.res == nullptr;
        }
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int row = wm * WTM + mi * 16 + l15;
            float v[4];
            if constexpr (INFER) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = fmaf(acc[mi][ni][j], sc4[j], b4[j]);
                    v[j] = relu_acc ? fmaxf(v[j], 0.f) : v[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[mi][ni][j] + b4[j];
            }
            if (do_stats) {
                // rows beyond M were fed zeros and there is no bias on this path: they add exactly 0
#pragma unroll
                for (int j = 0; j < 4; ++j) { s1[j] += v[j]; s2[j] = fmaf(v[j], v[j], s2[j]); }
            }
            uint2 pk;
            pk.x = pack2(v[0], v[1]);
            pk.y = pack2(v[2], v[3]);
            *reinterpret_cast<uint2*>(smem + row * C_LD + col * 2) = pk;
        }
        if (do_stats) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { s1[j] = row16_sum(s1[j]); s2[j] = row16_sum(s2[j]); }
            if (l15 == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    stat_scratch[(wm * 2 + 0) * BN + col + j] = s1[j];
                    stat_scratch[(wm * 2 + 1) * BN + col + j] = s2[j];
                }
            }
        }
    }
    __syncthreads();
    DP_STAMP_AT(4);
    if (do_stats) {
        // replica = tile_m mod R spreads the per-channel atomics of a large map over R cache lines
        float* dst = p.stats + (long)(tile_m % p.stat_reps) * 2 * p.Cout;
        for (int i = tid; i < 2 * BN; i += T) {
            const int which = i / BN, c = i - which * BN;
            float v = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < WGM; ++w2) v += stat_scratch[(w2 * 2 + which) * BN + c];
            if (n0 + c < p.Cout) atomicAdd(dst + which * p.Cout + n0 + c, v);
        }
    }
    {
        constexpr int TPR = BN / 8;               // threads per row (16-B pieces)
        constexpr int RPP = T / TPR;              // rows per pass
        const int cpiece = tid % TPR, rsub = tid / TPR;
        const int gcol = n0 + cpiece * 8;
        if (p.bnsums == nullptr) {
            float rsc[8], rsh[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool okc = p.res_scale != nullptr && gcol + j < p.Cout;
                rsc[j] = okc ? p.res_scale[gcol + j] : 1.f;
                rsh[j] = okc ? p.res_shift[gcol + j] : 0.f;
            }
            if (p.res == nullptr) {
#pragma unroll 4
                for (int r0 = 0; r0 < BM; r0 += RPP) {
                    const int row = r0 + rsub;
                    const int m = m0 + row;
                    if (m < p.M && gcol < p.Cout) {
                        const uint2 lo = *reinterpret_cast<const uint2*>(smem + row * C_LD + cpiece * 16);
                        const uint2 hi = *reinterpret_cast<const uint2*>(smem + row * C_LD + cpiece * 16 + 8);
                        *reinterpret_cast<uint4*>(p.y + (long)m * p.ldy + gcol) = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    }
                }
            } else {
                // residual (+ per-channel affine) added on the way out.  The loads of a whole batch of RG rows are
                // issued before the first is consumed: with one load per row in flight the loop was a chain of
                // exposed memory round trips (tools/stamp_conv.py: 14-16 us of a 256x256 block's 45 us against
                // 2.2 us for the plain store loop), i.e. bound by the bytes in flight per CU, not by HBM.
                const bf16_raw* __restrict__ resp = p.res;
                bf16_raw* __restrict__ yout = p.y;
                const bool affine = p.res_scale != nullptr;       // uniform
                constexpr int NR = BM / RPP;
                constexpr int RG = NR % 8 == 0 ? 8 : (NR % 6 == 0 ? 6 : (NR % 4 == 0 ? 4 : (NR % 3 == 0 ? 3 : (NR % 2 == 0 ? 2 : 1))));
                const bool col_ok = gcol < p.Cout;
                const int gcl = col_ok ? gcol : 0;
                for (int r0 = 0; r0 < BM; r0 += RG * RPP) {
                    uint4 vr[RG];
                    bool ok[RG];
                    // (no branch around a load: rows / columns beyond the tensor re-read its last row / first columns.
                    // With a load under a branch hipcc cannot count what is outstanding and waits vmcnt(0) before
                    // every row — i.e. for the previous row's STORE, one exposed write round trip per row.)
#pragma unroll
                    for (int g = 0; g < RG; ++g) {
                        const int m = m0 + r0 + g * RPP + rsub;
                        ok[g] = (m < p.M) && col_ok;
                        vr[g] = *reinterpret_cast<const uint4*>(resp + (long)min(m, p.M - 1) * p.ldres + gcl);
                    }
#pragma unroll
                    for (int g = 0; g < RG; ++g) {
                        if (!ok[g]) continue;
                        const int row = r0 + g * RPP + rsub;
                        const uint2 lo = *reinterpret_cast<const uint2*>(smem + row * C_LD + cpiece * 16);
                        const uint2 hi = *reinterpret_cast<const uint2*>(smem + row * C_LD + cpiece * 16 + 8);
                        float a[8], b[8];
                        unpack8(make_uint4(lo.x, lo.y, hi.x, hi.y), a);
                        unpack8(vr[g], b);
                        if (affine) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) a[j] += fmaf(b[j], rsc[j], rsh[j]);
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j) a[j] += b[j];
                        }
                        if constexpr (INFER) {
                            if (p.relu) {             // uniform: the block's ReLU behind the identity add
#pragma unroll
                                for (int j = 0; j < 8; ++j) a[j] = fmaxf(a[j], 0.f);
                            }
                        }
                        *reinterpret_cast<uint4*>(yout + (long)(m0 + row) * p.ldy + gcol) = pack8(a);
                    }
                }
            }
        } else {
            // ---- fused BN+ReLU backward prologue: dm = (dgrad [+ skip]) · mask, Σdm, Σdm·x̂ ----
            // restrict-qualified locals + a batch of GRP rows per round: all global loads of a round
            // are issued before the first is consumed (the store loop is otherwise one exposed memory
            // round trip per row)
            const bf16_raw* __restrict__ bnx = p.bnx;          // may be null (mask + Σdm only)
            const unsigned char* __restrict__ bnm = p.bnmask;
            const bf16_raw* __restrict__ bna = bnm ? nullptr : p.bna;
            const bf16_raw* __restrict__ resp = p.res;
            bf16_raw* __restrict__ yout = p.y;
            const bool col_ok = gcol < p.Cout;
            const int gcl = col_ok ? gcol : 0;
            const float msc = p.mask_scale;
            float mu[8], is[8], sc[8], sh[8], s1[8], s2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                mu[j] = col_ok && bnx ? p.bn_mean[gcol + j] : 0.f;
                is[j] = col_ok && bnx ? p.bn_rstd[gcol + j] : 0.f;
                sc[j] = (col_ok && p.bn_w ? p.bn_w[gcol + j] : 1.f) * is[j];
                sh[j] = fmaf(-mu[j], sc[j], col_ok && p.bn_b ? p.bn_b[gcol + j] : 0.f);
                s1[j] = 0.f; s2[j] = 0.f;
            }
            constexpr int NR = BM / RPP;
            constexpr int GRP = DP_BN_GRP(NR);
            for (int r0 = 0; r0 < BM; r0 += GRP * RPP) {
                uint4 vx[GRP], va[GRP], vr[GRP];
                unsigned vm[GRP];
                bool ok[GRP];
                // No per-row branch around a load (rows / columns beyond the tensor re-read its last row / first
                // columns; an absent operand is one uniform branch around its whole batch): with loads under
                // per-row branches hipcc cannot count what is outstanding and waits vmcnt(0) before every row — for
                // the previous row's STORE, one exposed write round trip per row.
                long off[GRP];
#pragma unroll
                for (int g = 0; g < GRP; ++g) {
                    const int m = m0 + r0 + g * RPP + rsub;
                    ok[g] = (m < p.M) && col_ok;
                    off[g] = (long)min(m, p.M - 1) * p.Cout + gcl;
                }
                if (bnx) {
#pragma unroll
                    for (int g = 0; g < GRP; ++g) vx[g] = *reinterpret_cast<const uint4*>(bnx + off[g]);
                } else {
#pragma unroll
                    for (int g = 0; g < GRP; ++g) vx[g] = make_uint4(0, 0, 0, 0);
                }
                if (bna) {
#pragma unroll
                    for (int g = 0; g < GRP; ++g) va[g] = *reinterpret_cast<const uint4*>(bna + off[g]);
                } else {
#pragma unroll
                    for (int g = 0; g < GRP; ++g) va[g] = make_uint4(0, 0, 0, 0);
                }
                if (bnm) {
#pragma unroll
                    for (int g = 0; g < GRP; ++g) vm[g] = bnm[off[g] >> 3];
                } else {
#pragma unroll
                    for (int g = 0; g < GRP; ++g) vm[g] = 0u;
                }
                if (resp) {
#pragma unroll
                    for (int g = 0; g < GRP; ++g)
                        vr[g] = *reinterpret_cast<const uint4*>(resp + (long)min(m0 + r0 + g * RPP + rsub, p.M - 1) * p.ldres + gcl);
                } else {
#pragma unroll
                    for (int g = 0; g < GRP; ++g) vr[g] = make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < GRP; ++g) {
                    if (!ok[g]) continue;
                    const int row = r0 + g * RPP + rsub;
                    const int m = m0 + row;
                    const uint2 lo = *reinterpret_cast<const uint2*>(smem + row * C_LD + cpiece * 16);
                    const uint2 hi = *reinterpret_cast<const uint2*>(smem + row * C_LD + cpiece * 16 + 8);
                    float a[8], xv[8], b[8];
                    unpack8(make_uint4(lo.x, lo.y, hi.x, hi.y), a);
                    unpack8(vx[g], xv);
                    unpack8(vr[g], b);
#pragma unroll
                    for (int j = 0; j < 8; ++j) a[j] += b[j];
                    if (bnm) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) a[j] = (vm[g] >> j) & 1u ? a[j] * msc : 0.f;
                    } else if (bna) {
                        float av[8];
                        unpack8(va[g], av);
#pragma unroll
                        for (int j = 0; j < 8; ++j) a[j] = av[j] > 0.f ? a[j] * msc : 0.f;
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) a[j] = fmaf(xv[j], sc[j], sh[j]) > 0.f ? a[j] * msc : 0.f;
                    }
                    const uint4 o = pack8(a);
                    unpack8(o, a);        // reduce exactly what BN backward will read back (bf16-rounded)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        s1[j] += a[j];
                        s2[j] = fmaf(a[j], (xv[j] - mu[j]) * is[j], s2[j]);
                    }
                    *reinterpret_cast<uint4*>(yout + (long)m * p.ldy + gcol) = o;
                }
            }
            __syncthreads();                              // every thread is done reading the C tile
            float* part = reinterpret_cast<float*>(smem); // [RPP][2][BN]
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                part[(rsub * 2 + 0) * BN + cpiece * 8 + j] = s1[j];
                part[(rsub * 2 + 1) * BN + cpiece * 8 + j] = s2[j];
            }
            __syncthreads();
            float* dst = p.bnsums + (long)(tile_m % p.bn_reps) * 2 * p.Cout;
            for (int i = tid; i < 2 * BN; i += T) {
                const int which = i / BN, c = i - which * BN;
                float v = 0.f;
#pragma unroll 4
                for (int r = 0; r < RPP; ++r) v += part[(r * 2 + which) * BN + c];
                if (n0 + c < p.Cout) atomicAdd(dst + which * p.Cout + n0 + c, v);
            }
        }
    }
#ifdef DP_STAMP
    DP_STAMP_AT(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DP_STAMP_AT(6);
    stamp_[8] = __builtin_amdgcn_s_memrealtime();
    if (p.dbg && tid == 0) {
#pragma unroll
        for (int i = 0; i < 11; ++i) p.dbg[(long)blockIdx.x * 16 + i] = stamp_[i];
    }
#endif
}

template <int BM, int BN, int WGM, int WGN, int NST, int MINW, int BK = 64>
int launch2(Conv2Args& p, bool unit, hipStream_t st) {
    constexpr int T = 64 * WGM * WGN;
    p.csteps = (p.Cin + BK - 1) / BK;
    p.ksteps = p.R * p.S * p.csteps;
    p.ctail = (p.Cin % BK) != 0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.Cout + BN - 1) / BN;
    const long nwg = (long)p.tiles_m * p.tiles_n;
    if (nwg <= 0 || nwg > 0x7fffffffL) return -4;
    constexpr int stage_bytes = NST * (BM + BN) * BK * 2;
    constexpr int epi_bytes = BM * (BN * 2 + 8) + WGM * 2 * BN * 4;
    constexpr int lds = stage_bytes > epi_bytes ? stage_bytes : epi_bytes;
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (p.w_img && p.img_px % BM != 0) return -2;          // a tile would straddle two images' weights
    if (p.oscale || p.relu) {
        if (unit) {
            auto kfn = conv_igemm2_kernel<BM, BN, WGM, WGN, NST, MINW, true, BK, true>;
            if (lds > 64 * 1024) {
                static bool once = false;
                if (!once) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds); once = true; }
            }
            hipLaunchKernelGGL(kfn, dim3((unsigned)nwg), dim3(T), lds, st, p);
        } else {
            auto kfn = conv_igemm2_kernel<BM, BN, WGM, WGN, NST, MINW, false, BK, true>;
            if (lds > 64 * 1024) {
                static bool once = false;
                if (!once) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds); once = true; }
            }
            hipLaunchKernelGGL(kfn, dim3((unsigned)nwg), dim3(T), lds, st, p);
        }
    } else if (unit) {
        auto kfn = conv_igemm2_kernel<BM, BN, WGM, WGN, NST, MINW, true, BK>;
        if (lds > 64 * 1024) {
            static bool once = false;
            if (!once) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds); once = true; }
        }
        hipLaunchKernelGGL(kfn, dim3((unsigned)nwg), dim3(T), lds, st, p);
    } else {
        auto kfn = conv_igemm2_kernel<BM, BN, WGM, WGN, NST, MINW, false, BK>;
        if (lds > 64 * 1024) {
            static bool once = false;
            if (!once) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds); once = true; }
        }
        hipLaunchKernelGGL(kfn, dim3((unsigned)nwg), dim3(T), lds, st, p);
    }
    return DP_LAUNCH_STATUS();
}

}  // namespace

// cfg: 1 = 128x128 4-wave 2-stage   3 = 256x128 8-wave 3-stage   4 = 128x256 8-wave 3-stage
//      5 = 256x256 8-wave 2-stage   7 = 128x64 4-wave 3-stage    8 = 160x128 / 9 = 192x128 4-wave 2-stage
//      (tile-count quantisation)    10 = 256x128 8-wave 2-stage  13 = 128x64 4-wave, 32-wide K steps, 4-stage ring (stem)
//      14 = 128x128 (32-wide K, 36 KiB LDS, 4 blocks/CU)  15 = 128x64 (48 KiB, 3 blocks/CU)  16 = 192x128 (32-wide K,
//      53 KiB, 3 blocks/CU)
//      (numbers 2, 6, 11, 12, 17, 18 were tiles no tuning run ever picked: removed in round 3)
//      small-LDS tiles for short-K layers whose time is the store loop (1x1 data gradients
//      with the fused BN-backward prologue read three tensors per output) — more resident blocks overlap one
//      block's K loop with another's stores
// Returns -2 for shapes this kernel does not take (R*S > 32, bias together with stats, Cout % 8):
// the caller then uses conv_igemm.hip.
DP_EXPORT int dp_conv_igemm2r(const void* x, const void* w, void* y, const float* bias, float* stats, int stat_reps,
                              int N, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int R, int S, int smul,
                              int sdiv, int pad_h, int pad_w, int dil_h, int dil_w, int ldy, int cfg,
                              const void* res, int ldres, hipStream_t st);

DP_EXPORT int dp_conv_igemm2(const void* x, const void* w, void* y, const float* bias, float* stats, int stat_reps,
                             int N, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int R, int S, int smul,
                             int sdiv, int pad_h, int pad_w, int dil_h, int dil_w, int ldy, int cfg, hipStream_t st) {
    return dp_conv_igemm2r(x, w, y, bias, stats, stat_reps, N, Hi, Wi, Cin, Ho, Wo, Cout, R, S, smul, sdiv, pad_h,
                           pad_w, dil_h, dil_w, ldy, cfg, nullptr, 0, st);
}

namespace {
// set by dp_conv_igemm2b for the launch it forwards to dp_conv_igemm2r (host-side, same thread)
thread_local const void* t_bnx = nullptr;
thread_local const void* t_bna = nullptr;
thread_local const void* t_bnmask = nullptr;
thread_local const float* t_bnp[4] = {nullptr, nullptr, nullptr, nullptr};
thread_local float* t_bnsums = nullptr;
thread_local int t_bn_reps = 0;
thread_local float t_mask_scale = 1.f;
thread_local int t_ldx = 0;      // set by dp_conv_igemm2w
thread_local const float* t_res_affine[2] = {nullptr, nullptr};   // set by dp_conv_igemm2ra
thread_local int t_w_img = 0;    // set by dp_conv_igemm2pi
thread_local const float* t_oscale = nullptr;   // set by dp_conv_igemm2e
thread_local int t_relu = 0;
int g_ablate = 0;
#ifdef DP_STAMP
unsigned long long* g_dbg = nullptr;
#endif
}  // namespace

DP_EXPORT int dp_conv_set_ablate(int mask) { g_ablate = mask; return 0; }
#ifdef DP_STAMP
DP_EXPORT int dp_conv_set_dbg(void* buf) { g_dbg = (unsigned long long*)buf; return 0; }
#endif

// + res: optional [N*Ho*Wo, ldres] bf16 tensor added to the output (not part of the fused statistics)
DP_EXPORT int dp_conv_igemm2r(const void* x, const void* w, void* y, const float* bias, float* stats, int stat_reps,
                              int N, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int R, int S, int smul,
                              int sdiv, int pad_h, int pad_w, int dil_h, int dil_w, int ldy, int cfg,
                              const void* res, int ldres, hipStream_t st) {
    if (res && (ldres < Cout || ldres % 8 != 0 || ((uintptr_t)res & 15) || stats)) return -1;
    if (Cin % 8 != 0 || Cin <= 0 || Cout <= 0 || N <= 0 || Ho <= 0 || Wo <= 0 || Hi <= 0 || Wi <= 0) return -1;
    if (ldy < Cout || R <= 0 || S <= 0 || smul < 1 || (sdiv != 1 && sdiv != 2 && sdiv != 4)) return -1;
    if (R * S > 32 || (bias && stats) || Cout % 8 != 0 || ldy % 8 != 0) return -2;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) return -3;
    const int ldx = t_ldx > 0 ? t_ldx : Cin;
    if (ldx > Cin || (ldx % 8 != 0 && ldx != 4)) return -1;
    if (ldx != Cin && (S != 1 || res || t_bnsums || sdiv != 1)) return -1;   // windows overlap along W only
    const long x_bytes = (long)N * Hi * Wi * ldx * 2, w_bytes = (long)Cout * R * S * Cin * 2;
    // 32-bit byte offsets below the out-of-range marker, with room for the largest tap offset
    if (x_bytes >= (1L << 31) - (1L << 24) || w_bytes >= (1L << 31) - (1L << 24) || (long)N * Ho * Wo >= (1L << 31) - 512)
        return -4;
    Conv2Args p{};
    p.x = (const bf16_raw*)x; p.w = (const bf16_raw*)w; p.y = (bf16_raw*)y; p.bias = bias; p.stats = stats;
    p.res = (const bf16_raw*)res; p.ldres = ldres;
    p.res_scale = res ? t_res_affine[0] : nullptr; p.res_shift = res ? t_res_affine[1] : nullptr;
    if (p.res_scale && t_bnsums) return -1;
    p.oscale = t_oscale; p.relu = t_relu;
    if ((p.oscale || p.relu) && (stats || t_bnsums || p.res_scale)) return -1;
    p.bnx = (const bf16_raw*)t_bnx; p.bna = (const bf16_raw*)t_bna; p.bnsums = t_bnsums;
    p.bnmask = (const unsigned char*)t_bnmask;
    p.bn_mean = t_bnp[0]; p.bn_rstd = t_bnp[1]; p.bn_w = t_bnp[2]; p.bn_b = t_bnp[3];
    p.bn_reps = t_bn_reps < 1 ? 1 : t_bn_reps;
    p.mask_scale = t_mask_scale;
    if (p.bnsums && (stats || ldy != Cout)) return -1;
    if (p.bnsums && !p.bnx && !p.bnmask && !p.bna) return -1;      // no pre-BN tensor: the mask needs a source
    p.stat_reps = stat_reps < 1 ? 1 : stat_reps;
    p.M = (int)((long)N * Ho * Wo);
    p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.R = R; p.S = S;
    p.ldx = ldx;
    p.ablate = g_ablate;
    p.w_img = t_w_img; p.img_px = Ho * Wo;
    if (p.w_img && (!(R == 1 && S == 1) || sdiv != 1 || smul != 1 || t_ldx > 0)) return -1;
#ifdef DP_STAMP
    p.dbg = g_dbg;
#endif
    p.smul = smul; p.sdl = sdiv == 1 ? 0 : (sdiv == 2 ? 1 : 2);
    p.pad_h = pad_h; p.pad_w = pad_w; p.dil_h = dil_h; p.dil_w = dil_w;
    p.ldy = ldy;
    p.x_bytes = (unsigned)x_bytes; p.w_bytes = (unsigned)w_bytes;
    const bool unit = (R == 1 && S == 1 && pad_h == 0 && pad_w == 0 && sdiv == 1 && ldx == Cin);
    switch (cfg) {
        case 1: return launch2<128, 128, 2, 2, 2, 2>(p, unit, st);
        case 3: return launch2<256, 128, 4, 2, 3, 2>(p, unit, st);
        case 4: return launch2<128, 256, 2, 4, 3, 2>(p, unit, st);
        case 5: return launch2<256, 256, 2, 4, 2, 2>(p, unit, st);
        case 7: return launch2<128, 64, 2, 2, 3, 1>(p, unit, st);
        case 8: return launch2<160, 128, 2, 2, 2, 2>(p, unit, st);
        case 9: return launch2<192, 128, 2, 2, 2, 2>(p, unit, st);
        case 10: return launch2<256, 128, 4, 2, 2, 2>(p, unit, st);
        case 13: return launch2<128, 64, 2, 2, 4, 1, 32>(p, unit, st);
        case 14: return launch2<128, 128, 2, 2, 2, 4, 32>(p, unit, st);
        case 15: return launch2<128, 64, 2, 2, 2, 3>(p, unit, st);
        case 16: return launch2<192, 128, 2, 2, 2, 3, 32>(p, unit, st);
        // (256x256 on FOUR waves — wave tile 128x128, accumulators in AGPRs, a third less fragment traffic — was
        // measured 2.4x SLOWER than case 5 on the K = 728 pointwise layers: 1150 vs 471 us, gpurun_out/n/shapes.log;
        // one wave per SIMD hides neither the LDS round trips nor the spills of its 512-register budget)
        default: return -2;
    }
}

// Data-gradient launch with the fused BN+ReLU backward prologue (see Conv2Args::bnx).  bnx / bna:
// [N*Ho*Wo, Cout] bf16 (bna may be null: mask recomputed from bnx), bn_mean / bn_rstd / bn_w / bn_b:
// fp32 [Cout] (bn_w, bn_b may be null), bnsums: fp32 [bn_reps, 2, Cout], zero-initialised.  bnmask (or null):
// positivity bits of a ([N*Ho*Wo, Cout/8] bytes, written by the BatchNorm apply kernel) — the ReLU mask then comes
// from them instead of from bna / a recomputation.
DP_EXPORT int dp_conv_igemm2b(const void* x, const void* w, void* y, int N, int Hi, int Wi, int Cin, int Ho, int Wo,
                              int Cout, int R, int S, int smul, int sdiv, int pad_h, int pad_w, int dil_h,
                              int dil_w, int cfg, const void* res, const void* bnx, const void* bna,
                              const void* bnmask, const float* bn_mean, const float* bn_rstd, const float* bn_w,
                              const float* bn_b, float* bnsums, int bn_reps, float mask_scale, hipStream_t st) {
    if (!bnsums || bn_reps < 1 || (bnx && (!bn_mean || !bn_rstd)) || !(mask_scale > 0.f)) return -1;
    if (!bnx && !bnmask && !bna) return -1;
    if (((uintptr_t)bnx | (uintptr_t)bna) & 15) return -3;
    t_bnmask = bnmask;
    t_bnx = bnx; t_bna = bna; t_bnsums = bnsums; t_bn_reps = bn_reps; t_mask_scale = mask_scale;
    t_bnp[0] = bn_mean; t_bnp[1] = bn_rstd; t_bnp[2] = bn_w; t_bnp[3] = bn_b;
    const int rc = dp_conv_igemm2r(x, w, y, nullptr, nullptr, 1, N, Hi, Wi, Cin, Ho, Wo, Cout, R, S, smul, sdiv,
                                   pad_h, pad_w, dil_h, dil_w, Cout, cfg, res, Cout, st);
    t_bnx = nullptr; t_bna = nullptr; t_bnsums = nullptr; t_bn_reps = 0; t_bnmask = nullptr; t_mask_scale = 1.f;
    t_bnp[0] = t_bnp[1] = t_bnp[2] = t_bnp[3] = nullptr;
    return rc;
}

// y = conv(x, w) + res * res_scale + res_shift (per output channel, fp32 [Cout] vectors) in the store loop
DP_EXPORT int dp_conv_igemm2ra(const void* x, const void* w, void* y, int N, int Hi, int Wi, int Cin, int Ho, int Wo,
                               int Cout, int R, int S, int smul, int sdiv, int pad_h, int pad_w, int dil_h,
                               int dil_w, int cfg, const void* res, const float* res_scale, const float* res_shift,
                               hipStream_t st) {
    if (!res || !res_scale || !res_shift) return -1;
    t_res_affine[0] = res_scale; t_res_affine[1] = res_shift;
    const int rc = dp_conv_igemm2r(x, w, y, nullptr, nullptr, 1, N, Hi, Wi, Cin, Ho, Wo, Cout, R, S, smul, sdiv,
                                   pad_h, pad_w, dil_h, dil_w, Cout, cfg, res, Cout, st);
    t_res_affine[0] = t_res_affine[1] = nullptr;
    return rc;
}

// Inference: y = act(conv(x, w) * oscale + oshift [+ res]) with act = ReLU when `relu` (oscale / oshift: fp32 [Cout] or
// null = 1 / 0; res: [N*Ho*Wo, ldres] bf16 or null).  The evaluation-mode BatchNorm + residual + ReLU of a block in the
// convolution's own epilogue.
DP_EXPORT int dp_conv_igemm2e(const void* x, const void* w, void* y, const float* oscale, const float* oshift, int relu,
                              int N, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int R, int S, int smul,
                              int sdiv, int pad_h, int pad_w, int dil_h, int dil_w, int ldy, int cfg,
                              const void* res, int ldres, hipStream_t st) {
    t_oscale = oscale; t_relu = relu ? 1 : 0;
    const int rc = dp_conv_igemm2r(x, w, y, oshift, nullptr, 1, N, Hi, Wi, Cin, Ho, Wo, Cout, R, S, smul, sdiv, pad_h,
                                   pad_w, dil_h, dil_w, ldy, cfg, res, ldres, st);
    t_oscale = nullptr; t_relu = 0;
    return rc;
}

// Forward convolution over OVERLAPPING pixel windows (the small-Cin network stem).  x: [N, Hi, Wi, ldx]
// with ldx = 4 (RGB / RGB+guidance padded to 4 channels, stored with the left padding physically
// present); a "pixel" of this launch is the Cin = 32-element window starting at a real pixel, i.e. a
// filter-row segment of 8 pixels, so a 7x7 filter is R = 7 taps of 32 "channels" (weights [Cout][R][1][32],
// zero where the window runs past the filter) instead of 49 taps of 8 zero-padded channels.  The last
// window of a row reads up to Cin - ldx elements past its pixel: the caller sizes Wi for it
// ((Wo-1)*smul + Cin/ldx <= Wi), so every read stays inside its own row.
DP_EXPORT int dp_conv_igemm2w(const void* x, const void* w, void* y, const float* bias, float* stats, int stat_reps,
                              int N, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int R, int S, int smul,
                              int sdiv, int pad_h, int pad_w, int dil_h, int dil_w, int ldy, int cfg, int ldx,
                              hipStream_t st) {
    if (ldx <= 0 || S != 1 || pad_w != 0 || dil_w != 1 || Cin % ldx != 0) return -1;
    if ((long)(Wo - 1) * smul + Cin / ldx > Wi || (Wi * ldx) % 8 != 0) return -1;   // last window ends inside its row
    t_ldx = ldx;
    const int rc = dp_conv_igemm2r(x, w, y, bias, stats, stat_reps, N, Hi, Wi, Cin, Ho, Wo, Cout, R, S, smul, sdiv,
                                   pad_h, pad_w, dil_h, dil_w, ldy, cfg, nullptr, 0, st);
    t_ldx = 0;
    return rc;
}

// A 1x1 convolution whose weights differ PER IMAGE (w: [N, Cout, Cin]; channel attention's attn · X) as ONE launch:
// tile rows never straddle images (Ho*Wo a multiple of the tile height, else -2), so a block picks its image's matrix.
DP_EXPORT int dp_conv_igemm2pi(const void* x, const void* w, void* y, int N, int Hi, int Wi, int Cin, int Cout, int cfg,
                               const void* res, hipStream_t st) {
    if ((long)N * Cout * Cin * 2 >= (1L << 31) - (1L << 24)) return -4;
    t_w_img = Cout * Cin;
    const int rc = dp_conv_igemm2r(x, w, y, nullptr, nullptr, 1, N, Hi, Wi, Cin, Hi, Wi, Cout, 1, 1, 1, 1, 0, 0, 1, 1,
                                   Cout, cfg, res, Cout, st);
    t_w_img = 0;
    return rc;
}
