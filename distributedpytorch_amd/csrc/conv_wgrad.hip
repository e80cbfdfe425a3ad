// Implicit-GEMM convolution weight gradient on MFMA for gfx950.
//
//   dW[k, (r,s,c)] = Σ_p dY[p, k] · X[pin(p, r, s), c]        p over batch·Ho·Wo output pixels
//
// GEMM view: M' = Cout, N' = R·S·Cin (the flattened KRSC weight row — tiles may straddle taps, so
// Cin = 8, 64 or 304 waste nothing), K' = pixels (very deep → split-K over pixel ranges).
//
// Both operands arrive "K-major": a pixel is a row and its channels are contiguous, but the MFMA
// fragments need 8 consecutive PIXELS per lane.  The tiles are therefore staged [pixel][channel]
// exactly as they lie in memory (global_load_lds_dwordx4, 16 B per lane) and read back with the
// CDNA4 transposing LDS read ds_read_b64_tr_b16 (4 pixels × 16 channels per 16-lane group, delivered
// pixel-major per lane) — no register shuffles, no pre-transposed copies of activations.
// LDS image: 256-B rows (128 channels), 16-B chunk index XOR-swizzled with
// f(row) = ((row&3)<<2)|((row>>2)&3) on the load SOURCE side and on the read side, which makes the
// transposed reads conflict-free.
//
// Split-K: each block owns (k-tile, n-tile, pixel range) and adds its fp32 128×128 partial tile into
// dW with no-return fp32 atomics; one accumulator register is two 128-B contiguous segments per
// wave-instruction, the shape that runs at the full atomic rate on MI355X.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int kThreads = 256;
constexpr int TM = 128;   // Cout tile
constexpr int TN = 128;   // flattened (r,s,c) tile
constexpr int TP = 64;    // pixels per K-step
constexpr int TILE_BYTES = TP * 256;        // one operand tile: 64 rows × 256 B
constexpr int STAGE = 2 * TILE_BYTES;

__device__ uint4 g_zero_page_w[16];

typedef __attribute__((ext_vector_type(4))) short s16x4;

struct WgradArgs {
    const bf16_raw* dy;   // [P, Cout]
    const bf16_raw* x;    // [N, Hi, Wi, Cin]
    float* dw;            // [Cout, R*S*Cin]
    int P;                // N*Ho*Wo
    int Hi, Wi, Cin, Ho, Wo, Cout, R, S;
    int stride_h, stride_w, pad_h, pad_w, dil_h, dil_w;
    int NF;               // R*S*Cin
    int tiles_k, tiles_n, splits, px_per_split;   // px_per_split multiple of TP
};

__device__ __forceinline__ void glds16w(const bf16_raw* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)src,
        (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int off0, int off1) {
    // two transposing reads = the 8 consecutive pixels of one channel this lane needs
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + off1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// (The first-generation kernel — global_load_lds with 64-bit addresses, 2-stage ring, no windows — was the route
// for tensors beyond the 32-bit offsets; ops._hip_conv cuts such batches into image chunks on the kernel below
// since round 2, nothing reached it any more: removed in round 3.)


// ------------------------------------------------------------------------------------------------
// Second generation (same tile geometry and fragment reads, new load path and pipeline).
// The ISA of the kernel above shows `s_waitcnt vmcnt(0)` between issue_stage() and the first
// transposing read of every K-step — hipcc cannot tell the LDS-DMA writes of stage t+1 from the
// reads of stage t — so its "prefetch" never overlapped anything, and every step paid ~100 VALU of
// pixel bookkeeping.  Here:
//   * operands arrive through buffer descriptors (`buffer_load_dwordx4 … lds`): the dy rows need NO
//     per-step VALU at all (fixed per-lane offset + a wave-uniform SGPR offset that advances by one
//     pixel block; the descriptor ends at the split's last pixel, so the tail rows read as zeros),
//     the x rows of 1×1/stride-1 convolutions likewise, and the general x rows keep (ox, oy, n)
//     incrementally and turn an invalid tap into an out-of-range offset;
//   * the wait for stage t is an explicit `s_waitcnt vmcnt(0)` placed BEFORE the barrier that opens
//     step t, the loads of step t+1 are issued right after that barrier and fly under the MFMAs;
//   * full tiles add their 64 accumulator registers with unconditional atomics (no 64 branches).
constexpr unsigned kOOBw = 0x80000000u;

struct Wgrad2Args {
    const bf16_raw* dy; const bf16_raw* x; float* dw;
    int P, Hi, Wi, Cin, Ho, Wo, Cout, R, S;
    int stride_h, stride_w, pad_h, pad_w, dil_h, dil_w;
    int NF, tiles_k, tiles_n, splits, px_per_split;
    unsigned x_bytes;
    int ldx;        // elements between neighbouring input pixels (Cin, or 4 for the overlapping stem windows)
#ifdef DP_STAMP
    unsigned long long* dbg;   // diagnostic build only (tools/stamp_conv.py --wgrad): phase stamps per block, stride 16
#endif
};

#ifdef DP_STAMP
#define WG_STAMP_DECL unsigned long long stamp_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}
#define WG_STAMP_AT(i) stamp_[i] = __builtin_amdgcn_s_memtime()
#else
#define WG_STAMP_DECL
#define WG_STAMP_AT(i)
#endif

__device__ __forceinline__ void lds_dma16w(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_wave_base, unsigned voff,
                                           unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff,
                                             soff, 0, 0);
}

// G = 2: two 128-channel dy sub-tiles share ONE x tile in a block of 8 waves (waves 0-3 own sub-tile 0,
// waves 4-7 sub-tile 1; each group is the 4-wave kernel unchanged, the x tile is loaded once by all 8).
// A K-step then moves 48 KiB for 256x128x64 MACs instead of 2 x 32 KiB: the kernel is bound by the
// L2->LDS stream (README, "Where the time goes"), and the weight gradients cost the critical path 26 ms
// of a 146 ms step even on their own stream (gpurun_out/s66: 120 ms without them).
//
// NST = 3 (G = 2 only: 3 x 48 KiB of the 160 KiB LDS, one block per CU): a three-deep ring.  With two stages
// one block has a single 48 KiB stage in flight while it computes — at one block per CU that is all the CU
// has outstanding, far below what the L2->LDS stream needs to cover its latency; the third stage keeps TWO
// stages (96 KiB) in flight across every barrier (counted vmcnt, the stage issued last stays outstanding).
// MF16: the products run on v_mfma_f32_16x16x32_bf16 (4 x 4 accumulator tiles of 16 x 16 per wave) instead of
// v_mfma_f32_32x32x16_bf16 (2 x 2 of 32 x 32): the same cycles per FLOP and the same number of transposed reads, but
// the chip holds a higher clock on the 16x16x32 shape under load (MI355X_MICROARCH, DVFS give-back item 7; the phase
// stamps read 1.4-1.9 GHz in this kernel against 1.75-2.1 GHz in the forward kernel, which uses 16x16x32).
template <bool UNIT, int G, int NST = 2, bool MF16 = false>
__global__ __launch_bounds__(kThreads * G, NST == 3 ? 1 : 2) void conv_wgrad2_kernel(Wgrad2Args p) {
    constexpr int XOFF = G * TILE_BYTES;                 // x tile offset inside a stage
    constexpr int STG = (G + 1) * TILE_BYTES;            // bytes per stage
    constexpr int NR = 4 / G;                            // x load rounds per thread (16*G rows each)
    constexpr int LPS = 4 + NR;                          // LDS-DMA instructions per thread per stage
    static_assert(NST == 2 || NST == 3, "ring depth");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    WG_STAMP_DECL;
#ifdef DP_STAMP
    stamp_[7] = __builtin_amdgcn_s_memrealtime();
#endif
    WG_STAMP_AT(0);

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tiles = p.tiles_k * p.tiles_n;
    const int split = bid / tiles;
    const int tile = bid - split * tiles;
    const int tile_k = tile % p.tiles_k, tile_n = tile / p.tiles_k;
    const int k0 = tile_k * (TM * G) + grp * TM, n0 = tile_n * TN;     // k0: this wave group's first channel
    const int p_begin = split * p.px_per_split;
    const int p_end = min(p.P, p_begin + p.px_per_split);
    const int nsteps = (p_end - p_begin + TP - 1) / TP;

    // descriptors: dy ends at this split's last pixel (tail rows of the last K-step read as zeros)
    const __amdgpu_buffer_rsrc_t rdy =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)((unsigned)p_end * (unsigned)p.Cout * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);

    const int pc = lane & 15;
    const int rsub = lane >> 4;
    const int fsw = (rsub << 2) | w4;                  // f(row): row & 3 = rsub, (row >> 2) & 3 = wave & 3
    const int lc = pc ^ fsw;
    const int row0 = w4 * 4 + rsub;                    // dy: + 16·i for load round i (per wave group)
    const int xrow0 = wave * 4 + rsub;                 // x:  + 16·G·i (all waves share the tile)

    // dy operand: channels k0 + lc*8 .. +7 of pixel (p_begin + row)
    const int kch = k0 + lc * 8;
    unsigned dy_voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        dy_voff[i] = kch < p.Cout ? (unsigned)(((p_begin + i * 16 + row0) * p.Cout + kch) * 2) : kOOBw;
    const unsigned dy_step = (unsigned)(TP * p.Cout * 2);

    // x operand: flattened weight-row index -> (tap, channel)
    const int flat = n0 + lc * 8;
    const bool n_ok = flat < p.NF;
    const int tap = flat / p.Cin;
    const int cch = flat - tap * p.Cin;
    const int tr = tap / p.S, ts = tap - tr * p.S;
    const int tdy = tr * p.dil_h - p.pad_h, tdx = ts * p.dil_w - p.pad_w;
    unsigned x_voff[NR];                                // UNIT: fixed offsets; general: byte offset of the
    int pox[NR], poy[NR];                               //   row's input pixel (oy·sh, ox·sw) before the tap
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int px = p_begin + i * (16 * G) + xrow0;
        if (UNIT) {
            x_voff[i] = n_ok ? (unsigned)((px * p.Cin + cch) * 2) : kOOBw;
        } else {
            pox[i] = px % p.Wo;
            const int t = px / p.Wo;
            poy[i] = t % p.Ho;
            x_voff[i] = (unsigned)(((((t / p.Ho) * p.Hi + poy[i] * p.stride_h) * p.Wi + pox[i] * p.stride_w) * p.ldx) * 2);
        }
    }
    const unsigned x_step = (unsigned)(TP * p.Cin * 2);
    const int adv_x = TP % p.Wo, adv_y = TP / p.Wo;
    // general path: the offset moves with the pixel by ADDS only (the PMC capture of the first version
    // showed 6.4 VALU per MFMA, most of it quarter-rate 32-bit multiplies rebuilding the address)
    const unsigned adv_bytes = (unsigned)((adv_y * p.stride_h * p.Wi + adv_x * p.stride_w) * p.ldx * 2);
    const unsigned wrapx_bytes = (unsigned)((p.stride_h * p.Wi - p.Wo * p.stride_w) * p.ldx * 2);
    const unsigned wrapy_bytes = (unsigned)((p.Hi - p.Ho * p.stride_h) * p.Wi * p.ldx * 2);
    const int tap_bytes = ((tdy * p.Wi + tdx) * p.ldx + cch) * 2;

    // The hardware range check looks at the per-lane offset only, so the running pixel offset is kept
    // in the VGPR offsets (one v_add per load and step) and the SGPR offset stays 0.  An out-of-range
    // lane starts at 2^31 and the tensors are < 2^31 - 2^24 bytes: it stays out of range, no wrap.
    auto issue_stage = [&](int buf) {
        unsigned char* dbase = smem + buf * STG + grp * TILE_BYTES + w4 * 1024;
        unsigned char* xbase = smem + buf * STG + XOFF + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lds_dma16w(rdy, dbase + i * 4096, dy_voff[i], 0);
            dy_voff[i] += dy_step;
        }
        if (UNIT) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                lds_dma16w(rx, xbase + i * (4096 * G), x_voff[i], 0);
                x_voff[i] += x_step;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int iy = poy[i] * p.stride_h + tdy, ix = pox[i] * p.stride_w + tdx;
                const bool ok = n_ok & ((unsigned)iy < (unsigned)p.Hi) & ((unsigned)ix < (unsigned)p.Wi);
                // pixels beyond P have n >= N: their offset lies beyond the descriptor and reads zeros
                lds_dma16w(rx, xbase + i * (4096 * G), ok ? x_voff[i] + (unsigned)tap_bytes : kOOBw, 0);
                pox[i] += adv_x; poy[i] += adv_y; x_voff[i] += adv_bytes;
                if (pox[i] >= p.Wo) { pox[i] -= p.Wo; poy[i] += 1; x_voff[i] += wrapx_bytes; }
                while (poy[i] >= p.Ho) { poy[i] -= p.Ho; x_voff[i] += wrapy_bytes; }
            }
        }
    };

    // the same stage in three parts (dy pieces 0-1, dy pieces 2-3, the x pieces), for the 16x16x32 loop below
    auto issue_dy2 = [&](int buf, int i0) {
        unsigned char* dbase = smem + buf * STG + grp * TILE_BYTES + w4 * 1024;
#pragma unroll
        for (int i = i0; i < i0 + 2; ++i) {
            lds_dma16w(rdy, dbase + i * 4096, dy_voff[i], 0);
            dy_voff[i] += dy_step;
        }
    };
    auto issue_x = [&](int buf) {
        unsigned char* xbase = smem + buf * STG + XOFF + wave * 1024;
        if (UNIT) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                lds_dma16w(rx, xbase + i * (4096 * G), x_voff[i], 0);
                x_voff[i] += x_step;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int iy = poy[i] * p.stride_h + tdy, ix = pox[i] * p.stride_w + tdx;
                const bool ok = n_ok & ((unsigned)iy < (unsigned)p.Hi) & ((unsigned)ix < (unsigned)p.Wi);
                lds_dma16w(rx, xbase + i * (4096 * G), ok ? x_voff[i] + (unsigned)tap_bytes : kOOBw, 0);
                pox[i] += adv_x; poy[i] += adv_y; x_voff[i] += adv_bytes;
                if (pox[i] >= p.Wo) { pox[i] -= p.Wo; poy[i] += 1; x_voff[i] += wrapx_bytes; }
                while (poy[i] >= p.Ho) { poy[i] -= p.Ho; x_voff[i] += wrapy_bytes; }
            }
        }
    };

    if constexpr (!MF16) {
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int wm = w4 >> 1, wn = w4 & 1;
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    int offA[2][2], offB[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int row = 8 * (g >> 1) + 4 * rd + q;
            const int f = (q << 2) | ((2 * (g >> 1) + rd) & 3);
            const int chA = ((wm * 64 + t * 32) >> 3) + 2 * (g & 1) + (pp >> 1);
            const int chB = ((wn * 64 + t * 32) >> 3) + 2 * (g & 1) + (pp >> 1);
            offA[t][rd] = row * 256 + ((chA ^ f) << 4) + (pp & 1) * 8;
            offB[t][rd] = row * 256 + ((chB ^ f) << 4) + (pp & 1) * 8;
        }
    }

    // Fragment reads are inline asm: hipcc cannot tell an LDS-DMA write from a later LDS read and puts
    // `s_waitcnt vmcnt(0)` before the first read it KNOWS of, which would drain the stage issued a few
    // instructions earlier.  The reads' own completion is tracked by hand (counted lgkmcnt, register
    // sets tied to the wait so the MFMAs cannot move above it).
    unsigned aA[2][2], aB[2][2];                        // LDS byte addresses (buffer 0)
    {
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                aA[t][rd] = base + (unsigned)(grp * TILE_BYTES) + (unsigned)offA[t][rd];
                aB[t][rd] = base + (unsigned)XOFF + (unsigned)offB[t][rd];
            }
    }
    s16x4 fa[2][2][2], fb[2][2][2];                     // [set][sub-tile][rd]
#define WG2_READS(SET, KK)                                                                                         \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                             \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fa[SET][i_][0]) : "v"(aA[i_][0]), "n"((KK) * 4096)); \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fa[SET][i_][1]) : "v"(aA[i_][1]), "n"((KK) * 4096)); \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fb[SET][i_][0]) : "v"(aB[i_][0]), "n"((KK) * 4096)); \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fb[SET][i_][1]) : "v"(aB[i_][1]), "n"((KK) * 4096)); \
    }
#define WG2_WAIT(SET, N)                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                       \
                 : "+v"(fa[SET][0][0]), "+v"(fa[SET][0][1]), "+v"(fa[SET][1][0]), "+v"(fa[SET][1][1]),             \
                   "+v"(fb[SET][0][0]), "+v"(fb[SET][0][1]), "+v"(fb[SET][1][0]), "+v"(fb[SET][1][1]))
#define WG2_MFMA(SET)                                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {           \
        typedef __attribute__((ext_vector_type(8))) short s16x8_;                                                  \
        const s16x8_ va = __builtin_shufflevector(fa[SET][i_][0], fa[SET][i_][1], 0, 1, 2, 3, 4, 5, 6, 7);        \
        const s16x8_ vb = __builtin_shufflevector(fb[SET][j_][0], fb[SET][j_][1], 0, 1, 2, 3, 4, 5, 6, 7);        \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, va),                      \
                                                              __builtin_bit_cast(bf16x8, vb), acc[i_][j_], 0, 0, 0); \
    }

    WG_STAMP_AT(1);
    if (nsteps > 0) issue_stage(0);
    if (NST == 3 && nsteps > 1) issue_stage(1);
    int cur = 0;                                        // ring slot of step t (uniform)
    for (int t = 0; t < nsteps; ++t) {
        // stage t landed (vmcnt) and my reads of the slot refilled below are done (every read is waited
        // for before the MFMAs that close a step), then everyone's
        if (NST == 3 && t + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef DP_STAMP
        if (t == 0) WG_STAMP_AT(2);
#endif
        if (NST == 3) {
            if (t + 2 < nsteps) issue_stage(cur == 0 ? 2 : cur - 1);      // (cur + 2) % 3: the slot read in step t-1
        } else {
            if (t + 1 < nsteps) issue_stage(cur ^ 1);
        }
        WG2_READS(0, 0)
        WG2_READS(1, 1)
        WG2_WAIT(0, 8);
        WG2_MFMA(0)
        WG2_READS(0, 2)
        WG2_WAIT(1, 8);
        WG2_MFMA(1)
        WG2_READS(1, 3)
        WG2_WAIT(0, 8);
        WG2_MFMA(0)
        WG2_WAIT(1, 0);
        WG2_MFMA(1)
        // move the fragment addresses to the next ring slot
        const bool wrap = cur == NST - 1;
        const unsigned step = wrap ? 0u - (unsigned)((NST - 1) * STG) : (unsigned)STG;
        cur = wrap ? 0 : cur + 1;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) { aA[a][b] += step; aB[a][b] += step; }
    }
#undef WG2_READS
#undef WG2_WAIT
#undef WG2_MFMA

    // ---- split-K combine ----
    WG_STAMP_AT(3);
    WG_STAMP_AT(4);
    const int frow = lane & 31, fhalf = lane >> 5;
    const bool full = (k0 + TM <= p.Cout) & (n0 + TN <= p.NF);     // uniform
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + frow;
            float* base = p.dw + (long)(k0 + wm * 64 + i * 32 + 4 * fhalf) * p.NF + col;
            if (full) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    atomicAdd(base + (long)((e & 3) + 8 * (e >> 2)) * p.NF, acc[i][j][e]);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = k0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fhalf;
                    if (row < p.Cout && col < p.NF) atomicAdd(p.dw + (long)row * p.NF + col, acc[i][j][e]);
                }
            }
        }
    }
    } else {
    // ---- 16x16x32 form ----
    typedef __attribute__((ext_vector_type(4))) float f32x4m;
    f32x4m acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4m{0.f, 0.f, 0.f, 0.f};
    const int wm = w4 >> 1, wn = w4 & 1;
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    // lane 4q + pp of 16-lane group g supplies pixel row 8g + 4 rd + q, channels 4 pp .. 4 pp + 3 of a 16-channel block:
    // after the two transposing reads lane t of the group holds pixels 8g .. 8g + 7 of channel t = the k-group g of
    // A[channel t][k] / B[k][channel t]
    unsigned aA[4][2], aB[4][2];
    {
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int row = 8 * g + 4 * rd + q;
                const int f = (q << 2) | ((2 * g + rd) & 3);
                const int chA = ((wm * 64 + t * 16) >> 3) + (pp >> 1);
                const int chB = ((wn * 64 + t * 16) >> 3) + (pp >> 1);
                aA[t][rd] = base + (unsigned)(grp * TILE_BYTES) + (unsigned)(row * 256 + ((chA ^ f) << 4) + (pp & 1) * 8);
                aB[t][rd] = base + (unsigned)XOFF + (unsigned)(row * 256 + ((chB ^ f) << 4) + (pp & 1) * 8);
            }
    }
    s16x4 fa[2][4][2], fb[2][4][2];                     // [set][16-channel block][rd]
#ifdef DP_WG_PRIO
#define W16_PRIO(P) __builtin_amdgcn_s_setprio(P)
#else
#define W16_PRIO(P)
#endif
#define W16_RD(DST, ADDR, KK) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"((KK) * 8192))
#define W16_READ_A(SET, KK, I0)                                                                                    \
    _Pragma("unroll") for (int i_ = (I0); i_ < (I0) + 2; ++i_) {                                                   \
        W16_RD(fa[SET][i_][0], aA[i_][0], KK); W16_RD(fa[SET][i_][1], aA[i_][1], KK); }
#define W16_READ_B(SET, KK)                                                                                        \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                             \
        W16_RD(fb[SET][j_][0], aB[j_][0], KK); W16_RD(fb[SET][j_][1], aB[j_][1], KK); }
    // the waits name exactly the registers whose reads they retire, so the MFMAs that consume them cannot move above
#define W16_WAIT_A01B(SET, N)                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                       \
                 : "+v"(fa[SET][0][0]), "+v"(fa[SET][0][1]), "+v"(fa[SET][1][0]), "+v"(fa[SET][1][1]),             \
                   "+v"(fb[SET][0][0]), "+v"(fb[SET][0][1]), "+v"(fb[SET][1][0]), "+v"(fb[SET][1][1]),             \
                   "+v"(fb[SET][2][0]), "+v"(fb[SET][2][1]), "+v"(fb[SET][3][0]), "+v"(fb[SET][3][1]))
#define W16_WAIT_A23(SET, N)                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                       \
                 : "+v"(fa[SET][2][0]), "+v"(fa[SET][2][1]), "+v"(fa[SET][3][0]), "+v"(fa[SET][3][1]))
#define W16_MFMA(SET, I0)                                                                                          \
    W16_PRIO(1);                                                                                                   \
    _Pragma("unroll") for (int i_ = (I0); i_ < (I0) + 2; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) { \
        typedef __attribute__((ext_vector_type(8))) short s16x8_;                                                  \
        const s16x8_ va = __builtin_shufflevector(fa[SET][i_][0], fa[SET][i_][1], 0, 1, 2, 3, 4, 5, 6, 7);        \
        const s16x8_ vb = __builtin_shufflevector(fb[SET][j_][0], fb[SET][j_][1], 0, 1, 2, 3, 4, 5, 6, 7);        \
        acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, va),                      \
                                                              __builtin_bit_cast(bf16x8, vb), acc[i_][j_], 0, 0, 0); \
    }                                                                                                              \
    W16_PRIO(0);
    WG_STAMP_AT(1);
    if (nsteps > 0) issue_stage(0);
    if (NST == 3 && nsteps > 1) issue_stage(1);
    int cur = 0;
    for (int t = 0; t < nsteps; ++t) {
        if (NST == 3 && t + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef DP_STAMP
        if (t == 0) WG_STAMP_AT(2);
#endif
        // The next stage's LDS-DMA pieces are issued in three groups BETWEEN the MFMA groups of the step, not as one
        // burst behind the barrier: eight waves issuing six 1-KiB pieces each at the same moment, next to their first
        // 16 fragment reads, fill the texture-address FIFOs (SQ_VMEM_TA_CMD/ADDR_FIFO_FULL) and each piece then holds
        // its wave for 100+ cycles (MI355X_MICROARCH: issue cost of an LDS-DMA piece by what the phase carries).
        const bool more3 = NST == 3 && t + 2 < nsteps;
        const int nslot = cur == 0 ? 2 : cur - 1;
        if (NST != 3 && t + 1 < nsteps) issue_stage(cur ^ 1);
        // pixels 0-31 of the step into set 0, 32-63 into set 1; at most 16 reads outstanding
        W16_READ_A(0, 0, 0)
        W16_READ_B(0, 0)
        W16_READ_A(0, 0, 2)
        W16_WAIT_A01B(0, 4);
        W16_MFMA(0, 0)
        if (more3) issue_dy2(nslot, 0);
        W16_READ_A(1, 1, 0)
        W16_READ_B(1, 1)
        W16_WAIT_A23(0, 12);
        W16_MFMA(0, 2)
        if (more3) issue_dy2(nslot, 2);
        W16_READ_A(1, 1, 2)
        W16_WAIT_A01B(1, 4);
        W16_MFMA(1, 0)
        if (more3) issue_x(nslot);
        W16_WAIT_A23(1, 0);
        W16_MFMA(1, 2)
        const bool wrap = cur == NST - 1;
        const unsigned step = wrap ? 0u - (unsigned)((NST - 1) * STG) : (unsigned)STG;
        cur = wrap ? 0 : cur + 1;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) { aA[a][b] += step; aB[a][b] += step; }
    }
#undef W16_RD
#undef W16_PRIO
#undef W16_READ_A
#undef W16_READ_B
#undef W16_WAIT_A01B
#undef W16_WAIT_A23
#undef W16_MFMA
    // ---- split-K combine: lane (g, c = lane & 15) holds D[4 g + r][c] of each 16 x 16 tile ----
    WG_STAMP_AT(3);
    WG_STAMP_AT(4);
    const int c16 = lane & 15;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int col = n0 + wn * 64 + b * 16 + c16;
            const int row0 = k0 + wm * 64 + a * 16 + 4 * g;
            if (col < p.NF) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row0 + r < p.Cout) atomicAdd(p.dw + (long)(row0 + r) * p.NF + col, acc[a][b][r]);
            }
        }
    }
    }
#ifdef DP_STAMP
    WG_STAMP_AT(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WG_STAMP_AT(6);
    stamp_[8] = __builtin_amdgcn_s_memrealtime();
    if (p.dbg && tid == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) p.dbg[(long)blockIdx.x * 16 + i] = stamp_[i];
        p.dbg[(long)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_getreg(63492);
        p.dbg[(long)blockIdx.x * 16 + 10] = __builtin_amdgcn_s_getreg(63508);
        p.dbg[(long)blockIdx.x * 16 + 11] = (unsigned long long)nsteps;
    }
#endif
}

}  // namespace

#ifdef DP_STAMP
namespace { unsigned long long* g_wg_dbg = nullptr; }
DP_EXPORT int dp_wgrad_set_dbg(void* buf) { g_wg_dbg = (unsigned long long*)buf; return 0; }
#endif

// splits: 0 = auto.  dw must be zero-initialised (or hold a gradient to accumulate into).
namespace {
thread_local int t_wg_ldx = 0;     // set by dp_conv_wgrad_win
}

DP_EXPORT int dp_conv_wgrad(const void* dy, const void* x, void* dw, int N, int Hi, int Wi, int Cin, int Ho, int Wo,
                            int Cout, int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int dil_h,
                            int dil_w, int splits, int reserved, hipStream_t st) {
    const int ldx = t_wg_ldx > 0 ? t_wg_ldx : Cin;
    if (Cin % 8 != 0 || Cout % 8 != 0 || Cin <= 0 || Cout <= 0 || N <= 0 || Ho <= 0 || Wo <= 0) return -1;
    if (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dw) & 15) return -3;
    const long P = (long)N * Ho * Wo;
    if (P >= (1L << 31) / (Cout > Cin ? Cout : Cin) || (long)N * Hi * Wi * ldx >= (1L << 31)) return -4;
    WgradArgs p{};
    p.dy = (const bf16_raw*)dy; p.x = (const bf16_raw*)x; p.dw = (float*)dw;
    p.P = (int)P; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.R = R; p.S = S;
    p.stride_h = stride_h; p.stride_w = stride_w; p.pad_h = pad_h; p.pad_w = pad_w; p.dil_h = dil_h; p.dil_w = dil_w;
    p.NF = R * S * Cin;
    // two wave groups per block (256 output channels, shared x tile) when the channels fill them
    const long x_bytes0 = (long)N * Hi * Wi * ldx * 2;
    (void)reserved;
    // 32-bit offsets of the LDS-DMA loads: the caller cuts larger batches into image chunks (ops._hip_conv.wgrad)
    if (x_bytes0 >= (1L << 31) - (1L << 24) || P * Cout * 2 >= (1L << 31) - (1L << 24)) return -4;
    // G = 2 also for channel counts that do not fill the last 256-channel block (Xception's 728 = 2.84 x 256): the
    // kernel's loads and its combine are bounds-checked per 128-channel sub-tile, and what the ragged block wastes
    // (768 vs 728 rows) the shared x tile and the 3-deep ring pay back many times — the 4-wave G = 1 kernel ran the
    // 728 -> 728 pointwise layers at 232 TFLOP/s (profiles/r2_x65_os8_b64_v28_profile.md).  Not when the padding to
    // 256 would add more than 1/8 to the rows the 128-blocks cover.
    const int pad256 = (Cout + 2 * TM - 1) / (2 * TM) * (2 * TM), pad128 = (Cout + TM - 1) / TM * TM;
    const bool g2_shape = Cout > TM && pad256 * 8 <= pad128 * 9;
    const int G = g2_shape ? 2 : 1;
    p.tiles_k = (Cout + TM * G - 1) / (TM * G);
    p.tiles_n = (p.NF + TN - 1) / TN;
    const int tiles = p.tiles_k * p.tiles_n;
    const int slots = G == 2 ? 256 : 512;          // resident blocks chip-wide (96 KiB vs 64 KiB of LDS)
    const int steps_total = (int)((P + TP - 1) / TP);
    if (splits <= 0) {
        // Cost model (µs), calibrated on MI355X: a K-step of one block costs ~0.5 µs (LDS-DMA bound,
        // ~1 µs when two blocks share a CU), a block has ~3 µs of fixed prologue/epilogue, and the
        // split-K combine moves splits·|dW| bytes through fp32 atomics at ~1.3 TB/s chip-wide
        // (MI355X_MICROARCH: global float atomics) — the term the first heuristic ignored: it asked
        // for 1536 blocks and spent 70 % of a layer-3 wgrad in atomics.
        const double dw_mb = (double)Cout * p.NF * 4.0 / 1e6;
        double best = 1e30;
        int best_s = 1;
        const int max_s = steps_total < 512 ? steps_total : 512;
        for (int s = 1; s <= max_s; ++s) {
            const int steps = (steps_total + s - 1) / s;
            const int real_s = (steps_total + steps - 1) / steps;
            const long blocks = (long)tiles * real_s;
            const long rounds = (blocks + slots - 1) / slots;
            const double share = G == 2 ? 1.0
                                        : (blocks >= 512 ? 1.0 : (blocks > 256 ? 0.5 + 0.5 * (blocks - 256) / 256.0 : 0.5));
            const double t = rounds * (steps * share + 3.0) + (real_s > 1 ? real_s * dw_mb / 1.3 : 0.0);
            if (t < best) { best = t; best_s = real_s; }
        }
        splits = best_s;
    }
    int steps_per = (steps_total + splits - 1) / splits;
    p.px_per_split = steps_per * TP;
    p.splits = (int)((P + p.px_per_split - 1) / p.px_per_split);
    const long nwg = (long)tiles * p.splits;
    if (nwg > 0x7fffffffL) return -4;
    const long x_bytes = (long)N * Hi * Wi * ldx * 2;
    Wgrad2Args q{};
    q.dy = p.dy; q.x = p.x; q.dw = p.dw; q.P = p.P; q.Hi = Hi; q.Wi = Wi; q.Cin = Cin; q.Ho = Ho; q.Wo = Wo;
    q.Cout = Cout; q.R = R; q.S = S; q.stride_h = stride_h; q.stride_w = stride_w; q.pad_h = pad_h; q.pad_w = pad_w;
    q.dil_h = dil_h; q.dil_w = dil_w; q.NF = p.NF; q.tiles_k = p.tiles_k; q.tiles_n = p.tiles_n; q.splits = p.splits;
    q.px_per_split = p.px_per_split; q.x_bytes = (unsigned)x_bytes; q.ldx = ldx;
#ifdef DP_STAMP
    q.dbg = g_wg_dbg;
#endif
    const bool unit = R == 1 && S == 1 && pad_h == 0 && pad_w == 0 && stride_h == 1 && stride_w == 1 && ldx == Cin;
    // Two wave groups: three-stage ring, 16x16x32 MFMA.  (Measured against the 32x32x16 shape on one box, R101 B128:
    // 952.8 / 950.5 vs 946.6 / 947.1 img/s — alone the kernels tie, the chip holds a higher clock on 16x16x32 — and
    // against the two-stage ring: -4...-6 % K loop.  Both alternatives were removed in round 3.)
    if (G == 2) {
        constexpr int lds3 = 3 * 3 * TILE_BYTES;      // 144 KiB: three stages, one block per CU
        static bool once16 = false;
        if (!once16) {
            (void)hipFuncSetAttribute((const void*)conv_wgrad2_kernel<true, 2, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3);
            (void)hipFuncSetAttribute((const void*)conv_wgrad2_kernel<false, 2, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3);
            once16 = true;
        }
        if (unit) hipLaunchKernelGGL((conv_wgrad2_kernel<true, 2, 3, true>), dim3((unsigned)nwg), dim3(2 * kThreads), lds3, st, q);
        else hipLaunchKernelGGL((conv_wgrad2_kernel<false, 2, 3, true>), dim3((unsigned)nwg), dim3(2 * kThreads), lds3, st, q);
        return DP_LAUNCH_STATUS();
    }
    if (unit) hipLaunchKernelGGL((conv_wgrad2_kernel<true, 1>), dim3((unsigned)nwg), dim3(kThreads), 2 * STAGE, st, q);
    else hipLaunchKernelGGL((conv_wgrad2_kernel<false, 1>), dim3((unsigned)nwg), dim3(kThreads), 2 * STAGE, st, q);
    return DP_LAUNCH_STATUS();
}

// Weight gradient over overlapping pixel windows (see dp_conv_igemm2w): x is [N, Hi, Wi, ldx], a tap's
// "channels" are the Cin elements starting at its pixel.  dw: fp32 [Cout][R][1][Cin].
DP_EXPORT int dp_conv_wgrad_win(const void* dy, const void* x, void* dw, int N, int Hi, int Wi, int Cin, int ldx,
                                int Ho, int Wo, int Cout, int R, int stride, int pad_h, int splits, hipStream_t st) {
    if (ldx <= 0 || ldx > Cin || Cin % ldx != 0 || (ldx % 8 != 0 && ldx != 4)) return -1;
    if ((long)(Wo - 1) * stride + Cin / ldx > Wi || (Wi * ldx) % 8 != 0) return -1;
    t_wg_ldx = ldx;
    const int rc = dp_conv_wgrad(dy, x, dw, N, Hi, Wi, Cin, Ho, Wo, Cout, R, 1, stride, stride, pad_h, 0, 1, 1, splits,
                                 0, st);
    t_wg_ldx = 0;
    return rc;
}
