// Fused BatchNorm kernels for NHWC bf16 activations on gfx950.
//
// All four kernels share ONE thread→data mapping: the activation is an [M, C] row-major matrix
// (M = N·H·W), viewed as 16-byte chunks of 8 channels.  A thread owns a fixed channel group
// (cg) for its whole life and walks rows with a block-wide stride, so
//   * every global access is a 16-B vector load/store, consecutive lanes → consecutive chunks
//     (full 1 KiB per wave-instruction when C ≥ 512, row-contiguous otherwise);
//   * per-channel constants (scale/shift, a/b/c of the backward) are computed once per thread and
//     live in registers — no per-element modulo, no constant-memory traffic;
//   * the reductions accumulate in fp32 registers, combine across the block through LDS and issue
//     one fp32 atomic per (block, channel): 256-B contiguous atomics per wave-instruction.
//
//   stats       x                    -> Σx, Σx²
//   apply       x,(stats|affine)     -> y = act(x·scale + shift (+res));  block 0 also finalises
//                                       mean / invstd / running statistics
//   bwd_reduce  dy, x (,y)           -> Σdy', Σdy'·x̂     (dy' = dy masked by the ReLU)
//   bwd_apply   dy, x (,y)           -> dx = a·dy' + b·x + c  (, d_residual = dy')
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxRowBlocks = 1024;
constexpr int kReduceSlab = 8;

struct Map {
    int cg;      // global channel-group index (8 channels)
    int rlane;   // row lane inside the block
    int rpp;     // rows per pass of the block
    int cgs;     // channel groups handled by this block
    bool active;
};

// SLAB = channel groups per block.  Streaming kernels use 256 (a block spans up to 2048 channels,
// rows fully contiguous).  Reduction kernels use 8 (a 64-channel = 128-byte slab per block, many
// rows per thread): the number of blocks that end up adding into one channel's accumulator is
// then the row-split count (tens), not the whole grid (a thousand) — contended fp32 atomics on
// one line run an order of magnitude below the spread rate on MI355X.
template <int SLAB>
__device__ __forceinline__ Map make_map(int C8) {
    Map m;
    const int base = blockIdx.y * SLAB;
    m.cgs = min(SLAB, C8 - base);
    m.rpp = kThreads / m.cgs;
    const int cgl = threadIdx.x % m.cgs;
    m.rlane = threadIdx.x / m.cgs;
    m.cg = base + cgl;
    m.active = m.rlane < m.rpp;
    return m;
}

__device__ __forceinline__ uint4 ldg16(const bf16_raw* p, long chunk) {
    return *reinterpret_cast<const uint4*>(p + chunk * 8);
}
__device__ __forceinline__ void stg16(bf16_raw* p, long chunk, const uint4& v) {
    *reinterpret_cast<uint4*>(p + chunk * 8) = v;
}

// Block-level combine of NACC per-thread accumulators over row lanes, then atomics.
// Fast path (full 8-group slab): lanes of a wave are [8 row lanes][8 channel groups], so three
// xor-shuffles (8, 16, 32) fold the row lanes in registers; the four waves then meet in 2 KB of LDS
// and 128 threads issue the atomics (one per channel and statistic).
template <int NACC>
__device__ __forceinline__ void block_reduce_atomic(const Map& m, float (&acc)[NACC], float* smem,
                                                    float* out0, float* out1, int C) {
    static_assert(NACC == 16, "two statistics x 8 channels");
    if (m.cgs == 8) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            float v = acc[j];
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[j] = v;
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane < 8) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) smem[(wave * 8 + lane) * NACC + j] = acc[j];
        }
        __syncthreads();
        if (threadIdx.x < 8 * NACC) {
            const int cgl = threadIdx.x / NACC, j = threadIdx.x % NACC;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < kThreads / 64; ++w) s += smem[(w * 8 + cgl) * NACC + j];
            const int c = (blockIdx.y * kReduceSlab + cgl) * 8 + (j & 7);
            atomicAdd((j < 8 ? out0 : out1) + c, s);
        }
        return;
    }
    // generic path (ragged last slab): smem layout [NACC][kThreads]; one thread per (statistic, channel)
    // folds the row lanes (the first version left all NACC x rpp additions to cgs threads: 672 serial LDS
    // reads per block for the decoder's 48-channel layer, 375 us for a 400 MB reduction)
#pragma unroll
    for (int j = 0; j < NACC; ++j) smem[j * kThreads + threadIdx.x] = m.active ? acc[j] : 0.f;
    __syncthreads();
    if ((int)threadIdx.x < NACC * m.cgs) {
        const int j = threadIdx.x / m.cgs, cgl = threadIdx.x - j * m.cgs;
        float s = 0.f;
        for (int rl = 0; rl < m.rpp; ++rl) s += smem[j * kThreads + rl * m.cgs + cgl];
        const int c = (blockIdx.y * kReduceSlab + cgl) * 8 + (j & 7);
        atomicAdd((j < 8 ? out0 : out1) + c, s);
    }
}

__device__ __forceinline__ void load8f(const float* p, float* f) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

__global__ __launch_bounds__(kThreads) void bn_stats_kernel(const bf16_raw* __restrict__ x, long M, int C,
                                                            float* __restrict__ sum, float* __restrict__ sq) {
    __shared__ float smem[16 * kThreads];
    const int C8 = C >> 3;
    const Map m = make_map<kReduceSlab>(C8);
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    if (m.active) {
        const long stride = (long)gridDim.x * m.rpp;
        long r = (long)blockIdx.x * m.rpp + m.rlane;
        // 4 independent 16-B loads in flight per thread
        for (; r + 3 * stride < M; r += 4 * stride) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ldg16(x, (r + u * stride) * C8 + m.cg);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[8];
                unpack8(v[u], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc[j] += f[j]; acc[8 + j] = fmaf(f[j], f[j], acc[8 + j]); }
            }
        }
        for (; r < M; r += stride) {
            float f[8];
            unpack8(ldg16(x, r * C8 + m.cg), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc[j] += f[j]; acc[8 + j] = fmaf(f[j], f[j], acc[8 + j]); }
        }
    }
    block_reduce_atomic<16>(m, acc, smem, sum, sq, C);
}

// Dropout fused into the BN+ReLU pass (the decoder's conv-BN-ReLU-Dropout): the keep decision of an
// element is a counter-based hash of (element index, seed), 16 bits per element, so the backward
// kernels RECOMPUTE it — no mask tensor, no separate dropout / masked-scale passes over the largest
// activations.  thr = round(p·65536) (0 = off), scale = 1 / (1 − thr/65536).
// (struct Drop, drop_hash, drop_keep8: common.h — the DANet head's dropout kernels use the same generator)

struct ApplyArgs {
    const bf16_raw* x; const bf16_raw* res; bf16_raw* y;
    long M; int C;
    // FROM_STATS: sum/sq/weight/bias (+ outputs mean/invstd, running stats); else scale/shift
    const float* sum; const float* sq; const float* weight; const float* bias;
    const float* scale; const float* shift;
    float* mean_out; float* invstd_out; float* running_mean; float* running_var;
    float inv_count, eps, momentum, unbias;
    int relu;
    int nrep;   // statistics replicas: sum/sq are [nrep][2][C] when produced by the conv epilogue
    long rep_stride;
    Drop drop;
    long y_ld8;   // output row stride in 16-byte chunks (C/8, or more when y is a channel slice of a
                  // wider concat buffer: the producer writes its slice in place, no cat kernel)
    unsigned char* mask;   // optional [M][C/8]: bit j of byte (row, chunk) = output element j is positive.  The
                           // consumer's data-gradient kernel masks with it instead of re-reading the whole
                           // activation (1/16 of the bytes; residual blocks, where the mask cannot be recomputed)
};

template <bool FROM_STATS>
__global__ __launch_bounds__(kThreads) void bn_apply_kernel(ApplyArgs a) {
    const int C8 = a.C >> 3;
    const Map m = make_map<kThreads>(C8);
    if (!m.active) return;
    float sc[8], sh[8];
    const int c0 = m.cg * 8;
    if (FROM_STATS) {
        float s1[8], s2[8], wv[8], bv[8];
        load8f(a.sum + c0, s1);
        load8f(a.sq + c0, s2);
        for (int r = 1; r < a.nrep; ++r) {
            float t1[8], t2[8];
            load8f(a.sum + r * a.rep_stride + c0, t1);
            load8f(a.sq + r * a.rep_stride + c0, t2);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1[j] += t1[j]; s2[j] += t2[j]; }
        }
        if (a.weight) load8f(a.weight + c0, wv);
        if (a.bias) load8f(a.bias + c0, bv);
        const bool finalise = blockIdx.x == 0 && m.rlane == 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float mean = s1[j] * a.inv_count;
            const float var = fmaxf(fmaf(-mean, mean, s2[j] * a.inv_count), 0.f);
            const float invstd = rsqrtf(var + a.eps);
            sc[j] = (a.weight ? wv[j] : 1.f) * invstd;
            sh[j] = fmaf(-mean, sc[j], a.bias ? bv[j] : 0.f);
            if (finalise) {
                a.mean_out[c0 + j] = mean;
                a.invstd_out[c0 + j] = invstd;
                if (a.running_mean) {
                    a.running_mean[c0 + j] = fmaf(a.momentum, mean - a.running_mean[c0 + j], a.running_mean[c0 + j]);
                    const float uv = var * a.unbias;
                    a.running_var[c0 + j] = fmaf(a.momentum, uv - a.running_var[c0 + j], a.running_var[c0 + j]);
                }
            }
        }
    } else {
        load8f(a.scale + c0, sc);
        load8f(a.shift + c0, sh);
    }
    const long stride = (long)gridDim.x * m.rpp;
    const bool has_res = a.res != nullptr;
    for (long r = (long)blockIdx.x * m.rpp + m.rlane; r < a.M; r += 2 * stride) {
        const long q0 = r * C8 + m.cg;
        const bool second = r + stride < a.M;
        const long q1 = second ? (r + stride) * C8 + m.cg : q0;
        uint4 v0 = ldg16(a.x, q0), v1 = ldg16(a.x, q1);
        uint4 r0 = v0, r1 = v1;
        if (has_res) { r0 = ldg16(a.res, q0); r1 = ldg16(a.res, q1); }
        float f0[8], f1[8], g0[8], g1[8];
        unpack8(v0, f0); unpack8(v1, f1);
        if (has_res) { unpack8(r0, g0); unpack8(r1, g1); }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float y0 = fmaf(f0[j], sc[j], sh[j]);
            float y1 = fmaf(f1[j], sc[j], sh[j]);
            if (has_res) { y0 += g0[j]; y1 += g1[j]; }
            if (a.relu) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
            f0[j] = y0; f1[j] = y1;
        }
        if (a.mask) {
            unsigned b0 = 0, b1 = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) { b0 |= (unsigned)(f0[j] > 0.f) << j; b1 |= (unsigned)(f1[j] > 0.f) << j; }
            a.mask[q0] = (unsigned char)b0;
            if (second) a.mask[q1] = (unsigned char)b1;
        }
        if (a.drop.thr) {
            const unsigned k0 = drop_keep8(q0, a.drop), k1 = drop_keep8(q1, a.drop);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f0[j] = (k0 >> j) & 1u ? f0[j] * a.drop.scale : 0.f;
                f1[j] = (k1 >> j) & 1u ? f1[j] * a.drop.scale : 0.f;
            }
        }
        stg16(a.y, r * a.y_ld8 + m.cg, pack8(f0));
        if (second) stg16(a.y, (r + stride) * a.y_ld8 + m.cg, pack8(f1));
    }
}

struct BwdArgs {
    const bf16_raw* dy; const bf16_raw* x; const bf16_raw* y;  // y: mask source (mask_mode 1)
    bf16_raw* dx; bf16_raw* dres;
    long M; int C;
    const float* mean; const float* invstd; const float* weight; const float* bias;
    const float* sdy; const float* sdyx;   // bwd_apply inputs (null => frozen statistics)
    float* out_sdy; float* out_sdyx;       // bwd_reduce outputs
    float inv_count;
    float* gw_sink; float* gb_sink;        // optional: += local Σdy'·x̂ / Σdy' (parameter-gradient arenas)
    const float* lsdy; const float* lsdyx; // sums feeding the sinks: local ones, or (SyncBN) the all-reduced
    float sink_scale;                      //   ones scaled by 1/world — the same value once DDP has averaged
    long dy_ld8;    // dy row stride in 16-byte chunks (C/8, or the concat buffer's when dy is a slice of it)
    int nrep;       // bwd_apply: sdy/sdyx (and lsdy/lsdyx) are nrep replicas, rep_stride floats apart — the
    long rep_stride;//   partial sums a data-gradient store loop left behind are folded here, no sum kernel
    int mask_mode;  // 0: no activation, 1: y > 0, 2: recompute x·scale+shift > 0, 3: as 2, and the fused
                    //    dropout's keep bit recomputed from the hash (dy' = keep·relu'·dy·scale)
    Drop drop;
};

__device__ __forceinline__ void masked_dy(const BwdArgs& a, int mode, const float* sc, const float* sh,
                                          const float* xf, const uint4& yv, float* dyf, long q) {
    if (mode == 1) {
        float yf[8];
        unpack8(yv, yf);
#pragma unroll
        for (int j = 0; j < 8; ++j) dyf[j] = yf[j] > 0.f ? dyf[j] : 0.f;
    } else if (mode == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dyf[j] = fmaf(xf[j], sc[j], sh[j]) > 0.f ? dyf[j] : 0.f;
    } else if (mode == 3) {
        const unsigned keep = drop_keep8(q, a.drop);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            dyf[j] = (fmaf(xf[j], sc[j], sh[j]) > 0.f && ((keep >> j) & 1u)) ? dyf[j] * a.drop.scale : 0.f;
    }
}

__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_kernel(BwdArgs a) {
    __shared__ float smem[16 * kThreads];
    const int C8 = a.C >> 3;
    const Map m = make_map<kReduceSlab>(C8);
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    if (m.active) {
        const int c0 = m.cg * 8;
        float mu[8], is[8], sc[8], sh[8], wv[8], bv[8];
        load8f(a.mean + c0, mu);
        load8f(a.invstd + c0, is);
        if (a.weight) load8f(a.weight + c0, wv);
        if (a.bias) load8f(a.bias + c0, bv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[j] = (a.weight ? wv[j] : 1.f) * is[j];
            sh[j] = fmaf(-mu[j], sc[j], a.bias ? bv[j] : 0.f);
        }
        const long stride = (long)gridDim.x * m.rpp;
        for (long r = (long)blockIdx.x * m.rpp + m.rlane; r < a.M; r += 2 * stride) {
            const long q0 = r * C8 + m.cg;
            const bool second = r + stride < a.M;
            const long q1 = second ? (r + stride) * C8 + m.cg : q0;
            uint4 d0 = ldg16(a.dy, r * a.dy_ld8 + m.cg), d1 = ldg16(a.dy, (second ? r + stride : r) * a.dy_ld8 + m.cg);
            uint4 x0 = ldg16(a.x, q0), x1 = ldg16(a.x, q1);
            uint4 y0 = x0, y1 = x1;
            if (a.mask_mode == 1) { y0 = ldg16(a.y, q0); y1 = ldg16(a.y, q1); }
            float df[8], xf[8];
            unpack8(d0, df); unpack8(x0, xf);
            masked_dy(a, a.mask_mode, sc, sh, xf, y0, df, q0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] += df[j];
                acc[8 + j] = fmaf(df[j], (xf[j] - mu[j]) * is[j], acc[8 + j]);
            }
            if (second) {
                unpack8(d1, df); unpack8(x1, xf);
                masked_dy(a, a.mask_mode, sc, sh, xf, y1, df, q1);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc[j] += df[j];
                    acc[8 + j] = fmaf(df[j], (xf[j] - mu[j]) * is[j], acc[8 + j]);
                }
            }
        }
    }
    block_reduce_atomic<16>(m, acc, smem, a.out_sdy, a.out_sdyx, a.C);
}

__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(BwdArgs a) {
    const int C8 = a.C >> 3;
    const Map m = make_map<kThreads>(C8);
    if (!m.active) return;
    const int c0 = m.cg * 8;
    float ka[8], kb[8], kc[8], sc[8], sh[8];
    {
        float mu[8], is[8], wv[8], bv[8], gs[8], gx[8];
        load8f(a.mean + c0, mu);
        load8f(a.invstd + c0, is);
        if (a.weight) load8f(a.weight + c0, wv);
        if (a.bias) load8f(a.bias + c0, bv);
        if (a.sdy) {
            load8f(a.sdy + c0, gs); load8f(a.sdyx + c0, gx);
            for (int r = 1; r < a.nrep; ++r) {
                float t1[8], t2[8];
                load8f(a.sdy + r * a.rep_stride + c0, t1);
                load8f(a.sdyx + r * a.rep_stride + c0, t2);
#pragma unroll
                for (int j = 0; j < 8; ++j) { gs[j] += t1[j]; gx[j] += t2[j]; }
            }
        }
        if (a.gw_sink && blockIdx.x == 0 && m.rlane == 0) {
            float lw[8], lb[8];
            load8f(a.lsdyx + c0, lw);
            load8f(a.lsdy + c0, lb);
            for (int r = 1; r < a.nrep; ++r) {
                float t1[8], t2[8];
                load8f(a.lsdyx + r * a.rep_stride + c0, t1);
                load8f(a.lsdy + r * a.rep_stride + c0, t2);
#pragma unroll
                for (int j = 0; j < 8; ++j) { lw[j] += t1[j]; lb[j] += t2[j]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a.gw_sink[c0 + j] += lw[j] * a.sink_scale;
                a.gb_sink[c0 + j] += lb[j] * a.sink_scale;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[j] = (a.weight ? wv[j] : 1.f) * is[j];
            sh[j] = fmaf(-mu[j], sc[j], a.bias ? bv[j] : 0.f);
            ka[j] = sc[j];
            if (a.sdy) {
                kb[j] = -sc[j] * is[j] * gx[j] * a.inv_count;
                kc[j] = -sc[j] * gs[j] * a.inv_count - kb[j] * mu[j];
            } else {
                kb[j] = 0.f; kc[j] = 0.f;
            }
        }
    }
    const long stride = (long)gridDim.x * m.rpp;
    for (long r = (long)blockIdx.x * m.rpp + m.rlane; r < a.M; r += 2 * stride) {
        const long q0 = r * C8 + m.cg;
        const bool second = r + stride < a.M;
        const long q1 = second ? (r + stride) * C8 + m.cg : q0;
        uint4 d0 = ldg16(a.dy, r * a.dy_ld8 + m.cg), d1 = ldg16(a.dy, (second ? r + stride : r) * a.dy_ld8 + m.cg);
        uint4 x0 = ldg16(a.x, q0), x1 = ldg16(a.x, q1);
        uint4 y0 = x0, y1 = x1;
        if (a.mask_mode == 1) { y0 = ldg16(a.y, q0); y1 = ldg16(a.y, q1); }
        float df[8], xf[8], o[8];
        unpack8(d0, df); unpack8(x0, xf);
        masked_dy(a, a.mask_mode, sc, sh, xf, y0, df, q0);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf(ka[j], df[j], fmaf(kb[j], xf[j], kc[j]));
        stg16(a.dx, q0, pack8(o));
        if (a.dres) stg16(a.dres, q0, pack8(df));
        if (second) {
            unpack8(d1, df); unpack8(x1, xf);
            masked_dy(a, a.mask_mode, sc, sh, xf, y1, df, q1);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = fmaf(ka[j], df[j], fmaf(kb[j], xf[j], kc[j]));
            stg16(a.dx, q1, pack8(o));
            if (a.dres) stg16(a.dres, q1, pack8(df));
        }
    }
}

inline dim3 grid_for(long M, int C, int rows_per_thread) {
    const int C8 = C / 8;
    const int cgs = C8 < kThreads ? C8 : kThreads;
    const int rpp = kThreads / cgs;
    long nb = (M + (long)rpp * rows_per_thread - 1) / ((long)rpp * rows_per_thread);
    if (nb < 1) nb = 1;
    if (nb > kMaxRowBlocks) nb = kMaxRowBlocks;
    return dim3((unsigned)nb, (unsigned)((C8 + kThreads - 1) / kThreads), 1);
}

// Streaming kernels: threads beyond rpp * cgs have no rows (make_map: rlane >= rpp).  With 91 channel
// groups (728 channels, Xception) that is 74 of 256 — launch only the waves that hold active threads
// (192), so the CU's wave slots go to blocks that move data.
inline dim3 block_for(int C) {
    const int C8 = C / 8;
    const int cgs = C8 < kThreads ? C8 : kThreads;
    const int used = (kThreads / cgs) * cgs;
    return dim3((unsigned)((used + 63) / 64 * 64));
}

// reduction kernels: grid = (row splits, 64-channel slabs); ~1024 blocks in total, ≥ 8 rows/thread
inline dim3 grid_reduce(long M, int C) {
    const int C8 = C / 8;
    const int slabs = (C8 + kReduceSlab - 1) / kReduceSlab;
    const int rpp = kThreads / kReduceSlab;
    long splits = (1024 + slabs - 1) / slabs;
    if (g_dp_deterministic) splits = 1;          // one block per channel slab: its atomics have no second writer
    const long max_by_rows = (M + (long)rpp * 8 - 1) / ((long)rpp * 8);
    if (splits > max_by_rows) splits = max_by_rows;
    if (splits < 1) splits = 1;
    return dim3((unsigned)splits, (unsigned)slabs, 1);
}

// ------------------------------------------------------------------------------------------------
// BatchNorm folded into the 1x1 convolution that consumes it (Xception's depthwise -> BN -> pointwise:
// no activation in between, so  W (s*z + t) = (W diag(s)) z + W t  and the normalised tensor never exists).
//   bn_fold      stats of z -> s = gamma*rstd, mean, rstd, running statistics; W' = W diag(s) (bf16);
//                b' = W beta - W' mean (the constant the next BatchNorm cancels; its running mean is corrected for it)
//   bn_fold_bwd  dW'(fp32, from the weight-gradient kernel run on raw z) -> dW += dW' diag(s),
//                ds_j = sum_k dW'_kj W_kj ; then c_j = -ds_j gamma_j rstd_j^3 / M (dz = W'^T dy + c (z - mean)),
//                dgamma_j += ds_j rstd_j
// ------------------------------------------------------------------------------------------------
struct FoldArgs {
    const float* stats; int nrep; long rep_stride;       // [nrep][2][Cin] partial sums of z, z^2
    const float* gamma; const float* beta;               // [Cin] (null: 1 / 0)
    float* rmean; float* rvar;                           // running statistics of the folded BN (or null)
    float inv_count, eps, momentum, unbias;
    const bf16_raw* w; bf16_raw* wf;                     // [Cout][Cin]
    float* s_out; float* mean_out; float* rstd_out;      // [Cin]
    float* bprime;                                       // [Cout] = W t
    float* rmean2; float rm2_scale;                      // next BN's running mean += rm2_scale * b' (or null)
    int Cout, Cin;
    bf16_raw* wfT;                                       // [Cin][Cout] or null: the folded weights transposed, for the
                                                         //   data gradient (saves a transposing copy per layer and step)
};

__global__ __launch_bounds__(256) void bn_fold_kernel(FoldArgs a) {
    extern __shared__ float fold_sh[];                   // s[Cin], beta[Cin], mean[Cin], red[4]
    float* s_sh = fold_sh;
    float* b_sh = fold_sh + a.Cin;
    float* m_sh = fold_sh + 2 * a.Cin;
    float* red = fold_sh + 3 * a.Cin;
    for (int j = threadIdx.x; j < a.Cin; j += 256) {
        float s1 = 0.f, s2 = 0.f;
        for (int r = 0; r < a.nrep; ++r) {
            s1 += a.stats[r * a.rep_stride + j];
            s2 += a.stats[r * a.rep_stride + a.Cin + j];
        }
        const float mean = s1 * a.inv_count;
        const float var = fmaxf(fmaf(-mean, mean, s2 * a.inv_count), 0.f);
        const float rstd = rsqrtf(var + a.eps);
        const float sc = (a.gamma ? a.gamma[j] : 1.f) * rstd;
        s_sh[j] = sc;
        b_sh[j] = a.beta ? a.beta[j] : 0.f;
        m_sh[j] = mean;
        if (blockIdx.x == 0) {
            a.s_out[j] = sc; a.mean_out[j] = mean; a.rstd_out[j] = rstd;
            if (a.rmean) {
                a.rmean[j] = fmaf(a.momentum, mean - a.rmean[j], a.rmean[j]);
                a.rvar[j] = fmaf(a.momentum, var * a.unbias - a.rvar[j], a.rvar[j]);
            }
        }
    }
    __syncthreads();
    for (int k = blockIdx.x; k < a.Cout; k += gridDim.x) {
        const bf16_raw* wr = a.w + (long)k * a.Cin;
        bf16_raw* wo = a.wf + (long)k * a.Cin;
        float acc = 0.f;
        for (int j8 = threadIdx.x * 8; j8 < a.Cin; j8 += 256 * 8) {
            // b' = sum_j W_kj beta_j - W'_kj mean_j with the ROUNDED W' the GEMM will use: mean(W' z) + b' is then
            // the exact mean of the unfolded layer's output (the running mean of the next BN relies on it)
            float f[8], g[8];
            unpack8(*reinterpret_cast<const uint4*>(wr + j8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc = fmaf(f[j], b_sh[j8 + j], acc);
                f[j] *= s_sh[j8 + j];
            }
            const uint4 packed = pack8(f);
            unpack8(packed, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = fmaf(-g[j], m_sh[j8 + j], acc);
            *reinterpret_cast<uint4*>(wo + j8) = packed;
            if (a.wfT) {
#pragma unroll
                for (int j = 0; j < 8; ++j) a.wfT[(long)(j8 + j) * a.Cout + k] = f2bf(g[j]);
            }
        }
        acc = wave_sum(acc);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float bp = red[0] + red[1] + red[2] + red[3];
            a.bprime[k] = bp;
            if (a.rmean2) a.rmean2[k] += a.rm2_scale * bp;
        }
    }
}

// 64 x 64 tile of the weight matrix per block: dW_sink += dW' * s (columns), ds[j] += sum_k dW'_kj W_kj
__global__ __launch_bounds__(256) void bn_fold_bwd_kernel(const float* __restrict__ dwf, const bf16_raw* __restrict__ w,
                                                          const float* __restrict__ s, float* __restrict__ dw_sink,
                                                          float* __restrict__ ds, int Cout, int Cin, float sink_scale) {
    __shared__ float part[4][64];
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    float acc = 0.f;
    if (j < Cin) {
        const float sj = s[j] * sink_scale;
        // (gridDim.y = 1 in deterministic mode: one block walks all row tiles of its 64 columns)
        for (int k0 = blockIdx.y * 64; k0 < Cout; k0 += gridDim.y * 64) {
            for (int k = k0 + rl; k < min(k0 + 64, Cout); k += 4) {
                const long idx = (long)k * Cin + j;
                const float g = dwf[idx];
                acc = fmaf(g, bf2f(w[idx]), acc);
                dw_sink[idx] += g * sj;
            }
        }
    }
    part[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && j < Cin) atomicAdd(ds + j, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// c_j = -ds_j gamma_j rstd_j^3 / M ; shift_j = -c_j mean_j ; dgamma_sink_j += ds_j rstd_j * sink_scale
__global__ void bn_fold_bwd_final_kernel(const float* __restrict__ ds, const float* __restrict__ gamma,
                                         const float* __restrict__ rstd, const float* __restrict__ mean,
                                         float* __restrict__ cvec, float* __restrict__ cshift,
                                         float* __restrict__ dgamma_sink, float inv_count, float sink_scale, int Cin) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Cin) return;
    const float g = gamma ? gamma[j] : 1.f, r = rstd[j];
    const float c = -ds[j] * g * r * r * r * inv_count;
    cvec[j] = c;
    cshift[j] = -c * mean[j];
    if (dgamma_sink) dgamma_sink[j] += ds[j] * r * sink_scale;
}

// ------------------------------------------------------------------------------------------------
// BatchNorm backward folded into the gradients of the 1x1 convolution in front of it (ops/bnfold.py: the tail of a
// ResNet bottleneck).  W: bf16 [K4][K1] weights, T = dm^T a (fp32 [K4][K1], from the weight-gradient kernel on the
// masked gradient), part: [reps][2][K4] partial sums whose first row is Σdm.
//   bn3fold_rows   sums[0][k] = Σdm_k,  sums[1][k] = Σ dm x̂ = rstd_k (Σ_j W_kj T_kj - mean_k Σdm_k)   (one wave per row)
//   bn3fold_coef   ka = g rstd, kb = -ka rstd Σ(dm x̂) / M, kc = -ka Σdm / M - kb mean; Wf = fp32(W), KBW = kb W,
//                  dgamma / dbeta sinks                                                                (one wave per row)
//   bn3fold_comb   dW_sink += ka T + kb WA + kc s^T ; Gb = bf16(G) ; c0 = W^T kc ; WpT = bf16(W^T diag(ka))
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn3fold_rows_kernel(const bf16_raw* __restrict__ W, const float* __restrict__ T,
                                                           const float* __restrict__ part, int reps,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           float* __restrict__ sums, int K4, int K1) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= K4) return;
    float dot = 0.f;
    for (int j = lane * 8; j < K1; j += 512) {
        float w[8];
        unpack8(*reinterpret_cast<const uint4*>(W + (long)k * K1 + j), w);
        const float4 t0 = *reinterpret_cast<const float4*>(T + (long)k * K1 + j);
        const float4 t1 = *reinterpret_cast<const float4*>(T + (long)k * K1 + j + 4);
        dot = fmaf(w[0], t0.x, dot); dot = fmaf(w[1], t0.y, dot); dot = fmaf(w[2], t0.z, dot); dot = fmaf(w[3], t0.w, dot);
        dot = fmaf(w[4], t1.x, dot); dot = fmaf(w[5], t1.y, dot); dot = fmaf(w[6], t1.z, dot); dot = fmaf(w[7], t1.w, dot);
    }
    dot = wave_sum(dot);
    if (lane == 0) {
        float sdy = 0.f;
        for (int r = 0; r < reps; ++r) sdy += part[(long)r * 2 * K4 + k];
        sums[k] = sdy;
        sums[K4 + k] = rstd[k] * (dot - mean[k] * sdy);
    }
}

__global__ __launch_bounds__(256) void bn3fold_coef_kernel(const bf16_raw* __restrict__ W, const float* __restrict__ sums,
                                                           const float* __restrict__ lsums,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, float inv_m, float sink_scale,
                                                           float* __restrict__ Wf, float* __restrict__ KBW,
                                                           float* __restrict__ kvec,
                                                           float* __restrict__ dgamma_sink, float* __restrict__ dbeta_sink,
                                                           int K4, int K1) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= K4) return;
    const float sdy = sums[k], sdyx = sums[K4 + k];
    const float ka = (gamma ? gamma[k] : 1.f) * rstd[k];
    const float kb = -ka * rstd[k] * sdyx * inv_m;
    const float kc = -ka * sdy * inv_m - kb * mean[k];
    if (lane == 0) {
        kvec[k] = ka; kvec[K4 + k] = kb; kvec[2 * K4 + k] = kc;
        // column K1 of the [K4][K1 + 8] matrix KBW holds kc: the GEMM G' = Wf^T KBW then delivers c0 = W^T kc as its
        // column K1 (a loop over k in one block took 200 us)
        *reinterpret_cast<float4*>(KBW + (long)k * (K1 + 8) + K1) = make_float4(kc, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(KBW + (long)k * (K1 + 8) + K1 + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        // affine gradients from the sums in `lsums` (local, or the all-reduced ones with sink_scale = 1 / world)
        if (dgamma_sink) dgamma_sink[k] += lsums[K4 + k] * sink_scale;
        if (dbeta_sink) dbeta_sink[k] += lsums[k] * sink_scale;
    }
    for (int j = lane * 8; j < K1; j += 512) {
        float w[8];
        unpack8(*reinterpret_cast<const uint4*>(W + (long)k * K1 + j), w);
        float4 a0 = make_float4(w[0], w[1], w[2], w[3]), a1 = make_float4(w[4], w[5], w[6], w[7]);
        *reinterpret_cast<float4*>(Wf + (long)k * K1 + j) = a0;
        *reinterpret_cast<float4*>(Wf + (long)k * K1 + j + 4) = a1;
        *reinterpret_cast<float4*>(KBW + (long)k * (K1 + 8) + j) = make_float4(kb * w[0], kb * w[1], kb * w[2], kb * w[3]);
        *reinterpret_cast<float4*>(KBW + (long)k * (K1 + 8) + j + 4) = make_float4(kb * w[4], kb * w[5], kb * w[6], kb * w[7]);
    }
}

__global__ __launch_bounds__(256) void bn3fold_comb_kernel(const float* __restrict__ T, const float* __restrict__ WA,
                                                           const float* __restrict__ s, const float* __restrict__ kvec,
                                                           const float* __restrict__ Wf, const float* __restrict__ G,
                                                           float* __restrict__ dw_sink, bf16_raw* __restrict__ Gb,
                                                           float* __restrict__ c0, const bf16_raw* __restrict__ WT,
                                                           bf16_raw* __restrict__ WpT, int K4, int K1, int blocks_dw,
                                                           int blocks_g, int blocks_t) {
    const int b = blockIdx.x;
    if (b < blocks_dw) {
        // dW[k][j] += ka_k T + kb_k WA + kc_k s_j : 4 consecutive elements per thread
        const long i4 = ((long)b * 256 + threadIdx.x) * 4;
        if (i4 < (long)K4 * K1) {
            const int k = (int)(i4 / K1), j = (int)(i4 - (long)k * K1);
            const float ka = kvec[k], kb = kvec[K4 + k], kc = kvec[2 * K4 + k];
            const float4 t = *reinterpret_cast<const float4*>(T + i4);
            const float4 wa = *reinterpret_cast<const float4*>(WA + i4);
            const float4 sv = *reinterpret_cast<const float4*>(s + j);
            float4 d = *reinterpret_cast<float4*>(dw_sink + i4);
            d.x += fmaf(ka, t.x, fmaf(kb, wa.x, kc * sv.x));
            d.y += fmaf(ka, t.y, fmaf(kb, wa.y, kc * sv.y));
            d.z += fmaf(ka, t.z, fmaf(kb, wa.z, kc * sv.z));
            d.w += fmaf(ka, t.w, fmaf(kb, wa.w, kc * sv.w));
            *reinterpret_cast<float4*>(dw_sink + i4) = d;
        }
    } else if (b < blocks_dw + blocks_g) {
        // G' is [K1][K1 + 8]: columns 0 .. K1-1 -> Gb (bf16), column K1 -> c0
        const long i = ((long)(b - blocks_dw) * 256 + threadIdx.x) * 2;
        if (i < (long)K1 * K1) {
            const int r = (int)(i / K1), c = (int)(i - (long)r * K1);
            const float2 g = *reinterpret_cast<const float2*>(G + (long)r * (K1 + 8) + c);
            *reinterpret_cast<unsigned*>(Gb + i) = pack2(g.x, g.y);
            if (c == 0) c0[r] = G[(long)r * (K1 + 8) + K1];
        }
    } else if (b < blocks_dw + blocks_g + blocks_t) {
        // WpT[j][k] = ka_k WT[j][k]: the data gradient's weights (W^T with its columns scaled), 8 elements per thread
        const long i8 = ((long)(b - blocks_dw - blocks_g) * 256 + threadIdx.x) * 8;
        if (i8 < (long)K1 * K4) {
            const int k = (int)(i8 % K4);
            float w[8];
            unpack8(*reinterpret_cast<const uint4*>(WT + i8), w);
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] *= kvec[k + e];
            *reinterpret_cast<uint4*>(WpT + i8) = pack8(w);
        }
    }
}

// set by dp_bn_set_dropout for the NEXT apply_stats / bwd_reduce / bwd_apply launch of this thread
thread_local Drop t_drop = {0u, 1.f, 0u, 0u, nullptr};

// row strides (in elements) of the NEXT launch's y (apply) / dy (backward): set by dp_bn_set_ld, 0 = C
thread_local long t_ld = 0;

// replica count / stride (floats) of the NEXT bwd_apply launch's sums: set by dp_bn_set_sum_replicas
thread_local int t_nrep = 1;
thread_local long t_rep_stride = 0;

inline long take_ld8(int C) {
    const long ld = t_ld > 0 ? t_ld : C;
    t_ld = 0;
    return ld / 8;
}

inline Drop take_drop() {
    Drop d = t_drop;
    t_drop = Drop{0u, 1.f, 0u, 0u, nullptr};
    d.word = d.thr ? g_dp_drop_word : nullptr;
    return d;
}

thread_local unsigned char* t_mask_out = nullptr;
inline unsigned char* take_mask() {
    unsigned char* m = t_mask_out;
    t_mask_out = nullptr;
    return m;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI.  Host-side shape checks live in ops/_hip_bn.py; here only C % 8 is re-checked because a
// violation would read out of bounds.
// ---------------------------------------------------------------------------------------------
DP_EXPORT int dp_bn_stats(const void* x, long M, int C, float* sum, float* sq, hipStream_t st) {
    if (C % 8 != 0 || M <= 0) return -1;
    hipLaunchKernelGGL(bn_stats_kernel, grid_reduce(M, C), dim3(kThreads), 0, st,
                       (const bf16_raw*)x, M, C, sum, sq);
    return DP_LAUNCH_STATUS();
}

// ld: row stride in elements (>= C, multiple of 8) of the output of the next apply launch or of the dy of
// the next backward launch of this thread — a channel slice of a wider NHWC buffer
DP_EXPORT int dp_bn_set_ld(long ld) {
    if (ld <= 0 || ld % 8 != 0) return -1;
    t_ld = ld;
    return 0;
}

// The next dp_bn_bwd_apply launch of this thread reads sdy / sdyx (and lsdy / lsdyx) as nrep partial sums,
// rep_stride floats apart, and folds them itself.
DP_EXPORT int dp_bn_set_sum_replicas(int nrep, long rep_stride) {
    if (nrep < 1 || nrep > 64 || rep_stride < 0 || rep_stride % 4 != 0) return -1;
    t_nrep = nrep;
    t_rep_stride = rep_stride;
    return 0;
}

// The next dp_bn_apply_stats launch of this thread also writes the positivity bit mask of its output
// ([M][C/8] bytes, ApplyArgs::mask).
DP_EXPORT int dp_bn_set_mask_out(void* mask) {
    t_mask_out = (unsigned char*)mask;
    return 0;
}

// thr16: round(p * 65536) in [1, 65535]; consumed (and cleared) by the next launch listed above
DP_EXPORT int dp_bn_set_dropout(unsigned thr16, unsigned long long seed) {
    if (thr16 == 0 || thr16 > 65535u) return -1;
    t_drop.thr = thr16;
    t_drop.scale = 65536.f / (float)(65536u - thr16);
    t_drop.seed_lo = (unsigned)seed;
    t_drop.seed_hi = (unsigned)(seed >> 32);
    return 0;
}

DP_EXPORT int dp_bn_apply_stats(const void* x, const void* res, void* y, long M, int C, const float* sum,
                                const float* sq, const float* weight, const float* bias, float* mean_out,
                                float* invstd_out, float* running_mean, float* running_var, float count,
                                float eps, float momentum, int relu, int nrep, long rep_stride, hipStream_t st) {
    // armed per-launch state (dp_bn_set_ld / dp_bn_set_dropout) is consumed even if this call is rejected
    const long ld8_ = take_ld8(C > 0 ? C : 8);
    const Drop drop_ = take_drop();
    unsigned char* mask_ = take_mask();
    if (C % 8 != 0 || M <= 0) return -1;
    ApplyArgs a{};
    a.mask = mask_;
    a.x = (const bf16_raw*)x; a.res = (const bf16_raw*)res; a.y = (bf16_raw*)y; a.M = M; a.C = C;
    a.sum = sum; a.sq = sq; a.weight = weight; a.bias = bias;
    a.mean_out = mean_out; a.invstd_out = invstd_out; a.running_mean = running_mean; a.running_var = running_var;
    a.inv_count = 1.f / count; a.eps = eps; a.momentum = momentum;
    a.unbias = count > 1.f ? count / (count - 1.f) : 1.f;
    a.relu = relu; a.nrep = nrep < 1 ? 1 : nrep; a.rep_stride = rep_stride;
    a.drop = drop_;
    a.y_ld8 = ld8_;
    if (a.y_ld8 * 8 < C || ((uintptr_t)y & 15)) return -1;
    hipLaunchKernelGGL(bn_apply_kernel<true>, grid_for(M, C, 4), block_for(C), 0, st, a);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_bn_apply_affine(const void* x, const void* res, void* y, long M, int C, const float* scale,
                                 const float* shift, int relu, hipStream_t st) {
    // armed per-launch state (dp_bn_set_ld / dp_bn_set_dropout) is consumed even if this call is rejected
    const long ld8_ = take_ld8(C > 0 ? C : 8);
    if (C % 8 != 0 || M <= 0) return -1;
    ApplyArgs a{};
    a.x = (const bf16_raw*)x; a.res = (const bf16_raw*)res; a.y = (bf16_raw*)y; a.M = M; a.C = C;
    a.scale = scale; a.shift = shift; a.relu = relu;
    a.y_ld8 = ld8_;
    if (a.y_ld8 * 8 < C || ((uintptr_t)y & 15)) return -1;
    hipLaunchKernelGGL(bn_apply_kernel<false>, grid_for(M, C, 4), block_for(C), 0, st, a);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_bn_bwd_reduce(const void* dy, const void* x, const void* y, long M, int C, const float* mean,
                               const float* invstd, const float* weight, const float* bias, float* out_sdy,
                               float* out_sdyx, int mask_mode, hipStream_t st) {
    // armed per-launch state (dp_bn_set_ld / dp_bn_set_dropout) is consumed even if this call is rejected
    const long ld8_ = take_ld8(C > 0 ? C : 8);
    const Drop drop_ = take_drop();
    if (C % 8 != 0 || M <= 0) return -1;
    BwdArgs a{};
    a.dy = (const bf16_raw*)dy; a.x = (const bf16_raw*)x; a.y = (const bf16_raw*)y; a.M = M; a.C = C;
    a.mean = mean; a.invstd = invstd; a.weight = weight; a.bias = bias;
    a.out_sdy = out_sdy; a.out_sdyx = out_sdyx; a.mask_mode = mask_mode;
    a.drop = drop_;
    a.dy_ld8 = ld8_;
    if ((mask_mode == 3 && a.drop.thr == 0) || a.dy_ld8 * 8 < C || ((uintptr_t)dy & 15)) return -1;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid_reduce(M, C), dim3(kThreads), 0, st, a);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_bn_bwd_apply(const void* dy, const void* x, const void* y, void* dx, void* dres, long M, int C,
                              const float* mean, const float* invstd, const float* weight, const float* bias,
                              const float* sdy, const float* sdyx, float count, int mask_mode, float* gw_sink,
                              float* gb_sink, const float* lsdy, const float* lsdyx, float sink_scale,
                              hipStream_t st) {
    // armed per-launch state (dp_bn_set_ld / dp_bn_set_dropout) is consumed even if this call is rejected
    const long ld8_ = take_ld8(C > 0 ? C : 8);
    const Drop drop_ = take_drop();
    const int nrep_ = t_nrep;
    const long rep_stride_ = t_rep_stride;
    t_nrep = 1; t_rep_stride = 0;
    if (C % 8 != 0 || M <= 0) return -1;
    if (nrep_ > 1 && (!sdy || rep_stride_ < 2L * C)) return -1;
    BwdArgs a{};
    a.nrep = nrep_; a.rep_stride = rep_stride_;
    a.dy = (const bf16_raw*)dy; a.x = (const bf16_raw*)x; a.y = (const bf16_raw*)y;
    a.dx = (bf16_raw*)dx; a.dres = (bf16_raw*)dres; a.M = M; a.C = C;
    a.mean = mean; a.invstd = invstd; a.weight = weight; a.bias = bias; a.sdy = sdy; a.sdyx = sdyx;
    a.inv_count = 1.f / count; a.mask_mode = mask_mode;
    a.gw_sink = gw_sink; a.gb_sink = gb_sink; a.lsdy = lsdy; a.lsdyx = lsdyx; a.sink_scale = sink_scale;
    a.drop = drop_;
    a.dy_ld8 = ld8_;
    if ((mask_mode == 3 && a.drop.thr == 0) || a.dy_ld8 * 8 < C || ((uintptr_t)dy & 15)) return -1;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, grid_for(M, C, 4), block_for(C), 0, st, a);
    return DP_LAUNCH_STATUS();
}

// Fold the BatchNorm described by `stats` (partial sums of z, z^2 over `count` rows) into the [Cout][Cin] bf16
// weights of the 1x1 convolution that consumes it.  s/mean/rstd: fp32 [Cin] outputs; bprime fp32 [Cout].
DP_EXPORT int dp_bn_fold(const float* stats, int nrep, long rep_stride, const float* gamma, const float* beta,
                         float* rmean, float* rvar, float count, float eps, float momentum, const void* w, void* wf,
                         float* s_out, float* mean_out, float* rstd_out, float* bprime, float* rmean2,
                         float rm2_scale, int Cout, int Cin, void* wfT, hipStream_t st) {
    if (!stats || !w || !wf || !s_out || !mean_out || !rstd_out || !bprime || nrep < 1 || Cin <= 0 || Cout <= 0 ||
        Cin % 8 != 0 || Cin > 8192 || count <= 0.f)
        return -1;
    if (((uintptr_t)w | (uintptr_t)wf) & 15) return -3;
    FoldArgs a{stats, nrep, rep_stride, gamma, beta, rmean, rvar, 1.f / count, eps, momentum,
               count > 1.f ? count / (count - 1.f) : 1.f, (const bf16_raw*)w, (bf16_raw*)wf, s_out, mean_out,
               rstd_out, bprime, rmean2, rm2_scale, Cout, Cin, (bf16_raw*)wfT};
    const int blocks = Cout < 512 ? Cout : 512;
    hipLaunchKernelGGL(bn_fold_kernel, dim3(blocks), dim3(256), (size_t)(3 * Cin + 8) * sizeof(float), st, a);
    return DP_LAUNCH_STATUS();
}

// dwf fp32 [Cout][Cin] (weight gradient w.r.t. the folded weights); dw_sink fp32 [Cout][Cin] accumulated;
// ds fp32 [Cin] zero-initialised scratch; cvec / cshift fp32 [Cin] outputs; dgamma_sink fp32 [Cin] accumulated or null
// stage: 0 = both kernels; 1 = the weight pass only (dw_sink, ds); 2 = the final pass only (ds -> cvec, cshift,
// dgamma_sink) — synchronised statistics all-reduce ds between the two.  `sink_scale` scales what goes into the
// sinks of THAT stage (the weight gradient is local, dgamma under SyncBN is the global sum / world).
DP_EXPORT int dp_bn_fold_bwd(const float* dwf, const void* w, const float* s, const float* gamma, const float* rstd,
                             const float* mean, float* dw_sink, float* ds, float* cvec, float* cshift,
                             float* dgamma_sink, float count, float sink_scale, int Cout, int Cin, int stage,
                             hipStream_t st) {
    if (!dwf || !w || !s || !rstd || !mean || !dw_sink || !ds || !cvec || !cshift || Cin <= 0 || Cout <= 0 || count <= 0.f ||
        stage < 0 || stage > 2)
        return -1;
    if (stage != 2)
        hipLaunchKernelGGL(bn_fold_bwd_kernel, dim3((Cin + 63) / 64, g_dp_deterministic ? 1 : (Cout + 63) / 64), dim3(256), 0, st,
                           dwf, (const bf16_raw*)w, s, dw_sink, ds, Cout, Cin, stage == 1 ? 1.f : sink_scale);
    if (stage != 1)
        hipLaunchKernelGGL(bn_fold_bwd_final_kernel, dim3((Cin + 255) / 256), dim3(256), 0, st, ds, gamma, rstd, mean, cvec,
                           cshift, dgamma_sink, 1.f / count, sink_scale, Cin);
    return DP_LAUNCH_STATUS();
}

// ops/bnfold.py, step 1: sums fp32 [2][K4] = (Σdm, Σ dm·x̂).  K1 % 8 == 0.
DP_EXPORT int dp_bn3fold_rows(const void* W, const float* T, const float* part, int reps, const float* mean,
                              const float* rstd, float* sums, int K4, int K1, hipStream_t st) {
    if (!W || !T || !part || !mean || !rstd || !sums || reps < 1 || K4 <= 0 || K1 <= 0 || K1 % 8 != 0) return -1;
    if (((uintptr_t)W | (uintptr_t)T) & 15) return -3;
    hipLaunchKernelGGL(bn3fold_rows_kernel, dim3((K4 + 3) / 4), dim3(256), 0, st, (const bf16_raw*)W, T, part, reps,
                       mean, rstd, sums, K4, K1);
    return DP_LAUNCH_STATUS();
}

// step 2: coefficients and operand matrices.  Wf fp32 [K4][K1]; KBW fp32 [K4][K1 + 8] = kb W | kc | 0;
// kvec fp32 [3][K4] = ka | kb | kc.
// lsums: the sums the affine gradients are formed from (== sums, or local / global as the caller decides).
DP_EXPORT int dp_bn3fold_coef(const void* W, const float* sums, const float* lsums, const float* mean,
                              const float* rstd, const float* gamma, float inv_m, float sink_scale, float* Wf,
                              float* KBW, float* kvec, float* dgamma_sink, float* dbeta_sink, int K4, int K1,
                              hipStream_t st) {
    if (!W || !sums || !lsums || !mean || !rstd || !Wf || !KBW || !kvec || K4 <= 0 || K1 <= 0 || K1 % 8 != 0)
        return -1;
    if (((uintptr_t)W | (uintptr_t)Wf | (uintptr_t)KBW) & 15) return -3;
    hipLaunchKernelGGL(bn3fold_coef_kernel, dim3((K4 + 3) / 4), dim3(256), 0, st, (const bf16_raw*)W, sums, lsums, mean,
                       rstd, gamma, inv_m, sink_scale, Wf, KBW, kvec, dgamma_sink, dbeta_sink, K4, K1);
    return DP_LAUNCH_STATUS();
}

// step 3 (after WA = Wf A and G' = Wf^T KBW, [K1][K1 + 8]): dw_sink += ka T + kb WA + kc s^T; Gb = bf16(G'[:, :K1]);
// c0 = G'[:, K1];
// WpT bf16 [K1][K4] = WT (the bf16 transposed weights [K1][K4]) with column k scaled by ka_k.  K4 % 8 == 0.
DP_EXPORT int dp_bn3fold_comb(const float* T, const float* WA, const float* s, const float* kvec, const float* Wf,
                              const float* G, float* dw_sink, void* Gb, float* c0, const void* WT, void* WpT, int K4,
                              int K1, hipStream_t st) {
    if (!T || !WA || !s || !kvec || !Wf || !G || !dw_sink || !Gb || !c0 || !WT || !WpT || K4 <= 0 || K1 <= 0 ||
        K1 % 8 != 0 || K4 % 8 != 0)
        return -1;
    if (((uintptr_t)T | (uintptr_t)WA | (uintptr_t)s | (uintptr_t)dw_sink | (uintptr_t)G | (uintptr_t)WT | (uintptr_t)WpT) & 15)
        return -3;
    const int bdw = (int)(((long)K4 * K1 / 4 + 255) / 256), bg = (int)(((long)K1 * K1 / 2 + 255) / 256);
    const int bt = (int)(((long)K1 * K4 / 8 + 255) / 256);
    hipLaunchKernelGGL(bn3fold_comb_kernel, dim3(bdw + bg + bt), dim3(256), 0, st, T, WA, s, kvec, Wf, G, dw_sink,
                       (bf16_raw*)Gb, c0, (const bf16_raw*)WT, (bf16_raw*)WpT, K4, K1, bdw, bg, bt);
    return DP_LAUNCH_STATUS();
}
