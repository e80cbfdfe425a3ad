// Fused bilinear upsample (align_corners=True) + ignore-index softmax cross-entropy (K6 + K7 fused).
//
// DeepLabv3+ ends in `loss = CE(upsample_x4(logits_129), labels_513)`.  Unfused, the 21-channel
// 513² logits map (354 MB per 32 images) is written by the upsample, read by the loss forward, read
// again and a same-sized dlogits written by the loss backward, and read once more by the upsample
// backward: ≈1.8 GB of HBM traffic and 2.1 ms per step (profiles/r1_native_v5_b32.md: bilinear_fwd
// 1.01 ms, bilinear_bwd 0.57 ms, softmax_ce_bwd 0.27 ms + the CE forward) for an op whose real
// inputs are 22 MB of low-resolution logits and 8 MB of labels.  Here the full-resolution logits
// never exist in memory:
//
//   forward   one thread per output pixel: the low-res footprint of an 8×32 output tile is staged in
//             LDS (fp32), each thread interpolates its C logits from 4 LDS taps and runs an online
//             softmax; Σloss and the valid count leave the block as one atomic pair.
//   backward  gather form, deterministic, no atomics: a block owns a 4×8 tile of LOW-res pixels,
//             phase 1 recomputes g = (softmax − onehot)·gloss/count for every output pixel whose
//             taps touch the tile (bf16 in LDS, the same rounding the unfused dlogits had),
//             phase 2 lets every (low-res pixel, class) sum its ≤ 7×7 weighted contributions.
//
// The low-res logits are addressed with a channel stride `ld` ≥ C so the classifier's padded
// (C=21 → 24) output is consumed in place; the gradient is written with `ldo` channels (pad = 0).
#include "common.h"

namespace {

constexpr int kT = 256;
constexpr int FT_H = 8, FT_W = 32;            // forward: output tile
constexpr int FF_H = FT_H + 1, FF_W = FT_W + 1;   // its low-res footprint for scale <= 1
constexpr int BT_H = 4, BT_W = 8;             // backward: low-res tile

struct UpCeArgs {
    const bf16_raw* x;       // low-res logits [N, Hi, Wi, ld]
    const void* target;      // [N, Ho, Wo]
    float* sums;             // [sum_reps][2]: Σloss, #valid (forward: block b adds into pair b % sum_reps)
    int sum_reps;
    const float* gloss;      // backward: upstream gradient (device scalar)
    bf16_raw* dx;            // backward: [N, Hi, Wi, ldo]
    int N, C, ld, ldo, Hi, Wi, Ho, Wo, ignore_index;
    float ry, rx;
    int tiles_y, tiles_x;    // per image
    int FH, FW;              // backward: footprint capacity (LDS carve)
};

template <typename LabelT>
__device__ __forceinline__ int ld_label(const void* p, long i) {
    return (int)reinterpret_cast<const LabelT*>(p)[i];
}

// CMAX: compile-time bound on C so a pixel's interpolated logits live in registers (one exp per
// class instead of the online softmax's two, one pass over the LDS taps); 0 = generic loop.
template <typename LabelT, int CMAX>
__global__ __launch_bounds__(kT) void upsample_ce_fwd_kernel(UpCeArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* tile = reinterpret_cast<float*>(smem_raw);           // [FF_H][FF_W][C]
    __shared__ float red[2 * (kT / kWave)];
    const int tid = threadIdx.x;
    const int C = p.C;
    const int ntiles = p.N * p.tiles_y * p.tiles_x;
    float loss = 0.f, cnt = 0.f;
    // persistent: a block walks tiles and leaves ONE atomic pair behind (35k same-address atomics
    // serialise into ~0.6 ms otherwise)
    for (int b0 = blockIdx.x; b0 < ntiles; b0 += gridDim.x) {
        int b = b0;
        const int tx = b % p.tiles_x; b /= p.tiles_x;
        const int ty = b % p.tiles_y;
        const int n = b / p.tiles_y;
        const int oy0 = ty * FT_H, ox0 = tx * FT_W;
        const int oy_last = min(oy0 + FT_H, p.Ho) - 1, ox_last = min(ox0 + FT_W, p.Wo) - 1;
        const int ylo = min((int)(oy0 * p.ry), p.Hi - 1), xlo = min((int)(ox0 * p.rx), p.Wi - 1);
        const int yhi = min(min((int)(oy_last * p.ry), p.Hi - 1) + 1, p.Hi - 1);
        const int xhi = min(min((int)(ox_last * p.rx), p.Wi - 1) + 1, p.Wi - 1);
        const int fh = min(yhi - ylo + 1, FF_H), fw = min(xhi - xlo + 1, FF_W);
        __syncthreads();                                         // previous tile's readers are done
        // stage the footprint: every row is one contiguous run of fw*ld elements in memory
        {
            const int run = (fw - 1) * p.ld + C;                 // elements of a row that matter
            for (int r = 0; r < fh; ++r) {
                const bf16_raw* src = p.x + (((long)n * p.Hi + ylo + r) * p.Wi + xlo) * p.ld;
                for (int e = tid; e < run; e += kT) {
                    const int xx = e / p.ld, c = e - xx * p.ld;
                    if (c < C) tile[(r * FF_W + xx) * C + c] = bf2f(src[e]);
                }
            }
        }
        __syncthreads();
        const int oy = oy0 + tid / FT_W, ox = ox0 + tid % FT_W;
        if (oy < p.Ho && ox < p.Wo) {
            const int t = ld_label<LabelT>(p.target, ((long)n * p.Ho + oy) * p.Wo + ox);
            if (t != p.ignore_index && t >= 0 && t < C) {
                const float sy = oy * p.ry, sx = ox * p.rx;
                const int y0 = min((int)sy, p.Hi - 1), x0 = min((int)sx, p.Wi - 1);
                const int y1 = min(y0 + 1, p.Hi - 1), x1 = min(x0 + 1, p.Wi - 1);
                const float fy = sy - y0, fx = sx - x0;
                const float w00 = (1.f - fy) * (1.f - fx), w01 = (1.f - fy) * fx, w10 = fy * (1.f - fx), w11 = fy * fx;
                const float* t00 = tile + (min(y0 - ylo, FF_H - 1) * FF_W + min(x0 - xlo, FF_W - 1)) * C;
                const float* t01 = tile + (min(y0 - ylo, FF_H - 1) * FF_W + min(x1 - xlo, FF_W - 1)) * C;
                const float* t10 = tile + (min(y1 - ylo, FF_H - 1) * FF_W + min(x0 - xlo, FF_W - 1)) * C;
                const float* t11 = tile + (min(y1 - ylo, FF_H - 1) * FF_W + min(x1 - xlo, FF_W - 1)) * C;
                if (CMAX > 0) {
                    float v[CMAX > 0 ? CMAX : 1];
                    float m = -3.0e38f, vt = 0.f;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) {
                        if (c < C) {
                            v[c] = w00 * t00[c] + w01 * t01[c] + w10 * t10[c] + w11 * t11[c];
                            m = fmaxf(m, v[c]);
                            if (c == t) vt = v[c];
                        }
                    }
                    float ssum = 0.f;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c)
                        if (c < C) ssum += __expf(v[c] - m);
                    loss += __logf(ssum) + m - vt;
                } else {
                    float m = -3.0e38f, ssum = 0.f, vt = 0.f;
                    for (int c = 0; c < C; ++c) {
                        const float v = w00 * t00[c] + w01 * t01[c] + w10 * t10[c] + w11 * t11[c];
                        if (c == t) vt = v;
                        const float mn = fmaxf(m, v);
                        ssum = ssum * __expf(m - mn) + __expf(v - mn);
                        m = mn;
                    }
                    loss += __logf(ssum) + m - vt;
                }
                cnt += 1.f;
            }
        }
    }
    loss = wave_sum(loss);
    cnt = wave_sum(cnt);
    const int wid = tid / kWave, lane = tid % kWave;
    if (lane == 0) { red[wid] = loss; red[kT / kWave + wid] = cnt; }
    __syncthreads();
    if (tid == 0) {
        float l = 0.f, c2 = 0.f;
        for (int w = 0; w < kT / kWave; ++w) { l += red[w]; c2 += red[kT / kWave + w]; }
        if (c2 > 0.f) {
            float* dst = p.sums + 2 * (blockIdx.x % (unsigned)p.sum_reps);
            atomicAdd(dst + 0, l);
            atomicAdd(dst + 1, c2);
        }
    }
}

// first / last output index whose source coordinate o·r lies strictly inside (i-1, i+1): the rows
// at the ends of the closed interval carry weight 0 (the 1e-3 slop only drops taps of weight < 1e-3)
__device__ __forceinline__ int fp_lo(int i, float inv_r) { return (int)floorf((i - 1) * inv_r + 1e-3f) + 1; }
__device__ __forceinline__ int fp_hi(int i, float inv_r) { return (int)ceilf((i + 1) * inv_r - 1e-3f) - 1; }

template <typename LabelT, int CMAX>
__global__ __launch_bounds__(kT) void upsample_ce_bwd_kernel(UpCeArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int C = p.C;
    float* lr = reinterpret_cast<float*>(smem_raw);                      // [BT_H+2][BT_W+2][C]
    int* ytab = reinterpret_cast<int*>(lr + (BT_H + 2) * (BT_W + 2) * C);  // [FH]: y0
    float* yfr = reinterpret_cast<float*>(ytab + p.FH);                  // [FH]: fy
    int* xtab = reinterpret_cast<int*>(yfr + p.FH);                      // [FW]
    float* xfr = reinterpret_cast<float*>(xtab + p.FW);                  // [FW]
    float* wyt = xfr + p.FW;                                             // [BT_H][FH]: weight of footprint row j for tile row r
    float* wxt = wyt + BT_H * p.FH;                                      // [BT_W][FW]
    bf16_raw* g = reinterpret_cast<bf16_raw*>(wxt + BT_W * p.FW);        // [FH][FW][C]
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int iy0 = ty * BT_H, ix0 = tx * BT_W;
    const int iy_last = min(iy0 + BT_H, p.Hi) - 1, ix_last = min(ix0 + BT_W, p.Wi) - 1;
    const float inv_ry = p.ry > 0.f ? 1.f / p.ry : 0.f, inv_rx = p.rx > 0.f ? 1.f / p.rx : 0.f;
    int oy_lo = 0, oy_hi = p.Ho - 1, ox_lo = 0, ox_hi = p.Wo - 1;
    if (p.ry > 0.f) { oy_lo = max(0, fp_lo(iy0, inv_ry)); oy_hi = min(p.Ho - 1, fp_hi(iy_last, inv_ry)); }
    if (p.rx > 0.f) { ox_lo = max(0, fp_lo(ix0, inv_rx)); ox_hi = min(p.Wo - 1, fp_hi(ix_last, inv_rx)); }
    const int fh = min(oy_hi - oy_lo + 1, p.FH), fw = min(ox_hi - ox_lo + 1, p.FW);
    const float k = p.gloss[0] / fmaxf(p.sums[1], 1.f);

    // low-res logits around the tile (clamped at the image border).  All of a thread's loads are issued before the
    // first LDS store (one flat index space, a fixed batch of independent loads): the row-by-row form was six
    // dependent global round trips per block in a kernel whose blocks live ~15 us.
    {
        const int run = (BT_W + 2) * C, total = (BT_H + 2) * run;
        constexpr int LB = 6;                             // loads in flight per thread and batch
        for (int e0 = tid; e0 < total; e0 += LB * kT) {
            bf16_raw v[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int e = min(e0 + u * kT, total - 1);
                const int r = e / run, ee = e - r * run;
                const int q = ee / C, c = ee - q * C;
                const int y = min(max(iy0 - 1 + r, 0), p.Hi - 1), x = min(max(ix0 - 1 + q, 0), p.Wi - 1);
                v[u] = p.x[(((long)n * p.Hi + y) * p.Wi + x) * p.ld + c];
            }
#pragma unroll
            for (int u = 0; u < LB; ++u)
                if (e0 + u * kT < total) lr[e0 + u * kT] = bf2f(v[u]);
        }
    }
    // tap tables and the per-(tile row, footprint row) / (tile col, footprint col) weights
    for (int j = tid; j < fh; j += kT) {
        const float sy = (oy_lo + j) * p.ry;
        const int y0 = min((int)sy, p.Hi - 1), y1 = min(y0 + 1, p.Hi - 1);
        const float fy = sy - y0;
        ytab[j] = y0; yfr[j] = fy;
#pragma unroll
        for (int r = 0; r < BT_H; ++r) {
            float w = 0.f;
            if (y0 == iy0 + r) w += 1.f - fy;
            if (y1 == iy0 + r) w += fy;
            wyt[r * p.FH + j] = w;
        }
    }
    for (int j = tid; j < fw; j += kT) {
        const float sx = (ox_lo + j) * p.rx;
        const int x0 = min((int)sx, p.Wi - 1), x1 = min(x0 + 1, p.Wi - 1);
        const float fx = sx - x0;
        xtab[j] = x0; xfr[j] = fx;
#pragma unroll
        for (int q = 0; q < BT_W; ++q) {
            float w = 0.f;
            if (x0 == ix0 + q) w += 1.f - fx;
            if (x1 == ix0 + q) w += fx;
            wxt[q * p.FW + j] = w;
        }
    }
    __syncthreads();

    // ---- phase 1: g for every output pixel of the footprint ----
    // (labels of a batch of LP pixels are loaded together, ahead of the softmax arithmetic)
    constexpr int LP = 4;
    for (int eb = tid; eb < fh * fw; eb += LP * kT) {
      int tl[LP];
#pragma unroll
      for (int u = 0; u < LP; ++u) {
          const int e = min(eb + u * kT, fh * fw - 1);
          const int j = e / fw, i = e - j * fw;
          tl[u] = ld_label<LabelT>(p.target, ((long)n * p.Ho + oy_lo + j) * p.Wo + ox_lo + i);
      }
#pragma unroll
      for (int u = 0; u < LP; ++u) {
        const int e = eb + u * kT;
        if (e >= fh * fw) break;
        const int j = e / fw, i = e - j * fw;
        bf16_raw* gp = g + (long)e * C;
        const int t = tl[u];
        if (t != p.ignore_index && t >= 0 && t < C) {
            const int y0 = ytab[j], x0 = xtab[i];
            const int y1 = min(y0 + 1, p.Hi - 1), x1 = min(x0 + 1, p.Wi - 1);
            const float fy = yfr[j], fx = xfr[i];
            const float w00 = (1.f - fy) * (1.f - fx), w01 = (1.f - fy) * fx, w10 = fy * (1.f - fx), w11 = fy * fx;
            const int ry0 = min(max(y0 - (iy0 - 1), 0), BT_H + 1), ry1 = min(max(y1 - (iy0 - 1), 0), BT_H + 1);
            const int rx0 = min(max(x0 - (ix0 - 1), 0), BT_W + 1), rx1 = min(max(x1 - (ix0 - 1), 0), BT_W + 1);
            const float* t00 = lr + (ry0 * (BT_W + 2) + rx0) * C;
            const float* t01 = lr + (ry0 * (BT_W + 2) + rx1) * C;
            const float* t10 = lr + (ry1 * (BT_W + 2) + rx0) * C;
            const float* t11 = lr + (ry1 * (BT_W + 2) + rx1) * C;
            if (CMAX > 0) {
                float v[CMAX > 0 ? CMAX : 1];
                float m = -3.0e38f;
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c < C) {
                        v[c] = w00 * t00[c] + w01 * t01[c] + w10 * t10[c] + w11 * t11[c];
                        m = fmaxf(m, v[c]);
                    }
                float ssum = 0.f;
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c < C) { v[c] = __expf(v[c] - m); ssum += v[c]; }
                const float inv = k / ssum;
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c < C) gp[c] = f2bf(v[c] * inv - (c == t ? k : 0.f));
            } else {
                float m = -3.0e38f, ssum = 0.f;
                for (int c = 0; c < C; ++c) {
                    const float v = w00 * t00[c] + w01 * t01[c] + w10 * t10[c] + w11 * t11[c];
                    const float mn = fmaxf(m, v);
                    ssum = ssum * __expf(m - mn) + __expf(v - mn);
                    m = mn;
                }
                const float inv = k / ssum;
                for (int c = 0; c < C; ++c) {
                    const float v = w00 * t00[c] + w01 * t01[c] + w10 * t10[c] + w11 * t11[c];
                    gp[c] = f2bf(__expf(v - m) * inv - (c == t ? k : 0.f));
                }
            }
        } else {
            for (int c = 0; c < C; ++c) gp[c] = 0;
        }
      }
    }
    __syncthreads();

    // ---- phase 2: every (low-res pixel, channel) gathers its contributions ----
    const int th = iy_last - iy0 + 1, tw = ix_last - ix0 + 1;
    for (int e = tid; e < th * tw * p.ldo; e += kT) {
        const int pix = e / p.ldo, c = e - pix * p.ldo;
        const int r = pix / tw, q = pix - r * tw;
        const int iy = iy0 + r, ix = ix0 + q;
        float acc = 0.f;
        if (c < C) {
            int jl = 0, jh = fh - 1, il = 0, ih = fw - 1;
            if (p.ry > 0.f) { jl = max(fp_lo(iy, inv_ry) - oy_lo, 0); jh = min(fp_hi(iy, inv_ry) - oy_lo, fh - 1); }
            if (p.rx > 0.f) { il = max(fp_lo(ix, inv_rx) - ox_lo, 0); ih = min(fp_hi(ix, inv_rx) - ox_lo, fw - 1); }
            const float* wy = wyt + r * p.FH;
            const float* wx = wxt + q * p.FW;
            for (int j = jl; j <= jh; ++j) {
                const bf16_raw* grow = g + ((long)j * fw) * C + c;
                float rowacc = 0.f;
                for (int i = il; i <= ih; ++i) rowacc = fmaf(wx[i], bf2f(grow[i * C]), rowacc);
                acc = fmaf(wy[j], rowacc, acc);
            }
        }
        p.dx[(((long)n * p.Hi + iy) * p.Wi + ix) * p.ldo + c] = f2bf(acc);
    }
}

// (Two pixel-centric forms of the backward for the exact scale-4 head were built and measured in round 3 — every
// thread one full-resolution pixel, contributions added into an LDS tile with ds_add_f32: 7.3 ms; the 4 x 4 pixels of a
// low-res cell reduced by DPP rotations inside a 16-lane row, no atomics: 3.85 ms — against 3.47 ms of the gather form
// above at B256.  Neither is kept; profiles/r3_loss_backward_scale4_negative.md has the kernels' description and times.)

inline bool fill_args(UpCeArgs& p, int N, int C, int ld, int Hi, int Wi, int Ho, int Wo, int ignore_index) {
    if (N <= 0 || C <= 0 || ld < C || Hi <= 0 || Wi <= 0 || Ho < Hi || Wo < Wi) return false;
    p.N = N; p.C = C; p.ld = ld; p.Hi = Hi; p.Wi = Wi; p.Ho = Ho; p.Wo = Wo; p.ignore_index = ignore_index;
    p.ry = Ho > 1 ? (float)(Hi - 1) / (Ho - 1) : 0.f;
    p.rx = Wo > 1 ? (float)(Wi - 1) / (Wo - 1) : 0.f;
    return true;
}

}  // namespace

// Returns -2 when the shape is outside what the fused kernels take (caller runs the unfused pair).
DP_EXPORT int dp_upsample_ce_fwd(const void* x, const void* target, int label_kind, int N, int C, int ld, int Hi,
                                 int Wi, int Ho, int Wo, int ignore_index, float* sums, int sum_reps, hipStream_t st) {
    UpCeArgs p{};
    if (sum_reps < 1) return -1;
    p.sum_reps = sum_reps;
    if (!fill_args(p, N, C, ld, Hi, Wi, Ho, Wo, ignore_index)) return -2;
    const size_t lds = (size_t)FF_H * FF_W * C * sizeof(float);
    if (lds > 60 * 1024) return -2;
    p.x = (const bf16_raw*)x; p.target = target; p.sums = sums;
    p.tiles_y = (Ho + FT_H - 1) / FT_H; p.tiles_x = (Wo + FT_W - 1) / FT_W;
    const long tiles = (long)N * p.tiles_y * p.tiles_x;
    if (tiles > 0x7fffffffL) return -2;
    const long grid = tiles < 2048 ? tiles : 2048;       // persistent: 8 blocks per CU
#define LAUNCH(T)                                                                                              \
    do {                                                                                                       \
        if (C <= 32) hipLaunchKernelGGL((upsample_ce_fwd_kernel<T, 32>), dim3((unsigned)grid), dim3(kT), lds, st, p); \
        else hipLaunchKernelGGL((upsample_ce_fwd_kernel<T, 0>), dim3((unsigned)grid), dim3(kT), lds, st, p);   \
    } while (0)
    if (label_kind == 0) LAUNCH(unsigned char);
    else if (label_kind == 1) LAUNCH(int);
    else LAUNCH(long long);
#undef LAUNCH
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_upsample_ce_bwd(const void* x, const void* target, int label_kind, int N, int C, int ld, int ldo,
                                 int Hi, int Wi, int Ho, int Wo, int ignore_index, const float* sums,
                                 const float* gloss, void* dx, hipStream_t st) {
    UpCeArgs p{};
    if (!fill_args(p, N, C, ld, Hi, Wi, Ho, Wo, ignore_index) || ldo < C) return -2;
    // footprint capacity: output pixels o with o·r strictly inside (i0-1, i_last+1)
    p.FH = p.ry > 0.f ? (int)((BT_H + 1) / p.ry) + 2 : Ho;
    p.FW = p.rx > 0.f ? (int)((BT_W + 1) / p.rx) + 2 : Wo;
    const size_t lds = (size_t)(BT_H + 2) * (BT_W + 2) * C * sizeof(float) + (size_t)(p.FH + p.FW) * 8 +
                       (size_t)(BT_H * p.FH + BT_W * p.FW) * sizeof(float) +
                       (size_t)p.FH * p.FW * C * sizeof(bf16_raw);
    if (lds > 64 * 1024) return -2;
    p.x = (const bf16_raw*)x; p.target = target; p.sums = const_cast<float*>(sums); p.gloss = gloss;
    p.dx = (bf16_raw*)dx; p.ldo = ldo;
    p.tiles_y = (Hi + BT_H - 1) / BT_H; p.tiles_x = (Wi + BT_W - 1) / BT_W;
    const long grid = (long)N * p.tiles_y * p.tiles_x;
    if (grid > 0x7fffffffL) return -2;
#define LAUNCH(T)                                                                                              \
    do {                                                                                                       \
        if (C <= 32) hipLaunchKernelGGL((upsample_ce_bwd_kernel<T, 32>), dim3((unsigned)grid), dim3(kT), lds, st, p); \
        else hipLaunchKernelGGL((upsample_ce_bwd_kernel<T, 0>), dim3((unsigned)grid), dim3(kT), lds, st, p);   \
    } while (0)
    if (label_kind == 0) LAUNCH(unsigned char);
    else if (label_kind == 1) LAUNCH(int);
    else LAUNCH(long long);
#undef LAUNCH
    return DP_LAUNCH_STATUS();
}
