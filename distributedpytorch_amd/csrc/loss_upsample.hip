// Ignore-index softmax cross-entropy (K7) and bilinear resize, align_corners=True (K6), for NHWC
// bf16 tensors on gfx950.
//
// softmax-CE: logits are an [M, C] row-major matrix with C small (21) so a row is 42 bytes — not a
// vector-load-friendly stride.  A 256-thread block therefore stages 256 consecutive rows (a
// contiguous, 16-B-aligned slab of 256·C·2 bytes) through LDS with coalesced 16-B loads, then each
// thread owns one pixel and walks its row in LDS (stride C·2 B → banks spread).  Forward produces
// Σloss and the valid-pixel count (one atomic pair per block); backward recomputes the softmax
// from the logits (cheaper than materialising dlogits in forward: 3 passes over the logits in
// total instead of 4, and no activation memory held between forward and backward) and writes
// dlogits = (p - onehot) · gloss / count, staged back through LDS for coalesced stores.
//
// bilinear: forward is a 4-tap gather; backward is scatter-free — every INPUT pixel gathers the
// output pixels whose footprint covers it (window ≤ ⌈2/r⌉ per axis) so there are no atomics and
// the result is deterministic.  Channels are vectorised 8-wide when C % 8 == 0; the logits map
// (C = 21) takes the scalar path.
#include "common.h"

namespace {

constexpr int kCeThreads = 256;

template <typename LabelT>
__device__ __forceinline__ int load_label(const void* p, long i) {
    return (int)reinterpret_cast<const LabelT*>(p)[i];
}

__device__ __forceinline__ void stage_rows_in(const bf16_raw* __restrict__ g, bf16_raw* lds, long row0, long M,
                                              int C) {
    // slab of up to 256 rows: [row0, row0+256) ∩ [0, M); base is 16-B aligned because 256·C·2 is
    const long rows = min((long)kCeThreads, M - row0);
    const long elems = rows * C;
    const bf16_raw* src = g + row0 * C;
    const long vec = elems >> 3;
    for (long i = threadIdx.x; i < vec; i += kCeThreads)
        reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(src)[i];
    for (long i = (vec << 3) + threadIdx.x; i < elems; i += kCeThreads) lds[i] = src[i];
}

__device__ __forceinline__ void stage_rows_out(bf16_raw* __restrict__ g, const bf16_raw* lds, long row0, long M,
                                               int C) {
    const long rows = min((long)kCeThreads, M - row0);
    const long elems = rows * C;
    bf16_raw* dst = g + row0 * C;
    const long vec = elems >> 3;
    for (long i = threadIdx.x; i < vec; i += kCeThreads)
        reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(lds)[i];
    for (long i = (vec << 3) + threadIdx.x; i < elems; i += kCeThreads) dst[i] = lds[i];
}

// out[0] += Σ loss, out[1] += #valid; out is [reps][2] and block b adds into pair b % reps
template <typename LabelT>
__global__ __launch_bounds__(kCeThreads) void softmax_ce_fwd_kernel(const bf16_raw* __restrict__ logits,
                                                                    const void* __restrict__ target, long M, int C,
                                                                    int ignore_index, float* __restrict__ out, int reps) {
    out += 2 * (blockIdx.x % (unsigned)reps);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_raw* tile = reinterpret_cast<bf16_raw*>(smem_raw);
    float loss = 0.f, cnt = 0.f;
    for (long row0 = (long)blockIdx.x * kCeThreads; row0 < M; row0 += (long)gridDim.x * kCeThreads) {
        __syncthreads();
        stage_rows_in(logits, tile, row0, M, C);
        __syncthreads();
        const long r = row0 + threadIdx.x;
        if (r < M) {
            const int t = load_label<LabelT>(target, r);
            if (t != ignore_index && t >= 0 && t < C) {
                const bf16_raw* row = tile + threadIdx.x * C;
                float mx = -3.0e38f;
                for (int c = 0; c < C; ++c) mx = fmaxf(mx, bf2f(row[c]));
                float s = 0.f;
                for (int c = 0; c < C; ++c) s += __expf(bf2f(row[c]) - mx);
                loss += __logf(s) + mx - bf2f(row[t]);
                cnt += 1.f;
            }
        }
    }
    loss = wave_sum(loss);
    cnt = wave_sum(cnt);
    // reduction scratch lives in the same dynamic LDS block (keeps the carve base 16-B aligned)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_raw);
    const int wid = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    if (lane == 0) { red[wid] = loss; red[8 + wid] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float l = 0.f, c = 0.f;
        for (int w = 0; w < kCeThreads / kWave; ++w) { l += red[w]; c += red[8 + w]; }
        atomicAdd(out + 0, l);
        atomicAdd(out + 1, c);
    }
}

// dlogits = (softmax - onehot) * gloss / max(count, 1); ignored rows -> 0
template <typename LabelT>
__global__ __launch_bounds__(kCeThreads) void softmax_ce_bwd_kernel(const bf16_raw* __restrict__ logits,
                                                                    const void* __restrict__ target, long M, int C,
                                                                    int ignore_index, const float* __restrict__ sums,
                                                                    const float* __restrict__ gloss,
                                                                    bf16_raw* __restrict__ dlogits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_raw* tile = reinterpret_cast<bf16_raw*>(smem_raw);
    const float k = gloss[0] / fmaxf(sums[1], 1.f);
    for (long row0 = (long)blockIdx.x * kCeThreads; row0 < M; row0 += (long)gridDim.x * kCeThreads) {
        __syncthreads();
        stage_rows_in(logits, tile, row0, M, C);
        __syncthreads();
        const long r = row0 + threadIdx.x;
        if (r < M) {
            bf16_raw* row = tile + threadIdx.x * C;
            const int t = load_label<LabelT>(target, r);
            if (t != ignore_index && t >= 0 && t < C) {
                float mx = -3.0e38f;
                for (int c = 0; c < C; ++c) mx = fmaxf(mx, bf2f(row[c]));
                float s = 0.f;
                for (int c = 0; c < C; ++c) s += __expf(bf2f(row[c]) - mx);
                const float inv = k / s;
                for (int c = 0; c < C; ++c) {
                    float p = __expf(bf2f(row[c]) - mx) * inv;
                    if (c == t) p -= k;
                    row[c] = f2bf(p);
                }
            } else {
                for (int c = 0; c < C; ++c) row[c] = 0;
            }
        }
        __syncthreads();
        stage_rows_out(dlogits, tile, row0, M, C);
    }
}

// ------------------------------- bilinear, align_corners=True -------------------------------
struct Resize {
    int N, C, Hi, Wi, Ho, Wo;
    float ry, rx;  // input step per output pixel: (Hi-1)/(Ho-1)
    int ldo;       // channel stride of the HIGH-resolution tensor (forward's y, backward's dy): C, or the
                   // row width of the concat buffer it is a slice of
};

template <int VEC>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const bf16_raw* __restrict__ x, bf16_raw* __restrict__ y,
                                                           Resize p) {
    const int CV = p.C / VEC;
    const long total = (long)p.N * p.Ho * p.Wo * CV;
    // index arithmetic in 32 bits whenever the element count allows: the three 64-bit div/mod pairs per
    // 16-byte output were the kernel's bound (≈300 VALU per thread, 527 us for the 1.1 GB ASPP upsample)
    const bool small = total <= 0x7fffffffL;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cv, ox, oy, n;
        if (small) {
            const unsigned u = (unsigned)i;
            unsigned pix = u / (unsigned)CV;
            cv = (int)(u - pix * (unsigned)CV);
            const unsigned row = pix / (unsigned)p.Wo;
            ox = (int)(pix - row * (unsigned)p.Wo);
            n = (int)(row / (unsigned)p.Ho);
            oy = (int)(row - (unsigned)n * (unsigned)p.Ho);
        } else {
            cv = (int)(i % CV);
            long pix = i / CV;
            ox = (int)(pix % p.Wo); pix /= p.Wo;
            oy = (int)(pix % p.Ho);
            n = (int)(pix / p.Ho);
        }
        const float sy = oy * p.ry, sx = ox * p.rx;
        const int y0 = min((int)sy, p.Hi - 1), x0 = min((int)sx, p.Wi - 1);
        const int y1 = min(y0 + 1, p.Hi - 1), x1 = min(x0 + 1, p.Wi - 1);
        const float fy = sy - y0, fx = sx - x0;
        const float w00 = (1.f - fy) * (1.f - fx), w01 = (1.f - fy) * fx, w10 = fy * (1.f - fx), w11 = fy * fx;
        const bf16_raw* base = x + (long)n * p.Hi * p.Wi * p.C + cv * VEC;
        const bf16_raw* p00 = base + ((long)y0 * p.Wi + x0) * p.C;
        const bf16_raw* p01 = base + ((long)y0 * p.Wi + x1) * p.C;
        const bf16_raw* p10 = base + ((long)y1 * p.Wi + x0) * p.C;
        const bf16_raw* p11 = base + ((long)y1 * p.Wi + x1) * p.C;
        bf16_raw* dst = y + (((long)n * p.Ho + oy) * p.Wo + ox) * p.ldo + cv * VEC;
        if (VEC == 8) {
            float a[8], b[8], c[8], d[8], o[8];
            unpack8(*reinterpret_cast<const uint4*>(p00), a);
            unpack8(*reinterpret_cast<const uint4*>(p01), b);
            unpack8(*reinterpret_cast<const uint4*>(p10), c);
            unpack8(*reinterpret_cast<const uint4*>(p11), d);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = w00 * a[j] + w01 * b[j] + w10 * c[j] + w11 * d[j];
            *reinterpret_cast<uint4*>(dst) = pack8(o);
        } else {
            dst[0] = f2bf(w00 * bf2f(p00[0]) + w01 * bf2f(p01[0]) + w10 * bf2f(p10[0]) + w11 * bf2f(p11[0]));
        }
    }
}

// gather-form backward: dx[n, iy, ix, c] = Σ_{oy,ox} wy(oy,iy)·wx(ox,ix)·dy[n, oy, ox, c]
template <int VEC>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const bf16_raw* __restrict__ dy, bf16_raw* __restrict__ dx,
                                                           Resize p) {
    const int CV = p.C / VEC;
    const long total = (long)p.N * p.Hi * p.Wi * CV;
    const float inv_ry = p.ry > 0.f ? 1.f / p.ry : 0.f, inv_rx = p.rx > 0.f ? 1.f / p.rx : 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        long pix = i / CV;
        const int ix = (int)(pix % p.Wi); pix /= p.Wi;
        const int iy = (int)(pix % p.Hi);
        const int n = (int)(pix / p.Hi);
        // output rows whose source coordinate sy = oy·ry lies in (iy-1, iy+1)
        int oy_lo, oy_hi, ox_lo, ox_hi;
        if (p.ry > 0.f) {
            oy_lo = max(0, (int)ceilf((iy - 1) * inv_ry - 1e-4f));
            oy_hi = min(p.Ho - 1, (int)floorf((iy + 1) * inv_ry + 1e-4f));
        } else { oy_lo = 0; oy_hi = p.Ho - 1; }
        if (p.rx > 0.f) {
            ox_lo = max(0, (int)ceilf((ix - 1) * inv_rx - 1e-4f));
            ox_hi = min(p.Wo - 1, (int)floorf((ix + 1) * inv_rx + 1e-4f));
        } else { ox_lo = 0; ox_hi = p.Wo - 1; }
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
        const bf16_raw* base = dy + (long)n * p.Ho * p.Wo * p.ldo + cv * VEC;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            // replicate the forward's tap selection exactly (including the clamped last row)
            const float sy = oy * p.ry;
            const int y0 = min((int)sy, p.Hi - 1), y1 = min(y0 + 1, p.Hi - 1);
            const float fy = sy - y0;
            float wy = 0.f;
            if (y0 == iy) wy += 1.f - fy;
            if (y1 == iy) wy += fy;
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const float sx = ox * p.rx;
                const int x0 = min((int)sx, p.Wi - 1), x1 = min(x0 + 1, p.Wi - 1);
                const float fx = sx - x0;
                float wx = 0.f;
                if (x0 == ix) wx += 1.f - fx;
                if (x1 == ix) wx += fx;
                if (wx == 0.f) continue;
                const float w = wy * wx;
                const bf16_raw* src = base + ((long)oy * p.Wo + ox) * p.ldo;
                if (VEC == 8) {
                    float f[8];
                    unpack8(*reinterpret_cast<const uint4*>(src), f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(w, f[j], acc[j]);
                } else {
                    acc[0] = fmaf(w, bf2f(src[0]), acc[0]);
                }
            }
        }
        bf16_raw* dst = dx + (((long)n * p.Hi + iy) * p.Wi + ix) * p.C + cv * VEC;
        if (VEC == 8) {
            *reinterpret_cast<uint4*>(dst) = pack8(acc);
        } else {
            dst[0] = f2bf(acc[0]);
        }
    }
}

// Channel counts that are not a multiple of 8 (the 21-class logits, 129^2 -> 513^2 in every evaluation forward): one
// block row per OUTPUT ROW — image, output row and the two source rows with their weights are block-uniform (scalar
// registers) — and the threads walk the row's Wo * C elements, so a thread pays ONE integer division (element -> pixel)
// where the flat kernel above pays three div/mod pairs: 678 us of a 9.8 ms R101 forward at batch 32 went there.
__global__ __launch_bounds__(256) void bilinear_fwd_row_kernel(const bf16_raw* __restrict__ x, bf16_raw* __restrict__ y,
                                                               Resize p) {
    const int rows = p.N * p.Ho;
    const unsigned rowlen = (unsigned)p.Wo * (unsigned)p.C;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        const int n = r / p.Ho, oy = r - n * p.Ho;
        const float sy = oy * p.ry;
        const int y0 = min((int)sy, p.Hi - 1), y1 = min(y0 + 1, p.Hi - 1);
        const float fy = sy - y0;
        const bf16_raw* row0 = x + ((long)n * p.Hi + y0) * p.Wi * p.C;
        const bf16_raw* row1 = x + ((long)n * p.Hi + y1) * p.Wi * p.C;
        bf16_raw* dst = y + (long)r * p.Wo * p.ldo;
        for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < rowlen; e += gridDim.x * blockDim.x) {
            const unsigned ox = e / (unsigned)p.C;
            const unsigned c = e - ox * (unsigned)p.C;
            const float sx = ox * p.rx;
            const int x0 = min((int)sx, p.Wi - 1), x1 = min(x0 + 1, p.Wi - 1);
            const float fx = sx - x0;
            const float top = bf2f(row0[x0 * p.C + c]) * (1.f - fx) + bf2f(row0[x1 * p.C + c]) * fx;
            const float bot = bf2f(row1[x0 * p.C + c]) * (1.f - fx) + bf2f(row1[x1 * p.C + c]) * fx;
            dst[(long)ox * p.ldo + c] = f2bf(top * (1.f - fy) + bot * fy);
        }
    }
}

inline int grid_1d(long total, int threads, int cap = 256 * 16) {
    long b = (total + threads - 1) / threads;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

// label_kind: 0 = uint8, 1 = int32, 2 = int64
DP_EXPORT int dp_softmax_ce_fwd(const void* logits, const void* target, int label_kind, long M, int C,
                                int ignore_index, float* out2, int reps, hipStream_t st) {
    if (M <= 0 || C <= 0 || C > 1024 || reps < 1) return -1;
    const size_t lds = ((size_t)kCeThreads * C * sizeof(bf16_raw) + 63) / 64 * 64;
    if (lds > 64 * 1024) return -2;
    const int grid = grid_1d(M, kCeThreads, 2048);
#define LAUNCH_FWD(T) hipLaunchKernelGGL(softmax_ce_fwd_kernel<T>, dim3(grid), dim3(kCeThreads), lds, st, \
                                         (const bf16_raw*)logits, target, M, C, ignore_index, out2, reps)
    if (label_kind == 0) LAUNCH_FWD(unsigned char);
    else if (label_kind == 1) LAUNCH_FWD(int);
    else LAUNCH_FWD(long long);
#undef LAUNCH_FWD
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_softmax_ce_bwd(const void* logits, const void* target, int label_kind, long M, int C,
                                int ignore_index, const float* sums, const float* gloss, void* dlogits,
                                hipStream_t st) {
    if (M <= 0 || C <= 0 || C > 1024) return -1;
    const size_t lds = ((size_t)kCeThreads * C * sizeof(bf16_raw) + 63) / 64 * 64;
    if (lds > 64 * 1024) return -2;
    const int grid = grid_1d(M, kCeThreads, 2048);
#define LAUNCH_BWD(T) hipLaunchKernelGGL(softmax_ce_bwd_kernel<T>, dim3(grid), dim3(kCeThreads), lds, st, \
                                         (const bf16_raw*)logits, target, M, C, ignore_index, sums, gloss, \
                                         (bf16_raw*)dlogits)
    if (label_kind == 0) LAUNCH_BWD(unsigned char);
    else if (label_kind == 1) LAUNCH_BWD(int);
    else LAUNCH_BWD(long long);
#undef LAUNCH_BWD
    return DP_LAUNCH_STATUS();
}

// ldo: channel stride (elements) of the high-resolution tensor — C, or the row width of the concat buffer
// that y (forward) / dy (backward) is a channel slice of
DP_EXPORT int dp_bilinear_fwd_ld(const void* x, void* y, int N, int C, int Hi, int Wi, int Ho, int Wo, int ldo,
                                 hipStream_t st) {
    if (N <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || ldo < C) return -1;
    if (C % 8 == 0 && (ldo % 8 != 0 || ((uintptr_t)y & 15))) return -1;
    Resize p{N, C, Hi, Wi, Ho, Wo, Ho > 1 ? (float)(Hi - 1) / (Ho - 1) : 0.f, Wo > 1 ? (float)(Wi - 1) / (Wo - 1) : 0.f,
             ldo};
    if (C % 8 == 0) {
        const long total = (long)N * Ho * Wo * (C / 8);
        hipLaunchKernelGGL(bilinear_fwd_kernel<8>, dim3(grid_1d(total, 256)), dim3(256), 0, st,
                           (const bf16_raw*)x, (bf16_raw*)y, p);
    } else {
        const long rowlen = (long)Wo * C;
        if (rowlen >= 256 && rowlen < (1L << 31) && (long)Wi * C < (1L << 31)) {
            const long rows = (long)N * Ho;
            const int gx = (int)((rowlen + 255) / 256 > 64 ? 64 : (rowlen + 255) / 256);
            hipLaunchKernelGGL(bilinear_fwd_row_kernel, dim3(gx, (unsigned)(rows > 65535 ? 65535 : rows)), dim3(256), 0, st,
                               (const bf16_raw*)x, (bf16_raw*)y, p);
        } else {
            const long total = (long)N * Ho * Wo * C;
            hipLaunchKernelGGL(bilinear_fwd_kernel<1>, dim3(grid_1d(total, 256)), dim3(256), 0, st,
                               (const bf16_raw*)x, (bf16_raw*)y, p);
        }
    }
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_bilinear_fwd(const void* x, void* y, int N, int C, int Hi, int Wi, int Ho, int Wo, hipStream_t st) {
    return dp_bilinear_fwd_ld(x, y, N, C, Hi, Wi, Ho, Wo, C, st);
}

DP_EXPORT int dp_bilinear_bwd_ld(const void* dy, void* dx, int N, int C, int Hi, int Wi, int Ho, int Wo, int ldo,
                                 hipStream_t st) {
    if (N <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || ldo < C) return -1;
    if (C % 8 == 0 && (ldo % 8 != 0 || ((uintptr_t)dy & 15))) return -1;
    Resize p{N, C, Hi, Wi, Ho, Wo, Ho > 1 ? (float)(Hi - 1) / (Ho - 1) : 0.f, Wo > 1 ? (float)(Wi - 1) / (Wo - 1) : 0.f,
             ldo};
    if (C % 8 == 0) {
        const long total = (long)N * Hi * Wi * (C / 8);
        hipLaunchKernelGGL(bilinear_bwd_kernel<8>, dim3(grid_1d(total, 256)), dim3(256), 0, st,
                           (const bf16_raw*)dy, (bf16_raw*)dx, p);
    } else {
        const long total = (long)N * Hi * Wi * C;
        hipLaunchKernelGGL(bilinear_bwd_kernel<1>, dim3(grid_1d(total, 256)), dim3(256), 0, st,
                           (const bf16_raw*)dy, (bf16_raw*)dx, p);
    }
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_bilinear_bwd(const void* dy, void* dx, int N, int C, int Hi, int Wi, int Ho, int Wo,
                              hipStream_t st) {
    return dp_bilinear_bwd_ld(dy, dx, N, C, Hi, Wi, Ho, Wo, C, st);
}

// ---- class-balanced sigmoid cross-entropy over up to 3 heads (the reference's instance task) ----------------------
// SegmentationMultiLosses(1) on DANet's 3-tuple (`/root/reference/train_pascal.py:119`, `:199`): per head
//   L_h = ((n_neg / n) * sum_{pos} bce + (n_pos / n) * sum_{neg} bce) / B,   L = sum_h w_h L_h.
// As stock tensor ops that is ~25 elementwise / reduction launches per head and direction over [B,1,512,512] fp32
// maps; here: one pass that leaves a row of partial sums PER BLOCK (no atomics: any two runs are bit-identical),
// one single-block pass that folds them in block order into the loss and the per-head coefficients, and one pass
// that writes the three gradients.
namespace {

constexpr int kBceHeads = 3;
constexpr int kBceCols = 2 + 2 * kBceHeads;      // n_pos, n_neg, then (sum_pos bce, sum_neg bce) per head

struct BceArgs {
    const void* x[kBceHeads];      // logits, fp32 or bf16 (is_bf16), M elements each
    const float* y;                // labels (>= 0.5 = positive)
    void* dx[kBceHeads];           // backward: gradients, same dtype as the logits
    float* part;                   // [blocks][kBceCols]
    float* out;                    // [1 + 2 * kBceHeads]: loss, then (c_pos, c_neg) per head
    const float* gloss;            // backward: upstream gradient (device scalar) or null (1)
    float w[kBceHeads];
    long M;
    int H, is_bf16, blocks;
    float inv_batch;
};

__device__ __forceinline__ float bce_ld(const void* p, long i, int is_bf16) {
    return is_bf16 ? bf2f(reinterpret_cast<const bf16_raw*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}

__global__ __launch_bounds__(256) void bce_sums_kernel(BceArgs a) {
    float acc[kBceCols];
#pragma unroll
    for (int c = 0; c < kBceCols; ++c) acc[c] = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.M; i += (long)gridDim.x * 256) {
        const bool pos = a.y[i] >= 0.5f;
        acc[pos ? 0 : 1] += 1.f;
#pragma unroll
        for (int h = 0; h < kBceHeads; ++h) {
            if (h < a.H) {
                const float x = bce_ld(a.x[h], i, a.is_bf16);
                // stable BCE with logits: softplus(-|x|) + relu(x) - x * y
                const float v = __logf(1.f + __expf(-fabsf(x))) + fmaxf(x, 0.f) - (pos ? x : 0.f);
                acc[2 + 2 * h + (pos ? 0 : 1)] += v;
            }
        }
    }
    __shared__ float red[4][kBceCols];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < kBceCols; ++c) {
        const float v = wave_sum(acc[c]);
        if (lane == 0) red[wave][c] = v;
    }
    __syncthreads();
    if (threadIdx.x < kBceCols)
        a.part[(long)blockIdx.x * kBceCols + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(64) void bce_final_kernel(BceArgs a) {
    __shared__ float tot[kBceCols];
    if (threadIdx.x < kBceCols) {
        float v = 0.f;
        for (int b = 0; b < a.blocks; ++b) v += a.part[(long)b * kBceCols + threadIdx.x];     // block order: fixed
        tot[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float n = fmaxf(tot[0] + tot[1], 1.f);
        float loss = 0.f;
        for (int h = 0; h < a.H; ++h) {
            const float cp = a.w[h] * (tot[1] / n) * a.inv_batch, cn = a.w[h] * (tot[0] / n) * a.inv_batch;
            loss += cp * tot[2 + 2 * h] + cn * tot[3 + 2 * h];
            a.out[1 + 2 * h] = cp;
            a.out[2 + 2 * h] = cn;
        }
        a.out[0] = loss;
    }
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(BceArgs a) {
    const float g = a.gloss ? a.gloss[0] : 1.f;
    float cp[kBceHeads], cn[kBceHeads];
#pragma unroll
    for (int h = 0; h < kBceHeads; ++h) { cp[h] = h < a.H ? g * a.out[1 + 2 * h] : 0.f; cn[h] = h < a.H ? g * a.out[2 + 2 * h] : 0.f; }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.M; i += (long)gridDim.x * 256) {
        const bool pos = a.y[i] >= 0.5f;
#pragma unroll
        for (int h = 0; h < kBceHeads; ++h) {
            if (h < a.H) {
                const float x = bce_ld(a.x[h], i, a.is_bf16);
                const float d = (pos ? cp[h] : cn[h]) * (1.f / (1.f + __expf(-x)) - (pos ? 1.f : 0.f));
                if (a.is_bf16) reinterpret_cast<bf16_raw*>(a.dx[h])[i] = f2bf(d);
                else reinterpret_cast<float*>(a.dx[h])[i] = d;
            }
        }
    }
}

inline int bce_blocks(long M) {
    long b = (M + 256 * 8 - 1) / (256 * 8);
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace

DP_EXPORT int dp_bce_blocks(long M) { return bce_blocks(M); }

// x0..x2: logits of up to 3 heads (H of them given), y: fp32 labels, part: fp32 [dp_bce_blocks(M)][8] scratch,
// out: fp32 [7] = loss, (c_pos, c_neg) per head (read back by dp_balanced_bce_bwd)
DP_EXPORT int dp_balanced_bce_fwd(const void* x0, const void* x1, const void* x2, int H, int is_bf16, const float* y,
                                  long M, int batch, float w0, float w1, float w2, float* part, float* out,
                                  hipStream_t st) {
    if (H < 1 || H > kBceHeads || !x0 || (H > 1 && !x1) || (H > 2 && !x2) || !y || M <= 0 || batch <= 0 || !part || !out)
        return -1;
    BceArgs a{};
    a.x[0] = x0; a.x[1] = x1; a.x[2] = x2; a.y = y; a.part = part; a.out = out;
    a.w[0] = w0; a.w[1] = w1; a.w[2] = w2; a.M = M; a.H = H; a.is_bf16 = is_bf16; a.blocks = bce_blocks(M);
    a.inv_batch = 1.f / (float)batch;
    hipLaunchKernelGGL(bce_sums_kernel, dim3(a.blocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(64), 0, st, a);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_balanced_bce_bwd(const void* x0, const void* x1, const void* x2, int H, int is_bf16, const float* y,
                                  long M, const float* out, const float* gloss, void* d0, void* d1, void* d2,
                                  hipStream_t st) {
    if (H < 1 || H > kBceHeads || !x0 || (H > 1 && !x1) || (H > 2 && !x2) || !y || M <= 0 || !out || !d0 || (H > 1 && !d1) ||
        (H > 2 && !d2))
        return -1;
    BceArgs a{};
    a.x[0] = x0; a.x[1] = x1; a.x[2] = x2; a.y = y; a.out = const_cast<float*>(out); a.gloss = gloss;
    a.dx[0] = d0; a.dx[1] = d1; a.dx[2] = d2; a.M = M; a.H = H; a.is_bf16 = is_bf16;
    long b = (M + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    hipLaunchKernelGGL(bce_bwd_kernel, dim3((unsigned)b), dim3(256), 0, st, a);
    return DP_LAUNCH_STATUS();
}
