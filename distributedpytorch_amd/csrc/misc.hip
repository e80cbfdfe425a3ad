// Small memory-bound kernels for NHWC bf16 on gfx950:
//   K10 fused flat SGD (momentum, weight decay, 1/world scale, bf16 refresh, grad zero)
//   K11 3×3/s2 max-pool fwd/bwd, global average pool fwd/bwd
//   K8/K9 fused inverse-affine augmentation (scale+rotate+crop+flip gather, normalise, NHWC bf16)
//   K12 confusion matrix (LDS histogram) for mIoU
//   plus a few glue ops (scale, channel pad) that would otherwise be ATen launches.
#include "common.h"

namespace {

inline int grid_1d(long total, int threads, int cap = 256 * 16) {
    long b = (total + threads - 1) / threads;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ------------------------------------------- SGD -------------------------------------------
__global__ __launch_bounds__(256) void sgd_flat_kernel(float* __restrict__ p, float* __restrict__ g,
                                                       float* __restrict__ m, bf16_raw* __restrict__ p16, long n,
                                                       float lr, float mu, float wd, float gscale, int zero_grad,
                                                       int first, const float* __restrict__ lr_scale /* = lr_dev */) {
    // lr_dev (or null): the learning rate in device memory — a captured step (hipGraph) freezes the `lr` argument, the
    // schedule then writes this scalar between replays (dp_sgd_set_lr_ptr; one per parameter group)
    if (lr_scale) lr = lr_scale[0];
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        float4 gv = reinterpret_cast<float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float* pp = &pv.x; float* gg = &gv.x; float* mm = &mv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = fmaf(wd, pp[j], gg[j] * gscale);
            const float mom = first ? d : fmaf(mu, mm[j], d);
            mm[j] = mom;
            pp[j] = fmaf(-lr, mu != 0.f ? mom : d, pp[j]);
        }
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        if (p16) {
            uint2 o;
            o.x = pack2(pp[0], pp[1]); o.y = pack2(pp[2], pp[3]);
            reinterpret_cast<uint2*>(p16)[i] = o;
        }
        if (zero_grad) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // tail (n not a multiple of 4)
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = fmaf(wd, p[i], g[i] * gscale);
        const float mom = first ? d : fmaf(mu, m[i], d);
        m[i] = mom;
        p[i] = fmaf(-lr, mu != 0.f ? mom : d, p[i]);
        if (p16) p16[i] = f2bf(p[i]);
        if (zero_grad) g[i] = 0.f;
    }
}

// ----------------------------------------- max-pool -----------------------------------------
struct Pool { int N, C, Hi, Wi, Ho, Wo; };

// forward also records, per output element, WHICH of the 9 taps was the (first, row-major) maximum:
// one byte, 8 channels = one 8-byte store.  Backward then never touches x: an input pixel looks at
// the <= 4 windows covering it and takes dy where the recorded tap is its own (ATen's tie rule, and
// 6 loads per thread instead of ~40 — the first version spent 754 us per step here).
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const bf16_raw* __restrict__ x, bf16_raw* __restrict__ y,
                                                          unsigned char* __restrict__ idx, Pool p) {
    const int C8 = p.C >> 3;
    const long total = (long)p.N * p.Ho * p.Wo * C8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % C8);
        long pix = i / C8;
        const int ox = (int)(pix % p.Wo); pix /= p.Wo;
        const int oy = (int)(pix % p.Ho);
        const int n = (int)(pix / p.Ho);
        float best[8];
        unsigned code[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { best[j] = -3.0e38f; code[j] = 4; }
        for (int r = 0; r < 3; ++r) {
            const int iy = oy * 2 - 1 + r;
            if (iy < 0 || iy >= p.Hi) continue;
            for (int s = 0; s < 3; ++s) {
                const int ix = ox * 2 - 1 + s;
                if (ix < 0 || ix >= p.Wi) continue;
                float f[8];
                unpack8(*reinterpret_cast<const uint4*>(x + (((long)n * p.Hi + iy) * p.Wi + ix) * p.C + cg * 8), f);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (f[j] > best[j]) { best[j] = f[j]; code[j] = (unsigned)(r * 3 + s); }
            }
        }
        const long o = (((long)n * p.Ho + oy) * p.Wo + ox) * p.C + cg * 8;
        *reinterpret_cast<uint4*>(y + o) = pack8(best);
        uint2 pk;
        pk.x = code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24);
        pk.y = code[4] | (code[5] << 8) | (code[6] << 16) | (code[7] << 24);
        *reinterpret_cast<uint2*>(idx + o) = pk;
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16_raw* __restrict__ dy,
                                                          const unsigned char* __restrict__ idx,
                                                          bf16_raw* __restrict__ dx, Pool p) {
    const int C8 = p.C >> 3;
    const long total = (long)p.N * p.Hi * p.Wi * C8;
    const bool small = total <= 0x7fffffffL;     // 32-bit index arithmetic (64-bit div/mod: ~100 VALU each)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cg, ix, iy, n;
        if (small) {
            const unsigned u = (unsigned)i;
            const unsigned pix = u / (unsigned)C8;
            cg = (int)(u - pix * (unsigned)C8);
            const unsigned row = pix / (unsigned)p.Wi;
            ix = (int)(pix - row * (unsigned)p.Wi);
            n = (int)(row / (unsigned)p.Hi);
            iy = (int)(row - (unsigned)n * (unsigned)p.Hi);
        } else {
            cg = (int)(i % C8);
            long pix = i / C8;
            ix = (int)(pix % p.Wi); pix /= p.Wi;
            iy = (int)(pix % p.Hi);
            n = (int)(pix / p.Hi);
        }
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        // windows (oy, ox) with oy*2-1 <= iy <= oy*2+1
        const int oy_lo = iy / 2, oy_hi = min(p.Ho - 1, (iy + 1) / 2);
        const int ox_lo = ix / 2, ox_hi = min(p.Wo - 1, (ix + 1) / 2);
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const unsigned mine = (unsigned)((iy - (oy * 2 - 1)) * 3 + (ix - (ox * 2 - 1)));
                const long o = (((long)n * p.Ho + oy) * p.Wo + ox) * p.C + cg * 8;
                const uint2 pk = *reinterpret_cast<const uint2*>(idx + o);
                float dv[8];
                unpack8(*reinterpret_cast<const uint4*>(dy + o), dv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned cj = ((j < 4 ? pk.x : pk.y) >> (8 * (j & 3))) & 0xffu;
                    if (cj == mine) acc[j] += dv[j];
                }
            }
        }
        *reinterpret_cast<uint4*>(dx + (((long)n * p.Hi + iy) * p.Wi + ix) * p.C + cg * 8) = pack8(acc);
    }
}

// ------------------------------------- global average pool -------------------------------------
// one block per (n, 32-channel-group slab): threads = 4 channel groups × 64 row lanes
__global__ __launch_bounds__(256) void gap_fwd_kernel(const bf16_raw* __restrict__ x, bf16_raw* __restrict__ y,
                                                      int C, int HW) {
    __shared__ float red[8][256];
    const int C8 = C >> 3;
    const int n = blockIdx.y;
    const int cgl = threadIdx.x & 3, rl = threadIdx.x >> 2;
    const int cg = blockIdx.x * 4 + cgl;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (cg < C8) {
        const bf16_raw* base = x + (long)n * HW * C + cg * 8;
        for (int r = rl; r < HW; r += 64) {
            float f[8];
            unpack8(*reinterpret_cast<const uint4*>(base + (long)r * C), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += f[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[j][threadIdx.x] = acc[j];
    __syncthreads();
    if (rl == 0 && cg < C8) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float s = 0.f;
            for (int k = 0; k < 64; ++k) s += red[j][k * 4 + cgl];
            o[j] = s / (float)HW;
        }
        *reinterpret_cast<uint4*>(y + (long)n * C + cg * 8) = pack8(o);
    }
}

__global__ __launch_bounds__(256) void gap_bwd_kernel(const bf16_raw* __restrict__ dy, bf16_raw* __restrict__ dx,
                                                      int N, int C, int HW) {
    const int C8 = C >> 3;
    const long total = (long)N * HW * C8;
    const float inv = 1.f / (float)HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % C8);
        const int n = (int)(i / ((long)HW * C8));
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(dy + (long)n * C + cg * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] *= inv;
        reinterpret_cast<uint4*>(dx)[i] = pack8(f);
    }
}

// --------------------------------------- augmentation ---------------------------------------
struct Aug {
    int B, Hi, Wi, Ho, Wo, Cout;   // Cout: channels written per pixel (3, or padded to 4/8 with zeros)
    float mean[3], inv_std[3];
};

__global__ __launch_bounds__(256) void affine_augment_kernel(const unsigned char* __restrict__ img,
                                                             const unsigned char* __restrict__ lab,
                                                             const float* __restrict__ mats,
                                                             bf16_raw* __restrict__ out_img,
                                                             unsigned char* __restrict__ out_lab, Aug p) {
    const long total = (long)p.B * p.Ho * p.Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % p.Wo);
        const int oy = (int)((i / p.Wo) % p.Ho);
        const int b = (int)(i / ((long)p.Wo * p.Ho));
        const float* m = mats + b * 6;
        const float sx = fmaf(m[0], (float)ox, fmaf(m[1], (float)oy, m[2]));
        const float sy = fmaf(m[3], (float)ox, fmaf(m[4], (float)oy, m[5]));
        const float fx0 = floorf(sx), fy0 = floorf(sy);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float ax = sx - fx0, ay = sy - fy0;
        const unsigned char* ib = img + (long)b * p.Hi * p.Wi * 3;
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
            const float w = ((t & 1) ? ax : 1.f - ax) * ((t >> 1) ? ay : 1.f - ay);
            if (xx >= 0 && xx < p.Wi && yy >= 0 && yy < p.Hi) {
                const unsigned char* px = ib + ((long)yy * p.Wi + xx) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = fmaf(w, ((float)px[c] - p.mean[c]) * p.inv_std[c], acc[c]);
            }  // out-of-bounds tap == mean colour == 0 after normalisation
        }
        bf16_raw* dst = out_img + i * p.Cout;
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[c] = f2bf(acc[c]);
        for (int c = 3; c < p.Cout; ++c) dst[c] = 0;
        if (lab) {
            const int xn = (int)floorf(sx + 0.5f), yn = (int)floorf(sy + 0.5f);
            unsigned char v = 255;
            if (xn >= 0 && xn < p.Wi && yn >= 0 && yn < p.Hi) v = lab[((long)b * p.Hi + yn) * p.Wi + xn];
            out_lab[i] = v;
        }
    }
}

// -------------------------------------- confusion matrix --------------------------------------
// pred_kind 0: logits bf16 [M, C] (arg-max on the fly), 1: uint8 class map, 2: int64 class map
template <typename LabelT>
__global__ __launch_bounds__(256) void confusion_kernel(const void* __restrict__ pred, int pred_kind,
                                                        const LabelT* __restrict__ target, long M, int C, int K,
                                                        int ignore_index, unsigned long long* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned int* hist = reinterpret_cast<unsigned int*>(smem_raw);
    for (int i = threadIdx.x; i < K * K; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < M; r += (long)gridDim.x * blockDim.x) {
        const int t = (int)target[r];
        if (t == ignore_index || t < 0 || t >= K) continue;
        int pc;
        if (pred_kind == 0) {
            const bf16_raw* row = reinterpret_cast<const bf16_raw*>(pred) + r * C;
            float best = bf2f(row[0]);
            pc = 0;
            for (int c = 1; c < C; ++c) {
                const float v = bf2f(row[c]);
                if (v > best) { best = v; pc = c; }
            }
        } else if (pred_kind == 1) {
            pc = (int)reinterpret_cast<const unsigned char*>(pred)[r];
        } else {
            pc = (int)reinterpret_cast<const long long*>(pred)[r];
        }
        if (pc >= 0 && pc < K) atomicAdd(&hist[t * K + pc], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K * K; i += blockDim.x)
        if (hist[i]) atomicAdd(out + i, (unsigned long long)hist[i]);
}

// channel pad: [M, Cin] -> [M, Cout] (zeros beyond Cin); used for the 3-channel network input
__global__ __launch_bounds__(256) void pad_channels_kernel(const bf16_raw* __restrict__ x, bf16_raw* __restrict__ y,
                                                           long M, int Cin, int Cout) {
    const long total = M * Cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cout);
        const long r = i / Cout;
        y[i] = c < Cin ? x[r * Cin + c] : (bf16_raw)0;
    }
}

// Stem input packing: [N, H, W, C] (C <= 4) -> [N, H, Wp, 4] with `left` zero pixels before each row's
// data and zeros after it: the layout the windowed stem convolution (dp_conv_igemm2w) reads — the
// horizontal padding is physical, so a filter row is ONE contiguous 64-byte segment per output pixel.
__global__ __launch_bounds__(256) void stem_pack_kernel(const bf16_raw* __restrict__ x, bf16_raw* __restrict__ y,
                                                        long rows, int W, int C, int Wp, int left) {
    const long total = rows * Wp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xp = (int)(i % Wp);
        const long r = i / Wp;
        const int xs = xp - left;
        unsigned lo = 0, hi = 0;
        if (xs >= 0 && xs < W) {
            const bf16_raw* src = x + (r * W + xs) * C;
            const unsigned v0 = src[0], v1 = C > 1 ? src[1] : 0u, v2 = C > 2 ? src[2] : 0u, v3 = C > 3 ? src[3] : 0u;
            lo = v0 | (v1 << 16);
            hi = v2 | (v3 << 16);
        }
        *reinterpret_cast<uint2*>(y + i * 4) = make_uint2(lo, hi);
    }
}

}  // namespace

int g_dp_deterministic = 0;
const unsigned* g_dp_drop_word = nullptr;
namespace { const float* g_lr_scale = nullptr; }
DP_EXPORT int dp_set_drop_word(const void* dev_word) { g_dp_drop_word = (const unsigned*)dev_word; return 0; }
DP_EXPORT int dp_sgd_set_lr_ptr(const void* dev_scalar) { g_lr_scale = (const float*)dev_scalar; return 0; }
DP_EXPORT int dp_set_deterministic(int on) {
    const int old = g_dp_deterministic;
    g_dp_deterministic = on ? 1 : 0;
    return old;
}

DP_EXPORT int dp_sgd_flat(float* p, float* g, float* m, void* p16, long n, float lr, float mu, float wd,
                          float gscale, int zero_grad, int first, hipStream_t st) {
    if (n <= 0) return 0;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m) & 15) return -3;
    if (p16 && ((uintptr_t)p16 & 7)) return -3;
    hipLaunchKernelGGL(sgd_flat_kernel, dim3(grid_1d(n / 4 + 1, 256, 2048)), dim3(256), 0, st, p, g, m,
                       (bf16_raw*)p16, n, lr, mu, wd, gscale, zero_grad, first, g_lr_scale);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_maxpool3x3s2_fwd(const void* x, void* y, void* idx, int N, int C, int Hi, int Wi, int Ho, int Wo,
                                  hipStream_t st) {
    if (C % 8 != 0 || Ho != (Hi + 2 - 3) / 2 + 1 || Wo != (Wi + 2 - 3) / 2 + 1) return -1;
    Pool p{N, C, Hi, Wi, Ho, Wo};
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_1d((long)N * Ho * Wo * (C / 8), 256)), dim3(256), 0, st,
                       (const bf16_raw*)x, (bf16_raw*)y, (unsigned char*)idx, p);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_maxpool3x3s2_bwd(const void* dy, const void* idx, void* dx, int N, int C, int Hi,
                                  int Wi, int Ho, int Wo, hipStream_t st) {
    if (C % 8 != 0 || Ho != (Hi + 2 - 3) / 2 + 1 || Wo != (Wi + 2 - 3) / 2 + 1) return -1;
    Pool p{N, C, Hi, Wi, Ho, Wo};
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_1d((long)N * Hi * Wi * (C / 8), 256)), dim3(256), 0, st,
                       (const bf16_raw*)dy, (const unsigned char*)idx, (bf16_raw*)dx, p);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_gap_fwd(const void* x, void* y, int N, int C, int HW, hipStream_t st) {
    if (C % 8 != 0 || N <= 0 || HW <= 0) return -1;
    hipLaunchKernelGGL(gap_fwd_kernel, dim3((C / 8 + 3) / 4, N), dim3(256), 0, st, (const bf16_raw*)x,
                       (bf16_raw*)y, C, HW);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_gap_bwd(const void* dy, void* dx, int N, int C, int HW, hipStream_t st) {
    if (C % 8 != 0 || N <= 0 || HW <= 0) return -1;
    hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_1d((long)N * HW * (C / 8), 256)), dim3(256), 0, st,
                       (const bf16_raw*)dy, (bf16_raw*)dx, N, C, HW);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_affine_augment(const void* img, const void* lab, const float* mats, void* out_img, void* out_lab,
                                int B, int Hi, int Wi, int Ho, int Wo, int Cout, const float* mean3,
                                const float* std3, hipStream_t st) {
    if (B <= 0 || Cout < 3 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return -1;
    Aug p{B, Hi, Wi, Ho, Wo, Cout, {mean3[0], mean3[1], mean3[2]}, {1.f / std3[0], 1.f / std3[1], 1.f / std3[2]}};
    hipLaunchKernelGGL(affine_augment_kernel, dim3(grid_1d((long)B * Ho * Wo, 256)), dim3(256), 0, st,
                       (const unsigned char*)img, (const unsigned char*)lab, mats, (bf16_raw*)out_img,
                       (unsigned char*)out_lab, p);
    return DP_LAUNCH_STATUS();
}

// target_kind: 0 uint8, 2 int64
DP_EXPORT int dp_confusion(const void* pred, int pred_kind, const void* target, int target_kind, long M, int C,
                           int K, int ignore_index, void* out_u64, hipStream_t st) {
    if (M <= 0 || K <= 0 || K > 128) return -1;
    const size_t lds = (size_t)K * K * sizeof(unsigned int);
    const int grid = grid_1d(M, 256, 1024);
    if (target_kind == 0)
        hipLaunchKernelGGL(confusion_kernel<unsigned char>, dim3(grid), dim3(256), lds, st, pred, pred_kind,
                           (const unsigned char*)target, M, C, K, ignore_index, (unsigned long long*)out_u64);
    else
        hipLaunchKernelGGL(confusion_kernel<long long>, dim3(grid), dim3(256), lds, st, pred, pred_kind,
                           (const long long*)target, M, C, K, ignore_index, (unsigned long long*)out_u64);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_stem_pack(const void* x, void* y, long rows, int W, int C, int Wp, int left, hipStream_t st) {
    if (rows <= 0 || W <= 0 || C < 1 || C > 4 || left < 0 || Wp < W + left || ((uintptr_t)y & 7)) return -1;
    hipLaunchKernelGGL(stem_pack_kernel, dim3(grid_1d(rows * Wp, 256)), dim3(256), 0, st, (const bf16_raw*)x,
                       (bf16_raw*)y, rows, W, C, Wp, left);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_pad_channels(const void* x, void* y, long M, int Cin, int Cout, hipStream_t st) {
    if (M <= 0 || Cin <= 0 || Cout < Cin) return -1;
    hipLaunchKernelGGL(pad_channels_kernel, dim3(grid_1d(M * Cout, 256)), dim3(256), 0, st, (const bf16_raw*)x,
                       (bf16_raw*)y, M, Cin, Cout);
    return DP_LAUNCH_STATUS();
}
