// Depthwise 3×3 (atrous) convolution for NHWC bf16 on gfx950 — the spatial
// half of every Xception separable unit.  Memory-bound (9 MACs per loaded element), so no MFMA:
// one thread owns 8 channels of one pixel, every access is a 16-B vector, weights are kept
// tap-major [9][C] so the 8 channels of a tap are one 16-B load.
//   fwd        y[n,oy,ox,c]  = Σ_t x[n, oy·s-p+r·d, ox·s-p+q·d, c] · w[t][c]
//   bwd_data   dx[n,iy,ix,c] = Σ_t dy[n,(iy+p-r·d)/s,(ix+p-q·d)/s,c] · w[t][c]   (where divisible)
//   bwd_weight dw[t][c]      = Σ_{n,oy,ox} dy · x      (column reduction, same mapping as BN stats)
#include "common.h"
#include <stdlib.h>

namespace {

struct Dw { int N, C, Hi, Wi, Ho, Wo, stride, pad_h, pad_w, dil; };

// Block -> (image band bx, 64-channel slab by) of the rolling-window kernels, from a 1-D grid.  The slabs of one band
// are CONSECUTIVE ids of one XCD (the same bijective remap as the dense kernels): with 728 channels a pixel's
// 128-byte slab segment starts 48 p mod 128 bytes into a cache line, so neighbouring slabs share every line at their
// seam — on one XCD, launched back to back, the second reader finds the line in that L2 and the two halves of a
// written line meet there; spread over XCDs each line crosses the fabric twice (728-channel forward 295 us against
// 311 us for 1024 channels with 40 % more bytes).
// With whole cache lines per slab segment (C % 64 == 0) there is no seam to share and the band-major order measured
// SLOWER (1024 channels: 280 -> 317 us, concurrent blocks walk the same DRAM pages): those keep the slab-major order.
__device__ __forceinline__ void dw_block_coords(int nb, int slabs, int C, int& bx, int& by) {
    const int total = nb * slabs, L = blockIdx.x;
    if ((C & 63) == 0) {
        by = L / nb;
        bx = L - by * nb;
        return;
    }
    const int q = total >> 3, r = total & 7, xcd = L & 7, idx = L >> 3;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    by = v % slabs;
    bx = v / slabs;
}

// sum over the 16 lanes of a DPP row (all lanes of the row end up with the total)
__device__ __forceinline__ float dw_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

inline int grid_1d(long total, int threads, int cap = 256 * 16) {
    long b = (total + threads - 1) / threads;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ w9,
                                                         bf16_raw* __restrict__ y, Dw p) {
    const int C8 = p.C >> 3;
    const long total = (long)p.N * p.Ho * p.Wo * C8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % C8);
        long pix = i / C8;
        const int ox = (int)(pix % p.Wo); pix /= p.Wo;
        const int oy = (int)(pix % p.Ho);
        const int n = (int)(pix / p.Ho);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = oy * p.stride - p.pad_h + r * p.dil;
            if ((unsigned)iy >= (unsigned)p.Hi) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int ix = ox * p.stride - p.pad_w + q * p.dil;
                if ((unsigned)ix >= (unsigned)p.Wi) continue;
                float xv[8], wv[8];
                unpack8(*reinterpret_cast<const uint4*>(x + (((long)n * p.Hi + iy) * p.Wi + ix) * p.C + cg * 8), xv);
                unpack8(*reinterpret_cast<const uint4*>(w9 + (r * 3 + q) * p.C + cg * 8), wv);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv[j], wv[j], acc[j]);
            }
        }
        reinterpret_cast<uint4*>(y)[i] = pack8(acc);
    }
}

__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(const bf16_raw* __restrict__ dy,
                                                              const bf16_raw* __restrict__ w9,
                                                              bf16_raw* __restrict__ dx, Dw p) {
    const int C8 = p.C >> 3;
    const long total = (long)p.N * p.Hi * p.Wi * C8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % C8);
        long pix = i / C8;
        const int ix = (int)(pix % p.Wi); pix /= p.Wi;
        const int iy = (int)(pix % p.Hi);
        const int n = (int)(pix / p.Hi);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ty = iy + p.pad_h - r * p.dil;
            if (ty < 0 || ty % p.stride) continue;
            const int oy = ty / p.stride;
            if (oy >= p.Ho) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int tx = ix + p.pad_w - q * p.dil;
                if (tx < 0 || tx % p.stride) continue;
                const int ox = tx / p.stride;
                if (ox >= p.Wo) continue;
                float dv[8], wv[8];
                unpack8(*reinterpret_cast<const uint4*>(dy + (((long)n * p.Ho + oy) * p.Wo + ox) * p.C + cg * 8), dv);
                unpack8(*reinterpret_cast<const uint4*>(w9 + (r * 3 + q) * p.C + cg * 8), wv);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(dv[j], wv[j], acc[j]);
            }
        }
        reinterpret_cast<uint4*>(dx)[i] = pack8(acc);
    }
}

// thread = (channel group, row lane); rows = output pixels; 9 taps × 8 channels of fp32 partials
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const bf16_raw* __restrict__ dy,
                                                                const bf16_raw* __restrict__ x,
                                                                float* __restrict__ dw9, Dw p) {
    __shared__ float smem[8 * 256];
    const int C8 = p.C >> 3;
    const int base = blockIdx.y * 256;
    const int cgs = min(256, C8 - base);
    const int rpp = 256 / cgs;
    const int cgl = threadIdx.x % cgs, rlane = threadIdx.x / cgs;
    const int cg = base + cgl;
    const bool active = rlane < rpp;
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    const long P = (long)p.N * p.Ho * p.Wo;
    if (active) {
        for (long r = (long)blockIdx.x * rpp + rlane; r < P; r += (long)gridDim.x * rpp) {
            const int ox = (int)(r % p.Wo);
            const long t2 = r / p.Wo;
            const int oy = (int)(t2 % p.Ho);
            const int n = (int)(t2 / p.Ho);
            float dv[8];
            unpack8(*reinterpret_cast<const uint4*>(dy + r * p.C + cg * 8), dv);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int iy = oy * p.stride - p.pad_h + a * p.dil;
                if ((unsigned)iy >= (unsigned)p.Hi) continue;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int ix = ox * p.stride - p.pad_w + q * p.dil;
                    if ((unsigned)ix >= (unsigned)p.Wi) continue;
                    float xv[8];
                    unpack8(*reinterpret_cast<const uint4*>(x + (((long)n * p.Hi + iy) * p.Wi + ix) * p.C + cg * 8), xv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[a * 3 + q][j] = fmaf(dv[j], xv[j], acc[a * 3 + q][j]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) smem[j * 256 + threadIdx.x] = active ? acc[t][j] : 0.f;
        __syncthreads();
        if (rlane == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float s = 0.f;
                for (int rl = 0; rl < rpp; ++rl) s += smem[j * 256 + rl * cgs + cgl];
                atomicAdd(dw9 + t * p.C + cg * 8 + j, s);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Second generation: strip mapping.  The rocprofv3 trace of Xception-65 (profiles/
// r1_native_x65_os8_b16.md) shows the kernels above 3-11x off the HBM time of their tensors
// (bwd_weight 445 us for 197 MB): a grid-stride loop sends consecutive iterations of a thread to pixels
// a thousand apart, so none of the 9 overlapping taps is ever found in the CU's L1, the 9 weight vectors
// are re-loaded per pixel, and every element pays three integer divisions.  Here a thread owns ONE
// channel group (8 channels) and a strip of kStrip CONSECUTIVE pixels: the weights (fwd / bwd_data) or
// the 72 partial sums (bwd_weight) live in registers for the whole strip, the pixel coordinates advance
// by increments, and the taps of neighbouring pixels hit in L1.  A block is 32 channel groups x 8
// strips (a wave-instruction moves 2 x 512 contiguous bytes).
constexpr int kSlab = 32;      // channel groups per block
constexpr int kStrips = 8;     // strips per block (kSlab * kStrips = 256 threads)

struct Strip { int cg; long p0, p1; bool active; };

template <int STRIP>
__device__ __forceinline__ Strip make_strip(int C8, long P) {
    Strip s;
    const int cgl = threadIdx.x % kSlab, st = threadIdx.x / kSlab;
    s.cg = blockIdx.y * kSlab + cgl;
    s.p0 = ((long)blockIdx.x * kStrips + st) * STRIP;
    s.p1 = min(P, s.p0 + STRIP);
    s.active = s.cg < C8 && s.p0 < P;
    return s;
}

template <int STRIP>
__global__ __launch_bounds__(256) void dwconv2_fwd_kernel(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ w9,
                                                          bf16_raw* __restrict__ y, Dw p) {
    const Strip s = make_strip<STRIP>(p.C >> 3, (long)p.N * p.Ho * p.Wo);
    if (!s.active) return;
    float w[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t) unpack8(*reinterpret_cast<const uint4*>(w9 + t * p.C + s.cg * 8), w[t]);
    int ox = (int)(s.p0 % p.Wo);
    long t2 = s.p0 / p.Wo;
    int oy = (int)(t2 % p.Ho);
    int n = (int)(t2 / p.Ho);
    for (long r = s.p0; r < s.p1; ++r) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        const bf16_raw* xn = x + (long)n * p.Hi * p.Wi * p.C + s.cg * 8;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int iy = oy * p.stride - p.pad_h + a * p.dil;
            if ((unsigned)iy >= (unsigned)p.Hi) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int ix = ox * p.stride - p.pad_w + q * p.dil;
                if ((unsigned)ix >= (unsigned)p.Wi) continue;
                float xv[8];
                unpack8(*reinterpret_cast<const uint4*>(xn + ((long)iy * p.Wi + ix) * p.C), xv);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv[j], w[a * 3 + q][j], acc[j]);
            }
        }
        *reinterpret_cast<uint4*>(y + r * p.C + s.cg * 8) = pack8(acc);
        if (++ox == p.Wo) { ox = 0; if (++oy == p.Ho) { oy = 0; ++n; } }
    }
}

// stride 1 only: dx[iy, ix] = Σ_t dy[iy + pad - a·dil, ix + pad - q·dil] · w[t]
template <int STRIP>
__global__ __launch_bounds__(256) void dwconv2_bwd_data_kernel(const bf16_raw* __restrict__ dy,
                                                               const bf16_raw* __restrict__ w9,
                                                               bf16_raw* __restrict__ dx, Dw p) {
    const Strip s = make_strip<STRIP>(p.C >> 3, (long)p.N * p.Hi * p.Wi);
    if (!s.active) return;
    float w[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t) unpack8(*reinterpret_cast<const uint4*>(w9 + t * p.C + s.cg * 8), w[t]);
    int ix = (int)(s.p0 % p.Wi);
    long t2 = s.p0 / p.Wi;
    int iy = (int)(t2 % p.Hi);
    int n = (int)(t2 / p.Hi);
    for (long r = s.p0; r < s.p1; ++r) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        const bf16_raw* dn = dy + (long)n * p.Ho * p.Wo * p.C + s.cg * 8;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int oy = iy + p.pad_h - a * p.dil;
            if ((unsigned)oy >= (unsigned)p.Ho) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int ox = ix + p.pad_w - q * p.dil;
                if ((unsigned)ox >= (unsigned)p.Wo) continue;
                float dv[8];
                unpack8(*reinterpret_cast<const uint4*>(dn + ((long)oy * p.Wo + ox) * p.C), dv);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(dv[j], w[a * 3 + q][j], acc[j]);
            }
        }
        *reinterpret_cast<uint4*>(dx + r * p.C + s.cg * 8) = pack8(acc);
        if (++ix == p.Wi) { ix = 0; if (++iy == p.Hi) { iy = 0; ++n; } }
    }
}

template <int STRIP>
__global__ __launch_bounds__(256) void dwconv2_bwd_weight_kernel(const bf16_raw* __restrict__ dy,
                                                                 const bf16_raw* __restrict__ x,
                                                                 float* __restrict__ dw9, Dw p) {
    __shared__ float red[kStrips][kSlab][9];       // one channel of the 8 at a time
    const Strip s = make_strip<STRIP>(p.C >> 3, (long)p.N * p.Ho * p.Wo);
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    if (s.active) {
        int ox = (int)(s.p0 % p.Wo);
        long t2 = s.p0 / p.Wo;
        int oy = (int)(t2 % p.Ho);
        int n = (int)(t2 / p.Ho);
        for (long r = s.p0; r < s.p1; ++r) {
            float dv[8];
            unpack8(*reinterpret_cast<const uint4*>(dy + r * p.C + s.cg * 8), dv);
            const bf16_raw* xn = x + (long)n * p.Hi * p.Wi * p.C + s.cg * 8;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int iy = oy * p.stride - p.pad_h + a * p.dil;
                if ((unsigned)iy >= (unsigned)p.Hi) continue;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int ix = ox * p.stride - p.pad_w + q * p.dil;
                    if ((unsigned)ix >= (unsigned)p.Wi) continue;
                    float xv[8];
                    unpack8(*reinterpret_cast<const uint4*>(xn + ((long)iy * p.Wi + ix) * p.C), xv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[a * 3 + q][j] = fmaf(dv[j], xv[j], acc[a * 3 + q][j]);
                }
            }
            if (++ox == p.Wo) { ox = 0; if (++oy == p.Ho) { oy = 0; ++n; } }
        }
    }
    // block combine over the 8 strips, one of the 8 channels per round; then one atomic per (tap, channel)
    const int cgl = threadIdx.x % kSlab, st = threadIdx.x / kSlab;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t) red[st][cgl][t] = acc[t][j];
        __syncthreads();
        for (int i = threadIdx.x; i < kSlab * 9; i += 256) {
            const int c = i / 9, t = i - c * 9;
            const int cg = blockIdx.y * kSlab + c;
            if (cg >= (p.C >> 3)) continue;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < kStrips; ++k) v += red[k][c][t];
            atomicAdd(dw9 + t * p.C + cg * 8 + j, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Third generation: LDS row tiles (stride 1, "same" padding: pad == dil — every Xception middle/exit
// unit).  Both mappings above leave the 9x reuse of an input element to the caches: with dilation d the
// three tap rows of a pixel are 2d image rows apart (d = 2, 728 channels, 65 columns: 376 KB), far beyond
// L1, so every tap is an L2 request — 8 TB/s of L2 traffic for a 100 MB tensor, 2.5x (fwd) to 8x (wgrad)
// off the HBM time (profiles/r1_native_x65_os8_b16_v2.md).  Here a block stages TH + 2d FULL-WIDTH image
// rows x 64 channels in LDS once (global reads: (TH + 2d) / TH of the tensor, 128 contiguous bytes per
// pixel), and all taps are conflict-free ds_read_b128: a wave reads 8 neighbouring pixels x 8 channel
// groups = 1 KiB contiguous.
//   MODE 0: out = depthwise conv of `in` (forward; data gradient = the same with the taps flipped)
//   MODE 2: dw[t][c] += sum over the tile of in2 (dy) x tap t of `in` (x): 72 partial sums per thread,
//           block combine through the (by then free) LDS tile, one atomic per (tap, channel) and block.
struct Dw3 {
    const bf16_raw* in; const bf16_raw* in2; const bf16_raw* w9; void* out;
    int N, C, H, W, dil, TH, tiles_h, flip;
    unsigned in_bytes;
    float* stats;           // MODE 0: optional [stat_reps][2][C] per-channel sums of out and out^2 (BatchNorm
    int stat_reps;          //   batch statistics for free, as in the dense kernels' epilogue)
};

constexpr unsigned kOOBd = 0x80000000u;
constexpr int kDw3MaxPx = 12;          // MODE 2: pixels per thread whose dy is pre-loaded into registers

template <int MODE>
__global__ __launch_bounds__(256, 2) void dwconv3_kernel(Dw3 p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    // [rows + 2d][W + 2d][8] 16-byte chunks: the horizontal padding is physically present (zeros), so the
    // nine taps of a pixel are nine FIXED offsets from its own chunk — no bounds tests in the tap loop
    const uint4* tile = reinterpret_cast<const uint4*>(smem3);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = blockIdx.y * 64;
    const int ncg = min(8, (p.C - c0) >> 3);
    const int n = blockIdx.x / p.tiles_h, th = blockIdx.x - n * p.tiles_h;
    const int y0 = th * p.TH;
    const int rows = min(p.TH, p.H - y0);
    const int d = p.dil;
    const int R = rows + 2 * d;
    const int Wt = p.W + 2 * d;
    const int cg = tid & 7, pl = tid >> 3;
    const bool cg_ok = cg < ncg;
    const int npix = rows * p.W;

    // MODE 2: this thread's dy vectors go to registers first — their latency hides under the staging
    uint4 dq[kDw3MaxPx];
    if (MODE == 2) {
#pragma unroll
        for (int k = 0; k < kDw3MaxPx; ++k) {
            const int pi = pl + 32 * k;
            dq[k] = make_uint4(0, 0, 0, 0);
            if (cg_ok && pi < npix)
                dq[k] = *reinterpret_cast<const uint4*>(p.in2 + ((long)n * p.H * p.W + (long)y0 * p.W + pi) * p.C + c0 + cg * 8);
        }
    }
    // ---- stage through LDS-DMA: one wave-instruction = 8 pixels x 128 bytes = 1 KiB, linear in LDS.
    // Rows outside the image, channel groups beyond C and the tail pixels get an out-of-range offset:
    // the buffer range check writes zeros.
    {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
        const int c8 = lane & 7;
        const int npx = R * Wt;
        const unsigned img = (unsigned)(((long)n * p.H * p.W * p.C + c0 + c8 * 8) * 2);
        for (int q0 = wave * 8; q0 < npx; q0 += 32) {
            const int q = q0 + (lane >> 3);
            const int r = q / Wt, px = q - r * Wt - d;
            const int iy = y0 - d + r;
            const bool ok = (c8 < ncg) & (q < npx) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)px < (unsigned)p.W);
            const unsigned off = img + (unsigned)((iy * p.W + px) * p.C * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem3 + q0 * 128), 16,
                                                     ok ? off : kOOBd, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (MODE == 0) {
        float w[9][8];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (cg_ok) unpack8(*reinterpret_cast<const uint4*>(p.w9 + (p.flip ? 8 - t : t) * p.C + c0 + cg * 8), w[t]);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) w[t][j] = 0.f;
            }
        }
        bf16_raw* out = reinterpret_cast<bf16_raw*>(p.out) + ((long)n * p.H + y0) * p.W * p.C + c0 + cg * 8;
        int toff[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int q = 0; q < 3; ++q) toff[a * 3 + q] = (a * d * Wt + q * d) * 8;
        float s1[8], s2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
        int y = pl / p.W, x = pl - y * p.W;             // pl < 32: at most a few rows down
        for (int pi = pl; pi < npix; pi += 32) {
            const uint4* t0 = tile + (y * Wt + x) * 8 + cg;
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float xv[8];
                unpack8(t0[toff[t]], xv);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv[j], w[t][j], acc[j]);
            }
            if (cg_ok) *reinterpret_cast<uint4*>(out + (long)pi * p.C) = pack8(acc);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1[j] += acc[j]; s2[j] = fmaf(acc[j], acc[j], s2[j]); }
            x += 32;
            while (x >= p.W) { x -= p.W; ++y; }
        }
        if (p.stats) {                                   // uniform
            __syncthreads();                             // everyone is done with the tile
            float* red = reinterpret_cast<float*>(smem3);       // [32 pixel lanes][8 cg][16]
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red[(pl * 8 + cg) * 16 + j] = s1[j];
                red[(pl * 8 + cg) * 16 + 8 + j] = s2[j];
            }
            __syncthreads();
            if (tid < 128) {
                const int g = tid >> 4, e = tid & 15;
                if (g < ncg) {
                    float v = 0.f;
#pragma unroll 8
                    for (int k = 0; k < 32; ++k) v += red[(k * 8 + g) * 16 + e];
                    float* dst = p.stats + (long)(blockIdx.x % p.stat_reps) * 2 * p.C;
                    atomicAdd(dst + (e >> 3) * p.C + c0 + g * 8 + (e & 7), v);
                }
            }
        }
    } else {
        float acc[9][8];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
        int toff[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int q = 0; q < 3; ++q) toff[a * 3 + q] = (a * d * Wt + q * d) * 8;
        int y = pl / p.W, x = pl - y * p.W;
#pragma unroll
        for (int k = 0; k < kDw3MaxPx; ++k) {
            const int pi = pl + 32 * k;
            if (pi < npix) {
                const uint4* t0 = tile + (y * Wt + x) * 8 + cg;
                float dv[8];
                unpack8(dq[k], dv);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    float xv[8];
                    unpack8(t0[toff[t]], xv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[t][j] = fmaf(dv[j], xv[j], acc[t][j]);
                }
            }
            x += 32;
            while (x >= p.W) { x -= p.W; ++y; }
        }
        // ---- block combine: red[pixel lane][cg][72] over the tile's LDS (the launch allocates >= 73,728 B)
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem3);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[(pl * 8 + cg) * 72 + t * 8 + j] = acc[t][j];
        __syncthreads();
        float* dw = reinterpret_cast<float*>(p.out);
        for (int o = tid; o < 8 * 72; o += 256) {
            const int g = o / 72, e = o - g * 72;
            if (g >= ncg) continue;
            float v = 0.f;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) v += red[(k * 8 + g) * 72 + e];
            const int t = e >> 3, j = e & 7;
            atomicAdd(dw + t * p.C + c0 + g * 8 + j, v);
        }
    }
}

// rows per tile, 0 = shape not taken (then the older kernels run).  Two blocks per CU when the tile fits
// in 80 KiB (one block stages while the other computes), else one block of up to 150 KiB.
inline int dw3_rows(int H, int W, int dil, bool wgrad) {
    const long row_bytes = (long)(W + 2 * dil) * 128;
    long th = (80 * 1024 - 1024) / row_bytes - 2 * dil;
    if (th < 1 || th < dil) th = (150 * 1024) / row_bytes - 2 * dil;
    if (th > H) th = H;
    if (wgrad && th * W > 32 * kDw3MaxPx) th = (32 * kDw3MaxPx) / W;
    if (th < 1) return 0;
    return (int)th;
}

template <int MODE>
inline int launch_dw3(const void* in, const void* in2, const void* w9, void* out, int N, int C, int H, int W, int dil,
                      int flip, int TH, hipStream_t st, float* stats = nullptr, int stat_reps = 1) {
    const long in_bytes = (long)N * H * W * C * 2;
    if (in_bytes >= (1L << 31) - (1L << 24)) return -2;
    Dw3 p{(const bf16_raw*)in, (const bf16_raw*)in2, (const bf16_raw*)w9, out, N, C, H, W, dil, TH, (H + TH - 1) / TH, flip,
          (unsigned)in_bytes, stats, stat_reps < 1 ? 1 : stat_reps};
    const long nb = (long)N * p.tiles_h;
    if (nb > 0x7fffffffL) return -4;
    int lds = (TH + 2 * dil) * (W + 2 * dil) * 128 + 1024;      // + one wave-instruction of slack for the tail
    if (MODE == 2 && lds < 32 * 8 * 72 * 4) lds = 32 * 8 * 72 * 4;
    if (MODE == 0 && stats && lds < 32 * 8 * 16 * 4) lds = 32 * 8 * 16 * 4;
    if (lds > 160 * 1024) return -2;
    auto kfn = dwconv3_kernel<MODE>;
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
    hipLaunchKernelGGL(kfn, dim3((unsigned)nb, (unsigned)((C + 63) / 64)), dim3(256), lds, st, p);
    return DP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// Fourth generation: rolling row window.  The row tiles above re-read their 2d halo rows per tile (TH = 4 at
// d = 2 on the 65-column maps: every element staged TWICE) and a block alternates "wait for the whole tile" /
// "compute" with nothing in flight while it computes: 427 us for a 394 MB tensor, 2.7x its HBM time
// (profiles/r2_x65_os8_b64_v29_profile.md).  Here a block walks DOWN one image (or one of `hsplit` horizontal
// bands) for a 64-channel slab with a ring of NR = 2 TS + 2d rows in LDS: TS + 2d rows feed the TS output rows
// of the current step while LDS-DMA brings the next TS rows in, so every element is fetched once (plus 2d rows
// per band) and the fetch overlaps the arithmetic.  Ring rows are rowpx = ceil8(W + 2d) pixels wide so one
// wave-instruction (8 pixels x 128 B) never crosses into the next ring slot.
// LDS reads of the rolling-window kernels go through inline asm with hand-placed s_waitcnt: hipcc guards every C++
// LDS load behind `s_waitcnt vmcnt(0)` while an LDS-DMA is pending (it cannot tell the slots apart), which made each
// pixel group wait for the NEXT step's rows it had just requested and for the previous pixel's global store.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4d;
__device__ __forceinline__ void dw_lds_rd16(u32x4d& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}
// the listed registers are valid once at most N LDS reads are outstanding (reads return in order)
template <int N>
__device__ __forceinline__ void dw_lds_wait(u32x4d& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void dw_lds_wait(u32x4d& a, u32x4d& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void dw_lds_wait(u32x4d& a, u32x4d& b, u32x4d& c) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void dw_lds_wait(u32x4d& a, u32x4d& b, u32x4d& c, u32x4d& d, u32x4d& e, u32x4d& f) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N) : "memory");
}
__device__ __forceinline__ unsigned wsel(const u32x4d& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }
// at most n (wave-uniform, 0..5) vector-memory operations stay outstanding: the youngest n, i.e. this wave's own
// global stores of the step behind it, not the LDS-DMA issued before them
__device__ __forceinline__ void dw_wait_vm_keep(int n) {
    // the builtin, not inline asm: hipcc's own wait-count bookkeeping sees it (with an opaque asm wait it assumed every
    // load of the previous loop iteration still pending and guarded each register reuse with vmcnt(0)).
    // gfx9 encoding: vmcnt = bits 3:0 (+15:14), expcnt 6:4 and lgkmcnt 11:8 left at their maxima
    if (n >= 5) __builtin_amdgcn_s_waitcnt(0x0F75);
    else if (n == 4) __builtin_amdgcn_s_waitcnt(0x0F74);
    else if (n == 3) __builtin_amdgcn_s_waitcnt(0x0F73);
    else if (n == 2) __builtin_amdgcn_s_waitcnt(0x0F72);
    else if (n == 1) __builtin_amdgcn_s_waitcnt(0x0F71);
    else __builtin_amdgcn_s_waitcnt(0x0F70);
}

constexpr int kDw4K = 5;                // pixels per thread and step: TS * W <= 32 * kDw4K

// The arithmetic is packed bf16 dot products (v_dot2c_f32_bf16: two bf16 products added to an fp32 accumulator)
// instead of "unpack to fp32, fma": v_perm_b32 pairs the SAME channel of two taps (forward: against a packed
// pair of tap weights) or of two pixels (weight gradient: against the same channel of two dy vectors).  72 MACs
// of a pixel cost 72 VALU operations instead of 144 — the kernels were VALU-bound once the row window removed
// the redundant fetches (322 us for a 394 MB tensor with ~185 operations per 16-byte chunk).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2bf(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}
__device__ __forceinline__ unsigned wsel(const uint4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }
__device__ __forceinline__ unsigned pair_lo(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }  // (a.lo, b.lo)
__device__ __forceinline__ unsigned pair_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }  // (a.hi, b.hi)

struct Dw4 {
    const bf16_raw* in; const bf16_raw* in2; const bf16_raw* w9; void* out;
    int N, C, H, W, dil, TS, NR, hsplit, rows_split, flip, rowpx;
    unsigned in_bytes;
    float* stats; int stat_reps;
    // MODE 1 (data gradient with the previous layer's BatchNorm+ReLU backward prologue, as in the dense kernels'
    // store loop): out = (conv + res) * [bna > 0]; stats <- per-channel sums of out and out * xhat,
    // xhat = (bnx - mean) * rstd.  res may be null; bna null = plain "conv + res" (no mask, no sums).
    // bna null with bnx given: the mask is recomputed, bnx * (w * rstd) + (b - mean * w * rstd) > 0 (no residual
    // entered that activation); bn_w / bn_b null = 1 / 0.
    const bf16_raw* res; const bf16_raw* bna; const bf16_raw* bnx; const float* bn_mean; const float* bn_rstd;
    const float* bn_w; const float* bn_b;
    int wsplit = 1, cols_split = 1 << 30;     // column bands (dwconv4 only: maps wider than 32 * kDw4K columns)
};

//   MODE 0: out = depthwise conv of `in` (+ BatchNorm statistics of out);  MODE 1: see Dw4::res;
//   MODE 2: dw[t][c] += sum of in2 (dy) x tap t of `in` (x)
template <int MODE>
__global__ __launch_bounds__(256, 2) void dwconv4_kernel(Dw4 p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem4[];
    const uint4* ring = reinterpret_cast<const uint4*>(smem4);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bxi, byi;
    dw_block_coords(p.N * p.hsplit * p.wsplit, (p.C + 63) >> 6, p.C, bxi, byi);
    const int c0 = byi * 64;
    const int ncg = min(8, (p.C - c0) >> 3);
    // bxi = (image, horizontal band hs, column band ws): maps wider than a step's 32 * kDw4K pixels are cut into
    // column bands of W <= 160 columns (+ d halo columns each side, fetched from the neighbouring band's pixels)
    const int bands = p.hsplit * p.wsplit;
    const int n = bxi / bands, hw = bxi - n * bands;
    const int hs = hw / p.wsplit, ws = hw - hs * p.wsplit;
    const int ya = hs * p.rows_split, yb = min(p.H, ya + p.rows_split);
    const int xa = ws * p.cols_split, Wi = p.W;
    const int W = min(Wi, xa + p.cols_split) - xa;          // columns of this band
    const int d = p.dil, TS = p.TS, NR = p.NR;
    const bool banded = p.wsplit > 1;                       // uniform
    // global pixel index (row-major in the image) of pixel pi of the step starting at row y0
    auto gpix = [&](int y0_, int pi) -> long {
        if (!banded) return (long)y0_ * Wi + pi;
        const int r = pi / W;
        return (long)(y0_ + r) * Wi + xa + (pi - r * W);
    };
    const int cg = tid & 7, pl = tid >> 3;
    const bool cg_ok = cg < ncg;
    const int ipr = p.rowpx >> 3;        // wave-instructions per staged row
    const int rowch = p.rowpx * 8;       // 16-byte chunks per ring row
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
    const int c8 = lane & 7;
    const unsigned img = (unsigned)(((long)n * p.H * Wi * p.C + c0 + c8 * 8) * 2);
    // Every other tensor of the kernel (dy, output, shortcut gradient, BatchNorm activation / input) has the input's
    // shape: per-pixel accesses go through buffer descriptors with an out-of-range offset for lanes without a pixel
    // (loads return zeros, stores are dropped) and a ZERO-SIZED descriptor for an operand that is not there — the
    // pixel loop has no branch around a memory instruction, so hipcc counts the outstanding operations exactly and
    // its s_waitcnt vmcnt(N) never has to assume the worst path.
    auto mkrs = [&](const void* ptr) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, ptr ? (int)p.in_bytes : 0, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rs_in2 = mkrs(MODE == 2 ? (const void*)p.in2 : nullptr);
    const __amdgpu_buffer_rsrc_t rs_out = mkrs(MODE <= 1 ? (const void*)p.out : nullptr);
    const __amdgpu_buffer_rsrc_t rs_res = mkrs(MODE == 1 ? (const void*)p.res : nullptr);
    const __amdgpu_buffer_rsrc_t rs_bna = mkrs(MODE == 1 ? (const void*)p.bna : nullptr);
    const __amdgpu_buffer_rsrc_t rs_bnx = mkrs(MODE == 1 ? (const void*)p.bnx : nullptr);
    const unsigned el0 = (unsigned)((((long)n * p.H * Wi) * p.C + c0 + (tid & 7) * 8) * 2);
    const bool cg_in = (tid & 7) < ncg;
    // byte offset of this thread's 8 channels of pixel pi of the step at row y0_ (out of range: no such pixel)
    auto goff = [&](int y0_, int pi, int npix_) -> unsigned {
        return (cg_in && pi < npix_) ? el0 + (unsigned)(gpix(y0_, pi) * p.C * 2) : kOOBd;
    };
    auto ldb = [&](__amdgpu_buffer_rsrc_t rs, unsigned off) -> uint4 {
        const u32x4d t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
        return make_uint4(t.x, t.y, t.z, t.w);
    };

    // image rows r_first .. r_first + nrows - 1 -> their ring slots (rows outside the image, the horizontal
    // padding and channel groups beyond C get an out-of-range offset: the buffer range check writes zeros)
    auto stage = [&](int r_first, int nrows) {
        for (int i = wave; i < nrows * ipr; i += 4) {
            const int rr = i / ipr, seg = i - rr * ipr;
            const int iy = r_first + rr;
            const int slot = (iy - (ya - d)) % NR;
            const int px = xa + seg * 8 + (lane >> 3) - d;            // image column (the halo may be a neighbour band's)
            const bool ok = (c8 < ncg) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)px < (unsigned)Wi);
            const unsigned off = img + (unsigned)((iy * Wi + px) * p.C * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc, (__attribute__((address_space(3))) void*)(smem4 + ((long)slot * rowch + seg * 64) * 16), 16,
                ok ? off : kOOBd, 0, 0, 0);
        }
    };
    auto load_dy = [&](uint4 (&dq)[kDw4K], int y0) {
        const int npix = min(TS, yb - y0) * W;
#pragma unroll
        for (int k = 0; k < kDw4K; ++k) dq[k] = ldb(rs_in2, goff(y0, pl + 32 * k, npix));
    };

    // MODE 0: tap weights as packed pairs — wp[tp][c] = (w[2 tp][c], w[2 tp + 1][c]); the ninth tap pairs with zero
    unsigned wp[MODE <= 1 ? 4 : 1][8], w8[MODE <= 1 ? 8 : 1];
    float mu[MODE == 1 ? 8 : 1], is[MODE == 1 ? 8 : 1], msc[MODE == 1 ? 8 : 1], msh[MODE == 1 ? 8 : 1];
    uint4 bx[MODE == 1 ? kDw4K : 1], bxn[MODE == 1 ? kDw4K : 1];     // bnx of this / the next step's pixels
    float acc2[MODE == 2 ? 9 : 1][8];
    float s1[8], s2[8];
    uint4 dq[kDw4K], dqn[kDw4K];
    auto load_bx = [&](uint4 (&q)[MODE == 1 ? kDw4K : 1], int y0) {
        const int npix = min(TS, yb - y0) * W;
#pragma unroll
        for (int k = 0; k < (MODE == 1 ? kDw4K : 1); ++k) q[k] = ldb(rs_bnx, goff(y0, pl + 32 * k, npix));
    };
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool on = cg_ok && p.bnx;
            mu[j] = on ? p.bn_mean[c0 + cg * 8 + j] : 0.f;
            is[j] = on ? p.bn_rstd[c0 + cg * 8 + j] : 0.f;
            msc[j] = (on && p.bn_w ? p.bn_w[c0 + cg * 8 + j] : 1.f) * is[j];
            msh[j] = fmaf(-mu[j], msc[j], on && p.bn_b ? p.bn_b[c0 + cg * 8 + j] : 0.f);
        }
        load_bx(bx, ya);          // (a zero-sized descriptor when there is no BatchNorm prologue)
    }
    if (MODE <= 1) {
        uint4 wr[9];
#pragma unroll
        for (int t = 0; t < 9; ++t)
            wr[t] = cg_ok ? *reinterpret_cast<const uint4*>(p.w9 + (p.flip ? 8 - t : t) * p.C + c0 + cg * 8)
                          : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int tp = 0; tp < 4; ++tp)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wp[tp][2 * i] = pair_lo(wsel(wr[2 * tp], i), wsel(wr[2 * tp + 1], i));
                wp[tp][2 * i + 1] = pair_hi(wsel(wr[2 * tp], i), wsel(wr[2 * tp + 1], i));
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) { w8[2 * i] = wsel(wr[8], i) & 0xffffu; w8[2 * i + 1] = wsel(wr[8], i) & 0xffff0000u; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    } else {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc2[t][j] = 0.f;
        load_dy(dq, ya);
    }
    stage(ya - d, TS + 2 * d);
    const long img_el = ((long)n * p.H * Wi) * p.C + c0 + cg * 8;
    bf16_raw* outp = MODE <= 1 ? reinterpret_cast<bf16_raw*>(p.out) + img_el : nullptr;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem4;
    const unsigned qd = (unsigned)d * 128u;          // bytes between the taps of a filter row
    int kept = 0;                                    // global stores this wave issued in the step behind it (uniform)
    for (int y0 = ya; y0 < yb; y0 += TS) {
        // the rows requested a step ago have landed; the stores issued after them may still be in flight
        dw_wait_vm_keep(MODE <= 1 ? kept : 0);
        __syncthreads();          // this step's rows have landed; everyone is done with the rows about to be replaced
        const bool more = y0 + TS < yb;
        if (more) {
            stage(y0 + TS + d, TS);
            if (MODE == 2) load_dy(dqn, y0 + TS);
            if (MODE == 1) load_bx(bxn, y0 + TS);
        }
        const int npix = min(TS, yb - y0) * W;
        const int rbase = (y0 - ya) % NR;
        int ry = pl / W, x = pl - ry * W;
        if (MODE <= 1) {
            // stores of this step, as the next step's wait counts them (every wave issues all of them: lanes
            // without a pixel store to an out-of-range offset)
            kept = kDw4K;
            // MODE 1: the mask / shortcut operands of pixel k + 1 are requested before pixel k is stored, so waiting
            // for them never waits for that store (the counter retires in issue order); two buffers by the parity
            // of k, so that no register is reused while a load into it is in flight
            uint4 gaq[2], grq[2];
            unsigned go_n = goff(y0, pl, npix);
            if (MODE == 1) { gaq[0] = ldb(rs_bna, go_n); grq[0] = ldb(rs_res, go_n); }
#pragma unroll
            for (int k = 0; k < kDw4K; ++k) {
                const int pi = pl + 32 * k;
                const bool ok = pi < npix;
                const unsigned go = go_n;
                if (k + 1 < kDw4K) {
                    go_n = goff(y0, pi + 32, npix);
                    if (MODE == 1) { gaq[(k + 1) & 1] = ldb(rs_bna, go_n); grq[(k + 1) & 1] = ldb(rs_res, go_n); }
                }
                const uint4& ga = gaq[k & 1];
                const uint4& gr = grq[k & 1];
                {
                    const int xe = ok ? x : 0, rye = ok ? ry : 0;     // lanes without a pixel recompute pixel 0
                    int r0 = rbase + rye; r0 -= r0 >= NR ? NR : 0;
                    int r1 = r0 + d;      r1 -= r1 >= NR ? NR : 0;
                    int r2 = r1 + d;      r2 -= r2 >= NR ? NR : 0;
                    const unsigned a0 = lds0 + (unsigned)(r0 * rowch + xe * 8 + cg) * 16u;
                    const unsigned a1 = lds0 + (unsigned)(r1 * rowch + xe * 8 + cg) * 16u;
                    const unsigned a2 = lds0 + (unsigned)(r2 * rowch + xe * 8 + cg) * 16u;
                    u32x4d v[9];
#pragma unroll
                    for (int q = 0; q < 3; ++q) dw_lds_rd16(v[q], a0 + q * qd);
#pragma unroll
                    for (int q = 0; q < 3; ++q) dw_lds_rd16(v[3 + q], a1 + q * qd);
#pragma unroll
                    for (int q = 0; q < 3; ++q) dw_lds_rd16(v[6 + q], a2 + q * qd);
                    float acc[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
                    for (int tp = 0; tp < 4; ++tp) {
                        if (tp == 0) dw_lds_wait<7>(v[0], v[1]);
                        if (tp == 1) dw_lds_wait<5>(v[2], v[3]);
                        if (tp == 2) dw_lds_wait<3>(v[4], v[5]);
                        if (tp == 3) dw_lds_wait<1>(v[6], v[7]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const unsigned a = wsel(v[2 * tp], i), b = wsel(v[2 * tp + 1], i);
                            acc[2 * i] = dot2bf(pair_lo(a, b), wp[tp][2 * i], acc[2 * i]);
                            acc[2 * i + 1] = dot2bf(pair_hi(a, b), wp[tp][2 * i + 1], acc[2 * i + 1]);
                        }
                    }
                    dw_lds_wait<0>(v[8]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[2 * i] = dot2bf(wsel(v[8], i), w8[2 * i], acc[2 * i]);
                        acc[2 * i + 1] = dot2bf(wsel(v[8], i), w8[2 * i + 1], acc[2 * i + 1]);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = ok ? acc[j] : 0.f;      // nothing of a lane without a pixel is summed
                    auto stb = [&](const uint4& o) {
                        u32x4d t; t.x = o.x; t.y = o.y; t.z = o.z; t.w = o.w;
                        __builtin_amdgcn_raw_buffer_store_b128(t, rs_out, (int)go, 0, 0);
                    };
                    if (MODE == 1) {
                        float rv[8];
                        unpack8(gr, rv);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] += rv[j];
                        if (p.bnx) {                     // uniform
                            float av[8], xv[8];
                            unpack8(bx[k], xv);
                            if (p.bna) {                 // uniform
                                unpack8(ga, av);
                            } else {
#pragma unroll
                                for (int j = 0; j < 8; ++j) av[j] = fmaf(xv[j], msc[j], msh[j]);
                            }
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[j] = av[j] > 0.f ? acc[j] : 0.f;
                            const uint4 o = pack8(acc);
                            unpack8(o, acc);             // reduce exactly what BN backward will read back
                            stb(o);
#pragma unroll
                            for (int j = 0; j < 8; ++j) { s1[j] += acc[j]; s2[j] = fmaf(acc[j], (xv[j] - mu[j]) * is[j], s2[j]); }
                        } else {
                            stb(pack8(acc));
                        }
                    } else {
                        stb(pack8(acc));
#pragma unroll
                        for (int j = 0; j < 8; ++j) { s1[j] += acc[j]; s2[j] = fmaf(acc[j], acc[j], s2[j]); }
                    }
                }
                x += 32;
                while (x >= W) { x -= W; ++ry; }
            }
        } else {
            // pixels in pairs (A, B): one dot product = dyA * xA + dyB * xB of one channel and tap
#pragma unroll
            for (int kp = 0; kp < (kDw4K + 1) / 2; ++kp) {
                const int piA = pl + 64 * kp, piB = piA + 32;
                const int ryA = ry, xA = x;
                x += 32;
                while (x >= W) { x -= W; ++ry; }
                const int ryB = ry, xB = x;
                x += 32;
                while (x >= W) { x -= W; ++ry; }
                if (piA < npix) {
                    int a0 = rbase + ryA;  a0 -= a0 >= NR ? NR : 0;
                    int a1 = a0 + d;       a1 -= a1 >= NR ? NR : 0;
                    int a2 = a1 + d;       a2 -= a2 >= NR ? NR : 0;
                    const unsigned tA[3] = {lds0 + (unsigned)(a0 * rowch + xA * 8 + cg) * 16u,
                                            lds0 + (unsigned)(a1 * rowch + xA * 8 + cg) * 16u,
                                            lds0 + (unsigned)(a2 * rowch + xA * 8 + cg) * 16u};
                    const bool pairB = (2 * kp + 1 < kDw4K) && piB < npix;
                    if (pairB) {
                        int b0 = rbase + ryB;  b0 -= b0 >= NR ? NR : 0;
                        int b1 = b0 + d;       b1 -= b1 >= NR ? NR : 0;
                        int b2 = b1 + d;       b2 -= b2 >= NR ? NR : 0;
                        const unsigned tB[3] = {lds0 + (unsigned)(b0 * rowch + xB * 8 + cg) * 16u,
                                                lds0 + (unsigned)(b1 * rowch + xB * 8 + cg) * 16u,
                                                lds0 + (unsigned)(b2 * rowch + xB * 8 + cg) * 16u};
                        const uint4 dA = dq[2 * kp], dB = dq[(2 * kp + 1 < kDw4K) ? 2 * kp + 1 : 2 * kp];
                        unsigned dl[4], dh[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) { dl[i] = pair_lo(wsel(dA, i), wsel(dB, i)); dh[i] = pair_hi(wsel(dA, i), wsel(dB, i)); }
                        // one filter row (3 taps of A, 3 of B) is requested ahead of the row being consumed
                        u32x4d fa[2][3], fb[2][3];
#pragma unroll
                        for (int q = 0; q < 3; ++q) { dw_lds_rd16(fa[0][q], tA[0] + q * qd); dw_lds_rd16(fb[0][q], tB[0] + q * qd); }
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            if (a < 2) {
#pragma unroll
                                for (int q = 0; q < 3; ++q) {
                                    dw_lds_rd16(fa[(a + 1) & 1][q], tA[a + 1] + q * qd);
                                    dw_lds_rd16(fb[(a + 1) & 1][q], tB[a + 1] + q * qd);
                                }
                                dw_lds_wait<6>(fa[a & 1][0], fa[a & 1][1], fa[a & 1][2], fb[a & 1][0], fb[a & 1][1], fb[a & 1][2]);
                            } else {
                                dw_lds_wait<0>(fa[a & 1][0], fa[a & 1][1], fa[a & 1][2], fb[a & 1][0], fb[a & 1][1], fb[a & 1][2]);
                            }
#pragma unroll
                            for (int q = 0; q < 3; ++q) {
                                const u32x4d va = fa[a & 1][q], vb = fb[a & 1][q];
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const unsigned ua = wsel(va, i), ub = wsel(vb, i);
                                    acc2[a * 3 + q][2 * i] = dot2bf(pair_lo(ua, ub), dl[i], acc2[a * 3 + q][2 * i]);
                                    acc2[a * 3 + q][2 * i + 1] = dot2bf(pair_hi(ua, ub), dh[i], acc2[a * 3 + q][2 * i + 1]);
                                }
                            }
                        }
                    } else {
                        const uint4 dA = dq[2 * kp];
                        u32x4d fa[9];
#pragma unroll
                        for (int a = 0; a < 3; ++a)
#pragma unroll
                            for (int q = 0; q < 3; ++q) dw_lds_rd16(fa[a * 3 + q], tA[a] + q * qd);
                        dw_lds_wait<3>(fa[0], fa[1], fa[2], fa[3], fa[4], fa[5]);
                        dw_lds_wait<0>(fa[6], fa[7], fa[8]);
#pragma unroll
                        for (int a = 0; a < 3; ++a)
#pragma unroll
                            for (int q = 0; q < 3; ++q) {
                                const u32x4d va = fa[a * 3 + q];
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const unsigned ua = wsel(va, i), ud = wsel(dA, i);
                                    acc2[a * 3 + q][2 * i] = dot2bf(ua & 0xffffu, ud, acc2[a * 3 + q][2 * i]);
                                    acc2[a * 3 + q][2 * i + 1] = dot2bf(ua & 0xffff0000u, ud, acc2[a * 3 + q][2 * i + 1]);
                                }
                            }
                    }
                }
            }
        }
        if (MODE == 1 && p.bnx) {
#pragma unroll
            for (int k = 0; k < (MODE == 1 ? kDw4K : 1); ++k) bx[k] = bxn[k];
        }
        if (MODE == 2) {
#pragma unroll
            for (int k = 0; k < kDw4K; ++k) dq[k] = dqn[k];
        }
    }
    __syncthreads();                                     // everyone is done with the ring
    if (MODE <= 1) {
        if (p.stats) {                                   // uniform
            float* red = reinterpret_cast<float*>(smem4);       // [32 pixel lanes][8 cg][16]
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red[(pl * 8 + cg) * 16 + j] = s1[j];
                red[(pl * 8 + cg) * 16 + 8 + j] = s2[j];
            }
            __syncthreads();
            if (tid < 128) {
                const int g = tid >> 4, e = tid & 15;
                if (g < ncg) {
                    float v = 0.f;
#pragma unroll 8
                    for (int k = 0; k < 32; ++k) v += red[(k * 8 + g) * 16 + e];
                    float* dst = p.stats + (long)(bxi % p.stat_reps) * 2 * p.C;
                    atomicAdd(dst + (e >> 3) * p.C + c0 + g * 8 + (e & 7), v);
                }
            }
        }
    } else {
        float* red = reinterpret_cast<float*>(smem4);           // [32 pixel lanes][8 cg][72]
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[(pl * 8 + cg) * 72 + t * 8 + j] = acc2[t][j];
        __syncthreads();
        float* dw = reinterpret_cast<float*>(p.out);
        for (int o = tid; o < 8 * 72; o += 256) {
            const int g = o / 72, e = o - g * 72;
            if (g >= ncg) continue;
            float v = 0.f;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) v += red[(k * 8 + g) * 72 + e];
            atomicAdd(dw + (e >> 3) * p.C + c0 + g * 8 + (e & 7), v);
        }
    }
}

struct Dw4Bn { const void* res; const void* bna; const void* bnx; const float* mean; const float* rstd; const float* w; const float* b; };

template <int MODE>
inline int launch_dw4(const void* in, const void* in2, const void* w9, void* out, int N, int C, int H, int W, int dil,
                      int flip, hipStream_t st, float* stats = nullptr, int stat_reps = 1, Dw4Bn bn = Dw4Bn{}) {
    const long in_bytes = (long)N * H * W * C * 2;
    if (in_bytes >= (1L << 31) - (1L << 24)) return -2;
    const int wsplit = (W + 32 * kDw4K - 1) / (32 * kDw4K);          // column bands of at most 160 columns
    const int Wb = (W + wsplit - 1) / wsplit;
    const int rowpx = (Wb + 2 * dil + 7) / 8 * 8;
    const long row_bytes = (long)rowpx * 128;
    int ts_max = (32 * kDw4K) / Wb;
    if (ts_max > 4) ts_max = 4;
    if (ts_max > H) ts_max = H;
    if (ts_max < 1) return -2;
    int TS = ts_max;                                    // two resident blocks per CU when a ring fits in 80 KiB
    while (TS >= 1 && (2 * TS + 2 * dil) * row_bytes > 80 * 1024) --TS;
    if (TS < 1) {
        TS = ts_max;
        while (TS >= 1 && (2 * TS + 2 * dil) * row_bytes > 160 * 1024) --TS;
        if (TS < 1) return -2;
    }
    const int NR = 2 * TS + 2 * dil;
    long lds = NR * row_bytes;
    const int bpc = lds * 2 <= 160 * 1024 ? 2 : 1;
    if (MODE == 2 && lds < 32 * 8 * 72 * 4) lds = 32 * 8 * 72 * 4;
    if (MODE <= 1 && stats && lds < 32 * 8 * 16 * 4) lds = 32 * 8 * 16 * 4;
    if (lds > 160 * 1024) return -2;
    // horizontal bands: the grid should fill whole rounds of the 256 x bpc resident block slots; a band costs its
    // rows plus the 2d halo rows it re-reads
    const long slabs = (C + 63) / 64, slots = 256L * bpc;
    int best_hs = 1;
    long best_cost = -1;
    for (int hs = 1; hs <= 4; ++hs) {
        const int rows = (H + hs - 1) / hs;
        if (hs > 1 && rows < 2 * TS) break;
        const long bands = (H + rows - 1) / rows;
        const long rounds = (N * slabs * bands * wsplit + slots - 1) / slots;
        const long cost = rounds * (rows + 2 * dil);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_hs = hs; }
    }
    const int rows_split = (H + best_hs - 1) / best_hs;
    const int hsplit = (H + rows_split - 1) / rows_split;
    const long nb = (long)N * hsplit * wsplit;
    if (nb > 0x7fffffffL) return -4;
    Dw4 p{(const bf16_raw*)in, (const bf16_raw*)in2, (const bf16_raw*)w9, out, N, C, H, W, dil, TS, NR, hsplit, rows_split,
          flip, rowpx, (unsigned)in_bytes, stats, stat_reps < 1 ? 1 : stat_reps,
          (const bf16_raw*)bn.res, (const bf16_raw*)bn.bna, (const bf16_raw*)bn.bnx, bn.mean, bn.rstd, bn.w, bn.b};
    p.wsplit = wsplit;
    p.cols_split = Wb;
    auto kfn = dwconv4_kernel<MODE>;
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
    if (nb * slabs > 0x7fffffffL) return -4;
    hipLaunchKernelGGL(kfn, dim3((unsigned)(nb * slabs)), dim3(256), (size_t)lds, st, p);
    return DP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// Fifth generation, weight gradient on the MATRIX cores.  dw[t][c] = sum_p dy[p][c] * x[p + t][c] is, for a group
// of 16 channels and one tap, the DIAGONAL of the 16x16 product  D[c][c'] = sum_p dy[p][c] * x[p + t][c']  — a GEMM
// over the pixel index.  15/16 of the MFMA work is thrown away and it is still ~5x cheaper than the packed dot
// products of dwconv4 MODE 2 (9 x v_mfma_f32_16x16x16_bf16 = 72 cycles per 16 pixels x 16 channels against 72
// v_perm + 72 v_dot2c for the same), which leaves the kernel where a reduction over two tensors belongs: on the
// HBM stream (dwconv4 MODE 2: 606 us per 728-channel 65x65 layer at batch 64 for 788 MB).
//   * x walks through the same rolling row ring as dwconv4; dy gets a ring of its own (2 TS rows) with the SAME row
//     geometry (rowpx pixels, d columns of physical zero padding on the left), so one pixel index addresses both.
//   * both operands are pixel-major in LDS ([pixel][64 channels], as they lie in memory) and are needed
//     channel-major by the MFMA: ds_read_b64_tr_b16 delivers 4 pixels of one channel per lane.  16-byte chunks are
//     XOR-swizzled by (pixel & 7) through the DMA's SOURCE address (the destination stays lane-linear), which makes
//     the transposed reads of 8 consecutive pixels conflict-free.
//   * a row is cut into 16-pixel K chunks; the last chunk of a row whose width is 8 mod 16 overlaps its
//     predecessor by 8 pixels and zeroes those in the dy fragment.  Padding columns hold dy = 0, so x may be
//     anything finite there (guard bands before / after the rings are zeroed once).
//   * 8 waves: wave & 3 = the 16-channel group of the 64-channel slab, wave >> 2 = which of the TS = 2 rows of a step.
// MEASURED (gpurun_out/ak, 728 channels, 65x65, batch 64, launchers timed alone): 318 us against 222 us for dwconv4
// MODE 2 — which in isolation already runs at 3.55 TB/s; its 606 us in the training step is the low-priority
// auxiliary stream sharing the chip, not arithmetic.  The two rings (x and dy) take 112 KiB, so ONE block per CU,
// and a block that alternates "wait for the rows" / "read + MFMA" has nothing else on its CU to fill the waits:
// 2.5 TB/s.  Kept selectable (DPAMD_DWCONV_GEN=5) and tested; not the default.
typedef short s16x4d __attribute__((ext_vector_type(4)));
typedef float f32x4d __attribute__((ext_vector_type(4)));

struct Dw5W {
    const bf16_raw* x; const bf16_raw* dy; float* dw;
    int N, C, H, W, dil, NR, hsplit, rows_split, rowpx;
    unsigned bytes;
};

__global__ __launch_bounds__(512) void dwconv5_wgrad_kernel(Dw5W p) {
    constexpr int TS = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem5[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cgw = wave & 3, rs = wave >> 2;
    const int c0 = blockIdx.y * 64;
    const int ncg = min(8, (p.C - c0) >> 3);
    const int n = blockIdx.x / p.hsplit, hs = blockIdx.x - n * p.hsplit;
    const int ya = hs * p.rows_split, yb = min(p.H, ya + p.rows_split);
    const int d = p.dil, W = p.W, NR = p.NR;
    const int rowb = p.rowpx * 128, ipr = p.rowpx >> 3;
    unsigned char* xring = smem5 + 1024;
    unsigned char* dring = xring + NR * rowb;
    // Reads of x run up to d pixels past either end of a ring row (against dy = 0): everything they can reach — the
    // guard bands, and ring slots no DMA has filled yet — must hold FINITE values, so the whole allocation is
    // zeroed once (a block lives for a whole image band).
    for (int i = tid * 16; i < 2048 + (NR + 2 * TS) * rowb; i += 512 * 16)
        *reinterpret_cast<uint4*>(smem5 + i) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.bytes, 0x00020000);
    const int c8s = (lane & 7) ^ (lane >> 3);          // logical 16-byte chunk this lane fetches into slot lane & 7
    const unsigned img = (unsigned)(((long)n * p.H * W * p.C + c0 + c8s * 8) * 2);
    auto stage = [&](__amdgpu_buffer_rsrc_t rsrc, unsigned char* ring, int r_first, int nrows, int row0, int nslots) {
        for (int i = wave; i < nrows * ipr; i += 8) {
            const int rr = i / ipr, seg = i - rr * ipr;
            const int iy = r_first + rr;
            const int slot = (iy - row0) % nslots;
            const int px = seg * 8 + (lane >> 3) - d;
            const bool ok = (c8s < ncg) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)px < (unsigned)W);
            const unsigned off = img + (unsigned)((iy * W + px) * p.C * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc, (__attribute__((address_space(3))) void*)(ring + (long)slot * rowb + seg * 1024), 16,
                ok ? off : kOOBd, 0, 0, 0);
        }
    };
    // transposed-read geometry: lane 4q + pp of 16-lane group gq supplies pixel 4 gq + q, channels 4 pp .. 4 pp + 3
    const int gq = lane >> 4, tl = lane & 15, q = tl >> 2, pp = tl & 3;
    const int kpix = 4 * gq + q;
    const int chunk = 2 * cgw + (pp >> 1);
    const unsigned lane_dy = (unsigned)(kpix * 128 + ((chunk ^ (kpix & 7)) << 4) + ((pp & 1) << 3));
    unsigned lane_x[3];
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2)
        lane_x[s2] = (unsigned)(kpix * 128 + ((chunk ^ ((kpix + (s2 - 1) * d) & 7)) << 4) + ((pp & 1) << 3));
    f32x4d acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4d{0.f, 0.f, 0.f, 0.f};
    const int nch = (p.rowpx + 15) >> 4;

    stage(rx, xring, ya - d, TS + 2 * d, ya - d, NR);
    stage(rd, dring, ya, TS, ya, 2 * TS);
    for (int y0 = ya; y0 < yb; y0 += TS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();          // this step's rows have landed; everyone is done with the rows about to be replaced
        if (y0 + TS < yb) {
            stage(rx, xring, y0 + TS + d, TS, ya - d, NR);
            stage(rd, dring, y0 + TS, TS, ya, 2 * TS);
        }
        const int y = y0 + rs;
        if (y < yb) {                                        // wave-uniform: EXEC stays full for the transposed reads
            const unsigned char* drow = dring + ((y - ya) % (2 * TS)) * rowb;
            const unsigned char* xrow[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) xrow[r] = xring + ((y - ya + r * d) % NR) * rowb;
            for (int ch = 0; ch < nch; ++ch) {
                int px0 = ch * 16;
                const bool overlap = px0 + 16 > p.rowpx;     // uniform
                if (overlap) px0 = p.rowpx - 16;
                s16x4d a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4d*)(drow + px0 * 128 + lane_dy));
                if (overlap && lane < 32) a = s16x4d{0, 0, 0, 0};      // pixels the previous chunk already took
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s2 = 0; s2 < 3; ++s2) {
                        const s16x4d b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4d*)(xrow[r] + (px0 + (s2 - 1) * d) * 128 + lane_x[s2]));
                        acc[r * 3 + s2] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc[r * 3 + s2], 0, 0, 0);
                    }
            }
        }
    }
    // the diagonal: lane (gq, tl) holds D[4 gq + j][tl] in register j, so channel tl sits in the lanes with tl >> 2 == gq
    if ((tl >> 2) == gq) {
        const int c = c0 + 16 * cgw + tl;
        if (c < p.C) {
            const int j = tl & 3;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float v = j == 0 ? acc[t][0] : (j == 1 ? acc[t][1] : (j == 2 ? acc[t][2] : acc[t][3]));
                atomicAdd(p.dw + t * p.C + c, v);
            }
        }
    }
}

inline int launch_dw5w(const void* x, const void* dy, void* dw, int N, int C, int H, int W, int dil, hipStream_t st) {
    constexpr int TS = 2;
    const long bytes = (long)N * H * W * C * 2;
    if (bytes >= (1L << 31) - (1L << 24) || H < TS) return -2;
    const int rowpx = (W + 2 * dil + 7) / 8 * 8;
    if (rowpx < 16) return -2;
    const int NR = 2 * TS + 2 * dil;
    const long lds = 2048 + (long)(NR + 2 * TS) * rowpx * 128;
    if (lds > 160 * 1024) return -2;
    const long slabs = (C + 63) / 64, slots = 256;
    int best_hs = 1;
    long best_cost = -1;
    for (int hs = 1; hs <= 4; ++hs) {
        const int rows = (H + hs - 1) / hs;
        if (hs > 1 && rows < 4 * TS) break;
        const long bands = (H + rows - 1) / rows;
        const long rounds = (N * slabs * bands + slots - 1) / slots;
        const long cost = rounds * (rows + 2 * dil);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_hs = hs; }
    }
    const int rows_split = ((H + best_hs - 1) / best_hs + TS - 1) / TS * TS;     // bands start on a step boundary
    const int hsplit = (H + rows_split - 1) / rows_split;
    const long nb = (long)N * hsplit;
    if (nb > 0x7fffffffL) return -4;
    Dw5W p{(const bf16_raw*)x, (const bf16_raw*)dy, (float*)dw, N, C, H, W, dil, NR, hsplit, rows_split, rowpx,
           (unsigned)bytes};
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)dwconv5_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    hipLaunchKernelGGL(dwconv5_wgrad_kernel, dim3((unsigned)nb, (unsigned)slabs), dim3(512), (size_t)lds, st, p);
    return DP_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// Generation 6: forward / data gradient on the MATRIX cores, same rolling row ring and two blocks per CU as
// generation 4.  For a group of 16 channels the depthwise sum  out[p][c] = sum_t w[t][c] x[p + t][c]  is the dense
// product  D[c][p] = sum_(t, c') A[c][(t, c')] B[(t, c')][p]  with the DIAGONAL weight matrix
// A[c][(t, c')] = w[t][c] [c' == c]  and  B[(t, c')][p] = x[p + t][c'] : K = 9 taps x 16 channels = 144 (+16 of
// padding) = 5 x v_mfma_f32_16x16x32_bf16 per 16 pixels x 16 channels, 80 MFMA cycles for 256 outputs against ~115
// VALU instructions per 8 outputs of a lane in dwconv4 (which is VALU-bound: 295 us for a 728-channel 65x65 layer at
// batch 64, 2.7 TB/s).
//   * B fragment of lane (pixel j = lane & 15, g = lane >> 4) for MFMA m: the 8 channels (g & 1) * 8 .. + 7 of the
//     wave's 16-channel group at pixel x0 + j shifted by tap 2 m + (g >> 1): ONE ds_read_b128 from the ring.  The
//     16-byte chunks of a pixel are XOR-swizzled by (pixel & 7) through the DMA's source address, which makes the 16
//     lanes x 2 chunk columns of a read group hit 16 different bank quads.
//   * A fragments (5 x 4 VGPRs per wave) are built once: one nonzero bf16 per lane.
//   * wave = 16-channel group of the 64-channel slab; a lane ends up with 4 consecutive channels of one pixel
//     (8-byte stores; the four waves of a block complete a pixel's 128 bytes).
//   MODE 0: out = conv (+ BatchNorm statistics of out); MODE 1: data gradient with the skip gradient and the previous
//   BatchNorm+ReLU's backward prologue, as dwconv4 MODE 1 (see Dw4).
template <int MODE>
__global__ __launch_bounds__(256, 2) void dwconv6_kernel(Dw4 p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];
    typedef __attribute__((ext_vector_type(4))) float f32x4w;
    typedef __attribute__((ext_vector_type(4))) int i32x4w;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bxi, byi;
    dw_block_coords(p.N * p.hsplit, (p.C + 63) >> 6, p.C, bxi, byi);
    const int c0 = byi * 64;
    const int ncg = min(8, (p.C - c0) >> 3);
    const int n = bxi / p.hsplit, hs = bxi - n * p.hsplit;
    const int ya = hs * p.rows_split, yb = min(p.H, ya + p.rows_split);
    const int d = p.dil, W = p.W, TS = p.TS, NR = p.NR;
    const int ipr = p.rowpx >> 3;        // wave-instructions per staged row
    const int rowb = p.rowpx * 128;      // bytes per ring row
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
    const int c8s = (lane & 7) ^ (lane >> 3);          // logical 16-byte chunk this lane fetches into slot lane & 7
    const unsigned img = (unsigned)(((long)n * p.H * W * p.C + c0 + c8s * 8) * 2);
    auto stage = [&](int r_first, int nrows) {
        for (int i = wave; i < nrows * ipr; i += 4) {
            const int rr = i / ipr, seg = i - rr * ipr;
            const int iy = r_first + rr;
            const int slot = (iy - (ya - d)) % NR;
            const int px = seg * 8 + (lane >> 3) - d;
            const bool ok = (c8s < ncg) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)px < (unsigned)W);
            const unsigned off = img + (unsigned)((iy * W + px) * p.C * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsrc, (__attribute__((address_space(3))) void*)(smem6 + (long)slot * rowb + seg * 1024), 16,
                ok ? off : kOOBd, 0, 0, 0);
        }
    };
    // ---- lane geometry ----
    const int j = lane & 15, g = lane >> 4;
    const int cw = c0 + 16 * wave;                        // first channel of this wave's group
    const int chunk = 2 * wave + (g & 1);                 // logical 16-byte chunk of the B fragment
    // A fragments: lane (i = j, g) holds A[i][k = 8 g + e]; nonzero only at e = i & 7 when (i >> 3) == (g & 1)
    i32x4w afr[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const int tap = 2 * m + (g >> 1);
        unsigned wv = 0;
        if (tap < 9 && (j >> 3) == (g & 1) && cw + j < p.C)
            wv = (unsigned)p.w9[(p.flip ? 8 - tap : tap) * p.C + cw + j];
        wv <<= 16 * (j & 1);
        const int dd = (j & 7) >> 1;
        afr[m] = i32x4w{dd == 0 ? (int)wv : 0, dd == 1 ? (int)wv : 0, dd == 2 ? (int)wv : 0, dd == 3 ? (int)wv : 0};
    }
    // B read offsets inside a ring row, per MFMA: pixel shift (s - 1) d and row r of tap 2 m + (g >> 1) (tap 9 = tap 8
    // again: its weights are zero); the swizzle term depends on (j + shift) & 7 only (x0 is a multiple of 16)
    int brow[5];                                          // tap row 0 / 1 / 2 of MFMA m's fragment
    unsigned boff[5];
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        int tap = 2 * m + (g >> 1);
        tap = tap > 8 ? 8 : tap;
        const int r = tap / 3, s2 = tap - 3 * r;
        brow[m] = r;
        const int px = j + s2 * d;                        // ring pixel (left padding d) of output pixel j, tap column s2
        boff[m] = (unsigned)(px * 128 + ((chunk ^ (px & 7)) << 4));
    }
    // per-lane channel quad (4 consecutive channels cq .. cq + 3 of pixel j)
    const int cq = cw + 4 * g;
    const bool cq_ok = cq < p.C;                          // C % 8 == 0 and cq % 4 == 0: a quad is valid as a whole ... or not
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2a[4] = {0.f, 0.f, 0.f, 0.f};
    float mu[4], is[4], msc[4], msh[4];
    if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool on = cq_ok && p.bnx;
            mu[e] = on ? p.bn_mean[cq + e] : 0.f;
            is[e] = on ? p.bn_rstd[cq + e] : 0.f;
            msc[e] = (on && p.bn_w ? p.bn_w[cq + e] : 1.f) * is[e];
            msh[e] = fmaf(-mu[e], msc[e], on && p.bn_b ? p.bn_b[cq + e] : 0.f);
        }
    }
    const int nseg = (W + 15) >> 4;
    bf16_raw* outp = reinterpret_cast<bf16_raw*>(p.out);
    const long img_px = (long)n * p.H * W;

    stage(ya - d, TS + 2 * d);
    for (int y0 = ya; y0 < yb; y0 += TS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();          // this step's rows have landed; everyone is done with the rows about to be replaced
        if (y0 + TS < yb) stage(y0 + TS + d, TS);
        const int nrow = min(TS, yb - y0);
        for (int ry = 0; ry < nrow; ++ry) {
            const int rb = (y0 - ya) + ry;                 // ring row index (mod NR) of tap row 0
            const int sl0 = (rb % NR) * rowb, sl1 = ((rb + d) % NR) * rowb, sl2 = ((rb + 2 * d) % NR) * rowb;   // uniform
            unsigned badr[5];
#pragma unroll
            for (int m = 0; m < 5; ++m) badr[m] = (unsigned)(brow[m] == 0 ? sl0 : (brow[m] == 1 ? sl1 : sl2)) + boff[m];
            for (int sg = 0; sg < nseg; ++sg) {
                const int x0 = sg * 16;
                f32x4w acc = f32x4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < 5; ++m) {
                    const i32x4w b = *reinterpret_cast<const i32x4w*>(smem6 + x0 * 128 + badr[m]);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, afr[m]),
                                                                  __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
                }
                const int x = x0 + j;
                const bool ok = cq_ok && x < W;
                const long e0 = (img_px + (long)(y0 + ry) * W + x) * p.C + cq;
                float v[4] = {acc[0], acc[1], acc[2], acc[3]};
                if (MODE == 0) {
                    if (ok) {
                        uint2 o;
                        o.x = pack2(v[0], v[1]); o.y = pack2(v[2], v[3]);
                        *reinterpret_cast<uint2*>(outp + e0) = o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { s1[e] += v[e]; s2a[e] = fmaf(v[e], v[e], s2a[e]); }
                    }
                } else if (ok) {
                    if (p.res) {
                        const uint2 r2 = *reinterpret_cast<const uint2*>(p.res + e0);
                        v[0] += bf2f((bf16_raw)(r2.x & 0xffffu)); v[1] += bf2f((bf16_raw)(r2.x >> 16));
                        v[2] += bf2f((bf16_raw)(r2.y & 0xffffu)); v[3] += bf2f((bf16_raw)(r2.y >> 16));
                    }
                    if (p.bnx) {
                        const uint2 x2 = *reinterpret_cast<const uint2*>(p.bnx + e0);
                        const float xv[4] = {bf2f((bf16_raw)(x2.x & 0xffffu)), bf2f((bf16_raw)(x2.x >> 16)),
                                             bf2f((bf16_raw)(x2.y & 0xffffu)), bf2f((bf16_raw)(x2.y >> 16))};
                        float av[4];
                        if (p.bna) {
                            const uint2 a2 = *reinterpret_cast<const uint2*>(p.bna + e0);
                            av[0] = bf2f((bf16_raw)(a2.x & 0xffffu)); av[1] = bf2f((bf16_raw)(a2.x >> 16));
                            av[2] = bf2f((bf16_raw)(a2.y & 0xffffu)); av[3] = bf2f((bf16_raw)(a2.y >> 16));
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) av[e] = fmaf(xv[e], msc[e], msh[e]);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = av[e] > 0.f ? v[e] : 0.f;
                        uint2 o;
                        o.x = pack2(v[0], v[1]); o.y = pack2(v[2], v[3]);
                        *reinterpret_cast<uint2*>(outp + e0) = o;
                        // reduce exactly what BN backward will read back (bf16-rounded)
                        const float rv[4] = {bf2f((bf16_raw)(o.x & 0xffffu)), bf2f((bf16_raw)(o.x >> 16)),
                                             bf2f((bf16_raw)(o.y & 0xffffu)), bf2f((bf16_raw)(o.y >> 16))};
#pragma unroll
                        for (int e = 0; e < 4; ++e) { s1[e] += rv[e]; s2a[e] = fmaf(rv[e], (xv[e] - mu[e]) * is[e], s2a[e]); }
                    } else {
                        uint2 o;
                        o.x = pack2(v[0], v[1]); o.y = pack2(v[2], v[3]);
                        *reinterpret_cast<uint2*>(outp + e0) = o;
                    }
                }
            }
        }
    }
    if (p.stats) {                                         // uniform
        // fold the 16 pixel lanes of a row (DPP), one atomic pair per channel and wave
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[e] = dw_row16_sum(s1[e]); s2a[e] = dw_row16_sum(s2a[e]); }
        if (j == 0 && cq_ok) {
            float* dst = p.stats + (long)(bxi % p.stat_reps) * 2 * p.C;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                atomicAdd(dst + cq + e, s1[e]);
                atomicAdd(dst + p.C + cq + e, s2a[e]);
            }
        }
    }
}

template <int MODE>
inline int launch_dw6(const void* in, const void* w9, void* out, int N, int C, int H, int W, int dil, int flip,
                      hipStream_t st, float* stats = nullptr, int stat_reps = 1, Dw4Bn bn = Dw4Bn{}) {
    const long in_bytes = (long)N * H * W * C * 2;
    if (in_bytes >= (1L << 31) - (1L << 24) || W < 8) return -2;
    const int rowpx = (W + 2 * dil + 7) / 8 * 8;
    const long row_bytes = (long)rowpx * 128;
    // the last 16-pixel segment of a row may read past the row's end — into the next ring slot, or, from the last
    // slot, into this guard band; what it reads there only reaches output pixels >= W, which are not stored
    const long guard = (long)(16 + 2 * dil) * 128;
    int TS = 4 < H ? 4 : H;                             // two resident blocks per CU when a ring fits in 80 KiB
    while (TS >= 1 && (2 * TS + 2 * dil) * row_bytes + guard > 80 * 1024) --TS;
    if (TS < 1) return -2;                              // generation 4 takes the wide maps (one block per CU)
    const int NR = 2 * TS + 2 * dil;
    const long lds = NR * row_bytes + guard;
    const long slabs = (C + 63) / 64, slots = 256L * 2;
    int best_hs = 1;
    long best_cost = -1;
    for (int hs = 1; hs <= 4; ++hs) {
        const int rows = (H + hs - 1) / hs;
        if (hs > 1 && rows < 2 * TS) break;
        const long bands = (H + rows - 1) / rows;
        const long rounds = (N * slabs * bands + slots - 1) / slots;
        const long cost = rounds * (rows + 2 * dil);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_hs = hs; }
    }
    const int rows_split = (H + best_hs - 1) / best_hs;
    const int hsplit = (H + rows_split - 1) / rows_split;
    const long nb = (long)N * hsplit;
    if (nb > 0x7fffffffL) return -4;
    Dw4 p{(const bf16_raw*)in, nullptr, (const bf16_raw*)w9, out, N, C, H, W, dil, TS, NR, hsplit, rows_split,
          flip, rowpx, (unsigned)in_bytes, stats, stat_reps < 1 ? 1 : stat_reps,
          (const bf16_raw*)bn.res, (const bf16_raw*)bn.bna, (const bf16_raw*)bn.bnx, bn.mean, bn.rstd, bn.w, bn.b};
    auto kfn = dwconv6_kernel<MODE>;
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); once = true; }
    if (nb * slabs > 0x7fffffffL) return -4;
    hipLaunchKernelGGL(kfn, dim3((unsigned)(nb * slabs)), dim3(256), (size_t)lds, st, p);
    return DP_LAUNCH_STATUS();
}

// 4 = rolling row window (default); 5 = as 4, with the weight gradient on the matrix cores (opt-in: measured
// slower, see dwconv5_wgrad_kernel); 3 = LDS row tiles, 2 / 1 = the strip / flat mappings
static int kDwGen = getenv("DPAMD_DWCONV_GEN") ? atoi(getenv("DPAMD_DWCONV_GEN")) : 4;

}  // namespace

// tests / tools: select the kernel generation at run time (returns the previous one)
DP_EXPORT int dp_dwconv_set_gen(int gen) {
    const int old = kDwGen;
    if (gen >= 1 && gen <= 7) kDwGen = gen;
    return old;
}

DP_EXPORT int dp_dwconv_fwd(const void* x, const void* w9, void* y, int N, int C, int Hi, int Wi, int Ho, int Wo,
                            int stride, int pad_h, int pad_w, int dil, hipStream_t st) {
    if (C % 8 != 0 || N <= 0 || stride < 1 || dil < 1) return -1;
    if (kDwGen >= 3 && stride == 1 && pad_h == dil && pad_w == dil && Ho == Hi && Wo == Wi) {
        if (kDwGen >= 6) {
            const int rc = launch_dw6<0>(x, w9, y, N, C, Hi, Wi, dil, 0, st);
            if (rc != -2) return rc;
        }
        if (kDwGen >= 4) {
            const int rc = launch_dw4<0>(x, nullptr, w9, y, N, C, Hi, Wi, dil, 0, st);
            if (rc != -2) return rc;
        }
        const int th = dw3_rows(Hi, Wi, dil, false);
        if (th > 0) {
            const int rc = launch_dw3<0>(x, nullptr, w9, y, N, C, Hi, Wi, dil, 0, th, st);
            if (rc != -2) return rc;
        }
    }
    Dw p{N, C, Hi, Wi, Ho, Wo, stride, pad_h, pad_w, dil};
    // strip mapping measured SLOWER here than the flat mapping (176 vs 111 us on the 728-channel 65x65
    // layers: a wave's loads split into two 512-B runs and a thread walks 8 pixels serially); it pays
    // only in bwd_weight, where it removes the per-pixel register spill of partial sums.  Opt-in: =2.
    static const bool gen2 = getenv("DPAMD_DWCONV_GEN") && atoi(getenv("DPAMD_DWCONV_GEN")) == 2;
    const long P = (long)N * Ho * Wo;
    const long nb = (P + kStrips * 8 - 1) / (kStrips * 8);
    if (gen2 && nb <= 0x7fffffffL) {
        hipLaunchKernelGGL(dwconv2_fwd_kernel<8>, dim3((unsigned)nb, (unsigned)((C / 8 + kSlab - 1) / kSlab)), dim3(256),
                           0, st, (const bf16_raw*)x, (const bf16_raw*)w9, (bf16_raw*)y, p);
        return DP_LAUNCH_STATUS();
    }
    hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(grid_1d((long)N * Ho * Wo * (C / 8), 256)), dim3(256), 0, st,
                       (const bf16_raw*)x, (const bf16_raw*)w9, (bf16_raw*)y, p);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_dwconv_bwd_data(const void* dy, const void* w9, void* dx, int N, int C, int Hi, int Wi, int Ho,
                                 int Wo, int stride, int pad_h, int pad_w, int dil, hipStream_t st) {
    if (C % 8 != 0 || N <= 0 || stride < 1 || dil < 1) return -1;
    if (kDwGen >= 3 && stride == 1 && pad_h == dil && pad_w == dil && Ho == Hi && Wo == Wi) {
        if (kDwGen >= 6) {                                           // = forward over dy with the taps flipped
            const int rc = launch_dw6<0>(dy, w9, dx, N, C, Hi, Wi, dil, 1, st);
            if (rc != -2) return rc;
        }
        if (kDwGen >= 4) {
            const int rc = launch_dw4<0>(dy, nullptr, w9, dx, N, C, Hi, Wi, dil, 1, st);
            if (rc != -2) return rc;
        }
        const int th = dw3_rows(Hi, Wi, dil, false);
        if (th > 0) {
            const int rc = launch_dw3<0>(dy, nullptr, w9, dx, N, C, Hi, Wi, dil, 1, th, st);
            if (rc != -2) return rc;
        }
    }
    Dw p{N, C, Hi, Wi, Ho, Wo, stride, pad_h, pad_w, dil};
    static const bool gen2 = getenv("DPAMD_DWCONV_GEN") && atoi(getenv("DPAMD_DWCONV_GEN")) == 2;   // see dp_dwconv_fwd
    const long Pin = (long)N * Hi * Wi;
    const long nbs = (Pin + kStrips * 8 - 1) / (kStrips * 8);
    if (gen2 && stride == 1 && nbs <= 0x7fffffffL) {
        hipLaunchKernelGGL(dwconv2_bwd_data_kernel<8>, dim3((unsigned)nbs, (unsigned)((C / 8 + kSlab - 1) / kSlab)),
                           dim3(256), 0, st, (const bf16_raw*)dy, (const bf16_raw*)w9, (bf16_raw*)dx, p);
        return DP_LAUNCH_STATUS();
    }
    hipLaunchKernelGGL(dwconv_bwd_data_kernel, dim3(grid_1d((long)N * Hi * Wi * (C / 8), 256)), dim3(256), 0, st,
                       (const bf16_raw*)dy, (const bf16_raw*)w9, (bf16_raw*)dx, p);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_dwconv_bwd_weight(const void* dy, const void* x, void* dw9, int N, int C, int Hi, int Wi, int Ho,
                                   int Wo, int stride, int pad_h, int pad_w, int dil, hipStream_t st) {
    if (C % 8 != 0 || N <= 0 || stride < 1 || dil < 1) return -1;
    if (kDwGen >= 3 && stride == 1 && pad_h == dil && pad_w == dil && Ho == Hi && Wo == Wi) {
        if (kDwGen == 5) {
            const int rc = launch_dw5w(x, dy, dw9, N, C, Hi, Wi, dil, st);
            if (rc != -2) return rc;
        }
        if (kDwGen >= 4) {
            const int rc = launch_dw4<2>(x, dy, nullptr, dw9, N, C, Hi, Wi, dil, 0, st);
            if (rc != -2) return rc;
        }
        const int th = dw3_rows(Hi, Wi, dil, true);
        if (th > 0) {
            const int rc = launch_dw3<2>(x, dy, nullptr, dw9, N, C, Hi, Wi, dil, 0, th, st);
            if (rc != -2) return rc;
        }
    }
    Dw p{N, C, Hi, Wi, Ho, Wo, stride, pad_h, pad_w, dil};
    const int C8 = C / 8;
    static const bool gen1 = getenv("DPAMD_DWCONV_GEN") && atoi(getenv("DPAMD_DWCONV_GEN")) == 1;
    {
        const long P = (long)N * Ho * Wo;
        const long nbw = (P + kStrips * 32 - 1) / (kStrips * 32);
        if (!gen1 && nbw <= 0x7fffffffL) {
            hipLaunchKernelGGL(dwconv2_bwd_weight_kernel<32>, dim3((unsigned)nbw, (unsigned)((C8 + kSlab - 1) / kSlab)),
                               dim3(256), 0, st, (const bf16_raw*)dy, (const bf16_raw*)x, (float*)dw9, p);
            return DP_LAUNCH_STATUS();
        }
    }
    const int cgs = C8 < 256 ? C8 : 256;
    const int rpp = 256 / cgs;
    long nb = ((long)N * Ho * Wo + (long)rpp * 8 - 1) / ((long)rpp * 8);
    if (nb < 1) nb = 1;
    if (nb > 512) nb = 512;
    hipLaunchKernelGGL(dwconv_bwd_weight_kernel, dim3((unsigned)nb, (unsigned)((C8 + 255) / 256)), dim3(256), 0, st,
                       (const bf16_raw*)dy, (const bf16_raw*)x, (float*)dw9, p);
    return DP_LAUNCH_STATUS();
}

// Forward with fused BatchNorm statistics: stats fp32 [stat_reps][2][C], zero-initialised, receives the
// per-channel sums of y and y^2.  Returns -2 when the LDS row-tiled kernel does not take the shape (the
// caller then runs dp_dwconv_fwd and a separate statistics pass).
DP_EXPORT int dp_dwconv_fwd_stats(const void* x, const void* w9, void* y, float* stats, int stat_reps, int N, int C,
                                  int Hi, int Wi, int Ho, int Wo, int stride, int pad_h, int pad_w, int dil,
                                  hipStream_t st) {
    if (C % 8 != 0 || N <= 0 || stride < 1 || dil < 1 || !stats || stat_reps < 1) return -1;
    if (!(kDwGen >= 3 && stride == 1 && pad_h == dil && pad_w == dil && Ho == Hi && Wo == Wi)) return -2;
    if (kDwGen >= 6) {
        const int rc = launch_dw6<0>(x, w9, y, N, C, Hi, Wi, dil, 0, st, stats, stat_reps);
        if (rc != -2) return rc;
    }
    if (kDwGen >= 4) {
        const int rc = launch_dw4<0>(x, nullptr, w9, y, N, C, Hi, Wi, dil, 0, st, stats, stat_reps);
        if (rc != -2) return rc;
    }
    const int th = dw3_rows(Hi, Wi, dil, false);
    if (th <= 0) return -2;
    return launch_dw3<0>(x, nullptr, w9, y, N, C, Hi, Wi, dil, 0, th, st, stats, stat_reps);
}

// Data gradient (stride 1, "same" padding) with the skip-path gradient `res` (or null) added and, when bnx is given,
// the previous layer's BatchNorm+ReLU backward prologue: dx = (convT(dy) + res) * mask and sums fp32 [reps][2][C]
// (zero-initialised) += per-channel (sum dx, sum dx * (bnx - mean) * rstd).  mask = [bna > 0] when bna (the
// activation itself) is given, else [bnx * w * rstd + b - mean * w * rstd > 0] (bn_w / bn_b may be null: 1 / 0).
// Returns -2 when the rolling-window kernel does not take the shape (nothing was launched).
DP_EXPORT int dp_dwconv_bwd_data_fused(const void* dy, const void* w9, void* dx, const void* res, const void* bna,
                                       const void* bnx, const float* bn_mean, const float* bn_rstd, const float* bn_w,
                                       const float* bn_b, float* sums, int reps, int N, int C, int H, int W, int dil,
                                       hipStream_t st) {
    if (C % 8 != 0 || N <= 0 || dil < 1 || H <= 0 || W <= 0) return -1;
    if (bna && !bnx) return -1;
    if (bnx && (!bn_mean || !bn_rstd || !sums || reps < 1)) return -1;
    if (!bnx && !res) return -1;
    if (((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)res | (uintptr_t)bna | (uintptr_t)bnx | (uintptr_t)w9) & 15) return -3;
    if (kDwGen < 4) return -2;
    if (kDwGen >= 7) {
        const int rc = launch_dw6<1>(dy, w9, dx, N, C, H, W, dil, 1, st, bnx ? sums : nullptr, reps,
                                     Dw4Bn{res, bna, bnx, bn_mean, bn_rstd, bn_w, bn_b});
        if (rc != -2) return rc;
    }
    return launch_dw4<1>(dy, nullptr, w9, dx, N, C, H, W, dil, 1, st, bnx ? sums : nullptr, reps,
                         Dw4Bn{res, bna, bnx, bn_mean, bn_rstd, bn_w, bn_b});
}
