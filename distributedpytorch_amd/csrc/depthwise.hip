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

// ------------------------------------------------------------------------------------------------
// Strip-mapped weight gradient (the general-shape fallback: stride 2, dilations whose row ring does not fit the
// LDS).  A thread owns ONE channel group (8 channels) and a strip of STRIP CONSECUTIVE output pixels: the 72
// partial sums live in registers for the whole strip, the pixel coordinates advance by increments, and the taps
// of neighbouring pixels hit in L1 (a grid-stride mapping spilled the partial sums per pixel: 445 us for 197 MB,
// profiles/r1_native_x65_os8_b16.md).  A block is 32 channel groups x 8 strips.
// `reps`: dw9 is [reps][9][C] and block b adds into replica b % reps (1 = everybody into one buffer; the
// deterministic mode gives every block its own and sums them in a fixed order afterwards).
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
__global__ __launch_bounds__(256) void dwconv2_bwd_weight_kernel(const bf16_raw* __restrict__ dy,
                                                                 const bf16_raw* __restrict__ x,
                                                                 float* __restrict__ dw9, Dw p, int reps, int tap_stride,
                                                                 int ch_stride) {
    // dw9[tap * tap_stride + channel * ch_stride]: (C, 1) = the [reps][9][C] scratch layout, (1, 9) = straight into
    // the parameter's [C][3][3] gradient (the arena slot: no transposing add behind the kernel; reps = 1)
    __shared__ float red[kStrips][kSlab][9];       // one channel of the 8 at a time
    dw9 += (long)(blockIdx.x % (unsigned)reps) * 9 * p.C;
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
            atomicAdd(dw9 + (long)t * tap_stride + (long)(cg * 8 + j) * ch_stride, v);
        }
    }
}

constexpr unsigned kOOBd = 0x80000000u;
// ------------------------------------------------------------------------------------------------
// Rolling row window (stride 1, "same" padding: every Xception middle / exit unit).  LDS row TILES (the generation
// before this one, removed) re-read their 2d halo rows per tile (TH = 4 at d = 2 on the 65-column maps: every element
// staged TWICE) and alternated "wait for the whole tile" / "compute" with nothing in flight while computing: 427 us
// for a 394 MB tensor, 2.7x its HBM time (profiles/r2_x65_os8_b64_v29_profile.md).  Here a block walks DOWN one image (or one of `hsplit` horizontal
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
    int dw_ts = 0, dw_cs = 1;                 // MODE 2: strides of the weight-gradient buffer (0: tap stride = C)
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
        // (stat_reps doubles as the replica count of the [reps][9][C] weight-gradient buffer)
        float* dw = reinterpret_cast<float*>(p.out) + (long)(bxi % p.stat_reps) * 9 * p.C;
        for (int o = tid; o < 8 * 72; o += 256) {
            const int g = o / 72, e = o - g * 72;
            if (g >= ncg) continue;
            float v = 0.f;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) v += red[(k * 8 + g) * 72 + e];
            atomicAdd(dw + (long)(e >> 3) * (p.dw_ts ? p.dw_ts : p.C) + (long)(c0 + g * 8 + (e & 7)) * p.dw_cs, v);
        }
    }
}

struct Dw4Bn { const void* res; const void* bna; const void* bnx; const float* mean; const float* rstd; const float* w; const float* b; };

template <int MODE>
inline int launch_dw4(const void* in, const void* in2, const void* w9, void* out, int N, int C, int H, int W, int dil,
                      int flip, hipStream_t st, float* stats = nullptr, int stat_reps = 1, Dw4Bn bn = Dw4Bn{},
                      int dw_ts = 0, int dw_cs = 1) {
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
    p.dw_ts = dw_ts; p.dw_cs = dw_cs;
    auto kfn = dwconv4_kernel<MODE>;
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
    if (nb * slabs > 0x7fffffffL) return -4;
    hipLaunchKernelGGL(kfn, dim3((unsigned)(nb * slabs)), dim3(256), (size_t)lds, st, p);
    return DP_LAUNCH_STATUS();
}


// rolling-window kernels on (default) / off (tests: the flat kernels on the same shapes)
static int kDwWindow = 1;

inline bool dw_same(int stride, int pad_h, int pad_w, int dil, int Hi, int Wi, int Ho, int Wo) {
    return kDwWindow && stride == 1 && pad_h == dil && pad_w == dil && Ho == Hi && Wo == Wi;
}

}  // namespace

// tests / tools: rolling-window kernels on or off at run time (returns the previous setting)
DP_EXPORT int dp_dwconv_set_window(int on) {
    const int old = kDwWindow;
    kDwWindow = on ? 1 : 0;
    return old;
}

DP_EXPORT int dp_dwconv_fwd(const void* x, const void* w9, void* y, int N, int C, int Hi, int Wi, int Ho, int Wo,
                            int stride, int pad_h, int pad_w, int dil, hipStream_t st) {
    if (C % 8 != 0 || N <= 0 || stride < 1 || dil < 1) return -1;
    if (dw_same(stride, pad_h, pad_w, dil, Hi, Wi, Ho, Wo)) {
        const int rc = launch_dw4<0>(x, nullptr, w9, y, N, C, Hi, Wi, dil, 0, st);
        if (rc != -2) return rc;
    }
    Dw p{N, C, Hi, Wi, Ho, Wo, stride, pad_h, pad_w, dil};
    hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(grid_1d((long)N * Ho * Wo * (C / 8), 256)), dim3(256), 0, st,
                       (const bf16_raw*)x, (const bf16_raw*)w9, (bf16_raw*)y, p);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_dwconv_bwd_data(const void* dy, const void* w9, void* dx, int N, int C, int Hi, int Wi, int Ho,
                                 int Wo, int stride, int pad_h, int pad_w, int dil, hipStream_t st) {
    if (C % 8 != 0 || N <= 0 || stride < 1 || dil < 1) return -1;
    if (dw_same(stride, pad_h, pad_w, dil, Hi, Wi, Ho, Wo)) {       // = forward over dy with the taps flipped
        const int rc = launch_dw4<0>(dy, nullptr, w9, dx, N, C, Hi, Wi, dil, 1, st);
        if (rc != -2) return rc;
    }
    Dw p{N, C, Hi, Wi, Ho, Wo, stride, pad_h, pad_w, dil};
    hipLaunchKernelGGL(dwconv_bwd_data_kernel, dim3(grid_1d((long)N * Hi * Wi * (C / 8), 256)), dim3(256), 0, st,
                       (const bf16_raw*)dy, (const bf16_raw*)w9, (bf16_raw*)dx, p);
    return DP_LAUNCH_STATUS();
}

// dw9: fp32 [reps][9][C], zero-initialised (or holding a gradient to accumulate into); block b adds into replica
// b % reps.  reps = 1 is the normal call; dp_dwconv_wgrad_blocks() returns the replica count that gives every
// block a buffer of its own (deterministic mode: no two blocks add into one address).
namespace {
int dw_bwd_weight(const void* dy, const void* x, void* dw9, int reps, int tap_stride, int ch_stride, int N, int C, int Hi,
                  int Wi, int Ho, int Wo, int stride, int pad_h, int pad_w, int dil, hipStream_t st) {
    if (C % 8 != 0 || N <= 0 || stride < 1 || dil < 1 || reps < 1 || !dw9) return -1;
    if (dw_same(stride, pad_h, pad_w, dil, Hi, Wi, Ho, Wo)) {
        const int rc = launch_dw4<2>(x, dy, nullptr, dw9, N, C, Hi, Wi, dil, 0, st, nullptr, reps, Dw4Bn{}, tap_stride,
                                     ch_stride);
        if (rc != -2) return rc;
    }
    Dw p{N, C, Hi, Wi, Ho, Wo, stride, pad_h, pad_w, dil};
    const int C8 = C / 8;
    const long P = (long)N * Ho * Wo;
    const long nbw = (P + kStrips * 32 - 1) / (kStrips * 32);
    if (nbw > 0x7fffffffL) return -4;
    hipLaunchKernelGGL(dwconv2_bwd_weight_kernel<32>, dim3((unsigned)nbw, (unsigned)((C8 + kSlab - 1) / kSlab)),
                       dim3(256), 0, st, (const bf16_raw*)dy, (const bf16_raw*)x, (float*)dw9, p, reps, tap_stride,
                       ch_stride);
    return DP_LAUNCH_STATUS();
}
}  // namespace

// dw9: fp32 [reps][9][C] (tap-major), accumulated into
DP_EXPORT int dp_dwconv_bwd_weight(const void* dy, const void* x, void* dw9, int reps, int N, int C, int Hi, int Wi,
                                   int Ho, int Wo, int stride, int pad_h, int pad_w, int dil, hipStream_t st) {
    return dw_bwd_weight(dy, x, dw9, reps, C, 1, N, C, Hi, Wi, Ho, Wo, stride, pad_h, pad_w, dil, st);
}

// The same, added straight into the parameter's gradient: fp32 [C][3][3] (channel-major, the arena slot) — no scratch
// buffer and no transposing add behind the kernel.
DP_EXPORT int dp_dwconv_bwd_weight_sink(const void* dy, const void* x, void* grad, int N, int C, int Hi, int Wi, int Ho,
                                        int Wo, int stride, int pad_h, int pad_w, int dil, hipStream_t st) {
    return dw_bwd_weight(dy, x, grad, 1, 1, 9, N, C, Hi, Wi, Ho, Wo, stride, pad_h, pad_w, dil, st);
}

// Upper bound of the number of row blocks (grid.x, or image bands of the rolling-window kernels) any depthwise
// launch of this shape uses: the replica count under which no two blocks share a statistics / weight-gradient
// replica.
DP_EXPORT long dp_dwconv_row_blocks(int N, int H, int W, int Ho, int Wo) {
    const long wsplit = (W + 32 * kDw4K - 1) / (32 * kDw4K);
    const long window = (long)N * 4 * wsplit;                                  // hsplit <= 4
    const long strip = ((long)N * Ho * Wo + kStrips * 32 - 1) / (kStrips * 32);
    return window > strip ? window : strip;
}

// Forward with fused BatchNorm statistics: stats fp32 [stat_reps][2][C], zero-initialised, receives the
// per-channel sums of y and y^2.  Returns -2 when the rolling-window kernel does not take the shape (the
// caller then runs dp_dwconv_fwd and a separate statistics pass).
DP_EXPORT int dp_dwconv_fwd_stats(const void* x, const void* w9, void* y, float* stats, int stat_reps, int N, int C,
                                  int Hi, int Wi, int Ho, int Wo, int stride, int pad_h, int pad_w, int dil,
                                  hipStream_t st) {
    if (C % 8 != 0 || N <= 0 || stride < 1 || dil < 1 || !stats || stat_reps < 1) return -1;
    if (!dw_same(stride, pad_h, pad_w, dil, Hi, Wi, Ho, Wo)) return -2;
    return launch_dw4<0>(x, nullptr, w9, y, N, C, Hi, Wi, dil, 0, st, stats, stat_reps);
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
    if (!kDwWindow) return -2;
    return launch_dw4<1>(dy, nullptr, w9, dx, N, C, H, W, dil, 1, st, bnx ? sums : nullptr, reps,
                         Dw4Bn{res, bna, bnx, bn_mean, bn_rstd, bn_w, bn_b});
}
