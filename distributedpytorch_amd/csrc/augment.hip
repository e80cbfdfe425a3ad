// GPU data path (generalised): everything the reference does per sample on CPU
// DataLoader workers with cv2 — flip (`custom_transforms.py:212`), scale+rotate warp (`:116-126`), mask crop
// (`:343-371`), fixed resize (`:156-196`), guidance map (`:30-51`), channel concat (`:307-323`), ToTensor
// (`:476-503`) — runs here as a few gathers over a DEVICE-RESIDENT uint8 copy of the dataset (288 GB of HBM
// hold all of VOC/SBD decoded: 10,582 x 0.8 MB = 8 GB), so the steady-state step has no host data path at all.
//
//   affine_augment2   one inverse-affine gather per batch: image bilinear or bicubic (Keys a = -0.75, OpenCV's
//                     INTER_CUBIC), label nearest, per-sample source slot + valid size (padded cache), optional
//                     canvas clip (out -> canvas -> source: the zero border of an intermediate warp), normalise,
//                     NHWC bf16 out (3 image channels + zero/guidance channels)
//   mask_bbox         per-sample bounding box of the nearest-warped mask (CropFromMaskStatic's get_bbox)
//   compose_crop      2x3 maps of "crop the relaxed bbox and resize it to Ho x Wo" composed with the warp
//   extreme_points    left / right / top / bottom extreme pixel of a mask, random or median candidate
//   nellipse_*        n-ellipse + Gaussian guidance channel from the four points (min / max reductions + write)
#include "common.h"

namespace {

inline int grid_1d(long total, int threads, int cap = 256 * 16) {
    long b = (total + threads - 1) / threads;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

struct Aug2 {
    int B, Hi, Wi, Ho, Wo, Cout;     // Hi/Wi: slot dimensions of the source buffers
    int interp;                      // 1 bilinear, 2 bicubic
    int lab_fill;                    // label value outside the source (255 ignore / 0 background)
    int clamp;                       // 1: clamp the interpolated image to 0..255 before normalising (cubic overshoot)
    float mean[3], inv_std[3], border[3];
};

__device__ __forceinline__ float cubic_w(float x) {      // Keys kernel, a = -0.75
    const float a = -0.75f;
    x = fabsf(x);
    if (x <= 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
    if (x < 2.f) return ((a * x - 5.f * a) * x + 8.f * a) * x - 4.f * a;
    return 0.f;
}

// mats  [B, 6]  out pixel -> source pixel (x_in = a x + b y + c ; y_in = d x + e y + f)
// clip  [B, 8]  optional: (a b c d e f  Wc Hc) out pixel -> canvas pixel and the canvas extent; output pixels whose
//               canvas position falls outside [0,Wc) x [0,Hc) take the border / fill value (the intermediate
//               warpAffine of the reference writes a canvas of the source's size with a zero border)
// src_idx [B] optional slot of sample b in the source buffers; sizes [B, 2] optional valid (h, w) of that slot
__global__ __launch_bounds__(256) void affine_augment2_kernel(const unsigned char* __restrict__ img,
                                                              const unsigned char* __restrict__ lab,
                                                              const int* __restrict__ src_idx,
                                                              const int* __restrict__ sizes,
                                                              const float* __restrict__ mats,
                                                              const float* __restrict__ clip,
                                                              bf16_raw* __restrict__ out_img,
                                                              unsigned char* __restrict__ out_lab, Aug2 p) {
    const long total = (long)p.B * p.Ho * p.Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % p.Wo);
        const int oy = (int)((i / p.Wo) % p.Ho);
        const int b = (int)(i / ((long)p.Wo * p.Ho));
        const float* m = mats + b * 6;
        const float sx = fmaf(m[0], (float)ox, fmaf(m[1], (float)oy, m[2]));
        const float sy = fmaf(m[3], (float)ox, fmaf(m[4], (float)oy, m[5]));
        const int slot = src_idx ? src_idx[b] : b;
        const int H = sizes ? sizes[2 * b] : p.Hi, W = sizes ? sizes[2 * b + 1] : p.Wi;
        bool inside = true;
        if (clip) {
            const float* q = clip + b * 8;
            const float cx = fmaf(q[0], (float)ox, fmaf(q[1], (float)oy, q[2]));
            const float cy = fmaf(q[3], (float)ox, fmaf(q[4], (float)oy, q[5]));
            inside = cx > -0.5f && cy > -0.5f && cx < q[6] - 0.5f && cy < q[7] - 0.5f;
        }
        const unsigned char* ib = img + (long)slot * p.Hi * p.Wi * 3;
        float acc[3];
        if (!inside) {
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = p.border[c];
        } else if (p.interp == 2) {
            const float fx0 = floorf(sx), fy0 = floorf(sy);
            const int x0 = (int)fx0, y0 = (int)fy0;
            const float ax = sx - fx0, ay = sy - fy0;
            float wx[4], wy[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { wx[t] = cubic_w(ax - (float)(t - 1)); wy[t] = cubic_w(ay - (float)(t - 1)); }
            acc[0] = acc[1] = acc[2] = 0.f;
#pragma unroll
            for (int ty = 0; ty < 4; ++ty) {
                const int yy = y0 + ty - 1;
#pragma unroll
                for (int tx = 0; tx < 4; ++tx) {
                    const int xx = x0 + tx - 1;
                    const float w = wx[tx] * wy[ty];
                    if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
                        const unsigned char* px = ib + ((long)yy * p.Wi + xx) * 3;
#pragma unroll
                        for (int c = 0; c < 3; ++c) acc[c] = fmaf(w, (float)px[c], acc[c]);
                    } else {
#pragma unroll
                        for (int c = 0; c < 3; ++c) acc[c] = fmaf(w, p.border[c], acc[c]);
                    }
                }
            }
        } else {
            const float fx0 = floorf(sx), fy0 = floorf(sy);
            const int x0 = (int)fx0, y0 = (int)fy0;
            const float ax = sx - fx0, ay = sy - fy0;
            acc[0] = acc[1] = acc[2] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
                const float w = ((t & 1) ? ax : 1.f - ax) * ((t >> 1) ? ay : 1.f - ay);
                if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
                    const unsigned char* px = ib + ((long)yy * p.Wi + xx) * 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[c] = fmaf(w, (float)px[c], acc[c]);
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[c] = fmaf(w, p.border[c], acc[c]);
                }
            }
        }
        bf16_raw* dst = out_img + i * p.Cout;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = acc[c];
            if (p.clamp) v = fminf(fmaxf(v, 0.f), 255.f);
            dst[c] = f2bf((v - p.mean[c]) * p.inv_std[c]);
        }
        for (int c = 3; c < p.Cout; ++c) dst[c] = 0;
        if (lab) {
            const int xn = (int)floorf(sx + 0.5f), yn = (int)floorf(sy + 0.5f);
            int v = p.lab_fill;
            if (inside && xn >= 0 && xn < W && yn >= 0 && yn < H) v = lab[((long)slot * p.Hi + yn) * p.Wi + xn];
            out_lab[i] = (unsigned char)v;
        }
    }
}

// bbox[b] = (x_min, y_min, x_max, y_max) over canvas pixels (x, y) in [0,W_b) x [0,H_b) whose nearest-warped mask
// value is > 0; initialised by the launcher to (INT_MAX, INT_MAX, -1, -1).  mats: canvas pixel -> source pixel.
__global__ __launch_bounds__(256) void mask_bbox_kernel(const unsigned char* __restrict__ mask,
                                                        const int* __restrict__ src_idx, const int* __restrict__ sizes,
                                                        const float* __restrict__ mats, int* __restrict__ bbox,
                                                        int B, int Hi, int Wi) {
    const int b = blockIdx.y;
    const int slot = src_idx ? src_idx[b] : b;
    const int H = sizes ? sizes[2 * b] : Hi, W = sizes ? sizes[2 * b + 1] : Wi;
    const float* m = mats + b * 6;
    int x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = -1, y1 = -1;
    const int total = H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int x = i % W, y = i / W;
        const float sx = fmaf(m[0], (float)x, fmaf(m[1], (float)y, m[2]));
        const float sy = fmaf(m[3], (float)x, fmaf(m[4], (float)y, m[5]));
        const int xn = (int)floorf(sx + 0.5f), yn = (int)floorf(sy + 0.5f);
        if (xn >= 0 && xn < W && yn >= 0 && yn < H && mask[((long)slot * Hi + yn) * Wi + xn] != 0) {
            x0 = min(x0, x); y0 = min(y0, y); x1 = max(x1, x); y1 = max(y1, y);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        x0 = min(x0, __shfl_xor(x0, o, 64)); y0 = min(y0, __shfl_xor(y0, o, 64));
        x1 = max(x1, __shfl_xor(x1, o, 64)); y1 = max(y1, __shfl_xor(y1, o, 64));
    }
    if ((threadIdx.x & 63) == 0 && x1 >= 0) {
        atomicMin(bbox + 4 * b + 0, x0); atomicMin(bbox + 4 * b + 1, y0);
        atomicMax(bbox + 4 * b + 2, x1); atomicMax(bbox + 4 * b + 3, y1);
    }
}

__global__ void fill_bbox_kernel(int* __restrict__ bbox, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { bbox[4 * b] = 0x7fffffff; bbox[4 * b + 1] = 0x7fffffff; bbox[4 * b + 2] = -1; bbox[4 * b + 3] = -1; }
}

// out -> source map of "crop bbox ± relax (zero-padded unless clipped) and resize to Ho x Wo (half-pixel
// centres)" composed with the canvas -> source warp `mats`; also the out -> canvas map + extent for the clip.
// Samples with an empty mask get a map that lands far outside (the crop of an empty mask is all zeros).
__global__ void compose_crop_kernel(const float* __restrict__ mats, const int* __restrict__ bbox,
                                    const int* __restrict__ sizes, float* __restrict__ out_mats,
                                    float* __restrict__ out_clip, int* __restrict__ out_bbox, int B, int Hi, int Wi,
                                    int Ho, int Wo, int relax, int zero_pad) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int H = sizes ? sizes[2 * b] : Hi, W = sizes ? sizes[2 * b + 1] : Wi;
    int x0 = bbox[4 * b], y0 = bbox[4 * b + 1], x1 = bbox[4 * b + 2], y1 = bbox[4 * b + 3];
    float* o = out_mats + 6 * b;
    float* q = out_clip + 8 * b;
    if (x1 < 0) {
        o[0] = 0.f; o[1] = 0.f; o[2] = -1.0e6f; o[3] = 0.f; o[4] = 0.f; o[5] = -1.0e6f;
        q[0] = 0.f; q[1] = 0.f; q[2] = -1.0e6f; q[3] = 0.f; q[4] = 0.f; q[5] = -1.0e6f; q[6] = (float)W; q[7] = (float)H;
        out_bbox[4 * b] = 0; out_bbox[4 * b + 1] = 0; out_bbox[4 * b + 2] = -1; out_bbox[4 * b + 3] = -1;
        return;
    }
    x0 -= relax; y0 -= relax; x1 += relax; y1 += relax;
    if (!zero_pad) { x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, W - 1); y1 = min(y1, H - 1); }
    out_bbox[4 * b] = x0; out_bbox[4 * b + 1] = y0; out_bbox[4 * b + 2] = x1; out_bbox[4 * b + 3] = y1;
    const float sx = (float)(x1 - x0 + 1) / (float)Wo, sy = (float)(y1 - y0 + 1) / (float)Ho;
    // canvas x = (u + 0.5) sx - 0.5 + x0
    const float ca = sx, cc = 0.5f * sx - 0.5f + (float)x0, ce = sy, cf = 0.5f * sy - 0.5f + (float)y0;
    q[0] = ca; q[1] = 0.f; q[2] = cc; q[3] = 0.f; q[4] = ce; q[5] = cf; q[6] = (float)W; q[7] = (float)H;
    const float* m = mats + 6 * b;
    o[0] = m[0] * ca; o[1] = m[1] * ce; o[2] = m[0] * cc + m[1] * cf + m[2];
    o[3] = m[3] * ca; o[4] = m[4] * ce; o[5] = m[3] * cc + m[4] * cf + m[5];
}

// One wave per (sample, side): side 0 left (x = xmin), 1 right (x = xmax), 2 top (y = ymin), 3 bottom (y = ymax).
// Candidates are the mask pixels on that line; the chosen one is candidate number (hash(seed, b, side) mod count)
// when `random`, else count / 2 (the median: helpers.extreme_points_fixed).  pts[b][side] = (x, y); (-1, -1) empty.
__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(64) void extreme_points_kernel(const unsigned char* __restrict__ mask,
                                                            const int* __restrict__ bbox, int* __restrict__ pts,
                                                            int B, int H, int W, int random, unsigned seed) {
    const int b = blockIdx.x >> 2, side = blockIdx.x & 3, lane = threadIdx.x;
    const int x0 = bbox[4 * b], y0 = bbox[4 * b + 1], x1 = bbox[4 * b + 2], y1 = bbox[4 * b + 3];
    int* dst = pts + (b * 4 + side) * 2;
    if (x1 < 0) { if (lane == 0) { dst[0] = -1; dst[1] = -1; } return; }
    const bool vertical = side < 2;                          // scan a column (left/right) or a row (top/bottom)
    const int fixed = side == 0 ? x0 : side == 1 ? x1 : side == 2 ? y0 : y1;
    const int len = vertical ? H : W;
    const unsigned char* mb = mask + (long)b * H * W;
    int count = 0;
    for (int base = 0; base < len; base += 64) {
        const int t = base + lane;
        const bool on = t < len && (vertical ? mb[(long)t * W + fixed] : mb[(long)fixed * W + t]) != 0;
        count += __popcll(__ballot(on));
    }
    if (count == 0) { if (lane == 0) { dst[0] = -1; dst[1] = -1; } return; }
    const int want = random ? (int)(hash32(seed ^ hash32((unsigned)(b * 4 + side) + 0x9e3779b9U)) % (unsigned)count)
                            : count / 2;
    int seen = 0;
    for (int base = 0; base < len; base += 64) {
        const int t = base + lane;
        const bool on = t < len && (vertical ? mb[(long)t * W + fixed] : mb[(long)fixed * W + t]) != 0;
        const unsigned long long bal = __ballot(on);
        const int before = seen + __popcll(bal & ((1ull << lane) - 1ull));
        if (on && before == want) { dst[0] = vertical ? fixed : t; dst[1] = vertical ? t : fixed; }
        seen += __popcll(bal);
        if (seen > want) break;
    }
}

// D(p) = sum_i |p - f_i| ; z1 = clip(1 - (D - Dmin) / (Dref - Dmin)), Dref = max_i D(f_i) ; z2 = max_i Gaussian ;
// z = 255 z1 + alpha 255 z2, rescaled so that max z = 255 (data/nellipse.py, data/transforms.py NEllipseWithGaussians).
__device__ __forceinline__ float sum_dist(const int* f, float x, float y) {
    float d = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float dx = x - (float)f[2 * k], dy = y - (float)f[2 * k + 1];
        d += sqrtf(dx * dx + dy * dy);
    }
    return d;
}

__device__ __forceinline__ float nell_z(const int* f, float x, float y, float dmin, float dref, float alpha,
                                        float inv_sigma2) {
    const float d = sum_dist(f, x, y);
    float z1;
    if (dref - dmin < 1e-6f) z1 = d <= dref ? 1.f : 0.f;
    else z1 = fminf(fmaxf(1.f - (d - dmin) / (dref - dmin), 0.f), 1.f);
    float z2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float dx = x - (float)f[2 * k], dy = y - (float)f[2 * k + 1];
        z2 = fmaxf(z2, __expf(-2.772588722f * (dx * dx + dy * dy) * inv_sigma2));     // 4 ln 2
    }
    return 255.f * z1 + alpha * 255.f * z2;
}

// pass 0: red[b][0] = min D (as uint bits; D >= 0) ; pass 1: red[b][1] = max z (uint bits)
__global__ __launch_bounds__(256) void nellipse_reduce_kernel(const int* __restrict__ pts, unsigned* __restrict__ red,
                                                              int H, int W, int pass, float alpha, float inv_sigma2) {
    const int b = blockIdx.y;
    const int* f = pts + b * 8;
    if (f[0] < 0 || f[2] < 0 || f[4] < 0 || f[6] < 0) return;
    float dmin = 0.f, dref = 0.f;
    if (pass == 1) {
        dmin = __uint_as_float(red[2 * b]);
#pragma unroll
        for (int k = 0; k < 4; ++k) dref = fmaxf(dref, sum_dist(f, (float)f[2 * k], (float)f[2 * k + 1]));
    }
    float v = pass == 0 ? 3.0e38f : 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
        const float x = (float)(i % W), y = (float)(i / W);
        if (pass == 0) v = fminf(v, sum_dist(f, x, y));
        else v = fmaxf(v, nell_z(f, x, y, dmin, dref, alpha, inv_sigma2));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float u = __shfl_xor(v, o, 64);
        v = pass == 0 ? fminf(v, u) : fmaxf(v, u);
    }
    if ((threadIdx.x & 63) == 0) {
        if (pass == 0) atomicMin(red + 2 * b, __float_as_uint(v));
        else atomicMax(red + 2 * b + 1, __float_as_uint(v));
    }
}

__global__ void nellipse_init_kernel(unsigned* __restrict__ red, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { red[2 * b] = 0x7f7fffffu; red[2 * b + 1] = 0u; }
}

// writes channel `ch` of the NHWC bf16 tensor out [B, H, W, C] (value range 0..255, then (v - mean) * inv_std)
__global__ __launch_bounds__(256) void nellipse_write_kernel(const int* __restrict__ pts, const unsigned* __restrict__ red,
                                                             bf16_raw* __restrict__ out, int B, int H, int W, int C,
                                                             int ch, float alpha, float inv_sigma2, float mean,
                                                             float inv_std) {
    const int b = blockIdx.y;
    const int* f = pts + b * 8;
    const bool empty = f[0] < 0 || f[2] < 0 || f[4] < 0 || f[6] < 0;
    float dmin = 0.f, dref = 0.f, zmax = 1.f;
    if (!empty) {
        dmin = __uint_as_float(red[2 * b]);
        zmax = fmaxf(__uint_as_float(red[2 * b + 1]), 1e-10f);
#pragma unroll
        for (int k = 0; k < 4; ++k) dref = fmaxf(dref, sum_dist(f, (float)f[2 * k], (float)f[2 * k + 1]));
    }
    const float k255 = 255.f / zmax;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
        float z = 0.f;
        if (!empty) z = fminf(nell_z(f, (float)(i % W), (float)(i / W), dmin, dref, alpha, inv_sigma2) * k255, 255.f);
        out[((long)b * H * W + i) * C + ch] = f2bf((z - mean) * inv_std);
    }
}

}  // namespace

DP_EXPORT int dp_affine_augment2(const void* img, const void* lab, const int* src_idx, const int* sizes,
                                 const float* mats, const float* clip, void* out_img, void* out_lab, int B, int Hi,
                                 int Wi, int Ho, int Wo, int Cout, int interp, int lab_fill, int clamp,
                                 const float* mean3, const float* std3, const float* border3, hipStream_t st) {
    if (B <= 0 || Cout < 3 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || (interp != 1 && interp != 2)) return -1;
    Aug2 p{B, Hi, Wi, Ho, Wo, Cout, interp, lab_fill, clamp,
           {mean3[0], mean3[1], mean3[2]}, {1.f / std3[0], 1.f / std3[1], 1.f / std3[2]},
           {border3[0], border3[1], border3[2]}};
    hipLaunchKernelGGL(affine_augment2_kernel, dim3(grid_1d((long)B * Ho * Wo, 256)), dim3(256), 0, st,
                       (const unsigned char*)img, (const unsigned char*)lab, src_idx, sizes, mats, clip,
                       (bf16_raw*)out_img, (unsigned char*)out_lab, p);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_mask_bbox(const void* mask, const int* src_idx, const int* sizes, const float* mats, int* bbox,
                           int B, int Hi, int Wi, hipStream_t st) {
    if (B <= 0 || Hi <= 0 || Wi <= 0) return -1;
    hipLaunchKernelGGL(fill_bbox_kernel, dim3((B + 63) / 64), dim3(64), 0, st, bbox, B);
    const int gx = (int)((((long)Hi * Wi + 255) / 256 + 3) / 4);
    hipLaunchKernelGGL(mask_bbox_kernel, dim3(gx < 1 ? 1 : (gx > 64 ? 64 : gx), B), dim3(256), 0, st,
                       (const unsigned char*)mask, src_idx, sizes, mats, bbox, B, Hi, Wi);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_compose_crop(const float* mats, const int* bbox, const int* sizes, float* out_mats, float* out_clip,
                              int* out_bbox, int B, int Hi, int Wi, int Ho, int Wo, int relax, int zero_pad,
                              hipStream_t st) {
    if (B <= 0 || Ho <= 0 || Wo <= 0) return -1;
    hipLaunchKernelGGL(compose_crop_kernel, dim3((B + 63) / 64), dim3(64), 0, st, mats, bbox, sizes, out_mats, out_clip,
                       out_bbox, B, Hi, Wi, Ho, Wo, relax, zero_pad);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_extreme_points(const void* mask, const int* bbox, int* pts, int B, int H, int W, int random,
                                unsigned seed, hipStream_t st) {
    if (B <= 0 || H <= 0 || W <= 0) return -1;
    hipLaunchKernelGGL(extreme_points_kernel, dim3(B * 4), dim3(64), 0, st, (const unsigned char*)mask, bbox, pts, B,
                       H, W, random, seed);
    return DP_LAUNCH_STATUS();
}

// guidance channel `ch` of out [B, H, W, C] bf16 from pts [B, 4, 2]; red: scratch [B, 2] uint32
DP_EXPORT int dp_nellipse_guidance(const int* pts, void* red, void* out, int B, int H, int W, int C, int ch,
                                   float alpha, float sigma, float mean, float stdv, hipStream_t st) {
    if (B <= 0 || H <= 0 || W <= 0 || ch < 0 || ch >= C || sigma <= 0.f) return -1;
    const float inv_sigma2 = 1.f / (sigma * sigma);
    unsigned* r = (unsigned*)red;
    hipLaunchKernelGGL(nellipse_init_kernel, dim3((B + 63) / 64), dim3(64), 0, st, r, B);
    int gx = (int)(((long)H * W + 255) / 256 / 4);
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    hipLaunchKernelGGL(nellipse_reduce_kernel, dim3(gx, B), dim3(256), 0, st, pts, r, H, W, 0, alpha, inv_sigma2);
    hipLaunchKernelGGL(nellipse_reduce_kernel, dim3(gx, B), dim3(256), 0, st, pts, r, H, W, 1, alpha, inv_sigma2);
    hipLaunchKernelGGL(nellipse_write_kernel, dim3(gx, B), dim3(256), 0, st, pts, r, (bf16_raw*)out, B, H, W, C, ch,
                       alpha, inv_sigma2, mean, 1.f / stdv);
    return DP_LAUNCH_STATUS();
}

// ---- validation of the instance task on the device (train_pascal.py:280-306 of the reference) -------------------
// Per object the reference pastes the crop-sized sigmoid back into a full-size canvas (helpers.crop2fullmask:
// bicubic resize of the crop to the bounding box, the `relax` border suppressed), thresholds it and takes the
// Jaccard index against the full-size mask with void pixels excluded — on the host, one object at a time, after a
// device->host copy of every prediction.  Here one kernel walks the full-size pixels of every object of the batch,
// samples the crop prediction where the canvas would hold it and counts intersection / union for all thresholds.
// Counts are integers (atomics: order-independent), one host read per epoch.
namespace {

struct PasteJac {
    const void* logits;            // [B, Hc, Wc] fp32 or bf16 (is_bf16)
    const unsigned char* gt;       // [slots, Hi, Wi] object mask (!= 0)
    const unsigned char* vd;       // [slots, Hi, Wi] void mask (!= 0) or null
    const int* src_idx;            // [B] slot of sample b, or null (slot = b)
    const int* sizes;              // [B, 2] (h, w) of the image inside its slot
    const int* bbox;               // [B, 4] inclusive crop box on the image (may reach outside it; x1 < x0: no object)
    int* counts;                   // [B, T, 2] (intersection, union)
    float th[4];
    int B, Hi, Wi, Hc, Wc, relax, T, is_bf16;
};

__global__ __launch_bounds__(256) void paste_jaccard_kernel(PasteJac p) {
    const int b = blockIdx.y;
    const int slot = p.src_idx ? p.src_idx[b] : b;
    const int H = p.sizes[2 * b], W = p.sizes[2 * b + 1];
    const int x0 = p.bbox[4 * b], y0 = p.bbox[4 * b + 1], x1 = p.bbox[4 * b + 2], y1 = p.bbox[4 * b + 3];
    const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
    const bool has_box = bw > 0 && bh > 0;
    const unsigned char* gt = p.gt + (size_t)slot * p.Hi * p.Wi;
    const unsigned char* vd = p.vd ? p.vd + (size_t)slot * p.Hi * p.Wi : nullptr;
    const size_t lbase = (size_t)b * p.Hc * p.Wc;
    int inter[4] = {0, 0, 0, 0}, uni[4] = {0, 0, 0, 0};
    const long npx = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        if (vd && vd[(size_t)y * p.Wi + x]) continue;
        const bool g = gt[(size_t)y * p.Wi + x] != 0;
        float val = 0.f;
        const int cx = x - x0, cy = y - y0;
        if (has_box && cx >= p.relax && cx < bw - p.relax && cy >= p.relax && cy < bh - p.relax) {
            // imgproc.resize(crop, (bh, bw), INTER_CUBIC): half-pixel centres, taps clamped to the edge
            const double sx = ((double)cx + 0.5) * ((double)p.Wc / bw) - 0.5, sy = ((double)cy + 0.5) * ((double)p.Hc / bh) - 0.5;
            const double fx = floor(sx), fy = floor(sy);
            const float ax = (float)(sx - fx), ay = (float)(sy - fy);
            const int ix = (int)fx, iy = (int)fy;
#pragma unroll
            for (int oy = -1; oy <= 2; ++oy) {
                const float wy = cubic_w(ay - (float)oy);
                const int yy = min(max(iy + oy, 0), p.Hc - 1);
#pragma unroll
                for (int ox = -1; ox <= 2; ++ox) {
                    const int xx = min(max(ix + ox, 0), p.Wc - 1);
                    const size_t off = lbase + (size_t)yy * p.Wc + xx;
                    const float z = p.is_bf16 ? bf2f(reinterpret_cast<const bf16_raw*>(p.logits)[off])
                                              : reinterpret_cast<const float*>(p.logits)[off];
                    val = fmaf(cubic_w(ax - (float)ox) * wy, 1.f / (1.f + __expf(-z)), val);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < p.T) {
                const bool pr = val > p.th[t];
                inter[t] += (pr && g) ? 1 : 0;
                uni[t] += (pr || g) ? 1 : 0;
            }
        }
    }
    __shared__ int part[4][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int a = inter[t], u = uni[t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); u += __shfl_xor(u, o, 64); }
        if (lane == 0) { part[wave][2 * t] = a; part[wave][2 * t + 1] = u; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * p.T) {
        const int v = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        if (v) atomicAdd(p.counts + (size_t)b * p.T * 2 + threadIdx.x, v);
    }
}

}  // namespace

// counts: int32 [B, T, 2], zero-initialised (or holding counts to add to).  T <= 4 thresholds.
DP_EXPORT int dp_paste_jaccard(const void* logits, int is_bf16, const void* gt, const void* vd, const int* src_idx,
                               const int* sizes, const int* bbox, int* counts, const float* th, int T, int B, int Hi,
                               int Wi, int Hc, int Wc, int relax, hipStream_t st) {
    if (!logits || !gt || !sizes || !bbox || !counts || !th || T < 1 || T > 4 || B <= 0 || Hi <= 0 || Wi <= 0 || Hc <= 0 ||
        Wc <= 0 || relax < 0)
        return -1;
    PasteJac p{};
    p.logits = logits; p.gt = (const unsigned char*)gt; p.vd = (const unsigned char*)vd; p.src_idx = src_idx;
    p.sizes = sizes; p.bbox = bbox; p.counts = counts;
    for (int t = 0; t < T; ++t) p.th[t] = th[t];
    p.B = B; p.Hi = Hi; p.Wi = Wi; p.Hc = Hc; p.Wc = Wc; p.relax = relax; p.T = T; p.is_bf16 = is_bf16;
    int gx = (int)(((long)Hi * Wi + 255) / 256);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(paste_jaccard_kernel, dim3(gx, B), dim3(256), 0, st, p);
    return DP_LAUNCH_STATUS();
}
