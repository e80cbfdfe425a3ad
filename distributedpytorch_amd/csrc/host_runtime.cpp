// Host-side native runtime (CPU): image geometry for the sample-dict transform pipeline.
//
// The reference calls OpenCV for every geometric transform (cv2.warpAffine / cv2.flip /
// cv2.resize behind helpers.fixed_resize — /root/reference/custom_transforms.py:116-126, :212,
// :186-193).  OpenCV is not available in this environment and the GPU path does the same work in
// one HIP gather kernel (csrc/misc.hip); this file is the CPU twin used by DataLoader workers and
// by the CPU/gloo configuration: an inverse-affine sampler with nearest / bilinear / bicubic
// (Keys a = -0.75, as OpenCV) taps and constant border, OpenMP over rows.  `resize` is the same
// sampler with the half-pixel-centre mapping.  float32 HWC in, float32 HWC out.
#include <algorithm>
#include <cmath>
#include <cstdint>

extern "C" {

static inline float cubic_w(float x) {
    const float a = -0.75f;
    x = std::fabs(x);
    if (x <= 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
    if (x < 2.f) return ((a * x - 5.f * a) * x + 8.f * a) * x - 4.f * a;
    return 0.f;
}

// dst[y, x, :] = sample(src, m[0]*x + m[1]*y + m[2], m[3]*x + m[4]*y + m[5])
// interp: 0 nearest, 1 bilinear, 2 bicubic.  m is the OUTPUT->INPUT map.
__attribute__((visibility("default")))
int dp_warp_affine_f32(const float* src, int hs, int ws, int c, float* dst, int hd, int wd, const double* m,
                       int interp, float border) {
    if (!src || !dst || hs <= 0 || ws <= 0 || hd <= 0 || wd <= 0 || c <= 0 || interp < 0 || interp > 2) return -1;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < hd; ++y) {
        for (int x = 0; x < wd; ++x) {
            const double sx = m[0] * x + m[1] * y + m[2];
            const double sy = m[3] * x + m[4] * y + m[5];
            float* out = dst + ((size_t)y * wd + x) * c;
            if (interp == 0) {
                const long xi = std::lround(sx), yi = std::lround(sy);
                const bool in = xi >= 0 && xi < ws && yi >= 0 && yi < hs;
                for (int k = 0; k < c; ++k) out[k] = in ? src[((size_t)yi * ws + xi) * c + k] : border;
                continue;
            }
            const int taps = interp == 1 ? 2 : 4;
            const int off = interp == 1 ? 0 : -1;
            const double fx0 = std::floor(sx), fy0 = std::floor(sy);
            const float ax = (float)(sx - fx0), ay = (float)(sy - fy0);
            float wx[4], wy[4];
            if (interp == 1) {
                wx[0] = 1.f - ax; wx[1] = ax; wy[0] = 1.f - ay; wy[1] = ay;
            } else {
                for (int t = 0; t < 4; ++t) { wx[t] = cubic_w(ax - (t - 1)); wy[t] = cubic_w(ay - (t - 1)); }
            }
            for (int k = 0; k < c; ++k) out[k] = 0.f;
            for (int ty = 0; ty < taps; ++ty) {
                const long yy = (long)fy0 + off + ty;
                for (int tx = 0; tx < taps; ++tx) {
                    const long xx = (long)fx0 + off + tx;
                    const float w = wx[tx] * wy[ty];
                    const bool in = xx >= 0 && xx < ws && yy >= 0 && yy < hs;
                    const float* px = in ? src + ((size_t)yy * ws + xx) * c : nullptr;
                    for (int k = 0; k < c; ++k) out[k] += w * (in ? px[k] : border);
                }
            }
        }
    }
    return 0;
}

// Bucket planner: greedy contiguous partition of parameter sizes (elements, already aligned) into
// buckets of at most `cap` elements; writes the bucket index of every parameter.  Returns #buckets.
__attribute__((visibility("default")))
int dp_plan_buckets(const int64_t* sizes, int n, int64_t cap, int32_t* bucket_of) {
    if (!sizes || !bucket_of || n <= 0 || cap <= 0) return -1;
    int b = 0;
    int64_t fill = 0;
    for (int i = 0; i < n; ++i) {
        if (fill > 0 && fill + sizes[i] > cap) { ++b; fill = 0; }
        bucket_of[i] = b;
        fill += sizes[i];
    }
    return b + 1;
}

}  // extern "C"
