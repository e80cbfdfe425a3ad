// Sanitizer driver for the host runtime: host_runtime.cpp is compiled together with
// this file under -fsanitize=address,undefined (CPU only; GPU sanitizers are not available on this
// platform) and every exported entry point is exercised on edge shapes: 1-pixel images, maps far
// outside the source (all-border output), odd channel counts, extreme scales, empty bucket caps.
// A heap overrun or UB in the samplers aborts the process; tests/test_host_sanitize.py runs it.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

extern "C" int dp_warp_affine_f32(const float* src, int hs, int ws, int c, float* dst, int hd, int wd,
                                  const double* m, int interp, float border);
extern "C" int dp_plan_buckets(const int64_t* sizes, int n, int64_t cap, int32_t* bucket_of);

static int check_warp(int hs, int ws, int c, int hd, int wd, const double* m, int interp) {
    std::vector<float> src((size_t)hs * ws * c), dst((size_t)hd * wd * c, -7.f);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (float)(i % 251);
    if (dp_warp_affine_f32(src.data(), hs, ws, c, dst.data(), hd, wd, m, interp, 255.f) != 0) return 1;
    for (float v : dst)
        if (!std::isfinite(v)) return 2;
    return 0;
}

int main() {
    const double ident[6] = {1, 0, 0, 0, 1, 0};
    const double far_away[6] = {1, 0, 1e6, 0, 1, -1e6};          // every sample lands outside: all border
    const double shrink[6] = {37.5, 0, -3.25, 0, 41.0, 2.5};     // output pixel = 40 input pixels
    const double rot[6] = {0.94, -0.34, 5.1, 0.34, 0.94, -3.7};
    int bad = 0;
    for (int interp = 0; interp < 3; ++interp) {
        bad += check_warp(1, 1, 1, 1, 1, ident, interp);
        bad += check_warp(1, 1, 3, 7, 5, rot, interp);
        bad += check_warp(5, 7, 3, 9, 4, ident, interp);
        bad += check_warp(17, 13, 4, 33, 29, rot, interp);
        bad += check_warp(9, 9, 1, 3, 3, shrink, interp);
        bad += check_warp(4, 6, 2, 8, 8, far_away, interp);
    }
    // argument validation must reject, not touch memory
    std::vector<float> one(1, 0.f);
    if (dp_warp_affine_f32(nullptr, 1, 1, 1, one.data(), 1, 1, ident, 0, 0.f) == 0) ++bad;
    if (dp_warp_affine_f32(one.data(), 1, 1, 1, one.data(), 0, 1, ident, 0, 0.f) == 0) ++bad;
    if (dp_warp_affine_f32(one.data(), 1, 1, 1, one.data(), 1, 1, ident, 3, 0.f) == 0) ++bad;

    const int64_t sizes[6] = {64, 1 << 20, 128, 5 << 20, 64, 64};
    int32_t bucket[6];
    const int nb = dp_plan_buckets(sizes, 6, 2 << 20, bucket);
    if (nb < 1 || bucket[0] != 0) ++bad;
    for (int i = 1; i < 6; ++i)
        if (bucket[i] < bucket[i - 1] || bucket[i] >= nb) ++bad;
    if (dp_plan_buckets(sizes, 6, 1, bucket) != 6) ++bad;          // cap below every tensor: one bucket each
    if (dp_plan_buckets(sizes, 0, 1, bucket) != -1) ++bad;
    std::printf(bad ? "host sanitize: %d FAILED\n" : "host sanitize: ok\n", bad);
    return bad ? 1 : 0;
}
