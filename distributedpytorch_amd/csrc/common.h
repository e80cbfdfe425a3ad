// Shared device helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DP_EXPORT extern "C" __attribute__((visibility("default")))

typedef unsigned short bf16_raw;  // storage type: tensors are passed as raw 16-bit words
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

static constexpr int kWave = 64;

__device__ __forceinline__ float bf2f(bf16_raw u) { return __uint_as_float(((unsigned)u) << 16); }

// plain cast: hipcc -O3 emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
__device__ __forceinline__ bf16_raw f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}

__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack2(f[0], f[1]); v.y = pack2(f[2], f[3]);
    v.z = pack2(f[4], f[5]); v.w = pack2(f[6], f[7]);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Counter-based dropout: the keep decision of an element is a hash of (element index, seed), 16 bits per element,
// so backward kernels RECOMPUTE it — no mask tensor.  thr = round(p·65536) (0 = off), scale = 1 / (1 − thr/65536).
struct Drop { unsigned thr; float scale; unsigned seed_lo, seed_hi;
              const unsigned* word; };   // optional device word mixed into the key (see g_dp_drop_word below), or null

__device__ __forceinline__ unsigned drop_hash(unsigned lo, unsigned hi, const Drop& d) {
    unsigned x = lo ^ d.seed_lo;
    x *= 0xcc9e2d51u; x = (x << 15) | (x >> 17); x *= 0x1b873593u;
    x ^= (hi ^ d.seed_hi) * 0x85ebca6bu;
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;      // murmur3 fmix32
    return x;
}

// keep bits of the 8 elements of 16-byte chunk q (bit j = element j)
__device__ __forceinline__ unsigned drop_keep8(long q, const Drop& d0) {
    const unsigned long long e = (unsigned long long)q * 4ull;
    Drop d = d0;
    if (d0.word) d.seed_hi ^= d0.word[0] * 0x9E3779B9u;       // a uniform load; forward and backward read the same word
    unsigned keep = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned h = drop_hash((unsigned)(e + k), (unsigned)((e + k) >> 32), d);
        keep |= (unsigned)((h & 0xffffu) >= d.thr) << (2 * k);
        keep |= (unsigned)((h >> 16) >= d.thr) << (2 * k + 1);
    }
    return keep;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Deterministic mode (ops._backend.set_deterministic / DPAMD_DETERMINISTIC=1; defined in misc.hip): launchers
// whose blocks would add into a shared address with float atomics shape their grids so that every address has ONE
// writer (row splits = 1, one replica per block, split-K = 1); the Python wrappers sum the replicas in a fixed
// order.  Two identical launches are then bit-identical.  Host-side flag only: no kernel reads it.
extern "C" int g_dp_deterministic;

// Captured training steps (parallel/engine.py, hipGraph replay): a kernel argument is frozen at capture, so whatever
// must change from step to step lives in device memory.  g_dp_drop_word (set by dp_set_drop_word; null = off): a device
// counter the captured step increments first thing, mixed into every fused dropout's key — a replay draws new masks,
// and its backward, reading the same word, recomputes the same ones.  Launchers copy the pointer into Drop::word.
extern "C" const unsigned* g_dp_drop_word;

// Launch-status helper: launchers return hipGetLastError() as int (0 = ok).
#define DP_LAUNCH_STATUS() ((int)hipGetLastError())
