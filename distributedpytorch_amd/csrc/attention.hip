// Dual-attention head of DANet (the model the reference trains: `/root/reference/train_pascal.py:86`; outputs
// `[1]`/`[2]` = position / channel attention branches, `:258-260`) as hand-written gfx950 kernels.
//
// POSITION attention: out[i] = gamma * sum_j softmax_j(q_i . k_j) v_j + x_i over the N = H*W positions of one
// image (N = 4096 at 512^2 / OS8), q/k 64 wide, v 512 wide, no 1/sqrt(d) scale.  The stock formulation
// materialises the N x N map per image (64 MB fp32 + 32 MB bf16 at N = 4096).  Here:
//
//   pam_fwd   flash-style: one workgroup = 128 queries x 256 value columns, keys streamed in blocks of 64
//             through LDS; S^T = K Q^T on v_mfma_f32_16x16x32_bf16 (keys on the accumulator registers, the query
//             on the lane), online softmax lane-local, and the probabilities feed O^T = V^T P^T STRAIGHT FROM
//             THE ACCUMULATOR REGISTERS (permuted k order shared with the V fragments, which come out of the
//             row-major LDS tile through ds_read_b64_tr_b16) — P never touches LDS, the N x N map never exists.
//             Epilogue writes o (bf16), y = gamma*o + x and the log-sum-exp per query.
//   pam_bwd   one workgroup = 64 keys x 512 value columns resident in LDS, queries streamed in blocks of 32:
//             S = Q K^T recomputed, P = exp(S - lse), dP = dO V^T, dS = P (dP - delta); dV^T, dK accumulate in
//             registers across the query loop (no cross-workgroup sum), dQ leaves through fp32 atomics
//             (64 adders per element; 1 MB per image).  gamma is folded into delta / dV / dP.
//   pam_delta delta'[q] = sum_c dy[q,c] o[q,c]  and  dgamma += sum_q delta'[q]
//
// CHANNEL attention: E = X^T X (C x C over N positions), A = softmax_j(max_j E - E), out = gamma * X A^T + x.
//   cam_gram     G = U^T V  (C x C, K = N, split-K + fp32 atomics, both operands through tr reads)
//   cam_softmax  row softmax of (max - E) -> bf16, and its backward dE = -A (dA - sum_j dA A)
//   the three N x C x C products (X A^T, dOut A, X (dE + dE^T)) run on the implicit-GEMM conv kernel as 1x1
//   convolutions with per-image weights (ops/attention.py).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4v;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int kD = 64;            // query / key width
constexpr int kBQ = 128;          // queries per workgroup (4 waves x 2 tiles of 16)
constexpr int kBK = 64;           // keys per LDS block
constexpr int kDVS = 256;         // value columns per workgroup
constexpr int kKLD = 144;         // LDS bytes per K row (128 + 16 pad)
constexpr int kVLD = kDVS * 2 + 32;   // LDS bytes per V row: +32 B makes rows 8 banks apart, tr reads conflict-free
constexpr float kNegBig = -1.0e30f;

__device__ __forceinline__ bf16x8 frag16(const unsigned char* p) { return *reinterpret_cast<const bf16x8*>(p); }

// two transposing reads: elements 0-3 = rows r0..r0+3, elements 4-7 = rows r1..r1+3 of this lane's column
__device__ __forceinline__ bf16x8 tr_frag2(const unsigned char* a0, const unsigned char* a1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 pack_frag(const f32x4v& a, const f32x4v& b) {
    uint4 u;
    u.x = pack2(a[0], a[1]); u.y = pack2(a[2], a[3]); u.z = pack2(b[0], b[1]); u.w = pack2(b[2], b[3]);
    return __builtin_bit_cast(bf16x8, u);
}

struct PamFwd {
    const bf16_raw* q; const bf16_raw* k; const bf16_raw* v;   // [B, N, 64], [B, N, 64], [B, N, DV]
    const bf16_raw* x;                                          // [B, N, DV] residual input (or null)
    bf16_raw* o; bf16_raw* y;                                   // attention output, gamma*o + x
    float* lse;                                                 // [B, N]
    const float* gamma;                                         // [1]
    int B, N, DV;
};

__global__ __launch_bounds__(256, 1) void pam_fwd_kernel(PamFwd p) {
    __shared__ __attribute__((aligned(16))) unsigned char Ks[kBK * kKLD];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[kBK * kVLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, dv0 = blockIdx.y * kDVS, q0 = blockIdx.x * kBQ + wave * 32;
    const long img = (long)b * p.N;

    // ---- this wave's queries as B operand fragments: Q[q = q0 + 16 qt + l15][d = 32 ks + 8 g + j] ----
    bf16x8 qf[2][2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + qt * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 u = make_uint4(0, 0, 0, 0);
            if (q < p.N) u = *reinterpret_cast<const uint4*>(p.q + (img + q) * kD + ks * 32 + g * 8);
            qf[qt][ks] = __builtin_bit_cast(bf16x8, u);
        }
    }
    f32x4v oacc[2][kDVS / 16];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int t = 0; t < kDVS / 16; ++t) oacc[qt][t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float mrun[2] = {kNegBig, kNegBig}, lrun[2] = {0.f, 0.f};

    // ---- staging registers: block kb+1 is fetched while block kb is consumed ----
    uint4 kreg[2], vreg[8];
    auto fetch = [&](int kb) {
        const int key0 = kb * kBK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * 256, row = c >> 3, ch = c & 7;
            kreg[i] = make_uint4(0, 0, 0, 0);
            if (key0 + row < p.N) kreg[i] = *reinterpret_cast<const uint4*>(p.k + (img + key0 + row) * kD + ch * 8);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + i * 256, row = c >> 5, ch = c & 31;
            vreg[i] = make_uint4(0, 0, 0, 0);
            if (key0 + row < p.N)
                vreg[i] = *reinterpret_cast<const uint4*>(p.v + (img + key0 + row) * p.DV + dv0 + ch * 8);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * 256, row = c >> 3, ch = c & 7;
            *reinterpret_cast<uint4*>(Ks + row * kKLD + ch * 16) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + i * 256, row = c >> 5, ch = c & 31;
            *reinterpret_cast<uint4*>(Vs + row * kVLD + ch * 16) = vreg[i];
        }
    };

    const int nkb = (p.N + kBK - 1) / kBK;
    const int qsel = l15 >> 2, pp = lane & 3;                         // tr-read role of this lane in its 16-lane group
    const unsigned char* vbase = Vs + (4 * g + qsel) * kVLD + pp * 8; // + (32 ks2 [+16]) rows, + dvt * 32 bytes
    fetch(0);
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();                       // every wave finished reading the previous block
        commit();
        __syncthreads();
        if (kb + 1 < nkb) fetch(kb + 1);

        // ---- S^T[key][q] = K Q^T : st[kt][qt][r] = key 16 kt + 4 g + r, query l15 of tile qt ----
        f32x4v st[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) st[kt][qt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const bf16x8 kf = frag16(Ks + (kt * 16 + l15) * kKLD + (ks * 32 + g * 8) * 2);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    st[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ks], st[kt][qt], 0, 0, 0);
            }
        if ((kb + 1) * kBK > p.N) {            // ragged last block: keys beyond N never win the softmax
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kb * kBK + kt * 16 + 4 * g + r >= p.N) { st[kt][0][r] = kNegBig; st[kt][1][r] = kNegBig; }
        }
        // ---- online softmax: the query sits on the lane (and on lanes +16, +32, +48 with other keys) ----
        bf16x8 pf[2][2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float mx = kNegBig;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][qt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(mrun[qt], mx);
            const float alpha = __expf(mrun[qt] - mnew);
            mrun[qt] = mnew;
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(st[kt][qt][r] - mnew);
                    st[kt][qt][r] = e;
                    ps += e;
                }
            lrun[qt] = fmaf(lrun[qt], alpha, ps);
#pragma unroll
            for (int t = 0; t < kDVS / 16; ++t) {
                oacc[qt][t][0] *= alpha; oacc[qt][t][1] *= alpha; oacc[qt][t][2] *= alpha; oacc[qt][t][3] *= alpha;
            }
            // k-step ks2 of the P V product takes keys {32 ks2 + 4g + j} (j < 4) and {32 ks2 + 16 + 4g + j - 4}
            pf[qt][0] = pack_frag(st[0][qt], st[1][qt]);
            pf[qt][1] = pack_frag(st[2][qt], st[3][qt]);
        }
        // ---- O^T[dv][q] += V^T P^T : A = V through transposing reads in the same permuted key order ----
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
            for (int t = 0; t < kDVS / 16; ++t) {
                const unsigned char* a0 = vbase + (ks2 * 32) * kVLD + t * 32;
                const bf16x8 vf = tr_frag2(a0, a0 + 16 * kVLD);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    oacc[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qt][ks2], oacc[qt][t], 0, 0, 0);
            }
    }
    // ---- epilogue: oacc[qt][t][r] = O[q = l15][dv = dv0 + 16 t + 4 g + r] ----
    const float gam = p.gamma ? p.gamma[0] : 1.f;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float l = lrun[qt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.f / l;
        const int q = q0 + qt * 16 + l15;
        if (q >= p.N) continue;
        if (blockIdx.y == 0 && g == 0) p.lse[img + q] = mrun[qt] + __logf(l);
        const long rowoff = (img + q) * p.DV + dv0 + 4 * g;
#pragma unroll
        for (int t = 0; t < kDVS / 16; ++t) {
            float o4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o4[r] = oacc[qt][t][r] * inv;
            uint2 ob;
            ob.x = pack2(o4[0], o4[1]); ob.y = pack2(o4[2], o4[3]);
            *reinterpret_cast<uint2*>(p.o + rowoff + t * 16) = ob;
            if (p.y) {
                float xin[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.x) {
                    const uint2 xb = *reinterpret_cast<const uint2*>(p.x + rowoff + t * 16);
                    xin[0] = __uint_as_float(xb.x << 16); xin[1] = __uint_as_float(xb.x & 0xffff0000u);
                    xin[2] = __uint_as_float(xb.y << 16); xin[3] = __uint_as_float(xb.y & 0xffff0000u);
                }
                // gamma * (bf16-rounded o) + x: the value the backward pass differentiates
                uint2 yb;
                yb.x = pack2(fmaf(gam, __uint_as_float(ob.x << 16), xin[0]),
                             fmaf(gam, __uint_as_float(ob.x & 0xffff0000u), xin[1]));
                yb.y = pack2(fmaf(gam, __uint_as_float(ob.y << 16), xin[2]),
                             fmaf(gam, __uint_as_float(ob.y & 0xffff0000u), xin[3]));
                *reinterpret_cast<uint2*>(p.y + rowoff + t * 16) = yb;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// backward of the position attention
// ------------------------------------------------------------------------------------------------
constexpr int kBQB = 32;                 // queries per step of the backward kernel
constexpr int kDVB = 512;                // value width handled by the backward kernel (whole rows resident)
constexpr int kWLD = kDVB * 2 + 32;      // LDS bytes per V / dy row: 8 banks apart (b128 and tr reads conflict-free)
constexpr int kSLD = 96;                 // LDS bytes per dS row ([key][32 queries] + pad)

struct PamBwd {
    const bf16_raw* q; const bf16_raw* k; const bf16_raw* v; const bf16_raw* dy;
    const float* lse; const float* delta; const float* gamma;
    float* dq;                  // [KB, B, N, 64] fp32 partials: key block kb STORES its share of dQ into slab kb (no
                                //   atomics: plain stores run at 4-5x the float-atomic rate and the sum over the
                                //   slabs, pam_dq_reduce_kernel, has a fixed order)
    bf16_raw* dk; bf16_raw* dv; // [B, N, 64], [B, N, DV]
    int B, N;
};

__global__ __launch_bounds__(256, 1) void pam_bwd_kernel(PamBwd p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Vs = smem;                                // [64][kWLD]
    unsigned char* Ks = Vs + kBK * kWLD;                     // [64][kKLD]
    unsigned char* Ys = Ks + kBK * kKLD;                     // [32][kWLD]   dy of the current query block
    unsigned char* Qs = Ys + kBQB * kWLD;                    // [32][kKLD]
    unsigned char* Ss = Qs + kBQB * kKLD;                    // [64][kSLD]   dS^T (bf16)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4, qsel = l15 >> 2, pp = lane & 3;
    const int b = blockIdx.y, key0 = blockIdx.x * kBK;
    const long img = (long)b * p.N;
    const float gam = p.gamma ? p.gamma[0] : 1.f;

    // ---- resident operands: this block's keys and values ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + i * 256, row = c >> 3, ch = c & 7;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (key0 + row < p.N) u = *reinterpret_cast<const uint4*>(p.k + (img + key0 + row) * kD + ch * 8);
        *reinterpret_cast<uint4*>(Ks + row * kKLD + ch * 16) = u;
    }
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int c = tid + i * 256, row = c >> 6, ch = c & 63;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (key0 + row < p.N) u = *reinterpret_cast<const uint4*>(p.v + (img + key0 + row) * kDVB + ch * 8);
        *reinterpret_cast<uint4*>(Vs + row * kWLD + ch * 16) = u;
    }
    __syncthreads();
    // this wave's 16 keys as the B operand of S = Q K^T (constant over the query loop)
    bf16x8 kf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) kf[ks] = frag16(Ks + (wave * 16 + l15) * kKLD + (ks * 32 + g * 8) * 2);
    const bool key_ok = key0 + wave * 16 + l15 < p.N;

    f32x4v dvacc[kDVB / 16], dkacc[kD / 16];
#pragma unroll
    for (int t = 0; t < kDVB / 16; ++t) dvacc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < kD / 16; ++t) dkacc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // lse / delta' of the block's 32 queries travel with the prefetch as well (8 of each per lane): read inside the
    // step under their bounds tests they were 16 dependent global loads with a wait each — 4 of a step's 6 us
    uint4 yreg[8], qreg;
    float lnext[8], dnext[8];
    auto fetch = [&](int qb) {
        const int qrow0 = qb * kBQB;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int qg = min(qrow0 + (i >> 2) * 16 + 4 * g + (i & 3), p.N - 1);
            lnext[i] = p.lse[img + qg];
            dnext[i] = p.delta[img + qg];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + i * 256, row = c >> 6, ch = c & 63;
            yreg[i] = make_uint4(0, 0, 0, 0);
            if (qrow0 + row < p.N) yreg[i] = *reinterpret_cast<const uint4*>(p.dy + (img + qrow0 + row) * kDVB + ch * 8);
        }
        {
            const int row = tid >> 3, ch = tid & 7;
            qreg = make_uint4(0, 0, 0, 0);
            if (qrow0 + row < p.N) qreg = *reinterpret_cast<const uint4*>(p.q + (img + qrow0 + row) * kD + ch * 8);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = tid + i * 256, row = c >> 6, ch = c & 63;
            *reinterpret_cast<uint4*>(Ys + row * kWLD + ch * 16) = yreg[i];
        }
        *reinterpret_cast<uint4*>(Qs + (tid >> 3) * kKLD + (tid & 7) * 16) = qreg;
    };

    const int nqb = (p.N + kBQB - 1) / kBQB;
    float* __restrict__ dqp = p.dq + (long)blockIdx.x * p.B * p.N * kD;      // this key block's slab
    fetch(0);
    for (int qb = 0; qb < nqb; ++qb) {
        __syncthreads();                         // the previous step's reads of Ys / Qs / Ss are done
        commit();
        float lcur[8], dcur[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { lcur[i] = lnext[i]; dcur[i] = dnext[i]; }
        __syncthreads();
        if (qb + 1 < nqb) fetch(qb + 1);
        const int qrow0 = qb * kBQB;

        // ---- S[q][key] = Q K^T and T[q][key] = dy V^T : register r of tile qt = query 16 qt + 4 g + r, key l15 ----
        f32x4v sv[2], tv[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) { sv[qt] = f32x4v{0.f, 0.f, 0.f, 0.f}; tv[qt] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const bf16x8 qfr = frag16(Qs + (qt * 16 + l15) * kKLD + (ks * 32 + g * 8) * 2);
                sv[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[ks], sv[qt], 0, 0, 0);
            }
        {
            // fragments of the NEXT pair of k-steps are requested before the MFMAs of the current pair (left alone
            // hipcc reads, waits lgkmcnt(0) and multiplies one k-step at a time: an LDS round trip per MFMA with one
            // wave per SIMD)
            constexpr int KS = kDVB / 32, GP = 2;
            bf16x8 vf[2][GP], yf[2][GP][2];
            auto rd = [&](int set, int ks0) {
#pragma unroll
                for (int u = 0; u < GP; ++u) {
                    vf[set][u] = frag16(Vs + (wave * 16 + l15) * kWLD + ((ks0 + u) * 32 + g * 8) * 2);
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
                        yf[set][u][qt] = frag16(Ys + (qt * 16 + l15) * kWLD + ((ks0 + u) * 32 + g * 8) * 2);
                }
            };
            rd(0, 0);
#pragma unroll
            for (int kg = 0; kg < KS / GP; ++kg) {
                if (kg + 1 < KS / GP) rd((kg + 1) & 1, (kg + 1) * GP);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < GP; ++u)
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
                        tv[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[kg & 1][u][qt], vf[kg & 1][u], tv[qt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- P = exp(S - lse), dS = gamma P (T - delta') ----
        f32x4v pv[2], dsv[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qg = qrow0 + qt * 16 + 4 * g + r;
                const bool on = qg < p.N && key_ok;
                const float pe = on ? __expf(sv[qt][r] - lcur[qt * 4 + r]) : 0.f;
                pv[qt][r] = pe;
                dsv[qt][r] = on ? gam * pe * (tv[qt][r] - dcur[qt * 4 + r]) : 0.f;
            }
        const bf16x8 pfr = pack_frag(pv[0], pv[1]);          // A[row = key l15][k = query (g, j)]
        const bf16x8 dsfr = pack_frag(dsv[0], dsv[1]);
        // ---- dV[key][dv] += P^T dy : B = dy through transposing reads (queries in the same permuted order) ----
        const unsigned char* ybase = Ys + (4 * g + qsel) * kWLD + pp * 8;
        {
            constexpr int NT = kDVB / 16, GT = 4;               // 4 value tiles (8 transposed reads) ahead
            bf16x8 yt[2][GT];
#pragma unroll
            for (int u = 0; u < GT; ++u) yt[0][u] = tr_frag2(ybase + u * 32, ybase + 16 * kWLD + u * 32);
#pragma unroll
            for (int tg = 0; tg < NT / GT; ++tg) {
                if (tg + 1 < NT / GT) {
#pragma unroll
                    for (int u = 0; u < GT; ++u) {
                        const int t = (tg + 1) * GT + u;
                        yt[(tg + 1) & 1][u] = tr_frag2(ybase + t * 32, ybase + 16 * kWLD + t * 32);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < GT; ++u)
                    dvacc[tg * GT + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfr, yt[tg & 1][u], dvacc[tg * GT + u], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- dK[key][d] += dS^T Q ----
        const unsigned char* qbase = Qs + (4 * g + qsel) * kKLD + pp * 8;
#pragma unroll
        for (int t = 0; t < kD / 16; ++t) {
            const bf16x8 qfr = tr_frag2(qbase + t * 32, qbase + 16 * kKLD + t * 32);
            dkacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsfr, qfr, dkacc[t], 0, 0, 0);
        }
        // ---- dS crosses LDS once, [key][query], for dQ = dS K (sum over all 64 keys of the block) ----
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            uint2 pk;
            pk.x = pack2(dsv[qt][0], dsv[qt][1]); pk.y = pack2(dsv[qt][2], dsv[qt][3]);
            *reinterpret_cast<uint2*>(Ss + (wave * 16 + l15) * kSLD + (qt * 16 + 4 * g) * 2) = pk;
        }
        __syncthreads();
        // wave w owns d columns 16 w .. 16 w + 15: dq[qt][r] = dQ[query 16 qt + 4 g + r][d = 16 w + l15]
        f32x4v dq[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            const unsigned char* kb0 = Ks + (ks2 * 32 + 4 * g + qsel) * kKLD + (wave * 16 + pp * 4) * 2;
            const bf16x8 kfr = tr_frag2(kb0, kb0 + 16 * kKLD);                     // B[k = key][col = d]
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const unsigned char* sb0 = Ss + (ks2 * 32 + 4 * g + qsel) * kSLD + (qt * 16 + pp * 4) * 2;
                const bf16x8 sfr = tr_frag2(sb0, sb0 + 16 * kSLD);                 // A[row = query][k = key]
                dq[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sfr, kfr, dq[qt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qg = qrow0 + qt * 16 + 4 * g + r;
                if (qg < p.N) dqp[(img + qg) * kD + wave * 16 + l15] = dq[qt][r];
            }
    }
    // ---- epilogue: dV, dK through LDS (the V / K images are dead) as coalesced 16-byte row stores ----
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kDVB / 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<bf16_raw*>(Vs + (wave * 16 + 4 * g + r) * kWLD + (t * 16 + l15) * 2) = f2bf(gam * dvacc[t][r]);
#pragma unroll
    for (int t = 0; t < kD / 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<bf16_raw*>(Ks + (wave * 16 + 4 * g + r) * kKLD + (t * 16 + l15) * 2) = f2bf(dkacc[t][r]);
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int c = tid + i * 256, row = c >> 6, ch = c & 63;
        if (key0 + row < p.N)
            *reinterpret_cast<uint4*>(p.dv + (img + key0 + row) * kDVB + ch * 8) =
                *reinterpret_cast<const uint4*>(Vs + row * kWLD + ch * 16);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + i * 256, row = c >> 3, ch = c & 7;
        if (key0 + row < p.N)
            *reinterpret_cast<uint4*>(p.dk + (img + key0 + row) * kD + ch * 8) =
                *reinterpret_cast<const uint4*>(Ks + row * kKLD + ch * 16);
    }
}

// dq[i] (bf16) = sum over the KB slabs of part[kb][i], in slab order (bit-reproducible)
__global__ __launch_bounds__(256) void pam_dq_reduce_kernel(const float* __restrict__ part, bf16_raw* __restrict__ dq,
                                                            long n8, int KB) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        const float* src = part + i * 8;
        for (int kb = 0; kb < KB; kb += 4) {                  // four slabs' loads in flight
            float4 v[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long off = (long)min(kb + u, KB - 1) * n8 * 8;
                v[u][0] = *reinterpret_cast<const float4*>(src + off);
                v[u][1] = *reinterpret_cast<const float4*>(src + off + 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (kb + u < KB) {
                    acc[0] += v[u][0].x; acc[1] += v[u][0].y; acc[2] += v[u][0].z; acc[3] += v[u][0].w;
                    acc[4] += v[u][1].x; acc[5] += v[u][1].y; acc[6] += v[u][1].z; acc[7] += v[u][1].w;
                }
            }
        }
        *reinterpret_cast<uint4*>(dq + i * 8) = pack8(acc);
    }
}

// delta'[row] = sum_c dy[row, c] o[row, c] (one wave per row of C = 512); dgamma_part[4 * block + wave] = that wave's
// share of sum(delta') (the caller sums the gridDim.x * 4 partials: no atomics, fixed order)
__global__ __launch_bounds__(256) void pam_delta_kernel(const bf16_raw* __restrict__ dy, const bf16_raw* __restrict__ o,
                                                        float* __restrict__ delta, float* __restrict__ dgamma,
                                                        long rows, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float total = 0.f;
    for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
        float acc = 0.f;
        for (int c = lane * 8; c < C; c += 512) {
            float a[8], bb[8];
            unpack8(*reinterpret_cast<const uint4*>(dy + r * C + c), a);
            unpack8(*reinterpret_cast<const uint4*>(o + r * C + c), bb);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = fmaf(a[j], bb[j], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) delta[r] = acc;
        total += acc;
    }
    if (dgamma && lane == 0) dgamma[blockIdx.x * 4 + wave] = total;
}


// ------------------------------------------------------------------------------------------------
// channel attention
// ------------------------------------------------------------------------------------------------
constexpr int kGT = 128;                 // gram output tile (channels x channels) per workgroup
constexpr int kGP = 32;                  // positions per K step
constexpr int kGLD = kGT * 2 + 32;       // LDS bytes per staged row (8 banks apart: tr reads conflict-free)

struct GramArgs {
    const bf16_raw* u; const bf16_raw* v;     // [B, N, C]
    float* g;                                 // [B, C, C] += U^T V
    int B, N, C, splits, steps_per_split;
};

// G[i][j] += sum_p U[p][i] V[p][j] over this workgroup's position range.  Both operands lie position-major in
// memory and are needed channel-major by the MFMA: staged as they lie, read back with ds_read_b64_tr_b16.
__global__ __launch_bounds__(256, 2) void cam_gram_kernel(GramArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char Us[kGP * kGLD];
    __shared__ __attribute__((aligned(16))) unsigned char Ws[kGP * kGLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4, qsel = l15 >> 2, pp = lane & 3;
    const int tiles = p.C / kGT;
    const int tile = blockIdx.x % (tiles * tiles), split = blockIdx.x / (tiles * tiles);
    const int i0 = (tile / tiles) * kGT, j0 = (tile % tiles) * kGT;
    const int b = blockIdx.y;
    const long img = (long)b * p.N;
    const int wm = wave >> 1, wn = wave & 1;                     // 2 x 2 waves, 64 x 64 each
    f32x4v acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int step0 = split * p.steps_per_split;
    const int step1 = min(step0 + p.steps_per_split, (p.N + kGP - 1) / kGP);
    for (int stp = step0; stp < step1; ++stp) {
        const int p0 = stp * kGP;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * 256, row = c >> 4, ch = c & 15;      // 32 rows x 16 chunks of 8 channels
            uint4 a = make_uint4(0, 0, 0, 0), w = make_uint4(0, 0, 0, 0);
            if (p0 + row < p.N) {
                a = *reinterpret_cast<const uint4*>(p.u + (img + p0 + row) * p.C + i0 + ch * 8);
                w = *reinterpret_cast<const uint4*>(p.v + (img + p0 + row) * p.C + j0 + ch * 8);
            }
            *reinterpret_cast<uint4*>(Us + row * kGLD + ch * 16) = a;
            *reinterpret_cast<uint4*>(Ws + row * kGLD + ch * 16) = w;
        }
        __syncthreads();
        bf16x8 af[4], bfr[4];
        const unsigned char* ub = Us + (4 * g + qsel) * kGLD + pp * 8 + wm * 128;
        const unsigned char* wb = Ws + (4 * g + qsel) * kGLD + pp * 8 + wn * 128;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            af[t] = tr_frag2(ub + t * 32, ub + 16 * kGLD + t * 32);
            bfr[t] = tr_frag2(wb + t * 32, wb + 16 * kGLD + t * 32);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[c], acc[a][c], 0, 0, 0);
    }
    float* gout = p.g + (long)b * p.C * p.C;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wm * 64 + a * 16 + 4 * g + r, j = j0 + wn * 64 + c * 16 + l15;
                atomicAdd(gout + (long)i * p.C + j, acc[a][c][r]);
            }
}

// One wave per row of E [rows, C] (C <= 2048, C % 64 == 0): A = softmax_j(max_j E - E) = exp(min_j E - E_j) / sum.
// Writes A (bf16) and gamma * A (bf16: the 1x1-conv weights of the forward product).
__global__ __launch_bounds__(256) void cam_softmax_fwd_kernel(const float* __restrict__ e, bf16_raw* __restrict__ a,
                                                              bf16_raw* __restrict__ ag, const float* __restrict__ gamma,
                                                              long rows, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float gam = gamma ? gamma[0] : 1.f;
    const float* er = e + row * C;
    float mn = 3.0e38f;
    for (int c = lane; c < C; c += 64) mn = fminf(mn, er[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o, 64));
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += __expf(mn - er[c]);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int c = lane; c < C; c += 64) {
        const float v = __expf(mn - er[c]) * inv;
        a[row * C + c] = f2bf(v);
        ag[row * C + c] = f2bf(gam * v);
    }
}

// dE = -A (dA - sum_j dA_j A_j) with dA = gamma * graw (graw = dy^T X from cam_gram); dgamma[row] = sum_j graw_j A_j
// (per-row partials: the caller sums them).
__global__ __launch_bounds__(256) void cam_softmax_bwd_kernel(const float* __restrict__ graw, const bf16_raw* __restrict__ a,
                                                              const float* __restrict__ gamma, float* __restrict__ de,
                                                              float* __restrict__ dgamma, long rows, int C) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float gam = gamma ? gamma[0] : 1.f;
    float dot = 0.f;
    for (int c = lane; c < C; c += 64) dot = fmaf(graw[row * C + c], bf2f(a[row * C + c]), dot);
    dot = wave_sum(dot);
    for (int c = lane; c < C; c += 64) {
        const float av = bf2f(a[row * C + c]);
        de[row * C + c] = -gam * av * (graw[row * C + c] - dot);
    }
    if (dgamma && lane == 0) dgamma[row] = dot;
}

// out[b][i][j] = alpha * in[b][i][j] + beta * in[b][j][i]  (C x C per image; bf16 out; fp32 or bf16 in)
template <typename T>
__global__ __launch_bounds__(256) void cam_mat_kernel(const T* __restrict__ in, bf16_raw* __restrict__ out, int C,
                                                      float alpha, float beta) {
    __shared__ float tile[32][33];
    const long base = (long)blockIdx.z * C * C;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;             // 32 x 8
    auto ld = [&](long idx) -> float {
        if constexpr (sizeof(T) == 4) return (float)in[idx];
        else return bf2f((bf16_raw)in[idx]);
    };
    for (int r = ty; r < 32; r += 8) tile[r][tx] = ld(base + (long)(j0 + r) * C + i0 + tx);    // in[j][i]
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long idx = base + (long)(i0 + r) * C + j0 + tx;
        out[idx] = f2bf(alpha * ld(idx) + beta * tile[tx][r]);
    }
}

}  // namespace

// q, k: [B, N, 64] bf16; v, x, o, y: [B, N, DV] bf16 (DV % 256 == 0); lse fp32 [B, N]; gamma fp32 [1].
DP_EXPORT int dp_pam_fwd(const void* q, const void* k, const void* v, const void* x, void* o, void* y, float* lse,
                         const float* gamma, int B, int N, int D, int DV, hipStream_t st) {
    if (D != kD || DV <= 0 || DV % kDVS != 0 || B <= 0 || N <= 0 || !q || !k || !v || !o || !lse) return -1;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)x | (uintptr_t)y) & 15) return -3;
    if ((long)B * N * DV >= (1L << 31) * 8L) return -4;
    PamFwd p{(const bf16_raw*)q, (const bf16_raw*)k, (const bf16_raw*)v, (const bf16_raw*)x, (bf16_raw*)o,
             (bf16_raw*)y, lse, gamma, B, N, DV};
    hipLaunchKernelGGL(pam_fwd_kernel, dim3((N + kBQ - 1) / kBQ, DV / kDVS, B), dim3(256), 0, st, p);
    return DP_LAUNCH_STATUS();
}

// dq_part: fp32 scratch [ceil(N / 64), B, N, 64] (every element is written); dq bf16 [B, N, 64]; dk bf16 [B, N, 64];
// dv bf16 [B, N, 512]
DP_EXPORT int dp_pam_bwd(const void* q, const void* k, const void* v, const void* dy, const float* lse,
                         const float* delta, const float* gamma, float* dq_part, void* dq, void* dk, void* dv, int B,
                         int N, int D, int DV, hipStream_t st) {
    if (D != kD || DV != kDVB || B <= 0 || N <= 0 || !q || !k || !v || !dy || !lse || !delta || !dq_part || !dq || !dk || !dv)
        return -1;
    if (((uintptr_t)dq_part | (uintptr_t)dq) & 15) return -3;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dy | (uintptr_t)dk | (uintptr_t)dv) & 15) return -3;
    PamBwd p{(const bf16_raw*)q, (const bf16_raw*)k, (const bf16_raw*)v, (const bf16_raw*)dy, lse, delta, gamma, dq_part,
             (bf16_raw*)dk, (bf16_raw*)dv, B, N};
    constexpr int lds = kBK * kWLD + kBK * kKLD + kBQB * kWLD + kBQB * kKLD + kBK * kSLD;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)pam_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        once = true;
    }
    const int KB = (N + kBK - 1) / kBK;
    hipLaunchKernelGGL(pam_bwd_kernel, dim3(KB, B), dim3(256), lds, st, p);
    const long n8 = (long)B * N * kD / 8;
    const long rb = (n8 + 255) / 256;
    hipLaunchKernelGGL(pam_dq_reduce_kernel, dim3((unsigned)(rb < 4096 ? rb : 4096)), dim3(256), 0, st, dq_part,
                       (bf16_raw*)dq, n8, KB);
    return DP_LAUNCH_STATUS();
}

// delta fp32 [rows]; dgamma fp32 [dp_pam_delta_parts(rows)] partial sums (may be null); C % 512 == 0
DP_EXPORT long dp_pam_delta_parts(long rows) {
    long blocks = (rows + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    return blocks * 4;
}

DP_EXPORT int dp_pam_delta(const void* dy, const void* o, float* delta, float* dgamma, long rows, int C,
                           hipStream_t st) {
    if (rows <= 0 || C <= 0 || C % 8 != 0 || !dy || !o || !delta) return -1;     // (a wave covers 512 channels per pass)
    long blocks = (rows + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pam_delta_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_raw*)dy,
                       (const bf16_raw*)o, delta, dgamma, rows, C);
    return DP_LAUNCH_STATUS();
}

// g fp32 [B, C, C] accumulated (zero it first); u, v bf16 [B, N, C]; C % 128 == 0
DP_EXPORT int dp_cam_gram(const void* u, const void* v, float* g, int B, int N, int C, hipStream_t st) {
    if (B <= 0 || N <= 0 || C <= 0 || C % kGT != 0 || !u || !v || !g) return -1;
    if (((uintptr_t)u | (uintptr_t)v) & 15) return -3;
    const int steps = (N + kGP - 1) / kGP;
    const int tiles = (C / kGT) * (C / kGT);
    int splits = (1024 + tiles * B - 1) / (tiles * B);         // ~1024 workgroups in total
    if (splits < 1 || g_dp_deterministic) splits = 1;         // (deterministic: one writer per output element)
    if (splits > steps) splits = steps;
    const int sps = (steps + splits - 1) / splits;
    splits = (steps + sps - 1) / sps;
    GramArgs p{(const bf16_raw*)u, (const bf16_raw*)v, g, B, N, C, splits, sps};
    hipLaunchKernelGGL(cam_gram_kernel, dim3(tiles * splits, B), dim3(256), 0, st, p);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_cam_softmax_fwd(const float* e, void* a, void* ag, const float* gamma, long rows, int C,
                                 hipStream_t st) {
    if (rows <= 0 || C <= 0 || !e || !a || !ag) return -1;
    hipLaunchKernelGGL(cam_softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, e, (bf16_raw*)a,
                       (bf16_raw*)ag, gamma, rows, C);
    return DP_LAUNCH_STATUS();
}

DP_EXPORT int dp_cam_softmax_bwd(const float* graw, const void* a, const float* gamma, float* de, float* dgamma,
                                 long rows, int C, hipStream_t st) {
    if (rows <= 0 || C <= 0 || !graw || !a || !de) return -1;
    hipLaunchKernelGGL(cam_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, graw,
                       (const bf16_raw*)a, gamma, de, dgamma, rows, C);
    return DP_LAUNCH_STATUS();
}

// in_kind: 0 = fp32, 1 = bf16; out bf16 [B, C, C]; C % 32 == 0
DP_EXPORT int dp_cam_mat(const void* in, int in_kind, void* out, int B, int C, float alpha, float beta,
                         hipStream_t st) {
    if (B <= 0 || C <= 0 || C % 32 != 0 || !in || !out || in == out) return -1;
    const dim3 grid(C / 32, C / 32, B);
    if (in_kind == 0)
        hipLaunchKernelGGL(cam_mat_kernel<float>, grid, dim3(256), 0, st, (const float*)in, (bf16_raw*)out, C, alpha, beta);
    else
        hipLaunchKernelGGL(cam_mat_kernel<bf16_raw>, grid, dim3(256), 0, st, (const bf16_raw*)in, (bf16_raw*)out, C,
                           alpha, beta);
    return DP_LAUNCH_STATUS();
}

// ---- DANet head tail: the three dropouts in front of the three classifiers, one pass -----------------------------
// The head ends in conv8(drop(sa + sc)), conv6(drop(sa)), conv7(drop(sc)) with independent masks.  As separate ops
// that is an add and three dropouts forward (5 reads + 4 writes of a [M, 512] tensor) and three masked scales plus
// two adds backward; here one kernel reads sa, sc and writes the three dropped tensors (2 reads + 3 writes), one
// kernel reads the three gradients and writes d_sa, d_sc (3 reads + 2 writes).  Masks are counter-based
// (common.h: Drop), stream k of a call uses seed_hi + k·0x9E3779B9, and are recomputed in backward.
namespace {

struct HeadDrop {
    const bf16_raw* a; const bf16_raw* b; const bf16_raw* c;    // fwd: sa, sc, -      bwd: g(sa), g(sc), g(sum)
    bf16_raw* o0; bf16_raw* o1; bf16_raw* o2;                   // fwd: drop(sa), drop(sc), drop(sa+sc)   bwd: d_sa, d_sc, -
    long n8;                                                     // 16-byte chunks
    Drop drop;
};

template <bool BWD>
__global__ __launch_bounds__(256) void head_drop_kernel(HeadDrop p) {
    Drop d0 = p.drop, d1 = p.drop, d2 = p.drop;
    d1.seed_hi += 0x9E3779B9u; d2.seed_hi += 2u * 0x9E3779B9u;
    const float sc = p.drop.scale;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < p.n8; q += (long)gridDim.x * blockDim.x) {
        const unsigned k0 = drop_keep8(q, d0), k1 = drop_keep8(q, d1), k2 = drop_keep8(q, d2);
        float a[8], b[8], o0[8], o1[8];
        unpack8(reinterpret_cast<const uint4*>(p.a)[q], a);
        unpack8(reinterpret_cast<const uint4*>(p.b)[q], b);
        if (!BWD) {
            float o2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o0[j] = (k0 >> j) & 1u ? a[j] * sc : 0.f;
                o1[j] = (k1 >> j) & 1u ? b[j] * sc : 0.f;
                o2[j] = (k2 >> j) & 1u ? bf2f(f2bf(a[j] + b[j])) * sc : 0.f;      // sa + sc rounds to bf16 first
            }
            reinterpret_cast<uint4*>(p.o0)[q] = pack8(o0);
            reinterpret_cast<uint4*>(p.o1)[q] = pack8(o1);
            reinterpret_cast<uint4*>(p.o2)[q] = pack8(o2);
        } else {
            float c[8];
            unpack8(reinterpret_cast<const uint4*>(p.c)[q], c);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float gs = (k2 >> j) & 1u ? c[j] * sc : 0.f;
                o0[j] = ((k0 >> j) & 1u ? a[j] * sc : 0.f) + gs;
                o1[j] = ((k1 >> j) & 1u ? b[j] * sc : 0.f) + gs;
            }
            reinterpret_cast<uint4*>(p.o0)[q] = pack8(o0);
            reinterpret_cast<uint4*>(p.o1)[q] = pack8(o1);
        }
    }
}

}  // namespace

// fwd (bwd = 0): a = sa, b = sc, c unused; o0 = drop(sa), o1 = drop(sc), o2 = drop(sa + sc).
// bwd (bwd = 1): a, b, c = gradients of o0, o1, o2; o0 = d_sa, o1 = d_sc.  n: elements (multiple of 8), all 16-B aligned.
DP_EXPORT int dp_head_dropout(const void* a, const void* b, const void* c, void* o0, void* o1, void* o2, long n,
                              unsigned thr, unsigned long long seed, int bwd, hipStream_t st) {
    if (!a || !b || !o0 || !o1 || n <= 0 || n % 8 || thr == 0 || thr > 65535) return -1;
    if (bwd ? !c : !o2) return -1;
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)o0 | (uintptr_t)o1 | (uintptr_t)o2) & 15) return -3;
    HeadDrop p{};
    p.a = (const bf16_raw*)a; p.b = (const bf16_raw*)b; p.c = (const bf16_raw*)c;
    p.o0 = (bf16_raw*)o0; p.o1 = (bf16_raw*)o1; p.o2 = (bf16_raw*)o2;
    p.n8 = n / 8;
    p.drop.thr = thr; p.drop.scale = 65536.f / (float)(65536u - thr);
    p.drop.seed_lo = (unsigned)seed; p.drop.seed_hi = (unsigned)(seed >> 32);
    p.drop.word = g_dp_drop_word;                     // captured steps: new masks per replay (common.h)
    long blocks = (p.n8 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (bwd) hipLaunchKernelGGL(head_drop_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(head_drop_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, p);
    return DP_LAUNCH_STATUS();
}
