// Implicit-GEMM (dilated) convolution on MFMA for gfx950 — forward (K1) and data-gradient (K2).
//
//   C[M, N] = A[M, K] · B[K, N]      M = batch·Ho·Wo output pixels, N = Cout, K = R·S·Cin
//
// Layouts: activations NHWC bf16, weights KRSC bf16 (= [N][K] row-major, K contiguous), output
// NHWC bf16 with an explicit row stride (so a conv can write straight into a channel slice of a
// concat buffer).  A is never materialised: each K-step covers 64 input channels of ONE filter tap
// (r, s), for which a pixel's operand row is 128 contiguous bytes of the input image — or zeros
// when the (dilated) tap falls into the padding.
//
// CDNA4 design points
//   * 256 threads = 4 wave64; block tile BM×BN ∈ {128×128, 128×64, 64×128(not yet)}, BK = 64;
//     each wave owns a (BM/WGM)×(BN/WGN) sub-tile built from v_mfma_f32_32x32x16_bf16.
//   * both operand tiles go global→LDS with global_load_lds_dwordx4 (16 B per lane, no VGPR
//     round trip).  The LDS image must be lane-linear for that instruction, so the bank-conflict
//     swizzle is applied on the per-lane SOURCE address (which 16-B chunk of its row a lane
//     fetches) and mirrored on the ds_read side: physical chunk = logical chunk ^ ((row>>1)&7).
//     With 128-B rows this makes every ds_read_b128 lane group hit 16 distinct 16-B slots.
//   * padding / M-tail / Cout-tail / Cin-tail lanes read from a 256-B device zero page instead of
//     branching: the instruction stream is identical for interior and border tiles.
//   * two LDS stages; the loads of K-step t+1 are issued right after the barrier that retires
//     step t and fly under the 16–32 MFMAs of step t (one barrier per K-step).
//   * epilogue: fp32 accumulators (+bias) → bf16 → LDS → coalesced 16-B row stores; optional fused
//     BatchNorm statistics (per-channel Σy, Σy² from the fp32 accumulators: each lane owns one
//     output channel column, so the reduction is in-register + one cross-half shuffle).
//   * blockIdx is remapped so that the blocks an XCD receives (round-robin dispatch) work on
//     neighbouring tiles and share their A panel in that XCD's L2.
//
// Data-gradient: the same kernel with dy as input and the tap-flipped, Cin↔Cout-transposed weight
// copy as B (built once per step by dp_weight_transpose); strided convolutions use `sdiv`: a tap
// contributes only where (out − pad + r·dil) is divisible by the forward stride.
#include "common.h"

namespace {

constexpr int BK = 64;            // K elements per step (= 128 B per operand row)
constexpr int kThreads = 256;

__device__ uint4 g_zero_page[16];  // 256 B of zeros (device globals are zero-initialised)

struct ConvArgs {
    const bf16_raw* x;      // [N, Hi, Wi, Cin]
    const bf16_raw* w;      // [Cout, R, S, Cin]
    bf16_raw* y;            // [M, ldy]
    const float* bias;      // [Cout] or null
    float* stats;           // [stat_reps, 2, Cout] (Σy, Σy²) or null
    int stat_reps;
    long M;
    int Hi, Wi, Cin, Ho, Wo, Cout, R, S;
    int smul, sdiv, pad_h, pad_w, dil_h, dil_w;
    int ldy;
    int csteps, ksteps;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ void glds16(const bf16_raw* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)src,
        (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BM, int BN, int WGM, int WGN, int SDIV>
__global__ __launch_bounds__(kThreads, 2) void conv_igemm_kernel(ConvArgs p) {
    static_assert(WGM * WGN == 4, "4 waves");
    constexpr int WTM = BM / WGM, WTN = BN / WGN;        // wave tile
    constexpr int MI = WTM / 32, NI = WTN / 32;          // 32×32 MFMA tiles per wave
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int AR = BM / 32, BR = BN / 32;            // load rounds (32 rows × 128 B per round)
    constexpr int C_LD = BN * 2 + 16;                    // epilogue LDS row stride (bytes)
    static_assert(BM * C_LD + WGM * BN * 8 <= 2 * STAGE, "epilogue tile + stats scratch must fit");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- XCD-aware tile mapping (bijective for any grid size) ----
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % p.tiles_n;
    const int tile_m = bid / p.tiles_n;
    const long m0 = (long)tile_m * BM;
    const int n0 = tile_n * BN;

    // ---- per-thread load geometry ----
    const int pc = tid & 7;                 // physical 16-B chunk in the LDS row
    const int rr = tid >> 3;                // row within a 32-row round
    const int lc = pc ^ ((rr >> 1) & 7);    // logical chunk this lane fetches (source-side swizzle)
    int a_by[AR], a_bx[AR], a_off[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = (int)m0 + i * 32 + rr;     // host guarantees M < 2^31 and |x| < 2^31 elements
        if (m < (int)p.M) {
            const int ox = m % p.Wo;
            const int t = m / p.Wo;
            const int oy = t % p.Ho;
            const int n = t / p.Ho;
            a_by[i] = oy * p.smul - p.pad_h;
            a_bx[i] = ox * p.smul - p.pad_w;
            if (SDIV == 1) a_off[i] = ((n * p.Hi + a_by[i]) * p.Wi + a_bx[i]) * p.Cin + lc * 8;
            else a_off[i] = n * p.Hi * p.Wi;
        } else {
            a_by[i] = -(1 << 28);  // always out of range -> zero page
            a_bx[i] = 0;
            a_off[i] = 0;
        }
    }
    // per-lane (non-uniform) zero-page address: keeps hipcc from splitting each load into a
    // scalar-base 'zero' instruction plus a vector-address instruction under divergent branches
    const bf16_raw* zero = reinterpret_cast<const bf16_raw*>(g_zero_page) + pc * 8;
    const int w_row_stride = p.R * p.S * p.Cin;
    int b_off[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int n = n0 + i * 32 + rr;
        b_off[i] = n < p.Cout ? n * w_row_stride + lc * 8 : -1;
    }

    // K-step iterator state (uniform): filter tap (kr, ks) and 64-channel slice kc
    int kr = 0, ks = 0, kc = 0;
    auto issue_stage = [&](int buf) {
        const unsigned cvalid = (unsigned)(kc * BK + lc * 8 < p.Cin);
        unsigned char* abase = smem + buf * STAGE + wave * (8 * 128);
        unsigned char* bbase = abase + A_BYTES;
        const int dy = kr * p.dil_h, dx = ks * p.dil_w;
        const int tap_off = (dy * p.Wi + dx) * p.Cin + kc * BK;           // SDIV == 1 only
        const int wtap_off = (kr * p.S + ks) * p.Cin + kc * BK;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            int iy = a_by[i] + dy, ix = a_bx[i] + dx;
            // branch-free validity (bitwise, not short-circuit): one v_cndmask pair per load
            unsigned ok = cvalid;
            int off;
            if (SDIV == 1) {
                off = a_off[i] + tap_off;
            } else {
                ok &= (unsigned)((iy % SDIV) == 0) & (unsigned)((ix % SDIV) == 0);
                iy /= SDIV; ix /= SDIV;
                off = (a_off[i] + iy * p.Wi + ix) * p.Cin + kc * BK + lc * 8;
            }
            ok &= (unsigned)((unsigned)iy < (unsigned)p.Hi) & (unsigned)((unsigned)ix < (unsigned)p.Wi);
            glds16(ok ? p.x + off : zero, abase + i * (32 * 128));
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const unsigned ok = cvalid & (unsigned)(b_off[i] >= 0);
            glds16(ok ? p.w + (b_off[i] + wtap_off) : zero, bbase + i * (32 * 128));
        }
        if (++kc == p.csteps) { kc = 0; if (++ks == p.S) { ks = 0; ++kr; } }
    };

    // ---- accumulators ----
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    const int wm = wave / WGN, wn = wave % WGN;
    const int frow = lane & 31;                 // fragment row (pixel for A, cout for B)
    const int fhalf = lane >> 5;                // which 8-wide k half of the 16-wide MFMA K
    const int fsw = (frow >> 1) & 7;            // read-side swizzle term (tile bases are multiples of 32)
    const int a_row_off = (wm * WTM + frow) * 128;
    const int b_row_off = (wn * WTN + frow) * 128;

    issue_stage(0);
    for (int t = 0; t < p.ksteps; ++t) {
        const int buf = t & 1;
        __syncthreads();   // hipcc emits s_waitcnt vmcnt(0) for the in-flight LDS-DMA, then s_barrier
        if (t + 1 < p.ksteps) issue_stage(buf ^ 1);
        const unsigned char* As = smem + buf * STAGE;
        const unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const int chunk = ((kk * 2 + fhalf) ^ fsw) * 16;
            bf16x8 af[MI], bfr[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                af[mi] = *reinterpret_cast<const bf16x8*>(As + a_row_off + mi * (32 * 128) + chunk);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                bfr[ni] = *reinterpret_cast<const bf16x8*>(Bs + b_row_off + ni * (32 * 128) + chunk);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
        }
    }
    __syncthreads();  // all waves done reading operand tiles; LDS is reused for the epilogue

    // ---- epilogue ----
    // accumulator element e of tile (mi, ni): row = wm*WTM + mi*32 + (e&3) + 8*(e>>2) + 4*fhalf,
    //                                         col = wn*WTN + ni*32 + frow
    const bool col_vec_ok = (p.Cout % 8 == 0) && (p.ldy % 8 == 0);
    const bool do_stats = p.stats != nullptr;            // uniform
    float* stat_scratch = reinterpret_cast<float*>(smem + BM * C_LD);   // [WGM][BN][2] floats
    float st1[NI], st2[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int col = wn * WTN + ni * 32 + frow;
        const int gcol = n0 + col;
        const float bv = (p.bias && gcol < p.Cout) ? p.bias[gcol] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm * WTM + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fhalf;
                const float v = acc[mi][ni][e] + bv;
                if (do_stats) {
                    const float vm = (m0 + row < p.M) ? v : 0.f;
                    s1 += vm; s2 = fmaf(vm, vm, s2);
                }
                *reinterpret_cast<bf16_raw*>(smem + row * C_LD + col * 2) = f2bf(v);
            }
        }
        if (do_stats) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            // combine the WGM waves that share this column through LDS: one atomic pair per
            // (block, channel), not per wave
            if (fhalf == 0 && wm > 0) {
                stat_scratch[((wm - 1) * BN + col) * 2 + 0] = s1;
                stat_scratch[((wm - 1) * BN + col) * 2 + 1] = s2;
            }
        }
        st1[ni] = s1; st2[ni] = s2;
    }
    __syncthreads();
    if (do_stats && wm == 0 && fhalf == 0) {
        // replica = tile_m mod R: spreads the per-channel atomics of the thousands of M-tiles of a
        // large feature map over R lines (contended single-line fp32 atomics are ~14× slower)
        float* dst = p.stats + (long)(tile_m % p.stat_reps) * 2 * p.Cout;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int col = wn * WTN + ni * 32 + frow;
            const int gcol = n0 + col;
            float s1 = st1[ni], s2 = st2[ni];
#pragma unroll
            for (int w2 = 1; w2 < WGM; ++w2) {
                s1 += stat_scratch[((w2 - 1) * BN + col) * 2 + 0];
                s2 += stat_scratch[((w2 - 1) * BN + col) * 2 + 1];
            }
            if (gcol < p.Cout) {
                atomicAdd(dst + gcol, s1);
                atomicAdd(dst + p.Cout + gcol, s2);
            }
        }
    }
    if (col_vec_ok) {
        constexpr int TPR = BN / 8;               // threads per row (16-B pieces)
        constexpr int RPP = kThreads / TPR;       // rows per pass
        const int cpiece = tid % TPR, rsub = tid / TPR;
        const int gcol = n0 + cpiece * 8;
#pragma unroll
        for (int r0 = 0; r0 < BM; r0 += RPP) {
            const int row = r0 + rsub;
            const long m = m0 + row;
            if (m < p.M && gcol < p.Cout) {
                const uint4 v = *reinterpret_cast<const uint4*>(smem + row * C_LD + cpiece * 16);
                *reinterpret_cast<uint4*>(p.y + m * p.ldy + gcol) = v;
            }
        }
    } else {
        for (int i = tid; i < BM * BN; i += kThreads) {
            const int row = i / BN, col = i % BN;
            const long m = m0 + row;
            if (m < p.M && n0 + col < p.Cout)
                p.y[m * p.ldy + n0 + col] = *reinterpret_cast<const bf16_raw*>(smem + row * C_LD + col * 2);
        }
    }
}

// Weight re-layout for the data gradient: wT[c][R-1-r][S-1-s][k] = w[k][r][s][c]
__global__ __launch_bounds__(256) void weight_transpose_kernel(const bf16_raw* __restrict__ w, bf16_raw* __restrict__ wt,
                                                               int K, int R, int S, int C) {
    const long total = (long)K * R * S * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // i indexes the DESTINATION [c][r'][s'][k] so stores are coalesced
        const int k = (int)(i % K);
        long t = i / K;
        const int s2 = (int)(t % S); t /= S;
        const int r2 = (int)(t % R);
        const int c = (int)(t / R);
        const int r = R - 1 - r2, s = S - 1 - s2;
        wt[i] = w[(((long)k * R + r) * S + s) * C + c];
    }
}

// All layers at once (one launch per optimiser step instead of one per convolution in backward):
// `table` holds, per weight tensor, {element offset in the arena, K, R, S, C, first tile}; a block
// transposes one 32(k) x 32(c) tile of one filter tap through LDS, so both the reads (c contiguous)
// and the writes (k contiguous) are coalesced.  src and dst arenas share offsets.  R < 0: |R| filter rows and NO tap
// flip — the depthwise weights ([C][3][3][1] in memory) ride along to their tap-major [9][C] form (csrc/depthwise.hip).
struct WtSlot { long off; int K, R, S, C; int tile0; int tiles_k; };

__global__ __launch_bounds__(256) void weight_transpose_batched_kernel(const bf16_raw* __restrict__ src,
                                                                       bf16_raw* __restrict__ dst,
                                                                       const WtSlot* __restrict__ table, int nslots) {
    __shared__ bf16_raw tile[32][33];
    const int b = blockIdx.x;
    int lo = 0, hi = nslots - 1;                 // last slot with tile0 <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].tile0 <= b) lo = mid; else hi = mid - 1;
    }
    const WtSlot sl = table[lo];
    int t = b - sl.tile0;
    const int tk = t % sl.tiles_k; t /= sl.tiles_k;
    const int tiles_c = (sl.C + 31) >> 5;
    const int tc = t % tiles_c;
    const int tap = t / tiles_c;                  // r*S + s
    const int R = sl.R < 0 ? -sl.R : sl.R;
    const int r = tap / sl.S, s2 = tap - r * sl.S;
    const int tapT = sl.R < 0 ? tap : (R - 1 - r) * sl.S + (sl.S - 1 - s2);
    const int taps = R * sl.S;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;      // 32 x 8
    const bf16_raw* w = src + sl.off;
    bf16_raw* wt = dst + sl.off;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = tk * 32 + ly + i * 8, c = tc * 32 + lx;
        tile[ly + i * 8][lx] = (k < sl.K && c < sl.C) ? w[((long)k * taps + tap) * sl.C + c] : (bf16_raw)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tc * 32 + ly + i * 8, k = tk * 32 + lx;
        if (k < sl.K && c < sl.C) wt[((long)c * taps + tapT) * sl.K + k] = tile[lx][ly + i * 8];
    }
}

template <int BM, int BN, int WGM, int WGN, int SDIV>
int launch(ConvArgs& p, hipStream_t st) {
    p.tiles_m = (int)((p.M + BM - 1) / BM);
    p.tiles_n = (p.Cout + BN - 1) / BN;
    const long nwg = (long)p.tiles_m * p.tiles_n;
    if (nwg <= 0 || nwg > 0x7fffffffL) return -4;
    constexpr int lds = 2 * (BM + BN) * BK * 2;
    auto kfn = conv_igemm_kernel<BM, BN, WGM, WGN, SDIV>;
    hipLaunchKernelGGL(kfn, dim3((unsigned)nwg), dim3(kThreads), lds, st, p);
    return DP_LAUNCH_STATUS();
}

}  // namespace

// tile: 0 = auto, 1 = 128x128, 2 = 128x64
DP_EXPORT int dp_conv_igemm(const void* x, const void* w, void* y, const float* bias, float* stats, int stat_reps,
                            int N, int Hi,
                            int Wi, int Cin, int Ho, int Wo, int Cout, int R, int S, int smul, int sdiv, int pad_h,
                            int pad_w, int dil_h, int dil_w, int ldy, int tile, hipStream_t st) {
    if (Cin % 8 != 0 || Cin <= 0 || Cout <= 0 || N <= 0 || Ho <= 0 || Wo <= 0 || Hi <= 0 || Wi <= 0) return -1;
    if (ldy < Cout || R <= 0 || S <= 0 || smul < 1 || sdiv < 1) return -1;
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) return -3;
    if ((long)N * Hi * Wi * Cin >= (1L << 31) || (long)N * Ho * Wo >= (1L << 31) ||
        (long)Cout * R * S * Cin >= (1L << 31)) return -4;   // 32-bit element offsets in the kernel
    ConvArgs p{};
    p.x = (const bf16_raw*)x; p.w = (const bf16_raw*)w; p.y = (bf16_raw*)y; p.bias = bias; p.stats = stats;
    p.stat_reps = stat_reps < 1 ? 1 : stat_reps;
    p.M = (long)N * Ho * Wo;
    p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout; p.R = R; p.S = S;
    p.smul = smul; p.sdiv = sdiv; p.pad_h = pad_h; p.pad_w = pad_w; p.dil_h = dil_h; p.dil_w = dil_w;
    p.ldy = ldy;
    p.csteps = (Cin + BK - 1) / BK;
    p.ksteps = R * S * p.csteps;
    if (tile == 0) tile = (Cout <= 64) ? 2 : 1;
    if (sdiv != 1 && sdiv != 2) return -2;
    if (tile == 1) return sdiv == 1 ? launch<128, 128, 2, 2, 1>(p, st) : launch<128, 128, 2, 2, 2>(p, st);
    if (tile == 2) return sdiv == 1 ? launch<128, 64, 2, 2, 1>(p, st) : launch<128, 64, 2, 2, 2>(p, st);
    return -2;
}

DP_EXPORT int dp_weight_transpose(const void* w, void* wt, int K, int R, int S, int C, hipStream_t st) {
    if (K <= 0 || R <= 0 || S <= 0 || C <= 0) return -1;
    const long total = (long)K * R * S * C;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(weight_transpose_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_raw*)w,
                       (bf16_raw*)wt, K, R, S, C);
    return DP_LAUNCH_STATUS();
}

// table: device array of nslots records {int64 off; int32 K, R, S, C, tile0, tiles_k} (32 bytes each)
DP_EXPORT int dp_weight_transpose_batched(const void* src, void* dst, const void* table, int nslots, int total_tiles,
                                          hipStream_t st) {
    if (nslots <= 0 || total_tiles <= 0) return 0;
    static_assert(sizeof(WtSlot) == 32, "table record layout");
    hipLaunchKernelGGL(weight_transpose_batched_kernel, dim3((unsigned)total_tiles), dim3(256), 0, st,
                       (const bf16_raw*)src, (bf16_raw*)dst, (const WtSlot*)table, nslots);
    return DP_LAUNCH_STATUS();
}
