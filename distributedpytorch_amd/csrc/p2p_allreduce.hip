// One-shot all-reduce for latency-bound messages over xGMI peer-to-peer stores (SURVEY.md §2.6 C4,
// §5.8 (c)): SyncBatchNorm moves 2C floats (<= 16 KB) per layer and direction, ~226 times per step.
// RCCL's ring/tree launch + protocol latency dominates such messages; here one small kernel per rank
// does the whole collective:
//
//   1. every rank writes its vector into a slot of EVERY peer's receive buffer (IPC-mapped device
//      memory of the peer, one xGMI hop, all 7 links used at once), as 8-byte granules
//      {value bits, sequence tag} — one naturally aligned 64-bit system-scope store each, so a granule
//      is either absent or complete and no fence / flag round trip is needed (the "LL" idea);
//   2. every rank polls its OWN buffer until the granule of each source carries this call's tag and
//      sums the sources in rank order — the same order on every rank, so all ranks get bit-identical
//      results (replicas must not drift).
//
// Slots are double-buffered by call parity: a rank can enter call k+2 (and overwrite slot k%2 of a
// peer) only after that peer's call k+1 granules arrived, i.e. after the peer finished reading call k.
// Every spin is bounded by the wall clock; a timeout raises an error word instead of hanging the GPU, and calls that
// find the word set return at once (one bound per lost peer, not one per call).
#include "common.h"

namespace {

constexpr int kMaxWorld = 16;
constexpr int kThreadsP2P = 256;

struct P2PArgs {
    float* data;                       // in/out, n floats (ordinary device memory)
    unsigned long long* peer[kMaxWorld];   // receive buffer of every rank (peer[rank] = my own)
    int n, maxn, rank, world;
    unsigned seq;                      // >= 1, increments per call
    unsigned* err;                     // device word: set to seq on timeout
    long long timeout_ticks;           // wall_clock64 ticks (100 MHz)
};

// receive buffer layout (per rank): [2 slots][world sources][maxn granules]
__global__ __launch_bounds__(kThreadsP2P) void p2p_allreduce_kernel(P2PArgs a) {
    // A peer was lost in an earlier call (its spin ran into the bound): the step's numbers are void already and the
    // host raises at its next check — do not wait the bound out again in each of the ~200 calls that follow.
    if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    const size_t slot_off = (size_t)(a.seq & 1u) * a.world * a.maxn;
    const int stride = gridDim.x * blockDim.x;
    const int t0 = blockIdx.x * blockDim.x + threadIdx.x;
    // ---- 1. scatter my vector to every rank (own buffer included: one code path, one summation order)
    for (int i = t0; i < a.n; i += stride) {
        const unsigned long long g = ((unsigned long long)a.seq << 32) | (unsigned long long)__float_as_uint(a.data[i]);
        for (int p = 0; p < a.world; ++p) {
            const int dst = (a.rank + 1 + p) % a.world;            // stagger the links
            __hip_atomic_store(a.peer[dst] + slot_off + (size_t)a.rank * a.maxn + i, g, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // ---- 2. gather: poll my buffer, sum the sources in rank order
    const unsigned long long* mine = a.peer[a.rank] + slot_off;
    const long long t_start = wall_clock64();
    for (int i = t0; i < a.n; i += stride) {
        float acc = 0.f;
        for (int r = 0; r < a.world; ++r) {
            const unsigned long long* src = mine + (size_t)r * a.maxn + i;
            unsigned long long g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            while ((unsigned)(g >> 32) != a.seq) {
                if (wall_clock64() - t_start > a.timeout_ticks) {   // peer never arrived: report, do not hang
                    atomicExch(a.err, a.seq);
                    return;
                }
                __builtin_amdgcn_s_sleep(2);
                g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            acc += __uint_as_float((unsigned)(g & 0xffffffffull));
        }
        a.data[i] = acc;
    }
}

}  // namespace

// ---- host side: fine-grained receive buffers and their IPC handles ------------------------------
DP_EXPORT int dp_p2p_alloc(long bytes, void** out) {
    if (bytes <= 0 || !out) return -1;
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    e = hipMemset(p, 0, (size_t)bytes);
    if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
    e = hipDeviceSynchronize();
    *out = p;
    return (int)e;
}

DP_EXPORT int dp_p2p_free(void* p) { return p ? (int)hipFree(p) : 0; }

// handle: 64 bytes (hipIpcMemHandle_t)
DP_EXPORT int dp_p2p_get_handle(void* p, void* handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    return (int)hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), p);
}

DP_EXPORT int dp_p2p_open_handle(const void* handle64, void** out) {
    const hipIpcMemHandle_t h = *reinterpret_cast<const hipIpcMemHandle_t*>(handle64);
    return (int)hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess);
}

DP_EXPORT int dp_p2p_close_handle(void* p) { return (int)hipIpcCloseMemHandle(p); }

// data: n fp32 on the current device, reduced in place.  peers: host array of `world` receive-buffer
// pointers (index = rank).  timeout_ms bounds every spin.
DP_EXPORT int dp_p2p_allreduce(void* data, int n, void* const* peers, int rank, int world, int maxn, unsigned seq,
                               unsigned* err, int timeout_ms, hipStream_t st) {
    if (!data || !peers || !err || n <= 0 || n > maxn || world < 1 || world > kMaxWorld || rank < 0 || rank >= world ||
        seq == 0)
        return -1;
    P2PArgs a{};
    a.data = (float*)data; a.n = n; a.maxn = maxn; a.rank = rank; a.world = world; a.seq = seq; a.err = err;
    for (int i = 0; i < world; ++i) {
        if (!peers[i]) return -1;
        a.peer[i] = (unsigned long long*)peers[i];
    }
    a.timeout_ticks = (long long)timeout_ms * 100000LL;       // 100 MHz constant clock
    int blocks = (n + kThreadsP2P - 1) / kThreadsP2P;
    if (blocks > 16) blocks = 16;                               // every block is resident: spins cannot starve
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(blocks), dim3(kThreadsP2P), 0, st, a);
    return DP_LAUNCH_STATUS();
}
