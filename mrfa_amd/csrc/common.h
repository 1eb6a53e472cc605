// Shared helpers for the gfx950 kernels of libmrfa_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "mrfa_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void mrfa_set_error(const char* fmt, ...);

// conv_split.hip: the 128 x 128 chunked implicit-GEMM tile on the bf16 matrix pipe with exactly split fp32 operands
int mrfa_conv_split_launch(hipStream_t st, const mrfa_conv_params& p, int KT, long long M, int splitk, int BN);
// conv_halo.hip: 3x3 stride-1 convolutions tiled over 2-D output patches, the input halo split once per channel chunk (same arithmetic)
bool mrfa_conv_halo_eligible(const mrfa_conv_params& p);
int mrfa_conv_halo_launch(hipStream_t st, const mrfa_conv_params& p);
int mrfa_tuning_conv_small();      // mrfa_set_tuning("conv_small", 0 / 1)
// wgrad_split.hip: the 128 x 128 chunked, row-aligned weight-gradient tile in the same arithmetic
int mrfa_wgrad_split_launch(hipStream_t st, const mrfa_wgrad_params& p, dim3 grid, long long M, long long kps, int tiles_n, int nsplit, int inner,
                            int total_splits, int taps, long long partial_stride, int BM, int BN);

// wgrad_halo.hip: weight gradient of the 3x3 stride-1 layers, all nine taps from one staging of X / dY (transposing LDS reads)
bool mrfa_wgrad_halo_eligible(const mrfa_wgrad_params& p);
int mrfa_wgrad_halo_launch(hipStream_t st, const mrfa_wgrad_params& p);
int mrfa_tuning_wgrad_halo_min(int set);   // mrfa_set_tuning("wgrad_halo_min_wgs", n)
int mrfa_tuning_wgrad_halo_target(int set);
int mrfa_tuning_wgrad_halo_phase(int set);
int mrfa_tuning_wgrad_halo(int set);       // mrfa_set_tuning("wgrad_halo", 0 / 1); set < 0: query

// conv_fewout3.hip: 3x3 layers with 1 / 2 output channels, channels across the lanes (true: handled, *rc = status)
bool mrfa_fewout3_wgrad(hipStream_t st, const float* x, int ldx, int N, int H, int W, int Cin, const float* dy, int lddy, int Cout, int R, int pad,
                        float* dw, float* dbias, int* rc);
int mrfa_tuning_fewout3(int set);          // mrfa_set_tuning("conv_fewout3", 0 / 1)

// conv_lean.hip: the keypoint encoder's small-channel 3x3 layers (<= 128 channels, ~1 GFLOP): four-wave patches, the halo of all input channels split once
// into LDS, weight fragments straight from the pre-split planes in global memory, split-operand arithmetic
bool mrfa_conv_lean_eligible(const mrfa_conv_params& p);
int mrfa_conv_lean_launch(hipStream_t st, const mrfa_conv_params& p);
int mrfa_tuning_conv_lean(int set);        // mrfa_set_tuning("conv_lean", 0 / 1); set < 0: query
int mrfa_tuning_conv_lean_min(int set);    // mrfa_set_tuning("conv_lean_min_wgs", n)
bool mrfa_gemm_lean_eligible(const mrfa_conv_params& p, long long M);      // 1x1 convolutions / linears: the K-pipelined kernel of conv_lean.hip
int mrfa_gemm_lean_launch(hipStream_t st, const mrfa_conv_params& p, long long M);
int mrfa_tuning_gemm_lean(int set);        // mrfa_set_tuning("gemm_lean", 0 / 1); set < 0: query
int mrfa_tuning_conv_lean_geo(int set);    // mrfa_set_tuning("conv_lean_geo", i): only geometry i of conv_lean.hip's table (-1: by workgroup count)

// conv_small.hip: one wave per 16..32-row output tile, operands straight from L1/L2 into v_mfma_f32_16x16x4_f32 (small problems)
bool mrfa_conv_small_eligible(const mrfa_conv_params& p, long long M);
int mrfa_conv_small_launch(hipStream_t st, const mrfa_conv_params& p, long long M);

// attention_mfma.hip: multi-head attention forward / backward on v_mfma_f32_16x16x4_f32 (the VALU kernels of tokenpose.hip are the fallback)
bool mrfa_attention_mfma_ok(int d, int n, const void* qkv, int ld, const void* out, int ldo, const void* dout, int lddo, const void* dqkv, int lddq);
int mrfa_attention_fwd_mfma(hipStream_t st, const float* qkv, int ld, int B, int n, int heads, int d, float scale, float* out, int ldo, float* lse);
int mrfa_attention_bwd_mfma(hipStream_t st, const float* qkv, int ld, const float* out, int ldo, const float* dout, int lddo, const float* lse,
                            float* delta, int B, int n, int heads, int d, float scale, float* dqkv, int lddq);
int mrfa_tuning_attention_mfma(int set);   // mrfa_set_tuning("attention_mfma", 0 / 1); set < 0: query

// wgrad_lean.hip: all nine taps of the keypoint encoder's <= 128-channel 3x3 layers per staging, many problems per launch, prologue per statistic group
bool mrfa_wgrad_lean_eligible(const mrfa_wgrad_params& p);
int mrfa_wgrad_lean_multi(hipStream_t st, const mrfa_wgrad_params* ps, int n, unsigned char* taken);      // launches the eligible ones, taken[i] = 1 for each
int mrfa_tuning_wgrad_lean(int set);       // mrfa_set_tuning("wgrad_lean", 0 / 1); set < 0: query

// wgrad_small.hip: one wave per 32 x 32 weight block of one tap over a pixel range (small problems)
bool mrfa_wgrad_small_eligible(const mrfa_wgrad_params& p, long long M);
int mrfa_wgrad_small_launch(hipStream_t st, const mrfa_wgrad_params& p, long long M);

#define MRFA_CHECK_ARG(cond, ...)                      \
    do {                                               \
        if (!(cond)) {                                 \
            mrfa_set_error(__VA_ARGS__);               \
            return 1;                                  \
        }                                              \
    } while (0)

#define MRFA_CHECK_LAUNCH(name)                                                   \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            mrfa_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return 2;                                                             \
        }                                                                         \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// grid size for streaming (HBM-bound) kernels: enough workgroups to fill 256 CUs x 8, grid-stride the rest
static inline int stream_grid(long long work_items, int block) {
    long long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return (int)g;
}

// Kernels of the latency-bound chains (the keypoint encoder's ~3 500 small launches per step) raise their wave priority: in the training step they run
// BESIDE chip-filling matrix kernels (the deferred weight gradients, one 256-VGPR workgroup per CU for hundreds of microseconds), and a chain kernel's few
// instructions otherwise take turns with that workgroup's on the same SIMD.  s_setprio only reorders instruction issue between co-resident waves.
__device__ __forceinline__ void chain_prio() {
    __builtin_amdgcn_s_setprio(3);
}

// Statistic groups (mrfa_hip.h, v7): the group of output row `row` of a launch with M = N Hout Wout rows, and the [MRFA_STATS_SLOTS][2 Cout] statistics
// blocks of that group.  The host side guarantees that no workgroup's tile straddles two groups, so `row` is any row of the tile.
__device__ __forceinline__ int stat_group(const mrfa_conv_params& p, long long row, long long M) {
    return p.groups > 1 ? (int)(row / (M / p.groups)) : 0;
}
__device__ __forceinline__ double* stat_slot(const mrfa_conv_params& p, int group, unsigned slot) {
    return p.stats + ((size_t)group * MRFA_STATS_SLOTS + slot % MRFA_STATS_SLOTS) * 2 * p.Cout;       // see MRFA_STATS_SLOTS (mrfa_hip.h)
}
// host side: rows of one group (0: the groups do not divide the batch)
static inline long long group_rows(const mrfa_conv_params& p, long long M) {
    if (p.groups <= 1) return M;
    return (p.N % p.groups) == 0 ? M / p.groups : 0;
}

// A K split that finishes inside its launch (mrfa_conv_params.sk_ticket, v8): called by EVERY thread of a workgroup behind its split-K atomics; true in the
// workgroup that drew the tile's last ticket -- all `nsplit` partial tiles are then summed in y.  The hand-off is the one of conv_small.hip's fused BatchNorm
// finalize (no fences: an agent-scope release / acquire writes back / invalidates a whole L2): the partial sums are device-scope atomics, performed at the
// memory side; every thread waits for the completion of its own (s_waitcnt vmcnt(0): returnless atomics count there) before the barrier that precedes the
// ticket, itself a device-scope atomic; the last workgroup then reads y with device-scope loads, which bypass the non-coherent per-XCD L2s.
__device__ __forceinline__ bool splitk_last_arriver(unsigned* ticket, unsigned nsplit) {
    __shared__ unsigned s_sk_ticket;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_sk_ticket = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return s_sk_ticket == nsplit - 1;
}

// BatchNorm finalize by the LAST workgroup of a launch that accumulates the statistics (mrfa_conv_params.fin_*): called by EVERY thread of each of the
// `participants` workgroups that added into p.stats, behind those atomics; whoever draws the last ticket reads all slots and writes what bn_finalize_kernel
// would have.  No release / acquire FENCES: an agent-scope fence writes back (release) or invalidates (acquire) the XCD's whole L2 -- with one per workgroup
// the training step went from 83 to 98 ms.  None is needed: the slot sums are device-scope atomics (performed at the memory side, never left dirty in an L2),
// every thread that issued some WAITS FOR THEIR COMPLETION (the explicit s_waitcnt vmcnt(0): returnless atomics count in vmcnt, and neither the back-off barrier
// of gfx950 nor a relaxed ticket makes hipcc emit that wait by itself -- round 4 shipped without it) before the barrier that precedes the ticket --
// also a device-scope atomic --, and the last workgroup reads the slots with device-scope loads, which bypass the non-coherent L2s (the guide's "sc1 stores AND
// sc1 loads" hand-off).  tests/test_wiring_cpu.py checks the compiled ISA of every kernel that calls this for the wait.
//
// Round 6 (measured on conv_lean.hip, 32 -> 32 @64^2 over 16 frames: 20.4 us with the finalize, 13.2 us without): what the 7 us were made of, and what replaced it.
//  * 256-512 workgroups drawing tickets from ONE word = that many returning atomics serialised on one address (~12 ns each: the guide's fan-in row, 3.3 us).
//    With a dense participant index `idx` the tickets are SHARDED: workgroup idx draws from word 1 + idx % 8, the last arriver of a shard draws from word 0,
//    and the last of those finalizes (fin_counter = MRFA_FIN_WORDS zeroed words).  idx < 0: the single-word form (K splits: the participants are the tiles'
//    last arrivers, nobody knows their indices).
//  * the last workgroup summed the 2 x 32 slot words of a channel as a chain of 64 dependent loads; they are now issued 32 at a time (one memory round trip
//    per batch), and with `scratch` (>= groups x 2 Cout doubles of LDS the caller no longer needs) every (group, statistic, channel) sum has its own thread.
__device__ __forceinline__ bool fin_last_arriver(unsigned* counter, unsigned participants, int idx) {
    __shared__ unsigned s_fin_ticket;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (idx < 0) {
            s_fin_ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == participants - 1;
        } else {
            const unsigned nsh = participants < 8u ? participants : 8u, sh = (unsigned)idx % nsh;
            const unsigned mine = participants / nsh + (sh < participants % nsh ? 1u : 0u);
            unsigned last = 0;
            if (__hip_atomic_fetch_add(counter + 1 + sh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine - 1)
                last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsh - 1;
            s_fin_ticket = last;
        }
    }
    __syncthreads();
    return s_fin_ticket != 0;
}

// sum of the MRFA_STATS_SLOTS slot words of one statistic of one channel: device-scope loads, all in flight together
__device__ __forceinline__ double fin_slot_sum(const double* first, size_t stride) {
    double v[MRFA_STATS_SLOTS];
#pragma unroll
    for (int s = 0; s < MRFA_STATS_SLOTS; ++s) v[s] = __hip_atomic_load(first + (size_t)s * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double t = 0.0;
#pragma unroll
    for (int s = 0; s < MRFA_STATS_SLOTS; ++s) t += v[s];
    return t;
}

__device__ __forceinline__ void fused_bn_finalize(const mrfa_conv_params& p, unsigned participants, int idx = -1, double* scratch = nullptr, int scratch_n = 0) {
    if (!fin_last_arriver(p.fin_counter, participants, idx)) return;
    const double cnt = (double)p.fin_count;
    const int G = p.groups > 1 ? p.groups : 1;
    const int C2 = 2 * p.Cout;
    const bool par = scratch != nullptr && scratch_n >= G * C2;
    if (par) {                                        // one thread per (group, statistic, channel) sum
        for (int o = threadIdx.x; o < G * C2; o += blockDim.x) scratch[o] = fin_slot_sum(p.stats + (size_t)(o / C2) * MRFA_STATS_SLOTS * C2 + o % C2, (size_t)C2);
        __syncthreads();
    }
    for (int c = threadIdx.x; c < p.Cout; c += blockDim.x) {
        float rm = p.fin_rmean ? p.fin_rmean[c] : 0.f, rv = p.fin_rmean ? p.fin_rvar[c] : 0.f;
        for (int g = 0; g < G; ++g) {                 // (statistic groups: one momentum update per group, in group order)
            const double* sg = p.stats + (size_t)g * MRFA_STATS_SLOTS * C2;
            const double t1 = par ? scratch[g * C2 + c] : fin_slot_sum(sg + c, (size_t)C2);
            const double t2 = par ? scratch[g * C2 + p.Cout + c] : fin_slot_sum(sg + p.Cout + c, (size_t)C2);
            const double m = t1 / cnt;
            double var = t2 / cnt - m * m;
            if (var < 0.0) var = 0.0;
            const float mean = (float)m, invstd = (float)(1.0 / sqrt(var + (double)p.fin_eps));
            const double unb = p.fin_count > 1 ? var * cnt / (cnt - 1.0) : var;
            rm = (1.f - p.fin_momentum) * rm + p.fin_momentum * mean;
            rv = (1.f - p.fin_momentum) * rv + p.fin_momentum * (float)unb;
            const float sc = p.fin_gamma[c] * invstd;
            const int gc = g * p.Cout + c;
            p.fin_scale[gc] = sc;
            p.fin_shift[gc] = p.fin_beta[c] - mean * sc;
            if (p.fin_mean) p.fin_mean[gc] = mean;
            if (p.fin_invstd) p.fin_invstd[gc] = invstd;
        }
        if (p.fin_rmean) { p.fin_rmean[c] = rm; p.fin_rvar[c] = rv; }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
