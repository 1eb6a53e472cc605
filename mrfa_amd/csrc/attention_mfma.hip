// Multi-head attention of the MTIA prior's token transformer (tokenpose_base.py:72-94: 276 tokens x 8 heads x 24, 12 layers, two encoder
// passes per step) on the fp32 matrix pipe: forward and both backward kernels with v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32
// accumulate -- the arithmetic of the VALU kernels in tokenpose.hip, which these replace).
//
// Why: the VALU kernels run one FMA per 4 LDS bytes (forward 33 us, backward 71 + 78 us per layer and pass: 4.4 ms per step); the same
// contractions are 252 / 360 / 504 MFMAs per 16-row tile.
//
// One wave owns a 16-row tile (queries in the forward and the query-side backward, keys in the key-side backward) and walks over the 16-row
// tiles of the other side.  The other side's operands of the head live in LDS as [rows up to a multiple of 16][d + 4] fp32 (rows past n zero):
// with (d + 4) / 4 odd and d + 4 = 4 or 12 mod 16 -- true for d = 16, 24, 32 -- BOTH fragment patterns are bank-conflict free:
//   pattern R ("row"):        lane (li, kq) reads  M[16 t + li][4 j + kq]          -- the A operand of  M . X^T   (contraction over d)
//   pattern T ("transposed"): lane (li, kq) reads  M[16 t + 4 kq + jj][16 dt + li] -- the A operand of  M^T . Y   (contraction over the rows)
// (columns >= d of pattern T read the next row: finite garbage that only reaches accumulator rows nobody stores).
// Every score tile is computed TRANSPOSED where needed so that its MFMA result layout (lane (li, kq) holds column li, rows 4 kq + r) is
// directly the B operand of the next product with the contraction index permuted (row 4 kq + jj at step jj; the A fragment uses the same
// permutation): the probabilities never leave the registers, the softmax never touches LDS or HBM.
//   forward           S^T = K Q^T (pattern R on K)  ->  online softmax per query lane  ->  O^T += V^T P^T (pattern T on V)
//   backward, queries S^T = K Q^T, dP^T = V dO^T (R on K, V)  ->  dS^T = P^T (dP^T - delta)  ->  dQ^T += K^T dS^T (T on K)
//   backward, keys    S = Q K^T, dP = dO V^T (R on Q, dO)  ->  P, dS  ->  dV^T += dO^T P, dK^T += Q^T dS (T on dO, Q)
#include <stdlib.h>
#include "common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int WPB = 5;                      // waves (16-row tiles) per workgroup: 276 tokens = 18 tiles -> 4 workgroups per (sample, head) = 256 at B = 8
constexpr int NT = WPB * 64;

__device__ __forceinline__ f32x4v mfma4(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// rows [0, n) of TWO (n x D) slices of row-major matrices (leading dimensions lda / ldb) -> LDS [np][D + 4] each, rows [n, np) zero.  All loads of
// a round (6 per thread and matrix) are issued before the first LDS store: the first version (load, store, next) exposed one global round trip per
// iteration, ~6 of the forward's 17 us
template <int D>
__device__ __forceinline__ void stage_pair(float* da, const float* __restrict__ a, int lda, float* db, const float* __restrict__ b, int ldb, int n, int np) {
    constexpr int DP = D + 4, Q = D / 4, U = 6;
    const int total = np * Q;
    for (int i0 = threadIdx.x; i0 < total; i0 += U * NT) {
        f32x4v va[U], vb[U];
        int off[U];
        bool st[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int i = i0 + k * NT;
            st[k] = i < total;
            const int ii = st[k] ? i : 0;
            const int r = ii / Q, c = (ii - r * Q) * 4;
            const bool in = st[k] && r < n;
            const int rr = in ? r : 0;                     // branch-free: a valid row, zeroed by the select below
            va[k] = *reinterpret_cast<const f32x4v*>(a + (size_t)rr * lda + c);
            vb[k] = *reinterpret_cast<const f32x4v*>(b + (size_t)rr * ldb + c);
            if (!in) { va[k] = f32x4v{0.f, 0.f, 0.f, 0.f}; vb[k] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
            off[k] = r * DP + c;
        }
#pragma unroll
        for (int k = 0; k < U; ++k)
            if (st[k]) {
                *reinterpret_cast<f32x4v*>(da + off[k]) = va[k];
                *reinterpret_cast<f32x4v*>(db + off[k]) = vb[k];
            }
    }
}

template <int D>
__global__ __launch_bounds__(NT) void att_fwd_mfma(const float* __restrict__ qkv, int ld, int n, int np, int heads, float scale,
                                                  float* __restrict__ out, int ldo, float* __restrict__ lse) {
    chain_prio();
    constexpr int DP = D + 4, KS = D / 4, DT = (D + 15) / 16;
    extern __shared__ float sm[];
    float* Ks = sm;
    float* Vs = sm + (size_t)np * DP;
    const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
    const float* base = qkv + (size_t)b * n * ld;
    const int inner = heads * D;
    stage_pair<D>(Ks, base + inner + h * D, ld, Vs, base + 2 * inner + h * D, ld, n, np);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int q0 = (blockIdx.y * WPB + wave) * 16;
    if (q0 >= n) return;
    const int qi = q0 + li, qc = qi < n ? qi : n - 1;
    float qf[KS];                                          // B operand of S^T = K Q^T: lane (column q = li, k = kq)
#pragma unroll
    for (int j = 0; j < KS; ++j) qf[j] = base[(size_t)qc * ld + h * D + 4 * j + kq] * scale;
    f32x4v acc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    // online softmax over the key tiles (a two-pass form -- row maximum first, no per-tile cross-lane reductions or rescales -- measured SLOWER:
    // 19.6 vs 17.2 us; its D / 4 extra MFMAs per tile cost more than the two ds_bpermute round trips they save).  Two independent accumulators per
    // score tile: a 16x16x4 MFMA chain on ONE accumulator issues every ~40 cycles.
    const int ntile = np >> 4;
    float m = -3.0e38f, l = 0.f;
    for (int t = 0; t < ntile; ++t) {
        f32x4v s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        const float* kr = Ks + (t * 16 + li) * DP + kq;
#pragma unroll
        for (int j = 0; j < KS; j += 2) {
            s0 = mfma4(kr[4 * j], qf[j], s0);
            if (j + 1 < KS) s1 = mfma4(kr[4 * j + 4], qf[j + 1], s1);
        }
        f32x4v s = s0 + s1;                               // s[r] = S[q = li][key = 16 t + 4 kq + r]
        float mt = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (t * 16 + 4 * kq + r >= n) s[r] = -3.0e38f;
            mt = fmaxf(mt, s[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float mn = fmaxf(m, mt);
        const float corr = __expf(m - mn);
        m = mn;
        float p[4], ps = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { p[r] = __expf(s[r] - mn); ps += p[r]; }
        l = l * corr + ps;                                 // (lane-local: the four k-quad partial sums are added once, after the loop)
        const float* vr = Vs + (t * 16 + 4 * kq) * DP + li;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[dt][r] *= corr;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) acc[dt] = mfma4(vr[jj * DP + 16 * dt], p[jj], acc[dt]);       // O^T[d = 16 dt + 4 kq + r][q = li]
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (qi < n) {
        const float inv = 1.f / l;
        float* o = out + ((size_t)b * n + qi) * ldo + h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = 16 * dt + 4 * kq;
            if (d0 < D) *reinterpret_cast<f32x4v*>(o + d0) = f32x4v{acc[dt][0] * inv, acc[dt][1] * inv, acc[dt][2] * inv, acc[dt][3] * inv};
        }
        if (kq == 0) lse[(size_t)bh * n + qi] = m + __logf(l);
    }
}

// backward, query side: dq_i += scale * sum_j dS_ij K_j ; delta_i = dO_i . O_i is written for the key side
template <int D>
__global__ __launch_bounds__(NT) void att_bwd_q_mfma(const float* __restrict__ qkv, int ld, const float* __restrict__ o, int ldo,
                                                    const float* __restrict__ dout, int lddo, const float* __restrict__ lse, float* __restrict__ delta,
                                                    int n, int np, int heads, float scale, float* __restrict__ dqkv, int lddq) {
    chain_prio();
    constexpr int DP = D + 4, KS = D / 4, DT = (D + 15) / 16;
    extern __shared__ float sm[];
    float* Ks = sm;
    float* Vs = sm + (size_t)np * DP;
    const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
    const float* base = qkv + (size_t)b * n * ld;
    const int inner = heads * D;
    stage_pair<D>(Ks, base + inner + h * D, ld, Vs, base + 2 * inner + h * D, ld, n, np);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int q0 = (blockIdx.y * WPB + wave) * 16;
    if (q0 >= n) return;
    const int qi = q0 + li, qc = qi < n ? qi : n - 1;
    float qf[KS], gf[KS], dl = 0.f;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        qf[j] = base[(size_t)qc * ld + h * D + 4 * j + kq] * scale;
        gf[j] = dout[((size_t)b * n + qc) * lddo + h * D + 4 * j + kq];
        dl += gf[j] * o[((size_t)b * n + qc) * ldo + h * D + 4 * j + kq];
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    const float L = lse[(size_t)bh * n + qc];
    f32x4v acc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int ntile = np >> 4;
    for (int t = 0; t < ntile; ++t) {
        f32x4v s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        const float* kr = Ks + (t * 16 + li) * DP + kq;
        const float* vr = Vs + (t * 16 + li) * DP + kq;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            s = mfma4(kr[4 * j], qf[j], s);                // S^T[key][q]
            dp = mfma4(vr[4 * j], gf[j], dp);              // dP^T[key][q] = V . dO^T
        }
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[r] = (t * 16 + 4 * kq + r < n) ? __expf(s[r] - L) * (dp[r] - dl) : 0.f;
        const float* kt = Ks + (t * 16 + 4 * kq) * DP + li;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[dt] = mfma4(kt[jj * DP + 16 * dt], ds[jj], acc[dt]);       // dQ^T[d][q]
    }
    if (qi < n) {
        if (kq == 0) delta[(size_t)bh * n + qi] = dl;
        float* g = dqkv + ((size_t)b * n + qi) * lddq + h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = 16 * dt + 4 * kq;
            if (d0 < D) {
                f32x4v v = *reinterpret_cast<f32x4v*>(g + d0);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += acc[dt][r] * scale;
                *reinterpret_cast<f32x4v*>(g + d0) = v;
            }
        }
    }
}

// backward, key side: dk_j += scale * sum_i dS_ij Q_i, dv_j += sum_i P_ij dO_i; Q, dO, lse, delta of the head in LDS
template <int D>
__global__ __launch_bounds__(NT) void att_bwd_kv_mfma(const float* __restrict__ qkv, int ld, const float* __restrict__ dout, int lddo,
                                                     const float* __restrict__ lse, const float* __restrict__ delta, int n, int np, int heads,
                                                     float scale, float* __restrict__ dqkv, int lddq) {
    chain_prio();
    constexpr int DP = D + 4, KS = D / 4, DT = (D + 15) / 16;
    extern __shared__ float sm[];
    float* Qs = sm;
    float* Gs = sm + (size_t)np * DP;
    float* Ls = sm + (size_t)2 * np * DP + 16;             // (+16: pattern T of the last rows reads up to 8 floats past Gs)
    float* Ds = Ls + np;
    const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
    const float* base = qkv + (size_t)b * n * ld;
    const int inner = heads * D;
    stage_pair<D>(Qs, base + h * D, ld, Gs, dout + (size_t)b * n * lddo + h * D, lddo, n, np);
    for (int i = threadIdx.x; i < np; i += NT) {
        Ls[i] = i < n ? lse[(size_t)bh * n + i] : 0.f;
        Ds[i] = i < n ? delta[(size_t)bh * n + i] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int k0 = (blockIdx.y * WPB + wave) * 16;
    if (k0 >= n) return;
    const int ki = k0 + li, kc = ki < n ? ki : n - 1;
    float kf[KS], vf[KS];                                  // B operands of S = Q K^T and dP = dO V^T: lane (column key = li, k = kq)
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        kf[j] = base[(size_t)kc * ld + inner + h * D + 4 * j + kq] * scale;
        vf[j] = base[(size_t)kc * ld + 2 * inner + h * D + 4 * j + kq];
    }
    f32x4v dk[DT], dv[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) { dk[t] = f32x4v{0.f, 0.f, 0.f, 0.f}; dv[t] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
    const int ntile = np >> 4;
    for (int t = 0; t < ntile; ++t) {
        f32x4v s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        const float* qr = Qs + (t * 16 + li) * DP + kq;
        const float* gr = Gs + (t * 16 + li) * DP + kq;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            s = mfma4(qr[4 * j], kf[j], s);                // s[r] = S[q = 16 t + 4 kq + r][key = li]
            dp = mfma4(gr[4 * j], vf[j], dp);
        }
        const f32x4v L4 = *reinterpret_cast<const f32x4v*>(Ls + t * 16 + 4 * kq);
        const f32x4v D4 = *reinterpret_cast<const f32x4v*>(Ds + t * 16 + 4 * kq);
        float p[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p[r] = (t * 16 + 4 * kq + r < n) ? __expf(s[r] - L4[r]) : 0.f;
            ds[r] = p[r] * (dp[r] - D4[r]);
        }
        const float* gt = Gs + (t * 16 + 4 * kq) * DP + li;
        const float* qt = Qs + (t * 16 + 4 * kq) * DP + li;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                dv[dt] = mfma4(gt[jj * DP + 16 * dt], p[jj], dv[dt]);       // dV^T[d][key]
                dk[dt] = mfma4(qt[jj * DP + 16 * dt], ds[jj], dk[dt]);      // dK^T[d][key]
            }
    }
    if (ki < n) {
        float* gk = dqkv + ((size_t)b * n + ki) * lddq + inner + h * D;
        float* gv = dqkv + ((size_t)b * n + ki) * lddq + 2 * inner + h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d0 = 16 * dt + 4 * kq;
            if (d0 < D) {
                f32x4v a = *reinterpret_cast<f32x4v*>(gk + d0), c = *reinterpret_cast<f32x4v*>(gv + d0);
#pragma unroll
                for (int r = 0; r < 4; ++r) { a[r] += dk[dt][r] * scale; c[r] += dv[dt][r]; }
                *reinterpret_cast<f32x4v*>(gk + d0) = a;
                *reinterpret_cast<f32x4v*>(gv + d0) = c;
            }
        }
    }
}

int g_att_mfma = -1;

template <typename K>
int raise_lds(K kernel, size_t lds) {
    if (lds <= 64 * 1024) return 0;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 0 : 1;
}

}  // namespace

int mrfa_tuning_attention_mfma(int set) {
    if (g_att_mfma < 0) { const char* e = getenv("MRFA_ATTENTION_MFMA"); g_att_mfma = !(e && e[0] == '0'); }
    const int prev = g_att_mfma;
    if (set >= 0) g_att_mfma = set != 0;
    return prev;
}

// the MFMA kernels take the call: float4 accesses on every operand (16-byte aligned views, leading dimensions % 4 == 0) and the other side's operands of a
// head -- tokens padded to a multiple of 16, + 16 floats, + 2 np floats in the key-side backward (dout != NULL) -- within the 160 KB of LDS.  Longer token
// lists stay on the VALU kernels of tokenpose.hip, whose LDS image is smaller (no padding): a call they can serve never fails here
bool mrfa_attention_mfma_ok(int d, int n, const void* qkv, int ld, const void* out, int ldo, const void* dout, int lddo, const void* dqkv, int lddq) {
    auto ok = [](const void* p, int l) { return p == nullptr || (aligned16(p) && (l % 4) == 0); };
    const size_t np = (size_t)(n + 15) / 16 * 16;
    const size_t lds = (2 * np * (d + 4) + 16 + (dout ? 2 * np : 0)) * sizeof(float);
    return mrfa_tuning_attention_mfma(-1) && (d == 16 || d == 24 || d == 32) && lds <= 160 * 1024 && ok(qkv, ld) && ok(out, ldo) && ok(dout, lddo) &&
           ok(dqkv, lddq);
}

#define ATT_MFMA_DISPATCH(KERNEL, LDS, ...)                                                                       \
    do {                                                                                                          \
        int rc_ = 0;                                                                                              \
        if (d == 24) { rc_ = raise_lds(KERNEL<24>, LDS); if (!rc_) hipLaunchKernelGGL((KERNEL<24>), grid, dim3(NT), LDS, st, __VA_ARGS__); }      \
        else if (d == 16) { rc_ = raise_lds(KERNEL<16>, LDS); if (!rc_) hipLaunchKernelGGL((KERNEL<16>), grid, dim3(NT), LDS, st, __VA_ARGS__); } \
        else { rc_ = raise_lds(KERNEL<32>, LDS); if (!rc_) hipLaunchKernelGGL((KERNEL<32>), grid, dim3(NT), LDS, st, __VA_ARGS__); }              \
        if (rc_) { mrfa_set_error("attention(mfma): cannot reserve %zu bytes of LDS", (size_t)(LDS)); return 1; }  \
    } while (0)

int mrfa_attention_fwd_mfma(hipStream_t st, const float* qkv, int ld, int B, int n, int heads, int d, float scale, float* out, int ldo, float* lse) {
    const int np = (n + 15) / 16 * 16;
    const size_t lds = ((size_t)2 * np * (d + 4) + 16) * sizeof(float);
    if (lds > 160 * 1024) { mrfa_set_error("attention_fwd: %d tokens x %d do not fit the 160 KB LDS", n, d); return 1; }
    const dim3 grid(B * heads, cdiv(np / 16, WPB));
    ATT_MFMA_DISPATCH(att_fwd_mfma, lds, qkv, ld, n, np, heads, scale, out, ldo, lse);
    MRFA_CHECK_LAUNCH("attention_fwd(mfma)");
    return 0;
}

int mrfa_attention_bwd_mfma(hipStream_t st, const float* qkv, int ld, const float* out, int ldo, const float* dout, int lddo, const float* lse,
                            float* delta, int B, int n, int heads, int d, float scale, float* dqkv, int lddq) {
    const int np = (n + 15) / 16 * 16;
    const dim3 grid(B * heads, cdiv(np / 16, WPB));
    const size_t lds_q = ((size_t)2 * np * (d + 4) + 16) * sizeof(float);
    const size_t lds_kv = ((size_t)2 * np * (d + 4) + 16 + 2 * np) * sizeof(float);
    if (lds_kv > 160 * 1024) { mrfa_set_error("attention_bwd: %d tokens x %d do not fit the 160 KB LDS", n, d); return 1; }
    ATT_MFMA_DISPATCH(att_bwd_q_mfma, lds_q, qkv, ld, out, ldo, dout, lddo, lse, delta, n, np, heads, scale, dqkv, lddq);
    MRFA_CHECK_LAUNCH("attention_bwd(q, mfma)");
    ATT_MFMA_DISPATCH(att_bwd_kv_mfma, lds_kv, qkv, ld, dout, lddo, lse, delta, n, np, heads, scale, dqkv, lddq);
    MRFA_CHECK_LAUNCH("attention_bwd(kv, mfma)");
    return 0;
}
