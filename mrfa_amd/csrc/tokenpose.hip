// K21: the kernels the MTIA prior (TokenPose_B = HRNet-W32 stem + 12-layer ViT; reference
// modules/transformer/{hr_base,tokenpose_base,pose_tokenpose_b}.py) needs beyond the convolution / BatchNorm kernels
// it shares with the rest of the path:
//   * stride-2 pixel subsampling (a stride-2 3x3 conv = the stride-1 conv kept at even pixels, hr_base.py:231,312,316),
//   * nearest-neighbour upsample fused with the branch sum and the ReLU of HighResolutionModule.forward
//     (hr_base.py:278-289), also used with factor 1 as a plain add + ReLU,
//   * LayerNorm over token rows (tokenpose_base.py:33,38), exact-erf GELU (tokenpose_base.py:51),
//   * multi-head attention over 276 tokens x 8 heads x 24 (tokenpose_base.py:72-94) computed per (sample, head) from the
//     packed qkv rows, softmax never materialised in HBM.
// All of it is HBM- or latency-bound work on <= 4 MB tensors; float4 accesses, one pass per tensor.
#include "common.h"

namespace {

#define GRID_STRIDE_U(i, n) for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += gridDim.x * blockDim.x)

// ---------------------------------------------------------------------------------------------- subsample
// y[n,oy,ox,:] = x[n,oy*s,ox*s,:]        (ACC: dx[n,oy*s,ox*s,:] += dy[n,oy,ox,:], called with the roles swapped)
template <bool SCATTER>
__global__ void subsample_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int H, int W, int Ho, int Wo,
                                 int C4, int s, unsigned total4) {
    chain_prio();
    GRID_STRIDE_U(i, total4) {
        const unsigned opix = i / (unsigned)C4;
        const unsigned c = (i - opix * (unsigned)C4) * 4u;
        const unsigned ox = opix % (unsigned)Wo, t = opix / (unsigned)Wo;
        const unsigned oy = t % (unsigned)Ho, n = t / (unsigned)Ho;
        const size_t big = ((size_t)n * H + oy * s) * W + ox * s;
        if (!SCATTER) {
            *reinterpret_cast<f32x4*>(dst + (size_t)opix * ldd + c) = *reinterpret_cast<const f32x4*>(src + big * lds + c);
        } else {
            f32x4* q = reinterpret_cast<f32x4*>(dst + big * ldd + c);
            f32x4 v = *q;
            v += *reinterpret_cast<const f32x4*>(src + (size_t)opix * lds + c);
            *q = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------- upsample + add + act
__global__ void ups_add_act_fwd_kernel(const float* __restrict__ lo, int ldl, int Hl, int Wl, int C4, int f, const float* __restrict__ base,
                                       int ldb, int relu, float* __restrict__ y, int ldy, unsigned total4) {
    chain_prio();
    const int H = Hl * f, W = Wl * f;
    GRID_STRIDE_U(i, total4) {
        const unsigned pix = i / (unsigned)C4;
        const unsigned c = (i - pix * (unsigned)C4) * 4u;
        const unsigned xx = pix % (unsigned)W, t = pix / (unsigned)W;
        const unsigned yy = t % (unsigned)H, n = t / (unsigned)H;
        const size_t lp = ((size_t)n * Hl + yy / f) * Wl + xx / f;
        f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)pix * ldb + c);
        v += *reinterpret_cast<const f32x4*>(lo + lp * ldl + c);
        if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        *reinterpret_cast<f32x4*>(y + (size_t)pix * ldy + c) = v;
    }
}

// one thread per low-resolution float4: walks its f x f block of the output gradient
__global__ void ups_add_act_bwd_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ dy, int lddy, int Hl, int Wl, int C4,
                                       int f, int relu, float* __restrict__ dlo, int lddl, float* dbase, int lddb, unsigned total4) {
    chain_prio();
    const int W = Wl * f, H = Hl * f;
    GRID_STRIDE_U(i, total4) {
        const unsigned lp = i / (unsigned)C4;
        const unsigned c = (i - lp * (unsigned)C4) * 4u;
        const unsigned xl = lp % (unsigned)Wl, t = lp / (unsigned)Wl;
        const unsigned yl = t % (unsigned)Hl, n = t / (unsigned)Hl;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < f; ++a) {
            for (int b = 0; b < f; ++b) {
                const size_t pix = ((size_t)n * H + yl * f + a) * W + xl * f + b;
                f32x4 g = *reinterpret_cast<const f32x4*>(dy + pix * lddy + c);
                if (relu) {
                    const f32x4 yy = *reinterpret_cast<const f32x4*>(y + pix * ldy + c);
#pragma unroll
                    for (int k = 0; k < 4; ++k) g[k] = yy[k] > 0.f ? g[k] : 0.f;
                }
                acc += g;
                if (dbase) {
                    f32x4* q = reinterpret_cast<f32x4*>(dbase + pix * lddb + c);
                    f32x4 cur = *q;
                    cur += g;
                    *q = cur;
                }
            }
        }
        if (dlo) {
            f32x4* q = reinterpret_cast<f32x4*>(dlo + (size_t)lp * lddl + c);
            f32x4 cur = *q;
            cur += acc;
            *q = cur;
        }
    }
}

// ---------------------------------------------------------------------------------------------- LayerNorm
// one wave per row, C <= 1024; biased variance, eps inside the square root (torch.nn.LayerNorm)
constexpr int LN_MAXK = 16;

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int ldx, long long rows, int C,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                           float* __restrict__ y, int ldy, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out) {
    chain_prio();
    const int lane = threadIdx.x & 63;
    const long long r = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    if (r >= rows) return;
    const float* xr = x + (size_t)r * ldx;
    float v[LN_MAXK];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXK; ++k) {
        const int c = k * 64 + lane;
        v[k] = c < C ? xr[c] : 0.f;
        s += v[k];
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXK; ++k) {
        const int c = k * 64 + lane;
        const float d = c < C ? v[k] - mean : 0.f;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int k = 0; k < LN_MAXK; ++k) {
        const int c = k * 64 + lane;
        if (c < C) y[(size_t)r * ldy + c] = (v[k] - mean) * rstd * gamma[c] + beta[c];
    }
    if (lane == 0) {
        mean_out[r] = mean;
        rstd_out[r] = rstd;
    }
}

// dx += rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy*gamma ; dgamma += sum_r dy*xhat ; dbeta += sum_r dy
// KMAX = 64-channel chunks a lane holds (4: C <= 256, the token transformer's 192; 16: C <= 1024); RU = rows of a wave whose loads are issued together (the
// loop was one row at a time: eight dependent load -> shuffle-reduce -> store round trips per wave, 59 us per launch over 4 416 x 192 in the training step).
// Parameter gradients: waves -> LDS -> per workgroup ONE atomic per channel, into slot (workgroup % LN_SLOTS) of `scratch` [LN_SLOTS][2][C] (138-276
// workgroups adding into the same 2 C words serialise at the memory side); the launch's last workgroup (ticket word behind the slots; the hand-off of
// common.h: fused_bn_finalize) sums the slots and adds them to dgamma / dbeta.  scratch == NULL: the atomics go to dgamma / dbeta directly.
constexpr int LN_SLOTS = MRFA_LN_SLOTS;
template <int KMAX, int RU>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                                                           long long rows, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           float* __restrict__ dx, int lddx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int rows_per_wave, float* __restrict__ scratch) {
    chain_prio();
    __shared__ float red[2][4][64 * 4];            // per wave partials for up to 256 channels at a time
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long w_id = (long long)blockIdx.x * 4 + wave;
    const long long r0 = w_id * rows_per_wave, r1 = min(rows, r0 + rows_per_wave);
    float pg[KMAX], pb[KMAX], gm[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        pg[k] = pb[k] = 0.f;
        gm[k] = (k * 64 + lane) < C ? gamma[k * 64 + lane] : 0.f;
    }
    for (long long rb = r0; rb < r1; rb += RU) {
        float xv[RU][KMAX], dv[RU][KMAX], ov[RU][KMAX], m[RU], rs[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const long long r = min(rb + u, r1 - 1);                      // (past the end: a re-read of the last row, never stored)
            m[u] = mean[r];
            rs[u] = rstd[r];
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int c = k * 64 + lane;
                const bool ok = c < C;
                xv[u][k] = ok ? x[(size_t)r * ldx + c] : 0.f;
                dv[u][k] = ok ? dy[(size_t)r * lddy + c] : 0.f;
                ov[u][k] = ok ? dx[(size_t)r * lddx + c] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const bool live = rb + u < r1;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const float xh = (xv[u][k] - m[u]) * rs[u];
                const float g = dv[u][k] * gm[k];
                if (live) { pg[k] += dv[u][k] * xh; pb[k] += dv[u][k]; }
                s1 += g;
                s2 += g * xh;
                xv[u][k] = xh;
                dv[u][k] = g;
            }
            const float k1 = wave_sum(s1) / (float)C, k2 = wave_sum(s2) / (float)C;
            if (live) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    const int c = k * 64 + lane;
                    if (c < C) dx[(size_t)(rb + u) * lddx + c] = ov[u][k] + rs[u] * (dv[u][k] - k1 - xv[u][k] * k2);
                }
            }
        }
    }
    // parameter gradients: waves of the block -> LDS -> one atomic per channel per block, 256 channels per round
    float* ag = scratch ? scratch + (size_t)(blockIdx.x % LN_SLOTS) * 2 * C : dgamma;
    float* ab = scratch ? ag + C : dbeta;
    for (int k0 = 0; k0 * 64 < C; k0 += 4) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k == k0 + kk) { a = pg[k]; b = pb[k]; }
            red[0][wave][kk * 64 + lane] = a;
            red[1][wave][kk * 64 + lane] = b;
        }
        __syncthreads();
        const int c = k0 * 64 + threadIdx.x;
        if (c < C) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a += red[0][w][threadIdx.x]; b += red[1][w][threadIdx.x]; }
            if (ag) atomicAdd(ag + c, a);
            if (ab) atomicAdd(ab + c, b);
        }
        __syncthreads();
    }
    if (scratch) {
        __shared__ unsigned s_ticket;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            s_ticket = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(scratch + (size_t)LN_SLOTS * 2 * C), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_ticket == gridDim.x - 1) {
            for (int c = threadIdx.x; c < 2 * C; c += 256) {
                float t = 0.f;
                for (int sl = 0; sl < LN_SLOTS; ++sl) t += __hip_atomic_load(scratch + (size_t)sl * 2 * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                float* dst = c < C ? dgamma : dbeta;
                if (dst) atomicAdd(dst + (c < C ? c : c - C), t);      // (2 C atomics per launch: two launches of one LayerNorm on different streams may overlap)
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- GELU (exact, erf)
__global__ void gelu_fwd_kernel(const float* __restrict__ x, int ldx, int C4, float* __restrict__ y, int ldy, unsigned total4) {
    chain_prio();
    GRID_STRIDE_U(i, total4) {
        const unsigned r = i / (unsigned)C4;
        const unsigned c = (i - r * (unsigned)C4) * 4u;
        f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r * ldx + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = 0.5f * v[k] * (1.f + erff(v[k] * 0.70710678118654752440f));
        *reinterpret_cast<f32x4*>(y + (size_t)r * ldy + c) = v;
    }
}

__global__ void gelu_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy, int C4, float* __restrict__ dx,
                                int lddx, unsigned total4) {
    chain_prio();
    GRID_STRIDE_U(i, total4) {
        const unsigned r = i / (unsigned)C4;
        const unsigned c = (i - r * (unsigned)C4) * 4u;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r * ldx + c);
        const f32x4 g = *reinterpret_cast<const f32x4*>(dy + (size_t)r * lddy + c);
        f32x4* q = reinterpret_cast<f32x4*>(dx + (size_t)r * lddx + c);
        f32x4 cur = *q;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float cdf = 0.5f * (1.f + erff(v[k] * 0.70710678118654752440f));
            const float pdf = 0.39894228040143267794f * __expf(-0.5f * v[k] * v[k]);
            cur[k] += g[k] * (cdf + v[k] * pdf);
        }
        *q = cur;
    }
}

// ---------------------------------------------------------------------------------------------- attention
// qkv: (B*n) rows x (3*heads*d) floats, q | k | v thirds, head h = columns [h*d, (h+1)*d) of its third ('b n (h d)',
// tokenpose_base.py:77-78).  One workgroup (512 threads) = one (sample, head) x 32 query rows; K and V of the head live
// in LDS.  A query row is shared by 16 lanes ("parts"), part p owning keys j = p, p+16, ...: 8 waves per workgroup and
// two workgroups per CU keep 4 waves on every SIMD, which hides the LDS latency a one-thread-per-row layout exposes
// (that version ran with 1 wave per SIMD and took 146 us per launch).  The per-row partial results (max, sum, P.V) are
// combined with xor-shuffles inside the 16-lane group.  LDS rows are padded to d+4 floats: the 16 distinct row
// addresses of a wave's ds_read_b128 then fall into 16 disjoint 4-bank groups.
constexpr int ATT_ROWS = 32;
constexpr int ATT_PARTS = 16;
constexpr int ATT_THREADS = ATT_ROWS * ATT_PARTS;

__device__ __forceinline__ float part_sum(float v) {
#pragma unroll
    for (int o = 1; o < ATT_PARTS; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float part_max(float v) {
#pragma unroll
    for (int o = 1; o < ATT_PARTS; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

template <int D>
__device__ __forceinline__ void att_stage(float* dst, const float* __restrict__ src, int ld, int n) {
    constexpr int DP = D + 4;
    for (int i = threadIdx.x; i < n * (D / 4); i += ATT_THREADS) {
        const int r = i / (D / 4), c = (i - r * (D / 4)) * 4;
        *reinterpret_cast<f32x4*>(dst + r * DP + c) = *reinterpret_cast<const f32x4*>(src + (size_t)r * ld + c);
    }
}

template <int D>
__device__ __forceinline__ float dot_lds(const float (&a)[D], const float* row) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        const f32x4 k = *reinterpret_cast<const f32x4*>(row + c);
        s += a[c] * k[0] + a[c + 1] * k[1] + a[c + 2] * k[2] + a[c + 3] * k[3];
    }
    return s;
}

template <int D>
__device__ __forceinline__ void axpy_lds(float (&acc)[D], float w, const float* row) {
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
        acc[c] += w * v[0]; acc[c + 1] += w * v[1]; acc[c + 2] += w * v[2]; acc[c + 3] += w * v[3];
    }
}

template <int D>
__global__ __launch_bounds__(ATT_THREADS) void attention_fwd_kernel(const float* __restrict__ qkv, int ld, int n, int heads, float scale,
                                                                  float* __restrict__ out, int ldo, float* __restrict__ lse) {
    constexpr int DP = D + 4;
    extern __shared__ float sm[];
    float* Ks = sm;
    float* Vs = sm + (size_t)n * DP;
    const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
    const float* base = qkv + (size_t)b * n * ld;
    const int inner = heads * D;
    att_stage<D>(Ks, base + inner + h * D, ld, n);
    att_stage<D>(Vs, base + 2 * inner + h * D, ld, n);
    __syncthreads();
    const int part = threadIdx.x & (ATT_PARTS - 1);
    const int i = blockIdx.y * ATT_ROWS + (threadIdx.x >> 4);
    const bool ok = i < n;                                   // whole 16-lane groups share i: shuffles stay inside live groups
    const int ii = ok ? i : n - 1;
    float q[D];
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)ii * ld + h * D + c);
        q[c] = v[0] * scale; q[c + 1] = v[1] * scale; q[c + 2] = v[2] * scale; q[c + 3] = v[3] * scale;
    }
    float m = -3.0e38f;
    for (int j = part; j < n; j += ATT_PARTS) m = fmaxf(m, dot_lds<D>(q, Ks + j * DP));
    m = part_max(m);
    float l = 0.f, acc[D];
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.f;
    for (int j = part; j < n; j += ATT_PARTS) {
        const float p = __expf(dot_lds<D>(q, Ks + j * DP) - m);
        l += p;
        axpy_lds<D>(acc, p, Vs + j * DP);
    }
    l = part_sum(l);
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = part_sum(acc[c]);
    if (ok && part == 0) {
        const float inv = 1.f / l;
        float* o = out + ((size_t)b * n + i) * ldo + h * D;
#pragma unroll
        for (int c = 0; c < D; ++c) o[c] = acc[c] * inv;
        lse[(size_t)bh * n + i] = m + __logf(l);
    }
}

// backward, query side: dq_i += scale * sum_j dS_ij K_j ; delta_i = dO_i . O_i is also written for the key side
template <int D>
__global__ __launch_bounds__(ATT_THREADS) void attention_bwd_q_kernel(const float* __restrict__ qkv, int ld, const float* __restrict__ o, int ldo,
                                                                    const float* __restrict__ dout, int lddo, const float* __restrict__ lse,
                                                                    float* __restrict__ delta, int n, int heads, float scale,
                                                                    float* __restrict__ dqkv, int lddq) {
    constexpr int DP = D + 4;
    extern __shared__ float sm[];
    float* Ks = sm;
    float* Vs = sm + (size_t)n * DP;
    const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
    const float* base = qkv + (size_t)b * n * ld;
    const int inner = heads * D;
    att_stage<D>(Ks, base + inner + h * D, ld, n);
    att_stage<D>(Vs, base + 2 * inner + h * D, ld, n);
    __syncthreads();
    const int part = threadIdx.x & (ATT_PARTS - 1);
    const int i = blockIdx.y * ATT_ROWS + (threadIdx.x >> 4);
    const bool ok = i < n;
    const int ii = ok ? i : n - 1;
    float q[D], dO[D], dq[D];
    float dl = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        q[c] = base[(size_t)ii * ld + h * D + c] * scale;
        dO[c] = dout[((size_t)b * n + ii) * lddo + h * D + c];
        dl += dO[c] * o[((size_t)b * n + ii) * ldo + h * D + c];
        dq[c] = 0.f;
    }
    const float L = lse[(size_t)bh * n + ii];
    for (int j = part; j < n; j += ATT_PARTS) {
        const float s = dot_lds<D>(q, Ks + j * DP);
        const float dp = dot_lds<D>(dO, Vs + j * DP);
        axpy_lds<D>(dq, __expf(s - L) * (dp - dl), Ks + j * DP);
    }
#pragma unroll
    for (int c = 0; c < D; ++c) dq[c] = part_sum(dq[c]);
    if (ok && part == 0) {
        delta[(size_t)bh * n + i] = dl;
        float* g = dqkv + ((size_t)b * n + i) * lddq + h * D;
#pragma unroll
        for (int c = 0; c < D; ++c) g[c] += dq[c] * scale;
    }
}

// backward, key side: 16 lanes share key/value row j, part p owning queries i = p, p+16, ...; Q and dO of the head in LDS
template <int D>
__global__ __launch_bounds__(ATT_THREADS) void attention_bwd_kv_kernel(const float* __restrict__ qkv, int ld, const float* __restrict__ dout,
                                                                     int lddo, const float* __restrict__ lse, const float* __restrict__ delta,
                                                                     int n, int heads, float scale, float* __restrict__ dqkv, int lddq) {
    constexpr int DP = D + 4;
    extern __shared__ float sm[];
    float* Qs = sm;
    float* Gs = sm + (size_t)n * DP;
    float* Ls = sm + (size_t)2 * n * DP;
    float* Ds = Ls + n;
    const int bh = blockIdx.x, b = bh / heads, h = bh - b * heads;
    const float* base = qkv + (size_t)b * n * ld;
    const int inner = heads * D;
    att_stage<D>(Qs, base + h * D, ld, n);
    att_stage<D>(Gs, dout + (size_t)b * n * lddo + h * D, lddo, n);
    for (int i = threadIdx.x; i < n; i += ATT_THREADS) {
        Ls[i] = lse[(size_t)bh * n + i];
        Ds[i] = delta[(size_t)bh * n + i];
    }
    __syncthreads();
    const int part = threadIdx.x & (ATT_PARTS - 1);
    const int j = blockIdx.y * ATT_ROWS + (threadIdx.x >> 4);
    const bool ok = j < n;
    const int jj = ok ? j : n - 1;
    float k[D], v[D], dk[D], dv[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        k[c] = base[(size_t)jj * ld + inner + h * D + c] * scale;
        v[c] = base[(size_t)jj * ld + 2 * inner + h * D + c];
        dk[c] = dv[c] = 0.f;
    }
    for (int i = part; i < n; i += ATT_PARTS) {
        const float s = dot_lds<D>(k, Qs + i * DP);
        const float dp = dot_lds<D>(v, Gs + i * DP);
        const float p = __expf(s - Ls[i]);
        axpy_lds<D>(dv, p, Gs + i * DP);
        axpy_lds<D>(dk, p * (dp - Ds[i]), Qs + i * DP);
    }
#pragma unroll
    for (int c = 0; c < D; ++c) {
        dk[c] = part_sum(dk[c]);
        dv[c] = part_sum(dv[c]);
    }
    if (ok && part == 0) {
        float* gk = dqkv + ((size_t)b * n + j) * lddq + inner + h * D;
        float* gv = dqkv + ((size_t)b * n + j) * lddq + 2 * inner + h * D;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            gk[c] += dk[c] * scale;
            gv[c] += dv[c];
        }
    }
}

bool vec_ok(const void* p, int ld) { return aligned16(p) && (ld % 4) == 0; }

}  // namespace

extern "C" int mrfa_subsample_fwd(void* stream, const float* x, int ldx, int N, int H, int W, int C, int stride, float* y, int ldy) {
    MRFA_CHECK_ARG(x && y && N > 0 && C > 0 && stride >= 1 && H % stride == 0 && W % stride == 0, "subsample_fwd: bad args");
    MRFA_CHECK_ARG(C % 4 == 0 && vec_ok(x, ldx) && vec_ok(y, ldy), "subsample_fwd: needs C %% 4 == 0 and 16-byte aligned views");
    const int Ho = H / stride, Wo = W / stride;
    const long long total4 = (long long)N * Ho * Wo * (C / 4);
    MRFA_CHECK_ARG(total4 < (1ll << 31), "subsample_fwd: tensor too large");
    hipLaunchKernelGGL((subsample_kernel<false>), dim3(stream_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, H, W, Ho, Wo,
                       C / 4, stride, (unsigned)total4);
    MRFA_CHECK_LAUNCH("subsample_fwd");
    return 0;
}

extern "C" int mrfa_subsample_bwd(void* stream, const float* dy, int lddy, int N, int H, int W, int C, int stride, float* dx, int lddx) {
    MRFA_CHECK_ARG(dy && dx && N > 0 && C > 0 && stride >= 1 && H % stride == 0 && W % stride == 0, "subsample_bwd: bad args");
    MRFA_CHECK_ARG(C % 4 == 0 && vec_ok(dy, lddy) && vec_ok(dx, lddx), "subsample_bwd: needs C %% 4 == 0 and 16-byte aligned views");
    const int Ho = H / stride, Wo = W / stride;
    const long long total4 = (long long)N * Ho * Wo * (C / 4);
    MRFA_CHECK_ARG(total4 < (1ll << 31), "subsample_bwd: tensor too large");
    hipLaunchKernelGGL((subsample_kernel<true>), dim3(stream_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, dy, lddy, dx, lddx, H, W, Ho, Wo,
                       C / 4, stride, (unsigned)total4);
    MRFA_CHECK_LAUNCH("subsample_bwd");
    return 0;
}

extern "C" int mrfa_upsample_add_act_fwd(void* stream, const float* lo, int ldl, int N, int Hl, int Wl, int C, int factor, const float* base,
                                         int ldb, int relu, float* y, int ldy) {
    MRFA_CHECK_ARG(lo && base && y && N > 0 && C > 0 && factor >= 1, "upsample_add_act_fwd: bad args");
    MRFA_CHECK_ARG(C % 4 == 0 && vec_ok(lo, ldl) && vec_ok(base, ldb) && vec_ok(y, ldy), "upsample_add_act_fwd: needs C %% 4 == 0, aligned views");
    const long long total4 = (long long)N * Hl * factor * Wl * factor * (C / 4);
    MRFA_CHECK_ARG(total4 < (1ll << 31), "upsample_add_act_fwd: tensor too large");
    hipLaunchKernelGGL(ups_add_act_fwd_kernel, dim3(stream_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, lo, ldl, Hl, Wl, C / 4, factor,
                       base, ldb, relu, y, ldy, (unsigned)total4);
    MRFA_CHECK_LAUNCH("upsample_add_act_fwd");
    return 0;
}

extern "C" int mrfa_upsample_add_act_bwd(void* stream, const float* y, int ldy, const float* dy, int lddy, int N, int Hl, int Wl, int C,
                                         int factor, int relu, float* dlo, int lddl, float* dbase, int lddb) {
    MRFA_CHECK_ARG(dy && (y || !relu) && N > 0 && C > 0 && factor >= 1, "upsample_add_act_bwd: bad args");
    MRFA_CHECK_ARG(C % 4 == 0 && vec_ok(dy, lddy) && (!relu || vec_ok(y, ldy)) && (!dlo || vec_ok(dlo, lddl)) && (!dbase || vec_ok(dbase, lddb)),
                   "upsample_add_act_bwd: needs C %% 4 == 0, aligned views");
    const long long total4 = (long long)N * Hl * Wl * (C / 4);
    MRFA_CHECK_ARG(total4 * factor * factor < (1ll << 31), "upsample_add_act_bwd: tensor too large");
    hipLaunchKernelGGL(ups_add_act_bwd_kernel, dim3(stream_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, y, ldy, dy, lddy, Hl, Wl, C / 4,
                       factor, relu, dlo, lddl, dbase, lddb, (unsigned)total4);
    MRFA_CHECK_LAUNCH("upsample_add_act_bwd");
    return 0;
}

extern "C" int mrfa_layernorm_fwd(void* stream, const float* x, int ldx, long long rows, int C, const float* gamma, const float* beta, float eps,
                                  float* y, int ldy, float* mean, float* rstd) {
    MRFA_CHECK_ARG(x && y && gamma && beta && mean && rstd && rows > 0 && C > 0 && C <= 64 * LN_MAXK, "layernorm_fwd: bad args (C <= 1024)");
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, C, gamma, beta, eps, y, ldy,
                       mean, rstd);
    MRFA_CHECK_LAUNCH("layernorm_fwd");
    return 0;
}

extern "C" int mrfa_layernorm_bwd(void* stream, const float* x, int ldx, const float* dy, int lddy, long long rows, int C, const float* gamma,
                                  const float* mean, const float* rstd, float* dx, int lddx, float* dgamma, float* dbeta, float* scratch) {
    MRFA_CHECK_ARG(x && dy && gamma && mean && rstd && dx && rows > 0 && C > 0 && C <= 64 * LN_MAXK, "layernorm_bwd: bad args (C <= 1024)");
    // rows per wave: with the slotted parameter-gradient partials more workgroups cost no more atomic contention
    const int rpw = scratch ? (rows >= 2048 ? 4 : 2) : (rows >= 4096 ? 8 : 2);
    const dim3 grid(cdiv(rows, 4 * rpw));
    if (C <= 256)
        hipLaunchKernelGGL((layernorm_bwd_kernel<4, 4>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, dy, lddy, rows, C, gamma, mean, rstd, dx, lddx, dgamma,
                           dbeta, rpw, scratch);
    else
        hipLaunchKernelGGL((layernorm_bwd_kernel<16, 1>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, dy, lddy, rows, C, gamma, mean, rstd, dx, lddx, dgamma,
                           dbeta, rpw, scratch);
    MRFA_CHECK_LAUNCH("layernorm_bwd");
    return 0;
}

extern "C" int mrfa_gelu_fwd(void* stream, const float* x, int ldx, long long rows, int C, float* y, int ldy) {
    MRFA_CHECK_ARG(x && y && rows > 0 && C > 0, "gelu_fwd: bad args");
    MRFA_CHECK_ARG(C % 4 == 0 && vec_ok(x, ldx) && vec_ok(y, ldy) && rows * (C / 4) < (1ll << 31), "gelu_fwd: needs C %% 4 == 0, aligned views");
    const long long total4 = rows * (C / 4);
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(stream_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, C / 4, y, ldy, (unsigned)total4);
    MRFA_CHECK_LAUNCH("gelu_fwd");
    return 0;
}

extern "C" int mrfa_gelu_bwd(void* stream, const float* x, int ldx, const float* dy, int lddy, long long rows, int C, float* dx, int lddx) {
    MRFA_CHECK_ARG(x && dy && dx && rows > 0 && C > 0, "gelu_bwd: bad args");
    MRFA_CHECK_ARG(C % 4 == 0 && vec_ok(x, ldx) && vec_ok(dy, lddy) && vec_ok(dx, lddx) && rows * (C / 4) < (1ll << 31),
                   "gelu_bwd: needs C %% 4 == 0, aligned views");
    const long long total4 = rows * (C / 4);
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(stream_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, dy, lddy, C / 4, dx, lddx,
                       (unsigned)total4);
    MRFA_CHECK_LAUNCH("gelu_bwd");
    return 0;
}

#define ATT_DISPATCH(D_, KERNEL, ...)                                                                                         \
    do {                                                                                                                      \
        if (d == 24) hipLaunchKernelGGL((KERNEL<24>), grid, dim3(ATT_THREADS), lds, (hipStream_t)stream, __VA_ARGS__);           \
        else if (d == 16) hipLaunchKernelGGL((KERNEL<16>), grid, dim3(ATT_THREADS), lds, (hipStream_t)stream, __VA_ARGS__);      \
        else hipLaunchKernelGGL((KERNEL<32>), grid, dim3(ATT_THREADS), lds, (hipStream_t)stream, __VA_ARGS__);                   \
    } while (0)

static int att_check(const char* what, int B, int n, int heads, int d, size_t lds_floats) {
    if (B <= 0 || n <= 0 || heads <= 0 || !(d == 16 || d == 24 || d == 32)) {
        mrfa_set_error("%s: bad shape B=%d n=%d heads=%d d=%d (d in {16, 24, 32})", what, B, n, heads, d);
        return 1;
    }
    if (lds_floats * 4 > 160 * 1024) {
        mrfa_set_error("%s: %d tokens x %d do not fit the 160 KB LDS", what, n, d);
        return 1;
    }
    return 0;
}

template <typename K>
static int att_attr(K kernel, size_t lds) {
    if (lds <= 64 * 1024) return 0;                     // within the default dynamic-LDS limit: nothing to raise
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 0 : 1;
}

extern "C" int mrfa_attention_fwd(void* stream, const float* qkv, int ld, int B, int n, int heads, int d, float scale, float* out, int ldo,
                                  float* lse) {
    MRFA_CHECK_ARG(qkv && out && lse && vec_ok(qkv, ld), "attention_fwd: null or misaligned pointer");
    if (att_check("attention_fwd", B, n, heads, d, (size_t)2 * n * (d + 4))) return 1;
    if (mrfa_attention_mfma_ok(d, n, qkv, ld, out, ldo, nullptr, 0, nullptr, 0))          // the matrix-pipe kernels (attention_mfma.hip)
        return mrfa_attention_fwd_mfma((hipStream_t)stream, qkv, ld, B, n, heads, d, scale, out, ldo, lse);
    const size_t lds = (size_t)2 * n * (d + 4) * sizeof(float);
    const dim3 grid(B * heads, cdiv(n, ATT_ROWS));
    int rc = d == 24 ? att_attr(attention_fwd_kernel<24>, lds) : d == 16 ? att_attr(attention_fwd_kernel<16>, lds) : att_attr(attention_fwd_kernel<32>, lds);
    MRFA_CHECK_ARG(rc == 0, "attention_fwd: cannot reserve %zu bytes of LDS", lds);
    ATT_DISPATCH(d, attention_fwd_kernel, qkv, ld, n, heads, scale, out, ldo, lse);
    MRFA_CHECK_LAUNCH("attention_fwd");
    return 0;
}

extern "C" int mrfa_attention_bwd(void* stream, const float* qkv, int ld, const float* out, int ldo, const float* dout, int lddo, const float* lse,
                                  float* delta, int B, int n, int heads, int d, float scale, float* dqkv, int lddq) {
    MRFA_CHECK_ARG(qkv && out && dout && lse && delta && dqkv && vec_ok(qkv, ld) && vec_ok(dout, lddo),
                   "attention_bwd: null or misaligned pointer");
    if (att_check("attention_bwd", B, n, heads, d, (size_t)2 * n * (d + 4) + 2 * n)) return 1;
    if (mrfa_attention_mfma_ok(d, n, qkv, ld, out, ldo, dout, lddo, dqkv, lddq))
        return mrfa_attention_bwd_mfma((hipStream_t)stream, qkv, ld, out, ldo, dout, lddo, lse, delta, B, n, heads, d, scale, dqkv, lddq);
    const dim3 grid(B * heads, cdiv(n, ATT_ROWS));
    {
        const size_t lds = (size_t)2 * n * (d + 4) * sizeof(float);
        int rc = d == 24 ? att_attr(attention_bwd_q_kernel<24>, lds) : d == 16 ? att_attr(attention_bwd_q_kernel<16>, lds)
                                                                                 : att_attr(attention_bwd_q_kernel<32>, lds);
        MRFA_CHECK_ARG(rc == 0, "attention_bwd: cannot reserve %zu bytes of LDS", lds);
        ATT_DISPATCH(d, attention_bwd_q_kernel, qkv, ld, out, ldo, dout, lddo, lse, delta, n, heads, scale, dqkv, lddq);
        MRFA_CHECK_LAUNCH("attention_bwd(q)");
    }
    {
        const size_t lds = ((size_t)2 * n * (d + 4) + 2 * n) * sizeof(float);
        int rc = d == 24 ? att_attr(attention_bwd_kv_kernel<24>, lds) : d == 16 ? att_attr(attention_bwd_kv_kernel<16>, lds)
                                                                                  : att_attr(attention_bwd_kv_kernel<32>, lds);
        MRFA_CHECK_ARG(rc == 0, "attention_bwd: cannot reserve %zu bytes of LDS", lds);
        ATT_DISPATCH(d, attention_bwd_kv_kernel, qkv, ld, dout, lddo, lse, delta, n, heads, scale, dqkv, lddq);
        MRFA_CHECK_LAUNCH("attention_bwd(kv)");
    }
    return 0;
}
