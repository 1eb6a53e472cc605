// BatchNorm2d (train + eval) on NHWC views, fused with ReLU, the DownBlock 2x2 average pool and the generator's
// occlusion blend.  HBM-bound: every kernel streams the activation once with 64 consecutive channels per wave
// (256-B coalesced rows) and reduces per-channel statistics with per-thread partial sums -> LDS -> one fp64 atomic per
// channel per workgroup.  Replaces aten::batch_norm / relu / avg_pool2d at modules/util.py:122,146-147,170,189-190,208
// and the blend at modules/generator.py:57, forward and backward.
#include "common.h"

namespace {

constexpr int CH = 64;        // channels per workgroup (one per lane)
constexpr int NW = 4;         // waves per workgroup, each walks its own rows

// ---------------------------------------------------------------------------------------------- statistics
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int ldx, long long rows, int C,
                                                      double* __restrict__ stats, int rows_per_block) {
    __shared__ float red[2][NW][CH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * CH + lane;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = min(rows, r0 + rows_per_block);
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        for (long long r = r0 + wave; r < r1; r += NW) {
            const float v = x[(size_t)r * ldx + c];
            s1 += v;
            s2 += v * v;
        }
    }
    red[0][wave][lane] = s1;
    red[1][wave][lane] = s2;
    __syncthreads();
    if (wave == 0 && c < C) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < NW; ++w) { a += red[0][w][lane]; b += red[1][w][lane]; }
        double* st = stats + (size_t)((blockIdx.y + blockIdx.x) % MRFA_STATS_SLOTS) * 2 * C;       // see MRFA_STATS_SLOTS (mrfa_hip.h)
        atomicAdd(st + c, a);
        atomicAdd(st + C + c, b);
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ stats, long long count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                   float momentum, float eps, int C, int train, int groups, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    chain_prio();
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    // statistic groups (train mode): one set of outputs and one momentum update of the running statistics per group, in group order --
    // what `groups` successive calls of the module do (reference model.py:185-186,234)
    float rm = rmean ? rmean[c] : 0.f, rv = rvar ? rvar[c] : 0.f;
    for (int g = 0; g < groups; ++g) {
        float mean, invstd;
        if (train) {
            const double* sg = stats + (size_t)g * MRFA_STATS_SLOTS * 2 * C;
            double t1 = 0.0, t2 = 0.0;
            for (int s = 0; s < MRFA_STATS_SLOTS; ++s) { t1 += sg[(size_t)s * 2 * C + c]; t2 += sg[(size_t)s * 2 * C + C + c]; }
            const double m = t1 / (double)count;
            double var = t2 / (double)count - m * m;
            if (var < 0.0) var = 0.0;
            mean = (float)m;
            invstd = (float)(1.0 / sqrt(var + (double)eps));
            const double unb = count > 1 ? var * (double)count / (double)(count - 1) : var;
            rm = (1.f - momentum) * rm + momentum * mean;
            rv = (1.f - momentum) * rv + momentum * (float)unb;
        } else {
            mean = rm;
            invstd = 1.0f / sqrtf(rv + eps);
        }
        const float sc = gamma[c] * invstd;
        scale[g * C + c] = sc;
        shift[g * C + c] = beta[c] - mean * sc;
        if (mean_out) mean_out[g * C + c] = mean;
        if (invstd_out) invstd_out[g * C + c] = invstd;
    }
    if (train && rmean) { rmean[c] = rm; rvar[c] = rv; }
}

// ---------------------------------------------------------------------------------------------- forward apply
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const mrfa_bnact_params p, long long total) {
    chain_prio();
    const int Ho = p.pool ? p.H / 2 : p.H, Wo = p.pool ? p.W / 2 : p.W;
    const long long gpix = p.groups > 1 ? (long long)(p.N / p.groups) * Ho * Wo : 0;      // output pixels per statistic group
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long opix = i / p.C;
        const int c = (int)(i - opix * p.C);
        const int gc = gpix ? (int)(opix / gpix) * p.C + c : c;
        const float sc = p.scale[gc], sh = p.shift[gc];
        float v;
        if (!p.pool) {
            v = p.x[(size_t)opix * p.ldx + c] * sc + sh;
            if (p.res) v += p.res[(size_t)opix * p.ldr + c];
            if (p.relu) v = fmaxf(v, 0.f);
        } else {
            const int ox = (int)(opix % Wo);
            const long long t = opix / Wo;
            const int oy = (int)(t % Ho);
            const long long n = t / Ho;
            const size_t base = ((size_t)n * p.H + 2 * oy) * p.W + 2 * ox;
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float u = p.x[(base + (q >> 1) * p.W + (q & 1)) * p.ldx + c] * sc + sh;
                if (p.relu) u = fmaxf(u, 0.f);
                a += u;
            }
            v = 0.25f * a;
        }
        if (p.blend_a) {
            const float o = p.occ[(size_t)opix * p.ldo];
            v = p.blend_a[(size_t)opix * p.lda + c] * o + v * (1.f - o);
        }
        p.y[(size_t)opix * p.ldy + c] = v;
    }
}

// float4 form of the plain / residual case (every BatchNorm of the keypoint encoders): one thread = 4 consecutive channels of a pixel
template <bool RES>
__global__ __launch_bounds__(256) void bn_act_fwd_vec_kernel(const mrfa_bnact_params p, long long total4, int c4) {
    chain_prio();
    const long long gpix = p.groups > 1 ? (long long)(p.N / p.groups) * p.H * p.W : 0;      // pixels per statistic group
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const long long opix = i / c4;
        const int c = (int)(i - opix * c4) * 4;
        const int gc = gpix ? (int)(opix / gpix) * p.C + c : c;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + gc), sh = *reinterpret_cast<const f32x4*>(p.shift + gc);
        f32x4 v = *reinterpret_cast<const f32x4*>(p.x + (size_t)opix * p.ldx + c) * sc + sh;
        if (RES) v += *reinterpret_cast<const f32x4*>(p.res + (size_t)opix * p.ldr + c);
        if (p.relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        *reinterpret_cast<f32x4*>(p.y + (size_t)opix * p.ldy + c) = v;
    }
}

// ---------------------------------------------------------------------------------------------- backward
// how many of the MRFA_STATS_SLOTS blocks of `red` a launch with `row_blocks` workgroups per channel chunk spreads its phase-1 atomics
// over (both phases run the same grid): summing 32 blocks costs phase 2 ~1 us, which only pays when >= 64 workgroups would otherwise
// queue on one address (measured: 32 ch @ 64^2 x 8: phase 1 10.7 -> 6.2 us; 128 ch @ 16^2 x 8 with 32 workgroups: nothing to gain)
__device__ __forceinline__ int red_slots(unsigned row_blocks) {
    return row_blocks >= 64 ? MRFA_STATS_SLOTS : (row_blocks >= 16 ? 8 : 1);
}

// Statistic groups (mrfa_bnbwd_params.groups): the grid's y axis holds one run of `row_blocks` workgroups per group, each run covering that group's
// rows [g cnt, (g + 1) cnt) in steps of rows_per_block.  Ungrouped: one run over all rows.
struct GroupSpan {
    int g;                  // this workgroup's group
    unsigned by, row_blocks; // its index inside the group's run, workgroups per run
    long long cnt, r0, r1;  // rows per group, this workgroup's rows
};
__device__ __forceinline__ GroupSpan group_span(const mrfa_bnbwd_params& p, long long rows, int rows_per_block) {
    GroupSpan s;
    const int G = p.groups > 1 ? p.groups : 1;
    s.row_blocks = gridDim.y / G;
    s.g = blockIdx.y / s.row_blocks;
    s.by = blockIdx.y - s.g * s.row_blocks;
    s.cnt = rows / G;
    s.r0 = s.g * s.cnt + (long long)s.by * rows_per_block;
    s.r1 = min((s.g + 1) * s.cnt, s.r0 + rows_per_block);
    return s;
}

// u = x*scale+shift ; a = relu(u) ; out = pool(a) or blend(A, a, occ).  PHASE 1: per-channel sum(du), sum(du*xhat),
// plus dA / docc of the blend.  PHASE 2: dx += gamma*invstd*(du - mean(du) - xhat*mean(du*xhat)) (train) or du*scale.
template <int PHASE>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const mrfa_bnbwd_params p, long long rows, int rows_per_block) {
    __shared__ float red[2][NW][CH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * CH + lane;
    const bool c_ok = c < p.C;
    const GroupSpan gs = group_span(p, rows, rows_per_block);
    const long long r0 = gs.r0, r1 = gs.r1;
    const int gc = gs.g * p.C + c;                           // this group's row of the [groups][C] per-channel arrays
    double* const red_g = p.red + (size_t)gs.g * MRFA_STATS_SLOTS * 2 * p.C;
    const int Wo = p.W / 2, Ho = p.H / 2;
    float sc = 0.f, sh = 0.f, mean = 0.f, invstd = 0.f, k1 = 0.f, k2 = 0.f, gi = 0.f;
    if (c_ok) {
        sc = p.scale[gc]; sh = p.shift[gc];
        if (p.mean) { mean = p.mean[gc]; invstd = p.invstd[gc]; }
        if (PHASE == 2 && p.train) {
            const double cnt = (double)gs.cnt * (double)(p.red_world > 1 ? p.red_world : 1);       // (SyncBatchNorm: `red` summed over the ranks)
            double t1 = 0.0, t2 = 0.0;
            for (int s = 0; s < MRFA_STATS_SLOTS; ++s) { t1 += red_g[(size_t)s * 2 * p.C + c]; t2 += red_g[(size_t)s * 2 * p.C + p.C + c]; }
            k1 = (float)(t1 / cnt);
            k2 = (float)(t2 / cnt);
            gi = p.gamma[c] * invstd;
        }
    }
    float s1 = 0.f, s2 = 0.f;
    for (long long r = r0 + wave; r < r1; r += NW) {
        // r indexes INPUT pixels (full resolution)
        long long opix = r;
        float gmul = 1.f;
        if (p.pool) {
            const int xx = (int)(r % p.W);
            const long long t = r / p.W;
            const int yy = (int)(t % p.H);
            const long long n = t / p.H;
            opix = ((long long)n * Ho + (yy >> 1)) * Wo + (xx >> 1);
            gmul = 0.25f;
        }
        float du = 0.f, xv = 0.f, docc_part = 0.f;
        if (c_ok) {
            xv = p.x[(size_t)r * p.ldx + c];
            float u = xv * sc + sh;
            if (p.res) u += p.res[(size_t)r * p.ldr + c];
            const float a = p.relu ? fmaxf(u, 0.f) : u;
            float da = p.dy[(size_t)opix * p.lddy + c] * gmul;
            if (p.blend_a) {
                const float o = p.occ[(size_t)opix * p.ldo];
                if (PHASE == 1) {
                    const float A = p.blend_a[(size_t)opix * p.lda + c];
                    if (p.dblend_a) p.dblend_a[(size_t)opix * p.ldda + c] += da * o;
                    docc_part = da * (A - a);
                }
                da *= (1.f - o);
            }
            du = (p.relu && u <= 0.f) ? 0.f : da;
            if (PHASE == 1 && p.dres) p.dres[(size_t)r * p.lddr + c] += du;
        }
        if (PHASE == 1) {
            if (p.blend_a && p.docc) {
                const float t = wave_sum(docc_part);
                if (lane == 0) atomicAdd(p.docc + (size_t)opix * p.lddo, t);
            }
            if (c_ok) {
                s1 += du;
                s2 += du * (xv - mean) * invstd;
            }
        } else if (c_ok) {
            float dx;
            if (p.train) dx = gi * (du - k1 - (xv - mean) * invstd * k2);
            else dx = du * sc;
            float* q = p.dx + (size_t)r * p.lddx + c;
            *q = p.dx_overwrite ? dx : *q + dx;
        }
    }
    if (PHASE == 1) {
        red[0][wave][lane] = s1;
        red[1][wave][lane] = s2;
        __syncthreads();
        if (wave == 0 && c_ok) {
            double a = 0.0, b = 0.0;
            for (int w = 0; w < NW; ++w) { a += red[0][w][lane]; b += red[1][w][lane]; }
            double* rd = red_g + (size_t)((gs.by + blockIdx.x) % red_slots(gs.row_blocks)) * 2 * p.C;       // see MRFA_STATS_SLOTS (mrfa_hip.h)
            atomicAdd(rd + c, a);
            atomicAdd(rd + p.C + c, b);
        }
    }
}


// float4 variant of the backward kernel: a lane owns 4 consecutive channels, 16 lanes cover the block's 64 channels of one
// row, so a wave streams 4 rows (4 x 256 B) per iteration and a workgroup 16; needs C % 4 == 0 and 16-byte aligned
// views.  Same arithmetic as the scalar kernel above, element by element.  Phase 2 also folds the parameter-gradient
// update (dgamma += sum du*xhat, dbeta += sum du) into the first row-block instead of a separate launch.
// MODE 0: plain, 1: residual, 2: 2x2 pool, 3: occlusion blend -- compile-time, so that the row loop of the common (plain / residual)
// case has no uniform branches between its loads (as run-time flags they kept hipcc from batching the x / dy / res loads of a row group)
template <int PHASE, int MODE>
__device__ __forceinline__ void bn_bwd_body(const mrfa_bnbwd_params& p, long long rows, int rows_per_block, float (&red)[2][16][CH],
                                            double (&redsum)[2][CH]) {
    constexpr bool POOL = MODE == 2, BLEND = MODE == 3, RES = MODE == 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane & 15, rsub = lane >> 4;
    const int slot = wave * 4 + rsub;                       // 16 row slots per workgroup
    const int c = blockIdx.x * CH + cg * 4;
    const bool c_ok = c < p.C;                              // C % 4 == 0: all four channels valid together
    const GroupSpan gs = group_span(p, rows, rows_per_block);
    const int gc = gs.g * p.C + c;                          // this group's row of the [groups][C] per-channel arrays
    double* const red_g = p.red + (size_t)gs.g * MRFA_STATS_SLOTS * 2 * p.C;
    if (PHASE == 2 && (p.train || p.dbeta || p.dgamma)) {
        // thread (channel, quarter) adds every 4th of the used slots; the 4 quarters of a channel are lanes 4k..4k+3 of one wave
        const int nslots = p.red_all ? MRFA_STATS_SLOTS : red_slots(gs.row_blocks);      // (red_all: the sums came from another kernel's slot choice)
        const int q = threadIdx.x & 3, col = (threadIdx.x >> 2) & 63;
        const int cc = blockIdx.x * CH + col;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            double t = 0.0;
            if (cc < p.C) {
                for (int s = q; s < nslots; s += 4) {
                    t += red_g[(size_t)s * 2 * p.C + which * p.C + cc];
                }
            }
            t += __shfl_xor(t, 1, 64);
            t += __shfl_xor(t, 2, 64);
            if (q == 0) redsum[which][col] = t;
        }
        __syncthreads();
    }
    const long long r0 = gs.r0, r1 = gs.r1;
    const int Wo = p.W / 2, Ho = p.H / 2;
    f32x4 sc = {0, 0, 0, 0}, sh = sc, mean = sc, invstd = sc, k1 = sc, k2 = sc, gi = sc;
    if (c_ok) {
        sc = *reinterpret_cast<const f32x4*>(p.scale + gc);
        sh = *reinterpret_cast<const f32x4*>(p.shift + gc);
        if (p.mean) {
            mean = *reinterpret_cast<const f32x4*>(p.mean + gc);
            invstd = *reinterpret_cast<const f32x4*>(p.invstd + gc);
        }
        if (PHASE == 2 && p.train) {
            const double cnt = (double)gs.cnt * (double)(p.red_world > 1 ? p.red_world : 1);       // (SyncBatchNorm: `red` summed over the ranks)
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                k1[k] = (float)(redsum[0][cg * 4 + k] / cnt);
                k2[k] = (float)(redsum[1][cg * 4 + k] / cnt);
                gi[k] = g[k] * invstd[k];
            }
        }
        if (PHASE == 2 && gs.by == 0 && slot == 0) {        // parameter gradients, once per channel (and statistic group: the groups' sums add up)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (p.dbeta) atomicAdd(p.dbeta + c + k, (float)redsum[0][cg * 4 + k]);         // atomic: see unpack_multi_kernel
                if (p.dgamma) atomicAdd(p.dgamma + c + k, (float)redsum[1][cg * 4 + k]);
            }
        }
    }
    f32x4 s1 = {0, 0, 0, 0}, s2 = s1;
    if constexpr (MODE <= 1) {
        // plain / residual (every BatchNorm of the keypoint encoders): two rows per trip, all loads of both rows issued before anything
        // is consumed, no branch around a load (idle lanes of a partial channel chunk read channel 0; rows past the end re-read the
        // last row).  With one row per trip inside `if (c_ok)` every trip was a serial L2 round trip.
        const int cl = c_ok ? c : 0;
        for (long long r = r0 + slot; r < r1; r += 32) {
            long long rr[2];
            bool live[2];
            f32x4 xv[2], da[2], rs[2], cur[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                live[u] = r + 16 * u < r1;
                rr[u] = live[u] ? r + 16 * u : r;
                xv[u] = *reinterpret_cast<const f32x4*>(p.x + (size_t)rr[u] * p.ldx + cl);
                da[u] = *reinterpret_cast<const f32x4*>(p.dy + (size_t)rr[u] * p.lddy + cl);
                rs[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (RES) rs[u] = *reinterpret_cast<const f32x4*>(p.res + (size_t)rr[u] * p.ldr + cl);
                cur[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (PHASE == 1 && RES && p.dres) cur[u] = *reinterpret_cast<const f32x4*>(p.dres + (size_t)rr[u] * p.lddr + cl);
                if (PHASE == 2 && !p.dx_overwrite) cur[u] = *reinterpret_cast<const f32x4*>(p.dx + (size_t)rr[u] * p.lddx + cl);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 du;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float uu = xv[u][k] * sc[k] + sh[k] + rs[u][k];
                    du[k] = (p.relu && uu <= 0.f) ? 0.f : da[u][k];
                }
                const bool st = c_ok && live[u];
                if (PHASE == 1) {
                    if (RES && p.dres && st) *reinterpret_cast<f32x4*>(p.dres + (size_t)rr[u] * p.lddr + c) = cur[u] + du;
                    if (st) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            s1[k] += du[k];
                            s2[k] += du[k] * (xv[u][k] - mean[k]) * invstd[k];
                        }
                    }
                } else if (st) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        cur[u][k] += p.train ? gi[k] * (du[k] - k1[k] - (xv[u][k] - mean[k]) * invstd[k] * k2[k]) : du[k] * sc[k];
                    *reinterpret_cast<f32x4*>(p.dx + (size_t)rr[u] * p.lddx + c) = cur[u];
                }
            }
        }
    } else
    for (long long r = r0 + slot; r < r1; r += 16) {
        long long opix = r;
        float gmul = 1.f;
        if (POOL) {
            const int xx = (int)(r % p.W);
            const long long t = r / p.W;
            const int yy = (int)(t % p.H);
            const long long n = t / p.H;
            opix = ((long long)n * Ho + (yy >> 1)) * Wo + (xx >> 1);
            gmul = 0.25f;
        }
        f32x4 du = {0, 0, 0, 0}, xv = du;
        float docc_part = 0.f;
        if (c_ok) {
            xv = *reinterpret_cast<const f32x4*>(p.x + (size_t)r * p.ldx + c);
            f32x4 da = *reinterpret_cast<const f32x4*>(p.dy + (size_t)opix * p.lddy + c);
            float o = 0.f;
            f32x4 A = {0, 0, 0, 0};
            if (BLEND) {
                o = p.occ[(size_t)opix * p.ldo];
                if (PHASE == 1) A = *reinterpret_cast<const f32x4*>(p.blend_a + (size_t)opix * p.lda + c);
            }
            f32x4 dA = {0, 0, 0, 0}, rs = {0, 0, 0, 0};
            if (RES) rs = *reinterpret_cast<const f32x4*>(p.res + (size_t)r * p.ldr + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float u = xv[k] * sc[k] + sh[k] + rs[k];
                const float a = p.relu ? fmaxf(u, 0.f) : u;
                float d = da[k] * gmul;
                if (BLEND) {
                    if (PHASE == 1) {
                        dA[k] = d * o;
                        docc_part += d * (A[k] - a);
                    }
                    d *= (1.f - o);
                }
                du[k] = (p.relu && u <= 0.f) ? 0.f : d;
            }
            if (PHASE == 1 && RES && p.dres) {
                f32x4* q = reinterpret_cast<f32x4*>(p.dres + (size_t)r * p.lddr + c);
                f32x4 cur = *q;
                cur += du;
                *q = cur;
            }
            if (PHASE == 1 && BLEND && p.dblend_a) {
                f32x4* q = reinterpret_cast<f32x4*>(p.dblend_a + (size_t)opix * p.ldda + c);
                f32x4 cur = *q;
                cur += dA;
                *q = cur;
            }
        }
        if (PHASE == 1) {
            if (BLEND && p.docc) {
                float t = docc_part;                         // sum over the 16 lanes (64 channels) of this row
#pragma unroll
                for (int o2 = 8; o2 > 0; o2 >>= 1) t += __shfl_xor(t, o2, 64);
                if (cg == 0) atomicAdd(p.docc + (size_t)opix * p.lddo, t);
            }
            if (c_ok) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    s1[k] += du[k];
                    s2[k] += du[k] * (xv[k] - mean[k]) * invstd[k];
                }
            }
        } else if (c_ok) {
            f32x4* q = reinterpret_cast<f32x4*>(p.dx + (size_t)r * p.lddx + c);
            f32x4 cur = {0.f, 0.f, 0.f, 0.f};
            if (!p.dx_overwrite) cur = *q;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dx = p.train ? gi[k] * (du[k] - k1[k] - (xv[k] - mean[k]) * invstd[k] * k2[k]) : du[k] * sc[k];
                cur[k] += dx;
            }
            *q = cur;
        }
    }
    if (PHASE == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[0][slot][cg * 4 + k] = s1[k];
            red[1][slot][cg * 4 + k] = s2[k];
        }
        __syncthreads();
        const int cc = blockIdx.x * CH + threadIdx.x;
        if (threadIdx.x < CH && cc < p.C) {
            double a = 0.0, b = 0.0;
            for (int w = 0; w < 16; ++w) { a += red[0][w][threadIdx.x]; b += red[1][w][threadIdx.x]; }
            double* rd = red_g + (size_t)((gs.by + blockIdx.x) % red_slots(gs.row_blocks)) * 2 * p.C;       // see MRFA_STATS_SLOTS (mrfa_hip.h)
            atomicAdd(rd + cc, a);
            atomicAdd(rd + p.C + cc, b);
        }
    }
}

template <int PHASE, int MODE>
__global__ __launch_bounds__(256) void bn_act_bwd_vec_kernel(const mrfa_bnbwd_params p, long long rows, int rows_per_block) {
    chain_prio();
    __shared__ float red[2][16][CH];
    __shared__ double redsum[2][CH];                         // phase 2: the MRFA_STATS_SLOTS partial sums of phase 1, added up once per workgroup
    bn_bwd_body<PHASE, MODE>(p, rows, rows_per_block, red, redsum);
}

__global__ void bn_param_grad_kernel(const double* __restrict__ red, float* __restrict__ dgamma, float* __restrict__ dbeta, int C, int nslots) {
    chain_prio();
    // red[C+c] = sum(du*xhat) is d(gamma); red[c] = sum(du) is d(beta); nslots = statistic groups x MRFA_STATS_SLOTS consecutive blocks
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double t1 = 0.0, t2 = 0.0;
    for (int s = 0; s < nslots; ++s) { t1 += red[(size_t)s * 2 * C + c]; t2 += red[(size_t)s * 2 * C + C + c]; }
    if (dbeta) atomicAdd(dbeta + c, (float)t1);
    if (dgamma) atomicAdd(dgamma + c, (float)t2);
}

extern "C" int mrfa_bn_param_grad_groups(void* stream, const double* red, int C, int groups, float* dgamma, float* dbeta) {
    MRFA_CHECK_ARG(red && C > 0 && groups >= 1 && (dgamma || dbeta), "bn_param_grad: bad args");
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, red, dgamma, dbeta, C, groups * MRFA_STATS_SLOTS);
    MRFA_CHECK_LAUNCH("bn_param_grad");
    return 0;
}

extern "C" int mrfa_bn_param_grad(void* stream, const double* red, int C, float* dgamma, float* dbeta) {
    return mrfa_bn_param_grad_groups(stream, red, C, 1, dgamma, dbeta);
}

static int pick_rows_per_block(long long rows, int chunks, int C) {
    // Every workgroup ends with one fp64 atomic per channel into the same 2C addresses, and same-address atomics serialise in
    // L2: 512 workgroups on a 4 MB tensor spent 9 of 15 us there (tools/bn_micro.py).  Total workgroups ~ one per 64 KB of
    // tensor, between 256 (enough to stream a few MB) and 1024 (134 MB: 55 us, vs 76 us with 2048); at least 64 rows each.
    long long target = rows * C * 4 / 65536;
    if (target < 256) target = 256;
    if (target > 1024) target = 1024;
    long long want = (target + chunks - 1) / chunks;
    long long rpb = (rows + want - 1) / want;
    if (rpb < 64) rpb = 64;
    if (rpb > 4096) rpb = 4096;
    return (int)rpb;
}

}  // namespace

extern "C" int mrfa_bn_stats(void* stream, const float* x, int ldx, long long rows, int C, double* stats) {
    MRFA_CHECK_ARG(x && stats && rows > 0 && C > 0, "bn_stats: bad args");
    const int chunks = cdiv(C, CH);
    const int rpb = pick_rows_per_block(rows, chunks, C);
    dim3 grid(chunks, cdiv(rows, rpb));
    hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, rows, C, stats, rpb);
    MRFA_CHECK_LAUNCH("bn_stats");
    return 0;
}

extern "C" int mrfa_bn_finalize(void* stream, const double* stats, long long count, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, int C, int train,
                                float* scale, float* shift, float* mean_out, float* invstd_out) {
    MRFA_CHECK_ARG(gamma && beta && scale && shift && C > 0, "bn_finalize: bad args");
    MRFA_CHECK_ARG(train ? stats != nullptr : (running_mean && running_var), "bn_finalize: missing statistics");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, stats, count, gamma, beta,
                       running_mean, running_var, momentum, eps, C, train, 1, scale, shift, mean_out, invstd_out);
    MRFA_CHECK_LAUNCH("bn_finalize");
    return 0;
}

extern "C" int mrfa_bn_finalize_groups(void* stream, const double* stats, long long count, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var, float momentum, float eps, int C, int groups,
                                       float* scale, float* shift, float* mean_out, float* invstd_out) {
    MRFA_CHECK_ARG(stats && gamma && beta && scale && shift && C > 0 && groups >= 1 && count > 0, "bn_finalize_groups: bad args");
    MRFA_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn_finalize_groups: running_mean / running_var come together");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, stats, count, gamma, beta,
                       running_mean, running_var, momentum, eps, C, 1, groups, scale, shift, mean_out, invstd_out);
    MRFA_CHECK_LAUNCH("bn_finalize_groups");
    return 0;
}

extern "C" int mrfa_bn_act_fwd(void* stream, const mrfa_bnact_params* pp) {
    const mrfa_bnact_params& p = *pp;
    MRFA_CHECK_ARG(p.x && p.y && p.scale && p.shift, "bn_act_fwd: null pointer");
    MRFA_CHECK_ARG(!p.pool || ((p.H % 2) == 0 && (p.W % 2) == 0), "bn_act_fwd: pool needs even H,W");
    MRFA_CHECK_ARG(!(p.pool && p.blend_a), "bn_act_fwd: pool and blend are exclusive");
    MRFA_CHECK_ARG(!(p.res && (p.pool || p.blend_a)), "bn_act_fwd: residual excludes pool and blend");
    MRFA_CHECK_ARG(p.groups <= 1 || (p.N % p.groups) == 0, "bn_act_fwd: %d statistic groups do not divide N = %d", p.groups, p.N);
    const long long opix = (long long)p.N * (p.pool ? p.H / 2 : p.H) * (p.pool ? p.W / 2 : p.W);
    const long long total = opix * p.C;
    if (!p.pool && !p.blend_a && (p.C % 4) == 0 && (p.ldx % 4) == 0 && (p.ldy % 4) == 0 && aligned16(p.x) && aligned16(p.y) && aligned16(p.scale) &&
        aligned16(p.shift) && (!p.res || ((p.ldr % 4) == 0 && aligned16(p.res)))) {
        const long long total4 = total / 4;
        if (p.res) hipLaunchKernelGGL((bn_act_fwd_vec_kernel<true>), dim3(stream_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, p, total4, p.C / 4);
        else hipLaunchKernelGGL((bn_act_fwd_vec_kernel<false>), dim3(stream_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, p, total4, p.C / 4);
        MRFA_CHECK_LAUNCH("bn_act_fwd(vec)");
        return 0;
    }
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, p, total);
    MRFA_CHECK_LAUNCH("bn_act_fwd");
    return 0;
}

extern "C" int mrfa_bn_act_bwd(void* stream, const mrfa_bnbwd_params* pp) {
    const mrfa_bnbwd_params& p = *pp;
    MRFA_CHECK_ARG(p.x && p.dy && p.scale && p.shift && p.red, "bn_act_bwd: null pointer");
    MRFA_CHECK_ARG(!p.train || (p.mean && p.invstd && p.gamma), "bn_act_bwd: train mode needs mean/invstd/gamma");
    MRFA_CHECK_ARG(!(p.res && (p.pool || p.blend_a)) && !(p.dres && !p.res), "bn_act_bwd: residual excludes pool and blend");
    MRFA_CHECK_ARG(!(p.pool && p.blend_a), "bn_act_bwd: pool and blend are exclusive (one compile-time MODE per launch)");
    const int G = p.groups > 1 ? p.groups : 1;
    MRFA_CHECK_ARG((p.N % G) == 0, "bn_act_bwd: %d statistic groups do not divide N = %d", G, p.N);
    const long long rows = (long long)p.N * p.H * p.W;
    const int chunks = cdiv(p.C, CH);
    // (statistic groups: the row blocks are cut per group, G runs of them along grid.y -- group_span())
    const int rpb = pick_rows_per_block(rows, chunks, p.C);
    dim3 grid(chunks, G * cdiv(rows / G, rpb));
    MRFA_CHECK_ARG(p.phase == 1 || p.phase == 2, "bn_act_bwd: phase %d", p.phase);
    const bool vec = (p.C % 4 == 0) && (p.ldx % 4 == 0) && (p.lddy % 4 == 0) && aligned16(p.x) && aligned16(p.dy) && aligned16(p.scale) &&
                     aligned16(p.shift) && (!p.mean || (aligned16(p.mean) && aligned16(p.invstd))) && (!p.gamma || aligned16(p.gamma)) &&
                     (p.phase == 1 || ((p.lddx % 4 == 0) && aligned16(p.dx))) &&
                     (!p.res || ((p.ldr % 4 == 0) && aligned16(p.res) && (!p.dres || ((p.lddr % 4 == 0) && aligned16(p.dres))))) &&
                     (!p.blend_a || ((p.lda % 4 == 0) && aligned16(p.blend_a) && (!p.dblend_a || ((p.ldda % 4 == 0) && aligned16(p.dblend_a)))));
    if (vec) {
        MRFA_CHECK_ARG(p.phase == 1 || p.dx != nullptr, "bn_act_bwd: phase 2 needs dx");
        const int mode = p.pool ? 2 : (p.blend_a ? 3 : (p.res ? 1 : 0));
#define BNB(PH, MD) hipLaunchKernelGGL((bn_act_bwd_vec_kernel<PH, MD>), grid, dim3(256), 0, (hipStream_t)stream, p, rows, rpb)
        if (p.phase == 1) { if (mode == 0) BNB(1, 0); else if (mode == 1) BNB(1, 1); else if (mode == 2) BNB(1, 2); else BNB(1, 3); }
        else { if (mode == 0) BNB(2, 0); else if (mode == 1) BNB(2, 1); else if (mode == 2) BNB(2, 2); else BNB(2, 3); }
#undef BNB
        MRFA_CHECK_LAUNCH("bn_act_bwd(vec)");
        return 0;
    }
    if (p.phase == 1) {
        hipLaunchKernelGGL((bn_act_bwd_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, p, rows, rpb);
    } else {
        MRFA_CHECK_ARG(p.dx != nullptr, "bn_act_bwd: phase 2 needs dx");
        hipLaunchKernelGGL((bn_act_bwd_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, p, rows, rpb);
        if (p.dgamma || p.dbeta) {
            hipLaunchKernelGGL(bn_param_grad_kernel, dim3(cdiv(p.C, 256)), dim3(256), 0, (hipStream_t)stream, p.red, p.dgamma, p.dbeta,
                               p.C, G * MRFA_STATS_SLOTS);
        }
    }
    MRFA_CHECK_LAUNCH("bn_act_bwd");
    return 0;
}
