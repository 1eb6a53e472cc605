// Weight-gradient convolution / batched TN GEMM on v_mfma_f32_32x32x2_f32.
//
//   dW[tap][co][ci] += alpha * sum_p dY[p][co] * X'[pix(p) + tap - pad][ci]        X' = relu(scale*ups(x)+shift)
//
// GEMM view: M = Cout, N = Cin (or taps*Cin in flat mode), K = N*Hout*Wout pixels.  Both operands are pixel-major
// NHWC, i.e. K is the STRIDED axis of both ("TN").  Each thread loads a 4(k) x 4(m) block with four coalesced float4
// row loads, transposes it in registers and stores four float4 {k..k+3} vectors, so the LDS image is
// [k/4][m%4][m/4 (+4 pad)][k%4]: stores are lane-contiguous, and an MFMA lane fetches FOUR consecutive k of its row
// with one conflict-free ds_read_b128 (16-B slot = (9*(m%4) + m/4) mod 16 distinct per lane group) -- the same
// operand cadence as the forward kernel: 16 ds_read_b128 feed 64 MFMAs per k-tile per wave, so the matrix pipe
// (64 cycles per MFMA per SIMD) is the bound.  The pixel reduction is split across workgroups (gridDim.z) and combined
// with fp32 atomics into a zero-initialised [tap][Cout][Cin] buffer; the bias gradient sum_p dY[p][co] falls out of
// the A-operand loads.
//
// Replaces the weight/bias-gradient half of aten::convolution_backward for every F.conv2d on the hot path
// (call sites listed in include/mrfa_hip.h) and d(k_s) of the correlation einsum (modules/raft.py:185).
#include "common.h"
#include <stdlib.h>
#include <algorithm>

namespace {

constexpr int WBK = 32;      // pixels per k-tile
constexpr int KG = WBK / 4;  // k-groups (planes) per tile

template <int BM, int BN, int WAVES_M, int WAVES_N, bool FLAT, bool ROWAL>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void wgrad_mfma_kernel(const mrfa_wgrad_params p, const long long M, const long long k_per_split,
                                                        const int tiles_n, const int nsplit, const int dy_scalar,
                                                        const int inner, const int total_splits, const int taps,
                                                        const long long partial_stride) {
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;   // per-wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int NT = WAVES_M * WAVES_N * 64;
    static_assert((NT == 256 || NT == 512) && TM >= 1 && TN >= 1, "tile");
    constexpr int SA = BM / 4 + 4, SB = BN / 4 + 4;         // float4 stride between the four (m%4) groups of a plane
    constexpr int PLA = 4 * SA, PLB = 4 * SB;               // float4 per plane
    constexpr int IA = (BM / 4) * KG, IB = (BN / 4) * KG;   // (column-group, k-group) work items per tile
    __shared__ f32x4 smem[2 * KG * (PLA + PLB)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // XCD-aware order: workgroup b runs on XCD b % 8.  All `inner` = tiles*taps workgroups that stream the SAME pixel
    // range (one split) are placed on one XCD, back to back, so dY and the tap-shifted X rows are fetched into that
    // XCD's L2 once and re-read by the other taps / channel tiles instead of 9-18x from HBM.
    int gsplit, t_in;
    if (total_splits >= 8) {
        const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
        gsplit = (jj / inner) * 8 + xcd;                    // global split index = batch * nsplit + split
        if (gsplit >= total_splits) return;
        t_in = jj - (jj / inner) * inner;
    } else {                                                // too few splits to give every XCD its own: plain order
        gsplit = blockIdx.x / inner;
        if (gsplit >= total_splits) return;
        t_in = blockIdx.x - gsplit * inner;
    }
    const int tile = t_in / taps;
    const int tap = FLAT ? 0 : (t_in - tile * taps);
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const int co0 = tile_m * BM, ci0 = tile_n * BN;
    const int NTOT = FLAT ? p.kflat : p.Cin;                // extent of the GEMM N axis
    const int bz = gsplit / nsplit, split = gsplit - bz * nsplit;
    const int r = tap / p.S, s = tap - r * p.S;

    const float* __restrict__ x = p.x + (size_t)bz * p.x_bs;
    const float* __restrict__ dy = p.dy + (size_t)bz * p.dy_bs;
    float* __restrict__ dw = p.dw + (size_t)bz * p.dw_bs;

    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;
    const int HWo = p.Hout * p.Wout;
    const int kb = (int)((long long)split * k_per_split);
    const int ke = (int)min(M, (long long)kb + k_per_split);

    // loader work item of this thread: column group (4 columns) x k-group (4 pixels)
    // 512-thread variant: threads 0-255 stage A, threads 256-511 stage B (4 loads each instead of 8)
    const int btid = NT == 512 ? tid - 256 : tid;
    const bool a_item = tid < IA, b_item = btid >= 0 && btid < IB;
    const int a_cg = tid % (BM / 4), a_kg = tid / (BM / 4);
    const int b_cg = (btid & 0x3ff) % (BN / 4), b_kg = (btid & 0x3ff) / (BN / 4);
    const int a_col = a_cg * 4, b_col = b_cg * 4;
    const bool do_bias = (p.dbias != nullptr) && tap == 0 && tile_n == 0;

    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    bool cmask[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cmask[q] = b_item && (ci0 + b_col + q) < NTOT;
    int fdy[4] = {0, 0, 0, 0}, fdx[4] = {0, 0, 0, 0}, fci[4] = {0, 0, 0, 0};
    if constexpr (FLAT) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (cmask[q]) {
                const int e = p.ktab[ci0 + b_col + q];
                fdy[q] = (e & 255) - 128; fdx[q] = ((e >> 8) & 255) - 128; fci[q] = e >> 16;
            }
    }
    if (p.in_scale) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (cmask[q]) {
                const int c = FLAT ? fci[q] : (ci0 + b_col + q);
                sc[q] = p.in_scale[c]; sh[q] = p.in_shift[c];
            }
    }
    bool amask[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) amask[q] = a_item && (co0 + a_col + q) < p.Cout;
    const bool a_any = amask[0], b_any = cmask[0];

    f32x4 ra[4], rb[4];                 // [row r of the k-group] x 4 columns
    f32x4 bias_acc = {0.f, 0.f, 0.f, 0.f};

    // pixel coordinates of this thread's first B row, advanced incrementally (no divisions in the loop)
    int b_n, b_oy, b_ox;
    {
        const int pp = kb + 4 * b_kg;
        b_n = pp / HWo;
        const int rem = pp - b_n * HWo;
        b_oy = rem / p.Wout;
        b_ox = rem - b_oy * p.Wout;
    }

    // ROWAL fast path (Wout % 32 == 0, every level >= 32^2): a k-tile is 32 consecutive pixels of ONE image row, so the
    // image / row part of every address and the vertical bounds test are wave-uniform scalars; only the horizontal
    // bound is per lane.  (n, oy, ox0) of the current tile are advanced with scalar adds.
    int t_n = 0, t_oy = 0, t_ox0 = 0;
    if constexpr (ROWAL) {
        t_n = kb / HWo;
        const int rem = kb - t_n * HWo;
        t_oy = rem / p.Wout;
        t_ox0 = rem - t_oy * p.Wout;
    }

    auto load_tiles = [&](int k0) {
        if constexpr (ROWAL) {
            if (a_any) {
                const float* ap = dy + (size_t)(k0 + 4 * a_kg) * p.ldy + co0 + a_col;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    f32x4 v;
                    if (!dy_scalar) {
                        v = *reinterpret_cast<const f32x4*>(ap + (size_t)rr * p.ldy);
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (!amask[q]) v[q] = 0.f;
                    } else {
                        v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (amask[q]) v[q] = ap[(size_t)rr * p.ldy + q];
                    }
                    ra[rr] = v;
                }
            } else {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) ra[rr] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const int ox = t_ox0 + 4 * b_kg;
            if constexpr (!FLAT) {
                const int iy = t_oy + r - p.pad;
                const bool rowok = b_any && (unsigned)iy < (unsigned)Hv;
                const int iyc = rowok ? (iy >> p.ups) : 0;
                const int colc = b_any ? (ci0 + b_col) : 0;
                const float* bp = x + ((size_t)t_n * p.Hin + iyc) * p.Win * p.ldx + colc;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int ix = ox + rr + s - p.pad;
                    const bool ok = rowok && (unsigned)ix < (unsigned)Wv;
                    // branch-free: out-of-bounds taps read column 0 of the row and are zeroed afterwards
                    f32x4 v = *reinterpret_cast<const f32x4*>(bp + (size_t)(ok ? (ix >> p.ups) : 0) * p.ldx);
                    if (p.in_scale) {
                        v = v * sc + sh;
                        if (p.in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (!cmask[q] || !ok) v[q] = 0.f;
                    rb[rr] = v;
                }
            } else {
                const float* bp = x + (size_t)t_n * p.Hin * p.Win * p.ldx;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int iy = t_oy + fdy[q], ix = ox + rr + fdx[q];
                        if (cmask[q] && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv) {
                            float t = bp[((size_t)(iy >> p.ups) * p.Win + (ix >> p.ups)) * p.ldx + fci[q]];
                            if (p.in_scale) { t = t * sc[q] + sh[q]; if (p.in_relu) t = fmaxf(t, 0.f); }
                            v[q] = t;
                        }
                    }
                    rb[rr] = v;
                }
            }
            t_ox0 += WBK;
            if (t_ox0 == p.Wout) { t_ox0 = 0; if (++t_oy == p.Hout) { t_oy = 0; ++t_n; } }
            return;
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int pp = k0 + 4 * a_kg + rr;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pp < ke && a_any) {
                if (!dy_scalar) {
                    v = *reinterpret_cast<const f32x4*>(dy + (size_t)pp * p.ldy + co0 + a_col);
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (!amask[q]) v[q] = 0.f;
                } else {       // unaligned dY view (e.g. a 1-channel slice at an odd channel offset): element loads
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (amask[q]) v[q] = dy[(size_t)pp * p.ldy + co0 + a_col + q];
                }
            }
            ra[rr] = v;
        }
        int n_img = b_n, oy = b_oy, ox = b_ox;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int pp = k0 + 4 * b_kg + rr;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pp < ke && b_any) {
                if constexpr (!FLAT) {
                    const int iy = oy + r - p.pad, ix = ox + s - p.pad;
                    if ((unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv) {
                        const size_t pix = (size_t)n_img * p.Hin * p.Win + (size_t)(iy >> p.ups) * p.Win + (ix >> p.ups);
                        v = *reinterpret_cast<const f32x4*>(x + pix * p.ldx + ci0 + b_col);
                        if (p.in_scale) {
                            v = v * sc + sh;
                            if (p.in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (!cmask[q]) v[q] = 0.f;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int iy = oy + fdy[q], ix = ox + fdx[q];
                        if (cmask[q] && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv) {
                            const size_t pix = (size_t)n_img * p.Hin * p.Win + (size_t)(iy >> p.ups) * p.Win + (ix >> p.ups);
                            float t = x[pix * p.ldx + fci[q]];
                            if (p.in_scale) { t = t * sc[q] + sh[q]; if (p.in_relu) t = fmaxf(t, 0.f); }
                            v[q] = t;
                        }
                    }
                }
            }
            rb[rr] = v;
            if (++ox == p.Wout) { ox = 0; if (++oy == p.Hout) { oy = 0; ++n_img; } }      // next pixel of the k-group
        }
        // advance the first row by one k-tile (WBK pixels)
        b_ox += WBK;
        while (b_ox >= p.Wout) {
            b_ox -= p.Wout;
            if (++b_oy == p.Hout) { b_oy = 0; ++b_n; }
        }
    };
    auto store_tiles = [&](int buf) {
        f32x4* As = smem + buf * KG * (PLA + PLB);
        f32x4* Bs = As + KG * PLA;
        if (a_item) {
#pragma unroll
            for (int e = 0; e < 4; ++e) As[a_kg * PLA + e * SA + a_cg] = f32x4{ra[0][e], ra[1][e], ra[2][e], ra[3][e]};
            if (do_bias) bias_acc += ra[0] + ra[1] + ra[2] + ra[3];
        }
        if (b_item) {
#pragma unroll
            for (int e = 0; e < 4; ++e) Bs[b_kg * PLB + e * SB + b_cg] = f32x4{rb[0][e], rb[1][e], rb[2][e], rb[3][e]};
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    if (kb < ke) {
        load_tiles(kb);
        store_tiles(0);
    }
    __syncthreads();
    const int fi = lane & 31, fh = lane >> 5;
    const int a_lane = (fi & 3) * SA + (fi >> 2) + ((wm * WTM) >> 2);
    const int b_lane = (fi & 3) * SB + (fi >> 2) + ((wn * WTN) >> 2);
    int cur = 0;
    for (int k0 = kb; k0 < ke; k0 += WBK) {
        const bool more = (k0 + WBK) < ke;
        if (more) load_tiles(k0 + WBK);
        const f32x4* As = smem + cur * KG * (PLA + PLB) + a_lane;
        const f32x4* Bs = smem + cur * KG * (PLA + PLB) + KG * PLA + b_lane;
#pragma unroll
        for (int q = 0; q < KG / 2; ++q) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(2 * q + fh) * PLA + i * 8];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[(2 * q + fh) * PLB + j * 8];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        if (more) store_tiles(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nn = ci0 + wn * WTN + j * 32 + (lane & 31);
        if (nn >= NTOT) continue;
        const int otap = FLAT ? nn / p.Cin : tap;
        const int ci = FLAT ? nn - otap * p.Cin : nn;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co = co0 + wm * WTM + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * fh;
                if (co < p.Cout) {
                    const size_t idx = ((size_t)otap * p.Cout + co) * p.Cin + ci;
                    if (partial_stride) p.ws[(size_t)split * partial_stride + idx] = acc[i][j][q] * p.alpha;   // two-stage reduction
                    else atomicAdd(dw + idx, acc[i][j][q] * p.alpha);
                }
            }
        }
    }
    // bias gradient: reduce the 8 k-group partials of every channel through LDS first, so a workgroup issues ONE atomic per
    // channel (per-thread atomics put 2048 x nsplit adds on each of the <= 128 addresses: ~0.6 ms on the 1x1 layers)
    if (p.dbias != nullptr && tap == 0 && tile_n == 0) {        // workgroup-uniform
        float* red = reinterpret_cast<float*>(smem);
        __syncthreads();
        if (a_item) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red[a_kg * BM + a_col + q] = bias_acc[q];
        }
        __syncthreads();
        if (tid < BM && co0 + tid < p.Cout) {
            float sacc = 0.f;
#pragma unroll
            for (int g = 0; g < KG; ++g) sacc += red[g * BM + tid];
            atomicAdd(p.dbias + co0 + tid, sacc);
        }
    }
}


// second stage of the two-stage split reduction: dw[i] += sum_s ws[s][i].  One thread per (i, group of 16 splits):
// 16 independent coalesced loads, then one atomic per group (<= nsplit/16-way contention instead of nsplit-way).
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long long n, int nsplit) {
    const int groups = (nsplit + 15) / 16;
    const long long total = n * groups;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(t / n);
        const long long i = t - (long long)g * n;
        const int k0 = g * 16, k1 = min(nsplit, k0 + 16);
        float s = 0.f;
#pragma unroll 16
        for (int k = k0; k < k1; ++k) s += ws[(size_t)k * n + i];
        atomicAdd(dw + i, s);
    }
}

}  // namespace

extern "C" int mrfa_conv2d_wgrad_lean_supported(const mrfa_wgrad_params* p) { return p && mrfa_wgrad_lean_eligible(*p) ? 1 : 0; }

extern "C" int mrfa_conv2d_wgrad_groups_supported(const mrfa_wgrad_params* p) {
    if (!p || p->groups <= 1 || !p->in_scale) return p ? 1 : 0;
    return mrfa_wgrad_lean_eligible(*p) ? 1 : 0;
}

extern "C" int mrfa_conv2d_wgrad_nhwc(void* stream, const mrfa_wgrad_params* pp) {
    const mrfa_wgrad_params& p = *pp;
    hipStream_t st = (hipStream_t)stream;
    MRFA_CHECK_ARG(p.x && p.dy && p.dw, "wgrad: null pointer");
    const int dy_scalar = ((p.ldy % 4) == 0 && aligned16(p.dy) && (p.dy_bs % 4) == 0) ? 0 : 1;
    const bool flat = p.kflat > 0;
    if (!flat) MRFA_CHECK_ARG((p.ldx % 4) == 0 && aligned16(p.x), "wgrad: x must be a 16-B aligned view with ld %% 4 == 0");
    else MRFA_CHECK_ARG(p.ktab != nullptr, "wgrad: flat mode needs ktab");
    const long long M = (long long)p.N * p.Hout * p.Wout;
    MRFA_CHECK_ARG(M < (1ll << 31) - 64, "wgrad: too many pixels");
    const int nb = p.nbatch > 1 ? p.nbatch : 1;
    // the keypoint encoder's <= 128-channel 3x3 layers WITH a prologue (a residual block's second convolution reading the raw output of its first): the
    // all-taps kernel of wgrad_lean.hip as a one-problem launch (without a prologue a lone problem stays on wgrad_small.hip: its 2 000 waves fill the chip)
    if (p.in_scale && mrfa_wgrad_lean_eligible(p)) {
        unsigned char taken = 0;
        return mrfa_wgrad_lean_multi(st, &p, 1, &taken);
    }
    MRFA_CHECK_ARG(p.groups <= 1 || !p.in_scale, "wgrad: groups = %d with a prologue is only implemented where mrfa_conv2d_wgrad_groups_supported() says so", p.groups);
    // small problems (the MTIA prior's layers): one wave per 32 x 32 weight block, no LDS staging, in-workgroup reduction (wgrad_small.hip)
    static const bool small_on = [] { const char* e = getenv("MRFA_CONV_SMALL"); return !(e && e[0] == '0'); }();
    if (small_on && mrfa_tuning_conv_small() && mrfa_wgrad_small_eligible(p, M)) return mrfa_wgrad_small_launch(st, p, M);
    MRFA_CHECK_ARG(p.stride <= 1, "wgrad: stride = %d is only implemented by the small-problem kernel: ask mrfa_conv2d_wgrad_stride_supported() first", p.stride);
    if (!flat && mrfa_wgrad_halo_eligible(p)) return mrfa_wgrad_halo_launch(st, p);      // 3x3 stride-1 layers: all nine taps per staging (wgrad_halo.hip)
    int taps = flat ? 1 : p.R * p.S;
    const int NTOT = flat ? p.kflat : p.Cin;
    // tile selection: (BM over Cout) x (BN over Cin or taps*Cin)
    auto pick = [](int n) {                 // padded size x small-tile penalty (loads ~ BM+BN, MFMAs ~ BM*BN)
        const int cands[4] = {128, 96, 64, 32};
        const double pen[4] = {1.0, 1.12, 1.35, 1.8};
        int best_t = 128;
        double best = 1e18;
        for (int i = 0; i < 4; ++i) {
            const double cost = (double)cdiv(n, cands[i]) * cands[i] * pen[i];
            if (cost < best) { best = cost; best_t = cands[i]; }
        }
        return best_t;
    };
    int BM = pick(p.Cout), BN = pick(NTOT);
    if (BM < 128 && BN < 128) {             // supported: one edge 128, or 64x64
        if (BM == 64 && BN == 64) {}
        else if (BM <= BN) BN = 128;
        else BM = 128;
    }
    if (flat && (BM == 96 || BN == 96)) { BM = BM == 96 ? 128 : BM; BN = BN == 96 ? 128 : BN; }
    // split-operand mode: the bf16x6 kernel is a 128 x 128 tile and ~1.6x faster than the fp32-MFMA tiles: prefer it over the
    // 64 / 96-wide tiles even at the cost of padding (eligibility: chunked K, Wout % 32 == 0, aligned dY)
    const bool split_mode = mrfa_get_mfma_mode() >= 1 && !flat && (p.Wout % 8) == 0 && (M % WBK) == 0 && !dy_scalar && p.Cout >= 32 && NTOT >= 32;
    if (split_mode) { BM = p.Cout <= 64 ? 64 : 128; BN = NTOT <= 64 ? 64 : 128; if (BM == 64 && BN == 64) BN = 128; }
    int tiles_m = cdiv(p.Cout, BM), tiles_n = cdiv(NTOT, BN);
    int taps_arg = taps;
    // split-operand kernel, several taps, Cin not a multiple of the tile width (64, 96, 160, 192 ...): tile the flattened [taps][Cin]
    // axis with 128-wide tiles (wgrad_split.hip "chunk-flat") when that removes >= 15 % of the padded tile work
    constexpr bool cflat_on = true;
    if (cflat_on && split_mode && taps > 1 && (p.Cin % 4) == 0) {
        // (a 64-wide tile does ~1.4x the work per column of a 128-wide one: measured 110 vs 190 TF/s)
        const double per_tap = (double)taps * tiles_n * BN * (BN == 64 ? 1.4 : 1.0), flat_w = (double)cdiv(taps * p.Cin, 128) * 128;
        if (flat_w <= 0.85 * per_tap) {
            BN = 128;
            tiles_n = cdiv(taps * p.Cin, 128);
            taps = 1;                       // (the cost model below counts tiles_m * tiles_n * taps workgroups per split)
            taps_arg = -1;
        }
    }
    const long long base = (long long)tiles_m * tiles_n * taps * nb;
    int nsplit = p.ksplit;
    if (nsplit <= 0) {
        // Choose the number of pixel-range splits by a small cost model: 2 workgroups fit a CU (LDS) => 64 run
        // concurrently per XCD, 512 per chip; with the XCD-aware order every XCD executes ceil(S/8) splits x `inner`
        // workgroups.  cost = rounds x k-tiles per workgroup (+ a term for the atomic epilogue of each workgroup).
        const long long max_split = std::max<long long>(1, M / (8 * WBK));     // >= 8 k-tiles per split
        double best = 1e30;
        nsplit = 1;
        const long long inner_ = (long long)tiles_m * tiles_n * taps;
        for (long long ns = 1; ns <= std::min<long long>(max_split, 1024); ++ns) {
            const long long S = ns * nb;
            const long long ktiles = ((M + ns - 1) / ns + WBK - 1) / WBK;
            double rounds;
            if (S >= 8) rounds = (double)(((S + 7) / 8 * inner_ + 63) / 64);
            else rounds = (double)((S * inner_ + 511) / 512);
            const double cost = rounds * (double)(ktiles + 6);
            if (cost < best * 0.999) { best = cost; nsplit = (int)ns; }
        }
    }
    long long kps = (M + nsplit - 1) / nsplit;
    kps = (kps + WBK - 1) / WBK * WBK;
    nsplit = (int)((M + kps - 1) / kps);
    const int inner = tiles_m * tiles_n * taps;
    const int total_splits = nsplit * nb;
    // Many splits of few tiles (1x1 and few-channel layers) would hammer the same few thousand addresses with hundreds of
    // atomics each: write per-split partial tiles with plain stores instead and reduce them in a second kernel.
    const long long wn = (long long)(flat ? 1 : taps) * 0 + (long long)p.R * p.S * p.Cout * p.Cin;
    const long long partial_stride = (nb == 1 && nsplit >= 24 && p.ws && wn * nsplit * 4 <= p.ws_bytes) ? wn : 0;
    dim3 grid((unsigned)(cdiv(total_splits, 8) * 8 * inner));
    const bool rowal = (p.Wout % WBK) == 0;
#define WLAUNCH(bm, bn, wm, wn, fl, ra)                                                                                     \
    hipLaunchKernelGGL((wgrad_mfma_kernel<bm, bn, wm, wn, fl, ra>), grid, dim3((wm) * (wn) * 64), 0, st, p, M, kps, tiles_n, nsplit, dy_scalar, \
                       inner, total_splits, taps, partial_stride)
#define WCFG(bm, bn, wm, wn)                                              \
    if (BM == bm && BN == bn) {                                           \
        if (flat && rowal) WLAUNCH(bm, bn, wm, wn, true, true);           \
        else if (flat) WLAUNCH(bm, bn, wm, wn, true, false);              \
        else if (rowal) WLAUNCH(bm, bn, wm, wn, false, true);             \
        else WLAUNCH(bm, bn, wm, wn, false, false);                       \
    }
    if (split_mode) {
        const int rc = mrfa_wgrad_split_launch(st, p, grid, M, kps, tiles_n, nsplit, inner, total_splits, taps_arg, partial_stride, BM, BN);
        if (rc) return rc;
    }
    else if (BM == 128 && BN == 128 && !flat && rowal && !p.tile8_off) { WLAUNCH(128, 128, 2, 4, false, true); }
    else WCFG(128, 128, 2, 2)
    else WCFG(128, 64, 2, 2)
    else WCFG(64, 128, 2, 2)
    else WCFG(64, 64, 2, 2)
    else WCFG(32, 128, 1, 4)
    else WCFG(128, 32, 4, 1)
    else WCFG(128, 96, 4, 1)
    else WCFG(96, 128, 1, 4)
    else { mrfa_set_error("wgrad: no tile config"); return 1; }
#undef WCFG
#undef WLAUNCH
    MRFA_CHECK_LAUNCH("mrfa_conv2d_wgrad_nhwc");
    if (partial_stride) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(stream_grid(wn * ((nsplit + 15) / 16), 256)), dim3(256), 0, st, p.ws, p.dw, wn, nsplit);
        MRFA_CHECK_LAUNCH("wgrad_reduce");
    }
    return 0;
}
