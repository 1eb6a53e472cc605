// Weight-gradient convolution / batched TN GEMM on v_mfma_f32_32x32x2_f32.
//
//   dW[tap][co][ci] += alpha * sum_p dY[p][co] * X'[pix(p) + tap - pad][ci]        X' = relu(scale*ups(x)+shift)
//
// GEMM view: M = Cout, N = Cin, K = N*Hout*Wout pixels.  Both operands are pixel-major NHWC, i.e. K is the strided
// axis of both ("TN"): tiles are staged in LDS exactly as they sit in memory, [k][m] and [k][n] (ds_write_b128 of
// coalesced float4 loads), and MFMA fragments are fetched with conflict-free ds_read_b32 (lane i reads column i of
// row k = 2j + lane/32).  One MFMA = 64 cycles on its SIMD against two 4-byte LDS reads, so the matrix pipe is the
// bound.  The reduction over pixels is split across workgroups (gridDim.z) and combined with fp32 atomics into a
// zero-initialised [tap][Cout][Cin] buffer; the same launch optionally produces the bias gradient sum_p dY[p][co].
//
// Replaces the weight/bias-gradient half of aten::convolution_backward for every F.conv2d on the hot path
// (call sites listed in include/mrfa_hip.h) and d(k_s) of the correlation einsum (modules/raft.py:185).
#include "common.h"

namespace {

constexpr int WBK = 32;

template <int BM, int BN, int WAVES_M, int WAVES_N, bool FLAT>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(const mrfa_wgrad_params p, const long long M, const long long k_per_split,
                                                        const int tiles_n, const int nsplit, const int dy_scalar) {
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;   // per-wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "tile");
    constexpr int RA = BM / 32, RB = BN / 32;      // float4 loads per thread: 32 rows x (BM/4) float4 / 256 threads
    constexpr int CA = BM / 4, CB = BN / 4;        // float4 columns per row
    __shared__ __attribute__((aligned(16))) float smem[2 * WBK * (BM + BN)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int co0 = tile_m * BM, ci0 = tile_n * BN;
    const int tap = FLAT ? 0 : blockIdx.y;
    const int NTOT = FLAT ? p.kflat : p.Cin;                // extent of the GEMM N axis
    const int bz = blockIdx.z / nsplit, split = blockIdx.z - bz * nsplit;
    const int r = tap / p.S, s = tap - r * p.S;

    const float* __restrict__ x = p.x + (size_t)bz * p.x_bs;
    const float* __restrict__ dy = p.dy + (size_t)bz * p.dy_bs;
    float* __restrict__ dw = p.dw + (size_t)bz * p.dw_bs;

    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;
    const int HWo = p.Hout * p.Wout;
    const long long kb = (long long)split * k_per_split;
    const long long ke = min(M, kb + k_per_split);

    // loader geometry: A: CA float4 per row -> rows per pass = 256/CA
    const int a_col = (tid % CA) * 4, a_row0 = tid / CA;
    constexpr int A_RSTEP = 256 / CA;
    const int b_col = (tid % CB) * 4, b_row0 = tid / CB;
    constexpr int B_RSTEP = 256 / CB;
    const bool do_bias = (p.dbias != nullptr) && tap == 0 && tile_n == 0;

    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    bool cmask[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cmask[q] = (ci0 + b_col + q) < NTOT;
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
    for (int q = 0; q < 4; ++q) amask[q] = (co0 + a_col + q) < p.Cout;
    const bool a_any = (co0 + a_col) < p.Cout, b_any = (ci0 + b_col) < NTOT;

    f32x4 ra[RA], rb[RB];
    f32x4 bias_acc = {0.f, 0.f, 0.f, 0.f};

    auto load_tiles = [&](long long k0) {
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const long long pp = k0 + a_row0 + j * A_RSTEP;
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
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const long long pp = k0 + b_row0 + j * B_RSTEP;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pp < ke && b_any) {
                const int n_img = (int)(pp / HWo);
                const int rem = (int)(pp - (long long)n_img * HWo);
                const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
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
            rb[j] = v;
        }
    };
    auto store_tiles = [&](int buf) {
        float* As = smem + buf * WBK * (BM + BN);
        float* Bs = As + WBK * BM;
#pragma unroll
        for (int j = 0; j < RA; ++j) *reinterpret_cast<f32x4*>(As + (a_row0 + j * A_RSTEP) * BM + a_col) = ra[j];
#pragma unroll
        for (int j = 0; j < RB; ++j) *reinterpret_cast<f32x4*>(Bs + (b_row0 + j * B_RSTEP) * BN + b_col) = rb[j];
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < RA; ++j) bias_acc += ra[j];
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
    int cur = 0;
    for (long long k0 = kb; k0 < ke; k0 += WBK) {
        const bool more = (k0 + WBK) < ke;
        if (more) load_tiles(k0 + WBK);
        const float* As = smem + cur * WBK * (BM + BN) + wm * WTM + fi;
        const float* Bs = smem + cur * WBK * (BM + BN) + WBK * BM + wn * WTN + fi;
#pragma unroll
        for (int kk = 0; kk < WBK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(2 * kk + fh) * BM + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[(2 * kk + fh) * BN + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
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
                if (co < p.Cout) atomicAdd(dw + ((size_t)otap * p.Cout + co) * p.Cin + ci, acc[i][j][q] * p.alpha);
            }
        }
    }
    if (do_bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (amask[q] && a_any) atomicAdd(p.dbias + co0 + a_col + q, bias_acc[q]);
    }
}

}  // namespace

extern "C" int mrfa_conv2d_wgrad_nhwc(void* stream, const mrfa_wgrad_params* pp) {
    const mrfa_wgrad_params& p = *pp;
    hipStream_t st = (hipStream_t)stream;
    MRFA_CHECK_ARG(p.x && p.dy && p.dw, "wgrad: null pointer");
    const int dy_scalar = ((p.ldy % 4) == 0 && aligned16(p.dy) && (p.dy_bs % 4) == 0) ? 0 : 1;
    const bool flat = p.kflat > 0;
    if (!flat) MRFA_CHECK_ARG((p.ldx % 4) == 0 && aligned16(p.x), "wgrad: x must be a 16-B aligned view with ld %% 4 == 0");
    else MRFA_CHECK_ARG(p.ktab != nullptr, "wgrad: flat mode needs ktab");
    const long long M = (long long)p.N * p.Hout * p.Wout;
    const int nb = p.nbatch > 1 ? p.nbatch : 1;
    const int taps = flat ? 1 : p.R * p.S;
    const int NTOT = flat ? p.kflat : p.Cin;
    // tile selection: (BM over Cout) x (BN over Cin or taps*Cin)
    int BM = p.Cout > 64 ? 128 : (p.Cout > 32 ? 64 : 32);
    int BN = NTOT > 64 ? 128 : 64;
    if (BM == 32) BN = 128;
    const int tiles_m = cdiv(p.Cout, BM), tiles_n = cdiv(NTOT, BN);
    const long long base = (long long)tiles_m * tiles_n * taps * nb;
    int nsplit = p.ksplit;
    if (nsplit <= 0) {
        nsplit = (int)((1536 + base - 1) / base);
        const long long max_split = (M + 255) / 256;      // at least 8 k-tiles per split
        if (nsplit > max_split) nsplit = (int)max_split;
        if (nsplit < 1) nsplit = 1;
        if (nsplit > 1024) nsplit = 1024;
    }
    long long kps = (M + nsplit - 1) / nsplit;
    kps = (kps + WBK - 1) / WBK * WBK;
    nsplit = (int)((M + kps - 1) / kps);
    MRFA_CHECK_ARG((long long)nsplit * nb <= 65535, "wgrad: grid.z too large");
    dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)taps, (unsigned)(nsplit * nb));
#define WCFG(bm, bn, wm, wn)                                                                                                       \
    if (BM == bm && BN == bn) {                                                                                                    \
        if (flat) hipLaunchKernelGGL((wgrad_mfma_kernel<bm, bn, wm, wn, true>), grid, dim3(256), 0, st, p, M, kps, tiles_n, nsplit, dy_scalar); \
        else hipLaunchKernelGGL((wgrad_mfma_kernel<bm, bn, wm, wn, false>), grid, dim3(256), 0, st, p, M, kps, tiles_n, nsplit, dy_scalar); \
    }
    WCFG(128, 128, 2, 2)
    else WCFG(128, 64, 2, 2)
    else WCFG(64, 128, 2, 2)
    else WCFG(64, 64, 2, 2)
    else WCFG(32, 128, 1, 4)
    else { mrfa_set_error("wgrad: no tile config"); return 1; }
#undef WCFG
    MRFA_CHECK_LAUNCH("mrfa_conv2d_wgrad_nhwc");
    return 0;
}
