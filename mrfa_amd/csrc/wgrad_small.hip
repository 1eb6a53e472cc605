// Small-problem weight gradient: dW[tap][co][ci] += alpha * sum_m dY[m][co] * X[m + tap][ci] for the MTIA prior's 0.1-0.6 GFLOP
// layers (32->32 @64^2, 64->64 @32^2, 128->128 @16^2 3x3 convolutions, 192<->576 linears over 2 208 token rows), which ran at
// 3-20 TF/s (28-57 us each) on the 64..128-wide LDS tiles of wgrad_mfma.hip / wgrad_split.hip: an output of 32 x 32 ... 128 x 128
// weights per tap pads a 128 x 128 tile by 4-16x, and the pixel axis (the GEMM's K) had to be split across workgroups and reduced
// by a second kernel.
//
// Here ONE WAVE owns a 32 (co) x 32 (ci) block of one tap over a range of pixels, with v_mfma_f32_16x16x4_f32 (exact fp32).  Both
// operands are pixel-major ([m][channel]), i.e. the GEMM's K axis is the slow one -- so a lane loads float2 = channels 2i, 2i+1 of
// pixel m (16 lanes x 8 B = one 128-byte line per pixel row; lane group q = lane >> 4 takes pixel m0 + 4j + q for MFMA j) and uses
// component t as the A (or B) operand of MFMA tile t: tile (t, u) then holds rows co0 + 2 i_a + t and columns ci0 + 2 i_b + u --
// a strided channel set per tile, which costs nothing (only the final store knows).  No LDS staging, no barrier in the loop.
// The four waves of a workgroup take four consecutive pixel ranges of the same block and add their 32 x 32 partials in LDS, so one
// workgroup issues 1 024 global atomics; the number of pixel ranges is chosen for ~2 000 waves per launch.
#include <vector>
#include "common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

// SPATIAL: the conv has taps or a size change (false: 1x1 same-size, X row = dY row); POW2: Hout, Wout are powers of two (pixel ->
// (image, y, x) by shifts).  Compile-time: as run-time flags they put 2-3 uniform branches around every load group, ~48 per trip of
// the 4-chunk loop, and hipcc would not move loads across them.
// one weight-gradient problem as the kernel sees it (the fields of mrfa_wgrad_params it reads + the launch geometry derived from them)
struct SmallProblem {
    const float* x; const float* dy; float* dw; float* dbias;
    int ldx, ldy, Hin, Win, Hout, Wout, Cin, Cout, R, S, pad, stride;
    float alpha;
    int M, chunks_per_wave, nblk_ci, nblk, pow2_w /* log2(Wout) */, pow2_hw, workgroups, groups8 /* range groups per XCD (multi-problem launches) */;
};

// ROWS: M % 16 == 0 and (same-size 1x1, or power-of-two output sizes with Wout % 16 == 0): a wave's 16-pixel chunk is then 16 CONSECUTIVE pixels of one
// image row, so image / row / first column are wave-uniform (scalar shifts of the chunk index) and a lane only adds its constant offset.  The general
// path decomposes every pixel of every load on the vector unit: 13.4 VALU instructions per MFMA against 0.29 M MFMAs per launch on the 32 -> 32 @64^2 layer
// (profiles/r4_pmc_wgrad_small: 3.96 M VALU, 42 % of the wave cycles waiting) -- the loop, not the 525 K atomics, was the kernel's 21.6 us.
// NT = 16-channel tiles per side of the wave's weight block: 2 (32 x 32 weights, float2 per lane and pixel) or 4 (64 x 64, float4: a wave-wide load then moves 64 B
// per L1 access instead of 32, and one loaded float feeds 4 instead of 2 MFMAs -- 2 instead of 8.25 L1 accesses per MFMA.  The L1's access rate is what bounds
// the 32 x 32 form once launch overhead is amortised: 4.07 M accesses per 32 -> 32 @64^2 problem = 6.6 us of its 10.5 us, tools/ubench/small_kernels 'multi')
template <bool SPATIAL, bool POW2, bool ROWS, int NT>
__device__ __forceinline__ void wgrad_small_body(const SmallProblem& p, const int g, const int tap, const int blk, float (*sacc)[16 * NT][16 * NT + 1], float* sbias) {
    typedef float VT __attribute__((ext_vector_type(NT)));
    constexpr int B = 16 * NT;
    const long long M = p.M;
    const int chunks_per_wave = p.chunks_per_wave, nblk_ci = p.nblk_ci, nblk = p.nblk, pow2_w = p.pow2_w, pow2_hw = p.pow2_hw;
    // two 32 x 32 partial buffers (plain stores: LDS float atomics cost 11 us here), filled in two rounds: 8.4 KB instead of 16.9 KB, so a
    // workgroup fits beside two resident 74 KB workgroups of the deferred wgrad_bf16x6 chain (12 KB of a CU's LDS stay free) instead of
    // waiting ~250 us for one of them to retire (measured tail of this kernel during the overlap: up to 530 us for a 15 us launch)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    // (range group g, tap, block): the 4 waves of a workgroup = pixel ranges 4g .. 4g+3 of the same (tap, block)
    const int co0 = (blk / nblk_ci) * B, ci0 = (blk % nblk_ci) * B;
    const int dr = tap / p.S - p.pad, ds = tap % p.S - p.pad;
    const int Mi = (int)M;                                 // (eligibility: M <= 65536) 32-bit pixel arithmetic: 64-bit division is a branchy routine
    const int range = (4 * g + wave) * chunks_per_wave * 16;
    const int HWo = p.Hout * p.Wout;
    const int wstride = p.stride > 1 ? p.stride : 1;       // strided layers (hr_base.py:241,253,302,305,365): dY pixel (oy, ox) <-> X pixel (s oy + r - pad, ..)

    if (threadIdx.x < B) sbias[threadIdx.x] = 0.f;
    __syncthreads();

    f32x4v acc[NT][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[t][u] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float bsum[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bsum[t] = 0.f;
    const VT vzero = {};
    const float* dyp = p.dy + co0 + NT * li;
    const float* xp = p.x + ci0 + NT * li;

    // ROWS: per-lane constants -- element offsets of this lane's pixel 4 j + kq inside a chunk
    int dy_lane[4], x_lane[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        dy_lane[j] = (4 * j + kq) * p.ldy + co0 + NT * li;
        x_lane[j] = (4 * j + kq) * p.ldx + ci0 + NT * li;           // (same-size 1x1 only; the spatial form computes its column below)
    }

    // BRANCH-FREE loads (a load inside a divergent branch makes hipcc wait vmcnt(0) at the join, which serialises the prefetch):
    // pixels past the end / taps outside the image read a valid row (the last pixel / the pixel itself) and are zeroed by a select
    auto load = [&](int m0c, VT (&a)[4], VT (&b)[4]) {
        if constexpr (ROWS) {
            int mc = __builtin_amdgcn_readfirstlane(m0c);           // the chunk index is wave-uniform: keep its arithmetic on the scalar unit
            mc = mc < Mi ? mc : Mi - 16;                            // past the end: re-read the last chunk (never consumed)
            const float* dyb = p.dy + (size_t)mc * p.ldy;
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const VT*>(dyb + dy_lane[j]);
            if constexpr (!SPATIAL) {
                const float* xb = p.x + (size_t)mc * p.ldx;
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const VT*>(xb + x_lane[j]);
            } else {
                const int img = mc >> pow2_hw, rem = mc & ((1 << pow2_hw) - 1);
                const int oy = rem >> pow2_w, ox0 = rem & ((1 << pow2_w) - 1);
                const int iy = oy * wstride + dr;
                const bool row_ok = (unsigned)iy < (unsigned)p.Hin;
                const float* xrow0 = p.x + (size_t)((img * p.Hin + (row_ok ? iy : 0)) * p.Win) * p.ldx + ci0 + NT * li;
                const int ix0 = ox0 * wstride + ds;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ix = ix0 + (4 * j + kq) * wstride;
                    const bool inb = row_ok && (unsigned)ix < (unsigned)p.Win;
                    const int ixc = ix < 0 ? 0 : (ix >= p.Win ? p.Win - 1 : ix);
                    const VT bv = *reinterpret_cast<const VT*>(xrow0 + (size_t)ixc * p.ldx);
                    b[j] = inb ? bv : vzero;
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mraw = m0c + 4 * j + kq;
            const bool ok = mraw < Mi;
            const int m = ok ? mraw : Mi - 1;
            const VT av = *reinterpret_cast<const VT*>(dyp + (size_t)m * p.ldy);
            a[j] = ok ? av : vzero;
            bool xok = ok;
            int xrow = m;
            if (SPATIAL) {
                int img, oy, ox;
                if (POW2) {
                    img = m >> pow2_hw;
                    const int rem = m & ((1 << pow2_hw) - 1);
                    oy = rem >> pow2_w;
                    ox = rem & ((1 << pow2_w) - 1);
                } else {
                    img = m / HWo;
                    const int rem = m - img * HWo;
                    oy = rem / p.Wout;
                    ox = rem - oy * p.Wout;
                }
                const int iy = oy * wstride + dr, ix = ox * wstride + ds;
                const bool inb = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                xok = ok && inb;
                xrow = inb ? (img * p.Hin + iy) * p.Win + ix : img * p.Hin * p.Win;
            }
            const VT bv = *reinterpret_cast<const VT*>(xp + (size_t)xrow * p.ldx);
            b[j] = xok ? bv : vzero;
        }
    };
    auto compute = [&](const VT (&a)[4], const VT (&b)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int u = 0; u < NT; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][t], b[j][u], acc[t][u], 0, 0, 0);
                bsum[t] += a[j][t];
            }
        }
    };

    // four-deep register ring over this wave's 16-pixel chunks (measured: with one chunk in flight the loop was load-latency bound,
    // ~1 us per round trip under load); loads past this wave's range are clamped by `load` itself and never consumed
    VT ra[4][4], rb[4][4];
    int nc = 0;
    if (range < Mi) {
        const int left = (Mi - range + 15) / 16;
        nc = left < chunks_per_wave ? left : chunks_per_wave;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) load(range + s * 16, ra[s], rb[s]);
    for (int c0 = 0; c0 < nc; c0 += 4) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            load(range + (c0 + s + 3) * 16, ra[(s + 3) & 3], rb[(s + 3) & 3]);
            if (c0 + s < nc) compute(ra[s], rb[s]);
        }
    }
    // C/D layout: column = lane & 15 (B row index i_b -> ci), row = (lane >> 4) * 4 + r (A row index i_a -> co)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (wave < 2) sacc[wave][NT * (kq * 4 + r) + t][NT * li + u] = acc[t][u][r];
    __syncthreads();
    if (wave >= 2) {                                   // round 2: waves 2, 3 add on top of waves 0, 1 (each element has one owner)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc[wave - 2][NT * (kq * 4 + r) + t][NT * li + u] += acc[t][u][r];
    }
    if (p.dbias && tap == 0 && ci0 == 0) {
        // column sums of dY over this wave's pixels: lane (li, kq) summed channels 2 li + {0, 1} of its pixels
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float v = bsum[t];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kq == 0) atomicAdd(&sbias[NT * li + t], v);
        }
    }
    __syncthreads();
    float* dw = p.dw + ((size_t)tap * p.Cout + co0) * p.Cin + ci0;
    for (int i = threadIdx.x; i < B * B; i += 256) {
        const int r = i / B, c = i % B;
        if (co0 + r < p.Cout && ci0 + c < p.Cin) atomicAdd(dw + (size_t)r * p.Cin + c, p.alpha * (sacc[0][r][c] + sacc[1][r][c]));
    }
    if (p.dbias && tap == 0 && ci0 == 0 && threadIdx.x < B && co0 + threadIdx.x < p.Cout) atomicAdd(p.dbias + co0 + threadIdx.x, sbias[threadIdx.x]);
}

template <bool SPATIAL, bool POW2, bool ROWS, int NT>
__global__ __launch_bounds__(256) void wgrad_small_kernel(const SmallProblem p) {
    chain_prio();
    __shared__ float sacc[2][16 * NT][16 * NT + 1];
    __shared__ float sbias[16 * NT];
    const int T = p.R * p.S, bid = blockIdx.x;
    wgrad_small_body<SPATIAL, POW2, ROWS, NT>(p, bid / (p.nblk * T), (bid / p.nblk) % T, bid % p.nblk, sacc, sbias);
}

// Up to SMALL_MULTI problems in ONE launch (mrfa_conv2d_wgrad_multi): the keypoint encoder's ~415 weight gradients per training step are
// independent of each other and issued together after its backward chains -- as single launches they were a 5 ms tail of the step (each pays the
// launch + ramp + drain of a 2 000-wave grid around ~10 us of work, four streams abreast gained 1.4x); here a workgroup finds its problem by its
// index (the record array travels in the kernel arguments) and runs the same body.
constexpr int SMALL_MULTI = 28;
struct SmallMulti {
    int n;
    int prefix[SMALL_MULTI + 1];                       // first workgroup of problem i
    SmallProblem d[SMALL_MULTI];
};

template <bool SPATIAL, bool POW2, bool ROWS, int NT>
__global__ __launch_bounds__(256) void wgrad_small_multi_kernel(const SmallMulti a) {
    chain_prio();
    __shared__ float sacc[2][16 * NT][16 * NT + 1];
    __shared__ float sbias[16 * NT];
    int lo = 0, hi = a.n;                              // largest i with prefix[i] <= blockIdx.x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int)blockIdx.x >= a.prefix[mid]) lo = mid; else hi = mid;
    }
    // XCD-AWARE decode.  Workgroups are dealt round-robin onto the 8 XCDs (blockIdx & 7), each with its own 4 MB L2.  With (tap, block) fastest, the T x nblk
    // workgroups of one pixel range sit on 8 XCDs and every L2 fetches ALL of X and dY: 0.9 M L2 misses = 116 MB per 32 -> 32 @64^2 problem
    // (profiles/r4_pmc_wgrad_small), which is what bounds this kernel once launch overhead is out of the way (10 us per problem in a 28-problem launch).
    // Every problem's workgroup count is a multiple of 8 (mrfa_conv2d_wgrad_multi pads: prefix[] stays 8-aligned), XCD x owns the contiguous range groups
    // [x' G8, (x' + 1) G8), x' = (x - problem index) mod 8 (rotated so that problems with fewer than 8 groups do not all land on XCD 0) with all their
    // taps and blocks; the padding exits here, before any barrier.
    const SmallProblem& p = a.d[lo];
    const int local = (int)blockIdx.x - a.prefix[lo];
    const int T = p.R * p.S, j = local >> 3;
    const int g = (((local & 7) - lo) & 7) * p.groups8 + j / (p.nblk * T);
    if ((long long)(4 * g) * p.chunks_per_wave * 16 >= p.M) return;
    wgrad_small_body<SPATIAL, POW2, ROWS, NT>(p, g, (j / p.nblk) % T, j % p.nblk, sacc, sbias);
}

// launch geometry of one problem -> kernel variant: 0 = spatial with power-of-two output sizes, 1 = spatial, 2 = same-size 1x1, 3 / 4 = the ROWS forms of 0 / 2,
// 5 / 6 = those with 64 x 64 weight blocks
constexpr int SMALL_VARIANTS = 7;
int small_problem(const mrfa_wgrad_params& p, long long M, SmallProblem* o) {
    const int T = p.R * p.S;
    constexpr bool rows_on = true, b64_on = true;     // (round 4's row-walking loads and 64 x 64 weight blocks: measured faster, no switch)
    const bool spatial0 = T > 1 || p.Hin != p.Hout || p.Win != p.Wout;
    auto lg2_ = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
    const bool rows0 = rows_on && (M % 16) == 0 && (!spatial0 || (lg2_(p.Wout) >= 0 && lg2_(p.Hout) >= 0 && (p.Wout % 16) == 0));
    // 64 x 64 weight blocks (float4 per lane and pixel): the ROWS forms of layers with 64-aligned channel counts on 16-byte aligned rows
    const bool b64 = b64_on && rows0 && (p.Cin % 64) == 0 && (p.Cout % 64) == 0 && (p.ldx % 4) == 0 && (p.ldy % 4) == 0 &&
                     (reinterpret_cast<uintptr_t>(p.x) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.dy) & 15) == 0;
    const int Bk = b64 ? 64 : 32;
    const int nblk_ci = p.Cin / Bk, nblk = (p.Cout / Bk) * nblk_ci;
    const long long chunks = (M + 15) / 16;
    long long groups = (2048 + (long long)nblk * T * 4 - 1) / ((long long)nblk * T * 4);      // workgroups (of 4 ranges) per block
    const long long max_groups = (chunks + 15) / 16;                                           // >= 4 chunks per wave
    if (groups > max_groups) groups = max_groups;
    if (groups < 1) groups = 1;
    const int chunks_per_wave = (int)((chunks + groups * 4 - 1) / (groups * 4));
    groups = (chunks + (long long)chunks_per_wave * 4 - 1) / ((long long)chunks_per_wave * 4);
    auto lg2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
    int pw = lg2(p.Wout), ph = lg2(p.Hout);
    if (ph < 0) pw = -1;
    const bool spatial = T > 1 || p.Hin != p.Hout || p.Win != p.Wout;
    o->x = p.x; o->dy = p.dy; o->dw = p.dw; o->dbias = p.dbias;
    o->ldx = p.ldx; o->ldy = p.ldy; o->Hin = p.Hin; o->Win = p.Win; o->Hout = p.Hout; o->Wout = p.Wout; o->Cin = p.Cin; o->Cout = p.Cout;
    o->R = p.R; o->S = p.S; o->pad = p.pad; o->stride = p.stride; o->alpha = p.alpha;
    o->M = (int)M; o->chunks_per_wave = chunks_per_wave; o->nblk_ci = nblk_ci; o->nblk = nblk;
    o->pow2_w = (spatial && pw >= 0) ? pw : 0; o->pow2_hw = (spatial && pw >= 0) ? pw + ph : 0;
    o->workgroups = (int)(groups * T * nblk);
    o->groups8 = (int)((groups + 7) / 8);
    const bool rows = rows0;
    if (!spatial) return rows ? (b64 ? 6 : 4) : 2;
    return pw >= 0 ? (rows ? (b64 ? 5 : 3) : 0) : 1;
}

}  // namespace

bool mrfa_wgrad_small_eligible(const mrfa_wgrad_params& p, long long M) {
    if (p.kflat > 0 || p.ups || p.in_scale || p.nbatch > 1 || p.ksplit > 0 || p.stride > 2) return false;
    if ((p.Cin & 31) || (p.Cout & 31) || (p.ldx & 1) || (p.ldy & 1)) return false;
    if ((reinterpret_cast<uintptr_t>(p.x) & 7) || (reinterpret_cast<uintptr_t>(p.dy) & 7)) return false;
    if (M > 65536 || p.Cout > 640 || p.Cin > 640) return false;
    // what the 128-wide tiles do well stays there: >= 128 x 128 weights per tap over many pixels
    if (p.stride == 2) return true;                  // (the only weight-gradient kernel with a strided gather)
    // (6 144 rows: the token transformer's 192 <-> 576 linears over the 16 x 276 rows of a batched source + driving encoder pass stay here -- the 128-row
    // tiles took 85 / 76 us for what this kernel does in ~15; round 4's limit was 4 096, one pass of 8 x 276 rows)
    if (p.Cout >= 128 && p.Cin >= 128 && M > 6144) return false;
    if (2.0 * (double)M * p.Cout * (double)p.Cin * p.R * p.S > 1.3e9) return false;
    return true;
}

extern "C" int mrfa_conv2d_wgrad_stride_supported(const mrfa_wgrad_params* p) {
    if (!p || p->stride != 2) return 0;
    if (p->Hout != (p->Hin + 2 * p->pad - p->R) / 2 + 1 || p->Wout != (p->Win + 2 * p->pad - p->S) / 2 + 1) return 0;
    return mrfa_wgrad_small_eligible(*p, (long long)p->N * p->Hout * p->Wout) ? 1 : 0;
}

int mrfa_wgrad_small_launch(hipStream_t st, const mrfa_wgrad_params& p, long long M) {
    SmallProblem sp;
    const int variant = small_problem(p, M, &sp);
    dim3 grid((unsigned)sp.workgroups);
    switch (variant) {
        case 0: hipLaunchKernelGGL((wgrad_small_kernel<true, true, false, 2>), grid, dim3(256), 0, st, sp); break;
        case 1: hipLaunchKernelGGL((wgrad_small_kernel<true, false, false, 2>), grid, dim3(256), 0, st, sp); break;
        case 2: hipLaunchKernelGGL((wgrad_small_kernel<false, false, false, 2>), grid, dim3(256), 0, st, sp); break;
        case 3: hipLaunchKernelGGL((wgrad_small_kernel<true, true, true, 2>), grid, dim3(256), 0, st, sp); break;
        case 4: hipLaunchKernelGGL((wgrad_small_kernel<false, false, true, 2>), grid, dim3(256), 0, st, sp); break;
        case 5: hipLaunchKernelGGL((wgrad_small_kernel<true, true, true, 4>), grid, dim3(256), 0, st, sp); break;
        default: hipLaunchKernelGGL((wgrad_small_kernel<false, false, true, 4>), grid, dim3(256), 0, st, sp); break;
    }
    MRFA_CHECK_LAUNCH("mrfa_conv2d_wgrad_nhwc(small)");
    return 0;
}

// n weight gradients in as few launches as possible: those the one-wave-per-block kernel takes are grouped by kernel variant and issued SMALL_MULTI at a
// time; every other one goes through mrfa_conv2d_wgrad_nhwc as if called alone.  The problems must be independent or accumulate atomically into shared
// outputs (dw / dbias are added with atomics: two problems may share them -- the two encoder passes of a training step do).
extern "C" int mrfa_conv2d_wgrad_multi(void* stream, const mrfa_wgrad_params* ps, int n) {
    MRFA_CHECK_ARG(n >= 0 && (n == 0 || ps), "wgrad_multi: bad args");
    hipStream_t st = (hipStream_t)stream;
    static const bool small_on = [] { const char* e = getenv("MRFA_CONV_SMALL"); return !(e && e[0] == '0'); }();
    SmallMulti batch[SMALL_VARIANTS];
    for (int v = 0; v < SMALL_VARIANTS; ++v) { batch[v].n = 0; batch[v].prefix[0] = 0; }
    auto flush = [&](int v) -> int {
        SmallMulti& b = batch[v];
        if (b.n == 0) return 0;
        dim3 grid((unsigned)b.prefix[b.n]);
        switch (v) {
            case 0: hipLaunchKernelGGL((wgrad_small_multi_kernel<true, true, false, 2>), grid, dim3(256), 0, st, b); break;
            case 1: hipLaunchKernelGGL((wgrad_small_multi_kernel<true, false, false, 2>), grid, dim3(256), 0, st, b); break;
            case 2: hipLaunchKernelGGL((wgrad_small_multi_kernel<false, false, false, 2>), grid, dim3(256), 0, st, b); break;
            case 3: hipLaunchKernelGGL((wgrad_small_multi_kernel<true, true, true, 2>), grid, dim3(256), 0, st, b); break;
            case 4: hipLaunchKernelGGL((wgrad_small_multi_kernel<false, false, true, 2>), grid, dim3(256), 0, st, b); break;
            case 5: hipLaunchKernelGGL((wgrad_small_multi_kernel<true, true, true, 4>), grid, dim3(256), 0, st, b); break;
            default: hipLaunchKernelGGL((wgrad_small_multi_kernel<false, false, true, 4>), grid, dim3(256), 0, st, b); break;
        }
        MRFA_CHECK_LAUNCH("mrfa_conv2d_wgrad_multi");
        b.n = 0;
        return 0;
    };
    // the residual blocks' 3x3 layers (<= 128 channels): all nine taps per staging, up to 24 problems per launch (wgrad_lean.hip)
    std::vector<unsigned char> taken((size_t)n, 0);
    if (n > 0) { const int rc = mrfa_wgrad_lean_multi(st, ps, n, taken.data()); if (rc) return rc; }
    for (int i = 0; i < n; ++i) {
        if (taken[i]) continue;
        const mrfa_wgrad_params& p = ps[i];
        MRFA_CHECK_ARG(p.x && p.dy && p.dw && p.N > 0 && p.Hout > 0 && p.Wout > 0, "wgrad_multi: problem %d: null pointer / bad sizes", i);
        const long long M = (long long)p.N * p.Hout * p.Wout;
        if (!(small_on && mrfa_tuning_conv_small() && mrfa_wgrad_small_eligible(p, M))) {
            const int rc = mrfa_conv2d_wgrad_nhwc(stream, &p);
            if (rc) return rc;
            continue;
        }
        SmallProblem sp;
        const int v = small_problem(p, M, &sp);
        SmallMulti& b = batch[v];
        b.d[b.n] = sp;
        b.prefix[b.n + 1] = b.prefix[b.n] + sp.groups8 * 8 * (p.R * p.S) * sp.nblk;      // (padded: see the kernel's XCD-aware decode)
        if (++b.n == SMALL_MULTI) { const int rc = flush(v); if (rc) return rc; }
    }
    for (int v = 0; v < SMALL_VARIANTS; ++v) { const int rc = flush(v); if (rc) return rc; }
    return 0;
}
