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
#include "common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

// SPATIAL: the conv has taps or a size change (false: 1x1 same-size, X row = dY row); POW2: Hout, Wout are powers of two (pixel ->
// (image, y, x) by shifts).  Compile-time: as run-time flags they put 2-3 uniform branches around every load group, ~48 per trip of
// the 4-chunk loop, and hipcc would not move loads across them.
template <bool SPATIAL, bool POW2>
__global__ __launch_bounds__(256) void wgrad_small_kernel(const mrfa_wgrad_params p, const long long M, const int chunks_per_wave, const int nblk_ci,
                                                         const int nblk, const int pow2_w /* log2(Wout) */, const int pow2_hw) {
    chain_prio();
    // two 32 x 32 partial buffers (plain stores: LDS float atomics cost 11 us here), filled in two rounds: 8.4 KB instead of 16.9 KB, so a
    // workgroup fits beside two resident 74 KB workgroups of the deferred wgrad_bf16x6 chain (12 KB of a CU's LDS stay free) instead of
    // waiting ~250 us for one of them to retire (measured tail of this kernel during the overlap: up to 530 us for a 15 us launch)
    __shared__ float sacc[2][32][33];
    __shared__ float sbias[32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;
    // blockIdx.x = (range group g, tap, block): the 4 waves of a workgroup = pixel ranges 4g .. 4g+3 of the same (tap, block)
    const int T = p.R * p.S;
    const int blk = blockIdx.x % nblk;
    const int tap = (blockIdx.x / nblk) % T;
    const int g = blockIdx.x / (nblk * T);
    const int co0 = (blk / nblk_ci) * 32, ci0 = (blk % nblk_ci) * 32;
    const int dr = tap / p.S - p.pad, ds = tap % p.S - p.pad;
    const int Mi = (int)M;                                 // (eligibility: M <= 65536) 32-bit pixel arithmetic: 64-bit division is a branchy routine
    const int range = (4 * g + wave) * chunks_per_wave * 16;
    const int HWo = p.Hout * p.Wout;
    const int wstride = p.stride > 1 ? p.stride : 1;       // strided layers (hr_base.py:241,253,302,305,365): dY pixel (oy, ox) <-> X pixel (s oy + r - pad, ..)

    if (threadIdx.x < 32) sbias[threadIdx.x] = 0.f;
    __syncthreads();

    f32x4v acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[t][u] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};
    const float* dyp = p.dy + co0 + 2 * li;
    const float* xp = p.x + ci0 + 2 * li;

    // BRANCH-FREE loads (a load inside a divergent branch makes hipcc wait vmcnt(0) at the join, which serialises the prefetch):
    // pixels past the end / taps outside the image read a valid row (the last pixel / the pixel itself) and are zeroed by a select
    auto load = [&](int m0c, f32x2v (&a)[4], f32x2v (&b)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mraw = m0c + 4 * j + kq;
            const bool ok = mraw < Mi;
            const int m = ok ? mraw : Mi - 1;
            const f32x2v av = *reinterpret_cast<const f32x2v*>(dyp + (size_t)m * p.ldy);
            a[j] = ok ? av : f32x2v{0.f, 0.f};
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
            const f32x2v bv = *reinterpret_cast<const f32x2v*>(xp + (size_t)xrow * p.ldx);
            b[j] = xok ? bv : f32x2v{0.f, 0.f};
        }
    };
    auto compute = [&](const f32x2v (&a)[4], const f32x2v (&b)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][t], b[j][u], acc[t][u], 0, 0, 0);
            bsum[0] += a[j][0];
            bsum[1] += a[j][1];
        }
    };

    // four-deep register ring over this wave's 16-pixel chunks (measured: with one chunk in flight the loop was load-latency bound,
    // ~1 us per round trip under load); loads past this wave's range are clamped by `load` itself and never consumed
    f32x2v ra[4][4], rb[4][4];
    int nc = 0;
    if (range < Mi) {
        const int left = (Mi - range + 15) / 16;
        nc = left < chunks_per_wave ? left : chunks_per_wave;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) load(range + s * 16, ra[s], rb[s]);
#ifdef MRFA_AB_NO_LOOP
    nc = nc > 1 ? 1 : nc;
#endif
    for (int c0 = 0; c0 < nc; c0 += 4) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            load(range + (c0 + s + 3) * 16, ra[(s + 3) & 3], rb[(s + 3) & 3]);
            if (c0 + s < nc) compute(ra[s], rb[s]);
        }
    }
    // C/D layout: column = lane & 15 (B row index i_b -> ci), row = (lane >> 4) * 4 + r (A row index i_a -> co)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (wave < 2) sacc[wave][2 * (kq * 4 + r) + t][2 * li + u] = acc[t][u][r];
    __syncthreads();
    if (wave >= 2) {                                   // round 2: waves 2, 3 add on top of waves 0, 1 (each element has one owner)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc[wave - 2][2 * (kq * 4 + r) + t][2 * li + u] += acc[t][u][r];
    }
    if (p.dbias && tap == 0 && ci0 == 0) {
        // column sums of dY over this wave's pixels: lane (li, kq) summed channels 2 li + {0, 1} of its pixels
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v = bsum[t];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kq == 0) atomicAdd(&sbias[2 * li + t], v);
        }
    }
    __syncthreads();
    float* dw = p.dw + ((size_t)tap * p.Cout + co0) * p.Cin + ci0;
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int r = i >> 5, c = i & 31;
#ifdef MRFA_AB_NO_ATOMICS
        if (co0 + r < p.Cout && ci0 + c < p.Cin) dw[(size_t)r * p.Cin + c] = p.alpha * (sacc[0][r][c] + sacc[1][r][c]);
#else
        if (co0 + r < p.Cout && ci0 + c < p.Cin) atomicAdd(dw + (size_t)r * p.Cin + c, p.alpha * (sacc[0][r][c] + sacc[1][r][c]));
#endif
    }
    if (p.dbias && tap == 0 && ci0 == 0 && threadIdx.x < 32 && co0 + threadIdx.x < p.Cout) atomicAdd(p.dbias + co0 + threadIdx.x, sbias[threadIdx.x]);
}

}  // namespace

bool mrfa_wgrad_small_eligible(const mrfa_wgrad_params& p, long long M) {
    if (p.kflat > 0 || p.ups || p.in_scale || p.nbatch > 1 || p.ksplit > 0 || p.stride > 2) return false;
    if ((p.Cin & 31) || (p.Cout & 31) || (p.ldx & 1) || (p.ldy & 1)) return false;
    if ((reinterpret_cast<uintptr_t>(p.x) & 7) || (reinterpret_cast<uintptr_t>(p.dy) & 7)) return false;
    if (M > 65536 || p.Cout > 640 || p.Cin > 640) return false;
    // what the 128-wide tiles do well stays there: >= 128 x 128 weights per tap over many pixels
    if (p.stride == 2) return true;                  // (the only weight-gradient kernel with a strided gather)
    if (p.Cout >= 128 && p.Cin >= 128 && M > 4096) return false;                  // (the only weight-gradient kernel with a strided gather)
    if (2.0 * (double)M * p.Cout * (double)p.Cin * p.R * p.S > 1.3e9) return false;
    return true;
}

extern "C" int mrfa_conv2d_wgrad_stride_supported(const mrfa_wgrad_params* p) {
    if (!p || p->stride != 2) return 0;
    if (p->Hout != (p->Hin + 2 * p->pad - p->R) / 2 + 1 || p->Wout != (p->Win + 2 * p->pad - p->S) / 2 + 1) return 0;
    return mrfa_wgrad_small_eligible(*p, (long long)p->N * p->Hout * p->Wout) ? 1 : 0;
}

int mrfa_wgrad_small_launch(hipStream_t st, const mrfa_wgrad_params& p, long long M) {
    const int T = p.R * p.S;
    const int nblk_ci = p.Cin / 32, nblk = (p.Cout / 32) * nblk_ci;
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
    dim3 grid((unsigned)(groups * T * nblk));
    const bool spatial = T > 1 || p.Hin != p.Hout || p.Win != p.Wout;
    if (!spatial) hipLaunchKernelGGL((wgrad_small_kernel<false, false>), grid, dim3(256), 0, st, p, M, chunks_per_wave, nblk_ci, nblk, 0, 0);
    else if (pw >= 0) hipLaunchKernelGGL((wgrad_small_kernel<true, true>), grid, dim3(256), 0, st, p, M, chunks_per_wave, nblk_ci, nblk, pw, pw + ph);
    else hipLaunchKernelGGL((wgrad_small_kernel<true, false>), grid, dim3(256), 0, st, p, M, chunks_per_wave, nblk_ci, nblk, 0, 0);
    MRFA_CHECK_LAUNCH("mrfa_conv2d_wgrad_nhwc(small)");
    return 0;
}
