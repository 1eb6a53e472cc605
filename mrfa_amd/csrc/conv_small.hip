// Small-problem implicit-GEMM convolution: ONE WAVE per output tile, operands straight from L1/L2 into MFMA registers.
//
// The MTIA prior (HRNet-W32 stages + the token transformer's nn.Linear layers) issues ~600 convolutions per pass with 0.1-0.6 GFLOP
// each: 32->32 @64^2, 64->64 @32^2, 128->128 @16^2 3x3 convolutions and 192<->576 linears over 2 208 token rows.  On the LDS-tiled
// 64..128-row workgroup tiles of conv_mfma.hip / conv_split.hip they ran at 10-25 TF/s (20-57 us each; profiles/r2_encoder_convs.log):
// too few workgroups to fill 256 CUs, a barrier and an exposed global-load round trip per 32-deep k-step, split-K init / epilogue
// launches around 6-step k-loops, and 50-75 % padding of 128-wide tiles for 32..64 output channels.
//
// Here a wave64 owns a (16 TMW) x (16 TNW) output tile built from v_mfma_f32_16x16x4_f32 (exact fp32, 32 cycles, 4 accumulator
// registers per 16x16 tile).  A and B fragments of that MFMA are ONE float per lane (lane l: row l & 15, k = l >> 4), so a lane
// loads float4 = k 4q..4q+3 of its row (q = l >> 4) for a 16-deep k-chunk and feeds components x, y, z, w to four successive MFMAs
// (A and B use the same k permutation, as in conv_mfma.hip).  No LDS, no barrier, no split-K: latency is hidden by an 8-chunk
// register prefetch ring and by 1 000+ independent waves per launch; the 64-byte row segments of a chunk are adjacent in the NHWC
// row, so successive chunks hit the lines the previous ones brought into L1.  Output tiles of 16 x 16 ... 32 x 32 per wave put
// 256-1 024 waves on a 0.6 GFLOP layer where the 128-row tiles had 64-256 workgroups.
//
// Same arguments, packed-weight layout ([tap][Cout up to a multiple of 128][Cin], pack mode 0 / 2) and epilogue semantics as the
// chunked path of mrfa_conv2d_nhwc (bias, output affine, residual, ReLU, accumulate, per-channel sum / sum-of-squares for
// train-mode BatchNorm): mrfa_conv2d_nhwc dispatches here (mrfa_conv_small_launch) when the shape qualifies.
#include "common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));


// DEPTH = k16 chunks in flight per wave: 4 for short K (a 12-chunk linear would mostly prefetch past its end), 8 for K >= 512 (the loop is
// load-latency bound at one wave per SIMD: ~1.5 us per round trip under load)
// CK = k16 chunks per loop step (2: the 16 x 16 / 16 x 32 wave tiles, whose 4-8 MFMAs per k16 did not cover the per-step loader advance:
// two scalar branches, the tap bookkeeping and a waitcnt per step)
template <int TMW, int TNW, int DEPTH, int CK>
__global__ __launch_bounds__(256) void conv_small_kernel(const mrfa_conv_params p, const long long M, const int tiles_n, const int wave_tiles_m) {
    chain_prio();
    constexpr int WM = 16 * TMW, WN = 16 * TNW;
    __shared__ float sred[4][2][WN];                  // per-wave column sums for the BatchNorm statistics
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kq = lane >> 4;         // MFMA operand / result role of this lane: row (column) li, k-quad kq
    // LOADER role: lane 4 r + q loads the float4 = k-quad q of row r, so four consecutive lanes read one contiguous 64-byte row segment
    // and a wave-wide load touches 16 cache lines.  (With the loader in the MFMA role -- lane = 16 q + r -- consecutive lanes read 16
    // different rows and every load instruction cost 64 L1 tag lookups: TCP_TOTAL_CACHE_ACCESSES / SQ_INSTS_VMEM_RD = 64.5, and lookups
    // + tag-conflict + pending-miss stalls added up to 97 % of the kernel's cycles per CU -- profiles/r2_pmc_conv_small_*.)  The
    // registers then move to the MFMA role with one ds_bpermute_b32 per component: lane (kq, li) takes from lane 4 li + kq.
    const int lr = lane >> 2, lk = lane & 3;
    const int perm_src = (4 * li + kq) * 4;           // ds_bpermute byte address of the source lane
    const int tile_n = blockIdx.x % tiles_n;
    const long long wt_m = (long long)(blockIdx.x / tiles_n) * 4 + wave;          // this wave's row tile
    const bool wave_on = wt_m < wave_tiles_m;
    const long long m0 = wt_m * WM;
    const int n0 = tile_n * WN;
    const int Wq = p.Wout;
    const int HWo = p.Hout * p.Wout;
    const int KC = p.Cin / (16 * CK);                 // loop steps (CK k16 chunks each) per tap
    const int ns = p.S;
    const int nq = p.R * p.S * KC;
    const int cstride = p.stride > 1 ? p.stride : 1;

    int a_oy[TMW], a_ox[TMW];
    long long a_img[TMW];
    bool a_ok[TMW];
#pragma unroll
    for (int a = 0; a < TMW; ++a) {
        const long long m = m0 + a * 16 + lr;
        a_ok[a] = wave_on && m < M;
        const long long mm = a_ok[a] ? m : 0;
        const int n_img = (int)(mm / HWo);
        const int rem = (int)(mm - (long long)n_img * HWo);
        a_oy[a] = rem / Wq;
        a_ox[a] = rem - a_oy[a] * Wq;
        a_img[a] = (long long)n_img * p.Hin * p.Win;
    }
    // B rows: packed weights are zero-padded to a multiple of 128 rows, so every row n0 + b*16 + li < w_rows is readable
    const float* wrow[TNW];
#pragma unroll
    for (int b = 0; b < TNW; ++b) {
        int n = n0 + b * 16 + lr;
        if (n >= p.w_rows) n = p.w_rows - 1;          // (its products land in columns >= Cout, which are never stored)
        wrow[b] = p.w + (size_t)n * p.w_ld + lk * 4;
    }

    f32x4v ra[DEPTH][CK][TMW], rb[DEPTH][CK][TNW];
    // loader state: tap (l_r, l_s), chunk l_c; per-tap A row pointers.  BRANCH-FREE: a row outside the image reads a valid address
    // (its image's pixel 0) and is zeroed by a select -- a load inside a divergent branch makes hipcc wait vmcnt(0) at the join, i.e.
    // for the newest prefetch instead of the oldest (measured: 20 us per layer instead of 8).  Past the last chunk the loader keeps
    // re-reading the last one (never consumed), so the k-loop needs no tail conditions either.
    int l_r = 0, l_s = 0, l_c = 0, l_q = 0;
    const float* arow[TMW];
    bool ainb[TMW];
    size_t w_tap_off = 0;
    auto set_tap = [&]() {
        const int dr = l_r - p.pad, ds = l_s - p.pad;
#pragma unroll
        for (int a = 0; a < TMW; ++a) {
            const int iy = a_oy[a] * cstride + dr, ix = a_ox[a] * cstride + ds;          // (strided layers: hr_base.py:241,253,302,305,365)
            ainb[a] = a_ok[a] && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            const long long pix = ainb[a] ? a_img[a] + (long long)iy * p.Win + ix : a_img[a];
            arow[a] = p.x + (size_t)pix * p.ldx + lk * 4;
        }
    };
    set_tap();
    auto load_chunk = [&](int slot) {
        const int c0 = l_c * 16 * CK;
#pragma unroll
        for (int h = 0; h < CK; ++h) {
#pragma unroll
            for (int a = 0; a < TMW; ++a) {
                const f32x4v v = *reinterpret_cast<const f32x4v*>(arow[a] + c0 + 16 * h);
                ra[slot][h][a] = ainb[a] ? v : f32x4v{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int b = 0; b < TNW; ++b) rb[slot][h][b] = *reinterpret_cast<const f32x4v*>(wrow[b] + w_tap_off + c0 + 16 * h);
        }
        if (l_q + 1 < nq) {                          // (uniform) advance; at the end: stay on the last chunk
            ++l_q;
            if (++l_c == KC) {
                l_c = 0;
                if (++l_s == ns) { l_s = 0; ++l_r; }
                w_tap_off += p.w_tap;
                set_tap();
            }
        }
    };

    f32x4v acc[TMW][TNW];
#pragma unroll
    for (int a = 0; a < TMW; ++a)
#pragma unroll
        for (int b = 0; b < TNW; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};
    f32x4v pa[2][CK][TMW], pb[2][CK][TNW];            // operands in the MFMA role, double-buffered: step q + 1 is permuted while q multiplies
    auto permute = [&](int slot, int pbuf) {
#pragma unroll
        for (int h = 0; h < CK; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int a = 0; a < TMW; ++a)
                    pa[pbuf][h][a][j] = __int_as_float(__builtin_amdgcn_ds_bpermute(perm_src, __float_as_int(ra[slot][h][a][j])));
#pragma unroll
                for (int b = 0; b < TNW; ++b)
                    pb[pbuf][h][b][j] = __int_as_float(__builtin_amdgcn_ds_bpermute(perm_src, __float_as_int(rb[slot][h][b][j])));
            }
    };
    auto compute = [&](int pbuf) {
#pragma unroll
        for (int h = 0; h < CK; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < TMW; ++a)
#pragma unroll
                    for (int b = 0; b < TNW; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[pbuf][h][a][j], pb[pbuf][h][b][j], acc[a][b], 0, 0, 0);
    };

    // register ring of DEPTH chunks: chunk q lives in slot q % DEPTH; the loop is unrolled by DEPTH so every index is static.
    // (waves past the last row tile run the loop on clamped addresses and store nothing: no divergent control flow around the loads)
#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s) load_chunk(s);
    permute(0, 0);
    for (int q0 = 0; q0 < nq; q0 += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            load_chunk((s + DEPTH - 1) % DEPTH);
            permute((s + 1) % DEPTH, (s + 1) & 1);    // chunk q0 + s + 1 (past the end: a clamped re-read, never multiplied)
            if (q0 + s < nq) compute(s & 1);
        }
    }

    // ---- epilogue: C/D layout of the 16x16 MFMA: column = lane & 15, row = (lane >> 4) * 4 + reg
    // (statistic groups: the four row tiles of a workgroup lie in one group -- the launcher checks rows per group % (4 WM) == 0)
    const int grp = stat_group(p, (long long)(blockIdx.x / tiles_n) * 4 * WM, M);
    float s1[TNW], s2[TNW];
#pragma unroll
    for (int b = 0; b < TNW; ++b) {
        const int c = n0 + b * 16 + li;
        const bool c_ok = wave_on && c < p.Cout;
        float bias = 0.f, osc = 1.f, osh = 0.f;
        float bsc = 0.f, bsh = 0.f, bmean = 0.f, binv = 0.f;          // bst_*: the BatchNorm whose output's gradient this launch writes
        if (c_ok) {
            if (p.bias) bias = p.bias[c];
            if (p.out_scale) { osc = p.out_scale[c]; osh = p.out_shift[c]; }
            if (p.bst_x) { const int gc = grp * p.Cout + c; bsc = p.bst_scale[gc]; bsh = p.bst_shift[gc]; bmean = p.bst_mean[gc]; binv = p.bst_invstd[gc]; }
        }
        s1[b] = 0.f; s2[b] = 0.f;
#pragma unroll
        for (int a = 0; a < TMW; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long m = m0 + a * 16 + kq * 4 + r;
                if (c_ok && m < M) {
                    float v = acc[a][b][r] * p.alpha + bias;
                    v = v * osc + osh;
                    float* dst = p.y + (size_t)m * p.ldy + c;
                    if (p.res) v += p.res[(size_t)m * p.ldr + c];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.accumulate) v += *dst;
                    *dst = v;
                    if (p.bst_x) {
                        // first phase of the BatchNorm backward (mrfa_conv_params.bst_*): v = d(act(bn(x))); through the activation, then the two sums
                        const float xr = p.bst_x[(size_t)m * p.bst_ldx + c];
                        const float du = (p.bst_relu && xr * bsc + bsh <= 0.f) ? 0.f : v;
                        s1[b] += du;
                        s2[b] += du * ((xr - bmean) * binv);
                    } else {
                        s1[b] += v;
                        s2[b] += v * v;
                    }
                }
            }
    }
    if (p.stats) {
        // column sums: over the 4 row groups of a wave (lanes li, li+16, li+32, li+48), then over the 4 waves through LDS: one
        // fp64 atomic per column per workgroup
#pragma unroll
        for (int b = 0; b < TNW; ++b) {
            s1[b] += __shfl_xor(s1[b], 16, 64); s1[b] += __shfl_xor(s1[b], 32, 64);
            s2[b] += __shfl_xor(s2[b], 16, 64); s2[b] += __shfl_xor(s2[b], 32, 64);
            if (kq == 0) { sred[wave][0][b * 16 + li] = s1[b]; sred[wave][1][b * 16 + li] = s2[b]; }
        }
        __syncthreads();
        if (threadIdx.x < 2 * WN) {
            const int which = threadIdx.x / WN, col = threadIdx.x % WN;
            const int c = n0 + col;
            if (c < p.Cout) {
                const double v = (double)sred[0][which][col] + (double)sred[1][which][col] + (double)sred[2][which][col] + (double)sred[3][which][col];
                atomicAdd(stat_slot(p, grp, blockIdx.x) + which * p.Cout + c, v);
            }
        }
        if (p.fin_scale) fused_bn_finalize(p, gridDim.x, (int)blockIdx.x);       // (common.h: the launch's last workgroup finishes the BatchNorm that follows)
    }
}

}  // namespace

// 1: the shape runs here.  Chunked (non-flat) layout, no fused upsample / pre-activation prologue / batched GEMM, channels in
// multiples of 16, and a problem small enough that the 128-row workgroup tiles cannot fill the chip.
bool mrfa_conv_small_eligible(const mrfa_conv_params& p, long long M) {
    if (p.kflat > 0 || p.ups || p.in_scale || p.nbatch > 1 || p.tile || p.splitk > 1 || p.mask) return false;
    if (p.stride > 2 || p.stride < 0) return false;
    if (p.groups > 1 && (group_rows(p, M) % 64) != 0) return false;      // statistic groups: a workgroup (4 waves x 16 / 32 rows) stays inside one group
    const bool strided = p.stride == 2;              // the only kernel with a strided gather: takes every such layer that fits its addressing
    if ((p.Cin & 15) || (p.ldx & 3) || (p.w_ld & 3)) return false;
    if (!aligned16(p.x) || !aligned16(p.w)) return false;
    const long long ktot = (long long)p.R * p.S * p.Cin;
    // what the big tiles do well stays there: long K with >= 256 row tiles of 128, or wide outputs at large M
    const long long big_tiles = ((M + 127) / 128) * ((p.Cout + 127) / 128);
    if (p.Cout > 640 || ktot > 1152) return false;
    if (M > 65536) return false;
    if (strided) return true;
    if (big_tiles >= 512 && p.Cout > 64) return false;
    // ~1.3 GFLOP at most: beyond that the LDS-tiled 128-row tiles (operand reuse across 4-8 waves, bf16 pipe) are the faster kernels
    if (2.0 * (double)M * p.Cout * (double)ktot > 1.3e9) return false;
    return true;
}

int mrfa_conv_small_launch(hipStream_t st, const mrfa_conv_params& p, long long M) {
    // wave tile: the largest of 32x32 / 16x32 / 16x16 that still yields >= ~2 000 waves (two per SIMD: measured best once the loads coalesce)
    const int ncols = (p.Cout + 15) / 16 * 16;
    auto waves = [&](int wm, int wn) { return ((M + wm - 1) / wm) * ((ncols + wn - 1) / wn); };
    int tm = 2, tn = 2;
    if (waves(32, 32) < 2048) { tm = 1; tn = 2; }
    if (tm == 1 && waves(16, 32) < 2048) { tn = 1; }
    if (ncols % 32 != 0 && tn == 2 && ncols < 32) tn = 1;
    if (p.groups > 1 && tm == 2 && (group_rows(p, M) % 128) != 0) tm = 1;       // (eligibility guarantees % 64)
    const int WM = 16 * tm, WN = 16 * tn;
    const int tiles_n = (ncols + WN - 1) / WN;
    const int wave_tiles_m = (int)((M + WM - 1) / WM);
    const long long blocks = (long long)((wave_tiles_m + 3) / 4) * tiles_n;
    dim3 grid((unsigned)blocks);
    const bool deep = (long long)p.R * p.S * p.Cin >= 512;
    // two k16 chunks per loop step: small wave tiles on deep K (measured: 64->64 3x3 14.8 -> 13.6 us, 128->128 15.6 -> 13.9, 576->192
    // linear 14.4 -> 13.0; short K gets slower: 32->32 3x3 14.0 -> 15.6, 192->576 12.6 -> 15.1)
    const bool ck2 = deep && (p.Cin % 32) == 0 && !(tm == 2 && tn == 2);
#define SMALL_LAUNCH(TM_, TN_)                                                                                                    \
    do {                                                                                                                          \
        if (ck2) hipLaunchKernelGGL((conv_small_kernel<TM_, TN_, 4, 2>), grid, dim3(256), 0, st, p, M, tiles_n, wave_tiles_m);   \
        else if (deep) hipLaunchKernelGGL((conv_small_kernel<TM_, TN_, 8, 1>), grid, dim3(256), 0, st, p, M, tiles_n, wave_tiles_m);    \
        else hipLaunchKernelGGL((conv_small_kernel<TM_, TN_, 4, 1>), grid, dim3(256), 0, st, p, M, tiles_n, wave_tiles_m);              \
    } while (0)
    if (tm == 2 && tn == 2) SMALL_LAUNCH(2, 2);
    else if (tm == 1 && tn == 2) SMALL_LAUNCH(1, 2);
    else SMALL_LAUNCH(1, 1);
#undef SMALL_LAUNCH
    MRFA_CHECK_LAUNCH("mrfa_conv2d_nhwc(small)");
    return 0;
}
