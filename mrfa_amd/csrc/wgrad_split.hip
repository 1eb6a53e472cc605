// Weight-gradient convolution on the bf16 matrix pipe with exactly split fp32 operands (see conv_split.hip for the
// arithmetic: x = x1 + x2 + x3 in bf16 pieces without error, six bf16 MFMA products, fp32 accumulate).
//
//   dW[tap][co][ci] += alpha * sum_p dY[p][co] * X'[pix(p) + tap - pad][ci]
//
// GEMM view as wgrad_mfma.hip: M = Cout, N = Cin, K = pixels, both operands pixel-major (K strided).  The bf16 MFMA wants
// EIGHT consecutive k per lane, so a loader thread owns an 8 (pixels) x 4 (channels) block: eight coalesced float4 row
// loads, transposed in registers into four 8-k vectors, each split into its three bf16x8 pieces and stored with one
// ds_write_b128 per (piece, channel).  LDS image per k16 slab and piece: [channel m][2 halves x 16 B] with a 48-byte row
// stride and a 32-byte skew between the two slabs: the 8 lanes of a ds_write_b128 group (4 k-groups x 2 channel quads)
// and the 16 rows of a ds_read_b128 fragment group both fall on distinct 16-byte bank slots.
// Covers the cases that dominate the step: 128/64-wide tiles, chunked (not flat) K, Wout % 8 == 0 and N*Hout*Wout % 32 == 0 (every
// 8-pixel k-group lies in one image row, no partial k-tiles); everything else stays on wgrad_mfma.hip.
#include "common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int WBK = 32;                       // pixels per k-tile (two k16 slabs)
constexpr int NT = 256;
constexpr int ROWB = 48;                      // LDS bytes per channel row (32 data + 16 pad)
constexpr int PLANE = 128 * ROWB;             // one piece of one operand of one slab
constexpr int SLAB = 6 * PLANE + 32;          // A pieces 1..3, B pieces 1..3 (+ skew: SLAB % 128 == 32)

__device__ __forceinline__ unsigned pack_hi16(float a, float b) {
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float chop_rest(float x) { return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

// 8 consecutive k -> the three bf16x8 pieces
__device__ __forceinline__ void split3x8(const float (&x)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = x[2 * q], b = x[2 * q + 1];
        p1[q] = pack_hi16(a, b);
        const float ra = chop_rest(a), rb = chop_rest(b);
        p2[q] = pack_hi16(ra, rb);
        p3[q] = pack_hi16(chop_rest(ra), chop_rest(rb));
    }
}

// BM x BN in {128 x 128, 64 x 128, 128 x 64}: waves 2 x 2, each (BM/2) x (BN/2); the 64-wide variants serve the layers with 64
// output or input channels (half of a 128-wide tile would be padding); loader items beyond the tile width stay idle
__device__ __forceinline__ unsigned rne16(float x) {          // plain bf16 (NP = 1): round to nearest even
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

template <int BM, int BN, int NP = 6>
__global__ __launch_bounds__(NT, 2) void wgrad_bf16x6_kernel(const mrfa_wgrad_params p, const long long M, const long long k_per_split,
                                                             const int tiles_n, const int nsplit, const int inner, const int total_splits,
                                                             const int taps, const long long partial_stride) {
    constexpr int TM = BM / 64, TN = BN / 64;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SLAB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int gsplit, t_in;
    if (total_splits >= 8) {                                // XCD-aware order (see wgrad_mfma.hip)
        const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
        gsplit = (jj / inner) * 8 + xcd;
        if (gsplit >= total_splits) return;
        t_in = jj - (jj / inner) * inner;
    } else {
        gsplit = blockIdx.x / inner;
        if (gsplit >= total_splits) return;
        t_in = blockIdx.x - gsplit * inner;
    }
    // taps < 0: "chunk-flat" N axis -- the tile's BN columns index (tap, ci) = divmod(column, Cin) of the flattened [taps][Cin] axis, so a
    // 192- or 160-channel input needs ceil(9 * Cin / 128) column tiles instead of 9 * 2 half-empty ones (Cin % 4 == 0: a loader
    // thread's channel quad never straddles two taps).  Only the loader's tap shift and the epilogue's address become per-column.
    const bool cflat = taps < 0;
    const int tps = cflat ? 1 : taps;
    const int tile = t_in / tps;
    const int tap = t_in - tile * tps;
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const int co0 = tile_m * BM, ci0 = tile_n * BN;
    const int bz = gsplit / nsplit, split = gsplit - bz * nsplit;
    const int NF = cflat ? p.R * p.S * p.Cin : p.Cin;        // length of the N axis this launch tiles

    const float* __restrict__ x = p.x + (size_t)bz * p.x_bs;
    const float* __restrict__ dy = p.dy + (size_t)bz * p.dy_bs;
    float* __restrict__ dw = p.dw + (size_t)bz * p.dw_bs;

    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;
    const int HWo = p.Hout * p.Wout;
    const int kb = (int)((long long)split * k_per_split);
    const int ke = (int)min(M, (long long)kb + k_per_split);

    // loader item: waves 0,1 stage A (dY), waves 2,3 stage B (X); lane -> (k-group of 8 pixels, quad of 4 channels)
    const bool is_a = tid < 128;
    const int it = tid & 127;
    // lane bits: b0 = low bit of the channel quad, b1 b2 = 8-pixel group, rest = channel quad >> 1.  Lane pairs then read 32 contiguous
    // bytes of a pixel row (with the pixel group in the low bits every one of the 64 lanes of a load hit its own cache line:
    // TCP_TOTAL_CACHE_ACCESSES / SQ_INSTS_VMEM = 58, and the L1 tag lookups alone outlasted the MFMA work of a k-tile), while any 8
    // consecutive lanes still hold {2 channel quads} x {4 pixel groups}: the ds_write_b128 bank pattern described above is unchanged.
    const int kg = (it >> 1) & 3;            // 8-pixel group 0..3 of the 32-pixel k-tile
    const int cg = ((it >> 3) << 1) | (it & 1);   // channel quad 0..31  (A/B on one box: step 101.1 -> 98.8 ms)
    const int col = cg * 4;
    const bool item = col < (is_a ? BM : BN);   // this thread has a column quad inside the tile
    const bool do_bias = (p.dbias != nullptr) && tap == 0 && tile_n == 0;
    // B loader threads: tap (r, s) and first input channel of this thread's column quad
    int r, s, bci;
    {
        const int nf = ci0 + col;
        const int tp = cflat ? min(nf / p.Cin, p.R * p.S - 1) : tap;
        bci = cflat ? nf - tp * p.Cin : nf;
        r = tp / p.S;
        s = tp - r * p.S;
    }

    bool cm[4];                              // channel masks of this thread's quad
#pragma unroll
    for (int q = 0; q < 4; ++q) cm[q] = item && (is_a ? (co0 + col + q) < p.Cout : (ci0 + col + q) < NF);
    const bool any = cm[0];
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (!is_a && p.in_scale) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (cm[q]) { sc[q] = p.in_scale[bci + q]; sh[q] = p.in_shift[bci + q]; }
    }

    f32x4 rg[8];                             // 8 pixel rows x 4 channels
    bool rok[8];
    f32x4 bias_acc = {0.f, 0.f, 0.f, 0.f};

    // Wout % 8 == 0: this thread's 8 consecutive pixels (k-group kg of the tile) lie in ONE image row; their (n, oy, ox0) advance
    // by one k-tile = 32 pixels per iteration (for Wout % 32 == 0 the whole tile stays in one row and the wrap never loops)
    int t_n, t_oy, t_ox0;
    {
        const int p0 = kb + 8 * kg;
        t_n = p0 / HWo;
        const int rem = p0 - t_n * HWo;
        t_oy = rem / p.Wout;
        t_ox0 = rem - t_oy * p.Wout;
    }

    auto load_tile = [&](int k0) {
        if (!item) {
        } else if (is_a) {
            const float* ap = dy + (size_t)(k0 + 8 * kg) * p.ldy + (any ? co0 + col : 0);
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                rg[rr] = *reinterpret_cast<const f32x4*>(ap + (size_t)rr * p.ldy);
                rok[rr] = true;
            }
        } else {
            const int ox = t_ox0;
            const int iy = t_oy + r - p.pad;
            const bool rowok = (unsigned)iy < (unsigned)Hv;
            const int iyc = rowok ? (iy >> p.ups) : 0;
            const float* bp = x + ((size_t)t_n * p.Hin + iyc) * p.Win * p.ldx + (any ? bci : 0);
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int ix = ox + rr + s - p.pad;
                const bool ok = rowok && (unsigned)ix < (unsigned)Wv;
                // out-of-bounds taps read column 0 of the row and are zeroed when stored
                rg[rr] = *reinterpret_cast<const f32x4*>(bp + (size_t)(ok ? (ix >> p.ups) : 0) * p.ldx);
                rok[rr] = ok;
            }
        }
        t_ox0 += WBK;
        while (t_ox0 >= p.Wout) {
            t_ox0 -= p.Wout;
            if (++t_oy == p.Hout) { t_oy = 0; ++t_n; }
        }
    };

    auto store_tile = [&]() {
        if (!item) return;
        // prologue / masks on the fp32 values, then transpose + split + store
        float v[4][8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            f32x4 t = rg[rr];
            if (!is_a && p.in_scale) {
                t = t * sc + sh;
                if (p.in_relu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) t[q] = fmaxf(t[q], 0.f);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q][rr] = (cm[q] && rok[rr]) ? t[q] : 0.f;
        }
        if (is_a && do_bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float sb = 0.f;
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) sb += v[q][rr];
                bias_acc[q] += sb;
            }
        }
        unsigned char* base = smem + (kg >> 1) * SLAB + (is_a ? 0 : 3 * PLANE) + (kg & 1) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x4 p1, p2, p3;
            unsigned char* d = base + (col + q) * ROWB;
            if constexpr (NP == 1) {
#pragma unroll
                for (int h = 0; h < 4; ++h) p1[h] = rne16(v[q][2 * h]) | (rne16(v[q][2 * h + 1]) << 16);
                *reinterpret_cast<u32x4*>(d + 0 * PLANE) = p1;
            } else {
                split3x8(v[q], p1, p2, p3);
                *reinterpret_cast<u32x4*>(d + 0 * PLANE) = p1;
                *reinterpret_cast<u32x4*>(d + 1 * PLANE) = p2;
                *reinterpret_cast<u32x4*>(d + 2 * PLANE) = p3;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    const int fi = lane & 31, fh = lane >> 5;
    auto compute_slab = [&](int sl) {
        const unsigned char* base = smem + sl * SLAB + fh * 16;
        constexpr int NPC = NP == 6 ? 3 : (NP == 3 ? 2 : 1);
        bf16x8 a[3][TM], b[3][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc)
                a[pc][i] = *reinterpret_cast<const bf16x8*>(base + pc * PLANE + (wm * (TM * 32) + i * 32 + fi) * ROWB);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc)
                b[pc][j] = *reinterpret_cast<const bf16x8*>(base + (3 + pc) * PLANE + (wn * (TN * 32) + j * 32 + fi) * ROWB);
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
        constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 6 - NP; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]][i], b[PB[t]][j], acc[i][j], 0, 0, 0);
    };

    if (kb < ke) {
        load_tile(kb);
        store_tile();
    }
    __syncthreads();
    for (int k0 = kb; k0 < ke; k0 += WBK) {
        const bool more = (k0 + WBK) < ke;
        if (more) load_tile(k0 + WBK);
        compute_slab(0);
        compute_slab(1);
        __syncthreads();
        if (more) store_tile();
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue (as wgrad_mfma.hip)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nfc = ci0 + wn * (TN * 32) + j * 32 + (lane & 31);
        if (nfc >= NF) continue;
        const int etap = cflat ? nfc / p.Cin : tap;
        const int ci = cflat ? nfc - etap * p.Cin : nfc;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int co = co0 + wm * (TM * 32) + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * fh;
                if (co < p.Cout) {
                    const size_t idx = ((size_t)etap * p.Cout + co) * p.Cin + ci;
                    if (partial_stride) p.ws[(size_t)split * partial_stride + idx] = acc[i][j][q] * p.alpha;
                    else atomicAdd(dw + idx, acc[i][j][q] * p.alpha);
                }
            }
        }
    }
    if (do_bias) {                                               // workgroup-uniform
        float* red = reinterpret_cast<float*>(smem);
        __syncthreads();
        if (is_a && item) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red[kg * BM + col + q] = bias_acc[q];
        }
        __syncthreads();
        if (tid < BM && co0 + tid < p.Cout) {
            float sacc = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) sacc += red[g * BM + tid];
            atomicAdd(p.dbias + co0 + tid, sacc);
        }
    }
}

}  // namespace

int mrfa_wgrad_split_launch(hipStream_t st, const mrfa_wgrad_params& p, dim3 grid, long long M, long long kps, int tiles_n, int nsplit, int inner,
                            int total_splits, int taps, long long partial_stride, int BM, int BN) {
    const bool three = mrfa_get_mfma_mode() == 2;        // bf16x3
    const bool one = mrfa_get_mfma_mode() == 3;          // plain bf16
    if (one && BM == 128 && BN == 128)
        hipLaunchKernelGGL((wgrad_bf16x6_kernel<128, 128, 1>), grid, dim3(NT), 0, st, p, M, kps, tiles_n, nsplit, inner, total_splits, taps, partial_stride);
    else if (one && BM == 64 && BN == 128)
        hipLaunchKernelGGL((wgrad_bf16x6_kernel<64, 128, 1>), grid, dim3(NT), 0, st, p, M, kps, tiles_n, nsplit, inner, total_splits, taps, partial_stride);
    else if (one && BM == 128 && BN == 64)
        hipLaunchKernelGGL((wgrad_bf16x6_kernel<128, 64, 1>), grid, dim3(NT), 0, st, p, M, kps, tiles_n, nsplit, inner, total_splits, taps, partial_stride);
    else if (three && BM == 128 && BN == 128)
        hipLaunchKernelGGL((wgrad_bf16x6_kernel<128, 128, 3>), grid, dim3(NT), 0, st, p, M, kps, tiles_n, nsplit, inner, total_splits, taps, partial_stride);
    else if (three && BM == 64 && BN == 128)
        hipLaunchKernelGGL((wgrad_bf16x6_kernel<64, 128, 3>), grid, dim3(NT), 0, st, p, M, kps, tiles_n, nsplit, inner, total_splits, taps, partial_stride);
    else if (three && BM == 128 && BN == 64)
        hipLaunchKernelGGL((wgrad_bf16x6_kernel<128, 64, 3>), grid, dim3(NT), 0, st, p, M, kps, tiles_n, nsplit, inner, total_splits, taps, partial_stride);
    else if (BM == 128 && BN == 128)
        hipLaunchKernelGGL((wgrad_bf16x6_kernel<128, 128>), grid, dim3(NT), 0, st, p, M, kps, tiles_n, nsplit, inner, total_splits, taps, partial_stride);
    else if (BM == 64 && BN == 128)
        hipLaunchKernelGGL((wgrad_bf16x6_kernel<64, 128>), grid, dim3(NT), 0, st, p, M, kps, tiles_n, nsplit, inner, total_splits, taps, partial_stride);
    else if (BM == 128 && BN == 64)
        hipLaunchKernelGGL((wgrad_bf16x6_kernel<128, 64>), grid, dim3(NT), 0, st, p, M, kps, tiles_n, nsplit, inner, total_splits, taps, partial_stride);
    else { mrfa_set_error("wgrad(bf16x6): no %dx%d tile", BM, BN); return 1; }
    MRFA_CHECK_LAUNCH("mrfa_conv2d_wgrad_nhwc(bf16x6)");
    return 0;
}
