// Lean patch kernel for the keypoint encoder's small-channel 3x3 layers (HRNet BasicBlock: hr_base.py:34-63, branches of
// HighResolutionModule 121-289): 32 -> 32 @64^2, 64 -> 64 @32^2, 128 -> 128 @16^2 over 8..24 frames, forward and data gradient.
//
// Those layers are ~1.2 GFLOP each: ONE 32 x 32 MFMA output tile per SIMD of the chip.  conv_small.hip ran them at 23-25 us on the fp32 matrix
// pipe with every wave re-reading its nine shifted activation rows and its weight rows from L1 / L2 (221 MB of L2 -> L1 traffic for 8 MB of
// tensors, profiles/r5 + DESIGN 3f).  Here a workgroup of FOUR waves owns a small 2-D output patch of one image:
//   * the patch's input HALO over ALL input channels is loaded once (every global load of the workgroup is issued before anything else),
//     run through the optional BatchNorm-apply + ReLU prologue (in_scale / in_shift: the bn_act launch between the two convolutions of a
//     residual block disappears), split exactly into its three bf16 pieces and stored in LDS -- once per workgroup, not once per tap and wave;
//   * the arithmetic is conv_halo.hip's: six v_mfma_f32_32x32x16_bf16 products per k16 step on exactly split fp32 operands (2 500 / 6 = 417 TF/s
//     pipe instead of the 157 TF/s fp32 pipe); the three 2^-16-class products accumulate in their own register tile (two independent MFMA
//     chains per output tile, and the small terms are summed among themselves before they meet the large ones);
//   * weight fragments come STRAIGHT FROM GLOBAL MEMORY: the pre-split planes of pack modes 8 / 9 are k16-chunk-major, so the fragment of a
//     (tap, chunk) for 32 output channels is one contiguous 1 KB run -- one 16-byte load per lane and piece, three steps ahead in a register
//     ring, served by L1 / L2 (every workgroup reads the same <= 0.9 MB).  No weight staging, hence ONE barrier per 16-channel super-chunk
//     instead of one per tap, and the LDS pipe carries the activation fragments only;
//   * a wave = one (or two) 32-pixel x 32-channel MFMA tile(s); the four waves of a workgroup divide patch rows (WPX), output-channel tiles
//     (WCO) and -- where a layer has fewer than 1 024 tiles -- the input channels (KS: partial tiles meet in LDS before the epilogue), so
//     that every shape puts >= 256 workgroups x 4 waves on the chip;
//   * 39-58 KB of LDS and <= 128 VGPRs per workgroup: two workgroups per CU.
// Epilogue = conv_small.hip's (bias, affine, residual, ReLU, accumulate, train-mode statistics with statistic groups, BatchNorm finalize by
// the launch's last workgroup, first phase of a BatchNorm backward in data-gradient launches: bst_*), with conv_halo.hip's 16-byte stores.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lean_rne16(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ unsigned lean_pack_hi16(float a, float b) {      // (bf16 chop of b) << 16 | (bf16 chop of a)
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float lean_chop_rest(float x) { return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

// The part of the small-problem kernels behind their k-loops: the two accumulator chains of a tile are added, the input-channel slices of a tile meet in LDS
// (fixed order: bit-identical run to run), then bias / affine / residual / ReLU / accumulate, 16-byte stores, the train-mode statistics (or the first phase of a
// BatchNorm backward: bst_*) with ONE atomic per statistic, channel and workgroup, and the BatchNorm finalize by the launch's last workgroup.
// Wave layout: wave = wk + KS * (wco + WCO * wpx); e_row[i] / e_ok[i]: output row of this lane in tile i (lane & 31) and whether it exists.
template <int MT, int WPX, int WCO, int KS>
__device__ __forceinline__ void lean_finish(const mrfa_conv_params& p, f32x16 (&acc)[MT][2], unsigned char* smem, int scratch_doubles, int wave, int lane, int tid, int wk,
                                            bool wg_on, int n0, int cb, int fhalf, const long long (&e_row)[MT], const bool (&e_ok)[MT], int grp,
                                            const f32x4 (&pre_res)[MT][4], const f32x4 (&pre_bx)[MT][4]) {
    // ---- the two chains of a tile, then the input-channel slices of a tile (through LDS: the halo images are dead behind the barrier)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] += acc[i][1][r];
    if constexpr (KS > 1) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);    // [wave][tile][16][64]
        if (wk > 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wave * MT + i) * 16 + r) * 64 + lane] = acc[i][0][r];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int k = 1; k < KS; ++k)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][0][r] += red[(((wave + k) * MT + i) * 16 + r) * 64 + lane];
        }
    }

    // ------------------------------------------------------------------ epilogue (waves with wk == 0)
    // lane = (output row e_row[i] of tile i, half); accumulator quad g = channels cb + 8 g .. + 3 of that row
    float s1[16], s2[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) s1[k] = s2[k] = 0.f;
    const bool wave_on = wg_on && wk == 0;
    if (wave_on) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (e_ok[i]) {
                const long long m = e_row[i];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = cb + 8 * g;
                    if (c0 < p.Cout) {                      // (Cout % 4 == 0: whole quads)
                        float* dst = p.y + (size_t)m * p.ldy + c0;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][0][4 * g + e] * p.alpha;
                        if (p.bias) {
                            const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + c0);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += b4[e];
                        }
                        if (p.out_scale) {
                            const f32x4 o4 = *reinterpret_cast<const f32x4*>(p.out_scale + c0), h4 = *reinterpret_cast<const f32x4*>(p.out_shift + c0);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] * o4[e] + h4[e];
                        }
                        if (p.res) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += pre_res[i][g][e];
                        }
                        if (p.relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        if (p.accumulate) {
                            const f32x4 o4 = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += o4[e];
                        }
                        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                        if (p.bst_x) {
                            // first phase of the BatchNorm backward (mrfa_conv_params.bst_*): v = d(act(bn(x))); through the activation, then the two sums
                            const int gc = grp * p.Cout + c0;
                            const f32x4 xr = pre_bx[i][g];
                            const f32x4 bsc = *reinterpret_cast<const f32x4*>(p.bst_scale + gc), bsh = *reinterpret_cast<const f32x4*>(p.bst_shift + gc);
                            const f32x4 bme = *reinterpret_cast<const f32x4*>(p.bst_mean + gc), biv = *reinterpret_cast<const f32x4*>(p.bst_invstd + gc);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float du = (p.bst_relu && xr[e] * bsc[e] + bsh[e] <= 0.f) ? 0.f : v[e];
                                s1[4 * g + e] += du;
                                s2[4 * g + e] += du * ((xr[e] - bme[e]) * biv[e]);
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { s1[4 * g + e] += v[e]; s2[4 * g + e] += v[e] * v[e]; }
                        }
                    }
                }
            }
        }
    }
    if (p.stats) {
        if (wave_on) {
            // per-channel sums over the 32 pixel lanes of a half: butterfly reduce-scatter; afterwards lane L holds channel index
            // kk = 8 b4 + 4 b3 + 2 b2 + b1 (bN = bit N of L), lanes L and L ^ 1 the same total
            auto stage = [&](float (&v)[16], auto W) {
                constexpr int w = decltype(W)::value;
                const bool hi = (lane & (2 * w)) != 0;
#pragma unroll
                for (int k = 0; k < w; ++k) {
                    const float send = hi ? v[k] : v[k + w];
                    const float keep = hi ? v[k + w] : v[k];
                    v[k] = keep + __shfl_xor(send, 2 * w, 64);
                }
            };
            auto reduce16 = [&](float (&v)[16]) {
                stage(v, std::integral_constant<int, 8>{});
                stage(v, std::integral_constant<int, 4>{});
                stage(v, std::integral_constant<int, 2>{});
                stage(v, std::integral_constant<int, 1>{});
                v[0] += __shfl_xor(v[0], 1, 64);
            };
            reduce16(s1);
            reduce16(s2);
        }
        // the pixel tiles of a workgroup that share their output channels (WPX waves) meet in LDS: ONE atomic per statistic, channel and workgroup
        __shared__ float s_st[4][2][32];
        if (wk == 0) {
            const int kk = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
            const int ch = 4 * fhalf + 8 * (kk >> 2) + (kk & 3);
            if ((lane & 1) == 0) { s_st[wave][0][ch] = wave_on ? s1[0] : 0.f; s_st[wave][1][ch] = wave_on ? s2[0] : 0.f; }
        }
        __syncthreads();
        if (tid < 64 * WCO) {
            const int which = tid / (32 * WCO), col = tid % (32 * WCO), wc = col >> 5, ch = col & 31;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < WPX; ++k) t += s_st[KS * (wc + WCO * k)][which][ch];
            const int cch = n0 + col;
            if (wg_on && cch < p.Cout) atomicAdd(stat_slot(p, grp, blockIdx.x) + which * p.Cout + cch, (double)t);
        }
        // (every workgroup of the grid, with all of its threads; the halo images are dead: their LDS is the finalize's scratch)
        if (p.fin_scale) fused_bn_finalize(p, gridDim.x, (int)blockIdx.x, reinterpret_cast<double*>(smem), scratch_doubles);
    }
}

// TW: patch width (32: a pixel tile is one row of 32; 16: two rows of 16).  The four waves: WPX (pixel tiles) x WCO (32-channel tiles) x KS (input-
// channel slices); MT pixel tiles per wave.  NSC super-chunks of 16 KS input channels (Cin = 16 KS NSC), the whole k-loop unrolled.
template <int TW, int WPX, int WCO, int KS, int MT, int NSC>
struct LeanGeo {
    static_assert(WPX * WCO * KS == 4, "four waves");
    static constexpr int TR = 32 / TW;                 // rows of one pixel tile
    static constexpr int NPT = WPX * MT;               // pixel tiles of the patch
    static constexpr int PR = NPT * TR;                // patch rows
    static constexpr int HP = TW + 2;                  // halo row pitch (pixels)
    static constexpr int HPIX = (PR + 2) * HP;
    static constexpr int AHALF = (HPIX * 16 - 64 + 127) / 128 * 128 + 64;      // half-plane (k 0..7 | k 8..15) stride: the halves sit in different bank halves
    static constexpr int UPS = HPIX * 4 * KS;          // float4 units of one super-chunk
    static constexpr int NUS = (UPS + 255) / 256;      // ... per thread
    static constexpr int NCH = NSC * KS;               // 16-channel chunks
    static_assert(AHALF % 128 == 64, "half-plane stride");
    static constexpr size_t LDS_BYTES(int npc) {       // dynamic LDS of a launch: the halo images, or the slices' partial tiles behind them
        const size_t halo = (size_t)NCH * npc * 2 * AHALF, red = KS > 1 ? (size_t)4 * MT * 16 * 64 * 4 : 0;
        return halo > red ? halo : red;
    }
    static_assert(NUS <= 9, "the units of the next super-chunk are stored one per tap");
};

template <int TW, int WPX, int WCO, int KS, int MT, int NSC, bool PRO, int NP>
__global__ __launch_bounds__(256, MT == 2 ? 1 : 2) void conv_lean_kernel(const mrfa_conv_params p, const int tiles_n, const int tiles_x, const int tiles_y, const int total_tiles) {
    using G = LeanGeo<TW, WPX, WCO, KS, MT, NSC>;
    constexpr int NPC = NP == 6 ? 3 : (NP == 3 ? 2 : 1);      // bf16 pieces kept
    constexpr int APLANE = 2 * G::AHALF, CHB = NPC * APLANE;  // bytes of one piece plane / of one 16-channel chunk
    constexpr int HP = G::HP, NUS = G::NUS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [chunk][piece][half][halo pixel][16 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave % KS, wco = (wave / KS) % WCO, wpx = wave / (KS * WCO);
    const int per_xcd = (int)gridDim.x >> 3;
    const int lin = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);      // an XCD's L2 sees a contiguous run of patches
    const bool wg_on = lin < total_tiles;                // (a workgroup past the end still walks the code on clamped indices: fused_bn_finalize counts tickets)
    const int lin_c = wg_on ? lin : 0;
    const int tile_n = lin_c % tiles_n;
    int t_ = lin_c / tiles_n;
    const int tx = t_ % tiles_x;
    t_ /= tiles_x;
    const int ty = t_ % tiles_y;
    const int n_img = t_ / tiles_y;
    const int y0 = ty * G::PR, x0 = tx * TW, n0 = tile_n * (32 * WCO);

    const float* __restrict__ x = p.x;
    // ---- halo units of this thread inside a super-chunk: (halo pixel, channel quad q of the 16 KS channels)
    int a_goff[NUS], a_loff[NUS];
    bool a_inb[NUS], a_val[NUS];
#pragma unroll
    for (int j = 0; j < NUS; ++j) {
        const int u = tid + j * 256;
        a_val[j] = u < G::UPS;
        const int uu = a_val[j] ? u : 0;
        const int q = uu % (4 * KS), hp = uu / (4 * KS);
        const int hy = hp / HP, hx = hp - hy * HP;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        a_inb[j] = a_val[j] && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
        // out-of-image pixels read a valid address (pixel 0 of the image) and are zeroed after the prologue: no load inside a branch
        const int pix = n_img * p.Hin * p.Win + (a_inb[j] ? iy * p.Win + ix : 0);
        a_goff[j] = pix * p.ldx + q * 4;
        const int q4 = q & 3;
        a_loff[j] = (q >> 2) * CHB + (q4 >> 1) * G::AHALF + hp * 16 + (q4 & 1) * 8;
    }
    f32x4 ra[NSC][NUS];
    auto load_sc = [&](auto SC) {
        constexpr int sc = decltype(SC)::value;
#pragma unroll
        for (int j = 0; j < NUS; ++j) ra[sc][j] = *reinterpret_cast<const f32x4*>(x + (size_t)a_goff[j] + sc * (16 * KS));
    };
    auto store_unit = [&](auto SC, int j) {
        constexpr int sc = decltype(SC)::value;
        f32x4 v = ra[sc][j];
        if constexpr (PRO) {                           // BatchNorm apply + ReLU of the producing layer (in_relu always comes with in_scale)
            // (statistic groups: the vectors are [groups][Cin], the patch lies in one image)
            const int c0 = (p.groups > 1 ? n_img / (p.N / p.groups) * p.Cin : 0) + sc * (16 * KS) + ((tid + j * 256) % (4 * KS)) * 4;
            const f32x4 psc = *reinterpret_cast<const f32x4*>(p.in_scale + c0), psh = *reinterpret_cast<const f32x4*>(p.in_shift + c0);
            v = v * psc + psh;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        v = a_inb[j] ? v : z;
        u32x2 p1, p2, p3;
        if constexpr (NP == 1) {
            p1[0] = lean_rne16(v.x) | (lean_rne16(v.y) << 16);
            p1[1] = lean_rne16(v.z) | (lean_rne16(v.w) << 16);
        } else {
            const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float a = xs[2 * h], b = xs[2 * h + 1];
                p1[h] = lean_pack_hi16(a, b);
                const float ar = lean_chop_rest(a), br = lean_chop_rest(b);
                p2[h] = lean_pack_hi16(ar, br);
                p3[h] = lean_pack_hi16(lean_chop_rest(ar), lean_chop_rest(br));
            }
        }
        if (a_val[j]) {
            unsigned char* dst = smem + sc * (KS * CHB) + a_loff[j];
            *reinterpret_cast<u32x2*>(dst) = p1;
            if constexpr (NPC >= 2) *reinterpret_cast<u32x2*>(dst + APLANE) = p2;
            if constexpr (NPC == 3) *reinterpret_cast<u32x2*>(dst + 2 * APLANE) = p3;
        }
    };

    // ---- weight fragments: lane (row = lane & 31, k half = lane >> 5) of the (tap, chunk) slab = 16 bytes at row * 32 + half * 16 of a contiguous run
    const int frow = lane & 31, fhalf = lane >> 5;
    const unsigned short* __restrict__ ws = reinterpret_cast<const unsigned short*>(p.w_split) + (size_t)(n0 + wco * 32 + frow) * 16 + fhalf * 8;
    const int w_tap = (int)p.w_tap, chunk_stride = p.w_rows * 16;
    const int w_piece = (int)p.w_piece;
    constexpr int STEPS = 9 * NSC, RING = 3;
    u32x4 rb[RING][NPC];
    auto load_b = [&](int step, int slot) {             // step = 9 * sc + tap: the chunk of this wave in super-chunk sc
        const int s = step < STEPS ? step : STEPS - 1;  // (past the end: a re-read nobody uses)
        const int sc = s / 9, tap = s - 9 * sc;
        const unsigned short* src = ws + (size_t)tap * w_tap + (size_t)(sc * KS + wk) * chunk_stride;
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) rb[slot][pc] = *reinterpret_cast<const u32x4*>(src + (size_t)pc * w_piece);
    };

    // issue order = the order the data is needed in (vmcnt retires in order): halo of super-chunk 0, the first weight fragments, the other halos
    load_sc(std::integral_constant<int, 0>{});
#pragma unroll
    for (int s = 0; s < RING; ++s) load_b(s, s);
    if constexpr (NSC > 1) load_sc(std::integral_constant<int, 1>{});
    if constexpr (NSC > 2) load_sc(std::integral_constant<int, 2>{});
    if constexpr (NSC > 3) load_sc(std::integral_constant<int, 3>{});
    static_assert(NSC <= 4, "halo register file: at most four super-chunks");

    f32x16 acc[MT][2];                                  // [..][0]: the three leading products, [..][1]: the three 2^-16-class products
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

    // A fragment of pixel tile (wpx * MT + i), tap (r, s), chunk ch: lane pixel j = lane & 31 -> halo slot (row + r) * HP + col + s
    const int j_row = TW == 32 ? 0 : (frow >> 4), j_col = TW == 32 ? frow : (frow & 15);
    const int a_frag = wk * CHB + fhalf * G::AHALF + ((wpx * MT * G::TR + j_row) * HP + j_col) * 16;

#pragma unroll
    for (int j = 0; j < NUS; ++j) store_unit(std::integral_constant<int, 0>{}, j);
    __syncthreads();

    bf16x8 af[2][MT][NPC];
    auto read_a = [&](int step, int buf) {
        const int s = step < STEPS ? step : STEPS - 1;
        const int sc = s / 9, tap = s - 9 * sc;
        const int r = tap / 3, c = tap - 3 * r;
        const unsigned char* A = smem + sc * (KS * CHB) + a_frag + (r * HP + c) * 16;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) af[buf][i][pc] = *reinterpret_cast<const bf16x8*>(A + pc * APLANE + i * (G::TR * HP * 16));
    };
    // six products (weights piece PB x activation piece PA), smallest first; t < 3 -> acc[..][1]
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
    auto mfma_step = [&](int buf, int slot) {
        bf16x8 b[NPC];
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) b[pc] = __builtin_bit_cast(bf16x8, rb[slot][pc]);
#pragma unroll
        for (int t = 6 - NP; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < MT; ++i)
                // D = W * X^T: rows = output channels, columns = pixels (a lane ends up with 4 consecutive channels of one pixel per accumulator quad)
                acc[i][t < 3 ? 1 : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[PB[t]], af[buf][i][PA[t]], acc[i][t < 3 ? 1 : 0], 0, 0, 0);
    };

    // epilogue operands with a row per output pixel (the residual; the BatchNorm input of bst_*): fetched at the head of the LAST super-chunk, so that the
    // ~1 us of their round trip lies under its MFMAs instead of behind them.  Lane = (pixel j_row / j_col of tile i, channel quad cb + 8 g): clamped addresses.
    const long long Mtot = (long long)p.N * p.Hout * p.Wout;
    const int cb = n0 + wco * 32 + 4 * fhalf;
    long long e_m[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int pyl = y0 + (wpx * MT + i) * G::TR + j_row;
        e_m[i] = ((long long)n_img * p.Hout + (pyl < p.Hout ? pyl : p.Hout - 1)) * p.Wout + x0 + j_col;
    }
    f32x4 pre_res[MT][4], pre_bx[MT][4];
    auto prefetch_epilogue = [&]() {
        if (p.res) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = cb + 8 * g < p.Cout ? cb + 8 * g : p.Cout - 4;
                    pre_res[i][g] = *reinterpret_cast<const f32x4*>(p.res + (size_t)e_m[i] * p.ldr + c0);
                }
        }
        if (p.bst_x) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = cb + 8 * g < p.Cout ? cb + 8 * g : p.Cout - 4;
                    pre_bx[i][g] = *reinterpret_cast<const f32x4*>(p.bst_x + (size_t)e_m[i] * p.bst_ldx + c0);
                }
        }
    };

    read_a(0, 0);
    auto run_sc = [&](auto SC) {
        constexpr int sc = decltype(SC)::value;
        if constexpr (sc == NSC - 1) prefetch_epilogue();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int step = 9 * sc + tap;
            // the next step's activation fragments (inside a super-chunk: its LDS image is complete; across the barrier they are read behind it)
            if (tap < 8) read_a(step + 1, (step + 1) & 1);
            mfma_step(step & 1, step % RING);
            load_b(step + RING, step % RING);
            if constexpr (sc + 1 < NSC) {              // the next super-chunk's halo: split and stored in the shadow of this one's MFMAs
                if (tap < NUS) store_unit(std::integral_constant<int, sc + 1>{}, tap);
            }
        }
        if constexpr (sc + 1 < NSC) {
            __syncthreads();
            read_a(9 * (sc + 1), (9 * (sc + 1)) & 1);
        }
    };
    run_sc(std::integral_constant<int, 0>{});
    if constexpr (NSC > 1) run_sc(std::integral_constant<int, 1>{});
    if constexpr (NSC > 2) run_sc(std::integral_constant<int, 2>{});
    if constexpr (NSC > 3) run_sc(std::integral_constant<int, 3>{});

    // ---- partial tiles -> output, statistics, finalize (lean_finish below: shared with the 1x1 / linear kernel)
    long long e_row[MT];
    bool e_ok[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int pyl = y0 + (wpx * MT + i) * G::TR + j_row;
        e_ok[i] = pyl < p.Hout;
        e_row[i] = ((long long)n_img * p.Hout + pyl) * p.Wout + x0 + j_col;
    }
    const int grp = stat_group(p, (long long)n_img * p.Hout * p.Wout, Mtot);      // (a patch lies in one image)
    lean_finish<MT, WPX, WCO, KS>(p, acc, smem, (int)(G::LDS_BYTES(NPC) / 8), wave, lane, tid, wk, wg_on, n0, cb, fhalf, e_row, e_ok, grp, pre_res, pre_bx);
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// The same arithmetic for 1x1 convolutions / linears (the token transformer's 192 <-> 576 linears over 16 x 276 token rows, HRNet's 1x1 fuse and
// bottleneck layers): K = Cin up to 640 does not fit the "whole halo in registers and LDS" scheme above, so the K axis is PIPELINED: a workgroup owns
// 64 output rows x 32 WCO output channels; per stage the four waves stage 32 input channels per input-channel slice (KS slices: waves = WCO x KS) of
// their 64 rows -- loads one stage ahead in registers, split, two LDS buffers, one barrier per stage -- and every wave multiplies TWO 32-row tiles by one
// 32-channel weight fragment per k16 step (fragments straight from the pre-split planes, one stage ahead in a second register set).  conv_small.hip ran
// these at 17-39 TF/s on the fp32 pipe (29 us for the 192 -> 576 linear); same epilogue as above (lean_finish).
template <int WCO, int KS, int NP>
__global__ __launch_bounds__(256, 2) void gemm_lean_kernel(const mrfa_conv_params p, const long long M, const int tiles_n, const int total_tiles) {
    static_assert(WCO * KS == 4, "four waves");
    constexpr int MT = 2;
    constexpr int NPC = NP == 6 ? 3 : (NP == 3 ? 2 : 1);
    constexpr int AHALF = 64 * 16 + 64;                        // half-plane (k 0..7 | k 8..15) of 64 rows; +64 B: the halves sit in different bank halves
    constexpr int APLANE = 2 * AHALF, CHB = NPC * APLANE;      // one piece plane / one 16-channel chunk
    constexpr int STAGE = KS * 2 * CHB;                        // one stage buffer: KS slices x two k16 chunks
    constexpr int NU = KS * 2;                                 // float4 units per thread and stage (64 rows x 8 KS channel quads)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [2 buffers][slice][k16 chunk][piece][half][row][16 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave % KS, wco = wave / KS;
    const int per_xcd = (int)gridDim.x >> 3;
    const int lin = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const bool wg_on = lin < total_tiles;
    const int lin_c = wg_on ? lin : 0;
    const int tile_n = lin_c % tiles_n;
    const long long m0 = (long long)(lin_c / tiles_n) * 64;
    const int n0 = tile_n * (32 * WCO);
    const int NS = p.Cin / (32 * KS);                          // stages: slice wk covers channels [wk * 32 NS, (wk + 1) * 32 NS)

    // ---- staging units: unit u = tid + 256 j -> (row = u / (8 KS), slice, channel quad of its 32 channels)
    size_t a_goff[NU];
    int a_loff[NU];
    bool a_ok[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        const int u = tid + j * 256;
        const int row = u / (8 * KS), r = u - row * (8 * KS), sl = r >> 3, q = r & 7;
        a_ok[j] = m0 + row < M;
        a_goff[j] = (size_t)(a_ok[j] ? m0 + row : 0) * p.ldx + (size_t)sl * 32 * NS + q * 4;
        a_loff[j] = sl * (2 * CHB) + (q >> 2) * CHB + ((q & 3) >> 1) * AHALF + row * 16 + (q & 1) * 8;
    }
    f32x4 ra[NU];
    auto load_stage = [&](int st) {
        const int s = st < NS ? st : NS - 1;                   // (past the end: a re-read nobody uses)
#pragma unroll
        for (int j = 0; j < NU; ++j) ra[j] = *reinterpret_cast<const f32x4*>(p.x + a_goff[j] + (size_t)s * 32);
    };
    auto store_stage = [&](int buf) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const f32x4 v = a_ok[j] ? ra[j] : z;
            u32x2 p1, p2, p3;
            if constexpr (NP == 1) {
                p1[0] = lean_rne16(v.x) | (lean_rne16(v.y) << 16);
                p1[1] = lean_rne16(v.z) | (lean_rne16(v.w) << 16);
            } else {
                const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float a = xs[2 * h], b = xs[2 * h + 1];
                    p1[h] = lean_pack_hi16(a, b);
                    const float ar = lean_chop_rest(a), br = lean_chop_rest(b);
                    p2[h] = lean_pack_hi16(ar, br);
                    p3[h] = lean_pack_hi16(lean_chop_rest(ar), lean_chop_rest(br));
                }
            }
            unsigned char* dst = smem + buf * STAGE + a_loff[j];
            *reinterpret_cast<u32x2*>(dst) = p1;
            if constexpr (NPC >= 2) *reinterpret_cast<u32x2*>(dst + APLANE) = p2;
            if constexpr (NPC == 3) *reinterpret_cast<u32x2*>(dst + 2 * APLANE) = p3;
        }
    };

    // ---- weight fragments of this wave: rows n0 + 32 wco .., chunks of slice wk; two k16 steps per stage, one stage ahead (two register sets)
    const int frow = lane & 31, fhalf = lane >> 5;
    const unsigned short* __restrict__ ws = reinterpret_cast<const unsigned short*>(p.w_split) + (size_t)(n0 + wco * 32 + frow) * 16 + fhalf * 8 +
                                            (size_t)(wk * 2 * NS) * ((size_t)p.w_rows * 16);
    const size_t chunk_stride = (size_t)p.w_rows * 16;
    const long long w_piece = p.w_piece;
    u32x4 rb[2][2][NPC];
    auto load_b = [&](int st, auto SET) {
        constexpr int set = decltype(SET)::value;
        const int s = st < NS ? st : NS - 1;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) rb[set][h][pc] = *reinterpret_cast<const u32x4*>(ws + (size_t)(2 * s + h) * chunk_stride + (size_t)pc * w_piece);
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
    const int a_frag = wk * (2 * CHB) + fhalf * AHALF + frow * 16;        // + i * 32 * 16 per row tile, + h * CHB per k16 step
    auto compute = [&](int buf, auto SET) {
        constexpr int set = decltype(SET)::value;
        const unsigned char* A = smem + buf * STAGE + a_frag;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 af[MT][NPC], b[NPC];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) af[i][pc] = *reinterpret_cast<const bf16x8*>(A + h * CHB + pc * APLANE + i * (32 * 16));
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) b[pc] = __builtin_bit_cast(bf16x8, rb[set][h][pc]);
#pragma unroll
            for (int t = 6 - NP; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    acc[i][t < 3 ? 1 : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[PB[t]], af[i][PA[t]], acc[i][t < 3 ? 1 : 0], 0, 0, 0);
        }
    };

    // epilogue operands with a row per output row: fetched before the last stage
    const int cb = n0 + wco * 32 + 4 * fhalf;
    long long e_row[MT];
    bool e_ok[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        e_row[i] = m0 + 32 * i + frow;
        e_ok[i] = e_row[i] < M;
    }
    f32x4 pre_res[MT][4], pre_bx[MT][4];
    auto prefetch_epilogue = [&]() {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = cb + 8 * g < p.Cout ? cb + 8 * g : p.Cout - 4;
                const long long mr = e_ok[i] ? e_row[i] : 0;
                if (p.res) pre_res[i][g] = *reinterpret_cast<const f32x4*>(p.res + (size_t)mr * p.ldr + c0);
                if (p.bst_x) pre_bx[i][g] = *reinterpret_cast<const f32x4*>(p.bst_x + (size_t)mr * p.bst_ldx + c0);
            }
    };

    // ---- pipeline: stage s in buffer s & 1; registers hold stage s + 1 (activations) and the weight set (s + 1) & 1
    load_stage(0);
    load_b(0, std::integral_constant<int, 0>{});
    store_stage(0);
    load_stage(1);
    __syncthreads();
    for (int s = 0; s < NS; s += 2) {
        // even stage s: weight set 0, buffer 0
        load_b(s + 1, std::integral_constant<int, 1>{});
        if (s + 1 < NS) store_stage(1);                        // (buffer 1 was last read as stage s - 1: barrier behind it)
        load_stage(s + 2);
        if (s + 1 >= NS) prefetch_epilogue();
        compute(0, std::integral_constant<int, 0>{});
        __syncthreads();
        if (s + 1 < NS) {
            load_b(s + 2, std::integral_constant<int, 0>{});
            if (s + 2 < NS) store_stage(0);
            load_stage(s + 3);
            if (s + 2 >= NS) prefetch_epilogue();
            compute(1, std::integral_constant<int, 1>{});
            __syncthreads();
        }
    }
    const int grp = stat_group(p, m0, M);                      // (the launcher keeps a workgroup's 64 rows inside one statistic group)
    lean_finish<MT, 1, WCO, KS>(p, acc, smem, (int)((size_t)2 * STAGE / 8), wave, lane, tid, wk, wg_on, n0, cb, fhalf, e_row, e_ok, grp, pre_res, pre_bx);
}

int g_lean_on = -1;              // -1: not initialised (MRFA_CONV_LEAN)
int g_lean_min_wgs = 128;
int g_lean_geo = -1;             // >= 0: only this entry of LEAN_CFGS (tests)

struct LeanCfg { int TW, WPX, WCO, KS, MT, NSC; };

// the instantiated geometries, in order of preference (fewest halo re-reads first); a geometry applies when Cin = 16 KS NSC, the output tiles into
// its patches and the launch has at least g_lean_min_wgs workgroups
constexpr LeanCfg LEAN_CFGS[] = {
    // two pixel tiles per wave: a weight fragment (16 bytes per lane from L1) feeds twelve MFMAs -- with one tile per wave the four SIMDs of a CU ask
    // the L1 for 3 KB per 192 matrix cycles each = its whole 64 bytes per clock
    {32, 4, 1, 1, 2, 2},      // 32 channels @64^2, >= 16 frames: 8 x 32 patches
    {32, 1, 2, 2, 2, 2},      // 64 channels @32^2, >= 16 frames: 2 x 32 patches x 64 output channels, the input channels in two slices
    {16, 1, 1, 4, 2, 2},      // 128 channels @16^2, >= 16 frames: 4 x 16 patches x 32 output channels, four slices
    // one pixel tile per wave (smaller batches: half the tiles)
    {32, 4, 1, 1, 1, 2},      // 32 channels @64^2: 4 x 32 patches
    {32, 2, 2, 1, 1, 4},      // 64 channels @32^2: 2 x 32 patches x 64 output channels
    {32, 1, 2, 2, 1, 2},      // 64 channels @32^2: 1 x 32 patches, two slices
    {16, 1, 2, 2, 1, 4},      // 128 channels @16^2: 2 x 16 patches x 64 output channels, two slices
    {16, 1, 1, 4, 1, 2},      // 128 channels @16^2: four slices
};
constexpr int N_LEAN_CFGS = sizeof(LEAN_CFGS) / sizeof(LEAN_CFGS[0]);
static_assert(N_LEAN_CFGS == 8, "mrfa_conv_lean_launch switches over the geometries");

long long lean_wgs(const mrfa_conv_params& p, const LeanCfg& c) {
    const int pr = c.WPX * c.MT * (32 / c.TW);
    return (long long)p.N * cdiv(p.Hout, pr) * (p.Wout / c.TW) * cdiv(p.Cout, 32 * c.WCO);
}

int lean_pick(const mrfa_conv_params& p) {
    for (int i = 0; i < N_LEAN_CFGS; ++i) {
        const LeanCfg& c = LEAN_CFGS[i];
        if (g_lean_geo >= 0 && i != g_lean_geo) continue;
        if (p.Cin != 16 * c.KS * c.NSC || (p.Wout % c.TW) != 0) continue;
        // two-tile geometries run one workgroup per CU: they need twice the workgroups
        if (g_lean_geo < 0 && lean_wgs(p, c) < (long long)g_lean_min_wgs * c.MT) continue;
        return i;
    }
    return -1;
}

bool lean_on() {
    if (g_lean_on < 0) { const char* e = getenv("MRFA_CONV_LEAN"); g_lean_on = !(e && e[0] == '0'); }
    return g_lean_on != 0;
}

template <int I, bool PRO, int NP>
int lean_launch_cfg(hipStream_t st, const mrfa_conv_params& p) {
    constexpr LeanCfg c = LEAN_CFGS[I];
    constexpr int npc = NP == 6 ? 3 : (NP == 3 ? 2 : 1);
    const int pr = c.WPX * c.MT * (32 / c.TW);
    const int tiles_n = cdiv(p.Cout, 32 * c.WCO), tiles_x = p.Wout / c.TW, tiles_y = cdiv(p.Hout, pr);
    const long long total = (long long)p.N * tiles_y * tiles_x * tiles_n;
    dim3 grid((unsigned)(cdiv(total, 8) * 8));
    constexpr size_t lds = LeanGeo<c.TW, c.WPX, c.WCO, c.KS, c.MT, c.NSC>::LDS_BYTES(npc);
    if constexpr (lds > 65536) {                    // (more than 64 KB of dynamic LDS must be asked for once per kernel)
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_lean_kernel<c.TW, c.WPX, c.WCO, c.KS, c.MT, c.NSC, PRO, NP>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (attr != hipSuccess) { mrfa_set_error("conv2d(lean): %zu bytes of LDS refused: %s", lds, hipGetErrorString(attr)); return 2; }
    }
    hipLaunchKernelGGL((conv_lean_kernel<c.TW, c.WPX, c.WCO, c.KS, c.MT, c.NSC, PRO, NP>), grid, dim3(256), lds, st, p, tiles_n, tiles_x, tiles_y, (int)total);
    return 0;
}

template <int I>
int lean_launch_i(hipStream_t st, const mrfa_conv_params& p, int mode) {
    const bool pro = p.in_scale != nullptr;
    if (mode == 3) return pro ? lean_launch_cfg<I, true, 1>(st, p) : lean_launch_cfg<I, false, 1>(st, p);
    if (mode == 2) return pro ? lean_launch_cfg<I, true, 3>(st, p) : lean_launch_cfg<I, false, 3>(st, p);
    return pro ? lean_launch_cfg<I, true, 6>(st, p) : lean_launch_cfg<I, false, 6>(st, p);
}

}  // namespace

int mrfa_tuning_conv_lean(int set) {
    const int prev = lean_on();
    if (set >= 0) g_lean_on = set != 0;
    return prev;
}
int mrfa_tuning_conv_lean_geo(int set) {
    const int prev = g_lean_geo;
    g_lean_geo = set;
    return prev;
}
int mrfa_tuning_conv_lean_min(int set) {
    const int prev = g_lean_min_wgs;
    if (set >= 0) g_lean_min_wgs = set;
    return prev;
}

// 1: the shape runs here.  3x3 / pad 1 / stride 1 in a split-operand (or plain bf16) mode with pre-split weights, 16-byte addressable tensors,
// whole channel quads, one of the instantiated geometries, and a problem small enough that conv_halo.hip's 8-row patches cannot fill the chip
bool mrfa_conv_lean_eligible(const mrfa_conv_params& p) {
    const int mode = mrfa_get_mfma_mode();
    if (!lean_on() || (mode != 1 && mode != 2 && mode != 3)) return false;
    if (p.kflat > 0 || p.R != 3 || p.S != 3 || p.pad != 1 || !p.w_split || p.nbatch > 1 || p.splitk > 1 || p.tile || p.stride > 1 || p.stride < 0 || p.ups || p.mask) return false;
    if (mode != 3 && p.w_piece <= 0) return false;
    if (p.Hout != p.Hin || p.Wout != p.Win) return false;
    if ((p.Cout % 32) != 0 || p.Cout > 128 || (p.Cin % 32) != 0) return false;
    if ((p.ldy % 4) != 0 || !aligned16(p.y) || (p.ldx % 4) != 0 || !aligned16(p.x)) return false;
    if (p.res && ((p.ldr % 4) != 0 || !aligned16(p.res))) return false;
    if (p.bias && !aligned16(p.bias)) return false;
    if (p.out_scale && (!aligned16(p.out_scale) || !aligned16(p.out_shift))) return false;
    if (p.in_scale && (!p.in_relu || !aligned16(p.in_scale) || !aligned16(p.in_shift))) return false;
    if (p.bst_x && ((p.bst_ldx % 4) != 0 || !aligned16(p.bst_x) || !aligned16(p.bst_scale) || !aligned16(p.bst_shift) || !aligned16(p.bst_mean) || !aligned16(p.bst_invstd)))
        return false;
    if (p.groups > 1 && (p.N % p.groups) != 0) return false;
    if (3 * p.w_piece >= (1ll << 31) || 9 * p.w_tap >= (1ll << 31) || (long long)p.N * p.Hin * p.Win * p.ldx >= (1ll << 31)) return false;
    // ~2.5 GFLOP at most: beyond that the 8-row patches of conv_halo.hip (weights shared through LDS by 8 waves) are the faster kernel
    if (2.0 * (double)p.N * p.Hout * p.Wout * p.Cout * 9.0 * p.Cin > 2.6e9) return false;
    return lean_pick(p) >= 0;
}

// ---- 1x1 convolutions / linears on gemm_lean_kernel
namespace {
int g_gemm_lean_on = -1;
bool gemm_lean_on() {
    if (g_gemm_lean_on < 0) { const char* e = getenv("MRFA_GEMM_LEAN"); g_gemm_lean_on = e ? atoi(e) : 1; }      // 0 off, 1 where measured faster, 2 wherever it can run
    return g_gemm_lean_on != 0;
}
// 0: 64 rows x 128 channels per workgroup; 1: 64 x 64 with the input channels in two slices (short N or long K: more workgroups, half the k-loop)
int gemm_lean_cfg(const mrfa_conv_params& p, long long M) {
    const long long rows = (M + 63) / 64;
    const bool can_b = (p.Cin % 64) == 0;
    if (!can_b) return 0;
    const long long wa = rows * cdiv(p.Cout, 128);
    return (p.Cout >= 128 && wa >= 256) ? 0 : 1;
}
template <int WCO, int KS>
int gemm_lean_launch_cfg(hipStream_t st, const mrfa_conv_params& p, long long M, int mode) {
    const int tiles_n = cdiv(p.Cout, 32 * WCO);
    const long long total = ((M + 63) / 64) * tiles_n;
    dim3 grid((unsigned)(cdiv(total, 8) * 8));
#define GL(NP_)                                                                                                                        \
    do {                                                                                                                               \
        constexpr int npc = NP_ == 6 ? 3 : (NP_ == 3 ? 2 : 1);                                                                          \
        constexpr size_t st_bytes = (size_t)2 * KS * 2 * npc * 2 * (64 * 16 + 64), red = KS > 1 ? (size_t)4 * 2 * 16 * 64 * 4 : 0;       \
        constexpr size_t lds = st_bytes > red ? st_bytes : red;                                                                        \
        hipLaunchKernelGGL((gemm_lean_kernel<WCO, KS, NP_>), grid, dim3(256), lds, st, p, M, tiles_n, (int)total);                     \
    } while (0)
    if (mode == 3) GL(1); else if (mode == 2) GL(3); else GL(6);
#undef GL
    return 0;
}
}  // namespace

int mrfa_tuning_gemm_lean(int set) {
    gemm_lean_on();
    const int prev = g_gemm_lean_on;
    if (set >= 0) g_gemm_lean_on = set > 2 ? 2 : set;
    return prev;
}

// 1: the 1x1 convolution / linear runs on gemm_lean_kernel: a split-operand (or plain bf16) mode with pre-split weights, 32-aligned channel counts, 16-byte
// addressable tensors, no prologue, and a problem small enough that the 128-row tiles of conv_split.hip cannot fill the chip with long k-loops
bool mrfa_gemm_lean_eligible(const mrfa_conv_params& p, long long M) {
    const int mode = mrfa_get_mfma_mode();
    if (!gemm_lean_on() || (mode != 1 && mode != 2 && mode != 3)) return false;
    if (p.kflat > 0 || p.R != 1 || p.S != 1 || p.pad != 0 || !p.w_split || p.nbatch > 1 || p.splitk > 1 || p.tile || p.stride > 1 || p.stride < 0 || p.ups || p.mask || p.in_scale) return false;
    if (mode != 3 && p.w_piece <= 0) return false;
    if (p.Hout != p.Hin || p.Wout != p.Win) return false;
    if ((p.Cout % 32) != 0 || (p.Cin % 32) != 0 || p.Cin > 1024 || M < 256) return false;
    if ((p.ldy % 4) != 0 || !aligned16(p.y) || (p.ldx % 4) != 0 || !aligned16(p.x)) return false;
    if (p.res && ((p.ldr % 4) != 0 || !aligned16(p.res))) return false;
    if (p.bias && !aligned16(p.bias)) return false;
    if (p.out_scale && (!aligned16(p.out_scale) || !aligned16(p.out_shift))) return false;
    if (p.bst_x && ((p.bst_ldx % 4) != 0 || !aligned16(p.bst_x) || !aligned16(p.bst_scale) || !aligned16(p.bst_shift) || !aligned16(p.bst_mean) || !aligned16(p.bst_invstd)))
        return false;
    if (p.groups > 1 && ((p.N % p.groups) != 0 || ((M / p.groups) % 64) != 0)) return false;      // a workgroup's 64 rows inside one statistic group
    if (3 * p.w_piece >= (1ll << 31) || M * p.ldx >= (1ll << 31) * 2) return false;
    if (2.0 * (double)M * p.Cout * (double)p.Cin > 2.6e9) return false;
    if (g_gemm_lean_on >= 2) return true;
    // measured (profiles/r6_gemm_lean_bench.txt): ahead on the transformer's token linears (4 416 rows, 192 / 576 channels: 17 against 25-27 us); behind the
    // 128-row tiles on the 65 536-row layer1 bottlenecks and behind conv_small's 32-row tiles where 64-row tiles leave most of the chip idle (the fuse layers)
    const long long wgs = ((M + 63) / 64) * cdiv(p.Cout, gemm_lean_cfg(p, M) == 0 ? 128 : 64);
    return wgs >= 192 && M <= 32768 && p.Cout >= 64;
}

int mrfa_gemm_lean_launch(hipStream_t st, const mrfa_conv_params& p, long long M) {
    const int mode = mrfa_get_mfma_mode();
    const int rc = gemm_lean_cfg(p, M) == 0 ? gemm_lean_launch_cfg<4, 1>(st, p, M, mode) : gemm_lean_launch_cfg<2, 2>(st, p, M, mode);
    if (rc) return rc;
    MRFA_CHECK_LAUNCH("mrfa_conv2d_nhwc(gemm_lean)");
    return 0;
}

int mrfa_conv_lean_launch(hipStream_t st, const mrfa_conv_params& p) {
    const int mode = mrfa_get_mfma_mode();
    const int i = lean_pick(p);
    int rc = 0;
    switch (i) {
        case 0: rc = lean_launch_i<0>(st, p, mode); break;
        case 1: rc = lean_launch_i<1>(st, p, mode); break;
        case 2: rc = lean_launch_i<2>(st, p, mode); break;
        case 3: rc = lean_launch_i<3>(st, p, mode); break;
        case 4: rc = lean_launch_i<4>(st, p, mode); break;
        case 5: rc = lean_launch_i<5>(st, p, mode); break;
        case 6: rc = lean_launch_i<6>(st, p, mode); break;
        case 7: rc = lean_launch_i<7>(st, p, mode); break;
        default: mrfa_set_error("conv2d(lean): no geometry"); return 1;
    }
    if (rc) return rc;
    MRFA_CHECK_LAUNCH("mrfa_conv2d_nhwc(lean)");
    return 0;
}
