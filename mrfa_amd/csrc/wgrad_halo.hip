// Weight gradient of the 3x3 stride-1 "same" convolutions on the bf16 matrix pipe with exactly split fp32 operands (the
// arithmetic of wgrad_split.hip), ALL NINE TAPS from one staging of the operands:
//
//   dW[tap (r,s)][co][ci] += alpha * sum_{n,y,x} dY[n,y,x][co] * X'[n, y+r-1, x+s-1][ci]          X' = prologue(ups(x)), zero outside
//
// Why: wgrad_bf16x6_kernel computes one tap per workgroup, so every activation and every output gradient is loaded, transposed in
// registers, split into its three bf16 pieces and stored to LDS nine times per (co, ci) tile: SQ_INSTS_VALU / MFMA = 6.1
// (profiles/r2_pmc_SQ_*), well above the ~5 instructions a wave can issue for free per 32-cycle MFMA.
//
// Here a workgroup (8 waves) owns a 32-pixel-wide column segment of one image, one (BCO x BCI) block of the weights and all nine
// taps, and walks down the segment one 32-pixel row strip at a time:
//   * per strip ONE new row of X' (34 pixels x BCI channels) and one row of dY (32 x BCO) are loaded (full 128-byte lines),
//     prologue'd, split and stored to LDS -- X' rows live in a 4-slot ring (rows y-1, y, y+1 in use, y+2 arriving), dY in two
//     buffers; every staged element feeds 9 (X') / 9 x BCI/32 ... MFMA tiles instead of one.
//   * LDS images stay PIXEL-major ([32-channel block][pixel][64 B]), i.e. exactly what the coalesced loads produce, and the
//     K-major MFMA fragments (8 consecutive pixels of one channel per lane) are formed by ds_read_b64_tr_b16, the gfx950
//     transposing LDS read: within a 16-lane group lane l supplies the address of (row l / 4, columns 4 (l % 4) ..) of a
//     [4 pixel][16 channel] block and receives column l of it (probed: tools/ubench/probe_tr.hip).  A tap (r, s) is then a ring
//     slot (r) plus a constant 64 * s byte offset -- no re-staging, no unaligned access.
//   * wave (c, b) accumulates the nine 32 (co) x 32 (ci) tap tiles of co block c and ci block b: 144 accumulator registers, one
//     dY fragment per k16 step serves all nine taps.  Per k16 step and wave: 60 transposing reads + 54 MFMAs.
//   * the accumulators leave through fp32 atomics into dW[tap][Cout][Cin] once per workgroup (pixel-range split as in
//     wgrad_split.hip), the bias gradient from the staged dY values.
// Used when the conv is 3x3 / pad 1 / stride 1 with Wout % 32 == 0 and Cin % 32 == 0; everything else stays on wgrad_split.hip.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4_t;

constexpr int NT = 512;
constexpr int XPIX = 34;                       // 32 pixels of the strip + one halo pixel on either side
constexpr int XBLK = XPIX * 64;                // one 32-channel block of an X' row: [34 pixels][32 bf16]
constexpr int DYBLK = 32 * 64;                 // one 32-channel block of a dY row

__device__ __forceinline__ unsigned pack_hi16(float a, float b) {
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float chop_rest(float x) { return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
__device__ __forceinline__ void split3(const f32x4 v, u32x2& p1, u32x2& p2, u32x2& p3) {
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float a = x[2 * q], b = x[2 * q + 1];
        p1[q] = pack_hi16(a, b);
        const float ra = chop_rest(a), rb = chop_rest(b);
        p2[q] = pack_hi16(ra, rb);
        p3[q] = pack_hi16(chop_rest(ra), chop_rest(rb));
    }
}

__device__ __forceinline__ unsigned rne16(float x) {          // plain bf16 (NP = 1): round to nearest even
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void round1(const f32x4 v, u32x2& p1) {
    p1[0] = rne16(v.x) | (rne16(v.y) << 16);
    p1[1] = rne16(v.z) | (rne16(v.w) << 16);
}

// transposing fragment read: rows (pixels) k0 .. k0+7 of this lane's channel column as one MFMA operand (two b64 reads)
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p) {
    typedef __attribute__((address_space(3))) bf16x4_t* lds4;
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(p));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(p + 4 * 64));
    bf16x8 r;
    __builtin_memcpy(&r, &lo, 8);
    __builtin_memcpy(reinterpret_cast<char*>(&r) + 8, &hi, 8);
    return r;
}

// NBO x NBI x TS = 8 waves: BCO = 32 NBO output channels x BCI = 32 NBI input channels per workgroup; TS = 2: the nine taps of a
// (co, ci) block are split over two waves (taps 0..4 / 5..8) -- the 64 x 64 block shape for layers with <= 64 output channels (a
// 64 x 128 block with all nine taps per wave needed 3 staging slots next to its 144 accumulators: 49 spilled registers, MFMA busy 29 %)
// PH: the weight gradient of a fused nearest-x2 upsample + 3x3 layer (ups = 1) in PHASE form (see conv_halo.hip): blockIdx.y = phase (py, px) of the
// output grid; the workgroup walks the LOW-resolution grid, dY is the phase image dY[2y + py][2x + px] (a stride-2 gather), X' the plain input, and
// only the four taps (py + a, px + b), a, b in {0, 1}, of the 3x3 neighbourhood are computed: 16 instead of 36 MACs per low-resolution pixel and
// channel pair.  The four accumulators are then added to every 3x3 tap (r, s) that reads the same source pixel: rows {0} | {1,2} for py = 0,
// {0,1} | {2} for py = 1, columns likewise -- the transpose of pack mode 12's weight sums.  No weights involved: same sums, different order.
template <int NBO, int NBI, int TS, bool PRO, int NP, bool PH = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_halo_kernel(const mrfa_wgrad_params p, const int tiles_ci,
                                                                                               const int ntiles, const int HS, const int segs_y,
                                                                                               const int tiles_x, const int total) {
    static_assert(NBO * NBI * TS == 8 && (TS == 1 || TS == 2), "8 waves");
    constexpr int NTAP = PH ? (TS == 1 ? 4 : 2) : (TS == 1 ? 9 : 5);            // accumulator tiles per wave
    constexpr int NPC = NP == 6 ? 3 : (NP == 3 ? 2 : 1);
    constexpr int XROW = NPC * NBI * XBLK;         // one ring slot: [piece][block][34 pixels][64 B]
    constexpr int DYROW = NPC * NBO * DYBLK;       // one dY buffer:  [piece][block][32 pixels][64 B]
    constexpr int NDY = NBO * 256 / NT;            // float4 units per thread: dY row (32 pixels x NBO x 8 quads)
    constexpr int NXU = NBI * XPIX * 8;            // units of an X' row
    constexpr int NX = (NXU + NT - 1) / NT;
    static_assert(NBO * 256 % NT == 0, "dY units fill whole rounds");
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * XROW + 2 * DYROW];
    unsigned char* const smX = smem;
    unsigned char* const smD = smem + 4 * XROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave / (NBO * NBI);               // tap half (TS = 2)
    const int wc = (wave / NBI) % NBO, wb = wave % NBI;
    const int per_xcd = (int)gridDim.x >> 3;
    const int lin = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);      // the tiles of one segment share an XCD's L2
    if (lin >= total) return;
    const int tile = lin % ntiles;
    int sg = lin / ntiles;
    const int ys = sg % segs_y;
    sg /= segs_y;
    const int xt = sg % tiles_x;
    const int n_img = sg / tiles_x;
    const int tile_co = tile / tiles_ci, tile_ci = tile - tile_co * tiles_ci;
    const int co0 = tile_co * (32 * NBO), ci0 = tile_ci * (32 * NBI);
    const int ph_y = PH ? (int)(blockIdx.y >> 1) : 0, ph_x = PH ? (int)(blockIdx.y & 1) : 0;
    const int Hgrid = PH ? p.Hin : p.Hout;           // rows of the grid the workgroup walks (PH: the low-resolution one)
    const int y0 = ys * HS, y1 = min(Hgrid, y0 + HS), x0 = xt * 32;

    const float* __restrict__ x = p.x;
    const float* __restrict__ dy = p.dy;
    const int ush = PH ? 0 : p.ups;
    const int Hv = p.Hin << ush, Wv = p.Win << ush;

    // ---- staging units of this thread.  dY: unit u = blk * 256 + px * 8 + quad;  X': u = blk * 272 + hp * 8 + quad
    int d_goff[NDY], d_loff[NDY], d_nval[NDY];
#pragma unroll
    for (int j = 0; j < NDY; ++j) {
        const int u = tid + j * NT;
        const int blk = u >> 8, px = (u >> 3) & 31, quad = u & 7;
        const int ch = co0 + blk * 32 + quad * 4;
        d_nval[j] = min(max(p.Cout - ch, 0), 4);                  // ragged Cout (126): the quad's tail channels belong to somebody else
        d_goff[j] = ((n_img * p.Hout) * p.Wout + (PH ? 2 * (x0 + px) + ph_x : x0 + px)) * p.ldy + (d_nval[j] > 0 ? ch : co0);
        d_loff[j] = blk * DYBLK + px * 64 + quad * 8;
    }
    int x_goff[NX], x_loff[NX];
    bool x_ok[NX], x_val[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int u = tid + j * NT;
        x_val[j] = u < NXU;
        const int uu = x_val[j] ? u : 0;
        const int blk = uu / (XPIX * 8), rem = uu - blk * (XPIX * 8);
        const int hp = rem >> 3, quad = rem & 7;
        const int ix = x0 - 1 + hp;
        const int ch = ci0 + blk * 32 + quad * 4;
        x_ok[j] = x_val[j] && (unsigned)ix < (unsigned)Wv && ch < p.Cin;          // (Cin % 32 == 0: whole quads)
        x_goff[j] = (n_img * p.Hin * p.Win + (x_ok[j] ? (ix >> ush) : 0)) * p.ldx + (x_ok[j] ? ch : ci0);
        x_loff[j] = blk * XBLK + hp * 64 + quad * 8;
    }
    f32x4 psc[NX], psh[NX];
    if constexpr (PRO) {
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int ch = x_goff[j] - (x_goff[j] / p.ldx) * p.ldx;            // channel of this unit (valid address even when masked)
            psc[j] = *reinterpret_cast<const f32x4*>(p.in_scale + ch);
            psh[j] = *reinterpret_cast<const f32x4*>(p.in_shift + ch);
        }
    }

    f32x4 rd[NDY], rx[NX];
    f32x4 bsum[NDY];
#pragma unroll
    for (int j = 0; j < NDY; ++j) bsum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = (p.dbias != nullptr) && tile_ci == 0;

    // rows: iy = input (virtual) row of X', oy = output row of dY; out-of-image X' rows are zero, loads of rows outside the image or the
    // segment read a clamped (valid) row and are masked / never used
    auto load_x = [&](int iy) {
        const int yc = min(max(iy, 0), Hv - 1) >> ush;
#pragma unroll
        for (int j = 0; j < NX; ++j) rx[j] = *reinterpret_cast<const f32x4*>(x + (size_t)x_goff[j] + (size_t)yc * p.Win * p.ldx);
    };
    auto store_x = [&](int iy) {
        const bool row_ok = (unsigned)iy < (unsigned)Hv;
        unsigned char* dst = smX + ((iy + 1) & 3) * XROW;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            f32x4 v = rx[j];
            if constexpr (PRO) {
                v = v * psc[j] + psh[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            v = (row_ok && x_ok[j]) ? v : z;
            u32x2 p1, p2, p3;
            if constexpr (NP == 1) round1(v, p1); else split3(v, p1, p2, p3);
            if (x_val[j]) {
                *reinterpret_cast<u32x2*>(dst + x_loff[j]) = p1;
                if constexpr (NPC >= 2) *reinterpret_cast<u32x2*>(dst + NBI * XBLK + x_loff[j]) = p2;
                if constexpr (NPC == 3) *reinterpret_cast<u32x2*>(dst + 2 * NBI * XBLK + x_loff[j]) = p3;
            }
        }
    };
    auto load_d = [&](int oy) {
        const int yg = min(oy, Hgrid - 1);
        const int yc = PH ? 2 * yg + ph_y : yg;
#pragma unroll
        for (int j = 0; j < NDY; ++j) rd[j] = *reinterpret_cast<const f32x4*>(dy + (size_t)d_goff[j] + (size_t)yc * p.Wout * p.ldy);
    };
    auto store_d = [&](int oy) {
        const bool row_ok = oy < y1;
        unsigned char* dst = smD + (oy & 1) * DYROW;
#pragma unroll
        for (int j = 0; j < NDY; ++j) {
            f32x4 v = rd[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (row_ok && e < d_nval[j]) ? v[e] : 0.f;
            bsum[j] += v;
            u32x2 p1, p2, p3;
            if constexpr (NP == 1) round1(v, p1); else split3(v, p1, p2, p3);
            *reinterpret_cast<u32x2*>(dst + d_loff[j]) = p1;
            if constexpr (NPC >= 2) *reinterpret_cast<u32x2*>(dst + NBO * DYBLK + d_loff[j]) = p2;
            if constexpr (NPC == 3) *reinterpret_cast<u32x2*>(dst + 2 * NBO * DYBLK + d_loff[j]) = p3;
        }
    };

    f32x16 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposing-read lane offset: 16-lane group g = lane >> 4: columns 16 (g & 1) .., pixel rows 8 (g >> 1) ..; lane l of the group
    // supplies the address of (row l / 4, columns 4 (l % 4) ..)
    const int l16 = lane & 15, grp = lane >> 4;
    const int frag_off = (8 * (grp >> 1) + (l16 >> 2)) * 64 + (grp & 1) * 32 + (l16 & 3) * 8;
    const int a_off = wc * DYBLK + frag_off;
    const int b_off = wb * XBLK + frag_off;

    // taps T0 .. T1-1 of this wave, in groups of <= 3 whose MFMAs interleave (no MFMA waits for its predecessor on the same accumulator)
    auto compute_taps = [&](int oy, auto T0_, auto T1_) {
        constexpr int T0 = decltype(T0_)::value, T1 = decltype(T1_)::value;
        const unsigned char* D = smD + (oy & 1) * DYROW + a_off;
        const unsigned char* XR[3] = {smX + ((oy + 0) & 3) * XROW + b_off,       // slot of input row oy - 1  (slot = (iy + 1) & 3)
                                      smX + ((oy + 1) & 3) * XROW + b_off, smX + ((oy + 2) & 3) * XROW + b_off};
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
        constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[NPC];
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) a[pc] = tr_frag(D + pc * NBO * DYBLK + kk * 16 * 64);
#pragma unroll
            for (int g0 = T0; g0 < T1; g0 += 3) {
                constexpr int GMAX = 3;
                bf16x8 b[NPC][GMAX];
#pragma unroll
                for (int q = 0; q < GMAX; ++q) {
                    const int tap = g0 + q;
                    if (tap < T1) {
#pragma unroll
                        for (int pc = 0; pc < NPC; ++pc) b[pc][q] = tr_frag(XR[tap / 3] + pc * NBI * XBLK + (kk * 16 + tap % 3) * 64);
                    }
                }
#pragma unroll
                for (int t = 6 - NP; t < 6; ++t)
#pragma unroll
                    for (int q = 0; q < GMAX; ++q)
                        if (g0 + q < T1)
                            acc[g0 + q - T0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]], b[PB[t]][q], acc[g0 + q - T0], 0, 0, 0);
            }
        }
    };
    // PH: the four taps (a, b) of this phase: X' row oy - 1 + ph_y + a = ring slot (oy + ph_y + a) & 3, pixel offset ph_x + b
    auto compute_phase = [&](int oy, auto Q0_, auto Q1_) {
        constexpr int Q0 = decltype(Q0_)::value, Q1 = decltype(Q1_)::value, NQ = Q1 - Q0;
        const unsigned char* D = smD + (oy & 1) * DYROW + a_off;
        const unsigned char* XR[2] = {smX + ((oy + ph_y) & 3) * XROW + b_off + ph_x * 64, smX + ((oy + ph_y + 1) & 3) * XROW + b_off + ph_x * 64};
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
        constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[NPC], b[NPC][NQ];
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) a[pc] = tr_frag(D + pc * NBO * DYBLK + kk * 16 * 64);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) b[pc][q] = tr_frag(XR[(Q0 + q) >> 1] + pc * NBI * XBLK + (kk * 16 + ((Q0 + q) & 1)) * 64);
#pragma unroll
            for (int t = 6 - NP; t < 6; ++t)
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]], b[PB[t]][q], acc[q], 0, 0, 0);
        }
    };
    auto compute = [&](int oy) {
        if constexpr (PH) {
            if constexpr (TS == 1) {
                compute_phase(oy, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
            } else {
                if (wt == 0) compute_phase(oy, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
                else compute_phase(oy, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
            }
        } else if constexpr (TS == 1) {
            compute_taps(oy, std::integral_constant<int, 0>{}, std::integral_constant<int, 9>{});
        } else {
            if (wt == 0) compute_taps(oy, std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
            else compute_taps(oy, std::integral_constant<int, 5>{}, std::integral_constant<int, 9>{});
        }
    };

    // ---- prologue: X' rows y0-1, y0, y0+1 and dY row y0 into LDS; X' row y0+2 and dY row y0+1 into registers
    load_x(y0 - 1);
    store_x(y0 - 1);
    load_x(y0);
    store_x(y0);
    load_x(y0 + 1);
    load_d(y0);
    store_x(y0 + 1);
    store_d(y0);
    load_x(y0 + 2);
    load_d(y0 + 1);
    __syncthreads();
    // ---- strips.  Strip oy reads ring slots of rows oy-1, oy, oy+1 and dY buffer oy & 1; meanwhile row oy+2 / dY row oy+1 (registers, loaded one
    // strip ago) go to the slot of row oy-2 / the other dY buffer (both last read in strip oy-1: barrier), and the next loads are issued.
    for (int oy = y0; oy < y1; ++oy) {
        store_x(oy + 2);
        store_d(oy + 1);
        load_x(oy + 3);
        load_d(oy + 2);
        compute(oy);
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue: atomics into dW[tap][Cout][Cin]
    float* __restrict__ dw = p.dw;
    const int ci = ci0 + wb * 32 + (lane & 31);
    const int half = lane >> 5;
    if constexpr (PH) {
        if (ci < p.Cin) {
            constexpr unsigned RM[4] = {0x1u, 0x6u, 0x3u, 0x4u};       // [phase bit * 2 + tap bit] -> mask over the 3x3 rows / columns that read that source pixel
            const int q0 = TS == 1 ? 0 : 2 * wt;
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                const int q = q0 + t;
                const unsigned rm = RM[ph_y * 2 + (q >> 1)], sm = RM[ph_x * 2 + (q & 1)];
#pragma unroll 1
                for (int t9 = 0; t9 < 9; ++t9) {             // (not unrolled: 4 x 9 x 16 guarded atomics with hoisted addresses spilled 100 registers)
                    const int r3 = t9 / 3, s3 = t9 - 3 * r3;
                        if (((rm >> r3) & 1u) && ((sm >> s3) & 1u)) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int co = co0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                                if (co < p.Cout) atomicAdd(dw + ((size_t)t9 * p.Cout + co) * p.Cin + ci, acc[t][r] * p.alpha);
                            }
                        }
                }
            }
        }
    } else if (ci < p.Cin) {
        const int tap0 = TS == 1 ? 0 : 5 * wt, ntap = TS == 1 ? 9 : (wt == 0 ? 5 : 4);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            if (t < ntap) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (co < p.Cout) atomicAdd(dw + ((size_t)(tap0 + t) * p.Cout + co) * p.Cin + ci, acc[t][r] * p.alpha);
                }
            }
        }
    }
    if (do_bias) {
        // bsum[j]: this thread's channel quad (blk, quad) over its pixel column px = (tid >> 3) & 31: reduce over px (lane bits 3..5, then
        // the waves through LDS), one atomic per channel
        float* red = reinterpret_cast<float*>(smem);          // the staging buffers are dead (all waves are past the last barrier)
        __syncthreads();
        for (int i = tid; i < 32 * NBO; i += NT) red[i] = 0.f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NDY; ++j) {
            f32x4 v = bsum[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = v[e];
                s += __shfl_xor(s, 8, 64);
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                v[e] = s;
            }
            if ((lane >> 3) == 0) {
                const int u = tid + j * NT;
                const int c = (u >> 8) * 32 + (u & 7) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(red + c + e, v[e]);
            }
        }
        __syncthreads();
        for (int i = tid; i < 32 * NBO; i += NT)
            if (co0 + i < p.Cout) atomicAdd(p.dbias + co0 + i, red[i]);        // (alpha scales dW only)
    }
}

int g_wgrad_halo_on = -1;
int g_wgrad_halo_min = 192;
int g_wgrad_halo_target = 256;      // workgroups per round (one per CU)
int g_wgrad_halo_phase = 1;         // weight gradient of fused-upsample layers in phase form

}  // namespace

int mrfa_tuning_wgrad_halo(int set) {
    if (g_wgrad_halo_on < 0) { const char* e = getenv("MRFA_WGRAD_HALO"); g_wgrad_halo_on = !(e && e[0] == '0'); }
    const int prev = g_wgrad_halo_on;
    if (set >= 0) g_wgrad_halo_on = set != 0;
    return prev;
}

int mrfa_tuning_wgrad_halo_min(int set) {
    const int prev = g_wgrad_halo_min;
    if (set >= 0) g_wgrad_halo_min = set;
    return prev;
}

int mrfa_tuning_wgrad_halo_phase(int set) {
    const int prev = g_wgrad_halo_phase;
    if (set >= 0) g_wgrad_halo_phase = set != 0;
    return prev;
}

int mrfa_tuning_wgrad_halo_target(int set) {
    const int prev = g_wgrad_halo_target;
    if (set > 0) g_wgrad_halo_target = set;
    return prev;
}

// fused-upsample layer whose LOW-resolution grid tiles into 32-pixel strips: phase form (wgrad_halo_kernel<..., PH = true>)
static bool wgrad_halo_phase(const mrfa_wgrad_params& p) {
    return g_wgrad_halo_phase && p.ups == 1 && !p.in_scale && (p.Win % 32) == 0 && p.Hin >= 8;
}

static void wgrad_halo_config(const mrfa_wgrad_params& p, int& NBO, int& HS, int& segs_y, long long& total) {
    // block shape: 128 (co) x 64 (ci), or 64 x 64 with the taps split over two waves when Cout <= 64 (half of a 128-row block would idle)
    NBO = p.Cout <= 64 ? 2 : 4;
    const int NBI = 2;
    const bool ph = wgrad_halo_phase(p);
    const int Hg = ph ? p.Hin : p.Hout, Wg = ph ? p.Win : p.Wout;        // the grid a workgroup walks
    const int ntiles = cdiv(p.Cout, 32 * NBO) * cdiv(p.Cin, 32 * NBI) * (ph ? 4 : 1);       // (x 4 phase workgroups per segment and block)
    const long long cols = (long long)p.N * (Wg / 32);
    // segment height: every segment pays ~6 row-strips of fixed cost (three-row prologue, 144 atomics per lane at the end) and the launch
    // runs in rounds of 256 workgroups (one per CU): minimise rounds x (HS + 6) over the power-of-two divisions of the column
    // (measured on 192->128 @256^2: 384 workgroups = 1.5 rounds 209 TF/s, 768 = 3 rounds 228; 128->128 @128^2: 256 workgroups 188, 512: 154)
    HS = Hg;
    double best = 1e30;
    for (int hs = Hg; hs >= 8; hs /= 2) {
        const long long tot = cols * ntiles * cdiv(Hg, hs);
        const double cost = (double)((tot + g_wgrad_halo_target - 1) / g_wgrad_halo_target) * (hs + 6);
        if (cost < best) { best = cost; HS = hs; }
        if (hs % 2) break;
    }
    segs_y = cdiv(Hg, HS);
    total = cols * segs_y * ntiles;                  // (phase form: counts the four phase workgroups)
}

bool mrfa_wgrad_halo_eligible(const mrfa_wgrad_params& p) {
    const int mode = mrfa_get_mfma_mode();
    if (!mrfa_tuning_wgrad_halo(-1) || (mode != 1 && mode != 2 && mode != 3)) return false;
    if (p.kflat > 0 || p.R != 3 || p.S != 3 || p.pad != 1 || p.nbatch > 1 || p.ksplit > 0) return false;
    if ((p.Wout % 32) != 0 || (p.Cin % 32) != 0 || p.Cout < 32 || p.Hout < 8) return false;
    if (p.Hout != (p.Hin << p.ups) || p.Wout != (p.Win << p.ups)) return false;
    if ((p.ldx % 4) != 0 || !aligned16(p.x) || (p.ldy % 4) != 0 || !aligned16(p.dy)) return false;
    if ((long long)p.N * p.Hin * p.Win * p.ldx >= (1ll << 31) || (long long)p.N * p.Hout * p.Wout * p.ldy >= (1ll << 31)) return false;
    if (p.in_scale && (!p.in_relu || !aligned16(p.in_scale) || !aligned16(p.in_shift))) return false;
    int NBO, HS, segs_y;
    long long total;
    wgrad_halo_config(p, NBO, HS, segs_y, total);
    return total >= g_wgrad_halo_min;                  // too few workgroups: the per-tap kernel splits the pixel range finer
}

int mrfa_wgrad_halo_launch(hipStream_t st, const mrfa_wgrad_params& p) {
    int NBO, HS, segs_y;
    long long total;
    wgrad_halo_config(p, NBO, HS, segs_y, total);
    const int NBI = 2;
    const int tiles_ci = cdiv(p.Cin, 32 * NBI), ntiles = cdiv(p.Cout, 32 * NBO) * tiles_ci;
    const bool ph = wgrad_halo_phase(p);
    if (ph) total /= 4;                              // per phase (blockIdx.y)
    dim3 grid((unsigned)(cdiv(total, 8) * 8), ph ? 4u : 1u);
    const int tiles_x = (ph ? p.Win : p.Wout) / 32;
    const bool three = mrfa_get_mfma_mode() == 2, one = mrfa_get_mfma_mode() == 3;
    const bool pro = p.in_scale != nullptr;
#define WH(NBO_, NBI_, TS_, PRO_, PH_)                                                                                                                     \
    do {                                                                                                                                              \
        if (one) hipLaunchKernelGGL((wgrad_halo_kernel<NBO_, NBI_, TS_, PRO_, 1, PH_>), grid, dim3(NT), 0, st, p, tiles_ci, ntiles, HS, segs_y, tiles_x, (int)total); \
        else if (three) hipLaunchKernelGGL((wgrad_halo_kernel<NBO_, NBI_, TS_, PRO_, 3, PH_>), grid, dim3(NT), 0, st, p, tiles_ci, ntiles, HS, segs_y, tiles_x, (int)total); \
        else hipLaunchKernelGGL((wgrad_halo_kernel<NBO_, NBI_, TS_, PRO_, 6, PH_>), grid, dim3(NT), 0, st, p, tiles_ci, ntiles, HS, segs_y, tiles_x, (int)total);      \
    } while (0)
    if (ph) { if (NBO == 4) WH(4, 2, 1, false, true); else WH(2, 2, 2, false, true); }
    else if (NBO == 4) { if (pro) WH(4, 2, 1, true, false); else WH(4, 2, 1, false, false); }
    else { if (pro) WH(2, 2, 2, true, false); else WH(2, 2, 2, false, false); }
#undef WH
    MRFA_CHECK_LAUNCH("mrfa_conv2d_wgrad_nhwc(halo)");
    return 0;
}
