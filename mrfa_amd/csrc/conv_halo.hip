// 3x3 "same" convolution on the bf16 matrix pipe with exactly split fp32 operands (the arithmetic of conv_split.hip), tiled over
// 2-D PATCHES of the output so that an input pixel is loaded, run through the fused prologue and split into its three bf16
// pieces ONCE per 16-channel chunk -- not once per tap as in the row-tiled implicit GEMM of conv_split.hip.
//
// Why: PMC on conv_bf16x6_kernel (profiles/r2_pmc_SQ_*): 3.7 non-MFMA VALU per MFMA, MFMA pipe busy 48 %.  Per 32x32x16 block
// the row-tiled kernel re-loads and re-splits the activation tile for each of the nine taps (5.5 VALU per element per tap) and
// writes 3 bf16 planes of it to LDS nine times.  One wave64 issues ~5 other instructions "for free" per 32-cycle MFMA
// (MI355X_MICROARCH.md, per-instruction table), so that work adds to the matrix time instead of hiding behind it.
//
// Here a workgroup owns PR x 32 output pixels of one image (PR = 8: 8 waves, or 4: 4 waves) x BN output channels:
//   * the (PR+2) x 34 input HALO of a 16-channel chunk is staged in LDS once (global fp32 -> prologue -> split -> three bf16
//     planes), double buffered across chunks, and serves all nine taps: a tap is a constant byte offset of the fragment read
//     ((r * 34 + s) * 16), i.e. an immediate of ds_read_b128.  A-side VALU / LDS-store / global-load work per MFMA drops by
//     9 * (PR * 32) / ((PR + 2) * 34) = 6.8x (PR = 8).
//   * the weights arrive pre-split (pack modes 8 / 9, as for conv_split.hip) and are copied per (chunk, tap) slab: 16-byte
//     loads one step ahead into registers, ds_write_b128 at the start of the next step, two slab buffers, ONE barrier per tap.
//   * LDS images are half-planes [k 0..7 | k 8..15][pixel or row][16 B]: the 16 lanes of a ds_read_b128 lane group read 16
//     consecutive-mod-16 pixels -> 16 distinct 16-byte bank slots, no swizzle, tap offsets stay additive.
// Per tap and wave: 12 ds_read_b128 + 24 MFMA + ~1.5 weight loads / stores + ~10 amortised halo instructions.
// Epilogue = conv_split.hip's (bias, eval-BN affine, residual, ReLU, accumulate, train-BN statistics).
// Used for the stride-1 3x3 layers with Wout % 32 == 0 and enough patches to fill the chip; everything else stays on
// conv_split.hip / conv_mfma.hip.  Also runs the data gradient (same conv, flipped / transposed pack).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int PW = 32;                             // patch width = the 32 rows of one MFMA tile
constexpr int HP = PW + 2;                         // halo row pitch (pixels)
template <int BN>
struct GeoB {
    static constexpr int BROWS = BN < 128 ? 128 : BN;
    static constexpr int BHALF = BROWS * 16 + 64;  // +64 B: the two halves of a row land in different bank halves (ds_write_b128 groups)
    static constexpr int BPLANE = 2 * BHALF;
    static constexpr int BSLAB = 3 * BPLANE;
};

// plain bf16 operands (NP = 1): round-to-nearest-even of the fp32 value, one product -- the arithmetic of a bf16 autocast
__device__ __forceinline__ unsigned rne16(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void round1(const f32x4 v, u32x2& p1) {
    p1[0] = rne16(v.x) | (rne16(v.y) << 16);
    p1[1] = rne16(v.z) | (rne16(v.w) << 16);
}

template <int PR>
struct Geo {
    static constexpr int NT = PR * 64;             // (PR / 2) x 2 waves, each 64 pixels (2 patch rows) x BN / 2 channels
    static constexpr int HPIX = (PR + 2) * HP;
    static constexpr int AHALF = HPIX * 16;        // bytes of one half-plane
    static constexpr int APLANE = 2 * AHALF;
    static constexpr int ABUF = 3 * APLANE;
    static constexpr int NU = (HPIX * 4 + NT - 1) / NT;      // float4 units (pixel, channel quad) per thread and chunk
    static_assert(AHALF % 128 == 64, "half-plane stride must put the k 8..15 half into the other bank half");
    static_assert(NU <= 4, "halo units are loaded at taps 0,2,4,6 and stored two taps later");
};

__device__ __forceinline__ unsigned pack_hi16(float a, float b) {      // (bf16 chop of b) << 16 | (bf16 chop of a)
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

// PH: "phase" form of a fused nearest-x2 upsample (UpBlock2d, util.py:172-176): blockIdx.y = output phase (py, px); the patch lives on the
// LOW-resolution input grid, every output pixel (2y + py, 2x + px) reads the 2x2 source pixels {y - 1 + py, y + py} x {x - 1 + px, x + px}
// with the pre-summed phase weights of pack mode 12: four taps per chunk instead of nine on four times fewer patch pixels = 16 / 36 of the MACs
// MODE 2 (PD): the DATA GRADIENT of such a layer, also in phase form: dx[y][x] (low resolution) = sum over the four phases of a 2x2
// transposed convolution of the phase image dY[2t + py][2u + px]; the K loop runs over (chunk, phase) pairs -- each stages the halo of its
// phase image (a stride-2 gather of the high-resolution gradient) and runs four taps with the transposed phase weights (pack mode 13).
// Replaces the 3x3 data gradient on the high-resolution grid + the 2x2 sum-pooling pass.
template <int PR, bool PRO, int BN, int NP, int MODE>
__global__ __launch_bounds__(PR * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_halo_kernel(const mrfa_conv_params p, const int tiles_n, const int tiles_x, const int tiles_y,
                                                              const int total_tiles) {
    using G = Geo<PR>;
    constexpr int BHALF = GeoB<BN>::BHALF, BPLANE = GeoB<BN>::BPLANE, BSLAB = GeoB<BN>::BSLAB;
    constexpr int NT = G::NT;
    constexpr int TN = BN / 64;                    // 32-column MFMA tiles per wave along N
    constexpr int NPC = NP == 6 ? 3 : (NP == 3 ? 2 : 1);     // bf16 pieces needed (bf16x3 drops the third; plain bf16: one rounded plane)
    constexpr int BUNITS = NPC * BN * 2;           // 16-byte units of one weight slab
    constexpr int NBU = (BUNITS + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G::ABUF + 2 * BSLAB];
    unsigned char* const smA = smem;
    unsigned char* const smB = smem + 2 * G::ABUF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int per_xcd = (int)gridDim.x >> 3;
    const int lin = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);      // an XCD's L2 sees a contiguous run of patches
    if (lin >= total_tiles) return;
    const int tile_n = lin % tiles_n;
    int t_ = lin / tiles_n;
    const int tx = t_ % tiles_x;
    t_ /= tiles_x;
    const int ty = t_ % tiles_y;
    const int n_img = t_ / tiles_y;
    const int y0 = ty * PR, x0 = tx * PW, n0 = tile_n * BN;

    constexpr bool PH = MODE == 1, PD = MODE == 2;
    constexpr int KT = (PH || PD) ? 4 : 9;                 // taps (= pipeline steps) per 16-channel chunk
    static_assert(!(PH || PD) || G::NU <= 3, "phase form: the halo units are loaded at tap 0 and stored at taps 1..3");
    const int ph_y = PH ? (int)(blockIdx.y >> 1) : 0, ph_x = PH ? (int)(blockIdx.y & 1) : 0;
    const int ush = (PH || PD) ? 0 : p.ups;        // phase forms: the patch is ON the low-resolution grid
    const float* __restrict__ x = p.x;
    const unsigned short* __restrict__ ws = reinterpret_cast<const unsigned short*>((PH || PD) ? p.w_phase : p.w_split) +
                                            (PH ? (size_t)(blockIdx.y * 4) * (size_t)p.w_tap : (size_t)0);
    // PD: x is the high-resolution gradient (Hin x Win = 2 Hout x 2 Wout), the halo lives on the output (low-resolution) grid
    const int Hv = PD ? p.Hout : (p.Hin << ush), Wv = PD ? p.Wout : (p.Win << ush);
    const int NC = PD ? (p.Cin >> 4) * 4 : (p.Cin >> 4);         // PD: virtual chunks vc = 4 * (16-channel chunk) + phase

    // ---- halo units of this thread: (halo pixel hp, channel quad q); q = tid & 3 for every unit (NT % 4 == 0)
    const int q4 = tid & 3;
    int a_goff[G::NU], a_loff[G::NU];
    bool a_inb[G::NU], a_val[G::NU];
#pragma unroll
    for (int j = 0; j < G::NU; ++j) {
        const int u = tid + j * NT;
        a_val[j] = u < G::HPIX * 4;
        const int hp = (a_val[j] ? u : 0) >> 2;
        const int hy = hp / HP, hx = hp - hy * HP;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        a_inb[j] = a_val[j] && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
        // out-of-image pixels read a valid address (pixel 0 of the image) and are zeroed after the prologue
        const int pix = n_img * p.Hin * p.Win + (a_inb[j] ? (PD ? (2 * iy) * p.Win + 2 * ix : (iy >> ush) * p.Win + (ix >> ush)) : 0);
        a_goff[j] = pix * p.ldx + q4 * 4;
        a_loff[j] = (q4 >> 1) * G::AHALF + hp * 16 + (q4 & 1) * 8;
    }
    // ---- weight-slab units of this thread: (piece, row, half)
    int b_goff[NBU], b_loff[NBU];
    bool b_val[NBU];
#pragma unroll
    for (int j = 0; j < NBU; ++j) {
        const int u = tid + j * NT;
        b_val[j] = u < BUNITS;
        const int uu = b_val[j] ? u : 0;
        const int pc = uu / (BN * 2), rem = uu - pc * (BN * 2);
        const int row = rem >> 1, half = rem & 1;
        // pre-split planes are k16-chunk-major (pack_multi.hip chunk_major()): the slab of (tap, chunk) = w_rows consecutive rows of 32 bytes,
        // ONE contiguous run (row-major planes gave 32 bytes out of every row: a quarter of each 128-byte line fetched from L2; +3-5 %)
        b_goff[j] = pc * (int)((PH || PD) ? p.w_phase_piece : p.w_piece) + (n0 + row) * 16 + half * 8;
        b_loff[j] = pc * BPLANE + half * BHALF + row * 16;
    }

    f32x4 ra[G::NU], psc, psh;
    u32x4 rb[NBU];
    const int w_tap = (int)p.w_tap;
    const int chunk_stride = p.w_rows * 16;        // elements between the k16 chunks of one tap

    // PD: virtual chunk c = 4 * (channel chunk) + phase: phase image offset (py * Win + px) pixels, masked units stay on their valid address
    auto load_a = [&](int j, int c) {
        if constexpr (PD) {
            const int po = a_inb[j] ? (((c >> 1) & 1) * p.Win + (c & 1)) * p.ldx : 0;
            ra[j] = *reinterpret_cast<const f32x4*>(x + (size_t)a_goff[j] + po + (c >> 2) * 16);
        } else {
            ra[j] = *reinterpret_cast<const f32x4*>(x + (size_t)a_goff[j] + c * 16);
        }
    };
    auto load_pro = [&](int c) {
        if constexpr (PRO) {
            psc = *reinterpret_cast<const f32x4*>(p.in_scale + c * 16 + q4 * 4);
            psh = *reinterpret_cast<const f32x4*>(p.in_shift + c * 16 + q4 * 4);
        }
    };
    auto store_a = [&](int j, int buf) {
        f32x4 v = ra[j];
        if constexpr (PRO) {                       // fused pre-activation BN + ReLU (in_relu is always set with in_scale)
            v = v * psc + psh;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        v = a_inb[j] ? v : z;
        u32x2 p1, p2, p3;
        if constexpr (NP == 1) round1(v, p1); else split3(v, p1, p2, p3);
        if (a_val[j]) {
            unsigned char* dst = smA + buf * G::ABUF + a_loff[j];
            *reinterpret_cast<u32x2*>(dst) = p1;
            if constexpr (NPC >= 2) *reinterpret_cast<u32x2*>(dst + G::APLANE) = p2;
            if constexpr (NPC == 3) *reinterpret_cast<u32x2*>(dst + 2 * G::APLANE) = p3;
        }
    };
    auto load_b = [&](int c, int tap) {
        const unsigned short* src = PD ? ws + (size_t)((c & 3) * 4 + tap) * w_tap + (size_t)(c >> 2) * chunk_stride : ws + (size_t)tap * w_tap + (size_t)c * chunk_stride;
#pragma unroll
        for (int j = 0; j < NBU; ++j) rb[j] = *reinterpret_cast<const u32x4*>(src + b_goff[j]);
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NBU; ++j)
            if (b_val[j]) *reinterpret_cast<u32x4*>(smB + buf * BSLAB + b_loff[j]) = rb[j];
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, fhalf = lane >> 5;
    const int a_frag = fhalf * G::AHALF + ((2 * wm + ph_y) * HP + frow + ph_x) * 16;      // + ((i + r) * HP + s) * 16 per (row tile, tap)
    const int b_frag = fhalf * BHALF + (wn * (TN * 32) + frow) * 16;            // + j * 512 per column tile

    auto compute = [&](auto TAP, int abuf, int bbuf, int vc) {
        constexpr int tap = decltype(TAP)::value;
        // PD: source of tap (a, b) of phase (py, px) is the phase-image pixel (y + 1 - py - a, x + 1 - px - b): halo offset (2 - a, 2 - b) minus the phase
        constexpr int r = PD ? 2 - (tap >> 1) : (PH ? tap >> 1 : tap / 3), s = PD ? 2 - (tap & 1) : (PH ? tap & 1 : tap % 3);
        const unsigned char* B = smB + bbuf * BSLAB + b_frag;
        const unsigned char* A = smA + abuf * G::ABUF + a_frag - (PD ? (((vc >> 1) & 1) * HP + (vc & 1)) * 16 : 0);
        constexpr int JG = TN == 3 ? 3 : (TN < 2 ? TN : 2);      // column tiles per pass: BN = 256 runs two passes over the same A fragments
        bf16x8 a[NPC][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) a[pc][i] = *reinterpret_cast<const bf16x8*>(A + pc * G::APLANE + ((i + r) * HP + s) * 16);
        // six products, smallest first; the accumulators interleave so no MFMA waits for its predecessor
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
        constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int j0 = 0; j0 < TN; j0 += JG) {
            bf16x8 b[NPC][JG];
#pragma unroll
            for (int j = 0; j < JG; ++j)
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) b[pc][j] = *reinterpret_cast<const bf16x8*>(B + pc * BPLANE + (j0 + j) * 512);
#pragma unroll
            for (int t = 6 - NP; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < JG; ++j)
                        // D = W * X^T: MFMA rows = output channels, columns = pixels, so a lane ends up with 4 CONSECUTIVE channels of one
                        // pixel per accumulator quad (16-byte stores in the epilogue instead of 4-byte ones)
                        acc[i][j0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[PB[t]][j], a[PA[t]][i], acc[i][j0 + j], 0, 0, 0);
        }
    };

    // ---- prologue: halo of chunk 0 -> A[0], slab (0, 0) -> B[0], slab (0, 1) -> registers
    load_pro(0);
#pragma unroll
    for (int j = 0; j < G::NU; ++j) load_a(j, 0);
    load_b(0, 0);
#pragma unroll
    for (int j = 0; j < G::NU; ++j) store_a(j, 0);
    store_b(0);
    load_b(0, 1);
    __syncthreads();

    // ---- main loop.  Step u = 9 c + tap reads A[c & 1] and B[u & 1]; during it the thread
    //   stores the slab of step u + 1 (registers, loaded during step u - 1) into B[(u + 1) & 1]   (last read in step u - 1: barrier),
    //   issues the loads of slab u + 2,
    //   and advances the halo of chunk c + 1: unit j is loaded at tap 2 j and split / stored into A[(c + 1) & 1] at tap 2 j + 2.
    // Loads past the end re-read the last chunk (clamped index: no branch around a load), their stores land in buffers nobody reads.
    int ubuf = 0;
    for (int c = 0; c < NC; ++c) {
        const int cn = (c + 1 < NC) ? c + 1 : c;
        const int abuf = c & 1;
        auto step = [&](auto TAP) {
            constexpr int tap = decltype(TAP)::value;
            store_b(ubuf ^ 1);
            if constexpr (tap + 2 < KT) load_b(c, tap + 2); else load_b(cn, tap + 2 - KT);
            if constexpr (tap == 0) load_pro(cn);      // (the stores of chunk c's halo, which used the previous pair, are all behind us)
            if constexpr (PH || PD) {                  // four steps per chunk: every unit of the next halo is loaded at tap 0, unit j stored at tap 1 + j
                if constexpr (tap >= 1 && tap - 1 < G::NU) store_a(tap - 1, abuf ^ 1);
                if constexpr (tap == 0) {
#pragma unroll
                    for (int j = 0; j < G::NU; ++j) load_a(j, cn);
                }
            } else {
                if constexpr (tap >= 2 && tap % 2 == 0 && (tap - 2) / 2 < G::NU) store_a((tap - 2) / 2, abuf ^ 1);
                if constexpr (tap % 2 == 0 && tap / 2 < G::NU) load_a(tap / 2, cn);
            }
            compute(TAP, abuf, ubuf, c);
            __syncthreads();
            ubuf ^= 1;
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        if constexpr (KT == 9) {
            step(std::integral_constant<int, 4>{});
            step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{});
            step(std::integral_constant<int, 7>{});
            step(std::integral_constant<int, 8>{});
        }
    }

    // ------------------------------------------------------------------ epilogue
    // lane = (pixel px = lane & 31 of patch row i, half); accumulator quad g of column tile j = channels cb + 8 g .. + 3 of that pixel
    float* __restrict__ y = p.y;
    const int px = lane & 31;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int cb = n0 + (wn * TN + j) * 32 + 4 * fhalf;
        float s1[16], s2[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) s1[k] = s2[k] = 0.f;
        // Round 6: everything a channel quad READS -- its per-channel constants, and for BOTH patch rows the ReLU mask, the residual and the old output of an
        // accumulating launch -- is issued first, from clamped (always valid) addresses, and waited for once.  Before, every load sat inside the per-row /
        // per-quad branches and the compiler waited for each before the next (218 s_waitcnt vmcnt(0) behind the last MFMA of this kernel): up to four
        // dependent round trips per quad, 8-16 quads per lane, with ONE workgroup per CU and nothing else to run meanwhile.
        long long m_row[2];
        bool row_ok[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // patch row; phase form: output pixel (2 pyl + ph_y, 2 (x0 + px) + ph_x)
            const int pyl = y0 + 2 * wm + i;
            row_ok[i] = pyl < (PH ? p.Hin : p.Hout);
            const int pyc = row_ok[i] ? pyl : y0;               // (a patch's first row always exists)
            m_row[i] = PH ? ((long long)n_img * p.Hout + 2 * pyc + ph_y) * p.Wout + 2 * (x0 + px) + ph_x
                          : ((long long)n_img * p.Hout + pyc) * p.Wout + x0 + px;
        }
        const bool mask_vec = p.mask && (p.ldm % 4) == 0 && (reinterpret_cast<uintptr_t>(p.mask) & 15) == 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = cb + 8 * g;
            const bool full = c0 + 3 < p.Cout;                  // whole quad inside Cout: 16-byte accesses (ldy, ldr % 4 == 0, 16-B aligned bases)
            const int cq = full ? c0 : 0;
            float bias[4], osc[4], osh[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { bias[e] = 0.f; osc[e] = 1.f; osh[e] = 0.f; }
            if (p.bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bias[e] = p.bias[c0 + e < p.Cout ? c0 + e : 0];
            }
            if (p.out_scale) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { osc[e] = p.out_scale[c0 + e < p.Cout ? c0 + e : 0]; osh[e] = p.out_shift[c0 + e < p.Cout ? c0 + e : 0]; }
            }
            f32x4 mk[2], r4[2], o4[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                mk[i] = f32x4{1.f, 1.f, 1.f, 1.f};
                r4[i] = o4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (p.mask) {                                       // fused ReLU backward of the producer of this gradient (data-gradient launches)
                if (mask_vec) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) mk[i] = *reinterpret_cast<const f32x4*>(p.mask + (size_t)m_row[i] * p.ldm + cq);
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) mk[i][e] = p.mask[(size_t)m_row[i] * p.ldm + (c0 + e < p.Cout ? c0 + e : 0)];
                }
            }
            if (p.res) {
#pragma unroll
                for (int i = 0; i < 2; ++i) r4[i] = *reinterpret_cast<const f32x4*>(p.res + (size_t)m_row[i] * p.ldr + cq);
            }
            if (p.accumulate) {
#pragma unroll
                for (int i = 0; i < 2; ++i) o4[i] = *reinterpret_cast<const f32x4*>(y + (size_t)m_row[i] * p.ldy + cq);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (!row_ok[i]) continue;
                float* dst = y + (size_t)m_row[i] * p.ldy + c0;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = (acc[i][j][4 * g + e] * p.alpha + bias[e]) * osc[e] + osh[e];
                    v[e] = mk[i][e] > 0.f ? v[e] : 0.f;
                }
                if (full) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += r4[i][e];
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += o4[i][e];
                    *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s1[4 * g + e] += v[e]; s2[4 * g + e] += v[e] * v[e]; }
                } else {                                       // ragged Cout (e.g. 126 channels next to 2 foreign ones in a wider buffer)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (c0 + e < p.Cout) {
                            float u = v[e];
                            if (p.res) u += p.res[(size_t)m_row[i] * p.ldr + c0 + e];
                            if (p.relu) u = fmaxf(u, 0.f);
                            if (p.accumulate) u += dst[e];
                            dst[e] = u;
                            s1[4 * g + e] += u;
                            s2[4 * g + e] += u * u;
                        }
                    }
                }
            }
        }
        if (p.stats) {
            // per-channel sums over the 32 pixel lanes of a half: butterfly reduce-scatter (16 + 16 shuffles for the 16 + 16 values);
            // after it lane L holds channel index kk = 8 b4 + 4 b3 + 2 b2 + b1 (bN = bit N of L), lanes L and L ^ 1 the same total
            auto stage = [&](float (&v)[16], auto W) {          // lanes L / L ^ 2W: the low one keeps v[0..W), the high one v[W..2W)
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
            const int kk = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
            const int cch = cb + 8 * (kk >> 2) + (kk & 3);
            if ((lane & 1) == 0 && cch < p.Cout) {
                double* st = stat_slot(p, stat_group(p, (long long)n_img * p.Hout * p.Wout, (long long)p.N * p.Hout * p.Wout), blockIdx.x);   // (a patch lies in one image)
                atomicAdd(st + cch, (double)s1[0]);
                atomicAdd(st + p.Cout + cch, (double)s2[0]);
            }
        }
    }
    if (p.fin_scale) fused_bn_finalize(p, (unsigned)total_tiles * gridDim.y, lin + total_tiles * (int)blockIdx.y, reinterpret_cast<double*>(smem), (int)(sizeof(smem) / 8));      // (every workgroup with a tile reaches this point with all of its threads)
}

int g_halo_on = -1;              // -1: not initialised (MRFA_CONV_HALO)
int g_halo_min_tiles = 128;       // see halo_enough()
int g_halo_pr = 0;               // 0 = by workgroup count, 4 / 8 = forced patch height
int g_conv_small = 1;

int g_halo_bn256 = 1;
int g_halo_bn192 = 1;
int g_halo_bn64_fill = 1;

// (patch rows, BN): PR = 8 (8 waves, 1 workgroup / CU) when that still gives every CU a workgroup, else PR = 4 (4 waves, 2 workgroups
// per CU); BN = 256 (each wave 64 pixels x 128 channels: the halo is staged once for twice the MFMAs, 18 instead of 24 fragment reads
// per 48 MFMAs) when Cout pads to a multiple of 256 anyway and the workgroup count allows
int g_halo_phase = 1;

// the phase form of a fused upsample applies: pre-summed phase weights present, the LOW-resolution grid tiles into 8 x 32 patches
bool halo_phase(const mrfa_conv_params& p) {
    const int mode = mrfa_get_mfma_mode();
    return (mode == 1 || mode == 2) && g_halo_phase && p.ups == 1 && p.w_phase != nullptr && (p.Win % PW) == 0 && (p.Hin % 8) == 0 && 3 * p.w_phase_piece < (1ll << 31);
}

// enough workgroups for a launch with `tiles` workgroups of width bn?  One per CU (2 x g_halo_min_tiles = 256) in general; half a chip's worth is
// still better than the row-tiled kernel for <= 128-wide tiles on >= 64-pixel-wide outputs with >= 64 channels on both sides (measured: 256->128
// @64^2 forward 174 -> 148 us, 192->128 152 -> 115, the dense-motion up block 512->128 @32->64 in phase form 284 -> 145; but 512->512 @32^2
// 259 -> 285 and 32->32 @64^2 15 -> 23 us)
bool halo_enough(const mrfa_conv_params& p, long long tiles, int bn) {
    if (tiles >= 2ll * g_halo_min_tiles) return true;
    return bn <= 128 && tiles >= g_halo_min_tiles && p.Wout >= 64 && p.Cin >= 64 && p.Cout >= 64;
}

void halo_config(const mrfa_conv_params& p, int& PR, int& BN) {
    PR = 0;
    BN = p.Cout <= 64 ? 64 : 128;
    const bool wide = g_halo_bn256 && p.Cout > 128 && cdiv(p.Cout, 256) * 256 == cdiv(p.Cout, 128) * 128;
    if (halo_phase(p)) {                           // four phase workgroups per low-resolution patch
        const long long patches = (long long)p.N * (p.Hin / 8) * (p.Win / PW) * 4;
        if (wide && halo_enough(p, patches * cdiv(p.Cout, 256), 256)) { PR = 8; BN = 256; return; }
        if (halo_enough(p, patches * cdiv(p.Cout, BN), BN)) { PR = 8; return; }
    }
    const long long patches8 = (long long)p.N * cdiv(p.Hout, 8) * (p.Wout / PW), patches4 = (long long)p.N * cdiv(p.Hout, 4) * (p.Wout / PW);
    if (g_halo_pr != 4 && p.Hout % 8 == 0) {
        // 192-wide tiles where they pad less than 128- / 256-wide ones (data gradients into 160 / 192 channels: 256 -> 192 columns of MFMA work)
        if (g_halo_bn192 && !p.in_scale && cdiv(p.Cout, 192) * 192 < cdiv(p.Cout, 128) * 128 && halo_enough(p, patches8 * cdiv(p.Cout, 192), 192)) {
            PR = 8;
            BN = 192;
            return;
        }
        if (wide && halo_enough(p, patches8 * cdiv(p.Cout, 256), 256)) { PR = 8; BN = 256; return; }
        // too few 128-wide tiles for one workgroup per CU but enough 64-wide ones: fill the chip with the narrower tile (more fragment reads per
        // MFMA, twice the workgroups) -- the low-resolution levels (512 -> 512 @32^2, 256 -> 128 @64^2)
        if (g_halo_bn64_fill && BN == 128 && patches8 * cdiv(p.Cout, 128) < 2ll * g_halo_min_tiles &&
            patches8 * cdiv(p.Cout, 64) >= 2ll * g_halo_min_tiles) {
            PR = 8;
            BN = 64;
            return;
        }
        if (halo_enough(p, patches8 * cdiv(p.Cout, BN), BN)) { PR = 8; return; }
    }
    // 4-row patches run two workgroups per CU: they need twice the workgroups to fill the chip (measured: 512->512 @32^2, 256 workgroups of
    // 4 rows lose to the row-tiled kernel with its K split)
    if (g_halo_pr != 8 && patches4 * cdiv(p.Cout, BN) >= 4ll * g_halo_min_tiles) PR = 4;
}

// (is the launch the phase form?  halo_config picks it first; it falls through to the plain form when the phase grid is too small)
bool halo_uses_phase(const mrfa_conv_params& p, int PR, int BN) {
    if (!halo_phase(p) || PR != 8) return false;
    const long long patches = (long long)p.N * (p.Hin / 8) * (p.Win / PW) * 4;
    return halo_enough(p, patches * cdiv(p.Cout, BN), BN);
}

}  // namespace

static bool halo_on() {
    if (g_halo_on < 0) { const char* e = getenv("MRFA_CONV_HALO"); g_halo_on = !(e && e[0] == '0'); }
    return g_halo_on != 0;
}

extern "C" int mrfa_set_tuning(const char* key, int value) {
    if (!key) return -1;
    if (!strcmp(key, "conv_halo")) { const int prev = halo_on(); g_halo_on = value != 0; return prev; }
    if (!strcmp(key, "conv_halo_min_tiles")) { const int prev = g_halo_min_tiles; g_halo_min_tiles = value; return prev; }
    if (!strcmp(key, "conv_halo_phase")) { const int prev = g_halo_phase; g_halo_phase = value != 0; return prev; }
    if (!strcmp(key, "conv_halo_bn256")) { const int prev = g_halo_bn256; g_halo_bn256 = value != 0; return prev; }
    if (!strcmp(key, "conv_halo_bn64_fill")) { const int prev = g_halo_bn64_fill; g_halo_bn64_fill = value != 0; return prev; }
    if (!strcmp(key, "conv_halo_bn192")) { const int prev = g_halo_bn192; g_halo_bn192 = value != 0; return prev; }
    if (!strcmp(key, "conv_halo_pr")) { const int prev = g_halo_pr; g_halo_pr = value; return prev; }
    if (!strcmp(key, "wgrad_halo")) return mrfa_tuning_wgrad_halo(value != 0);
    if (!strcmp(key, "wgrad_halo_min_wgs")) return mrfa_tuning_wgrad_halo_min(value);
    if (!strcmp(key, "wgrad_halo_target_wgs")) return mrfa_tuning_wgrad_halo_target(value);
    if (!strcmp(key, "wgrad_halo_phase")) return mrfa_tuning_wgrad_halo_phase(value);
    if (!strcmp(key, "conv_fewout3")) return mrfa_tuning_fewout3(value != 0);
    if (!strcmp(key, "attention_mfma")) return mrfa_tuning_attention_mfma(value != 0);
    if (!strcmp(key, "conv_small")) { const int prev = g_conv_small; g_conv_small = value != 0; return prev; }
    if (!strcmp(key, "conv_lean")) return mrfa_tuning_conv_lean(value != 0);
    if (!strcmp(key, "conv_lean_min_wgs")) return mrfa_tuning_conv_lean_min(value);
    if (!strcmp(key, "conv_lean_geo")) return mrfa_tuning_conv_lean_geo(value);
    if (!strcmp(key, "wgrad_lean")) return mrfa_tuning_wgrad_lean(value != 0);
    if (!strcmp(key, "gemm_lean")) return mrfa_tuning_gemm_lean(value);
    return -1;
}

int mrfa_tuning_conv_small() { return g_conv_small; }

// ups == 2: phase DATA GRADIENT of a fused-upsample 3x3 layer (x = high-resolution gradient, y = low-resolution input gradient)
static bool halo_phase_dgrad(const mrfa_conv_params& p) {
    const int mode = mrfa_get_mfma_mode();
    if (!halo_on() || !g_halo_phase || (mode != 1 && mode != 2)) return false;
    if (p.ups != 2 || !p.w_phase || p.kflat > 0 || p.R != 3 || p.S != 3 || p.pad != 1 || p.nbatch > 1 || p.splitk > 1 || p.tile || p.in_scale) return false;
    if (p.Hin != 2 * p.Hout || p.Win != 2 * p.Wout || (p.Wout % PW) != 0 || (p.Hout % 8) != 0 || (p.Cin % 32) != 0 || p.Cout < 32) return false;
    if ((p.ldy % 4) != 0 || !aligned16(p.y) || (p.ldx % 4) != 0 || !aligned16(p.x) || 3 * p.w_phase_piece >= (1ll << 31) || p.w_tap >= (1ll << 31)) return false;
    if ((long long)p.N * p.Hin * p.Win * p.ldx >= (1ll << 31)) return false;
    return true;
}

extern "C" int mrfa_conv2d_phase_dgrad_supported(const mrfa_conv_params* p) { return p && halo_phase_dgrad(*p) ? 1 : 0; }

extern "C" int mrfa_conv2d_mask_supported(const mrfa_conv_params* p) { return p && p->kflat == 0 && mrfa_conv_halo_eligible(*p) ? 1 : 0; }

bool mrfa_conv_halo_eligible(const mrfa_conv_params& p) {
    if (p.ups == 2) return halo_phase_dgrad(p);
    const int mode = mrfa_get_mfma_mode();
    if (!halo_on() || (mode != 1 && mode != 2 && mode != 3)) return false;       // (mode 3: w_split = ONE plane rounded to nearest even, pack modes 14 / 15)
    if (p.kflat > 0 || p.R != 3 || p.S != 3 || p.pad != 1 || !p.w_split || p.nbatch > 1 || p.splitk > 1 || p.tile || p.stride > 1 || p.stride < 0) return false;
    if ((p.Wout % PW) != 0 || p.Hout < 4 || (p.Cin % 32) != 0 || p.Cout < 32) return false;
    if (p.Hout != (p.Hin << p.ups) || p.Wout != (p.Win << p.ups)) return false;
    if ((p.ldy % 4) != 0 || !aligned16(p.y) || (p.res && ((p.ldr % 4) != 0 || !aligned16(p.res)))) return false;       // float4 epilogue
    if ((p.ldx % 4) != 0 || !aligned16(p.x) || 3 * p.w_piece >= (1ll << 31) || p.w_tap >= (1ll << 31)) return false;
    if ((long long)p.N * p.Hin * p.Win * p.ldx >= (1ll << 31)) return false;
    if (p.in_scale && !p.in_relu) return false;
    int PR, BN;
    halo_config(p, PR, BN);
    return PR != 0;
}

int mrfa_conv_halo_launch(hipStream_t st, const mrfa_conv_params& p) {
    int PR, BN;
    if (p.ups == 2) {                                // phase data gradient: 8-row patches on the output grid, <= 128-wide tiles
        PR = 8;
        BN = p.Cout <= 64 ? 64 : 128;
    } else {
        halo_config(p, PR, BN);
    }
    const bool phase = p.ups != 2 && halo_uses_phase(p, PR, BN);
    const int tiles_n = cdiv(p.Cout, BN), tiles_x = (phase ? p.Win : p.Wout) / PW, tiles_y = cdiv(phase ? p.Hin : p.Hout, PR);
    const long long total = (long long)p.N * tiles_y * tiles_x * tiles_n;
    dim3 grid((unsigned)(cdiv(total, 8) * 8), phase ? 4u : 1u);
    const bool three = mrfa_get_mfma_mode() == 2, one = mrfa_get_mfma_mode() == 3;
    const bool pro = p.in_scale != nullptr;
#define HALO_LAUNCH(PR_, PRO_, BN_, MODE_)                                                                                                                      \
    do {                                                                                                                                                     \
        if constexpr ((MODE_) == 0) {                                                                                                                        \
            if (one) { hipLaunchKernelGGL((conv_halo_kernel<PR_, PRO_, BN_, 1, 0>), grid, dim3(PR_ * 64), 0, st, p, tiles_n, tiles_x, tiles_y, (int)total); break; } \
        }                                                                                                                                                    \
        if (three) hipLaunchKernelGGL((conv_halo_kernel<PR_, PRO_, BN_, 3, MODE_>), grid, dim3(PR_ * 64), 0, st, p, tiles_n, tiles_x, tiles_y, (int)total);    \
        else hipLaunchKernelGGL((conv_halo_kernel<PR_, PRO_, BN_, 6, MODE_>), grid, dim3(PR_ * 64), 0, st, p, tiles_n, tiles_x, tiles_y, (int)total);          \
    } while (0)
#define HALO_BN(PR_, PRO_, MODE_)                                                     \
    do {                                                                              \
        if (BN == 64) HALO_LAUNCH(PR_, PRO_, 64, MODE_);                              \
        else if (BN == 256) HALO_LAUNCH(PR_, PRO_, 256, MODE_);                       \
        else HALO_LAUNCH(PR_, PRO_, 128, MODE_);                                      \
    } while (0)
    if (p.ups == 2) { if (BN == 64) HALO_LAUNCH(8, false, 64, 2); else HALO_LAUNCH(8, false, 128, 2); }
    else if (phase) { if (pro) HALO_BN(8, true, 1); else HALO_BN(8, false, 1); }
    else if (PR == 8 && BN == 192) HALO_LAUNCH(8, false, 192, 0);
    else if (PR == 8) { if (pro) HALO_BN(8, true, 0); else HALO_BN(8, false, 0); }
    else {
        if (BN == 64) { if (pro) HALO_LAUNCH(4, true, 64, 0); else HALO_LAUNCH(4, false, 64, 0); }
        else { if (pro) HALO_LAUNCH(4, true, 128, 0); else HALO_LAUNCH(4, false, 128, 0); }
    }
#undef HALO_BN
#undef HALO_LAUNCH
    MRFA_CHECK_LAUNCH("mrfa_conv2d_nhwc(halo)");
    return 0;
}
