// fp32 implicit-GEMM convolution on the bf16 matrix pipe: operands split EXACTLY into three bf16 pieces, six products.
//
// A float has 24 significand bits = 3 x 8: chopping x to its top 16 bits gives a bf16 x1 with x - x1 exact, and twice
// more gives x = x1 + x2 + x3 with NO error (bf16 has fp32's exponent range).  Then
//     a*b = a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1) + [a2b3 + a3b2 + a3b3]
// where every kept product of two 8-bit significands is exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16 and the
// dropped bracket is < 2^-23 |ab| -- below the rounding of a single fp32 multiply.  The result is an fp32-accurate dot
// product (same error class as the v_mfma_f32_32x32x2_f32 kernel of conv_mfma.hip: only accumulation order differs),
// computed with six bf16 MFMAs per 32x32x16 block: the bf16 pipe is 16x the fp32 matrix pipe (2.5 PF vs 157 TF dense),
// so the ceiling is 2.5 PF / 6 = 417 TF/s of fp32-equivalent work.  (Same idea as "3xTF32" / cuBLAS "BF16x9".)
//
// Structure (128 x 128 tile, 4 wave64, each wave 64 x 64 = 2 x 2 MFMA tiles, BK = 32):
//   global fp32 tile -> registers (same loaders / fused prologues as conv_mfma.hip) -> split with VALU (and / sub / perm)
//   -> LDS as three bf16 planes per operand, rows of 32 k = 64 B, 16-byte slots XOR-swizzled with (row >> 2) & 3 so that
//   both the 4-lanes-per-row writes and the 16-rows-per-group ds_read_b128 fragment reads are bank-conflict free with no
//   padding (48 KB per workgroup, 3 workgroups = 12 waves per CU) -> 12 fragment reads and 24 MFMAs per k16 step per wave.
// Epilogue identical to conv_mfma.hip (C/D layout of the 32x32 MFMAs does not depend on the input type).
#include <type_traits>
#include "common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, NT = 256;
constexpr int PLANE = 128 * 32;                    // bytes per (operand, piece) plane of one k16 slab: 128 rows x 16 bf16
constexpr int SLAB = 6 * PLANE;                    // A pieces 1..3, B pieces 1..3

__device__ __forceinline__ unsigned pack_hi16(float a, float b) {      // (bf16 chop of b) << 16 | (bf16 chop of a)
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float chop_rest(float x) { return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

// 4 consecutive k of one row -> the three bf16x4 pieces (2 dwords each)
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

// plain bf16 operands (NP = 1): round-to-nearest-even of the fp32 value, one product -- the arithmetic of a bf16 autocast
__device__ __forceinline__ unsigned rne16(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void round1(const f32x4 v, u32x2& p1) {
    p1[0] = rne16(v.x) | (rne16(v.y) << 16);
    p1[1] = rne16(v.z) | (rne16(v.w) << 16);
}

// BN = 128: waves 2 x 2, each 64 x 64.  BN = 64 (layers with 64 output channels: half of a 128-wide tile would be padding):
// waves 2 x 2, each 64 x 32 -- same loaders, the B operand simply has 64 rows.
// TR: transposed product (D = W * X^T) + 16-byte epilogue accesses -- every launch WITHOUT a K split.  Split-K launches keep the row-major
// product: their epilogue is fp32 atomics, and with the transposed layout one atomic instruction would touch 64 cache lines (32 pixels x
// 2 halves, ldy apart) instead of two 128-byte rows (measured: 512->512 @8^2 47 -> 224 us).
template <bool PRO, bool WS, int BN, int NP = 6, bool TR = true>
__global__ __launch_bounds__(NT, 3) void conv_bf16x6_kernel(const mrfa_conv_params p, const int KT, const int kt_per_split, const long long M,
                                                            const int tiles_n, const int total_tiles) {
    constexpr int TM = 2, TN = BN / 64;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SLAB];           // two k16 slab buffers (48 KB)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int per_xcd = (int)gridDim.x >> 3;
    const int lin = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (lin >= total_tiles) return;
    const int tile_m = lin / tiles_n;
    const int tile_n = lin - tile_m * tiles_n;
    const long long m0 = (long long)tile_m * BM;
    const int n0 = tile_n * BN;
    const int bz = blockIdx.y;

    const float* __restrict__ x = p.x + (size_t)bz * p.x_bs;
    const float* __restrict__ w = p.w + (size_t)bz * p.w_bs;
    float* __restrict__ y = p.y + (size_t)bz * p.y_bs;

    // Loader mapping: thread (lrow = tid >> 2, q = tid & 3) owns, for rows lrow and lrow + 64, the float4 chunks q
    // (k = 4q..4q+3, first k16 slab of the 32-wide k-tile) and q + 4 (second slab): every thread has the same amount of
    // split / LDS-store work in both slabs.
    const int lrow = tid >> 2;    // 0..63
    const int q4 = tid & 3;
    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;
    const int HWo = p.Hout * p.Wout;
    const int KC = p.Cin >> 5;

    int a_oy[2], a_ox[2], a_base[2];
    bool a_ok[2], b_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const long long m = m0 + lrow + 64 * j;
        a_ok[j] = m < M;
        const long long mm = a_ok[j] ? m : 0;
        const int n_img = (int)(mm / HWo);
        const int rem = (int)(mm - (long long)n_img * HWo);
        a_oy[j] = rem / p.Wout;
        a_ox[j] = rem - a_oy[j] * p.Wout;
        a_base[j] = n_img * p.Hin * p.Win;
        b_ok[j] = (lrow + 64 * j) < BN && (n0 + lrow + 64 * j) < p.w_rows;
    }

    f32x4 ra[2][2], rb[2][2], psc[2], psh[2];      // [row j][slab h]
    // WS: weights arrive pre-split (pack modes 8 / 9): thread (brow = tid >> 1, bhalf = tid & 1) copies, per slab and piece,
    // the 8 bf16 of row brow that form one MFMA fragment half -- no VALU work on the B operand at all
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 wsp[2][3];
    const int brow = tid >> 1, bhalf = tid & 1;
    const bool b_loader = brow < BN;
    const unsigned short* l_ws = reinterpret_cast<const unsigned short*>(p.w_split) + (size_t)bz * p.w_bs;
    bool a_inb[2];
    int l_r = 0, l_s = 0, l_c = 0;
    const float* l_w = w;

    // ISSUE the global loads of one 32-wide k-tile; nothing here consumes a loaded register (masking, the fused BN+ReLU
    // prologue and the bf16 split happen in store_slab), so the loads stay in flight across the matrix work
    auto load_tile = [&]() {
        const int c0 = l_c * 32 + q4 * 4;
        if constexpr (PRO) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                psc[h] = *reinterpret_cast<const f32x4*>(p.in_scale + c0 + 16 * h);
                psh[h] = *reinterpret_cast<const f32x4*>(p.in_shift + c0 + 16 * h);
            }
        }
        const int dr = l_r - p.pad, ds = l_s - p.pad;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int iy = a_oy[j] + dr;
            const int ix = a_ox[j] + ds;
            const bool inb = a_ok[j] && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
            a_inb[j] = inb;
            // out-of-bounds rows read a valid address (pixel 0 of their image) and are zeroed when stored
            const int pix = inb ? (a_base[j] + (iy >> p.ups) * p.Win + (ix >> p.ups)) : a_base[j];
            const float* src = x + (size_t)pix * p.ldx + c0;
            ra[j][0] = *reinterpret_cast<const f32x4*>(src);
            ra[j][1] = *reinterpret_cast<const f32x4*>(src + 16);
        }
        if constexpr (WS) {
            if (b_loader) {
                // k16-chunk-major planes (pack_multi.hip chunk_major()): chunk 2 l_c + h of this tap, row n0 + brow, half bhalf
                const unsigned short* src = l_ws + (size_t)(2 * l_c) * (p.w_rows * 16) + (size_t)(n0 + brow) * 16 + bhalf * 8;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) wsp[h][pc] = *reinterpret_cast<const u32x4*>(src + (size_t)pc * p.w_piece + (size_t)h * (p.w_rows * 16));
            }
        } else {
            const float* wt = l_w + c0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = b_ok[j] ? (n0 + lrow + 64 * j) : 0;
                const float* src = wt + (size_t)row * p.w_ld;
                rb[j][0] = *reinterpret_cast<const f32x4*>(src);
                rb[j][1] = *reinterpret_cast<const f32x4*>(src + 16);
            }
        }
        // K order: the R*S taps of one 32-channel chunk are consecutive k-tiles (channel-chunk-major).  The shifted re-reads of
        // a chunk by the next tap then hit the XCD's L2 while it still holds that chunk: with tap-major order the working set of
        // the 32 workgroups of an XCD between two visits of a row was the whole Cin depth of ~34 image rows (> the 4 MB L2).
        // Measured on the pre-split micro-benchmark (tools/ubench/split_gemm.hip): +7 % at Cout = 128, +2 % at 256.
        l_w += p.w_tap;
        l_ws += p.w_tap;
        if (++l_s == p.S) {
            l_s = 0;
            if (++l_r == p.R) {
                l_r = 0;
                ++l_c;
                l_w -= (size_t)p.R * p.S * p.w_tap;
                l_ws -= (size_t)p.R * p.S * p.w_tap;
            }
        }
    };

    // LDS: two slab buffers (k16 each), each 6 planes (A pieces 1..3, B pieces 1..3) of 128 rows x 16 bf16 = 32 B per row;
    // the two 16-byte halves of a row are swapped for rows with bit 3 set, so the 16 rows of a ds_read_b128 lane group
    // (and the 4-lanes-per-row ds_write_b64 groups) hit 16 distinct 16-byte bank slots
    auto store_slab = [&](int h, int buf) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        unsigned char* base = smem + buf * SLAB;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = lrow + 64 * j;
            const int off = row * 32 + ((((q4 >> 1) ^ (row >> 3)) & 1) << 4) + ((q4 & 1) << 3);
            f32x4 v = ra[j][h];
            if constexpr (PRO) {                   // fused pre-activation BN + ReLU (in_relu is always set with in_scale)
                v = v * psc[h] + psh[h];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            v = a_inb[j] ? v : z;
            u32x2 p1, p2, p3;
            if constexpr (NP == 1) {
                round1(v, p1);
                *reinterpret_cast<u32x2*>(base + 0 * PLANE + off) = p1;
                static_assert(NP != 1 || !WS, "the pre-split weight pieces are truncations: plain bf16 rounds the fp32 weights itself");
                round1(b_ok[j] ? rb[j][h] : z, p1);
                *reinterpret_cast<u32x2*>(base + 3 * PLANE + off) = p1;
            } else {
                split3(v, p1, p2, p3);
                *reinterpret_cast<u32x2*>(base + 0 * PLANE + off) = p1;
                *reinterpret_cast<u32x2*>(base + 1 * PLANE + off) = p2;
                *reinterpret_cast<u32x2*>(base + 2 * PLANE + off) = p3;
                if constexpr (!WS) {
                    split3(b_ok[j] ? rb[j][h] : z, p1, p2, p3);
                    *reinterpret_cast<u32x2*>(base + 3 * PLANE + off) = p1;
                    *reinterpret_cast<u32x2*>(base + 4 * PLANE + off) = p2;
                    *reinterpret_cast<u32x2*>(base + 5 * PLANE + off) = p3;
                }
            }
        }
        if constexpr (WS) {
            if (b_loader) {
                const int off = brow * 32 + (((bhalf ^ (brow >> 3)) & 1) << 4);
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x4*>(base + (3 + pc) * PLANE + off) = wsp[h][pc];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_row = lane & 31;
    const int frag_half = lane >> 5;
    auto compute_slab = [&](int buf) {
        const unsigned char* base = smem + buf * SLAB;
        constexpr int NPC = NP == 6 ? 3 : (NP == 3 ? 2 : 1);      // pieces needed: all three, the two leading ones, or one
        bf16x8 a[3][TM], b[3][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * 64 + i * 32 + frag_row;
            const int off = row * 32 + (((frag_half ^ (row >> 3)) & 1) << 4);
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) a[pc][i] = *reinterpret_cast<const bf16x8*>(base + pc * PLANE + off);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = wn * (TN * 32) + j * 32 + frag_row;
            const int off = row * 32 + (((frag_half ^ (row >> 3)) & 1) << 4);
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) b[pc][j] = *reinterpret_cast<const bf16x8*>(base + (3 + pc) * PLANE + off);
        }
        // six products, smallest first; the four accumulators interleave so no MFMA waits for its predecessor
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
        constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 6 - NP; t < 6; ++t)                    // NP = 3: a1*b0 + a0*b1 + a0*b0 (drops the three ~2^-16 terms)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    // D = W * X^T (rows = output channels, columns = pixels): a lane ends up with 4 consecutive channels of one pixel per
                    // accumulator quad -> 16-byte epilogue accesses (as conv_halo.hip; the short-K launches left on this kernel -- 1x1
                    // layers, low-resolution levels -- are dominated by their store tail)
                    if constexpr (TR) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[PB[t]][j], a[PA[t]][i], acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]][i], b[PB[t]][j], acc[i][j], 0, 0, 0);
    };

    const int kt_begin = blockIdx.z * kt_per_split;
    const int kt_end = min(KT, kt_begin + kt_per_split);
    {
        const int T = p.R * p.S;
        l_c = kt_begin / T;
        const int tap0 = kt_begin - l_c * T;
        l_r = tap0 / p.S;
        l_s = tap0 - l_r * p.S;
        l_w = w + (size_t)tap0 * p.w_tap;
        l_ws += (size_t)tap0 * p.w_tap;
    }
    // Software pipeline over k16 slabs u = 2 kt + h, LDS buffer u & 1:
    //   iteration kt:   compute(slab 2kt)   | split + store slab 2kt+1 (registers of tile kt)   ; issue loads of tile kt+1 ; barrier
    //                   compute(slab 2kt+1) | split + store slab 2kt+2 (registers of tile kt+1) ; barrier
    // one barrier per slab; the VALU split / LDS stores of the next slab sit in the same basic block as the 24 MFMAs of
    // the current one, and the global loads have a whole slab of matrix work to land.
    if (kt_begin < kt_end) {
        load_tile();
        store_slab(0, 0);
    }
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = (kt + 1) < kt_end;
        compute_slab(0);
        store_slab(1, 1);
        if (more) load_tile();
        __syncthreads();
        compute_slab(1);
        store_slab(0, 0);            // (after the last tile: stale registers into a buffer nobody reads again)
        __syncthreads();
    }

    if constexpr (!TR) {
    // ------------------------------------------------------------------ epilogue of the split-K launches (as conv_mfma.hip)
    const int half = lane >> 5;
    const bool splitk = p.splitk > 1;
    if (splitk) {
        // partial tile: device-scope atomics into y (zeros, or the bias written by the init pass)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = n0 + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (c < p.Cout && m < M) atomicAdd(y + (size_t)m * p.ldy + c, acc[i][j][r] * p.alpha);
                }
        }
        // v8: the workgroup that completes the tile applies the epilogue itself (otherwise splitk_epilogue_kernel does, in a pass of its own)
        if (!p.sk_ticket || !splitk_last_arriver(p.sk_ticket + (size_t)bz * total_tiles + lin, gridDim.z)) return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = n0 + (wn * TN + j) * 32 + (lane & 31);
        const bool c_ok = c < p.Cout;
        float bias = 0.f, osc = 1.f, osh = 0.f;
        if (c_ok) {
            if (p.bias) bias = p.bias[c];
            if (p.out_scale) { osc = p.out_scale[c]; osh = p.out_shift[c]; }
        }
        float s1 = 0.f, s2 = 0.f;
        // Everything a 32-row block of this column tile READS first, all of it in flight together, from clamped (always valid) addresses.  With the loads inside the
        // per-element branch the compiler waited for each one before the next was issued: the last arriver of a K split read its 64 values per lane
        // back one memory-side round trip at a time -- ~30 us of a 65 us launch on the low-resolution levels (round 6, the ISA of this loop:
        // global_load_dword ... sc1 ; s_waitcnt vmcnt(0), 64 times; tools/sweep_splitk.py FUSED=1 before / after in profiles/r6_sweep_splitk_*)
        const int cq = c_ok ? c : 0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float yv[16], rv[16];
            auto row_of = [&](int r) {                     // (rows past M read the last row: a valid address, never stored)
                const long long m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                return m < M ? m : M - 1;
            };
            if (splitk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) yv[r] = __hip_atomic_load(y + (size_t)row_of(r) * p.ldy + cq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (p.accumulate) {
#pragma unroll
                for (int r = 0; r < 16; ++r) yv[r] = y[(size_t)row_of(r) * p.ldy + cq];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) yv[r] = 0.f;
            }
            if (p.res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = p.res[(size_t)row_of(r) * p.ldr + cq];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (c_ok && m < M) {
                    float* dst = y + (size_t)m * p.ldy + c;
                    float v = splitk ? yv[r] : acc[i][j][r] * p.alpha;
                    v += bias;
                    v = v * osc + osh;
                    v += rv[r];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (!splitk) v += yv[r];                // (accumulate; zeros otherwise)
                    *dst = v;
                    s1 += v;
                    s2 += v * v;
                }
            }
        }
        if (p.stats) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (half == 0 && c_ok) {
                double* st = stat_slot(p, stat_group(p, m0, M), blockIdx.x);       // (statistic groups: the launcher keeps a tile inside one group)
                atomicAdd(st + c, (double)s1);
                atomicAdd(st + p.Cout + c, (double)s2);
            }
        }
    }
        if (p.fin_scale) fused_bn_finalize(p, (unsigned)total_tiles);      // (a K split that finishes inside the launch: one last-arriving workgroup per tile gets here)
        return;
    } else {
    // ------------------------------------------------------------------ epilogue
    // lane = (pixel row px = lane & 31 of row tile i, half); accumulator quad g of column tile j = channels cb + 8 g .. + 3 of that pixel
    const int half = lane >> 5, px = lane & 31;
    const bool splitk = p.splitk > 1;
    const bool vec_ok = (p.ldy % 4) == 0 && ((reinterpret_cast<uintptr_t>(y) & 15) == 0) &&
                        (!p.res || ((p.ldr % 4) == 0 && (reinterpret_cast<uintptr_t>(p.res) & 15) == 0));
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int cb = n0 + (wn * TN + j) * 32 + 4 * half;
        float s1[16], s2[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) s1[k] = s2[k] = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const long long m = m0 + (wm * TM + i) * 32 + px;
            // (as in the epilogue above: the residual / old-output quads of this pixel are read first, together, from clamped addresses)
            const long long mc = m < M ? m : M - 1;
            f32x4 r4q[4], o4q[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) r4q[g] = o4q[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.res && vec_ok && !splitk) {
#pragma unroll
                for (int g = 0; g < 4; ++g) r4q[g] = *reinterpret_cast<const f32x4*>(p.res + (size_t)mc * p.ldr + (cb + 8 * g + 3 < p.Cout ? cb + 8 * g : 0));
            }
            if (p.accumulate && vec_ok && !splitk) {
#pragma unroll
                for (int g = 0; g < 4; ++g) o4q[g] = *reinterpret_cast<const f32x4*>(y + (size_t)mc * p.ldy + (cb + 8 * g + 3 < p.Cout ? cb + 8 * g : 0));
            }
            if (m < M) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = cb + 8 * g;
                    float* dst = y + (size_t)m * p.ldy + c0;
                    float v[4], bias[4], osc[4], osh[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool c_ok = c0 + e < p.Cout && !splitk;
                        bias[e] = (p.bias && c_ok) ? p.bias[c0 + e] : 0.f;
                        osc[e] = (p.out_scale && c_ok) ? p.out_scale[c0 + e] : 1.f;
                        osh[e] = (p.out_scale && c_ok) ? p.out_shift[c0 + e] : 0.f;
                        v[e] = acc[i][j][4 * g + e] * p.alpha;
                    }
                    if (splitk) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c0 + e < p.Cout) atomicAdd(dst + e, v[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (v[e] + bias[e]) * osc[e] + osh[e];
                        if (vec_ok && c0 + 3 < p.Cout) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += r4q[g][e];
                            if (p.relu) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += o4q[g][e];
                            *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
#pragma unroll
                            for (int e = 0; e < 4; ++e) { s1[4 * g + e] += v[e]; s2[4 * g + e] += v[e] * v[e]; }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (c0 + e < p.Cout) {
                                    float u = v[e];
                                    if (p.res) u += p.res[(size_t)m * p.ldr + c0 + e];
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
            }
        }
        if (p.stats && !splitk) {
            // per-channel sums over the 32 pixel lanes of a half: butterfly reduce-scatter (see conv_halo.hip); afterwards lane L holds
            // channel index kk = 8 b4 + 4 b3 + 2 b2 + b1 (bN = bit N of L)
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
            const int kk = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
            const int cch = cb + 8 * (kk >> 2) + (kk & 3);
            if ((lane & 1) == 0 && cch < p.Cout) {
                double* st = stat_slot(p, stat_group(p, m0, M), blockIdx.x);       // (statistic groups: the launcher keeps a tile inside one group)
                atomicAdd(st + cch, (double)s1[0]);
                atomicAdd(st + p.Cout + cch, (double)s2[0]);
            }
        }
    }
    if (p.fin_scale) fused_bn_finalize(p, (unsigned)total_tiles);
    }
}

}  // namespace

// called by mrfa_conv2d_nhwc when the split-operand mode is on and the launch is a chunked 128 x {128, 64} tile
int mrfa_conv_split_launch(hipStream_t st, const mrfa_conv_params& p, int KT, long long M, int splitk, int BN) {
    const int tiles_n = cdiv(p.Cout, BN);
    const long long tiles_m = (M + BM - 1) / BM;
    const int total_tiles = (int)(tiles_m * tiles_n);
    dim3 grid((unsigned)(cdiv(total_tiles, 8) * 8), (unsigned)(p.nbatch > 1 ? p.nbatch : 1), (unsigned)splitk);
    const int kps = cdiv(KT, splitk);
    mrfa_conv_params q = p;
    q.splitk = splitk;
    if (p.in_scale && !p.in_relu) { mrfa_set_error("conv2d(bf16x6): in_scale without in_relu is not used by the path"); return 1; }
    const bool three = mrfa_get_mfma_mode() == 2;        // bf16x3
    const bool one = mrfa_get_mfma_mode() == 3;          // plain bf16
#define SPLIT_LAUNCH_TR(PRO_, WS_, BN_, TR_)                                                                                                       \
    do {                                                                                                                                           \
        if (one) hipLaunchKernelGGL((conv_bf16x6_kernel<PRO_, false, BN_, 1, TR_>), grid, dim3(NT), 0, st, q, KT, kps, M, tiles_n, total_tiles);    \
        else if (three) hipLaunchKernelGGL((conv_bf16x6_kernel<PRO_, WS_, BN_, 3, TR_>), grid, dim3(NT), 0, st, q, KT, kps, M, tiles_n, total_tiles); \
        else hipLaunchKernelGGL((conv_bf16x6_kernel<PRO_, WS_, BN_, 6, TR_>), grid, dim3(NT), 0, st, q, KT, kps, M, tiles_n, total_tiles);          \
    } while (0)
#define SPLIT_LAUNCH(PRO_, WS_, BN_) do { if (splitk > 1) SPLIT_LAUNCH_TR(PRO_, WS_, BN_, false); else SPLIT_LAUNCH_TR(PRO_, WS_, BN_, true); } while (0)
    const bool pro = p.in_scale != nullptr, ws = p.w_split != nullptr;
    if (BN == 64) {
        if (pro && ws) SPLIT_LAUNCH(true, true, 64); else if (pro) SPLIT_LAUNCH(true, false, 64);
        else if (ws) SPLIT_LAUNCH(false, true, 64); else SPLIT_LAUNCH(false, false, 64);
    } else {
        if (pro && ws) SPLIT_LAUNCH(true, true, 128); else if (pro) SPLIT_LAUNCH(true, false, 128);
        else if (ws) SPLIT_LAUNCH(false, true, 128); else SPLIT_LAUNCH(false, false, 128);
    }
#undef SPLIT_LAUNCH
#undef SPLIT_LAUNCH_TR
    MRFA_CHECK_LAUNCH("mrfa_conv2d_nhwc(bf16x6)");
    return 0;
}
