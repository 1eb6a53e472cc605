// Weight gradient of the keypoint encoder's small-channel 3x3 layers (HRNet BasicBlock: hr_base.py:34-63: 32 -> 32 @64^2, 64 -> 64 @32^2,
// 128 -> 128 @16^2 over 8..24 frames), ALL NINE TAPS from one staging of the operands, on the split-operand pipe, many problems per launch:
//
//   dW[tap (r,s)][co][ci] += alpha * sum_{n,y,x} dY[n,y,x][co] * X'[n, y+r-1, x+s-1][ci]          X' = relu(in_scale x + in_shift) or x, zero outside
//
// wgrad_small.hip computes one tap and one 32 x 32 (64 x 64) block per wave on the fp32 matrix pipe: X and dY are read nine times from L1 / L2 and
// the ~112 such problems of a training step (the deferred weight gradients of the encoder's residual blocks) took 3.4 ms of kernel time, most of
// the step's 3.5 ms tail.  Here, as in wgrad_halo.hip (whose 128 x 64 blocks over 32-pixel columns of >= 8 rows these shapes cannot fill):
//   * a workgroup of FOUR waves walks a run of small 2-D patches of one problem.  Per patch the X' halo and the dY pixels are loaded (the loads of
//     the next patch are in flight while this one multiplies), run through the optional BatchNorm-apply + ReLU prologue (per statistic group),
//     split exactly into three bf16 pieces and stored PIXEL-major in LDS ([32-channel block][pixel][64 B]);
//   * the K-major MFMA fragments (8 consecutive pixels of one channel per lane) are formed by ds_read_b64_tr_b16, the transposing LDS read; a tap
//     (r, s) is a constant (r HP + s) * 64 byte offset of the X' fragment: one dY fragment per k16 step serves all nine taps;
//   * a wave owns the nine 32 (co) x 32 (ci) tap tiles of one block pair: 144 accumulator registers.  The four waves are 2 x 2 block pairs of a
//     64 x 64 weight block over the same pixels (64 / 128 channels), or four pixel tiles of the patch (32 channels: summed through LDS at the end);
//   * the problems of a launch travel as a table in the kernel arguments (mrfa_conv2d_wgrad_multi): a problem is cut into G pixel ranges so that
//     the launch has ~768 workgroups whatever the number of problems; 9 x 1 024 fp32 atomics per (block, range) into dW[tap][Cout][Cin].
// The prologue is what lets a residual block skip the BatchNorm-apply launch between its two convolutions (conv_lean.hip takes in_scale as well).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4_t;

constexpr int WL_MAXP = 24;                        // problems per launch

struct LeanW {
    const float* x; const float* dy; float* dw;
    const float* in_scale; const float* in_shift;
    int ldx, ldy, N, H, W, Cin, Cout, groups;
    float alpha;
    int G;                                         // pixel ranges (workgroups) per 64 x 64 block of the weights
    int Q;                                         // patches of the problem
};
struct LeanWMulti {
    int n;
    int prefix[WL_MAXP + 1];                       // first workgroup of problem i
    LeanW p[WL_MAXP];
};

__device__ __forceinline__ unsigned wl_pack_hi16(float a, float b) { return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u); }
__device__ __forceinline__ float wl_chop_rest(float x) { return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
__device__ __forceinline__ unsigned wl_rne16(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
template <int NP>
__device__ __forceinline__ void wl_pieces(const f32x4 v, u32x2& p1, u32x2& p2, u32x2& p3) {
    if constexpr (NP == 1) {
        p1[0] = wl_rne16(v.x) | (wl_rne16(v.y) << 16);
        p1[1] = wl_rne16(v.z) | (wl_rne16(v.w) << 16);
    } else {
        const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float a = xs[2 * h], b = xs[2 * h + 1];
            p1[h] = wl_pack_hi16(a, b);
            const float ar = wl_chop_rest(a), br = wl_chop_rest(b);
            p2[h] = wl_pack_hi16(ar, br);
            p3[h] = wl_pack_hi16(wl_chop_rest(ar), wl_chop_rest(br));
        }
    }
}

// transposing fragment read: pixels k0 .. k0+7 of this lane's channel column as one MFMA operand (two b64 reads; see wgrad_halo.hip)
__device__ __forceinline__ bf16x8 wl_tr_frag(const unsigned char* p) {
    typedef __attribute__((address_space(3))) bf16x4_t* lds4;
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(p));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(p + 4 * 64));
    bf16x8 r;
    __builtin_memcpy(&r, &lo, 8);
    __builtin_memcpy(reinterpret_cast<char*>(&r) + 8, &hi, 8);
    return r;
}

// NB: 32-channel blocks per side of the workgroup's weight block (1: 32 x 32, the four waves take four pixel tiles; 2: 64 x 64, one block pair per wave)
// TW: patch width (32: a pixel tile is one row of 32; 16: two rows of 16)
template <int NB, int TW, bool PRO, int NP>
__global__ __launch_bounds__(256, 1) void wgrad_lean_kernel(const LeanWMulti a) {
    constexpr int NPXW = NB == 1 ? 4 : 1;          // pixel tiles of a patch (= waves along pixels)
    constexpr int TR = 32 / TW, PR = NPXW * TR, HP = TW + 2, HPIX = (PR + 2) * HP, DPIX = NPXW * 32;
    constexpr int XBLK = HPIX * 64, DBLK = DPIX * 64;            // one 32-channel block of one piece
    constexpr int NPC = NP == 6 ? 3 : (NP == 3 ? 2 : 1);
    constexpr int XPLANE = NB * XBLK, DPLANE = NB * DBLK;
    constexpr int NXU = NB * HPIX * 8, NDU = NB * DPIX * 8;      // float4 units of a patch
    constexpr int NX = (NXU + 255) / 256, ND = (NDU + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // two image buffers [X' | dY] (patch q in buffer q & 1): the next patch is split and stored while this one multiplies -- ONE workgroup per CU
    // (144 accumulator + ~90 staging / fragment registers do not fit twice into a SIMD's file), so the overlap has to come from inside the wave
    constexpr int IMG = NPC * (XPLANE + DPLANE);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int pi = 0;
    while (pi + 1 < a.n && (int)blockIdx.x >= a.prefix[pi + 1]) ++pi;
    const LeanW& p = a.p[pi];
    const int local = (int)blockIdx.x - a.prefix[pi];
    const int nci = p.Cin / (32 * NB);
    const int bg = local / p.G, jr = local - bg * p.G;
    const int co0 = (bg / nci) * (32 * NB), ci0 = (bg % nci) * (32 * NB);
    const int q0 = (int)((long long)jr * p.Q / p.G), q1 = (int)((long long)(jr + 1) * p.Q / p.G);
    const int tiles_x = p.W / TW, tiles_y = (p.H + PR - 1) / PR;
    const int wpx = NB == 1 ? wave : 0, wco = NB == 1 ? 0 : (wave >> 1), wci = NB == 1 ? 0 : (wave & 1);

    // ---- staging units of this thread: unit u = tid + 256 j -> (32-channel block, halo pixel or patch pixel, channel quad); the same for every patch, and
    // recomputed from tid where needed (constant divisors) rather than kept: the nine accumulator tiles leave ~100 registers for everything else
    auto x_unit = [&](int j, int& blk, int& hp, int& quad) -> bool {
        const int u = tid + j * 256;
        const int uu = u < NXU ? u : 0;
        blk = uu / (HPIX * 8);
        const int rem = uu - blk * (HPIX * 8);
        hp = rem >> 3;
        quad = rem & 7;
        return u < NXU;
    };
    auto d_unit = [&](int j, int& blk, int& px, int& quad) -> bool {
        const int u = tid + j * 256;
        const int uu = u < NDU ? u : 0;
        blk = uu / (DPIX * 8);
        const int rem = uu - blk * (DPIX * 8);
        px = rem >> 3;
        quad = rem & 7;
        return u < NDU;
    };
    // prologue vectors of this workgroup's input channels, [groups][2][32 NB] floats in LDS behind the images (read at store time)
    float* const smP = reinterpret_cast<float*>(smem + 2 * IMG);
    if constexpr (PRO) {
        const int G = p.groups > 1 ? p.groups : 1;
        for (int i = tid; i < G * 2 * 32 * NB; i += 256) {
            const int g = i / (2 * 32 * NB), r = i - g * (2 * 32 * NB), which = r / (32 * NB), c = r - which * (32 * NB);
            smP[i] = (which ? p.in_shift : p.in_scale)[g * p.Cin + ci0 + c];
        }
        __syncthreads();
    }

    f32x4 rx[NX], rd[ND];
    unsigned okmask = 0;                           // bit j: X' unit j inside the image; bit 16 + j: dY unit j inside the image
    int cur_grp = 0;
    auto load_patch = [&](int q) {                 // q clamped by the caller: loads past the run re-read its last patch
        const int tx = q % tiles_x;
        int t = q / tiles_x;
        const int ty = t % tiles_y, n = t / tiles_y;
        const int y0 = ty * PR, x0 = tx * TW;
        const float* xb = p.x + (size_t)n * p.H * p.W * p.ldx;
        const float* db = p.dy + (size_t)n * p.H * p.W * p.ldy;
        okmask = 0;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            int blk, hp, quad;
            const bool val = x_unit(j, blk, hp, quad);
            const int hy = hp / HP, hx = hp - hy * HP;
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            const bool ok = val && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            okmask |= ok ? (1u << j) : 0u;
            rx[j] = *reinterpret_cast<const f32x4*>(xb + (size_t)(ok ? iy * p.W + ix : 0) * p.ldx + ci0 + blk * 32 + quad * 4);
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            int blk, px, quad;
            const bool val = d_unit(j, blk, px, quad);
            const int r = px / TW, c = px - r * TW;
            const int oy = y0 + r;
            const bool ok = val && oy < p.H;
            okmask |= ok ? (1u << (16 + j)) : 0u;
            rd[j] = *reinterpret_cast<const f32x4*>(db + (size_t)(ok ? oy * p.W + x0 + c : 0) * p.ldy + co0 + blk * 32 + quad * 4);
        }
        cur_grp = p.groups > 1 ? n / (p.N / p.groups) : 0;
    };
    // unit k of the combined list (X' units 0 .. NX-1, dY units NX .. NX+ND-1) of the patch in the registers -> image buffer `buf`
    auto store_unit = [&](int k, int buf) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        unsigned char* const smX = smem + buf * IMG;
        unsigned char* const smD = smX + NPC * XPLANE;
        if (k < NX) {
            const int j = k;
            int blk, hp, quad;
            const bool val = x_unit(j, blk, hp, quad);
            f32x4 v = rx[j];
            if constexpr (PRO) {
                const float* ps = smP + cur_grp * (2 * 32 * NB) + blk * 32 + quad * 4;
                const f32x4 psc = *reinterpret_cast<const f32x4*>(ps), psh = *reinterpret_cast<const f32x4*>(ps + 32 * NB);
                v = v * psc + psh;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            v = (okmask >> j) & 1u ? v : z;
            u32x2 p1, p2, p3;
            wl_pieces<NP>(v, p1, p2, p3);
            if (val) {
                unsigned char* dst = smX + blk * XBLK + hp * 64 + quad * 8;
                *reinterpret_cast<u32x2*>(dst) = p1;
                if constexpr (NPC >= 2) *reinterpret_cast<u32x2*>(dst + XPLANE) = p2;
                if constexpr (NPC == 3) *reinterpret_cast<u32x2*>(dst + 2 * XPLANE) = p3;
            }
        } else {
            const int j = k - NX;
            int blk, px, quad;
            const bool val = d_unit(j, blk, px, quad);
            const f32x4 v = (okmask >> (16 + j)) & 1u ? rd[j] : z;
            u32x2 p1, p2, p3;
            wl_pieces<NP>(v, p1, p2, p3);
            if (val) {
                unsigned char* dst = smD + blk * DBLK + px * 64 + quad * 8;
                *reinterpret_cast<u32x2*>(dst) = p1;
                if constexpr (NPC >= 2) *reinterpret_cast<u32x2*>(dst + DPLANE) = p2;
                if constexpr (NPC == 3) *reinterpret_cast<u32x2*>(dst + 2 * DPLANE) = p3;
            }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposing-read lane offset: 16-lane group g = lane >> 4: channels 16 (g & 1) .., pixels 8 (g >> 1) ..; lane l of the group supplies the address of
    // (pixel l / 4, channels 4 (l % 4) ..) of a [4 pixel][16 channel] block
    const int l16 = lane & 15, grp16 = lane >> 4;
    const int frag_off = (8 * (grp16 >> 1) + (l16 >> 2)) * 64 + (grp16 & 1) * 32 + (l16 & 3) * 8;
    const int a_off = wco * DBLK + wpx * (32 * 64) + frag_off;                    // + kk * 16 * 64
    const int b_off = wci * XBLK + frag_off;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
    constexpr int TG = 3, NGRP = 3;                // taps per fragment group, groups per k16 step
    constexpr int NUNIT = NX + ND, UPG = (NUNIT + 2 * NGRP - 1) / (2 * NGRP);      // staging units stored behind each of the 2 x 3 tap groups
    // patch in buffer `buf` x this wave's nine tap tiles; `stage`: the next patch (in the registers) goes to the other buffer, a few units per tap group
    auto compute = [&](int buf, bool stage) {
        const unsigned char* const smX = smem + buf * IMG;
        const unsigned char* const smD = smX + NPC * XPLANE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // k16 step kk of this wave's pixel tile: 16 consecutive halo slots per tap row
            const int row0 = TW == 32 ? wpx : wpx * 2 + kk, col0 = TW == 32 ? 16 * kk : 0;
            bf16x8 af[NPC];
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) af[pc] = wl_tr_frag(smD + pc * DPLANE + a_off + kk * 16 * 64);
#pragma unroll
            for (int gi = 0; gi < NGRP; ++gi) {
                const int g0 = gi * TG;
                bf16x8 bfr[NPC][TG];
#pragma unroll
                for (int q = 0; q < TG; ++q) {
                    const int tap = g0 + q, r = tap / 3, s = tap - 3 * r;
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc) bfr[pc][q] = wl_tr_frag(smX + pc * XPLANE + b_off + ((row0 + r) * HP + col0 + s) * 64);
                }
#pragma unroll
                for (int t = 6 - NP; t < 6; ++t)
#pragma unroll
                    for (int q = 0; q < TG; ++q) acc[g0 + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[t]], bfr[PB[t]][q], acc[g0 + q], 0, 0, 0);
                if (stage) {
#pragma unroll
                    for (int k = (kk * NGRP + gi) * UPG; k < (kk * NGRP + gi + 1) * UPG; ++k)
                        if (k < NUNIT) store_unit(k, buf ^ 1);
                }
            }
        }
    };

    if (q0 < q1) {
        load_patch(q0);
#pragma unroll
        for (int k = 0; k < NUNIT; ++k) store_unit(k, q0 & 1);
        if (q0 + 1 < q1) load_patch(q0 + 1);
        __syncthreads();
        for (int q = q0; q < q1; ++q) {
            const bool more = q + 1 < q1;
            compute(q & 1, more);                  // (the registers hold patch q + 1: stored into the other buffer, last read as patch q - 1)
            if (q + 2 < q1) load_patch(q + 2);     // in flight over the barrier and the first tap groups of the next patch
            __syncthreads();
        }
    }

    // ------------------------------------------------------------------ epilogue
    // lane = (ci = lane & 31, half); accumulator quad g of a tap tile = output channels 8 g + 4 half .. + 3
    if constexpr (NB == 1) {
        // the four pixel tiles of the patch hold partial sums of the SAME 32 x 32 block: waves 1..3 hand theirs to wave 0 through LDS, one at a time
        float* red = reinterpret_cast<float*>(smem);            // [9][16][64]
        for (int w = 1; w < 4; ++w) {
            __syncthreads();
            if (wave == w) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = acc[t][r];
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += red[(t * 16 + r) * 64 + lane];
            }
        }
        if (wave != 0) return;
    }
    if (q0 >= q1) return;
    const int ci = ci0 + wci * 32 + (lane & 31), half = lane >> 5;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co0 + wco * 32 + 8 * g + 4 * half + e;
                atomicAdd(p.dw + ((size_t)t * p.Cout + co) * p.Cin + ci, p.alpha * acc[t][4 * g + e]);
            }
}

int g_wgrad_lean_on = -1;

template <int NB, int TW>
constexpr size_t wl_lds(int npc) {
    constexpr int NPXW = NB == 1 ? 4 : 1, TR = 32 / TW, PR = NPXW * TR, HP = TW + 2, HPIX = (PR + 2) * HP, DPIX = NPXW * 32;
    const size_t st = (size_t)2 * npc * NB * (HPIX + DPIX) * 64 + (size_t)3 * 2 * 32 * NB * 4, red = NB == 1 ? (size_t)9 * 16 * 64 * 4 : 0;  // (two image buffers + prologue vectors of <= 3 groups)
    return st > red ? st : red;
}

// geometry of a problem: 0: 32 channels (32 x 32 block, 4-row patches of 32), 1: 64-channel blocks on 32-wide patches, 2: on 16-wide patches
int wl_variant(const mrfa_wgrad_params& p) {
    if (p.Cin == 32 && p.Cout == 32) return (p.Wout % 32) == 0 ? 0 : -1;
    if ((p.Cin % 64) || (p.Cout % 64)) return -1;
    return (p.Wout % 32) == 0 ? 1 : ((p.Wout % 16) == 0 ? 2 : -1);
}

template <int NB, int TW>
int wl_launch(hipStream_t st, const LeanWMulti& m, bool pro, int mode) {
    dim3 grid((unsigned)m.prefix[m.n]);
#define WL(PRO_, NP_)                                                                                                                              \
    do {                                                                                                                                          \
        constexpr size_t lds = wl_lds<NB, TW>(NP_ == 6 ? 3 : (NP_ == 3 ? 2 : 1));                                                                  \
        if constexpr (lds > 65536) {                /* (more than 64 KB of dynamic LDS must be asked for once per kernel) */                        \
            static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_lean_kernel<NB, TW, PRO_, NP_>),               \
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                             \
            if (attr != hipSuccess) { mrfa_set_error("wgrad(lean): %zu bytes of LDS refused: %s", lds, hipGetErrorString(attr)); return 2; }      \
        }                                                                                                                                         \
        hipLaunchKernelGGL((wgrad_lean_kernel<NB, TW, PRO_, NP_>), grid, dim3(256), lds, st, m);                                                  \
    } while (0)
    if (mode == 3) { if (pro) WL(true, 1); else WL(false, 1); }
    else if (mode == 2) { if (pro) WL(true, 3); else WL(false, 3); }
    else { if (pro) WL(true, 6); else WL(false, 6); }
#undef WL
    return 0;
}

}  // namespace

int mrfa_tuning_wgrad_lean(int set) {
    if (g_wgrad_lean_on < 0) { const char* e = getenv("MRFA_WGRAD_LEAN"); g_wgrad_lean_on = !(e && e[0] == '0'); }
    const int prev = g_wgrad_lean_on;
    if (set >= 0) g_wgrad_lean_on = set != 0;
    return prev;
}

// 1: the problem runs here: 3x3 / pad 1 / stride 1, 32 -> 32 channels or 64-aligned channel counts up to 128, 16-byte addressable rows, no bias gradient
bool mrfa_wgrad_lean_eligible(const mrfa_wgrad_params& p) {
    const int mode = mrfa_get_mfma_mode();
    if (!mrfa_tuning_wgrad_lean(-1) || (mode != 1 && mode != 2 && mode != 3)) return false;
    if (p.kflat > 0 || p.ups || p.R != 3 || p.S != 3 || p.pad != 1 || p.nbatch > 1 || p.ksplit > 0 || p.stride > 1 || p.dbias) return false;
    if (p.Hout != p.Hin || p.Wout != p.Win || p.Cin > 128 || p.Cout > 128) return false;
    if ((p.ldx % 4) != 0 || !aligned16(p.x) || (p.ldy % 4) != 0 || !aligned16(p.dy)) return false;
    if (p.in_scale && (!p.in_relu || !aligned16(p.in_scale) || !aligned16(p.in_shift))) return false;
    if (p.groups > 1 && ((p.N % p.groups) != 0 || p.groups > 3)) return false;
    if ((long long)p.N * p.Hin * p.Win * p.ldx >= (1ll << 31) || (long long)p.N * p.Hout * p.Wout * p.ldy >= (1ll << 31)) return false;
    if (2.0 * (double)p.N * p.Hout * p.Wout * p.Cout * 9.0 * p.Cin > 2.6e9) return false;
    return wl_variant(p) >= 0;
}

// the eligible problems of ps[0..n) (taken[i] set for each), grouped by geometry and prologue, WL_MAXP per launch
int mrfa_wgrad_lean_multi(hipStream_t st, const mrfa_wgrad_params* ps, int n, unsigned char* taken) {
    const int mode = mrfa_get_mfma_mode();
    int neligible = 0;
    for (int i = 0; i < n; ++i) neligible += (taken[i] = mrfa_wgrad_lean_eligible(ps[i]) ? 1 : 0);
    if (!neligible) return 0;
    for (int v = 0; v < 3; ++v)
        for (int pro = 0; pro < 2; ++pro) {
            int idx[4096], cnt = 0;
            for (int i = 0; i < n && cnt < 4096; ++i)
                if (taken[i] && wl_variant(ps[i]) == v && (ps[i].in_scale != nullptr) == (pro != 0)) idx[cnt++] = i;
            for (int b0 = 0; b0 < cnt; b0 += WL_MAXP) {
                const int nb = cnt - b0 < WL_MAXP ? cnt - b0 : WL_MAXP;
                LeanWMulti m;
                m.n = nb;
                m.prefix[0] = 0;
                // ~512 workgroups per launch (two rounds of one per CU), at least one and at most 64 pixel ranges per weight block
                long long blocks = 0;
                for (int k = 0; k < nb; ++k) {
                    const mrfa_wgrad_params& q = ps[idx[b0 + k]];
                    blocks += v == 0 ? 1 : (q.Cin / 64) * (q.Cout / 64);
                }
                int G = (int)((512 + blocks - 1) / blocks);
                if (G > 64) G = 64;
                for (int k = 0; k < nb; ++k) {
                    const mrfa_wgrad_params& q = ps[idx[b0 + k]];
                    LeanW& w = m.p[k];
                    const int PR = v == 0 ? 4 : (v == 1 ? 1 : 2), TW = v == 2 ? 16 : 32;
                    w.x = q.x; w.dy = q.dy; w.dw = q.dw; w.in_scale = q.in_scale; w.in_shift = q.in_shift;
                    w.ldx = q.ldx; w.ldy = q.ldy; w.N = q.N; w.H = q.Hout; w.W = q.Wout; w.Cin = q.Cin; w.Cout = q.Cout; w.groups = q.groups;
                    w.alpha = q.alpha;
                    w.Q = q.N * (q.Wout / TW) * cdiv(q.Hout, PR);
                    w.G = G < w.Q ? G : w.Q;
                    const int nblocks = v == 0 ? 1 : (q.Cin / 64) * (q.Cout / 64);
                    m.prefix[k + 1] = m.prefix[k] + w.G * nblocks;
                }
                int rc = 0;
                if (v == 0) rc = wl_launch<1, 32>(st, m, pro != 0, mode);
                else if (v == 1) rc = wl_launch<2, 32>(st, m, pro != 0, mode);
                else rc = wl_launch<2, 16>(st, m, pro != 0, mode);
                if (rc) return rc;
                MRFA_CHECK_LAUNCH("mrfa_conv2d_wgrad_multi(lean)");
            }
        }
    return 0;
}
