// 3x3 / pad 1 convolutions with ONE or TWO output channels over 64 / 128 / 256 input channels (refine.conv2 128->2, refine.convo2 128->1
// at all six pyramid levels, raft.py:76,78): data gradient and (one output channel) weight gradient with the CHANNELS across the lanes.
//
// These layers are pure HBM streams (2.4 GFLOP against 268 MB of activations at 256^2, B = 8).  Their data gradient took the generic MFMA
// route with K = 9 * Cout = 18 (336 us at 256^2 = 0.8 TB/s of dx writes); here it takes 150 us.  (A forward kernel of the same shape was
// measured equal to conv_fewout.hip's tile kernel -- 135 vs 133 us -- and dropped.)  LPP = Cin / 4 lanes form a pixel group (a wave holds 64 / LPP groups), every lane keeps the
// weights of its four channels for all nine taps in registers (72 VGPRs for two output channels), and a group slides along an image
// row with a 3 x 3 window of float4 registers: three new 16-byte loads per pixel (each a contiguous 4 * Cin-byte row segment across the
// group), 72 FMAs, one LPP-lane butterfly per output channel.  No LDS, no barrier, every input byte is requested three times (rows
// y-1, y, y+1) from L1 / L2.
//   dgrad     dx[p][c]  (+)= sum_{tap,co} dy[p - tap][co] w[co][tap][c]          (window of 9 x Cout broadcast scalars)
//   wgrad     dw[co][tap][c] += sum_p dy[p][co] x[p + tap][c]                     (accumulators in registers, one LDS reduction and one set
//                                                                                  of atomics per workgroup), dbias[co] += sum_p dy[p][co]
#include "common.h"

namespace {

constexpr int RUN = 16;            // pixels a group walks per work unit

struct Unit {                      // (image n, row y, first column x0) of work unit u
    int n, y, x0;
};
__device__ __forceinline__ Unit unit_of(long long u, int H, int runs_x) {
    const int rx = (int)(u % runs_x);
    const long long t = u / runs_x;
    return Unit{(int)(t / H), (int)(t % H), rx * RUN};
}


// column x of rows y-1, y, y+1 (zero outside the image); `base` = first float of this lane's channel quad in image n
__device__ __forceinline__ void load_col(const float* __restrict__ base, int ld, int H, int W, int y, int x, f32x4 (&col)[3]) {
    const bool xok = (unsigned)x < (unsigned)W;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yy = y + r - 1;
        const bool ok = xok && (unsigned)yy < (unsigned)H;
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (ok ? ((size_t)yy * W + x) * ld : (size_t)0));     // clamped address, selected value
        col[r] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// RD: the launch reads a ReLU mask and / or the old gradient (their batched read-ahead costs 32 registers: kept out of the launches that read neither)
template <int COUT, int LPP, bool RD>
__global__ __launch_bounds__(256) void fewout3_dgrad_kernel(const float* __restrict__ dy, int lddy, int N, int H, int W, int Cin,
                                                           const float* __restrict__ w, float* __restrict__ dx, int lddx, int accumulate,
                                                           long long units, int runs_x, const float* __restrict__ mask, int ldm) {
    constexpr int G = 64 / LPP;
    const int lane = threadIdx.x & 63, cl = lane % LPP, grp = lane / LPP;
    const int c = cl * 4;
    f32x4 wr[COUT][9];
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[co][t] = *reinterpret_cast<const f32x4*>(w + ((size_t)co * 9 + t) * Cin + c);
    const long long gid = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * G + grp;
    const long long gstep = (long long)gridDim.x * (blockDim.x >> 6) * G;
    for (long long u = gid; u < units; u += gstep) {
        const Unit ut = unit_of(u, H, runs_x);
        const float* gb = dy + (size_t)ut.n * H * W * lddy;
        // dx[y][x] = sum_{r,s} dy[y + 1 - r][x + 1 - s] w[r][s]: window of dy columns x-1, x, x+1 (rows y-1, y, y+1), all lanes of the group
        // read the same addresses (one broadcast transaction)
        float win[3][3][COUT];
        auto load_dcol = [&](int xx, float (&col)[3][COUT]) {
            const bool xok = (unsigned)xx < (unsigned)W;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int yy = ut.y + r - 1;
                const bool ok = xok && (unsigned)yy < (unsigned)H;
                const float* pp = gb + (ok ? ((size_t)yy * W + xx) * lddy : (size_t)0);
#pragma unroll
                for (int co = 0; co < COUT; ++co) { const float v = pp[co]; col[r][co] = ok ? v : 0.f; }
            }
        };
        load_dcol(ut.x0 - 1, win[0]);
        load_dcol(ut.x0, win[1]);
        // what four pixels' stores need (ReLU mask, old gradient of an accumulating launch) is read together, ahead of them, from clamped addresses --
        // inside the `xo < W` branch each read was a round trip of its own per pixel (round 6)
        const size_t pix0 = (size_t)(ut.n * H + ut.y) * W;
#pragma unroll
        for (int i0 = 0; i0 < RUN; i0 += 4) {
            f32x4 mk[RD ? 4 : 1], old[RD ? 4 : 1];
            if constexpr (RD) {
            if (mask) {
#pragma unroll
                for (int q = 0; q < 4; ++q) mk[q] = *reinterpret_cast<const f32x4*>(mask + (pix0 + min(ut.x0 + i0 + q, W - 1)) * ldm + c);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) mk[q] = f32x4{1.f, 1.f, 1.f, 1.f};
            }
            if (accumulate) {
#pragma unroll
                for (int q = 0; q < 4; ++q) old[q] = *reinterpret_cast<const f32x4*>(dx + (pix0 + min(ut.x0 + i0 + q, W - 1)) * lddx + c);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) old[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q;
                const int xo = ut.x0 + i;
                float(&c0)[3][COUT] = win[i % 3];
                float(&c1)[3][COUT] = win[(i + 1) % 3];
                float(&c2)[3][COUT] = win[(i + 2) % 3];
                load_dcol(xo + 1, c2);
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int co = 0; co < COUT; ++co)
#pragma unroll
                    for (int r = 0; r < 3; ++r)          // window row r' = y - 1 + r' is the source of tap row r = 2 - r'; column slot likewise
                        a += wr[co][(2 - r) * 3 + 2] * c0[r][co] + wr[co][(2 - r) * 3 + 1] * c1[r][co] + wr[co][(2 - r) * 3 + 0] * c2[r][co];
                if (xo < W) {
                    float* d = dx + (pix0 + xo) * lddx + c;
                    if constexpr (RD) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) a[e] = mk[q][e] > 0.f ? a[e] : 0.f;     // fused ReLU backward of the producer of x (mrfa_conv_params.mask semantics)
                        a += old[q];
                    }
                    *reinterpret_cast<f32x4*>(d) = a;
                }
            }
        }
    }
}

template <int COUT, int LPP>
__global__ __launch_bounds__(256) void fewout3_wgrad_kernel(const float* __restrict__ x, int ldx, int N, int H, int W, int Cin,
                                                           const float* __restrict__ dy, int lddy, float* __restrict__ dw,
                                                           float* __restrict__ dbias, long long units, int runs_x) {
    constexpr int G = 64 / LPP;
    __shared__ float red[COUT * 9 * 256 + 4];        // [co][tap][channel] partial sums of the workgroup (Cin <= 256) + bias sums
    const int lane = threadIdx.x & 63, cl = lane % LPP, grp = lane / LPP;
    const int c = cl * 4;
    f32x4 acc[COUT][9];
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[co][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bs[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) bs[co] = 0.f;
    const long long gid = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * G + grp;
    const long long gstep = (long long)gridDim.x * (blockDim.x >> 6) * G;
    for (long long u = gid; u < units; u += gstep) {
        const Unit ut = unit_of(u, H, runs_x);
        const float* base = x + (size_t)ut.n * H * W * ldx + c;
        const float* gb = dy + ((size_t)(ut.n * H + ut.y) * W) * lddy;
        f32x4 win[3][3];
        load_col(base, ldx, H, W, ut.y, ut.x0 - 1, win[0]);
        load_col(base, ldx, H, W, ut.y, ut.x0, win[1]);
#pragma unroll
        for (int i = 0; i < RUN; ++i) {
            const int xo = ut.x0 + i;
            f32x4(&c0)[3] = win[i % 3];
            f32x4(&c1)[3] = win[(i + 1) % 3];
            f32x4(&c2)[3] = win[(i + 2) % 3];
            load_col(base, ldx, H, W, ut.y, xo + 1, c2);
            const bool ok = xo < W;
            const float* gp = gb + (size_t)(ok ? xo : 0) * lddy;
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const float g = ok ? gp[co] : 0.f;
                bs[co] += g;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    acc[co][r * 3 + 0] += c0[r] * g;
                    acc[co][r * 3 + 1] += c1[r] * g;
                    acc[co][r * 3 + 2] += c2[r] * g;
                }
            }
        }
    }
    // workgroup reduction through LDS (the groups of a wave and the four waves hold sums over different pixels), then one atomic per weight
    for (int i = threadIdx.x; i < COUT * 9 * Cin + 4; i += blockDim.x) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(red + (co * 9 + t) * Cin + c + e, acc[co][t][e]);
    if (cl == 0) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) atomicAdd(red + COUT * 9 * Cin + co, bs[co]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < COUT * 9 * Cin; i += blockDim.x) atomicAdd(dw + i, red[i]);
    if (dbias && threadIdx.x < COUT) atomicAdd(dbias + threadIdx.x, red[COUT * 9 * Cin + threadIdx.x]);
}

int g_fewout3_on = 1;

bool eligible(const float* x, int ldx, int Cin, int Cout, int R, int pad, int W) {
    return g_fewout3_on && R == 3 && pad == 1 && (Cout == 1 || Cout == 2) && (Cin == 64 || Cin == 128 || Cin == 256) && (ldx % 4) == 0 &&
           aligned16(x) && W >= 4;
}

long long grid_for(long long units, int lpp) {
    const long long groups_per_wg = 4ll * (64 / lpp);
    long long wgs = (units + groups_per_wg - 1) / groups_per_wg;
    if (wgs > 256 * 8) wgs = 256 * 8;              // grid-stride beyond 8 workgroups per CU
    return wgs < 1 ? 1 : wgs;
}

}  // namespace

int mrfa_tuning_fewout3(int set) {
    const int prev = g_fewout3_on;
    if (set >= 0) g_fewout3_on = set != 0;
    return prev;
}

#define FEW3R(KERNEL, RD, ...)                                                                                          \
    do {                                                                                                                \
        const int lpp = Cin / 4;                                                                                        \
        const dim3 grid((unsigned)grid_for(units, lpp));                                                                \
        if (Cout == 1) {                                                                                                \
            if (lpp == 16) hipLaunchKernelGGL((KERNEL<1, 16, RD>), grid, dim3(256), 0, st, __VA_ARGS__);                 \
            else if (lpp == 32) hipLaunchKernelGGL((KERNEL<1, 32, RD>), grid, dim3(256), 0, st, __VA_ARGS__);            \
            else hipLaunchKernelGGL((KERNEL<1, 64, RD>), grid, dim3(256), 0, st, __VA_ARGS__);                           \
        } else {                                                                                                        \
            if (lpp == 16) hipLaunchKernelGGL((KERNEL<2, 16, RD>), grid, dim3(256), 0, st, __VA_ARGS__);                 \
            else if (lpp == 32) hipLaunchKernelGGL((KERNEL<2, 32, RD>), grid, dim3(256), 0, st, __VA_ARGS__);            \
            else hipLaunchKernelGGL((KERNEL<2, 64, RD>), grid, dim3(256), 0, st, __VA_ARGS__);                           \
        }                                                                                                               \
    } while (0)
#define FEW3(KERNEL, ...)                                                                                               \
    do {                                                                                                                \
        const int lpp = Cin / 4;                                                                                        \
        const dim3 grid((unsigned)grid_for(units, lpp));                                                                \
        if (Cout == 1) {                                                                                                \
            if (lpp == 16) hipLaunchKernelGGL((KERNEL<1, 16>), grid, dim3(256), 0, st, __VA_ARGS__);                     \
            else if (lpp == 32) hipLaunchKernelGGL((KERNEL<1, 32>), grid, dim3(256), 0, st, __VA_ARGS__);                \
            else hipLaunchKernelGGL((KERNEL<1, 64>), grid, dim3(256), 0, st, __VA_ARGS__);                               \
        } else {                                                                                                        \
            if (lpp == 16) hipLaunchKernelGGL((KERNEL<2, 16>), grid, dim3(256), 0, st, __VA_ARGS__);                     \
            else if (lpp == 32) hipLaunchKernelGGL((KERNEL<2, 32>), grid, dim3(256), 0, st, __VA_ARGS__);                \
            else hipLaunchKernelGGL((KERNEL<2, 64>), grid, dim3(256), 0, st, __VA_ARGS__);                               \
        }                                                                                                               \
    } while (0)

bool mrfa_fewout3_wgrad(hipStream_t st, const float* x, int ldx, int N, int H, int W, int Cin, const float* dy, int lddy, int Cout, int R, int pad,
                        float* dw, float* dbias, int* rc) {
    // one output channel only: with two, the 72 accumulator + 36 window registers leave one wave per SIMD and the tile kernel of
    // conv_fewout.hip wins (128->2 @256^2: 332 vs 252 us; 128->1: 221 vs 288 us)
    if (!eligible(x, ldx, Cin, Cout, R, pad, W) || Cout != 1) return false;
    const int runs_x = cdiv(W, RUN);
    const long long units = (long long)N * H * runs_x;
    FEW3(fewout3_wgrad_kernel, x, ldx, N, H, W, Cin, dy, lddy, dw, dbias, units, runs_x);
    *rc = hipGetLastError() == hipSuccess ? 0 : 2;
    if (*rc) mrfa_set_error("conv_fewout_wgrad(3x3 channel-lane): launch failed");
    return true;
}

extern "C" int mrfa_conv_fewout_dgrad(void* stream, const float* dy, int lddy, int N, int H, int W, int Cout, const float* w, float* dx, int lddx,
                                      int Cin, int R, int pad, int accumulate, const float* mask, int ldm) {
    MRFA_CHECK_ARG(dy && w && dx, "conv_fewout_dgrad: null pointer");
    MRFA_CHECK_ARG(eligible(dx, lddx, Cin, Cout, R, pad, W) && aligned16(w),
                   "conv_fewout_dgrad: 3x3 / pad 1, Cout in {1, 2}, Cin in {64, 128, 256}, 16-B aligned dx / w with lddx %% 4 == 0 (got Cin %d Cout %d R %d)",
                   Cin, Cout, R);
    hipStream_t st = (hipStream_t)stream;
    const int runs_x = cdiv(W, RUN);
    const long long units = (long long)N * H * runs_x;
    MRFA_CHECK_ARG(!mask || ((ldm % 4) == 0 && aligned16(mask)), "conv_fewout_dgrad: mask must be a 16-B aligned view with ldm %% 4 == 0");
    if (mask || accumulate) FEW3R(fewout3_dgrad_kernel, true, dy, lddy, N, H, W, Cin, w, dx, lddx, accumulate, units, runs_x, mask, ldm);
    else FEW3R(fewout3_dgrad_kernel, false, dy, lddy, N, H, W, Cin, w, dx, lddx, accumulate, units, runs_x, mask, ldm);
    MRFA_CHECK_LAUNCH("conv_fewout_dgrad");
    return 0;
}

extern "C" int mrfa_conv_fewout_dgrad_supported(int Cin, int Cout, int R, int pad, int W, int lddx) {
    return g_fewout3_on && R == 3 && pad == 1 && (Cout == 1 || Cout == 2) && (Cin == 64 || Cin == 128 || Cin == 256) && (lddx % 4) == 0 && W >= 4;
}
