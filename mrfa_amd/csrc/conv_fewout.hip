// Direct (VALU) convolution kernels for layers with <= 4 output channels, where a 32-wide MFMA tile would waste >= 8x:
//   generator.final 64->3 7x7, refine.conv2 128->2, refine.convo2 128->1, dense_motion.occlusion 108->1 7x7
//   (modules/generator.py:32, raft.py:76,78, dense_motion.py:25) and the data gradient of corr_enc.convf1 (2 input
//   channels, raft.py:56), which is a 128->2 7x7 convolution over dY.
//
// forward  (K4 of SURVEY 2.2): one workgroup = a 16x16 output tile of one image; the input halo tile is staged in LDS
//   in 16-channel chunks (pixel stride 20 floats => the 16 lanes of a ds_read_b128 group hit distinct 16-B slots),
//   every thread owns one output pixel and all Cout accumulators; weights are wave-uniform (scalar loads).
// wgrad: persistent workgroups walk their share of 16x16 tiles; a thread owns one (tap, 4-channel group) of the
//   current channel chunk and accumulates dW[co][tap][c..c+3] over the tile's pixels from LDS (x halo tile + dY tile),
//   so only one atomic per weight per workgroup per chunk is issued.
// Both are bound by LDS / VALU issue, not HBM: every input element is fetched from HBM once.
#include "common.h"

namespace {

constexpr int FT = 16;          // output tile edge
constexpr int FCC = 16;         // channels per LDS chunk
constexpr int FPS = FCC + 4;    // padded pixel stride in floats (20): conflict-free b128 reads of consecutive pixels

// PY = output rows per thread (the workgroup's tile is 16 x 16 PY outputs): the PY outputs of a thread lie BELOW each other, so that input row i of the halo
// tile serves tap row i of the first, i - 1 of the second, ... -- R + PY - 1 LDS reads and R weight loads per (tap column, 4-channel group) for PY R
// multiply-add groups instead of R reads and R loads for R.  The 7x7 layers of the path (generator.final 64 -> 3, the data gradient of corr_enc.convf1 =
// a 128 -> 2 7x7 convolution over dY: 0.37 and 0.52 ms per step at 256^2) issue one LDS read and one scalar weight load per 8-12 multiply-adds with PY = 1 and
// are bound by those, not by the multiply-adds; consecutive lanes still read consecutive pixels (the conflict-free 20-float pixel stride).
template <int COUT, int PY, int RT>
__global__ __launch_bounds__(256) void conv_fewout_fwd_kernel(const float* __restrict__ x, int ldx, int H, int W, int Cin,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ y, int ldy, int Ho, int Wo, int Rdyn, int pad,
                                                             int accumulate, int tiles_x, int tiles_y, int cper) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int R = RT > 0 ? RT : Rdyn;                            // RT: the kernel size at compile time (7: the tap loops unroll, weight loads are hoisted)
    const int HTW = FT + R - 1, HTH = FT * PY + R - 1;           // halo tile: columns, rows
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int bt = blockIdx.x;
    const int n = bt / (tiles_x * tiles_y);
    const int trem = bt - n * tiles_x * tiles_y;
    const int ty0 = (trem / tiles_x) * FT * PY, tx0 = (trem % tiles_x) * FT;
    const int T = R * R;
    float acc[PY][COUT];
#pragma unroll
    for (int j = 0; j < PY; ++j)
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[j][c] = 0.f;
    const float* xin = x + (size_t)n * H * W * ldx;
    // gridDim.y > 1: the channel chunks are split over blockIdx.y and the partial sums meet in y through atomics (y holds the bias /
    // the running sum already).  Low-resolution levels have 8-32 tiles, and one workgroup walking all Cin x R x R taps alone is a
    // 180-220 us chain of dependent LDS / scalar loads whatever the image size (measured per level of the refinement pyramid).
    const int cbeg = blockIdx.y * cper;
    const int cend = min(Cin, cbeg + cper);

    for (int c0 = cbeg; c0 < cend; c0 += FCC) {
        __syncthreads();
        // stage the halo tile: HTH*HTW pixels x 4 float4
        for (int i = tid; i < HTH * HTW * 4; i += 256) {
            const int c4 = i & 3, pix = i >> 2;
            const int hy = pix / HTW, hx = pix - hy * HTW;
            const int iy = ty0 + hy - pad, ix = tx0 + hx - pad;
            // branch-free: an out-of-image / past-Cin element reads a valid address and is zeroed by a select (a load inside a divergent
            // branch is waited for at the join, one serial round trip per trip of this loop)
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && c0 + c4 * 4 < Cin;
            const f32x4 ld = *reinterpret_cast<const f32x4*>(xin + (ok ? ((size_t)iy * W + ix) * ldx + c0 + c4 * 4 : (size_t)0));
            *reinterpret_cast<f32x4*>(smem + pix * FPS + c4 * 4) = ok ? ld : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        if constexpr (PY == 1) {
#pragma unroll
            for (int tap = 0; tap < (RT > 0 ? RT * RT : T); ++tap) {
                const int r = tap / R, s = tap - r * R;
                const float* px = smem + ((ty + r) * HTW + tx + s) * FPS;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    if (c0 + c4 * 4 >= Cin) break;       // (uniform) Cin % 16 != 0: no weights exist there -- reading on past the row would
                                                         // multiply the staged zeros by whatever follows the weight buffer (NaN * 0)
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(px + c4 * 4);
#pragma unroll
                    for (int co = 0; co < COUT; ++co) {
                        // wave-uniform address -> scalar loads
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + ((size_t)co * T + tap) * Cin + c0 + c4 * 4);
                        acc[0][co] = fmaf(xv.x, wv.x, acc[0][co]);
                        acc[0][co] = fmaf(xv.y, wv.y, acc[0][co]);
                        acc[0][co] = fmaf(xv.z, wv.z, acc[0][co]);
                        acc[0][co] = fmaf(xv.w, wv.w, acc[0][co]);
                    }
                }
            }
        } else {
            for (int s = 0; s < R; ++s) {
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    if (c0 + c4 * 4 >= Cin) break;       // (uniform, as above)
                    const float* px = smem + ((ty * PY) * HTW + tx + s) * FPS + c4 * 4;
                    const float* wp = w + (size_t)s * Cin + c0 + c4 * 4;          // + (co T + r R) Cin
                    // input rows i = 0 .. R + PY - 2 of this thread's column: row i is tap row i - j of output row j
#pragma unroll
                    for (int i = 0; i < (RT > 0 ? RT : R) + PY - 1; ++i) {
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(px + (size_t)i * HTW * FPS);
#pragma unroll
                        for (int j = 0; j < PY; ++j) {
                            const int r = i - j;
                            if (r >= 0 && r < R) {                               // (uniform)
#pragma unroll
                                for (int co = 0; co < COUT; ++co) {
                                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + ((size_t)co * T + r * R) * Cin);
                                    acc[j][co] = fmaf(xv.x, wv.x, acc[j][co]);
                                    acc[j][co] = fmaf(xv.y, wv.y, acc[j][co]);
                                    acc[j][co] = fmaf(xv.z, wv.z, acc[j][co]);
                                    acc[j][co] = fmaf(xv.w, wv.w, acc[j][co]);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < PY; ++j) {
        const int oy = ty0 + ty * PY + j, ox = tx0 + tx;
        if (oy < Ho && ox < Wo) {
            float* d = y + ((size_t)(n * Ho + oy) * Wo + ox) * ldy;
            if (gridDim.y > 1) {
#pragma unroll
                for (int co = 0; co < COUT; ++co) atomicAdd(d + co, acc[j][co] + ((bias && blockIdx.y == 0 && accumulate) ? bias[co] : 0.f));
            } else {
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    float v = acc[j][co] + (bias ? bias[co] : 0.f);
                    d[co] = accumulate ? d[co] + v : v;
                }
            }
        }
    }
}

// y[p][co] = bias[co] (or 0): initialises the output of a channel-split forward launch
__global__ void fewout_init_kernel(float* __restrict__ y, int ldy, long long pixels, int Cout, const float* __restrict__ bias) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= pixels * Cout) return;
    const long long pix = i / Cout;
    const int co = (int)(i - pix * Cout);
    y[(size_t)pix * ldy + co] = bias ? bias[co] : 0.f;
}

// dW[co][tap][ci] += sum_p dY[p][co] * X[p + tap - pad][ci]
template <int COUT>
__global__ __launch_bounds__(256) void conv_fewout_wgrad_kernel(const float* __restrict__ x, int ldx, int H, int W, int Cin,
                                                               const float* __restrict__ dy, int lddy, int Ho, int Wo, int R, int pad,
                                                               float* __restrict__ dw, float* __restrict__ dbias, int tiles_x,
                                                               int tiles_y, int ntiles, int cc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HT = FT + R - 1;
    const int T = R * R;
    const int ps = cc + 4;                                  // pixel stride of the x tile (floats)
    float* xs = smem;                                        // HT*HT*ps
    float* dys = smem + HT * HT * ps;                        // 256 * 4
    const int tid = threadIdx.x;
    const int groups = cc / 4;                               // float4 groups per chunk
    const int items = T * groups;                            // (tap, group) work items (<= 256)
    const bool active = tid < items;
    const int tap = active ? tid / groups : 0, g4 = active ? tid - (tid / groups) * groups : 0;
    const int r = tap / R, s = tap - r * R;

    // blockIdx.y = channel chunk: every workgroup stages its tiles for ONE chunk (the tile was re-staged per chunk anyway), so a
    // low-resolution level with 8 tiles runs Cin / cc times as many workgroups instead of 8 long ones
    {
        const int c0 = blockIdx.y * cc;
        f32x4 acc[COUT];
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        float bsum = 0.f;
        for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
            const int n = t / (tiles_x * tiles_y);
            const int trem = t - n * tiles_x * tiles_y;
            const int ty0 = (trem / tiles_x) * FT, tx0 = (trem % tiles_x) * FT;
            const float* xin = x + (size_t)n * H * W * ldx;
            __syncthreads();
            for (int i = tid; i < HT * HT * groups; i += 256) {
                const int gg = i % groups, pix = i / groups;
                const int hy = pix / HT, hx = pix - hy * HT;
                const int iy = ty0 + hy - pad, ix = tx0 + hx - pad;
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && c0 + gg * 4 < Cin;      // branch-free, as in the forward
                const f32x4 ld = *reinterpret_cast<const f32x4*>(xin + (ok ? ((size_t)iy * W + ix) * ldx + c0 + gg * 4 : (size_t)0));
                *reinterpret_cast<f32x4*>(xs + pix * ps + gg * 4) = ok ? ld : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            {
                const int py = tid >> 4, pxx = tid & 15;
                const int oy = ty0 + py, ox = tx0 + pxx;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (oy < Ho && ox < Wo) {
                    const float* d = dy + ((size_t)(n * Ho + oy) * Wo + ox) * lddy;
#pragma unroll
                    for (int c = 0; c < COUT; ++c) v[c] = d[c];
                }
                *reinterpret_cast<f32x4*>(dys + tid * 4) = v;
            }
            __syncthreads();
            if (active) {
                for (int py = 0; py < FT; ++py) {
                    const float* xrow = xs + ((py + r) * HT + s) * ps + g4 * 4;
                    const float* drow = dys + py * FT * 4;
#pragma unroll 4
                    for (int pxx = 0; pxx < FT; ++pxx) {
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(xrow + pxx * ps);
                        const f32x4 dv = *reinterpret_cast<const f32x4*>(drow + pxx * 4);      // broadcast
#pragma unroll
                        for (int c = 0; c < COUT; ++c) acc[c] += xv * dv[c];
                    }
                }
            }
            if (dbias && c0 == 0 && tid < COUT) {
                float sacc = 0.f;
                for (int q = 0; q < 256; ++q) sacc += dys[q * 4 + tid];
                bsum += sacc;
            }
        }
        if (active) {
#pragma unroll
            for (int c = 0; c < COUT; ++c) {
                float* d = dw + ((size_t)c * T + tap) * Cin + c0 + g4 * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (c0 + g4 * 4 + q < Cin) atomicAdd(d + q, acc[c][q]);
            }
        }
        if (dbias && c0 == 0 && tid < COUT) atomicAdd(dbias + tid, bsum);
    }
}

}  // namespace

extern "C" int mrfa_conv_fewout_fwd(void* stream, const float* x, int ldx, int N, int H, int W, int Cin, const float* w,
                                    const float* bias, float* y, int ldy, int Cout, int R, int pad, int accumulate) {
    MRFA_CHECK_ARG(x && w && y && Cout >= 1 && Cout <= 4 && (Cin % 4) == 0 && (ldx % 4) == 0 && aligned16(x) && aligned16(w),
                   "conv_fewout_fwd: needs Cout <= 4, Cin %% 4 == 0, 16-B aligned x / w");
    const int Ho = H + 2 * pad - R + 1, Wo = W + 2 * pad - R + 1;
    // output rows per thread: 2 for the big-kernel layers with one or two outputs (measured, 128 -> 2 7x7, B = 8: 448 -> 437 us @256^2, 200 -> 117 @128^2 with the
    // workgroup target below, 52 -> 26 @64^2; three outputs are faster with one row: 280 vs 332 us, 64 -> 3 @256^2)
    const int PY = (R >= 5 && Cout <= 2 && Ho >= 32) ? 2 : 1;
    const bool rt7 = true;                                   // kernel size 7 at compile time: 554 -> 448 us (128 -> 2 @256^2), 327 -> 280 (64 -> 3)
    const int tiles_x = cdiv(Wo, FT), tiles_y = cdiv(Ho, FT * PY);
    const size_t lds = (size_t)(FT * PY + R - 1) * (FT + R - 1) * FPS * sizeof(float);
    if (lds > 64 * 1024) {
        MRFA_CHECK_ARG(lds <= 160 * 1024, "conv_fewout_fwd: halo tile does not fit the LDS");
    }
    hipStream_t st = (hipStream_t)stream;
    const int ntiles = N * tiles_x * tiles_y;
    const int chunks = cdiv(Cin, FCC);
    const int want = 1024;                                   // workgroups wanted: below half of that the channel chunks are split over gridDim.y
    int csplit = 1;
    if (ntiles < want / 2 && chunks > 1) {
        csplit = cdiv(want, ntiles);
        if (csplit > chunks) csplit = chunks;
    }
    const int cper = cdiv(chunks, csplit) * FCC;
    csplit = cdiv(Cin, cper);
    if (csplit > 1 && !accumulate) {
        const long long tot = (long long)N * Ho * Wo * Cout;
        hipLaunchKernelGGL(fewout_init_kernel, dim3((unsigned)cdiv(tot, 256)), dim3(256), 0, st, y, ldy, (long long)N * Ho * Wo, Cout, bias);
    }
    dim3 grid((unsigned)ntiles, (unsigned)csplit);
#define FWD3(co, py, rt)                                                                                                                        \
    do {                                                                                                                                    \
        if (lds > 64 * 1024)                                                                                                                \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fewout_fwd_kernel<co, py, rt>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((conv_fewout_fwd_kernel<co, py, rt>), grid, dim3(256), lds, st, x, ldx, H, W, Cin, w, bias, y, ldy, Ho, Wo, R, pad,   \
                           accumulate, tiles_x, tiles_y, cper);                                                                             \
    } while (0)
#define FWD2(co, py) do { if (R == 7 && rt7) FWD3(co, py, 7); else FWD3(co, py, 0); } while (0)
#define FWD(co) do { if (PY == 4) FWD2(co, 4); else if (PY == 2) FWD2(co, 2); else FWD2(co, 1); } while (0)
    switch (Cout) { case 1: FWD(1); break; case 2: FWD(2); break; case 3: FWD(3); break; default: FWD(4); break; }
#undef FWD
#undef FWD2
#undef FWD3
    MRFA_CHECK_LAUNCH("conv_fewout_fwd");
    return 0;
}

extern "C" int mrfa_conv_fewout_wgrad(void* stream, const float* x, int ldx, int N, int H, int W, int Cin, const float* dy, int lddy,
                                      int Cout, int R, int pad, float* dw, float* dbias) {
    MRFA_CHECK_ARG(x && dy && dw && Cout >= 1 && Cout <= 4 && (Cin % 4) == 0 && (ldx % 4) == 0 && aligned16(x),
                   "conv_fewout_wgrad: needs Cout <= 4, Cin %% 4 == 0, 16-B aligned x");
    {
        int rc3 = 0;
        if (mrfa_fewout3_wgrad((hipStream_t)stream, x, ldx, N, H, W, Cin, dy, lddy, Cout, R, pad, dw, dbias, &rc3)) return rc3;
    }
    const int Ho = H + 2 * pad - R + 1, Wo = W + 2 * pad - R + 1;
    const int tiles_x = cdiv(Wo, FT), tiles_y = cdiv(Ho, FT);
    const int ntiles = N * tiles_x * tiles_y;
    const int T = R * R;
    MRFA_CHECK_ARG(T * 1 <= 256, "conv_fewout_wgrad: kernel too large");
    int cc = 4;                                   // largest chunk with T * cc/4 <= 256 work items, <= 64 channels
    const int HT = FT + R - 1;
    auto lds_of = [&](int c) { return ((size_t)HT * HT * (c + 4) + 256 * 4) * sizeof(float); };
    // few tiles (low-resolution levels): 16-channel chunks, i.e. more workgroups with shorter chains
    const int cc_max = ntiles < 64 ? 16 : 64;
    while (cc * 2 <= cc_max && T * (cc * 2 / 4) <= 256 && cc * 2 <= ((Cin + 3) / 4 * 4) && lds_of(cc * 2) <= 64 * 1024) cc *= 2;
    const size_t lds = lds_of(cc);
    MRFA_CHECK_ARG(lds <= 64 * 1024, "conv_fewout_wgrad: LDS tile too large");
    dim3 grid((unsigned)(ntiles < 512 ? ntiles : 512), (unsigned)cdiv(Cin, cc));
    hipStream_t st = (hipStream_t)stream;
#define WG(co) hipLaunchKernelGGL((conv_fewout_wgrad_kernel<co>), grid, dim3(256), lds, st, x, ldx, H, W, Cin, dy, lddy, Ho, Wo, R, pad, dw, \
                                  dbias, tiles_x, tiles_y, ntiles, cc)
    switch (Cout) { case 1: WG(1); break; case 2: WG(2); break; case 3: WG(3); break; default: WG(4); break; }
#undef WG
    MRFA_CHECK_LAUNCH("conv_fewout_wgrad");
    return 0;
}
