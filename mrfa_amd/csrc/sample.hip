// Gather kernels of the MRFA hot path on NHWC views: bilinear grid_sample (both coordinate conventions the reference
// uses), align_corners=True bilinear resize, and the correlation-pyramid window lookup.  All HBM/L2-bound gathers:
// one thread per (output pixel, channel) with channels fastest so every bilinear tap is a contiguous 4*C-byte read
// (the NCHW reference gathers one 4-byte element per tap per channel plane).
// Replaces aten::grid_sampler_2d (+backward) at modules/util.py:34, dense_motion.py:83, raft.py:166,168,247,260,271,302,
// aten::upsample_bilinear2d (+backward) at raft.py:161-162,205-206,228,243,266-267,279-295,308 and CorrBlock.__call__
// (raft.py:23-48) including its avg_pool2d pyramid (raft.py:20) and the two 64 MiB/sample transposes (raft.py:208,235).
#include "common.h"

namespace {

struct Taps {
    int x0, y0;
    float fx, fy;   // fractional parts; tap weights (1-fx)(1-fy) etc.
};

__device__ __forceinline__ Taps make_taps(float ix, float iy) {
    Taps t;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    t.x0 = (int)fx0;
    t.y0 = (int)fy0;
    t.fx = ix - fx0;
    t.fy = iy - fy0;
    return t;
}


// Branch-free tap set: the four tap addresses are always valid (out-of-image taps are clamped onto the neighbouring in-image tap,
// a pixel wholly outside samples pixel (0, 0)) and the weights of the taps that do not exist are zero -- a load inside a divergent branch
// makes hipcc wait for it (vmcnt(0)) at the join, which turned the four taps of a pixel into four serial memory round trips.
struct Taps4 {
    size_t o00, o01, o10, o11;      // pixel offsets (units of pixels) of the four taps
    float w00, w01, w10, w11;       // bilinear weights, 0 for taps outside the image
    float fx, fy;
    bool ok00, ok01, ok10, ok11;
};

__device__ __forceinline__ Taps4 make_taps4(float ix, float iy, int Wi, int Hi) {
    const bool inr = ix > -1.f && iy > -1.f && ix < (float)Wi && iy < (float)Hi;      // NaN / huge coordinates: "outside"
    const Taps tp = make_taps(inr ? ix : 0.f, inr ? iy : 0.f);
    const bool x0ok = inr && tp.x0 >= 0, x1ok = inr && tp.x0 + 1 < Wi, y0ok = tp.y0 >= 0, y1ok = tp.y0 + 1 < Hi;
    const int x0 = tp.x0 < 0 ? 0 : tp.x0, y0 = tp.y0 < 0 ? 0 : tp.y0;
    const int x1 = tp.x0 + 1 < Wi ? tp.x0 + 1 : Wi - 1, y1 = tp.y0 + 1 < Hi ? tp.y0 + 1 : Hi - 1;
    Taps4 t;
    t.fx = tp.fx; t.fy = tp.fy;
    t.ok00 = y0ok && x0ok; t.ok01 = y0ok && x1ok; t.ok10 = y1ok && x0ok; t.ok11 = y1ok && x1ok;
    t.o00 = (size_t)y0 * Wi + x0; t.o01 = (size_t)y0 * Wi + x1; t.o10 = (size_t)y1 * Wi + x0; t.o11 = (size_t)y1 * Wi + x1;
    t.w00 = t.ok00 ? (1.f - tp.fx) * (1.f - tp.fy) : 0.f;
    t.w01 = t.ok01 ? tp.fx * (1.f - tp.fy) : 0.f;
    t.w10 = t.ok10 ? (1.f - tp.fx) * tp.fy : 0.f;
    t.w11 = t.ok11 ? tp.fx * tp.fy : 0.f;
    return t;
}

__device__ __forceinline__ void sample_coords(const float* __restrict__ grid, int ldg, long long opix, int ox, int oy, int Wi, int Hi,
                                              int mode, float& ix, float& iy) {
    const float gx = grid[(size_t)opix * ldg], gy = grid[(size_t)opix * ldg + 1];
    if (mode == 0) {   // normalised, align_corners=False
        ix = ((gx + 1.f) * (float)Wi - 1.f) * 0.5f;
        iy = ((gy + 1.f) * (float)Hi - 1.f) * 0.5f;
    } else {           // flow in pixels added to the identity grid, align_corners=True
        ix = (float)ox + gx;
        iy = (float)oy + gy;
    }
}

__global__ __launch_bounds__(256) void grid_sample_fwd_kernel(const float* __restrict__ in, int ldi, long long in_bstride, int in_rep,
                                                             int Hi, int Wi, int C, const float* __restrict__ grid, int ldg, int N,
                                                             int Ho, int Wo, float* __restrict__ out, int ldo, int mode, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long opix = i / C;
        const int c = (int)(i - opix * C);
        const int ox = (int)(opix % Wo);
        const long long t = opix / Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float ix, iy;
        sample_coords(grid, ldg, opix, ox, oy, Wi, Hi, mode, ix, iy);
        const Taps4 tp = make_taps4(ix, iy, Wi, Hi);
        const float* base = in + (size_t)(n / in_rep) * in_bstride + c;
        // taps that do not exist read a clamped (valid) address and are SELECTED away, not multiplied by 0: a non-finite value
        // at the clamped pixel must not leak into a zero-padded output (0 * inf = NaN)
        const float l00 = base[tp.o00 * ldi], l01 = base[tp.o01 * ldi], l10 = base[tp.o10 * ldi], l11 = base[tp.o11 * ldi];
        const float v = tp.w00 * (tp.ok00 ? l00 : 0.f) + tp.w01 * (tp.ok01 ? l01 : 0.f) + tp.w10 * (tp.ok10 ? l10 : 0.f) + tp.w11 * (tp.ok11 ? l11 : 0.f);
        out[(size_t)opix * ldo + c] = v;
    }
}

// float4 variants (feature warps: C in {64, 128, 256, 512, ...}): a lane owns 4 consecutive channels, LPP = min(64, C/4) lanes
// cover one pixel (chunk), so a wave streams 64 / LPP pixels; 16-byte taps instead of 4-byte ones (4x fewer memory
// instructions); d(grid) reduces over the LPP lanes of a pixel with shuffles.
template <int LPP>
__global__ __launch_bounds__(256) void grid_sample_fwd_vec_kernel(const float* __restrict__ in, int ldi, long long in_bstride, int in_rep,
                                                                 int Hi, int Wi, int C, const float* __restrict__ grid, int ldg,
                                                                 long long npix, int Ho, int Wo, float* __restrict__ out, int ldo, int mode) {
    const int chunks = C / (4 * LPP);
    const long long items = npix * chunks;                 // (pixel, chunk of 4*LPP channels)
    const int sub = threadIdx.x % LPP;
    const long long first = (blockIdx.x * (long long)blockDim.x + threadIdx.x) / LPP;
    const long long step = ((long long)gridDim.x * blockDim.x) / LPP;
    for (long long it = first; it < items; it += step) {
        const long long opix = it / chunks;
        const int c = (int)(it - opix * chunks) * 4 * LPP + sub * 4;
        const int ox = (int)(opix % Wo);
        const long long t = opix / Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float ix, iy;
        sample_coords(grid, ldg, opix, ox, oy, Wi, Hi, mode, ix, iy);
        const Taps4 tp = make_taps4(ix, iy, Wi, Hi);
        const float* base = in + (size_t)(n / in_rep) * in_bstride + c;
        const f32x4 t00 = *reinterpret_cast<const f32x4*>(base + tp.o00 * ldi), t01 = *reinterpret_cast<const f32x4*>(base + tp.o01 * ldi);
        const f32x4 t10 = *reinterpret_cast<const f32x4*>(base + tp.o10 * ldi), t11 = *reinterpret_cast<const f32x4*>(base + tp.o11 * ldi);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 v = tp.w00 * (tp.ok00 ? t00 : z) + tp.w01 * (tp.ok01 ? t01 : z) + tp.w10 * (tp.ok10 ? t10 : z) + tp.w11 * (tp.ok11 ? t11 : z);
        *reinterpret_cast<f32x4*>(out + (size_t)opix * ldo + c) = v;
    }
}

template <int LPP>
__global__ __launch_bounds__(256) void grid_sample_bwd_vec_kernel(const float* __restrict__ in, int ldi, long long in_bstride, int in_rep,
                                                                 int Hi, int Wi, int C, const float* __restrict__ grid, int ldg,
                                                                 long long npix, int Ho, int Wo, const float* __restrict__ dout, int lddo,
                                                                 int mode, float* __restrict__ din, int lddi, long long din_bstride,
                                                                 float* __restrict__ dgrid, int lddg) {
    const int chunks = C / (4 * LPP);
    const long long items = npix * chunks;
    const int sub = threadIdx.x % LPP;
    const long long first = (blockIdx.x * (long long)blockDim.x + threadIdx.x) / LPP;
    const long long step = ((long long)gridDim.x * blockDim.x) / LPP;
    // all lanes of a wave run the same number of iterations (items are handed out per LPP-lane group, the tail is masked)
    const long long iters = (items + step - 1) / step;
    for (long long k = 0; k < iters; ++k) {
        const long long it = first + k * step;
        const bool live = it < items;
        const long long opix = live ? it / chunks : 0;
        const int c = (int)((live ? it : 0) - opix * chunks) * 4 * LPP + sub * 4;
        const int ox = (int)(opix % Wo);
        const long long t = opix / Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float ix, iy;
        sample_coords(grid, ldg, opix, ox, oy, Wi, Hi, mode, ix, iy);
        float gxs = 0.f, gys = 0.f;
        {
            const Taps4 tp = make_taps4(live ? ix : -2.f, iy, Wi, Hi);       // dead items: every tap weight 0, nothing stored
            const f32x4 g = *reinterpret_cast<const f32x4*>(dout + (size_t)opix * lddo + c);
            const size_t ib = (size_t)(n / in_rep) * in_bstride + c;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 l00 = *reinterpret_cast<const f32x4*>(in + ib + tp.o00 * ldi), l01 = *reinterpret_cast<const f32x4*>(in + ib + tp.o01 * ldi);
            const f32x4 l10 = *reinterpret_cast<const f32x4*>(in + ib + tp.o10 * ldi), l11 = *reinterpret_cast<const f32x4*>(in + ib + tp.o11 * ldi);
            const f32x4 v00 = tp.ok00 ? l00 : z, v01 = tp.ok01 ? l01 : z, v10 = tp.ok10 ? l10 : z, v11 = tp.ok11 ? l11 : z;
            if (din) {
                float* db = din + (size_t)(n / in_rep) * din_bstride + c;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (tp.ok00) atomicAdd(db + tp.o00 * lddi + q, g[q] * tp.w00);
                    if (tp.ok01) atomicAdd(db + tp.o01 * lddi + q, g[q] * tp.w01);
                    if (tp.ok10) atomicAdd(db + tp.o10 * lddi + q, g[q] * tp.w10);
                    if (tp.ok11) atomicAdd(db + tp.o11 * lddi + q, g[q] * tp.w11);
                }
            }
            const bool any = tp.ok00 || tp.ok01 || tp.ok10 || tp.ok11;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gxs += g[q] * ((v01[q] - v00[q]) * (1.f - tp.fy) + (v11[q] - v10[q]) * tp.fy);
                gys += g[q] * ((v10[q] - v00[q]) * (1.f - tp.fx) + (v11[q] - v01[q]) * tp.fx);
            }
            if (!any) { gxs = 0.f; gys = 0.f; }
        }
        if (dgrid) {
#pragma unroll
            for (int o = LPP / 2; o > 0; o >>= 1) {
                gxs += __shfl_xor(gxs, o, 64);
                gys += __shfl_xor(gys, o, 64);
            }
            if (sub == 0 && live) {
                const float mx = mode == 0 ? 0.5f * (float)Wi : 1.f, my = mode == 0 ? 0.5f * (float)Hi : 1.f;
                if (chunks == 1) {
                    dgrid[(size_t)opix * lddg] += gxs * mx;
                    dgrid[(size_t)opix * lddg + 1] += gys * my;
                } else {
                    atomicAdd(dgrid + (size_t)opix * lddg, gxs * mx);
                    atomicAdd(dgrid + (size_t)opix * lddg + 1, gys * my);
                }
            }
        }
    }
}

// one wave per RUN of consecutive output pixels of one row x chunk of 64 channels: lanes = channels, so d(grid) reduces with wave shuffles
// and the tap coordinates are wave-uniform.  The input-gradient scatter merges neighbours before it reaches memory: with a smooth flow
// the right-hand taps (y0 | y1, x0 + 1) of output pixel ox are the left-hand taps of pixel ox + 1, so the right column is carried in two
// registers and added to the next pixel's left column when the coordinates match (a wave-uniform test) -- ~2 instead of 4 atomics per
// (pixel, channel) on the feature warps of the path (was: 18 launches x 117 us per step, ~1 TB/s).
constexpr int GS_RUN = 8;
__global__ __launch_bounds__(256) void grid_sample_bwd_kernel(const float* __restrict__ in, int ldi, long long in_bstride, int in_rep,
                                                             int Hi, int Wi, int C, const float* __restrict__ grid, int ldg,
                                                             long long npix, int Ho, int Wo, const float* __restrict__ dout, int lddo,
                                                             int mode, float* __restrict__ din, int lddi, long long din_bstride,
                                                             float* __restrict__ dgrid, int lddg) {
    const int lane = threadIdx.x & 63;
    const long long wave_id = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int chunks = (C + 63) / 64;
    const int runs_x = (Wo + GS_RUN - 1) / GS_RUN;
    const long long nruns = (long long)(npix / Wo) * runs_x;          // (image row, run of GS_RUN pixels)
    for (long long wi = wave_id; wi < nruns * chunks; wi += nwaves) {
        const long long run = wi / chunks;
        const int c = (int)(wi - run * chunks) * 64 + lane;
        const bool c_ok = c < C;
        const int cc = c_ok ? c : 0;                                          // lanes past C read channel 0 and contribute nothing
        const long long rowi = run / runs_x;                                  // n * Ho + oy
        const int ox0 = (int)(run - rowi * runs_x) * GS_RUN;
        const int oy = (int)(rowi % Ho);
        const int n = (int)(rowi / Ho);
        const size_t ib = (size_t)(n / in_rep) * in_bstride + cc;
        float* db = din ? din + (size_t)(n / in_rep) * din_bstride + cc : nullptr;
        // carried right column: contributions to input pixels (py0, px) and (py1, px), valid flags per row
        float ctop = 0.f, cbot = 0.f;
        long long ptop = -1, pbot = -1;                                       // pixel offsets (y * Wi + x), -1 = nothing pending
        const int nrun = min(GS_RUN, Wo - ox0);
        for (int k = 0; k < nrun; ++k) {
            const int ox = ox0 + k;
            const long long opix = rowi * Wo + ox;
            float ix, iy;
            sample_coords(grid, ldg, opix, ox, oy, Wi, Hi, mode, ix, iy);
            const Taps4 tp = make_taps4(ix, iy, Wi, Hi);                      // (wave-uniform: one pixel per wave and step)
            const float g = c_ok ? dout[(size_t)opix * lddo + cc] : 0.f;
            const float l00 = in[ib + tp.o00 * ldi], l01 = in[ib + tp.o01 * ldi], l10 = in[ib + tp.o10 * ldi], l11 = in[ib + tp.o11 * ldi];
            const float v00 = tp.ok00 ? l00 : 0.f, v01 = tp.ok01 ? l01 : 0.f, v10 = tp.ok10 ? l10 : 0.f, v11 = tp.ok11 ? l11 : 0.f;
            if (db) {
                float a00 = g * tp.w00, a10 = g * tp.w10;
                const long long q00 = tp.ok00 ? (long long)tp.o00 : -2, q10 = tp.ok10 ? (long long)tp.o10 : -2;
                // merge the carried column into this pixel's left column where they are the same input pixel, flush it otherwise
                if (ptop >= 0) { if (ptop == q00) a00 += ctop; else if (c_ok) atomicAdd(db + ptop * lddi, ctop); }
                if (pbot >= 0) { if (pbot == q10) a10 += cbot; else if (c_ok) atomicAdd(db + pbot * lddi, cbot); }
                if (c_ok) {
                    if (tp.ok00) atomicAdd(db + tp.o00 * lddi, a00);
                    if (tp.ok10) atomicAdd(db + tp.o10 * lddi, a10);
                }
                ptop = tp.ok01 ? (long long)tp.o01 : -1;
                pbot = tp.ok11 ? (long long)tp.o11 : -1;
                ctop = g * tp.w01;
                cbot = g * tp.w11;
            }
            if (dgrid) {
                // d val / d ix = (v01 - v00)(1-fy) + (v11 - v10) fy ; d val / d iy = (v10 - v00)(1-fx) + (v11 - v01) fx
                float gxs = wave_sum(g * ((v01 - v00) * (1.f - tp.fy) + (v11 - v10) * tp.fy));
                float gys = wave_sum(g * ((v10 - v00) * (1.f - tp.fx) + (v11 - v01) * tp.fx));
                if (lane == 0) {
                    const float mx = mode == 0 ? 0.5f * (float)Wi : 1.f, my = mode == 0 ? 0.5f * (float)Hi : 1.f;
                    if (chunks == 1) {
                        dgrid[(size_t)opix * lddg] += gxs * mx;
                        dgrid[(size_t)opix * lddg + 1] += gys * my;
                    } else {
                        atomicAdd(dgrid + (size_t)opix * lddg, gxs * mx);
                        atomicAdd(dgrid + (size_t)opix * lddg + 1, gys * my);
                    }
                }
            }
        }
        if (db && c_ok) {
            if (ptop >= 0) atomicAdd(db + ptop * lddi, ctop);
            if (pbot >= 0) atomicAdd(db + pbot * lddi, cbot);
        }
    }
}

// ---------------------------------------------------------------------------------------------- bilinear resize (ac=True)
__device__ __forceinline__ void resize_src(int o, int Ni, int No, int& i0, int& i1, float& f) {
    const float s = No > 1 ? (float)(Ni - 1) / (float)(No - 1) : 0.f;
    const float src = s * (float)o;
    i0 = (int)src;                      // src >= 0
    if (i0 > Ni - 1) i0 = Ni - 1;
    i1 = i0 + (i0 < Ni - 1 ? 1 : 0);
    f = src - (float)i0;
}

__global__ void resize_fwd_kernel(const float* __restrict__ in, int ldi, int N, int Hi, int Wi, int C, float* __restrict__ out, int ldo,
                                  int Ho, int Wo, float mul, int acc, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long opix = i / C;
        const int c = (int)(i - opix * C);
        const int ox = (int)(opix % Wo);
        const long long t = opix / Wo;
        const int oy = (int)(t % Ho);
        const long long n = t / Ho;
        int x0, x1, y0, y1;
        float fx, fy;
        resize_src(ox, Wi, Wo, x0, x1, fx);
        resize_src(oy, Hi, Ho, y0, y1, fy);
        const float* b = in + (size_t)n * Hi * Wi * ldi + c;
        const float top = b[((size_t)y0 * Wi + x0) * ldi] * (1.f - fx) + b[((size_t)y0 * Wi + x1) * ldi] * fx;
        const float bot = b[((size_t)y1 * Wi + x0) * ldi] * (1.f - fx) + b[((size_t)y1 * Wi + x1) * ldi] * fx;
        const float v = (top * (1.f - fy) + bot * fy) * mul;
        float* d = out + (size_t)opix * ldo + c;
        *d = acc ? (*d + v) : v;
    }
}

__global__ void resize_bwd_kernel(const float* __restrict__ dout, int lddo, int N, int Hi, int Wi, int C, float* __restrict__ din,
                                  int lddi, int Ho, int Wo, float mul, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long opix = i / C;
        const int c = (int)(i - opix * C);
        const int ox = (int)(opix % Wo);
        const long long t = opix / Wo;
        const int oy = (int)(t % Ho);
        const long long n = t / Ho;
        int x0, x1, y0, y1;
        float fx, fy;
        resize_src(ox, Wi, Wo, x0, x1, fx);
        resize_src(oy, Hi, Ho, y0, y1, fy);
        const float g = dout[(size_t)opix * lddo + c] * mul;
        float* b = din + (size_t)n * Hi * Wi * lddi + c;
        atomicAdd(b + ((size_t)y0 * Wi + x0) * lddi, g * (1.f - fx) * (1.f - fy));
        atomicAdd(b + ((size_t)y0 * Wi + x1) * lddi, g * fx * (1.f - fy));
        atomicAdd(b + ((size_t)y1 * Wi + x0) * lddi, g * (1.f - fx) * fy);
        atomicAdd(b + ((size_t)y1 * Wi + x1) * lddi, g * fx * fy);
    }
}

// Up-sampling backward as a GATHER (no atomics): one thread per (input pixel, channel) visits the output pixels whose
// two source taps include it.  Used when Ho >= Hi and Wo >= Wi (the x2 / x4 resizes of the flow recomposition and the
// 98-channel correlation features, raft.py:243,279-295); the scatter version above remains for down-sampling.
__global__ void resize_bwd_gather_kernel(const float* __restrict__ dout, int lddo, int N, int Hi, int Wi, int C, float* __restrict__ din,
                                         int lddi, int Ho, int Wo, float mul, long long total) {
    const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long ipix = i / C;
        const int c = (int)(i - ipix * C);
        const int ix = (int)(ipix % Wi);
        const long long t = ipix / Wi;
        const int iy = (int)(t % Hi);
        const long long n = t / Hi;
        // candidate output range: src in (i-1, i+1)  (one extra on each side against rounding)
        int oy_lo = sy > 0.f ? (int)floorf((float)(iy - 1) / sy) - 1 : 0, oy_hi = sy > 0.f ? (int)ceilf((float)(iy + 1) / sy) + 1 : Ho - 1;
        int ox_lo = sx > 0.f ? (int)floorf((float)(ix - 1) / sx) - 1 : 0, ox_hi = sx > 0.f ? (int)ceilf((float)(ix + 1) / sx) + 1 : Wo - 1;
        oy_lo = max(oy_lo, 0); oy_hi = min(oy_hi, Ho - 1); ox_lo = max(ox_lo, 0); ox_hi = min(ox_hi, Wo - 1);
        const float* g = dout + (size_t)n * Ho * Wo * lddo + c;
        float acc = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1; float fy;
            resize_src(oy, Hi, Ho, y0, y1, fy);
            const float wy = (y0 == iy ? 1.f - fy : 0.f) + (y1 == iy ? fy : 0.f);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1; float fx;
                resize_src(ox, Wi, Wo, x0, x1, fx);
                const float wx = (x0 == ix ? 1.f - fx : 0.f) + (x1 == ix ? fx : 0.f);
                if (wx != 0.f) row += wx * g[((size_t)oy * Wo + ox) * lddo];
            }
            acc += wy * row;
        }
        din[(size_t)ipix * lddi + c] += acc * mul;
    }
}

// ---------------------------------------------------------------------------------------------- sums of resizes, many per launch (v8)
// The flow / occlusion re-composition between two refinement levels (raft.py:276-295) and the update of the running flow (raft.py:258-262) are ~16 copies and
// resizes of 1- and 2-channel maps per level, each a launch of its own on a chain where nothing else runs (~170 launches per step in both directions, 5-18 us
// each).  Here ONE launch evaluates a table of "dst (=|+=) sum_k mul_k resize(src_k)" records, one thread per dst element, the terms in table order -- the
// arithmetic (and its order) of the launches it replaces.  Forward: dst = an output, src_k = inputs at any size.  Backward: dst = an input's gradient,
// src_k = output gradients at sizes >= dst's (the gather form of resize_bwd_gather_kernel; identity where the sizes agree).
struct ResizeSumArgs {
    int n;
    int prefix[MRFA_RESIZE_SUM_MAX + 1];               // first 256-element block of record i
    mrfa_resize_sum_desc d[MRFA_RESIZE_SUM_MAX];
};

template <bool BWD>
__global__ __launch_bounds__(256) void resize_sum_multi_kernel(const ResizeSumArgs a) {
    int lo = 0, hi = a.n;                              // largest i with prefix[i] <= blockIdx.x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int)blockIdx.x >= a.prefix[mid]) lo = mid; else hi = mid;
    }
    const mrfa_resize_sum_desc& d = a.d[lo];
    const long long i = (long long)((int)blockIdx.x - a.prefix[lo]) * 256 + threadIdx.x;
    const long long total = (long long)d.N * d.Hd * d.Wd * d.C;
    if (i >= total) return;
    const long long pix = i / d.C;
    const int c = (int)(i - pix * d.C);
    const int px = (int)(pix % d.Wd);
    const long long t = pix / d.Wd;
    const int py = (int)(t % d.Hd);
    const long long n = t / d.Hd;
    float* dst = d.dst + (size_t)pix * d.ldd + c;
    float v = d.overwrite ? 0.f : *dst;
    bool first = d.overwrite != 0;
    for (int k = 0; k < d.nterm; ++k) {
        const float* __restrict__ src = d.term[k].src;
        const int lds = d.term[k].lds, Hs = d.term[k].Hs, Ws = d.term[k].Ws;
        const float mul = d.term[k].mul;
        float r;
        if (Hs == d.Hd && Ws == d.Wd) {
            r = src[(size_t)pix * lds + c] * mul;                                   // same size: a copy (mrfa_copy_view)
        } else if (!BWD) {
            int x0, x1, y0, y1;
            float fx, fy;
            resize_src(px, Ws, d.Wd, x0, x1, fx);
            resize_src(py, Hs, d.Hd, y0, y1, fy);
            const float* b = src + (size_t)n * Hs * Ws * lds + c;
            const float top = b[((size_t)y0 * Ws + x0) * lds] * (1.f - fx) + b[((size_t)y0 * Ws + x1) * lds] * fx;
            const float bot = b[((size_t)y1 * Ws + x0) * lds] * (1.f - fx) + b[((size_t)y1 * Ws + x1) * lds] * fx;
            r = (top * (1.f - fy) + bot * fy) * mul;                                // (resize_fwd_kernel)
        } else {
            // gather form of the up-sampling backward (resize_bwd_gather_kernel): dst = an input pixel (py, px) of a Hd x Wd map that was resized to Hs x Ws
            const int Hi = d.Hd, Wi = d.Wd, Ho = Hs, Wo = Ws;
            const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
            int oy_lo = sy > 0.f ? (int)floorf((float)(py - 1) / sy) - 1 : 0, oy_hi = sy > 0.f ? (int)ceilf((float)(py + 1) / sy) + 1 : Ho - 1;
            int ox_lo = sx > 0.f ? (int)floorf((float)(px - 1) / sx) - 1 : 0, ox_hi = sx > 0.f ? (int)ceilf((float)(px + 1) / sx) + 1 : Wo - 1;
            oy_lo = max(oy_lo, 0); oy_hi = min(oy_hi, Ho - 1); ox_lo = max(ox_lo, 0); ox_hi = min(ox_hi, Wo - 1);
            const float* g = src + (size_t)n * Ho * Wo * lds + c;
            float acc = 0.f;
            for (int oy = oy_lo; oy <= oy_hi; ++oy) {
                int y0, y1; float fy;
                resize_src(oy, Hi, Ho, y0, y1, fy);
                const float wy = (y0 == py ? 1.f - fy : 0.f) + (y1 == py ? fy : 0.f);
                if (wy == 0.f) continue;
                float row = 0.f;
                for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                    int x0, x1; float fx;
                    resize_src(ox, Wi, Wo, x0, x1, fx);
                    const float wx = (x0 == px ? 1.f - fx : 0.f) + (x1 == px ? fx : 0.f);
                    if (wx != 0.f) row += wx * g[((size_t)oy * Wo + ox) * lds];
                }
                acc += wy * row;
            }
            r = acc * mul;
        }
        v = first ? r : v + r;
        first = false;
    }
    *dst = v;
}

// ---------------------------------------------------------------------------------------------- correlation window lookup
// One wave per query pixel.  Lane e < (2r+1)^2 owns window element (a,b) = (e / (2r+1), e % (2r+1)) sampled at
// (x + a - r, y + b - r) on its level's source map; all 49 lanes share the fractional offsets, so the wave touches an
// 8x8 block of the 16 KiB map row (L2/L1 resident).  Channel order of the reference: lvl*49 + a*7 + b (raft.py:31-37).
template <bool BWD>
__global__ __launch_bounds__(256) void corr_lookup_kernel(const float* __restrict__ vol0, const float* __restrict__ vol1, int Hs, int Ws,
                                                         const float* __restrict__ coords, int ldc, long long Q, int radius,
                                                         float* __restrict__ out, int ldo, const float* __restrict__ dout, int lddo,
                                                         float* __restrict__ dvol0, float* __restrict__ dvol1,
                                                         float* __restrict__ dcoords, int lddc) {
    const int lane = threadIdx.x & 63;
    const long long wave_id = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int win = 2 * radius + 1, nwin = win * win;
    const int a = lane / win, b = lane - a * win;
    for (long long q = wave_id; q < Q; q += nwaves) {
        const float cx = coords[(size_t)q * ldc], cy = coords[(size_t)q * ldc + 1];
        float gx = 0.f, gy = 0.f;
        const bool active = lane < nwin;
        // both pyramid levels: tap sets first (branch-free: clamped addresses, zero weights -- see make_taps4), then all eight loads,
        // then the arithmetic; idle lanes (lane >= nwin) sample far outside and contribute nothing
        Taps4 tp[2];
        float l[2][4], g[2];
#pragma unroll
        for (int lvl = 0; lvl < 2; ++lvl) {
            const int H = Hs >> lvl, W = Ws >> lvl;
            const float inv = lvl == 0 ? 1.f : 0.5f;
            const float ix = cx * inv + (float)(a - radius), iy = cy * inv + (float)(b - radius);
            tp[lvl] = make_taps4(active ? ix : -2.f, iy, W, H);
            const float* vol = (lvl == 0 ? vol0 : vol1) + (size_t)q * H * W;
            l[lvl][0] = vol[tp[lvl].o00]; l[lvl][1] = vol[tp[lvl].o01]; l[lvl][2] = vol[tp[lvl].o10]; l[lvl][3] = vol[tp[lvl].o11];
            g[lvl] = (BWD && active) ? dout[(size_t)q * lddo + lvl * nwin + lane] : 0.f;
        }
#pragma unroll
        for (int lvl = 0; lvl < 2; ++lvl) {
            const int H = Hs >> lvl, W = Ws >> lvl;
            const float inv = lvl == 0 ? 1.f : 0.5f;
            const Taps4& t = tp[lvl];
            const float v00 = t.ok00 ? l[lvl][0] : 0.f, v01 = t.ok01 ? l[lvl][1] : 0.f, v10 = t.ok10 ? l[lvl][2] : 0.f, v11 = t.ok11 ? l[lvl][3] : 0.f;
            if (!BWD) {
                const float v = v00 * t.w00 + v01 * t.w01 + v10 * t.w10 + v11 * t.w11;
                if (active) out[(size_t)q * ldo + lvl * nwin + lane] = v;
            } else {
                float* dvol = dvol0 ? ((lvl == 0 ? dvol0 : dvol1) + (size_t)q * H * W) : nullptr;
                if (dvol) {
                    if (t.ok00) atomicAdd(dvol + t.o00, g[lvl] * t.w00);
                    if (t.ok01) atomicAdd(dvol + t.o01, g[lvl] * t.w01);
                    if (t.ok10) atomicAdd(dvol + t.o10, g[lvl] * t.w10);
                    if (t.ok11) atomicAdd(dvol + t.o11, g[lvl] * t.w11);
                }
                gx += g[lvl] * ((v01 - v00) * (1.f - t.fy) + (v11 - v10) * t.fy) * inv;
                gy += g[lvl] * ((v10 - v00) * (1.f - t.fx) + (v11 - v01) * t.fx) * inv;
            }
        }
        if (BWD && dcoords) {
            gx = wave_sum(gx);
            gy = wave_sum(gy);
            if (lane == 0) {
                dcoords[(size_t)q * lddc] += gx;
                dcoords[(size_t)q * lddc + 1] += gy;
            }
        }
    }
}

// Transform.transform_frame (model.py:44-48): F.grid_sample(frame, grid, padding_mode="reflection"), bilinear, align_corners=False, NCHW in and out, forward only
// (the frame is data and the warp's parameters are random constants: nothing differentiates through it).  One thread per output pixel, all channels.
__device__ __forceinline__ float reflect_coord(float v, float size) {
    // torch's reflect_coordinates(v, -1, 2 size - 1) followed by clip_coordinates: reflection about the pixel EDGES -0.5 and size - 0.5
    const float span = size;                        // (twice_high - twice_low) / 2
    v = fabsf(v + 0.5f);                            // |v - min|, min = -0.5
    const float flips = floorf(v / span);
    const float extra = v - flips * span;           // fmodf(v, span) for v >= 0
    v = (((int)flips) & 1) ? (span - extra - 0.5f) : (extra - 0.5f);
    return fminf(fmaxf(v, 0.f), size - 1.f);
}

__global__ void warp_frame_reflect_kernel(const float* __restrict__ in, int C, int H, int W, const float* __restrict__ grid, int Ho, int Wo,
                                          float* __restrict__ out, long long npix) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / ((long long)Ho * Wo);
        const float gx = grid[2 * i], gy = grid[2 * i + 1];
        const float x = reflect_coord(((gx + 1.f) * W - 1.f) * 0.5f, (float)W), y = reflect_coord(((gy + 1.f) * H - 1.f) * 0.5f, (float)H);
        const float x0f = floorf(x), y0f = floorf(y);
        const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = x - x0f, wy1 = y - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const bool x1ok = x1 < W, y1ok = y1 < H;                     // (after the clip x0, y0 are in range; the far neighbours may be one past the edge: weight 0)
        const float* pin = in + n * C * (long long)H * W;
        float* pout = out + n * C * (long long)Ho * Wo + (i - n * (long long)Ho * Wo);
        for (int c = 0; c < C; ++c) {
            const float* pc = pin + (long long)c * H * W;
            float v = pc[y0 * W + x0] * wx0 * wy0;
            if (x1ok) v += pc[y0 * W + x1] * wx1 * wy0;
            if (y1ok) v += pc[y1 * W + x0] * wx0 * wy1;
            if (x1ok && y1ok) v += pc[y1 * W + x1] * wx1 * wy1;
            pout[(long long)c * Ho * Wo] = v;
        }
    }
}

}  // namespace

extern "C" int mrfa_warp_frame_reflect(void* stream, const float* in, int N, int C, int H, int W, const float* grid, int Ho, int Wo, float* out) {
    MRFA_CHECK_ARG(in && grid && out && N > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "warp_frame_reflect: bad args");
    const long long npix = (long long)N * Ho * Wo;
    hipLaunchKernelGGL(warp_frame_reflect_kernel, dim3(stream_grid(npix, 256)), dim3(256), 0, (hipStream_t)stream, in, C, H, W, grid, Ho, Wo, out, npix);
    MRFA_CHECK_LAUNCH("warp_frame_reflect");
    return 0;
}

extern "C" int mrfa_grid_sample_fwd(void* stream, const float* in, int ldi, long long in_bstride, int in_rep, int Hi, int Wi, int C,
                                    const float* grid, int ldg, int N, int Ho, int Wo, float* out, int ldo, int mode) {
    MRFA_CHECK_ARG(in && grid && out && C > 0 && N > 0 && in_rep >= 1, "grid_sample_fwd: bad args");
    const long long total = (long long)N * Ho * Wo * C;
    const int lpp = C % 256 == 0 ? 64 : (C == 128 ? 32 : (C == 64 ? 16 : 0));
    if (lpp && ldi % 4 == 0 && ldo % 4 == 0 && in_bstride % 4 == 0 && aligned16(in) && aligned16(out)) {
        const long long npix = (long long)N * Ho * Wo;
        dim3 g(stream_grid(total / 4, 256));
#define GSF(L) hipLaunchKernelGGL((grid_sample_fwd_vec_kernel<L>), g, dim3(256), 0, (hipStream_t)stream, in, ldi, in_bstride, in_rep, Hi, Wi, C, \
                                   grid, ldg, npix, Ho, Wo, out, ldo, mode)
        if (lpp == 64) GSF(64); else if (lpp == 32) GSF(32); else GSF(16);
#undef GSF
        MRFA_CHECK_LAUNCH("grid_sample_fwd(vec)");
        return 0;
    }
    hipLaunchKernelGGL(grid_sample_fwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, in, ldi, in_bstride,
                       in_rep, Hi, Wi, C, grid, ldg, N, Ho, Wo, out, ldo, mode, total);
    MRFA_CHECK_LAUNCH("grid_sample_fwd");
    return 0;
}

extern "C" int mrfa_grid_sample_bwd(void* stream, const float* in, int ldi, long long in_bstride, int in_rep, int Hi, int Wi, int C,
                                    const float* grid, int ldg, int N, int Ho, int Wo, const float* dout, int lddo, int mode,
                                    float* din, int lddi, long long din_bstride, float* dgrid, int lddg) {
    MRFA_CHECK_ARG(in && grid && dout && C > 0 && N > 0 && in_rep >= 1, "grid_sample_bwd: bad args");
    const long long npix = (long long)N * Ho * Wo;
    const int lpp = C % 256 == 0 ? 64 : (C == 128 ? 32 : (C == 64 ? 16 : 0));
    if (lpp && !din && ldi % 4 == 0 && lddo % 4 == 0 && in_bstride % 4 == 0 && aligned16(in) && aligned16(dout)) {
        // (only when no input gradient is wanted: with din, a lane's four atomics land 16 bytes apart from its neighbours', i.e. each
        //  atomic instruction touches 4x the cache lines of the scalar kernel's 64 consecutive floats -- measured 6 % slower step)
        dim3 g(stream_grid(npix * C / 4, 256));
#define GSB(L) hipLaunchKernelGGL((grid_sample_bwd_vec_kernel<L>), g, dim3(256), 0, (hipStream_t)stream, in, ldi, in_bstride, in_rep, Hi, Wi, C, \
                                   grid, ldg, npix, Ho, Wo, dout, lddo, mode, din, lddi, din_bstride, dgrid, lddg)
        if (lpp == 64) GSB(64); else if (lpp == 32) GSB(32); else GSB(16);
#undef GSB
        MRFA_CHECK_LAUNCH("grid_sample_bwd(vec)");
        return 0;
    }
    const long long waves = (long long)N * Ho * cdiv(Wo, GS_RUN) * cdiv(C, 64);
    hipLaunchKernelGGL(grid_sample_bwd_kernel, dim3(stream_grid(waves * 64, 256)), dim3(256), 0, (hipStream_t)stream, in, ldi, in_bstride,
                       in_rep, Hi, Wi, C, grid, ldg, npix, Ho, Wo, dout, lddo, mode, din, lddi, din_bstride, dgrid, lddg);
    MRFA_CHECK_LAUNCH("grid_sample_bwd");
    return 0;
}

extern "C" int mrfa_resize_bilinear_fwd(void* stream, const float* in, int ldi, int N, int Hi, int Wi, int C, float* out, int ldo,
                                        int Ho, int Wo, float scale_mul, int accumulate) {
    MRFA_CHECK_ARG(in && out && C > 0 && N > 0, "resize_fwd: bad args");
    const long long total = (long long)N * Ho * Wo * C;
    hipLaunchKernelGGL(resize_fwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, in, ldi, N, Hi, Wi, C, out, ldo,
                       Ho, Wo, scale_mul, accumulate, total);
    MRFA_CHECK_LAUNCH("resize_fwd");
    return 0;
}

extern "C" int mrfa_resize_bilinear_bwd(void* stream, const float* dout, int lddo, int N, int Hi, int Wi, int C, float* din, int lddi,
                                        int Ho, int Wo, float scale_mul) {
    MRFA_CHECK_ARG(dout && din && C > 0 && N > 0, "resize_bwd: bad args");
    if (Ho >= Hi && Wo >= Wi) {
        const long long total_in = (long long)N * Hi * Wi * C;
        hipLaunchKernelGGL(resize_bwd_gather_kernel, dim3(stream_grid(total_in, 256)), dim3(256), 0, (hipStream_t)stream, dout, lddo, N, Hi,
                           Wi, C, din, lddi, Ho, Wo, scale_mul, total_in);
        MRFA_CHECK_LAUNCH("resize_bwd_gather");
        return 0;
    }
    const long long total = (long long)N * Ho * Wo * C;
    hipLaunchKernelGGL(resize_bwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, lddo, N, Hi, Wi, C, din,
                       lddi, Ho, Wo, scale_mul, total);
    MRFA_CHECK_LAUNCH("resize_bwd");
    return 0;
}

static int resize_sum_launch(void* stream, const mrfa_resize_sum_desc* descs, int n, bool bwd, const char* what) {
    MRFA_CHECK_ARG(n >= 0 && (n == 0 || descs), "%s: bad args", what);
    for (int base = 0; base < n; base += MRFA_RESIZE_SUM_MAX) {
        ResizeSumArgs a;
        a.n = n - base < MRFA_RESIZE_SUM_MAX ? n - base : MRFA_RESIZE_SUM_MAX;
        int blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            const mrfa_resize_sum_desc& d = descs[base + i];
            MRFA_CHECK_ARG(d.dst && d.N > 0 && d.Hd > 0 && d.Wd > 0 && d.C > 0 && d.nterm >= 1 && d.nterm <= MRFA_RESIZE_SUM_TERMS, "%s: record %d: bad sizes / term count", what, base + i);
            for (int k = 0; k < d.nterm; ++k) {
                MRFA_CHECK_ARG(d.term[k].src && d.term[k].Hs > 0 && d.term[k].Ws > 0, "%s: record %d term %d: null source / bad size", what, base + i, k);
                if (bwd) MRFA_CHECK_ARG(d.term[k].Hs >= d.Hd && d.term[k].Ws >= d.Wd, "%s: record %d term %d: the backward takes up-sampling (or same-size) terms only", what, base + i, k);
            }
            a.d[i] = d;
            a.prefix[i] = blocks;
            const long long total = (long long)d.N * d.Hd * d.Wd * d.C;
            MRFA_CHECK_ARG(total < (1ll << 31) - 256 && blocks + (total + 255) / 256 < (1ll << 30), "%s: record %d too large", what, base + i);
            blocks += (int)((total + 255) / 256);
        }
        a.prefix[a.n] = blocks;
        if (bwd) hipLaunchKernelGGL((resize_sum_multi_kernel<true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((resize_sum_multi_kernel<false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
        MRFA_CHECK_LAUNCH(what);
    }
    return 0;
}

extern "C" int mrfa_resize_sum_multi(void* stream, const mrfa_resize_sum_desc* descs, int n) {
    return resize_sum_launch(stream, descs, n, false, "resize_sum_multi");
}

extern "C" int mrfa_resize_sum_multi_bwd(void* stream, const mrfa_resize_sum_desc* descs, int n) {
    return resize_sum_launch(stream, descs, n, true, "resize_sum_multi_bwd");
}

extern "C" int mrfa_corr_lookup_fwd(void* stream, const float* vol0, const float* vol1, int Hs, int Ws, const float* coords, int ldc,
                                    long long Q, int radius, float* out, int ldo) {
    MRFA_CHECK_ARG(vol0 && vol1 && coords && out && Q > 0, "corr_lookup_fwd: bad args");
    MRFA_CHECK_ARG((2 * radius + 1) * (2 * radius + 1) <= 64, "corr_lookup: window must fit one wave (radius <= 3)");
    hipLaunchKernelGGL((corr_lookup_kernel<false>), dim3(stream_grid(Q * 64, 256)), dim3(256), 0, (hipStream_t)stream, vol0, vol1, Hs, Ws,
                       coords, ldc, Q, radius, out, ldo, nullptr, 0, nullptr, nullptr, nullptr, 0);
    MRFA_CHECK_LAUNCH("corr_lookup_fwd");
    return 0;
}

extern "C" int mrfa_corr_lookup_bwd(void* stream, const float* vol0, const float* vol1, int Hs, int Ws, const float* coords, int ldc,
                                    long long Q, int radius, const float* dout, int lddo, float* dvol0, float* dvol1, float* dcoords,
                                    int lddc) {
    MRFA_CHECK_ARG(vol0 && vol1 && coords && dout && Q > 0, "corr_lookup_bwd: bad args");
    MRFA_CHECK_ARG((dvol0 == nullptr) == (dvol1 == nullptr), "corr_lookup_bwd: dvol0/dvol1 must both be given or both null");
    MRFA_CHECK_ARG((2 * radius + 1) * (2 * radius + 1) <= 64, "corr_lookup: window must fit one wave (radius <= 3)");
    hipLaunchKernelGGL((corr_lookup_kernel<true>), dim3(stream_grid(Q * 64, 256)), dim3(256), 0, (hipStream_t)stream, vol0, vol1, Hs, Ws,
                       coords, ldc, Q, radius, nullptr, 0, dout, lddo, dvol0, dvol1, dcoords, lddc);
    MRFA_CHECK_LAUNCH("corr_lookup_bwd");
    return 0;
}
