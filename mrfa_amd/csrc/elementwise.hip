// Small HBM-bound elementwise / pooling kernels on NHWC views (channels fastest => coalesced rows):
// 2x2 pools (DownBlock2d pool and the data-gradient of the fused nearest-x2 upsample), bias+activation with optional
// BN statistics, activation backward, occlusion blend (generator.py:47,57,63), per-channel column sums and the
// strided anti-alias downsample (util.py:318-326) that computes only the outputs the nearest decimation keeps.
#include "common.h"

namespace {

#define GRID_STRIDE(i, total) \
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (total); i += (long long)gridDim.x * blockDim.x)

__global__ void avgpool2_fwd_kernel(const float* __restrict__ x, int ldx, int H, int W, int C, float* __restrict__ y, int ldy,
                                    long long total) {
    const int Ho = H / 2, Wo = W / 2;
    GRID_STRIDE(i, total) {
        const long long opix = i / C;
        const int c = (int)(i - opix * C);
        const int ox = (int)(opix % Wo);
        const long long t = opix / Wo;
        const int oy = (int)(t % Ho);
        const long long n = t / Ho;
        const size_t b = (((size_t)n * H + 2 * oy) * W + 2 * ox) * ldx + c;
        y[(size_t)opix * ldy + c] = 0.25f * (x[b] + x[b + ldx] + x[b + (size_t)W * ldx] + x[b + (size_t)(W + 1) * ldx]);
    }
}

// y[n,oy,ox,c] += mul * sum_{2x2} x[n,2oy+dy,2ox+dx,c]     (x is (2Ho x 2Wo))
__global__ void sumpool2_acc_kernel(const float* __restrict__ x, int ldx, int Ho, int Wo, int C, float* __restrict__ y, int ldy, float mul,
                                    long long total) {
    const int H = 2 * Ho, W = 2 * Wo;
    GRID_STRIDE(i, total) {
        const long long opix = i / C;
        const int c = (int)(i - opix * C);
        const int ox = (int)(opix % Wo);
        const long long t = opix / Wo;
        const int oy = (int)(t % Ho);
        const long long n = t / Ho;
        const size_t b = (((size_t)n * H + 2 * oy) * W + 2 * ox) * ldx + c;
        y[(size_t)opix * ldy + c] += mul * (x[b] + x[b + ldx] + x[b + (size_t)W * ldx] + x[b + (size_t)(W + 1) * ldx]);
    }
}

// dx[n,y,x,c] += mul * dy[n,y/2,x/2,c]      (dx is (2Ho x 2Wo))
__global__ void unpool2_acc_kernel(const float* __restrict__ dy, int lddy, int Ho, int Wo, int C, float* __restrict__ dx, int lddx,
                                   float mul, long long total) {
    const int H = 2 * Ho, W = 2 * Wo;
    GRID_STRIDE(i, total) {
        const long long pix = i / C;
        const int c = (int)(i - pix * C);
        const int xx = (int)(pix % W);
        const long long t = pix / W;
        const int yy = (int)(t % H);
        const long long n = t / H;
        dx[(size_t)pix * lddx + c] += mul * dy[(((size_t)n * Ho + (yy >> 1)) * Wo + (xx >> 1)) * lddy + c];
    }
}

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 1.f / (1.f + __expf(-v));
    return v;
}

__global__ void bias_act_kernel(const float* __restrict__ x, int ldx, long long rows, int C, const float* __restrict__ bias, int act,
                                float* __restrict__ y, int ldy, long long total) {
    GRID_STRIDE(i, total) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        float v = x[(size_t)r * ldx + c];
        if (bias) v += bias[c];
        y[(size_t)r * ldy + c] = act_apply(v, act);
    }
}

__global__ void act_bwd_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ dy, int lddy, long long rows, int C, int act,
                               float* __restrict__ dx, int lddx, int acc, long long total) {
    GRID_STRIDE(i, total) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        const float yy = y[(size_t)r * ldy + c];
        float g = dy[(size_t)r * lddy + c];
        if (act == 1) g = yy > 0.f ? g : 0.f;
        else if (act == 2) g = g * yy * (1.f - yy);
        float* d = dx + (size_t)r * lddx + c;
        *d = acc ? (*d + g) : g;
    }
}

// float4 form: C % 4 == 0, 16-byte aligned views, rows * C / 4 < 2^31 (32-bit index arithmetic: no 64-bit divisions).
// dx may alias dy (in-place ReLU backward of a conv output), hence no __restrict__.
__global__ void act_bwd_vec_kernel(const float* y, int ldy, const float* dy, int lddy, int C4, int act, float* dx, int lddx, int acc,
                                   unsigned total4) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
        const unsigned r = i / (unsigned)C4;
        const unsigned c = (i - r * (unsigned)C4) * 4u;
        const f32x4 yy = *reinterpret_cast<const f32x4*>(y + (size_t)r * ldy + c);
        f32x4 g = *reinterpret_cast<const f32x4*>(dy + (size_t)r * lddy + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (act == 1) g[k] = yy[k] > 0.f ? g[k] : 0.f;
            else if (act == 2) g[k] = g[k] * yy[k] * (1.f - yy[k]);
        }
        f32x4* d = reinterpret_cast<f32x4*>(dx + (size_t)r * lddx + c);
        if (acc) g += *d;
        *d = g;
    }
}

__global__ void blend_fwd_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, const float* __restrict__ occ,
                                 int ldo, int C, float* __restrict__ y, int ldy, long long total) {
    GRID_STRIDE(i, total) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        const float o = occ[(size_t)r * ldo];
        const float bv = b ? b[(size_t)r * ldb + c] : 0.f;
        y[(size_t)r * ldy + c] = a[(size_t)r * lda + c] * o + bv * (1.f - o);
    }
}

// one wave per (row, 64-channel chunk): docc reduces over channels with shuffles
__global__ __launch_bounds__(256) void blend_bwd_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                       const float* __restrict__ occ, int ldo, const float* __restrict__ dy, int lddy,
                                                       long long rows, int C, float* __restrict__ da, int ldda, float* __restrict__ db,
                                                       int lddb, float* __restrict__ docc, int lddo) {
    const int lane = threadIdx.x & 63;
    const long long wave_id = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int chunks = (C + 63) / 64;
    for (long long wi = wave_id; wi < rows * chunks; wi += nwaves) {
        const long long r = wi / chunks;
        const int c = (int)(wi - r * chunks) * 64 + lane;
        float part = 0.f;
        if (c < C) {
            const float o = occ[(size_t)r * ldo];
            const float g = dy[(size_t)r * lddy + c];
            const float av = a[(size_t)r * lda + c];
            const float bv = b ? b[(size_t)r * ldb + c] : 0.f;
            if (da) da[(size_t)r * ldda + c] += g * o;
            if (db) db[(size_t)r * lddb + c] += g * (1.f - o);
            part = g * (av - bv);
        }
        if (docc) {
            part = wave_sum(part);
            if (lane == 0) atomicAdd(docc + (size_t)r * lddo, part);
        }
    }
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ldx, long long rows, int C, float* __restrict__ out,
                                                    int rows_per_block) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const long long r0 = (long long)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float s = 0.f;
    if (c < C)
        for (long long r = r0 + wave; r < r1; r += 4) s += x[(size_t)r * ldx + c];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < C) atomicAdd(out + c, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}

// AntiAliasInterpolation2d: out[n,oy,ox,c] = sum_{i,j} K[i,j] * in[n,c,stride*oy+i-k/2, stride*ox+j-k/2] (zero padded)
__global__ void antialias_kernel(const float* __restrict__ x, int C, int H, int W, const float* __restrict__ kern, int k, int stride,
                                 float* __restrict__ y, int ldy, int Ho, int Wo, long long total) {
    const int ka = k / 2;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const long long opix = i / C;
        const int ox = (int)(opix % Wo);
        const long long t = opix / Wo;
        const int oy = (int)(t % Ho);
        const long long n = t / Ho;
        const float* src = x + ((size_t)n * C + c) * H * W;
        float acc = 0.f;
        for (int a = 0; a < k; ++a) {
            const int iy = stride * oy + a - ka;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int b = 0; b < k; ++b) {
                const int ix = stride * ox + b - ka;
                if ((unsigned)ix < (unsigned)W) acc += kern[a * k + b] * src[(size_t)iy * W + ix];
            }
        }
        y[(size_t)opix * ldy + c] = acc;
    }
}

}  // namespace

extern "C" int mrfa_avgpool2_fwd(void* stream, const float* x, int ldx, int N, int H, int W, int C, float* y, int ldy) {
    MRFA_CHECK_ARG(x && y && (H % 2) == 0 && (W % 2) == 0, "avgpool2: bad args");
    const long long total = (long long)N * (H / 2) * (W / 2) * C;
    hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, H, W, C, y, ldy, total);
    MRFA_CHECK_LAUNCH("avgpool2_fwd");
    return 0;
}

extern "C" int mrfa_sumpool2_acc(void* stream, const float* x, int ldx, int N, int Ho, int Wo, int C, float* y, int ldy, float mul) {
    MRFA_CHECK_ARG(x && y, "sumpool2: bad args");
    const long long total = (long long)N * Ho * Wo * C;
    hipLaunchKernelGGL(sumpool2_acc_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, Ho, Wo, C, y, ldy, mul,
                       total);
    MRFA_CHECK_LAUNCH("sumpool2_acc");
    return 0;
}

extern "C" int mrfa_unpool2_acc(void* stream, const float* dy, int lddy, int N, int Ho, int Wo, int C, float* dx, int lddx, float mul) {
    MRFA_CHECK_ARG(dy && dx, "unpool2: bad args");
    const long long total = (long long)N * Ho * Wo * 4 * C;
    hipLaunchKernelGGL(unpool2_acc_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, lddy, Ho, Wo, C, dx, lddx,
                       mul, total);
    MRFA_CHECK_LAUNCH("unpool2_acc");
    return 0;
}

extern "C" int mrfa_bias_act(void* stream, const float* x, int ldx, long long rows, int C, const float* bias, int act, float* y, int ldy,
                             double* stats) {
    MRFA_CHECK_ARG(x && y && rows > 0 && C > 0, "bias_act: bad args");
    const long long total = rows * C;
    hipLaunchKernelGGL(bias_act_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, C, bias, act, y, ldy,
                       total);
    MRFA_CHECK_LAUNCH("bias_act");
    if (stats) return mrfa_bn_stats(stream, y, ldy, rows, C, stats);
    return 0;
}

extern "C" int mrfa_act_bwd(void* stream, const float* y, int ldy, const float* dy, int lddy, long long rows, int C, int act, float* dx,
                            int lddx, int accumulate) {
    MRFA_CHECK_ARG(y && dy && dx && rows > 0 && C > 0, "act_bwd: bad args");
    const long long total = rows * C;
    if (C % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && aligned16(y) && aligned16(dy) && aligned16(dx) &&
        total / 4 < (1ll << 31)) {
        hipLaunchKernelGGL(act_bwd_vec_kernel, dim3(stream_grid(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, y, ldy, dy, lddy, C / 4,
                           act, dx, lddx, accumulate, (unsigned)(total / 4));
        MRFA_CHECK_LAUNCH("act_bwd(vec)");
        return 0;
    }
    hipLaunchKernelGGL(act_bwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, y, ldy, dy, lddy, rows, C, act, dx,
                       lddx, accumulate, total);
    MRFA_CHECK_LAUNCH("act_bwd");
    return 0;
}

extern "C" int mrfa_blend_fwd(void* stream, const float* a, int lda, const float* b, int ldb, const float* occ, int ldo, long long rows,
                              int C, float* y, int ldy) {
    MRFA_CHECK_ARG(a && occ && y && rows > 0 && C > 0, "blend_fwd: bad args");
    const long long total = rows * C;
    hipLaunchKernelGGL(blend_fwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, occ, ldo, C, y,
                       ldy, total);
    MRFA_CHECK_LAUNCH("blend_fwd");
    return 0;
}

extern "C" int mrfa_blend_bwd(void* stream, const float* a, int lda, const float* b, int ldb, const float* occ, int ldo, const float* dy,
                              int lddy, long long rows, int C, float* da, int ldda, float* db, int lddb, float* docc, int lddo) {
    MRFA_CHECK_ARG(a && occ && dy && rows > 0 && C > 0, "blend_bwd: bad args");
    const long long waves = rows * cdiv(C, 64);
    hipLaunchKernelGGL(blend_bwd_kernel, dim3(stream_grid(waves * 64, 256)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, occ, ldo, dy,
                       lddy, rows, C, da, ldda, db, lddb, docc, lddo);
    MRFA_CHECK_LAUNCH("blend_bwd");
    return 0;
}

extern "C" int mrfa_colsum(void* stream, const float* x, int ldx, long long rows, int C, float* out) {
    MRFA_CHECK_ARG(x && out && rows > 0 && C > 0, "colsum: bad args");
    const int chunks = cdiv(C, 64);
    long long want = (1024 + chunks - 1) / chunks;
    long long rpb = (rows + want - 1) / want;
    if (rpb < 64) rpb = 64;
    dim3 grid(chunks, cdiv(rows, rpb));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, rows, C, out, (int)rpb);
    MRFA_CHECK_LAUNCH("colsum");
    return 0;
}

extern "C" int mrfa_antialias_down(void* stream, const float* x_nchw, int N, int C, int H, int W, const float* kern, int k, int stride,
                                   float* y, int ldy) {
    MRFA_CHECK_ARG(x_nchw && kern && y && (H % stride) == 0 && (W % stride) == 0, "antialias_down: bad args");
    const int Ho = H / stride, Wo = W / stride;
    const long long total = (long long)N * Ho * Wo * C;
    hipLaunchKernelGGL(antialias_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x_nchw, C, H, W, kern, k, stride, y,
                       ldy, Ho, Wo, total);
    MRFA_CHECK_LAUNCH("antialias_down");
    return 0;
}
