// K22: kernels of the reference's training losses (reference modules/model.py:26-141,219-254):
//   * 2x2 max-pooling of the VGG19 perceptual network (torchvision vgg19.features[4,9,18,27], model.py:88-105) forward and
//     backward; ties (frequent after ReLU: all-zero windows) route the gradient to the FIRST maximum in (dy,dx) scan order, as
//     ATen's max_pool2d_with_indices does;
//   * mean|x - y| between two feature maps (the perceptual term, model.py:225-227) as one reduction pass, and its gradient;
//   * the gradient of AntiAliasInterpolation2d (ImagePyramide of the GENERATED image, model.py:123-141, util.py:318-326).
// All HBM-bound streaming kernels, float4 where the layout allows.
#include "common.h"

namespace {

#define GRID_STRIDE_U(i, n) for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += gridDim.x * blockDim.x)

__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, int ldx, int H, int W, int C4, float* __restrict__ y, int ldy, unsigned total4) {
    const int Ho = H / 2, Wo = W / 2;
    GRID_STRIDE_U(i, total4) {
        const unsigned opix = i / (unsigned)C4;
        const unsigned c = (i - opix * (unsigned)C4) * 4u;
        const unsigned ox = opix % (unsigned)Wo, t = opix / (unsigned)Wo;
        const unsigned oy = t % (unsigned)Ho, n = t / (unsigned)Ho;
        const float* b = x + (((size_t)n * H + 2 * oy) * W + 2 * ox) * ldx + c;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(b), v1 = *reinterpret_cast<const f32x4*>(b + ldx);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(b + (size_t)W * ldx), v3 = *reinterpret_cast<const f32x4*>(b + (size_t)(W + 1) * ldx);
        f32x4 m;
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = fmaxf(fmaxf(v0[k], v1[k]), fmaxf(v2[k], v3[k]));
        *reinterpret_cast<f32x4*>(y + (size_t)opix * ldy + c) = m;
    }
}

// dx[first argmax of the window] += dy
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, int ldx, int H, int W, int C4, const float* __restrict__ dy, int lddy,
                                    float* __restrict__ dx, int lddx, unsigned total4) {
    const int Ho = H / 2, Wo = W / 2;
    GRID_STRIDE_U(i, total4) {
        const unsigned opix = i / (unsigned)C4;
        const unsigned c = (i - opix * (unsigned)C4) * 4u;
        const unsigned ox = opix % (unsigned)Wo, t = opix / (unsigned)Wo;
        const unsigned oy = t % (unsigned)Ho, n = t / (unsigned)Ho;
        const size_t p0 = ((size_t)n * H + 2 * oy) * W + 2 * ox;
        const size_t off[4] = {p0, p0 + 1, p0 + W, p0 + W + 1};
        f32x4 v[4], g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] = *reinterpret_cast<const f32x4*>(x + off[q] * ldx + c);
            g[q] = *reinterpret_cast<const f32x4*>(dx + off[q] * lddx + c);
        }
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + (size_t)opix * lddy + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int best = 0;
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (v[q][k] > v[best][k]) best = q;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q == best) g[q][k] += d[k];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(dx + off[q] * lddx + c) = g[q];
    }
}

// nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of torchvision's resnet18 stem (BGMotionPredictor, bg_motion_predictor.py:12):
// out (Ho = (H-1)/2+1) = max over the 3x3 window clipped to the image; backward: first maximum in (row, column) scan order
__global__ void maxpool3s2_fwd_kernel(const float* __restrict__ x, int ldx, int H, int W, int Ho, int Wo, int C4, float* __restrict__ y, int ldy,
                                      unsigned total4) {
    GRID_STRIDE_U(i, total4) {
        const unsigned opix = i / (unsigned)C4;
        const unsigned c = (i - opix * (unsigned)C4) * 4u;
        const int ox = (int)(opix % (unsigned)Wo);
        const unsigned t = opix / (unsigned)Wo;
        const int oy = (int)(t % (unsigned)Ho);
        const unsigned n = t / (unsigned)Ho;
        f32x4 m = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        for (int a = 0; a < 3; ++a) {
            const int iy = 2 * oy - 1 + a;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int b = 0; b < 3; ++b) {
                const int ix = 2 * ox - 1 + b;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + iy) * W + ix) * ldx + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], v[k]);
            }
        }
        *reinterpret_cast<f32x4*>(y + (size_t)opix * ldy + c) = m;
    }
}

// windows overlap (stride 2 < kernel 3): the scatter into dx is atomic
__global__ void maxpool3s2_bwd_kernel(const float* __restrict__ x, int ldx, int H, int W, int Ho, int Wo, int C4, const float* __restrict__ dy,
                                      int lddy, float* __restrict__ dx, int lddx, unsigned total4) {
    GRID_STRIDE_U(i, total4) {
        const unsigned opix = i / (unsigned)C4;
        const unsigned c = (i - opix * (unsigned)C4) * 4u;
        const int ox = (int)(opix % (unsigned)Wo);
        const unsigned t = opix / (unsigned)Wo;
        const int oy = (int)(t % (unsigned)Ho);
        const unsigned n = t / (unsigned)Ho;
        f32x4 m = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        int arg[4] = {-1, -1, -1, -1};
        for (int a = 0; a < 3; ++a) {
            const int iy = 2 * oy - 1 + a;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int b = 0; b < 3; ++b) {
                const int ix = 2 * ox - 1 + b;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + iy) * W + ix) * ldx + c);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (v[k] > m[k] || arg[k] < 0) { m[k] = v[k]; arg[k] = iy * W + ix; }
            }
        }
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + (size_t)opix * lddy + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(dx + ((size_t)n * H * W + arg[k]) * lddx + c + k, d[k]);
    }
}

// out[0] += coef * sum |x - y| (fp64)      /      dx += scale[0] * coef * sign(x - y)
__global__ __launch_bounds__(256) void l1_diff_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y, int ldy, int C4,
                                                         double* __restrict__ out, double coef, unsigned total4) {
    __shared__ float red[4];
    float s = 0.f;
    GRID_STRIDE_U(i, total4) {
        const unsigned r = i / (unsigned)C4;
        const unsigned c = (i - r * (unsigned)C4) * 4u;
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + (size_t)r * ldx + c), b = *reinterpret_cast<const f32x4*>(y + (size_t)r * ldy + c);
        s += fabsf(a[0] - b[0]) + fabsf(a[1] - b[1]) + fabsf(a[2] - b[2]) + fabsf(a[3] - b[3]);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, coef * (double)(red[0] + red[1] + red[2] + red[3]));
}

__global__ void l1_diff_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y, int ldy, int C4, const float* __restrict__ gscale,
                                   float coef, float* __restrict__ dx, int lddx, unsigned total4) {
    const float g = (gscale ? gscale[0] : 1.f) * coef;
    GRID_STRIDE_U(i, total4) {
        const unsigned r = i / (unsigned)C4;
        const unsigned c = (i - r * (unsigned)C4) * 4u;
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + (size_t)r * ldx + c), b = *reinterpret_cast<const f32x4*>(y + (size_t)r * ldy + c);
        f32x4* q = reinterpret_cast<f32x4*>(dx + (size_t)r * lddx + c);
        f32x4 cur = *q;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = a[k] - b[k];
            cur[k] += d > 0.f ? g : (d < 0.f ? -g : 0.f);
        }
        *q = cur;
    }
}

// dx (NCHW image gradient) += transpose of antialias_down: y[n,oy,ox,c] = sum_{i,j} k[i][j] x[n,c,oy*s+i-ka,ox*s+j-ka]
// one thread per input pixel (n,c,Y,X): gathers the outputs whose window covers it
__global__ void antialias_down_bwd_kernel(const float* __restrict__ dy, int lddy, int N, int C, int H, int W, const float* __restrict__ kern, int k,
                                          int stride, float* __restrict__ dx, long long total) {
    const int ka = k / 2;
    const int Ho = H / stride, Wo = W / stride;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int X = (int)(i % W);
        long long t = i / W;
        const int Y = (int)(t % H);
        t /= H;
        const int c = (int)(t % C);
        const int n = (int)(t / C);
        // oy*s + a - ka == Y  ->  oy in [ceil((Y + ka - k + 1)/s), floor((Y + ka)/s)]
        int oy0 = Y + ka - k + 1;
        oy0 = oy0 <= 0 ? 0 : (oy0 + stride - 1) / stride;
        const int oy1 = min(Ho - 1, (Y + ka) / stride);
        int ox0 = X + ka - k + 1;
        ox0 = ox0 <= 0 ? 0 : (ox0 + stride - 1) / stride;
        const int ox1 = min(Wo - 1, (X + ka) / stride);
        float s = 0.f;
        for (int oy = oy0; oy <= oy1; ++oy) {
            const int a = Y + ka - oy * stride;
            for (int ox = ox0; ox <= ox1; ++ox) {
                const int b = X + ka - ox * stride;
                s += kern[a * k + b] * dy[(((size_t)n * Ho + oy) * Wo + ox) * lddy + c];
            }
        }
        dx[i] += s;
    }
}

bool v4(const void* p, int ld) { return aligned16(p) && (ld % 4) == 0; }

}  // namespace

extern "C" int mrfa_maxpool2_fwd(void* stream, const float* x, int ldx, int N, int H, int W, int C, float* y, int ldy) {
    MRFA_CHECK_ARG(x && y && N > 0 && (H % 2) == 0 && (W % 2) == 0 && C > 0, "maxpool2_fwd: bad args");
    MRFA_CHECK_ARG(C % 4 == 0 && v4(x, ldx) && v4(y, ldy), "maxpool2_fwd: needs C %% 4 == 0 and 16-byte aligned views");
    const long long total4 = (long long)N * (H / 2) * (W / 2) * (C / 4);
    MRFA_CHECK_ARG(total4 < (1ll << 31), "maxpool2_fwd: tensor too large");
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(stream_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, H, W, C / 4, y, ldy,
                       (unsigned)total4);
    MRFA_CHECK_LAUNCH("maxpool2_fwd");
    return 0;
}

extern "C" int mrfa_maxpool2_bwd(void* stream, const float* x, int ldx, int N, int H, int W, int C, const float* dy, int lddy, float* dx, int lddx) {
    MRFA_CHECK_ARG(x && dy && dx && N > 0 && (H % 2) == 0 && (W % 2) == 0 && C > 0, "maxpool2_bwd: bad args");
    MRFA_CHECK_ARG(C % 4 == 0 && v4(x, ldx) && v4(dy, lddy) && v4(dx, lddx), "maxpool2_bwd: needs C %% 4 == 0 and 16-byte aligned views");
    const long long total4 = (long long)N * (H / 2) * (W / 2) * (C / 4);
    MRFA_CHECK_ARG(total4 < (1ll << 31), "maxpool2_bwd: tensor too large");
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(stream_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, H, W, C / 4, dy, lddy, dx,
                       lddx, (unsigned)total4);
    MRFA_CHECK_LAUNCH("maxpool2_bwd");
    return 0;
}

extern "C" int mrfa_maxpool3s2_fwd(void* stream, const float* x, int ldx, int N, int H, int W, int C, float* y, int ldy) {
    MRFA_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0, "maxpool3s2_fwd: bad args");
    MRFA_CHECK_ARG(C % 4 == 0 && v4(x, ldx) && v4(y, ldy), "maxpool3s2_fwd: needs C %% 4 == 0 and 16-byte aligned views");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total4 = (long long)N * Ho * Wo * (C / 4);
    MRFA_CHECK_ARG(total4 < (1ll << 31), "maxpool3s2_fwd: tensor too large");
    hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3(stream_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, H, W, Ho, Wo, C / 4, y,
                       ldy, (unsigned)total4);
    MRFA_CHECK_LAUNCH("maxpool3s2_fwd");
    return 0;
}

extern "C" int mrfa_maxpool3s2_bwd(void* stream, const float* x, int ldx, int N, int H, int W, int C, const float* dy, int lddy, float* dx,
                                   int lddx) {
    MRFA_CHECK_ARG(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0, "maxpool3s2_bwd: bad args");
    MRFA_CHECK_ARG(C % 4 == 0 && v4(x, ldx) && v4(dy, lddy), "maxpool3s2_bwd: needs C %% 4 == 0 and 16-byte aligned views");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total4 = (long long)N * Ho * Wo * (C / 4);
    MRFA_CHECK_ARG(total4 < (1ll << 31), "maxpool3s2_bwd: tensor too large");
    hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(stream_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, H, W, Ho, Wo, C / 4, dy,
                       lddy, dx, lddx, (unsigned)total4);
    MRFA_CHECK_LAUNCH("maxpool3s2_bwd");
    return 0;
}

extern "C" int mrfa_l1_diff_fwd(void* stream, const float* x, int ldx, const float* y, int ldy, long long rows, int C, double coef,
                                double* out_sum) {
    MRFA_CHECK_ARG(x && y && out_sum && rows > 0 && C > 0, "l1_diff_fwd: bad args");
    MRFA_CHECK_ARG(C % 4 == 0 && v4(x, ldx) && v4(y, ldy) && rows * (C / 4) < (1ll << 31), "l1_diff_fwd: needs C %% 4 == 0, aligned views");
    const long long total4 = rows * (C / 4);
    long long g = (total4 + 256 * 8 - 1) / (256 * 8);              // >= 8 float4 per thread: one fp64 atomic per workgroup
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(l1_diff_fwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, C / 4, out_sum, coef, (unsigned)total4);
    MRFA_CHECK_LAUNCH("l1_diff_fwd");
    return 0;
}

extern "C" int mrfa_l1_diff_bwd(void* stream, const float* x, int ldx, const float* y, int ldy, long long rows, int C, const float* gscale,
                                float coef, float* dx, int lddx) {
    MRFA_CHECK_ARG(x && y && dx && rows > 0 && C > 0, "l1_diff_bwd: bad args");
    MRFA_CHECK_ARG(C % 4 == 0 && v4(x, ldx) && v4(y, ldy) && v4(dx, lddx) && rows * (C / 4) < (1ll << 31),
                   "l1_diff_bwd: needs C %% 4 == 0, aligned views");
    const long long total4 = rows * (C / 4);
    hipLaunchKernelGGL(l1_diff_bwd_kernel, dim3(stream_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, C / 4, gscale, coef,
                       dx, lddx, (unsigned)total4);
    MRFA_CHECK_LAUNCH("l1_diff_bwd");
    return 0;
}

extern "C" int mrfa_antialias_down_bwd(void* stream, const float* dy, int lddy, int N, int C, int H, int W, const float* kern, int k, int stride,
                                       float* dx) {
    MRFA_CHECK_ARG(dy && kern && dx && N > 0 && C > 0 && k >= 1 && (k & 1) && stride >= 1 && H % stride == 0 && W % stride == 0,
                   "antialias_down_bwd: bad args (odd kernel, H and W multiples of the stride)");
    const long long total = (long long)N * C * H * W;
    hipLaunchKernelGGL(antialias_down_bwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, lddy, N, C, H, W, kern, k,
                       stride, dx, total);
    MRFA_CHECK_LAUNCH("antialias_down_bwd");
    return 0;
}
