// Layout kernels: conv-weight (un)packing between the reference's OIHW parameters and the MFMA kernel layouts,
// NCHW <-> NHWC conversion at the module boundary, strided view copies.  All HBM-bound, all tiny next to the convs.
#include "common.h"

namespace {

// one thread per destination element of the PADDED destination; pads are written as zero
__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int R, int S,
                                   int mode, int CoP, int CiP, int KP, long long total) {
    const int T = R * S;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (mode == 0) {            // [tap][CoP][CiP]
            const int ci = (int)(i % CiP);
            const long long t1 = i / CiP;
            const int co = (int)(t1 % CoP);
            const int tap = (int)(t1 / CoP);
            if (ci < Cin && co < Cout) v = src[((size_t)co * Cin + ci) * T + tap];
        } else if (mode == 1) {     // [CoP][KP], k = tap*Cin + ci
            const int k = (int)(i % KP);
            const int co = (int)(i / KP);
            if (k < T * Cin && co < Cout) {
                const int tap = k / Cin, ci = k - tap * Cin;
                v = src[((size_t)co * Cin + ci) * T + tap];
            }
        } else if (mode == 2) {     // dgrad chunked: [tap'][CiP(128-padded Cin)][CoP(32-padded Cout)], tap' flipped
            const int co = (int)(i % CoP);
            const long long t1 = i / CoP;
            const int ci = (int)(t1 % CiP);
            const int tapf = (int)(t1 / CiP);
            const int tap = T - 1 - tapf;
            if (ci < Cin && co < Cout) v = src[((size_t)co * Cin + ci) * T + tap];
        } else if (mode == 3) {     // dgrad flat: [CiP][KP], k = tap'*Cout + co
            const int k = (int)(i % KP);
            const int ci = (int)(i / KP);
            if (k < T * Cout && ci < Cin) {
                const int tapf = k / Cout, co = k - tapf * Cout;
                const int tap = T - 1 - tapf;
                v = src[((size_t)co * Cin + ci) * T + tap];
            }
        } else if (mode == 5) {     // few-output direct kernels: [Cout][tap][Cin]
            const int ci = (int)(i % Cin);
            const long long t1 = i / Cin;
            const int tap = (int)(t1 % T);
            const int co = (int)(t1 / T);
            v = src[((size_t)co * Cin + ci) * T + tap];
        } else if (mode == 7) {     // few-INPUT data gradient as a few-output conv over dY: [Cin][tap'][Cout], tap' flipped
            const int co = (int)(i % Cout);
            const long long t1 = i / Cout;
            const int tapf = (int)(t1 % T);
            const int ci = (int)(t1 / T);
            v = src[((size_t)co * Cin + ci) * T + (T - 1 - tapf)];
        }
        dst[i] = v;
    }
}

// grad [Cout][tap][Cin] -> OIHW, dst += src
__global__ void unpack_wgrad_fewout_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int T, long long total,
                                           int overwrite) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % T);
        const long long t1 = i / T;
        const int ci = (int)(t1 % Cin);
        const int co = (int)(t1 / Cin);
        const float v = src[((size_t)co * T + tap) * Cin + ci];
        if (overwrite) dst[i] = v; else atomicAdd(dst + i, v);
    }
}

// grad [tap][Cout][Cin] -> OIHW, dst += src   (one thread per OIHW element)
__global__ void unpack_wgrad_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int T, long long total,
                                    int overwrite) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % T);
        const long long t1 = i / T;
        const int ci = (int)(t1 % Cin);
        const int co = (int)(t1 / Cin);
        const float v = src[((size_t)tap * Cout + co) * Cin + ci];
        if (overwrite) dst[i] = v; else atomicAdd(dst + i, v);
    }
}

// NCHW -> NHWC view, tiled through LDS so both sides are coalesced (32 pixels x 32 channels per tile)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int ldd, int C, int HW, int acc) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, pp = p0 + tx;
        tile[j][tx] = (c < C && pp < HW) ? src[((size_t)n * C + c) * HW + pp] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int pp = p0 + j, c = c0 + tx;
        if (c < C && pp < HW) {
            float* d = dst + ((size_t)n * HW + pp) * ldd + c;
            *d = acc ? (*d + tile[tx][j]) : tile[tx][j];
        }
    }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int C, int HW, int acc) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int pp = p0 + j, c = c0 + tx;
        tile[j][tx] = (c < C && pp < HW) ? src[((size_t)n * HW + pp) * lds_ + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, pp = p0 + tx;
        if (c < C && pp < HW) {
            float* d = dst + ((size_t)n * C + c) * HW + pp;
            *d = acc ? (*d + tile[tx][j]) : tile[tx][j];
        }
    }
}

__global__ void copy_view_kernel(const float* __restrict__ x, int ldx, long long rows, int C, float* __restrict__ y, int ldy,
                                 float mul, int acc) {
    const long long total = rows * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        const float v = x[(size_t)r * ldx + c] * mul;
        float* d = y + (size_t)r * ldy + c;
        *d = acc ? (*d + v) : v;
    }
}

__global__ void copy_view_vec_kernel(const float* __restrict__ x, int ldx, int C4, float* __restrict__ y, int ldy, float mul, int acc,
                                     unsigned total4) {
    chain_prio();
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
        const unsigned r = i / (unsigned)C4;
        const unsigned c = (i - r * (unsigned)C4) * 4u;
        f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r * ldx + c) * mul;
        f32x4* d = reinterpret_cast<f32x4*>(y + (size_t)r * ldy + c);
        if (acc) v += *d;
        *d = v;
    }
}

}  // namespace

extern "C" int mrfa_pack_conv_weight(void* stream, const float* src, float* dst, int Cout, int Cin, int R, int S, int mode) {
    hipStream_t st = (hipStream_t)stream;
    MRFA_CHECK_ARG(src && dst && Cout > 0 && Cin > 0 && R > 0 && S > 0, "pack_conv_weight: bad args");
    const int T = R * S;
    long long total;
    int CoP = 0, CiP = 0, KP = 0;
    const int overwrite = (mode & 16) ? 1 : 0;      // modes 4|16, 6|16: dst = unpacked gradient instead of dst += ...
    mode &= 15;
    if (mode == 0) { CoP = cdiv(Cout, 128) * 128; CiP = cdiv(Cin, 32) * 32; total = (long long)T * CoP * CiP; }
    else if (mode == 1) { CoP = cdiv(Cout, 128) * 128; KP = cdiv((long long)T * Cin, 32) * 32; total = (long long)CoP * KP; }
    else if (mode == 2) { CiP = cdiv(Cin, 128) * 128; CoP = cdiv(Cout, 32) * 32; total = (long long)T * CiP * CoP; }
    else if (mode == 3) { CiP = cdiv(Cin, 128) * 128; KP = cdiv((long long)T * Cout, 32) * 32; total = (long long)CiP * KP; }
    else if (mode == 4) {
        total = (long long)Cout * Cin * T;
        hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, st, src, dst, Cout, Cin, T, total, overwrite);
        MRFA_CHECK_LAUNCH("unpack_wgrad");
        return 0;
    } else if (mode == 6) {
        total = (long long)Cout * Cin * T;
        hipLaunchKernelGGL(unpack_wgrad_fewout_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, st, src, dst, Cout, Cin, T, total, overwrite);
        MRFA_CHECK_LAUNCH("unpack_wgrad_fewout");
        return 0;
    } else if (mode == 5 || mode == 7) {
        total = (long long)Cout * Cin * T;
    } else { mrfa_set_error("pack_conv_weight: unknown mode %d", mode); return 1; }
    hipLaunchKernelGGL(pack_weight_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, st, src, dst, Cout, Cin, R, S, mode, CoP, CiP,
                       KP, total);
    MRFA_CHECK_LAUNCH("pack_conv_weight");
    return 0;
}

extern "C" int mrfa_build_ktab(int* tab, int C, int R, int S, int pad, int flip) {
    (void)flip;
    if (!tab || C <= 0 || C >= 32768 || R > 15 || S > 15) { mrfa_set_error("build_ktab: bad args"); return 1; }
    const int K = R * S * C;
    const int KP = cdiv(K, 32) * 32;
    for (int k = 0; k < KP; ++k) {
        if (k >= K) { tab[k] = -1; continue; }
        const int tap = k / C, c = k - tap * C;
        const int r = tap / S, s = tap - r * S;
        tab[k] = ((r - pad) + 128) | (((s - pad) + 128) << 8) | (c << 16);
    }
    return 0;
}

extern "C" int mrfa_nchw_to_nhwc(void* stream, const float* src, float* dst, int ldd, int N, int C, int H, int W, int accumulate) {
    MRFA_CHECK_ARG(src && dst && ldd >= C, "nchw_to_nhwc: bad args");
    dim3 grid(cdiv((long long)H * W, 32), cdiv(C, 32), N);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, ldd, C, H * W, accumulate);
    MRFA_CHECK_LAUNCH("nchw_to_nhwc");
    return 0;
}

extern "C" int mrfa_nhwc_to_nchw(void* stream, const float* src, int lds_, float* dst, int N, int C, int H, int W, int accumulate) {
    MRFA_CHECK_ARG(src && dst && lds_ >= C, "nhwc_to_nchw: bad args");
    dim3 grid(cdiv((long long)H * W, 32), cdiv(C, 32), N);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, lds_, dst, C, H * W, accumulate);
    MRFA_CHECK_LAUNCH("nhwc_to_nchw");
    return 0;
}

__global__ void timestamp_kernel(unsigned long long* dst) { *dst = wall_clock64(); }

extern "C" int mrfa_timestamp(void* stream, unsigned long long* dst) {
    MRFA_CHECK_ARG(dst, "timestamp: null dst");
    hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dst);
    MRFA_CHECK_LAUNCH("timestamp");
    return 0;
}

extern "C" int mrfa_copy_view(void* stream, const float* x, int ldx, long long rows, int C, float* y, int ldy, float mul, int accumulate) {
    MRFA_CHECK_ARG(x && y && rows >= 0 && C > 0, "copy_view: bad args");
    if (rows == 0) return 0;
    if (C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && aligned16(x) && aligned16(y) && rows * C / 4 < (1ll << 31)) {
        hipLaunchKernelGGL(copy_view_vec_kernel, dim3(stream_grid(rows * C / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, C / 4, y, ldy,
                           mul, accumulate, (unsigned)(rows * C / 4));
        MRFA_CHECK_LAUNCH("copy_view(vec)");
        return 0;
    }
    hipLaunchKernelGGL(copy_view_kernel, dim3(stream_grid(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, C, y, ldy,
                       mul, accumulate);
    MRFA_CHECK_LAUNCH("copy_view");
    return 0;
}
