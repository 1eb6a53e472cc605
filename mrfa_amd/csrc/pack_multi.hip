// Batched weight (un)packing: every convolution's OIHW parameter -> all of its kernel layouts in ONE launch per <= 48
// convolutions (and the reverse for the weight gradients), instead of one small launch per (convolution, layout).
//
// The single-tensor kernels of layout.hip spend their time on 64-bit index divisions and 36-byte-strided reads
// (170 us for the 4.7 M-element 1024->512 3x3 weight = 0.2 TB/s) and cost 270 launches per training step.  Here a
// workgroup owns a (TCO output channels) x (32 input channels) x (all taps) tile of one convolution:
//   * OIHW side: each output channel's 32*T floats are contiguous        -> coalesced 128 B+ segments,
//   * the tile is transposed through LDS (row stride odd: conflict free),
//   * kernel-layout side: 32 consecutive ci (modes 0, 1, 5) or TCO consecutive co (modes 2, 3, 7) per segment.
// Padding elements of the destination layouts are never written: the host allocates those buffers zero-filled.
// The descriptor table travels BY VALUE in the kernel arguments (<= 4 KB), so a launch is self-contained and can be
// recorded in a hipGraph without any host->device copy.
#include <type_traits>
#include "common.h"

namespace {

constexpr int TCI = 32;
constexpr int MAXD = MRFA_PACK_MAX_DESCS;
constexpr int LDS_FLOATS = 12832;              // 51.3 KB: 32 x (32*9 + 1), 16 x (32*25 + 1) or 8 x (32*49 + 1)

struct PackArgs {
    int n;
    int tile_prefix[MAXD + 1];                 // workgroup b handles desc d with tile_prefix[d] <= b < tile_prefix[d+1]
    mrfa_pack_desc d[MAXD];
};

__host__ __device__ inline int tco_for(int T) { return T <= 9 ? 32 : (T <= 25 ? 16 : 8); }
__host__ __device__ inline int rup(int a, int b) { return (a + b - 1) / b * b; }

// destination index of weight element (co, ci, tap t) in kernel layout `mode` (see mrfa_pack_conv_weight)
__device__ __forceinline__ long long dst_index(int mode, int co, int ci, int t, int Cout, int Cin, int T) {
    switch (mode) {
        case 0: return ((long long)t * rup(Cout, 128) + co) * rup(Cin, 32) + ci;
        case 1: return (long long)co * rup(T * Cin, 32) + t * Cin + ci;
        case 2: return ((long long)(T - 1 - t) * rup(Cin, 128) + ci) * rup(Cout, 32) + co;
        case 3: return (long long)ci * rup(T * Cout, 32) + (T - 1 - t) * Cout + co;
        case 5: return ((long long)co * T + t) * Cin + ci;
        default: return ((long long)ci * T + (T - 1 - t)) * Cout + co;          // 7
    }
}

__device__ __forceinline__ int find_desc(const int* prefix, int n, int b) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (prefix[mid] <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void pack_multi_kernel(const PackArgs a) {
    __shared__ float lds[LDS_FLOATS];
    const int di = find_desc(a.tile_prefix, a.n, blockIdx.x);
    const mrfa_pack_desc& d = a.d[di];
    const int T = d.R * d.S, Cout = d.Cout, Cin = d.Cin;
    const int TCO = tco_for(T);
    const int tiles_ci = (Cin + TCI - 1) / TCI;
    const int tile = blockIdx.x - a.tile_prefix[di];
    const int co0 = (tile / tiles_ci) * TCO, ci0 = (tile % tiles_ci) * TCI;
    const int nco = min(TCO, Cout - co0), nci = min(TCI, Cin - ci0);
    const int row = TCI * T + 1;                                   // odd LDS row stride
    const int seg = nci * T;                                       // contiguous floats per output channel in OIHW
    // OIHW -> LDS [co_l][ci_l * T + t]
    for (int i = threadIdx.x; i < nco * seg; i += 256) {
        const int co_l = i / seg, r = i - co_l * seg;
        lds[co_l * row + r] = d.src[((size_t)(co0 + co_l) * Cin + ci0) * T + r];
    }
    __syncthreads();
    for (int k = 0; k < d.ndst; ++k) {
        const int mode = d.mode[k];
        float* __restrict__ dst = d.dst[k];
        if (mode == 8 || mode == 9) {                              // bf16 pieces for the split-operand kernels
            unsigned short* __restrict__ d16 = reinterpret_cast<unsigned short*>(dst);
            const long long piece = mode == 8 ? (long long)T * rup(Cout, 128) * rup(Cin, 32) : (long long)T * rup(Cin, 128) * rup(Cout, 32);
            for (int i = threadIdx.x; i < nco * T * nci; i += 256) {
                int co_l, ci_l, t;
                if (mode == 8) { ci_l = i % nci; const int q = i / nci; t = q % T; co_l = q / T; }
                else { co_l = i % nco; const int q = i / nco; ci_l = q % nci; t = q / nci; }
                const float v = lds[co_l * row + ci_l * T + t];
                const unsigned b1 = __float_as_uint(v) & 0xffff0000u;
                const float r1 = v - __uint_as_float(b1);
                const unsigned b2 = __float_as_uint(r1) & 0xffff0000u;
                const float r2 = r1 - __uint_as_float(b2);
                const long long idx = dst_index(mode == 8 ? 0 : 2, co0 + co_l, ci0 + ci_l, t, Cout, Cin, T);
                d16[idx] = (unsigned short)(b1 >> 16);
                d16[piece + idx] = (unsigned short)(b2 >> 16);
                d16[2 * piece + idx] = (unsigned short)(__float_as_uint(r2) >> 16);
            }
            continue;
        }
        if (mode == 0 || mode == 1 || mode == 5) {                 // ci fastest
            for (int i = threadIdx.x; i < nco * T * nci; i += 256) {
                const int ci_l = i % nci, q = i / nci;
                const int t = q % T, co_l = q / T;
                dst[dst_index(mode, co0 + co_l, ci0 + ci_l, t, Cout, Cin, T)] = lds[co_l * row + ci_l * T + t];
            }
        } else {                                                   // co fastest (data-gradient layouts)
            for (int i = threadIdx.x; i < nci * T * nco; i += 256) {
                const int co_l = i % nco, q = i / nco;
                const int ci_l = q % nci, t = q / nci;
                dst[dst_index(mode, co0 + co_l, ci0 + ci_l, t, Cout, Cin, T)] = lds[co_l * row + ci_l * T + t];
            }
        }
    }
}

struct UnpackArgs {
    int n;
    int tile_prefix[MAXD + 1];
    mrfa_unpack_desc d[MAXD];
};

// dst (OIHW gradient) += src (accumulator in [tap][Cout][Cin], or [Cout][tap][Cin] when fewout)
__global__ __launch_bounds__(256) void unpack_multi_kernel(const UnpackArgs a) {
    __shared__ float lds[LDS_FLOATS];
    const int di = find_desc(a.tile_prefix, a.n, blockIdx.x);
    const mrfa_unpack_desc& d = a.d[di];
    const int T = d.T, Cout = d.Cout, Cin = d.Cin;
    const int TCO = tco_for(T);
    const int tiles_ci = (Cin + TCI - 1) / TCI;
    const int tile = blockIdx.x - a.tile_prefix[di];
    const int co0 = (tile / tiles_ci) * TCO, ci0 = (tile % tiles_ci) * TCI;
    const int nco = min(TCO, Cout - co0), nci = min(TCI, Cin - ci0);
    const int row = TCI * T + 1;
    const int seg = nci * T;
    for (int i = threadIdx.x; i < nco * T * nci; i += 256) {      // ci fastest on the accumulator side
        const int ci_l = i % nci, q = i / nci;
        const int t = q % T, co_l = q / T;
        const size_t s = d.fewout ? ((size_t)(co0 + co_l) * T + t) * Cin + ci0 + ci_l : ((size_t)t * Cout + co0 + co_l) * Cin + ci0 + ci_l;
        lds[co_l * row + ci_l * T + t] = d.src[s];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nco * seg; i += 256) {
        const int co_l = i / seg, r = i - co_l * seg;
        // atomic: the two keypoint-encoder passes of a training step run their backward on two streams and accumulate into
        // the same weight.grad (mrfa_amd/train.py HotPath.encode_pair)
        atomicAdd(d.dst + ((size_t)(co0 + co_l) * Cin + ci0) * T + r, lds[co_l * row + r]);
    }
}

template <typename Args, typename Desc, typename Kernel>
int launch_chunks(hipStream_t st, const Desc* descs, int n, Kernel kernel, const char* what) {
    for (int base = 0; base < n; base += MAXD) {
        Args a;
        a.n = min(MAXD, n - base);
        int tiles = 0;
        for (int i = 0; i < a.n; ++i) {
            const Desc& d = descs[base + i];
            a.d[i] = d;
            a.tile_prefix[i] = tiles;
            int T, Cout = d.Cout, Cin = d.Cin;
            if constexpr (std::is_same<Desc, mrfa_pack_desc>::value) T = d.R * d.S; else T = d.T;
            if (Cout <= 0 || Cin <= 0 || T <= 0 || tco_for(T) * (TCI * T + 1) > LDS_FLOATS) {
                mrfa_set_error("%s: desc %d: bad shape Cout=%d Cin=%d taps=%d", what, base + i, Cout, Cin, T);
                return 1;
            }
            tiles += cdiv(Cout, tco_for(T)) * cdiv(Cin, TCI);
        }
        a.tile_prefix[a.n] = tiles;
        hipLaunchKernelGGL(kernel, dim3(tiles), dim3(256), 0, st, a);
        MRFA_CHECK_LAUNCH(what);
    }
    return 0;
}

}  // namespace

extern "C" int mrfa_pack_conv_weights_multi(void* stream, const mrfa_pack_desc* descs, int n) {
    MRFA_CHECK_ARG(n >= 0 && (n == 0 || descs), "pack_conv_weights_multi: bad args");
    for (int i = 0; i < n; ++i) {
        MRFA_CHECK_ARG(descs[i].src && descs[i].ndst >= 1 && descs[i].ndst <= 3, "pack_conv_weights_multi: desc %d: null src or ndst %d", i,
                       descs[i].ndst);
        for (int k = 0; k < descs[i].ndst; ++k) {
            const int m = descs[i].mode[k];
            MRFA_CHECK_ARG(descs[i].dst[k] && (m == 0 || m == 1 || m == 2 || m == 3 || m == 5 || m == 7 || m == 8 || m == 9),
                           "pack_conv_weights_multi: desc %d: null dst or mode %d", i, m);
        }
    }
    return launch_chunks<PackArgs>((hipStream_t)stream, descs, n, pack_multi_kernel, "pack_conv_weights_multi");
}

extern "C" int mrfa_unpack_wgrads_multi(void* stream, const mrfa_unpack_desc* descs, int n) {
    MRFA_CHECK_ARG(n >= 0 && (n == 0 || descs), "unpack_wgrads_multi: bad args");
    for (int i = 0; i < n; ++i) MRFA_CHECK_ARG(descs[i].src && descs[i].dst, "unpack_wgrads_multi: desc %d: null pointer", i);
    return launch_chunks<UnpackArgs>((hipStream_t)stream, descs, n, unpack_multi_kernel, "unpack_wgrads_multi");
}
