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
// The descriptor table travels BY VALUE in the kernel arguments (MRFA_PACK_MAX_DESCS = 240 descriptors: 16 KB -- gfx950 / ROCm 7 take kernel-argument
// segments of that size, also as hipGraph kernel nodes; through round 4 the table held 48 = 3 KB), so a launch is self-contained and can be recorded in a
// hipGraph without any host->device copy.  A launch lasts as long as ONE workgroup's tile takes (~42 us for a 32 x 32 x 9 tile with three layouts) whatever
// the number of tiles: the keypoint encoder's ~500 descriptors were 11 launches back to back at the head of every step (0.51 ms), now 3 (0.29 ms).
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

// bf16 planes (modes 8 / 9 / 12 / 13 / 14 / 15) are K16-CHUNK-MAJOR: element (tap tt, row r, column f) of a [taps][rowsP][colsP] matrix sits at
// tt * rowsP * colsP + (f / 16) * (rowsP * 16) + r * 16 + f % 16: the 16-channel slab of a tap that a conv workgroup stages is ONE contiguous run
// of rows x 32 bytes (with row-major planes it was 32 bytes out of every row's Cin * 2: a quarter of each 128-byte line fetched from L2)
__device__ __forceinline__ long long chunk_major(int tt, int r, int f, int rowsP, int colsP) {
    return (long long)tt * rowsP * colsP + (long long)(f >> 4) * (rowsP * 16) + r * 16 + (f & 15);
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
    // Store loops without per-element divisions: a thread keeps its fastest-axis position (lane & 31 -- or a PAIR of positions for the
    // bf16 planes, written as one 32-bit store) and walks the two slower axes incrementally.  (The first version divided three
    // times per element and stored the bf16 pieces 2 bytes per lane: 131 us per 48-convolution launch.)
    for (int k = 0; k < d.ndst; ++k) {
        const int mode = d.mode[k];
        float* __restrict__ dst = d.dst[k];
        if (mode == 12 || mode == 13) {
            // nearest-x2 upsample + 3x3 conv = four 2x2 "phase" convolutions on the low-resolution input (UpBlock2d, util.py:172-176): output
            // pixel (2y + py, 2x + px) reads rows {y - 1 + py, y + py} x columns {x - 1 + px, x + px}; the weight of phase tap (a, b) is the
            // sum of the 3x3 taps that land on that source pixel: rows R(py, a) = {0} {1,2} | {0,1} {2}, columns likewise.  Layout: three
            // bf16 planes [piece][(py*2+px)*4 + a*2+b][CoutPad128][CinPad32] (mode 12: the split of mode 8 applied to the summed weights) or
            // transposed [piece][..][CinPad128][CoutPad32] (mode 13: for the phase data gradient, the split of mode 9's role), each tap's matrix
            // k16-chunk-major (chunk_major()).
            constexpr unsigned RM[4] = {0x1u, 0x6u, 0x3u, 0x4u};         // [py*2 + a] -> bit mask over r
            const bool tr = mode == 13;
            unsigned short* __restrict__ d16 = reinterpret_cast<unsigned short*>(dst);
            const int rowsP = tr ? rup(Cin, 128) : rup(Cout, 128), colsP = tr ? rup(Cout, 32) : rup(Cin, 32);
            const long long piece = 16ll * rowsP * colsP;
            const int nfast = tr ? nco : nci, nslow = tr ? nci : nco;     // fastest destination axis: ci (12) / co (13)
            const int f0 = (threadIdx.x & 15) * 2;                        // pair along the fastest axis
            for (int q = threadIdx.x >> 4; q < nslow * 16; q += 16) {
                const int sl = q >> 4, pt = q & 15;
                if (f0 >= nfast) continue;
                const int ph = pt >> 2, a_ = (pt >> 1) & 1, b_ = pt & 1;
                const unsigned rm = RM[(ph >> 1) * 2 + a_], sm = RM[(ph & 1) * 2 + b_];
                float v[2] = {0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (f0 + e < nfast) {
                        const int co_l = tr ? f0 + e : sl, ci_l = tr ? sl : f0 + e;
                        const float* wl = lds + co_l * row + ci_l * T;
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int s_ = 0; s_ < 3; ++s_)
                                if (((rm >> r) & 1u) && ((sm >> s_) & 1u)) v[e] += wl[r * 3 + s_];
                    }
                }
                const long long idx = tr ? chunk_major(pt, ci0 + sl, co0 + f0, rowsP, colsP) : chunk_major(pt, co0 + sl, ci0 + f0, rowsP, colsP);
                float x0 = v[0], x1 = v[1];
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
                    const unsigned b0 = __float_as_uint(x0) & 0xffff0000u, b1 = __float_as_uint(x1) & 0xffff0000u;
                    *reinterpret_cast<unsigned*>(d16 + pc * piece + idx) = (b0 >> 16) | b1;       // idx is even; a lone last element pairs with 0
                    x0 -= __uint_as_float(b0);
                    x1 -= __uint_as_float(b1);
                }
            }
            continue;
        }
        const bool ci_fast = mode == 0 || mode == 1 || mode == 5 || mode == 8 || mode == 14;
        const bool bf = mode == 8 || mode == 9 || mode == 14 || mode == 15;
        const bool rne1 = mode == 14 || mode == 15;               // ONE plane, round-to-nearest-even (plain bf16 mode), layout of mode 8 / 9
        const int nf = ci_fast ? nci : nco;                       // extent of the fastest (contiguous) destination axis in this tile
        const int lanes_f = bf ? 16 : 32;                         // threads along it (bf16: two elements each)
        const int f0 = (threadIdx.x & (lanes_f - 1)) * (bf ? 2 : 1);
        const int rows_per_pass = 256 / lanes_f;
        const int r0 = threadIdx.x / lanes_f;
        // slower axes as (outer, inner): ci-fast layouts walk (co_l, t); co-fast layouts walk (t, ci_l)
        const int n_inner = ci_fast ? T : nci;
        const int n_rows = ci_fast ? nco * T : T * nci;
        int outer = r0 / n_inner, inner = r0 - outer * n_inner;
        const int base_mode = (mode == 8 || mode == 14) ? 0 : ((mode == 9 || mode == 15) ? 2 : mode);
        unsigned short* __restrict__ d16 = reinterpret_cast<unsigned short*>(dst);
        const long long piece = base_mode == 0 ? (long long)T * rup(Cout, 128) * rup(Cin, 32) : (long long)T * rup(Cin, 128) * rup(Cout, 32);
        for (int q = r0; q < n_rows; q += rows_per_pass) {
            if (f0 < nf) {
                const int co_l = ci_fast ? outer : f0, t = ci_fast ? inner : outer, ci_l = ci_fast ? f0 : inner;
                // neighbour along the fastest axis: +1 in ci (stride T in LDS) or +1 in co (stride `row`)
                const int lstep = ci_fast ? T : row;
                const float v0 = lds[co_l * row + ci_l * T + t];
                const long long idx = !bf ? dst_index(base_mode, co0 + co_l, ci0 + ci_l, t, Cout, Cin, T)
                                      : (base_mode == 0 ? chunk_major(t, co0 + co_l, ci0 + ci_l, rup(Cout, 128), rup(Cin, 32))
                                                        : chunk_major(T - 1 - t, ci0 + ci_l, co0 + co_l, rup(Cin, 128), rup(Cout, 32)));
                if (!bf) {
                    dst[idx] = v0;
                } else {
                    const bool two = f0 + 1 < nf;
                    const float v1 = two ? lds[co_l * row + ci_l * T + t + lstep] : 0.f;
                    if (rne1) {
                        const unsigned ua = __float_as_uint(v0), ub = __float_as_uint(v1);
                        const unsigned ra = (ua + 0x7fffu + ((ua >> 16) & 1u)) >> 16, rb = (ub + 0x7fffu + ((ub >> 16) & 1u)) >> 16;
                        if (two) *reinterpret_cast<unsigned*>(d16 + idx) = ra | (rb << 16);
                        else d16[idx] = (unsigned short)ra;
                        inner += rows_per_pass;
                        while (inner >= n_inner) { inner -= n_inner; ++outer; }
                        continue;
                    }
                    unsigned h[3], l[3];
                    float a = v0, b = v1;
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) {
                        const unsigned ba = __float_as_uint(a) & 0xffff0000u, bb = __float_as_uint(b) & 0xffff0000u;
                        l[pc] = ba >> 16;
                        h[pc] = bb >> 16;
                        a -= __uint_as_float(ba);
                        b -= __uint_as_float(bb);
                    }
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) {
                        if (two) *reinterpret_cast<unsigned*>(d16 + pc * piece + idx) = l[pc] | (h[pc] << 16);    // idx is even
                        else d16[pc * piece + idx] = (unsigned short)l[pc];
                    }
                }
            }
            inner += rows_per_pass;
            while (inner >= n_inner) { inner -= n_inner; ++outer; }
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
    {                                                              // ci fastest on the accumulator side; (co_l, t) walked incrementally
        const int ci_l = threadIdx.x & 31, r0 = threadIdx.x >> 5;
        int co_l = r0 / T, t = r0 - co_l * T;
        for (int q = r0; q < nco * T; q += 8) {
            if (ci_l < nci) {
                const size_t s = d.fewout ? ((size_t)(co0 + co_l) * T + t) * Cin + ci0 + ci_l : ((size_t)t * Cout + co0 + co_l) * Cin + ci0 + ci_l;
                lds[co_l * row + ci_l * T + t] = d.src[s];
            }
            t += 8;
            while (t >= T) { t -= T; ++co_l; }
        }
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
            MRFA_CHECK_ARG(descs[i].dst[k] && (m == 0 || m == 1 || m == 2 || m == 3 || m == 5 || m == 7 || m == 8 || m == 9 || m == 14 || m == 15 || ((m == 12 || m == 13) && descs[i].R == 3 && descs[i].S == 3)),
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
