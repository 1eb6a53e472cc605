// Implicit-GEMM convolution / batched NT GEMM on the CDNA4 fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
//   out[p, co] = alpha * sum_{r,s,ci} in'[pix(p)+(r,s)-pad, ci] * W[co, r, s, ci]   (+bias, affine, ReLU, residual)
//
// GEMM view: M = N*Hout*Wout output pixels, N = Cout, K = R*S*Cin.  NHWC activations make K (channels) the
// contiguous axis of both operands, so both tiles are staged in LDS as [row][k] (k contiguous, row stride 36
// floats = conflict-free for ds_read_b128) and every lane fetches FOUR consecutive k of its MFMA row with one
// ds_read_b128: lane l (row i = l&31, half h = l>>5) reads k = 8q+4h..8q+4h+3 and feeds them to four successive
// 32x32x2 MFMAs; A and B use the same k permutation, so the sum over k is complete and exact fp32 FMA chains.
//
// 256 threads = 4 wave64 per workgroup; block tile BM x BN (128x128 default: each wave owns a 64x64 sub-tile =
// 2x2 MFMA tiles = 64 accumulator VGPRs), BK = 32, register-staged double-buffered LDS, one barrier per k-tile.
// fp32 MFMA issues once per 64 cycles per SIMD and needs only 2 operand floats per lane per issue, so this kernel
// is bounded by the matrix pipe (157.3 TF/s peak), not by LDS or L2: per k-tile a workgroup moves 32 KiB from L2
// and spends 64 MFMAs x 64 cycles = 4096 cycles per wave on it (8 B/cycle/CU).
//
// Reference call sites replaced: every F.conv2d on the hot path (see include/mrfa_hip.h) and the all-pairs
// correlation einsum (modules/raft.py:185).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;   // 36 floats = 144 B row stride: 16-B slots (9*i + h) mod 16 are distinct per lane group

template <int BM, int BN, int WAVES_M, int WAVES_N, bool FLAT>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void conv_mfma_kernel(const mrfa_conv_params p, const int KT, const int kt_per_split,
                                                       const long long M, const int tiles_n, const int total_tiles) {
    constexpr int TM = BM / WAVES_M / 32;
    constexpr int TN = BN / WAVES_N / 32;
    constexpr int NT = WAVES_M * WAVES_N * 64;       // threads per workgroup (256 or 512)
    constexpr int RSTEP = NT / 8;                    // rows covered by one pass of the loader (8 float4 per 32-wide row)
    constexpr int RA = BM / RSTEP;                   // float4 global loads per thread per k-tile (A)
    constexpr int RB = BN / RSTEP;                   // (B)
    static_assert(WAVES_M * WAVES_N == 4 || WAVES_M * WAVES_N == 8, "4 or 8 waves");
    static_assert(RA >= 1 && RB >= 1, "tile too small for the workgroup");
    static_assert(TM >= 1 && TN >= 1, "tile");

    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDK];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave % WAVES_N;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (each XCD has a private 4 MiB L2).  Give every XCD one
    // contiguous band of output tiles so the 3x3 halo rows and the tile's Cout-siblings are re-read from ITS L2
    // instead of being fetched by all eight.
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

    const int lrow = tid >> 3;   // 0..31
    const int kq = tid & 7;      // which float4 of the 32-wide k-tile
    const int Hv = p.Hin << p.ups, Wv = p.Win << p.ups;
    const int HWo = p.Hout * p.Wout;
    const int KC = FLAT ? 1 : (p.Cin >> 5);

    int a_oy[RA], a_ox[RA], a_base[RA];
    bool a_ok[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
        const long long m = m0 + lrow + RSTEP * j;
        a_ok[j] = m < M;
        const long long mm = a_ok[j] ? m : 0;
        const int n_img = (int)(mm / HWo);
        const int rem = (int)(mm - (long long)n_img * HWo);
        a_oy[j] = rem / p.Wout;
        a_ox[j] = rem - a_oy[j] * p.Wout;
        if (FLAT && p.stride == 2) { a_oy[j] *= 2; a_ox[j] *= 2; }       // strided gather (flat-K launches only: HRNet's 3 -> 64 stem, hr_base.py:302)
        a_base[j] = n_img * p.Hin * p.Win;
    }
    bool b_ok[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) b_ok[j] = (n0 + lrow + RSTEP * j) < p.w_rows;

    f32x4 ra[RA], rb[RB];

    // (tap, channel-chunk) of the NEXT k-tile to load, advanced incrementally: no integer divisions in the k-loop
    int l_r = 0, l_s = 0, l_c = 0;           // tap row / col, chunk index
    const float* l_w = w;                    // packed-weight pointer of the current tap
    const int a_koff = kq * 4;

    auto load_tiles = [&](int kt) {
        if constexpr (!FLAT) {
            const int c0 = l_c * 32 + a_koff;
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (p.in_scale) {
                sc = *reinterpret_cast<const f32x4*>(p.in_scale + c0);
                sh = *reinterpret_cast<const f32x4*>(p.in_shift + c0);
            }
            const int dr = l_r - p.pad, ds = l_s - p.pad;
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                const int iy = a_oy[j] + dr;
                const int ix = a_ox[j] + ds;
                const bool inb = a_ok[j] && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
                // branch-free: out-of-bounds rows read a valid address (pixel 0 of their image) and are zeroed afterwards
                const int pix = inb ? (a_base[j] + (iy >> p.ups) * p.Win + (ix >> p.ups)) : a_base[j];
                f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)pix * p.ldx + c0);
                if (p.in_scale) {
                    v = v * sc + sh;
                    if (p.in_relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                }
                ra[j] = inb ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const float* wt = l_w + c0;
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int row = b_ok[j] ? (n0 + lrow + RSTEP * j) : 0;
                const f32x4 v = *reinterpret_cast<const f32x4*>(wt + (size_t)row * p.w_ld);
                rb[j] = b_ok[j] ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // advance (chunk, s, r)
            if (++l_c == KC) {
                l_c = 0;
                l_w += p.w_tap;
                if (++l_s == p.S) { l_s = 0; ++l_r; }
            }
        } else {
            const int k0 = kt * 32 + kq * 4;
            const int4 ent = *reinterpret_cast<const int4*>(p.ktab + k0);
            const int e[4] = {ent.x, ent.y, ent.z, ent.w};
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                float vv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = 0.f;
                    if (e[q] >= 0 && a_ok[j]) {
                        const int dy = (e[q] & 255) - 128, dx = ((e[q] >> 8) & 255) - 128, ci = e[q] >> 16;
                        const int iy = a_oy[j] + dy, ix = a_ox[j] + dx;
                        if ((unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv) {
                            const size_t pix = (size_t)(a_base[j] + (iy >> p.ups) * p.Win + (ix >> p.ups));
                            v = x[pix * p.ldx + ci];
                            if (p.in_scale) {
                                v = v * p.in_scale[ci] + p.in_shift[ci];
                                if (p.in_relu) v = fmaxf(v, 0.f);
                            }
                        }
                    }
                    vv[q] = v;
                }
                ra[j] = f32x4{vv[0], vv[1], vv[2], vv[3]};
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (b_ok[j]) v = *reinterpret_cast<const f32x4*>(w + (size_t)(n0 + lrow + RSTEP * j) * p.w_ld + k0);
                rb[j] = v;
            }
        }
    };

    auto store_tiles = [&](int buf) {
        float* As = smem + buf * (BM + BN) * LDK;
        float* Bs = As + BM * LDK;
#pragma unroll
        for (int j = 0; j < RA; ++j) *reinterpret_cast<f32x4*>(As + (lrow + RSTEP * j) * LDK + kq * 4) = ra[j];
#pragma unroll
        for (int j = 0; j < RB; ++j) *reinterpret_cast<f32x4*>(Bs + (lrow + RSTEP * j) * LDK + kq * 4) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kt_begin = blockIdx.z * kt_per_split;
    const int kt_end = min(KT, kt_begin + kt_per_split);

    if constexpr (!FLAT) {
        const int tap0 = kt_begin / KC;
        l_c = kt_begin - tap0 * KC;
        l_r = tap0 / p.S;
        l_s = tap0 - l_r * p.S;
        l_w = w + (size_t)tap0 * p.w_tap;
    }
    if (kt_begin < kt_end) {
        load_tiles(kt_begin);
        store_tiles(0);
    }
    __syncthreads();

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = (kt + 1) < kt_end;
        if (more) load_tiles(kt + 1);
        const float* As = smem + cur * (BM + BN) * LDK + (wm * TM * 32 + frag_row) * LDK + frag_k;
        const float* Bs = smem + cur * (BM + BN) * LDK + BM * LDK + (wn * TN * 32 + frag_row) * LDK + frag_k;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(As + i * 32 * LDK + q * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDK + q * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        if (more) store_tiles(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ------------------------------------------------------------------ epilogue
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
    if (p.fin_scale) fused_bn_finalize(p, (unsigned)total_tiles);      // (without a split: every workgroup with a tile; with one: the tile's last arriver)
}

// y = bias (or 0) broadcast: initialises the output of a split-K launch
__global__ void splitk_init_kernel(float* y, int ldy, long long rows, int C, const float* bias) {
    const long long total = rows * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        y[(size_t)r * ldy + c] = bias ? bias[c] : 0.f;
    }
}

// second pass of a split-K launch: affine / residual / ReLU / BN statistics on the summed output
__global__ void splitk_epilogue_kernel(float* y, int ldy, long long rows, int C, const float* osc, const float* osh,
                                       const float* res, int ldr, int relu, double* stats) {
    // one thread per channel-column chunk; rows strided over blockIdx.y so statistics reduce per thread first
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = osc ? osc[c] : 1.f, sh = osh ? osh[c] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    for (long long r = blockIdx.y; r < rows; r += gridDim.y) {
        float v = y[(size_t)r * ldy + c] * sc + sh;
        if (res) v += res[(size_t)r * ldr + c];
        if (relu) v = fmaxf(v, 0.f);
        y[(size_t)r * ldy + c] = v;
        s1 += v;
        s2 += v * v;
    }
    if (stats) {
        double* st = stats + (size_t)(blockIdx.y % MRFA_STATS_SLOTS) * 2 * C;
        atomicAdd(st + c, (double)s1);
        atomicAdd(st + C + c, (double)s2);
    }
}

template <int BM, int BN, int WM_, int WN_>
int launch_cfg(hipStream_t st, const mrfa_conv_params& p, int KT, long long M, int splitk) {
    const int tiles_n = cdiv(p.Cout, BN);
    const long long tiles_m = (M + BM - 1) / BM;
    const int total_tiles = (int)(tiles_m * tiles_n);
    dim3 grid((unsigned)(cdiv(total_tiles, 8) * 8), (unsigned)(p.nbatch > 1 ? p.nbatch : 1), (unsigned)splitk);
    const int kps = cdiv(KT, splitk);
    mrfa_conv_params q = p;
    q.splitk = splitk;
    if (p.kflat > 0)
        hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM_, WN_, true>), grid, dim3(WM_ * WN_ * 64), 0, st, q, KT, kps, M, tiles_n, total_tiles);
    else
        hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM_, WN_, false>), grid, dim3(WM_ * WN_ * 64), 0, st, q, KT, kps, M, tiles_n, total_tiles);
    MRFA_CHECK_LAUNCH("mrfa_conv2d_nhwc");
    return 0;
}

}  // namespace

// 0: v_mfma_f32_32x32x2_f32 everywhere; 1: eligible launches (chunked K, 128-row tile) run the bf16x6 split-operand kernel;
// 2: the same kernels keeping the three leading products only (bf16x3: 2^-16-class product error instead of 2^-24);
// 3: plain bf16 operands (round-to-nearest-even), one product: the arithmetic of a bf16 autocast (BASELINE config 4)
static int g_mfma_mode = 0;
static int g_split_target_256 = getenv("MRFA_SPLIT_TARGET_256") ? atoi(getenv("MRFA_SPLIT_TARGET_256")) : 1;
extern "C" int mrfa_set_mfma_mode(int mode) {
    if (mode < 0 || mode > 3) { mrfa_set_error("set_mfma_mode: unknown mode %d", mode); return 1; }
    g_mfma_mode = mode;
    return 0;
}
extern "C" int mrfa_get_mfma_mode(void) { return g_mfma_mode; }

static thread_local int g_last_tile = 0;
// (BM << 16) | (BN << 4) | (flat << 1) | (splitk > 1) of the most recent mrfa_conv2d_nhwc launch on this thread
extern "C" int mrfa_conv2d_last_config(void) { return g_last_tile; }

extern "C" int mrfa_conv2d_stride_supported(const mrfa_conv_params* p) {
    if (!p || p->stride != 2) return 0;
    if (p->Hout != (p->Hin + 2 * p->pad - p->R) / 2 + 1 || p->Wout != (p->Win + 2 * p->pad - p->S) / 2 + 1) return 0;
    if (p->kflat > 0) return (!p->ups && p->nbatch <= 1 && p->splitk <= 1 && p->ktab) ? 1 : 0;      // flat-K gather of the fp32 tile kernel
    if (!mrfa_tuning_conv_small()) return 0;
    static const bool small_on = [] { const char* e = getenv("MRFA_CONV_SMALL"); return !(e && e[0] == '0'); }();
    return small_on && mrfa_conv_small_eligible(*p, (long long)p->N * p->Hout * p->Wout) ? 1 : 0;
}

static int conv2d_dispatch(void* stream, const mrfa_conv_params* pp, bool* fin_done, int* dry_split = nullptr, int* dry_reads_w = nullptr);

extern "C" int mrfa_conv2d_bwdstats_supported(const mrfa_conv_params* p) {
    if (!p || !p->stats || p->fin_scale || p->stride < 0 || p->kflat > 0) return 0;
    if (mrfa_conv_lean_eligible(*p) || mrfa_gemm_lean_eligible(*p, (long long)p->N * p->Hout * p->Wout)) return 1;
    static const bool small_on = [] { const char* e = getenv("MRFA_CONV_SMALL"); return !(e && e[0] == '0'); }();
    return small_on && mrfa_tuning_conv_small() && mrfa_conv_small_eligible(*p, (long long)p->N * p->Hout * p->Wout) ? 1 : 0;
}

// statistic groups: every kernel's output tile is at most 128 rows (and divides 128), the patch-tiled kernel's lies inside one image; a K split sums
// its partial tiles in a pass whose workgroups stride over ALL rows, so grouped launches never split K (the automatic choice is switched off for them)
extern "C" int mrfa_conv2d_groups_supported(const mrfa_conv_params* p) {
    if (!p || p->groups <= 1) return p ? 1 : 0;
    if (!p->stats && !p->fin_scale && !p->bst_x && !p->in_scale) return 1;       // (nothing per group in this call)
    if (p->nbatch > 1 || p->splitk > 1) return 0;
    const long long M = (long long)p->N * p->Hout * p->Wout;
    const long long rows = group_rows(*p, M);
    if (rows <= 0) return 0;
    if (p->kflat == 0 && mrfa_conv_lean_eligible(*p)) return 1;                                      // (a patch lies inside one image)
    if (p->kflat == 0 && mrfa_gemm_lean_eligible(*p, M)) return 1;                                   // (its own rule: rows of a group % 64 == 0)
    if (p->in_scale) return 0;                                                                       // (prologue vectors per group: conv_lean.hip only)
    if ((rows % 128) == 0) return 1;
    // shorter groups: the kernels whose tiles are smaller than 128 rows, where the dispatch would pick them
    static const bool small_on = [] { const char* e = getenv("MRFA_CONV_SMALL"); return !(e && e[0] == '0'); }();
    if (small_on && mrfa_tuning_conv_small() && mrfa_conv_small_eligible(*p, M)) return 1;       // (its own rule: rows % 64 == 0)
    return p->kflat == 0 && p->stride <= 1 && mrfa_conv_halo_eligible(*p) ? 1 : 0;                // (a patch lies inside one image)
}

// v8: the K slices conv2d_dispatch would use for these parameters (the dispatch itself, stopped before its first launch)
extern "C" int mrfa_conv2d_split_k(const mrfa_conv_params* p) {
    if (!p) return 1;
    if (p->splitk > 1) return p->splitk;
    int k = 1;
    bool fin_done = false;
    return conv2d_dispatch(nullptr, p, &fin_done, &k) ? 1 : k;
}

// v9: does the kernel a call with these parameters would run read the fp32 weight layout `w` at all?  0: it reads the pre-split planes only (conv_halo.hip,
// conv_lean.hip, the row-tiled split-operand tile with w_split in the split modes) -- the caller may pass any non-NULL `w` and need not keep (or refresh, once
// per optimizer step) that layout: for the decoder's ~100 M parameters that is 8 of the 28 bytes per parameter the per-step re-packing moved.
extern "C" int mrfa_conv2d_reads_fp32_weights(const mrfa_conv_params* p) {
    if (!p || (!p->w_split && !p->w_phase)) return 1;
    int k = 1, reads = 1;
    bool fin_done = false;
    return conv2d_dispatch(nullptr, p, &fin_done, &k, &reads) ? 1 : reads;
}

extern "C" int mrfa_conv2d_nhwc(void* stream, const mrfa_conv_params* pp) {
    MRFA_CHECK_ARG(pp, "conv2d: null parameter block");
    if (pp->groups > 1 && (pp->stats || pp->fin_scale || pp->bst_x || pp->in_scale))
        MRFA_CHECK_ARG(mrfa_conv2d_groups_supported(pp), "conv2d: groups = %d is not implemented for these parameters (N = %d, %d x %d outputs, splitk %d): ask "
                       "mrfa_conv2d_groups_supported() first", pp->groups, pp->N, pp->Hout, pp->Wout, pp->splitk);
    if (pp->fin_scale) {
        MRFA_CHECK_ARG(pp->stats && pp->fin_shift && pp->fin_gamma && pp->fin_beta && pp->fin_counter && pp->fin_count > 0,
                       "conv2d: fin_scale needs stats, fin_shift, fin_gamma, fin_beta, fin_counter and fin_count > 0");
        MRFA_CHECK_ARG((pp->fin_rmean == nullptr) == (pp->fin_rvar == nullptr), "conv2d: fin_rmean / fin_rvar come together");
    }
    if (pp->bst_x) {
        MRFA_CHECK_ARG(pp->stats && pp->bst_scale && pp->bst_shift && pp->bst_mean && pp->bst_invstd && !pp->fin_scale,
                       "conv2d: bst_x needs stats, bst_scale / _shift / _mean / _invstd and no fin_*");
        MRFA_CHECK_ARG(mrfa_conv2d_bwdstats_supported(pp), "conv2d: bst_* is only implemented by the small-problem kernels: ask mrfa_conv2d_bwdstats_supported() first");
    }
    bool fin_done = false;
    const int rc = conv2d_dispatch(stream, pp, &fin_done);
    if (rc || !pp->fin_scale || fin_done) return rc;
    // every kernel but the one-wave-per-tile one: the finalize launch behind the convolution, inside the call
    return mrfa_bn_finalize_groups(stream, pp->stats, pp->fin_count, pp->fin_gamma, pp->fin_beta, pp->fin_rmean, pp->fin_rvar, pp->fin_momentum, pp->fin_eps,
                                   pp->Cout, pp->groups > 1 ? pp->groups : 1, pp->fin_scale, pp->fin_shift, pp->fin_mean, pp->fin_invstd);
}

static int conv2d_dispatch(void* stream, const mrfa_conv_params* pp, bool* fin_done, int* dry_split, int* dry_reads_w) {
    const mrfa_conv_params& p = *pp;
    hipStream_t st = (hipStream_t)stream;
    MRFA_CHECK_ARG(p.x && p.w && (p.y || dry_split), "conv2d: null pointer");
    MRFA_CHECK_ARG(p.N > 0 && p.Cin > 0 && p.Cout > 0 && p.Hout > 0 && p.Wout > 0, "conv2d: bad sizes");
    MRFA_CHECK_ARG(p.R >= 1 && p.S >= 1 && p.R <= 15 && p.S <= 15, "conv2d: kernel size %dx%d unsupported", p.R, p.S);
    MRFA_CHECK_ARG((p.w_ld % 4) == 0 && aligned16(p.w), "conv2d: packed weight must be 16-B aligned, w_ld %% 4 == 0");
    const bool flat = p.kflat > 0;
    if (!flat) {
        MRFA_CHECK_ARG((p.Cin % 32) == 0, "conv2d: chunked mode needs Cin %% 32 == 0 (got %d); use flat mode", p.Cin);
        MRFA_CHECK_ARG((p.ldx % 4) == 0 && aligned16(p.x), "conv2d: chunked mode needs 16-B aligned x and ldx %% 4 == 0");
        if (p.in_scale) MRFA_CHECK_ARG(aligned16(p.in_scale) && aligned16(p.in_shift), "conv2d: in_scale/in_shift alignment");
    } else {
        MRFA_CHECK_ARG(p.ktab != nullptr && aligned16(p.ktab), "conv2d: flat mode needs a 16-B aligned ktab");
    }
    const long long M = (long long)p.N * p.Hout * p.Wout;
    const int Ktot = flat ? p.kflat : p.R * p.S * p.Cin;
    const int KT = (Ktot + BK - 1) / BK;
    const int nb = p.nbatch > 1 ? p.nbatch : 1;

    // ---- the keypoint encoder's <= 128-channel 3x3 layers in a split-operand mode: four-wave patches on the bf16 pipe (conv_lean.hip)
    if (!flat && mrfa_conv_lean_eligible(p)) {
        if (dry_split) { *dry_split = 1; if (dry_reads_w) *dry_reads_w = 0; return 0; }
        g_last_tile = (32 << 16) | (32 << 4) | 4 | (1 << 27);    // bit 27: conv_lean
        *fin_done = p.fin_scale != nullptr;                      // (finished by the launch's last workgroup)
        return mrfa_conv_lean_launch(st, p);
    }
    // ---- 1x1 convolutions / linears of the keypoint encoder in a split-operand mode: K-pipelined four-wave tiles on the bf16 pipe (conv_lean.hip)
    if (!flat && mrfa_gemm_lean_eligible(p, M)) {
        if (dry_split) { *dry_split = 1; if (dry_reads_w) *dry_reads_w = 0; return 0; }
        g_last_tile = (64 << 16) | (64 << 4) | 4 | (1 << 26);    // bit 26: gemm_lean
        *fin_done = p.fin_scale != nullptr;                      // (finished by the launch's last workgroup)
        return mrfa_gemm_lean_launch(st, p, M);
    }
    // ---- small problems (the MTIA prior's 0.1-0.6 GFLOP layers): one wave per output tile, no LDS / barrier / split-K (conv_small.hip)
    static const bool small_on = [] { const char* e = getenv("MRFA_CONV_SMALL"); return !(e && e[0] == '0'); }();
    if (small_on && mrfa_tuning_conv_small() && mrfa_conv_small_eligible(p, M)) {
        if (dry_split) { *dry_split = 1; return 0; }
        g_last_tile = (16 << 16) | (16 << 4) | 8;                // bit 3: conv_small
        *fin_done = p.fin_scale != nullptr;                      // (finished by the launch's last workgroup)
        return mrfa_conv_small_launch(st, p, M);
    }
    if (p.stride < 0 || p.stride > 2 || (p.stride == 2 && !(flat && !p.ups && p.nbatch <= 1))) {
        mrfa_set_error("conv2d: stride = %d is only implemented by the one-wave-per-tile kernel and by flat-K launches: ask mrfa_conv2d_stride_supported() first", p.stride);
        return 1;
    }
    if (p.mask && !(!flat && mrfa_conv_halo_eligible(p))) {
        mrfa_set_error("conv2d: `mask` is only honoured by the patch-tiled kernel: ask mrfa_conv2d_mask_supported() first");
        return 1;
    }
    if (p.ups == 2 && !(!flat && mrfa_conv_halo_eligible(p))) {
        mrfa_set_error("conv2d: ups = 2 (phase data gradient of a fused-upsample layer) is only implemented for the shapes "
                       "mrfa_conv2d_phase_dgrad_supported() reports");
        return 1;
    }
    // ---- 3x3 stride-1 layers with 32-aligned rows in split-operand mode: patch-tiled kernel, input halo split once per chunk (conv_halo.hip)
    if (!flat && mrfa_conv_halo_eligible(p)) {
        if (dry_split) { *dry_split = 1; if (dry_reads_w) *dry_reads_w = 0; return 0; }
        *fin_done = p.fin_scale != nullptr;                      // (finished by the launch's last workgroup, common.h: fused_bn_finalize)
        g_last_tile = (128 << 16) | ((p.Cout <= 64 ? 64 : 128) << 4) | 4 | (1 << 28);       // bit 28: conv_halo
        return mrfa_conv_halo_launch(st, p);
    }
    // ---- tile selection: largest BN whose padding waste is small, then BM by how many workgroups result
    int BN = 128;
    {
        const int cands[4] = {128, 96, 64, 32};
        const double pen[4] = {1.0, 1.08, 1.16, 1.3};       // measured per-tile efficiency relative to 128x128
        double best = 1e18;
        for (int i = 0; i < 4; ++i) {
            if (cands[i] == 96 && flat) continue;
            const double cost = (double)cdiv(p.Cout, cands[i]) * cands[i] * pen[i];
            if (cost < best) { best = cost; BN = cands[i]; }
        }
    }
    // split-operand mode: the bf16x6 kernel only exists as a 128-wide tile and is ~1.5x faster than the fp32-MFMA tiles, which
    // outweighs the padding of 64 / 96 / 160 / 192-channel outputs to a multiple of 128
    if (g_mfma_mode >= 1 && !flat && p.Cout >= 32) BN = p.Cout <= 64 ? 64 : 128;
    auto ntiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * cdiv(p.Cout, bn) * nb; };
    // Few output tiles (low-resolution hourglass / generator levels): every M-tile re-reads the whole weight tensor, so
    // keep the tile tall and split K across workgroups first; shrink BM only when K is too short to split.
    int BM = 128;
    if (BN == 96 && ntiles(128, 96) < 384) BN = 128;     // few tiles: the 128-wide family has the 64/32-row variants
    if (BN == 128) { if (M <= 32) BM = 32; else if (M <= 64) BM = 64; }
    else if (BN == 64 && M <= 64) BM = 64;
    int splitk = 1;
    const bool auto_split = (p.splitk == 0);
    if (p.splitk > 1) splitk = p.splitk;
    {
        // Short K loop over many pixels (HRNet's 32..64-channel 3x3 convs at 64^2 / 32^2): a K split would add a zero-init and
        // a reduction/epilogue pass over the whole output (2 x 20 us measured) to a 20 us kernel -- keep one launch
        const bool short_k_big_m = KT < 32 && M > 4096;
        const bool grouped_stats = p.groups > 1 && p.stats && !(p.sk_ticket && !p.accumulate);       // (the split-K epilogue PASS does not keep statistic groups apart; the fused one does)
        const int max_split = (auto_split && !short_k_big_m && !grouped_stats) ? (KT / 2 > 0 ? KT / 2 : 1) : 1;
        long long t = ntiles(BM, BN);
        // (round 4, tools/sweep_splitk.py: with the split forced per launch the automatic choice is within 10 % of the best on six of eight low-resolution
        // shapes -- 128 -> 32 @64^2 fused upsample would prefer no split (74 -> 54 us), 256 -> 512 @8^2 a quarter of the slices (40 -> 30 us).  Lowering the
        // target here to catch those two moved OTHER layers onto the 64-row fp32 tiles through the BM loop below: +0.8 ms per step.  Left as it was.)
        if (t < 384 && !(short_k_big_m && t >= 128)) {
            // 512 workgroups -- or 256 for the deep-K launches that finish their split themselves (sk_ticket): their partial sums pass the memory-side atomic
            // units in one burst behind the k-loop (12 + 6 us of a 51 us launch, profiles/r6_splitk_launch_timeline.txt), and with the 128-row tile kept half the
            // slices are 5-6 us faster on 8 <= t <= 16 tiles (tools/sweep_splitk.py TILE128=1 COLD=1 FUSED=1: 1024 -> 1024 @4^2 52.3 -> 47.1, 512 -> 512 @8^2
            // 51.4 -> 46.7, 1024 -> 256 @8^2 up 53.0 -> 46.5, 256 -> 512 @8^2 42.8 -> 36.0)
            const bool half_target = g_split_target_256 && auto_split && p.sk_ticket && !p.accumulate && KT >= 72 && KT <= 320 && BM == 128 && t >= 8 && t <= 16;      // (deeper K: 2048 -> 512 @8^2 up wants its 32 slices, 86 against 101 us)
            int want = (int)(((half_target ? 256 : 512) + t - 1) / t);
            if (auto_split) splitk = want <= max_split ? want : max_split;
            const int min_bm = (BN == 128) ? 32 : (BN == 64 ? 64 : 128);
            while (!half_target && ntiles(BM, BN) * splitk < 384 && BM > min_bm) {
                BM >>= 1;
                if (auto_split) {
                    t = ntiles(BM, BN);
                    want = (int)((512 + t - 1) / t);
                    splitk = want <= max_split ? want : max_split;
                }
            }
            if (splitk < 1) splitk = 1;
        }
    }
    bool w8 = false;                                         // 8-wave (512-thread) variant of the 128x128 tile
    if (p.tile) {
        BM = p.tile >> 16; BN = p.tile & 0x7fff; w8 = (p.tile & 0x8000) != 0;
        if (p.splitk >= 1) splitk = p.splitk;
    }
    if (dry_split) {
        *dry_split = splitk;
        // (the row-tiled split-operand tile copies pre-split weight planes when it has them -- except in plain-bf16 mode, which rounds the fp32 layout itself)
        if (dry_reads_w) *dry_reads_w = !(g_mfma_mode >= 1 && g_mfma_mode != 3 && BM == 128 && (BN == 128 || BN == 64) && !flat && p.w_split);
        return 0;
    }
    // v8: with sk_ticket the tile's last workgroup applies bias / affine / residual / ReLU / statistics (no epilogue pass); with y_zero the output already holds
    // zeros (no init pass: the bias then comes with the fused epilogue)
    const bool fused = splitk > 1 && p.sk_ticket && !p.accumulate;
    mrfa_conv_params pk = p;                                 // what the kernels see
    if (splitk > 1) {
        const bool init = !p.accumulate && !(p.y_zero && (fused || !p.bias));
        if (init) {
            const long long rows = M * nb;
            MRFA_CHECK_ARG(nb == 1 || p.y_bs == (long long)M * p.ldy, "conv2d: split-K batched output must be dense");
            hipLaunchKernelGGL(splitk_init_kernel, dim3(stream_grid(rows * p.Cout, 256)), dim3(256), 0, st, p.y, p.ldy, rows,
                               p.Cout, p.bias);
            MRFA_CHECK_LAUNCH("splitk_init");
            pk.bias = nullptr;                               // (written by the init pass)
        }
        if (!fused) pk.sk_ticket = nullptr;
    }
    // the BatchNorm that follows (fin_*): finished by the launch's last workgroup, except behind a K split with an epilogue pass and in batched launches
    if (p.fin_scale && nb == 1 && (splitk == 1 || fused)) *fin_done = true;
    else pk.fin_scale = nullptr;
    g_last_tile = (BM << 16) | (BN << 4) | ((flat ? 1 : 0) << 1) | (splitk > 1 ? 1 : 0);
    int rc = 1;
#define CFG(bm, bn, wm, wn) if (BM == bm && BN == bn) rc = launch_cfg<bm, bn, wm, wn>(st, pk, KT, M, splitk)
    if (!p.tile && BM == 128 && BN == 128 && !flat) w8 = true;      // 8 waves: 4 waves/SIMD hide the load/barrier phases (+4..13 %)
    if (g_mfma_mode >= 1 && BM == 128 && (BN == 128 || BN == 64) && !flat) {
        g_last_tile |= 4;                                            // bit 2: split-operand kernel
        rc = mrfa_conv_split_launch(st, pk, KT, M, splitk, BN);
    } else if (w8 && BM == 128 && BN == 128) rc = launch_cfg<128, 128, 2, 4>(st, pk, KT, M, splitk);
    else CFG(128, 128, 2, 2);
    else CFG(128, 96, 4, 1);
    else CFG(128, 64, 2, 2);
    else CFG(128, 32, 4, 1);
    else CFG(64, 128, 2, 2);
    else CFG(64, 64, 2, 2);
    else CFG(32, 128, 1, 4);
    else { mrfa_set_error("conv2d: no tile config %dx%d", BM, BN); return 1; }
#undef CFG
    if (rc) return rc;
    if (splitk > 1 && !fused && (p.relu || p.stats || p.out_scale || p.res)) {
        MRFA_CHECK_ARG(!p.accumulate, "conv2d: split-K with accumulate cannot apply an epilogue");
        const long long rows = M * nb;
        dim3 grid(cdiv(p.Cout, 64), (unsigned)(rows < 256 ? rows : 256));
        hipLaunchKernelGGL(splitk_epilogue_kernel, grid, dim3(64), 0, st, p.y, p.ldy, rows, p.Cout, p.out_scale, p.out_shift,
                           p.res, p.ldr, p.relu, p.stats);
        MRFA_CHECK_LAUNCH("splitk_epilogue");
    }
    return 0;
}
