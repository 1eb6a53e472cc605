// 3x3 / pad 1 / stride 1 convolutions of the MTIA prior's HRNet trunk (32->32 @64^2, 64->64 @32^2, 128->128 @16^2, 64->64 @64^2, the
// 3x3 transition / fusion layers at those sizes; hr_base.py:26-54,120-289): exact fp32 on v_mfma_f32_16x16x4_f32 with the INPUT HALO
// STAGED IN LDS ONCE per workgroup.
//
// Why (round 4): conv_small.hip runs these 0.6 GFLOP layers at 22-32 TF/s (14-17 us).  It feeds the MFMAs straight from L1: every one
// of the nine taps re-reads its activation rows through the vector-memory path, 1 KB per wave and k16 chunk for 4-16 MFMAs, which is
// 64-96 B/clk per CU against an L1 that delivers 64 -- the kernel sits on the load path, not on the matrix pipe ("per-CU
// vector load path").  Here the (rows + 2) x (width + 2) x Cin halo of a workgroup's output rows goes global -> LDS once (float4 loads
// along the channels, the fused pre-activation BatchNorm + ReLU of the input applied on the way: once per element instead of once per
// tap), and all nine taps read it with ds_read_b128: a tap is an address offset.  Only the weights still come through L1 (every wave
// of every workgroup reads the same <= 590 KB: L1 / L2 hits, a four-deep register ring).
//
// Layout: LDS pixel stride = Cin + 4 floats, so the 8 lanes of a ds_read_b128 issue group (8 consecutive pixels, one 16-byte
// channel quad each) start 4 words apart modulo the 32 banks: conflict free, and the staging ds_write_b128 (consecutive channel quads
// of one pixel) are conflict free as well.  A lane (i = lane & 15, kq = lane >> 4) reads the float4 = channels c0 + 4 kq .. + 3 of
// pixel i for a 16-channel chunk and feeds component j to MFMA j; the weight fragment uses the same k permutation (as conv_small.hip).
// The product is issued transposed (D = W X^T: rows = output channels, columns = pixels), which leaves a lane with 4 consecutive
// output channels of one pixel per accumulator: 16-byte epilogue stores.
//
// Wave tile = 16 TM pixels of one image row x 16 TN output channels; workgroup = WM x WN waves = whole rows (or a row segment) x
// 16 TN WN channels.  Same arguments, packed-weight layout (pack mode 0 / 2) and epilogue semantics as conv_small.hip (bias, output
// affine, residual, ReLU, accumulate, slotted BatchNorm statistics) plus the pre-activation prologue (in_scale / in_shift / in_relu),
// so that the data gradient (flipped / transposed pack) runs here too.  mrfa_conv2d_nhwc dispatches here before conv_small.
#include <stdlib.h>
#include "common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int RING = 8;              // weight fragments in flight (k16 steps)

struct LdsGeo {
    int PR;          // output rows per workgroup
    int PWW;         // output pixels per row per workgroup (the whole row, or a segment of WM px-tiles)
    int HPW;         // halo row pitch in pixels = PWW + 2
    int CC;          // channels staged per pass (Cin, or a divisor of it when the halo would not fit)
    int STR;         // LDS pixel stride in floats = CC + 4
    int segs;        // row segments per image row (1: whole rows)
    int bands;       // row bands per image
    int abl;         // (timing ablations, MRFA_LDS_ABL: bit 0 no staging, bit 1 no main loop, bit 2 no epilogue)
};

// KC = k16 steps per tap and staging pass (CC / 16), compile time: the 9 KC steps of a pass are straight-line code with static ring slots --
// with ONE wave per SIMD nobody else fills issue slots, so the loads / LDS reads of the next steps must sit BETWEEN the MFMAs of this one, which
// the compiler only does inside one basic block (the first version walked the taps with a run-time state machine: ~40 scalar / address
// instructions per step issued while the matrix pipe idled, 58 % pipe utilisation in the loop)
template <int TM, int TN, int WM, int WN, bool PRO, int KC>
__global__ __launch_bounds__(256) void conv_lds_kernel(const mrfa_conv_params p, const LdsGeo g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float sred[4][2][TN * 16];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int wm = wave % WM, wn = wave / WM;
    // blockIdx.x = ((image, band), segment); blockIdx.y = output-channel group
    const int seg = blockIdx.x % g.segs;
    const int bb = blockIdx.x / g.segs;
    const int band = bb % g.bands, n_img = bb / g.bands;
    const int y0 = band * g.PR, x0 = seg * g.PWW;
    const int n0 = blockIdx.y * (WN * TN * 16);
    constexpr int PXT = 16 * TM;
    const int tpr = g.PWW / PXT;                       // px-tiles per staged row
    const int py = wm / tpr, pxo = (wm % tpr) * PXT;   // this wave's tile: row py of the band, pixels pxo .. pxo + PXT - 1 of the segment
    const bool row_ok = y0 + py < p.Hout;

    // weight fragment rows (rows past w_rows are clamped: their products land in channels >= Cout, never stored)
    const float* wrow[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        int n = n0 + (wn * TN + b) * 16 + li;
        if (n >= p.w_rows) n = p.w_rows - 1;
        wrow[b] = p.w + (size_t)n * p.w_ld + 4 * kq;
    }
    f32x4v acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};

    const int HP = (g.PR + 2) * g.HPW;                 // halo pixels
    const int q_per_px = g.CC >> 2;                    // float4 units per pixel
    const int units = HP * q_per_px;
    constexpr int NQ = 9 * KC;                         // k16 steps per pass
    const f32x4v* const sm4 = reinterpret_cast<const f32x4v*>(smem);   // LDS in float4 units: STR, 4 kq, 16 kc are all multiples of 4 floats
    const int STR4 = g.STR >> 2;
    const int a_base = (py * g.HPW + pxo + li) * STR4 + kq;      // + (r * HPW + s + 16 a) * STR4 + 4 kc   (float4 units)

    for (int c_base = 0; c_base < p.Cin; c_base += g.CC) {
        if (c_base) __syncthreads();                   // the previous pass's readers are done
        // ---- stage the halo: 4 units in flight per thread
        for (int u0 = tid; u0 < ((g.abl & 1) ? 0 : units); u0 += 4 * 256) {
            f32x4v v[4];
            int dst[4];
            bool inb[4], val[4];
            int qd[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int u = u0 + j * 256;
                val[j] = u < units;
                const int uu = val[j] ? u : 0;
                const int hp = uu / q_per_px;
                qd[j] = uu - hp * q_per_px;
                const int hy = hp / g.HPW, hx = hp - hy * g.HPW;
                const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
                inb[j] = val[j] && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                const long long pix = (long long)n_img * p.Hin * p.Win + (inb[j] ? iy * p.Win + ix : 0);      // branch-free: a valid address
                v[j] = *reinterpret_cast<const f32x4v*>(p.x + (size_t)pix * p.ldx + c_base + qd[j] * 4);
                dst[j] = hp * (g.STR >> 2) + qd[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4v t = v[j];
                if constexpr (PRO) {                   // pre-activation BatchNorm + ReLU of the input (zero padding applies AFTER it)
                    const f32x4v sc = *reinterpret_cast<const f32x4v*>(p.in_scale + c_base + qd[j] * 4);
                    const f32x4v sh = *reinterpret_cast<const f32x4v*>(p.in_shift + c_base + qd[j] * 4);
                    t = t * sc + sh;
                    if (p.in_relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = fmaxf(t[e], 0.f);
                    }
                }
                t = inb[j] ? t : f32x4v{0.f, 0.f, 0.f, 0.f};
                if (val[j]) reinterpret_cast<f32x4v*>(smem)[dst[j]] = t;
            }
        }
        // ---- weights of this pass: ring of RING k16 steps in registers, the first ones loaded while the halo lands
        f32x4v rb[RING][TN];
        auto load_b = [&](int q, int slot) {           // q, slot: compile-time after unrolling
            const int tap = q / KC, kc = q % KC;
#pragma unroll
            for (int b = 0; b < TN; ++b) rb[slot][b] = *reinterpret_cast<const f32x4v*>(wrow[b] + (size_t)c_base + (size_t)tap * p.w_tap + 16 * kc);
        };
        auto load_a = [&](int q, int buf, f32x4v (&xa)[2][TM]) {
            const int tap = q / KC, kc = q % KC;
            const int r = tap / 3, sx = tap % 3;
            const int off = a_base + (r * g.HPW + sx) * STR4 + 4 * kc;
#pragma unroll
            for (int a = 0; a < TM; ++a) xa[buf][a] = sm4[off + 16 * a * STR4];
        };
#pragma unroll
        for (int q = 0; q < RING - 1 && q < NQ; ++q) load_b(q, q % RING);
        __syncthreads();
        if (!(g.abl & 2)) {
            f32x4v xa[2][TM];
            load_a(0, 0, xa);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q + RING - 1 < NQ) load_b(q + RING - 1, (q + RING - 1) % RING);
                if (q + 1 < NQ) load_a(q + 1, (q + 1) & 1, xa);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb[q % RING][b][j], xa[q & 1][a][j], acc[a][b], 0, 0, 0);
            }
        }
    }

    if (g.abl & 4) return;
    // ---- epilogue.  D = W X^T: lane (li, kq) holds, per tile (a, b), output channels cb + 4 kq + {0..3} of pixel pxo + 16 a + li
    float s1[TN][4], s2[TN][4];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int c0 = n0 + (wn * TN + b) * 16 + 4 * kq;
        float bias[4], osc[4], osh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool c_ok = c0 + e < p.Cout;
            bias[e] = (p.bias && c_ok) ? p.bias[c0 + e] : 0.f;
            osc[e] = (p.out_scale && c_ok) ? p.out_scale[c0 + e] : 1.f;
            osh[e] = (p.out_scale && c_ok) ? p.out_shift[c0 + e] : 0.f;
            s1[b][e] = s2[b][e] = 0.f;
        }
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int ox = x0 + pxo + 16 * a + li;
            if (!row_ok || ox >= p.Wout) continue;
            const long long m = ((long long)n_img * p.Hout + y0 + py) * p.Wout + ox;
            float* dst = p.y + (size_t)m * p.ldy + c0;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (acc[a][b][e] * p.alpha + bias[e]) * osc[e] + osh[e];
            if (c0 + 3 < p.Cout) {
                if (p.res) {
                    const f32x4v r4 = *reinterpret_cast<const f32x4v*>(p.res + (size_t)m * p.ldr + c0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += r4[e];
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (p.accumulate) {
                    const f32x4v o4 = *reinterpret_cast<const f32x4v*>(dst);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += o4[e];
                }
                *reinterpret_cast<f32x4v*>(dst) = f32x4v{v[0], v[1], v[2], v[3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1[b][e] += v[e]; s2[b][e] += v[e] * v[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (c0 + e < p.Cout) {
                        float u = v[e];
                        if (p.res) u += p.res[(size_t)m * p.ldr + c0 + e];
                        if (p.relu) u = fmaxf(u, 0.f);
                        if (p.accumulate) u += dst[e];
                        dst[e] = u;
                        s1[b][e] += u;
                        s2[b][e] += u * u;
                    }
                }
            }
        }
    }
    if (p.stats) {
        // per-channel sums over the 16 pixel lanes of a k-quad group, then over the WM waves of this channel group through LDS: one fp64
        // atomic per channel, sum and workgroup (slotted: MRFA_STATS_SLOTS)
        // butterfly reduce-scatter over the 16 pixel lanes: TN * 4 values per statistic, (TN * 4) / 2 + ... shuffles instead of 4 per value
        constexpr int NV = TN * 4;
        float v1[NV], v2[NV];
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) { v1[b * 4 + e] = s1[b][e]; v2[b * 4 + e] = s2[b][e]; }
        int have = NV, idx = 0;                         // this lane keeps values [idx, idx + have) of the NV
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) {
            if (have > 1) {                            // split: the lane with bit o clear keeps the low half
                const bool hi = (li & o) != 0;
                const int half = have / 2;
#pragma unroll
                for (int k = 0; k < NV / 2; ++k) {
                    if (k < half) {
                        const float send1 = hi ? v1[k] : v1[k + half], keep1 = hi ? v1[k + half] : v1[k];
                        const float send2 = hi ? v2[k] : v2[k + half], keep2 = hi ? v2[k + half] : v2[k];
                        v1[k] = keep1 + __shfl_xor(send1, o, 64);
                        v2[k] = keep2 + __shfl_xor(send2, o, 64);
                    }
                }
                idx = idx * 2 + (hi ? 1 : 0);
                have = half;
            } else {                                   // one value left: plain butterfly
                v1[0] += __shfl_xor(v1[0], o, 64);
                v2[0] += __shfl_xor(v2[0], o, 64);
            }
        }
        // NV = 8: after o = 8, 4, 2 every lane holds ONE channel, index = bits (8, 4, 2) of li; NV = 4: after o = 8, 4, bits (8, 4)
        {
            int ch;
            if (NV == 8) ch = ((li >> 3) & 1) * 4 + ((li >> 2) & 1) * 2 + ((li >> 1) & 1);
            else ch = ((li >> 3) & 1) * 2 + ((li >> 2) & 1);
            const bool writer = NV == 8 ? (li & 1) == 0 : (li & 3) == 0;
            if (writer) {
                const int b = ch >> 2, e = ch & 3;
                sred[wave][0][b * 16 + 4 * kq + e] = v1[0];
                sred[wave][1][b * 16 + 4 * kq + e] = v2[0];
            }
        }
        __syncthreads();
        constexpr int WNC = WN * TN * 16;              // channels of this workgroup
        if (tid < 2 * WNC) {
            const int which = tid / WNC, col = tid % WNC;
            const int wn_ = col / (TN * 16), cc = col % (TN * 16);
            const int c = n0 + col;
            if (c < p.Cout) {
                double v = 0.0;
#pragma unroll
                for (int w = 0; w < WM; ++w) v += (double)sred[wn_ * WM + w][which][cc];
                const unsigned slot = (blockIdx.x + blockIdx.y * gridDim.x) % MRFA_STATS_SLOTS;
                atomicAdd(p.stats + (size_t)slot * 2 * p.Cout + which * p.Cout + c, v);
            }
        }
    }
}

int g_conv_lds = -1;

struct LdsCfg { int tm, tn, wm, wn; LdsGeo g; int grid_x, grid_y; size_t lds; };

// picks the wave / workgroup tile: the largest wave tile that still gives ~1 000 waves (one per SIMD), whole rows per workgroup where a
// row holds at most WM px-tiles, row segments otherwise
bool lds_config(const mrfa_conv_params& p, LdsCfg& c) {
    const int W = p.Wout, H = p.Hout;
    const long long M = (long long)p.N * H * W;
    const int ncols = (p.Cout + 15) / 16 * 16;
    // (TM, TN, WM, WN), in order of preference; measured per launch in the C++ loop (tools/ubench/small_kernels.cpp, MRFA_LDS_CFG sweep; conv_small: 13.9 /
    // 13.9 / 13.5 us): 32 @64^2: (2,1,2,2) 12.3, (2,1,4,1) 12.7, (2,2,4,1) 13.8;  64 @32^2: (2,1,1,4) 11.0, (2,1,2,2) 11.3, (1,1,1,4) 12.8;  128 @16^2:
    // (1,1,1,4) 12.9, (1,1,2,2) 13.5.  32-pixel x 16-channel wave tiles win where the row is >= 32 pixels wide: twice the workgroups of the 32 x 32 tile
    // (two per CU: one workgroup's staging and epilogue overlap the other's k-loop) at the same weight traffic per MFMA.
    static const int cand[6][4] = {{2, 1, 1, 4}, {2, 1, 2, 2}, {2, 1, 4, 1}, {1, 1, 1, 4}, {1, 1, 2, 2}, {2, 2, 4, 1}};
    static const int first = [] { const char* e = getenv("MRFA_LDS_CAND"); return e ? atoi(e) : 0; }();
    static int forced[4] = {0, 0, 0, 0};
    static const bool has_forced = [] { const char* e = getenv("MRFA_LDS_CFG"); return e && sscanf(e, "%d,%d,%d,%d", forced, forced + 1, forced + 2, forced + 3) == 4; }();
    for (int i = first; i < 6; ++i) {
        const int tm = has_forced ? forced[0] : cand[i][0], tn = has_forced ? forced[1] : cand[i][1], wm = has_forced ? forced[2] : cand[i][2],
                  wn = has_forced ? forced[3] : cand[i][3];
        const int pxt = 16 * tm;
        if (W % pxt) continue;
        const int tpr = W / pxt;                       // px-tiles per image row
        int PR, PWW, segs;
        if (wm % tpr == 0) { PR = wm / tpr; PWW = W; segs = 1; }
        else if (tpr % wm == 0) { PR = 1; PWW = wm * pxt; segs = tpr / wm; }
        else continue;
        if (H % PR) continue;
        const int wgc = wn * tn * 16;                  // output channels per workgroup
        if (wgc >= 2 * ncols && i + 1 < 6 && !has_forced) continue;      // half of the workgroup's channel axis (or more) would be padding
        const long long waves = (M / pxt) * ((ncols + tn * 16 - 1) / (tn * 16));
        if (waves < 900 && i + 1 < 6 && !has_forced) continue;
        // channels per staging pass: the whole Cin if the halo fits into 64 KB (two workgroups per CU), else halves / quarters
        auto bytes = [&](int cc) { return (size_t)(PR + 2) * (PWW + 2) * (cc + 4) * 4; };
        int CC = 0;                                    // (compile-time k loops exist for 32 / 64 / 128 channels per pass)
        for (int cc = 128; cc >= 32 && !CC; cc >>= 1)
            if (p.Cin % cc == 0 && bytes(cc) <= 64 * 1024) CC = cc;
        if (!CC) continue;
        c.tm = tm; c.tn = tn; c.wm = wm; c.wn = wn;
        static const int abl = [] { const char* e = getenv("MRFA_LDS_ABL"); return e ? atoi(e) : 0; }();
        c.g = LdsGeo{PR, PWW, PWW + 2, CC, CC + 4, segs, H / PR, abl};
        c.grid_x = p.N * (H / PR) * segs;
        c.grid_y = (ncols + wgc - 1) / wgc;
        c.lds = bytes(CC);
        return true;
    }
    return false;
}

}  // namespace

int mrfa_tuning_conv_lds(int set) {
    // OFF by default (MRFA_CONV_LDS=1 / mrfa_set_tuning("conv_lds", 1)): measured on one box, C++ launch loop: 13.8 / 14.9 / 13.6 us against conv_small's
    // 13.9 / 13.8 / 13.5 (32 @64^2 / 64 @32^2 / 128 @16^2), and 86.5 vs 83.9-84.0 ms in the training step -- see DESIGN 3d
    if (g_conv_lds < 0) { const char* e = getenv("MRFA_CONV_LDS"); g_conv_lds = (e && e[0] == '1'); }
    const int prev = g_conv_lds;
    if (set >= 0) g_conv_lds = set != 0;
    return prev;
}

bool mrfa_conv_lds_eligible(const mrfa_conv_params& p) {
    if (!mrfa_tuning_conv_lds(-1)) return false;
    if (p.kflat > 0 || p.ups || p.nbatch > 1 || p.splitk > 1 || p.tile || p.mask || p.stride > 1 || p.stride < 0) return false;
    if (p.R != 3 || p.S != 3 || p.pad != 1 || p.Hout != p.Hin || p.Wout != p.Win) return false;
    if ((p.Cin & 31) || p.Cin > 256 || p.Cout < 16 || p.Cout > 256 || p.Wout > 64 || (p.Wout & 15)) return false;
    if ((p.ldx & 3) || (p.w_ld & 3) || !aligned16(p.x) || !aligned16(p.w)) return false;
    if ((p.ldy & 3) || !aligned16(p.y) || (p.res && ((p.ldr & 3) || !aligned16(p.res)))) return false;
    if (p.in_scale && (!aligned16(p.in_scale) || !aligned16(p.in_shift))) return false;
    const long long M = (long long)p.N * p.Hout * p.Wout;
    if (M > 65536 * 2) return false;
    // what the bf16-pipe patch kernel does well stays there: long K over many pixels
    if (2.0 * (double)M * p.Cout * 9.0 * p.Cin > 2.6e9) return false;
    LdsCfg c;
    return lds_config(p, c);
}

int mrfa_conv_lds_launch(hipStream_t st, const mrfa_conv_params& p) {
    LdsCfg c;
    if (!lds_config(p, c)) { mrfa_set_error("conv2d(lds): no configuration"); return 1; }
    const dim3 grid((unsigned)c.grid_x, (unsigned)c.grid_y);
    const bool pro = p.in_scale != nullptr;
#define LDS_LAUNCH_K(TM_, TN_, WM_, WN_, KC_)                                                                                    \
    do {                                                                                                                        \
        if (pro) hipLaunchKernelGGL((conv_lds_kernel<TM_, TN_, WM_, WN_, true, KC_>), grid, dim3(256), c.lds, st, p, c.g);       \
        else hipLaunchKernelGGL((conv_lds_kernel<TM_, TN_, WM_, WN_, false, KC_>), grid, dim3(256), c.lds, st, p, c.g);          \
    } while (0)
#define LDS_LAUNCH(TM_, TN_, WM_, WN_)                                                                                          \
    do {                                                                                                                        \
        if (kc == 2) LDS_LAUNCH_K(TM_, TN_, WM_, WN_, 2);                                                                       \
        else if (kc == 4) LDS_LAUNCH_K(TM_, TN_, WM_, WN_, 4);                                                                  \
        else LDS_LAUNCH_K(TM_, TN_, WM_, WN_, 8);                                                                               \
    } while (0)
    const int kc = c.g.CC / 16;
#define LDS_W(TM_, TN_)                                                          \
    do {                                                                         \
        if (c.wm == 4) LDS_LAUNCH(TM_, TN_, 4, 1);                               \
        else if (c.wm == 2) LDS_LAUNCH(TM_, TN_, 2, 2);                          \
        else LDS_LAUNCH(TM_, TN_, 1, 4);                                         \
    } while (0)
    if (c.tm == 2 && c.tn == 2) LDS_W(2, 2);
    else if (c.tm == 2) LDS_W(2, 1);
    else if (c.tn == 2) LDS_W(1, 2);
    else LDS_W(1, 1);
#undef LDS_W
#undef LDS_LAUNCH_K
#undef LDS_LAUNCH
    MRFA_CHECK_LAUNCH("mrfa_conv2d_nhwc(lds)");
    return 0;
}
