/* mrfa_hip.h -- C ABI of libmrfa_hip.so: the MI355X (gfx950) kernels behind the MRFA hot path.
 *
 * The reference (JialeTao/MRFA) has no FFI: its "kernels" are the ATen ops reached from
 * modules/{util,kp_detector,dense_motion,raft,generator}.py.  Each entry point below replaces one ATen op class
 * on that path and cites the reference call sites it stands in for.
 *
 * Conventions
 *   - all activations are fp32 NHWC "views": base pointer + leading dimension `ld` (floats between consecutive
 *     pixels) so producers can write straight into a slice of a concatenated buffer (removes the torch.cat copies
 *     at util.py:262, raft.py:66,68,83,86, generator.py:51,60);
 *   - every function takes the hipStream_t to launch on (as void*), returns 0 on success, non-zero on error;
 *     mrfa_last_error() returns a thread-local message.  Nothing synchronises the device, nothing allocates.
 *   - plain pointers and sizes only; no torch types.
 */
#ifndef MRFA_HIP_H
#define MRFA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* mrfa_last_error(void);
/* ABI version of THIS header; mrfa_version() returns the one the library was built from and a client must refuse a mismatch (the
 * ctypes binding mrfa_amd/hip.py does): parameter structs are passed by pointer without a size field, so a client compiled against an
 * older header would hand over a shorter struct.
 *   1  rounds 1-2.
 *   3  round 3 (shipped as "1"): mrfa_conv_params += mask, ldm, w_phase, w_phase_piece, stride; mrfa_wgrad_params += stride;
 *      mrfa_bnbwd_params += sync; pack modes 8 / 9 (pre-split bf16 weight planes) became k16-chunk-major -- plane[tap][k16 chunk][row][16]
 *      -- and are only meaningful together with w_rows; pack modes 12-15 added.
 *   4  round 4: mrfa_conv_params += w_wino, w_wino_piece (pack modes 16 / 17) with a `..._wino_supported` query; stride = -2.
 *   5  round 4: mrfa_timestamp() added (no struct changed: a version-4 client still works against this library, not the reverse).
 *   6  round 4: mrfa_conv_params += fin_* (BatchNorm finalize inside the convolution call); mrfa_conv2d_wgrad_multi(); mrfa_warp_frame_reflect(); mrfa_bnbwd_params += red_world, red_all, mrfa_bn_param_grad(); mrfa_conv_params += bst_*.
 *   7  round 5: STATISTIC GROUPS -- mrfa_conv_params / mrfa_bnact_params / mrfa_bnbwd_params += groups, mrfa_bn_finalize_groups(), mrfa_bn_param_grad_groups(),
 *      mrfa_conv2d_groups_supported() (see "Statistic groups" below).  REMOVED (measured slower or neutral in round 4, never on by default): stride = -2 (the
 *      strided data gradient), mrfa_bnbwd_params.phase = 3 with its `..._fused_supported` query (`sync` stays in the struct, reserved), the LDS-staged
 *      small convolution behind mrfa_set_tuning("conv_lds"), the Winograd-along-x form (w_wino stays in the struct, reserved; pack modes 16 / 17 and its
 *      `..._wino_supported` query are gone).
 *   8  round 5: mrfa_conv_params += sk_ticket, y_zero (a K split that finishes inside its launch: no init pass, no epilogue pass) with the query
 *      mrfa_conv2d_split_k(); mrfa_layernorm_bwd += scratch; mrfa_resize_sum_multi() / _bwd(); MRFA_PACK_MAX_DESCS 48 -> 240.
 *   9  round 6: mrfa_conv_params.fin_counter points to MRFA_FIN_WORDS zeroed words (was: one); new kernel family behind mrfa_conv2d_nhwc for the keypoint
 *      encoder's <= 128-channel 3x3 layers (conv_lean.hip; mrfa_conv2d_last_config() bit 27; it honours in_scale / in_shift, stats / fin_* / bst_* / groups),
 *      tuning keys "conv_lean", "conv_lean_min_wgs", "conv_lean_geo"; mrfa_wgrad_params += groups with mrfa_conv2d_wgrad_groups_supported(); with
 *      groups > 1 the prologue vectors in_scale / in_shift of mrfa_conv_params are [groups][Cin] (only where mrfa_conv2d_groups_supported() says so:
 *      conv_lean.hip); all-taps multi-problem weight gradient of the same layers (wgrad_lean.hip, tuning key "wgrad_lean").                               */
#define MRFA_ABI_VERSION 9
int mrfa_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * K1/K2/K3/K12: implicit-GEMM convolution / batched NT GEMM on v_mfma_f32_32x32x2_f32.
 * Replaces F.conv2d at util.py:118,142,144,168,187,206; raft.py:54-58,73-78,123-124,138; generator.py:13,32;
 * dense_motion.py:22,25; kp_detector.py:29,36 and the einsum/bmm at raft.py:185 (taps=1, nbatch=B).
 * Also used as the data-gradient conv (weights packed flipped/transposed).                                      */
typedef struct {
    const float* x;        /* input NHWC view                                                                   */
    int ldx;               /* floats between consecutive input pixels                                           */
    int Hin, Win;          /* stored input spatial size                                                         */
    int ups;               /* 1: fused nearest x2 upsample of the input (UpBlock2d, util.py:173); 2: the DATA GRADIENT of such a
                              layer in phase form: x = gradient on the 2Hout x 2Wout grid, y = its 3x3 data gradient summed over
                              the 2x2 pixels of every low-resolution pixel (needs w_phase in pack mode 13; only the shapes
                              mrfa_conv2d_phase_dgrad_supported() reports)                                        */
    int N, Cin;
    const float* w;        /* packed weights, see mrfa_pack_conv_weight                                         */
    int w_ld;              /* floats between consecutive output-channel rows of the packed weight               */
    long long w_tap;       /* floats between consecutive taps (chunked mode)                                    */
    int w_rows;            /* rows present in the packed weight (>= Cout; rows beyond are read as zero)         */
    float* y;              /* output NHWC view                                                                  */
    int ldy;
    int Cout;
    int Hout, Wout;
    int R, S, pad;
    const float* in_scale; /* optional per-Cin affine + ReLU applied to in-bounds inputs (pre-activation BN:    */
    const float* in_shift; /*   ResBlock2d / ChannelBlock2d, util.py:126-128,150-155); v9: [groups][Cin] with groups > 1 */
    int in_relu;
    const float* bias;     /* optional per-Cout bias                                                            */
    const float* out_scale;/* optional per-Cout affine on the output (eval-mode BN folded into the epilogue)    */
    const float* out_shift;
    int relu;              /* ReLU on the output                                                                */
    const float* res;      /* optional residual NHWC view added to the output (ResBlock2d skip, util.py:156)    */
    int ldr;
    double* stats;         /* optional [2*Cout] per-channel sum / sum-of-squares of the stored output (BN train) */
    float alpha;           /* scale on the accumulator before bias (corr volume 1/sqrt(dim), raft.py:185)       */
    int accumulate;        /* 1: y += result (gradient accumulation)                                            */
    int nbatch;            /* >1: batched GEMM; x/w/y advance by the strides below per batch                    */
    long long x_bs, w_bs, y_bs;
    int splitk;            /* >1: K split over gridDim.z, partial sums atomically added into y (y pre-initialised,
                              epilogue options other than alpha are ignored)                                    */
    const int* ktab;       /* flat-K mode (Cin % 32 != 0): table of (dy,dx,ci) per k, see mrfa_build_ktab       */
    int kflat;             /* R*S*Cin in flat mode, 0 in chunked mode                                           */
    int tile;              /* 0 = auto; else force a tile config (tests / tuning)                               */
    const void* w_split;   /* optional (chunked mode): the same weights pre-split into three bf16 pieces (pack modes 8 / 9);  */
    long long w_piece;     /*   bf16 elements between consecutive pieces.  Used by the split-operand kernel when present.     */
    const float* mask;     /* optional (data-gradient launches): forward values of the tensor whose gradient this launch writes, a ReLU   */
    int ldm;               /*   output: the result is multiplied by (mask > 0) BEFORE any accumulation -- the ReLU backward of the producer, */
                           /*   fused into its consumer's data gradient (only where mrfa_conv2d_mask_supported() says so)                    */
    const void* w_phase;   /* optional, with ups = 1 and a 3x3 / pad 1 kernel: the 16 phase-tap weights of the four 2x2 convolutions */
    long long w_phase_piece; /* that equal nearest-x2 + 3x3 (pack mode 12, bf16 pieces).  When present (and the patch-tiled kernel  */
                           /*   applies) the layer runs 16 instead of 36 taps per low-resolution pixel -- same result up to fp32 rounding */
    int stride;            /* 0 / 1: stride 1.  2: strided convolution, Hout = (Hin + 2 pad - R) / 2 + 1 (HRNet's downsampling 3x3 layers,       */
                           /*   hr_base.py:241,253,302,305,365) -- only where mrfa_conv2d_stride_supported() says so.  (-2, the strided data       */
                           /*   gradient of v4-v6, is gone in v7: a stride-1 launch over the zero-stuffed dY is what callers use)                  */
    const void* w_wino;    /* (reserved: the Winograd F(2, 3)-along-x form of v4-v6 -- 1.10-1.16x per launch, nothing in the training step on a    */
    long long w_wino_piece;/*   power-limited part -- is gone in v7; must be NULL / 0)                                                              */
    /* v6, optional, together with `stats` (fin_scale != NULL): FINISH the train-mode BatchNorm that follows this convolution as part of this call --   */
    /* exactly what mrfa_bn_finalize(stats, fin_count, fin_gamma, fin_beta, fin_rmean, fin_rvar, fin_momentum, fin_eps, Cout, 1, fin_scale, fin_shift,   */
    /* fin_mean, fin_invstd) would do after it (fin_rmean / fin_rvar / fin_mean / fin_invstd may be NULL).  The small-problem kernel does it in the     */
    /* SAME launch: the last workgroup to finish (fin_counter: MRFA_FIN_WORDS zero-initialised 32-bit words per call, e.g. behind the statistics block) reduces the */
    /* slots -- the keypoint encoder's ~180 BatchNorm layers per pass then cost two launches each instead of three; every other kernel is followed by   */
    /* the finalize launch inside the call                                                                                                               */
    const float* fin_gamma; const float* fin_beta; float* fin_rmean; float* fin_rvar;
    float fin_momentum, fin_eps; long long fin_count;
    float* fin_scale; float* fin_shift; float* fin_mean; float* fin_invstd;
    unsigned int* fin_counter;
    /* v6, optional, for DATA-GRADIENT launches that are the only writer of y = d(output of a BatchNorm + activation): the first phase of that BatchNorm's  */
    /* backward in this launch's epilogue.  bst_x (row stride bst_ldx) = the BatchNorm's INPUT at the pixels of y; with u = x bst_scale + bst_shift and       */
    /* du = (bst_relu && u <= 0) ? 0 : y, the launch adds sum(du) and sum(du (x - bst_mean) bst_invstd) per channel into `stats` -- laid out and consumed   */
    /* like mrfa_bnbwd_params.red, so that mrfa_bn_act_bwd runs phase 2 only.  One launch less per layer on the keypoint encoder's backward chains.        */
    /* Only where mrfa_conv2d_bwdstats_supported() says so (the small-problem kernel); not together with fin_*.                                               */
    const float* bst_x; int bst_ldx; const float* bst_scale; const float* bst_shift; const float* bst_mean; const float* bst_invstd; int bst_relu;
    int groups;            /* v7: statistic groups of the N samples (0 / 1: one).  See "Statistic groups" below: `stats` = [groups][MRFA_STATS_SLOTS][2 Cout]    */
                           /*   (fin_counter behind ALL of it), fin_scale / _shift / _mean / _invstd and bst_scale / _shift / _mean / _invstd = [groups][Cout],  */
                           /*   fin_count = rows of ONE group, fin_rmean / fin_rvar updated once per group in group order.  Only where                           */
                           /*   mrfa_conv2d_groups_supported() says so                                                                                            */
    /* v8, optional: a K SPLIT THAT FINISHES INSIDE ITS LAUNCH.  The low-resolution layers (4^2 .. 32^2: too few output tiles to fill 256 CUs) split K over       */
    /* gridDim.z and add partial tiles into y with device-scope atomics; through v7 that cost a pass before (y = bias) and a pass behind (affine / residual / */
    /* ReLU / statistics) the launch -- 107 launches per training step on the serial low-resolution chains.  sk_ticket: ZEROED 32-bit words, one per output  */
    /* tile (at least ceil(N Hout Wout / 32) ceil(Cout / 32) max(nbatch, 1) of them cover every tile shape), fresh for every call: each workgroup takes a    */
    /* ticket of its tile once its atomics have completed, and the one that draws the last re-reads the summed tile (device-scope loads), adds the bias,     */
    /* applies out_scale / res / relu, stores it and accumulates `stats` (one statistic group per tile: groups > 1 is honoured).  y_zero = 1: the caller      */
    /* states that y already holds zeros (e.g. a slice of a zero-filled arena): the init pass is dropped as well (with a bias: only together with            */
    /* sk_ticket).  Ignored by launches that do not split (mrfa_conv2d_split_k() tells) and with accumulate = 1.                                              */
    unsigned int* sk_ticket;
    int y_zero;
} mrfa_conv_params;
int mrfa_conv2d_split_k(const mrfa_conv_params* p);                  /* K slices a call with these parameters would use (1: no split; y / sk_ticket may be NULL) */
int mrfa_conv2d_reads_fp32_weights(const mrfa_conv_params* p);       /* v9: 0 = the kernel this call would run reads w_split / w_phase only: `w` may be any non-NULL
                                                                        pointer and the fp32 layout need not exist (y may be NULL for the query)            */

/* Statistic groups (v7).  The reference runs its keypoint encoder as separate calls on the source frames, the driving frames (and the transformed driving frames
 * of the equivariance loss): modules/model.py:185-186,234 -- so in train mode every BatchNorm normalises each of those batches with ITS OWN batch statistics and
 * updates its running statistics once per call, in call order.  Here the calls travel as ONE batch of N = groups x B samples (half / a third of the launches of a
 * latency-bound chain, twice / three times the rows per launch): samples [g N / groups, (g + 1) N / groups) form statistic group g, and every train-mode BatchNorm
 * quantity is kept per group: statistics blocks [groups][MRFA_STATS_SLOTS][2 C], scale / shift / mean / invstd [groups][C], the backward's `red`
 * [groups][MRFA_STATS_SLOTS][2 C]; running_mean / running_var take `groups` momentum updates in group order (r <- (1 - m) r + m stat_g, g = 0, 1, ..), gamma / beta
 * gradients are summed over the groups.  Everything without cross-sample coupling (convolutions, activations, LayerNorm, attention, weight gradients) is
 * unchanged by construction.  Requirements: N % groups == 0 and no output tile of the launch may straddle two groups (the query functions check the kernels' tile
 * heights: rows of one group % 128 == 0 always qualifies).  groups = 0 / 1 is the ungrouped behaviour of v6.                                                    */
int mrfa_conv2d_groups_supported(const mrfa_conv_params* p);         /* 1: a call with these parameters honours groups > 1 for stats / fin_* / bst_*  */

/* BatchNorm statistics buffers (`stats` of mrfa_conv_params, mrfa_bias_act, mrfa_bn_stats, mrfa_bn_finalize; `red` of mrfa_bnbwd_params): MRFA_STATS_SLOTS
 * consecutive blocks of 2*C doubles ([sum | sum of squares] per channel), zero-initialised by the caller.  A producing workgroup adds
 * into block (workgroup index % MRFA_STATS_SLOTS) and mrfa_bn_finalize sums the blocks: device-scope atomics on ONE address are
 * performed at the memory side at ~20-40 ns each (the per-XCD L2s are not coherent), so 256-512 workgroups adding into the same
 * 2*C words cost 5-21 us per launch on the MTIA prior's 0.6-GFLOP layers (tools/ubench/small_kernels.cpp) -- more than the layer. */
#define MRFA_STATS_SLOTS 32
/* v9: mrfa_conv_params.fin_counter = MRFA_FIN_WORDS zeroed 32-bit words per call (one through v8): the kernels that know their workgroups' indices draw the
 * finalize tickets from eight of them -- 256-512 returning atomics on ONE word cost 3.3 us per launch -- and only the shards' last arrivers from word 0   */
#define MRFA_FIN_WORDS 16
int mrfa_conv2d_nhwc(void* stream, const mrfa_conv_params* p);
int mrfa_conv2d_stride_supported(const mrfa_conv_params* p);         /* 1: a call with these parameters honours stride = 2                */
int mrfa_conv2d_mask_supported(const mrfa_conv_params* p);           /* 1: a call with these parameters honours `mask`                    */
int mrfa_conv2d_phase_dgrad_supported(const mrfa_conv_params* p);    /* 1: a call with these parameters (ups = 2) is implemented          */
int mrfa_conv2d_bwdstats_supported(const mrfa_conv_params* p);   /* 1: a call with these parameters honours bst_*                                          */
/* Matrix-pipe selection for the 128 x 128 chunked tiles of mrfa_conv2d_nhwc and mrfa_conv2d_wgrad_nhwc (process-wide):
 *   0  v_mfma_f32_32x32x2_f32 (fp32 operands; 157 TF/s pipe)
 *   1  fp32 operands split exactly into 3 bf16 pieces, 6 v_mfma_f32_32x32x16_bf16 products, fp32 accumulate
 *      (fp32-accurate: the dropped cross terms are < 2^-23 of each product; 2.5 PF/s pipe / 6)
 *   2  the same kernels keeping only a1*b0 + a0*b1 + a0*b0 ("bf16x3"): product error ~2^-16 (a 16-17 bit significand, 32x
 *      finer than TF32), 1.3x faster tiles; an opt-in mode, NOT the default
 *   3  plain bf16: operands rounded to nearest-even bf16, ONE product, fp32 accumulate and fp32 storage ("MFMA bf16 conv
 *      tiles", BASELINE config 4); bf16-autocast accuracy (2^-9 operand error), opt-in                              */
int mrfa_set_mfma_mode(int mode);
int mrfa_get_mfma_mode(void);
/* (BM << 16) | (BN << 4) | (flat << 1) | (splitk > 1) chosen by the most recent call on this thread (for roofline accounting);
 * bit 2: split-operand kernel, bit 3: small-problem kernel, bit 28: patch-tiled 3x3 kernel (conv_halo.hip)                       */
int mrfa_conv2d_last_config(void);
/* Kernel-selection knobs (process-wide; tests and tuning -- the defaults are the measured choices).  Returns the previous value,
 * -1 for an unknown key.  Keys:
 *   "conv_halo"            1 / 0: patch-tiled 3x3 kernel on / off (default 1; also MRFA_CONV_HALO=0 in the environment)
 *   "conv_halo_min_tiles"  workgroup-count unit of the patch-tiled kernel's selection rule (default 128: 2x = one per CU in general,
 *                          1x for <= 128-wide tiles on >= 64-pixel-wide, >= 64-channel layers)
 *   "conv_halo_pr"         0 = patch height by workgroup count (default), 4 / 8 = forced
 *   "conv_halo_phase"      1 / 0: phase form (four 2x2 convolutions) of fused-upsample layers that carry w_phase (default 1)
 *   "conv_halo_bn256"      1 / 0: 256-channel workgroup tiles where Cout pads to 256 anyway (default 1)
 *   "wgrad_halo"           1 / 0: all-taps weight-gradient kernel of the 3x3 layers on / off (default 1; MRFA_WGRAD_HALO=0)
 *   "wgrad_halo_min_wgs"   fewest workgroups for which that kernel is chosen (default 192)
 *   "conv_fewout3"         1 / 0: channel-lane kernels of the 3x3 layers with 1 / 2 output channels on / off (default 1)
 *   "conv_small"           1 / 0: one-wave-per-tile small-problem kernels on / off (default 1)                                    */
int mrfa_set_tuning(const char* key, int value);

/* weight-gradient (and TN GEMM): dW[tap][co][ci] += sum_p dY[p][co] * X'[p + tap][ci]   (X' = prologue(ups(x)))
 * Replaces the weight-gradient half of conv2d backward for every call site above, and d(k_s) of raft.py:185.    */
typedef struct {
    const float* x; int ldx; int Hin, Win; int ups; int N, Cin;
    const float* in_scale; const float* in_shift; int in_relu;
    const float* dy; int ldy; int Cout; int Hout, Wout;
    int R, S, pad;
    float* dw;             /* [taps][Cout][Cin] fp32, accumulated with atomics (caller zero-initialises)         */
    float* dbias;          /* optional [Cout], accumulated with atomics                                          */
    float alpha;
    int nbatch; long long x_bs, dy_bs, dw_bs;
    int ksplit;            /* 0 = auto: number of pixel-range splits                                            */
    const int* ktab;       /* flat mode (small / odd Cin): GEMM N axis = taps*Cin gathered through the table      */
    int kflat;             /* R*S*Cin in flat mode, else 0                                                      */
    int tile8_off;         /* tuning: 1 disables the 8-wave variant of the 128x128 tile                         */
    float* ws;             /* optional scratch for the two-stage split reduction (used when many splits hit few   */
    long long ws_bytes;    /*   weights: 1x1 / few-channel layers); NULL => atomics only                          */
    int stride;            /* as mrfa_conv_params.stride: dY pixel (oy, ox) pairs with X pixel (stride * oy + r - pad, ...); only where   */
                           /*   mrfa_conv2d_wgrad_stride_supported() says so                                                                 */
    int groups;            /* v9: statistic groups of the N samples for the PROLOGUE (0 / 1: one): in_scale / in_shift = [groups][Cin], sample n   */
                           /*   uses row n / (N / groups) -- the BatchNorm-apply + ReLU of a batched keypoint-encoder pass folded into the weight  */
                           /*   gradient of the convolution that consumes it.  Only where mrfa_conv2d_wgrad_groups_supported() says so             */
} mrfa_wgrad_params;

int mrfa_conv2d_wgrad_nhwc(void* stream, const mrfa_wgrad_params* p);
/* v6: n independent weight gradients (or ones that share dw / dbias, which are accumulated atomically) in as few launches as possible: the problems the
 * one-wave-per-block kernel takes run up to 28 per launch, every other one as if mrfa_conv2d_wgrad_nhwc had been called for it (the keypoint encoder's
 * ~415 small weight gradients per training step, issued together after its backward chains)                                                          */
int mrfa_conv2d_wgrad_multi(void* stream, const mrfa_wgrad_params* ps, int n);
int mrfa_conv2d_wgrad_stride_supported(const mrfa_wgrad_params* p);  /* 1: a call with these parameters honours stride = 2                       */
int mrfa_conv2d_wgrad_groups_supported(const mrfa_wgrad_params* p);  /* 1: a call with these parameters honours groups > 1 (prologue rows per group) */
int mrfa_conv2d_wgrad_lean_supported(const mrfa_wgrad_params* p);    /* 1: mrfa_conv2d_wgrad_multi runs this problem on the all-taps kernel of the keypoint
                                                                        encoder's <= 128-channel 3x3 layers (wgrad_lean.hip) in the current matrix mode    */

/* weight (un)packing between the reference's OIHW parameter layout and the kernel layouts.
 * mode 0: OIHW -> fwd  chunked [tap][CoutPad][CinPad]            (CoutPad % 128 == 0, CinPad % 32 == 0)
 * mode 1: OIHW -> fwd  flat    [CoutPad][KPad], k = tap*Cin+ci   (KPad % 32 == 0)
 * mode 2: OIHW -> dgrad chunked [tap'][CinPad128][CoutPad32], tap' = flipped tap (data gradient = conv with
 *         180-degree rotated, in/out-transposed weights)
 * mode 3: OIHW -> dgrad flat   [CinPad128][KPad], k = tap'*Cout+co
 * mode 4: grad [tap][Cout][Cin] -> OIHW, accumulating (dst += src)
 * mode 5: OIHW -> [Cout][tap][Cin]            (few-output direct kernels)
 * mode 6: grad [Cout][tap][Cin] -> OIHW, accumulating
 * mode 7: OIHW -> [Cin][tap'][Cout] flipped   (data gradient of a few-INPUT conv run as a few-output conv over dY)
 * modes 4|16 and 6|16: as 4 / 6 but overwriting (dst = ...) instead of accumulating
 * mode 8: as mode 0 but split for the bf16x6 kernels: three planes [piece][tap][CoutPad][CinPad] of bf16 with
 *         w = piece0 + piece1 + piece2 exactly (piece_k = top 16 bits of the residual); mode 9: likewise for mode 2
 *         (modes 8 / 9: batched entry point only).  Every bf16 plane (modes 8, 9, 12, 13, 14, 15) stores a tap's [rows][cols] matrix
 *         K16-CHUNK-MAJOR: element (row r, column f) at (f / 16) * (rows * 16) + r * 16 + f % 16 -- the 16-channel slab of a tap that a
 *         conv workgroup stages is one contiguous run of rows x 32 bytes (mrfa_conv_params.w_rows = rows must be set with w_split / w_phase)
 * modes 14 / 15: the layouts of modes 8 / 9 as ONE bf16 plane rounded to nearest even (plain bf16 mode, mrfa_set_mfma_mode(3));
 *         passed through mrfa_conv_params.w_split with w_piece = 0 (batched entry point only)
 * mode 13: mode 12 transposed ([piece][16][CinPad128][CoutPad32]) for the phase data gradient (ups = 2)
 * mode 12: 3x3 only: the phase weights of UpBlock2d's nearest-x2 + conv (see mrfa_conv_params.w_phase): three bf16 planes
 *         [piece][16 phase taps][CoutPad128][CinPad32], phase tap = (py*2+px)*4 + a*2+b, weight = sum of the 3x3 taps that read
 *         the same low-resolution pixel (batched entry point only)
 * (modes 16 / 17, the Winograd-along-x weights of v4-v6, are gone in v7)                                                              */
int mrfa_pack_conv_weight(void* stream, const float* src, float* dst, int Cout, int Cin, int R, int S, int mode);

/* Batched forms: all layouts of many convolutions per launch (descriptor table passed by value in the kernel
 * arguments, MRFA_PACK_MAX_DESCS per launch: 240 since round 5 (a 16 KB argument block; 48 before) because a launch lasts as long as one workgroup's tile
 * whatever the number of tiles), used once per training step for all convolutions of the path (~600 descriptors with the MTIA prior).
 * mrfa_pack_conv_weights_multi: for every desc, src (OIHW) -> dst[k] in layout mode[k] (0,1,2,3,5,7 as above).  Padding
 *   elements of the destinations are NOT written: allocate those buffers zero-filled.
 * mrfa_unpack_wgrads_multi: for every desc, dst (OIHW gradient) += src (accumulator, [tap][Cout][Cin], or
 *   [Cout][tap][Cin] when fewout != 0)  -- modes 4 / 6 above.                                                         */
#define MRFA_PACK_MAX_DESCS 240
typedef struct {
    const float* src;
    float* dst[3];
    int mode[3];
    int ndst;
    int Cout, Cin, R, S;
} mrfa_pack_desc;
typedef struct {
    const float* src;
    float* dst;
    int Cout, Cin, T, fewout;
} mrfa_unpack_desc;
int mrfa_pack_conv_weights_multi(void* stream, const mrfa_pack_desc* descs /* host array */, int n);
int mrfa_unpack_wgrads_multi(void* stream, const mrfa_unpack_desc* descs /* host array */, int n);
/* host-side helper: fills tab[KPad] for flat mode; entry = (dy+128) | (dx+128)<<8 | c<<16, invalid k -> -1      */
int mrfa_build_ktab(int* tab_host, int C, int R, int S, int pad, int flip);

/* K4: direct (VALU/LDS) convolution for layers with <= 4 output channels (generator.final 64->3 7x7, refine.conv2
 * 128->2, refine.convo2 128->1, dense_motion.occlusion 108->1; modules/generator.py:32, raft.py:76,78, dense_motion.py:25)
 * and, with mode-7 weights, the data gradient of corr_enc.convf1 (2 input channels, raft.py:56).
 * w: [Cout][R*R][Cin] (pack mode 5 / 7); Cin %% 4 == 0; stride 1; square kernel.                                 */
int mrfa_conv_fewout_fwd(void* stream, const float* x, int ldx, int N, int H, int W, int Cin, const float* w,
                         const float* bias, float* y, int ldy, int Cout, int R, int pad, int accumulate);
/* dw [Cout][R*R][Cin] += sum_p dY[p][co] * X[p+tap-pad][ci] (atomics; caller zero-initialises); dbias optional    */
int mrfa_conv_fewout_wgrad(void* stream, const float* x, int ldx, int N, int H, int W, int Cin, const float* dy, int lddy,
                           int Cout, int R, int pad, float* dw, float* dbias);
/* data gradient of the few-output 3x3 layers: dx[p][ci] (+)= sum_{tap,co} dy[p - tap + pad][co] * w[co][tap][ci]   (w in pack mode 5;
 * refine.conv2 / convo2, raft.py:76,78).  Only the shapes mrfa_conv_fewout_dgrad_supported() reports (3x3 / pad 1, Cout in {1, 2},
 * Cin in {64, 128, 256}, lddx % 4 == 0): the caller takes the generic conv2d route (pack mode 2 / 3) otherwise.                      */
int mrfa_conv_fewout_dgrad(void* stream, const float* dy, int lddy, int N, int H, int W, int Cout, const float* w, float* dx, int lddx,
                           int Cin, int R, int pad, int accumulate, const float* mask /* optional, as mrfa_conv_params.mask */, int ldm);
int mrfa_conv_fewout_dgrad_supported(int Cin, int Cout, int R, int pad, int W, int lddx);

/* ------------------------------------------------------------------------------------------------------------
 * K5/K6/K7: BatchNorm (train + eval) with fused ReLU / 2x2 avg-pool / occlusion blend.
 * Replaces batch_norm at util.py:122,146-147,170,189,208, relu, avg_pool2d (util.py:190) and the blend at
 * generator.py:57.                                                                                              */
int mrfa_bn_stats(void* stream, const float* x, int ldx, long long rows, int C, double* stats /*[2C] zeroed*/);
int mrfa_bn_finalize(void* stream, const double* stats, long long count, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, int C, int train,
                     float* scale, float* shift, float* mean_out, float* invstd_out);
/* v7, train mode with statistic groups: stats [groups][MRFA_STATS_SLOTS][2C], count = rows of ONE group, outputs [groups][C]; the running statistics (may be
 * NULL) take one momentum update per group, in group order -- what `groups` successive train-mode calls of the module do to them (model.py:185-186,234)          */
int mrfa_bn_finalize_groups(void* stream, const double* stats, long long count, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, int C, int groups,
                            float* scale, float* shift, float* mean_out, float* invstd_out);
typedef struct {
    const float* x; int ldx; int N, H, W, C;
    const float* scale; const float* shift; int relu;
    int pool;                        /* 1: 2x2 average pool after the activation (DownBlock2d)                   */
    const float* blend_a; int lda;   /* optional: y = a*occ + act(x)*(1-occ)  (generator.py:57)                  */
    const float* occ; int ldo;       /* occ: one channel per pixel                                               */
    float* y; int ldy;
    const float* res; int ldr;       /* optional residual added BEFORE the activation: y = act(bn(x) + res) (HRNet
                                        BasicBlock / Bottleneck, transformer/hr_base.py:50-51,92-93); not with pool/blend */
    int groups;                      /* v7: statistic groups (0 / 1: one): scale / shift = [groups][C], sample n uses row n / (N / groups); N % groups == 0 */
} mrfa_bnact_params;
int mrfa_bn_act_fwd(void* stream, const mrfa_bnact_params* p);

typedef struct {
    const float* x; int ldx; int N, H, W, C;     /* pre-BN activation (saved)                                    */
    const float* scale; const float* shift; int relu; int pool;
    const float* mean; const float* invstd; const float* gamma;
    const float* dy; int lddy;                   /* gradient wrt the op output (pooled size if pool)             */
    const float* blend_a; int lda; const float* occ; int ldo;
    float* dblend_a; int ldda;                   /* += dy*occ                                                    */
    float* docc; int lddo;                       /* += sum_c dy*(a - act)                                        */
    double* red;                                 /* [MRFA_STATS_SLOTS][2C] zeroed: sum(dz), sum(dz*xhat); phase 1 adds
                                                    into block (workgroup % MRFA_STATS_SLOTS), phase 2 sums the blocks */
    float* dx; int lddx;                         /* += BN input gradient                                         */
    float* dgamma; float* dbeta;                 /* += (phase 2)                                                 */
    int train;                                   /* 0: eval-mode BN (no batch-statistics terms)                  */
    int phase;                                   /* 1: reductions, 2: apply.  (3, both in one launch with a grid-wide barrier, is gone in v7:
                                                    measured 6 ms slower per training step beside chip-filling kernels)                 */
    int dx_overwrite;                            /* phase 2: dx = ... instead of dx += ... (this BN is the only writer
                                                    of its input's gradient: no zero fill, no read of dx)              */
    const float* res; int ldr;                   /* the forward's residual (needed for the ReLU mask)            */
    float* dres; int lddr;                       /* += gradient wrt the residual (phase 1), may be null          */
    unsigned int* sync;                          /* (reserved: the arrival counter of the removed phase 3)       */
    int red_world;                               /* v6, phase 2: > 1 = `red` holds the sums of that many ranks (SyncBatchNorm: the caller all-reduced the
                                                    slots between the phases): the batch means divide by red_world x the local row count.  dgamma / dbeta
                                                    are LOCAL sums: take them from the local `red` with mrfa_bn_param_grad() before the exchange and pass NULL here */
    int red_all;                                 /* v6, phase 2: 1 = sum ALL MRFA_STATS_SLOTS blocks of `red` (phase 1 of this entry point uses -- and phase 2 by
                                                    default sums -- only as many as the launch geometry needs; sums accumulated by a convolution's bst_* epilogue
                                                    are spread over all of them)                                                                              */
    int groups;                                  /* v7: statistic groups (0 / 1: one): scale / shift / mean / invstd = [groups][C], red = [groups][MRFA_STATS_SLOTS][2C];
                                                    the batch means of phase 2 are per group (count = rows / groups), dgamma / dbeta += the sums of ALL groups      */
} mrfa_bnbwd_params;
int mrfa_bn_act_bwd(void* stream, const mrfa_bnbwd_params* p);
/* v6: dgamma[c] += sum over the slots of red[.][C + c], dbeta[c] += sum of red[.][c] (what phase 2 adds when it is given dgamma / dbeta)                  */
int mrfa_bn_param_grad(void* stream, const double* red, int C, float* dgamma, float* dbeta);
int mrfa_bn_param_grad_groups(void* stream, const double* red, int C, int groups, float* dgamma, float* dbeta);     /* v7: red = [groups][MRFA_STATS_SLOTS][2C] */

/* ------------------------------------------------------------------------------------------------------------
 * K10: bilinear grid_sample, zeros padding, NHWC, channel-vectorised.
 * mode 0: grid = normalised (x,y) in [-1,1], align_corners=False  (dense_motion.py:83; raft.py:166,168,271)
 * mode 1: grid = flow (dx,dy) in pixels; sample at (x+dx, y+dy), align_corners=True semantics
 *         (bilinear_sampler(img, flow+coords_grid), util.py:26-38; raft.py:247,260,302)
 * grid is (N,Ho,Wo,2) fp32 with leading dimension ldg.  Output image n samples input image n / in_rep (in_rep = 11
 * for DenseMotion's repeated source, dense_motion.py:80-81, else 1); in_bstride = floats between input images.     */
int mrfa_grid_sample_fwd(void* stream, const float* in, int ldi, long long in_bstride, int in_rep, int Hi, int Wi, int C,
                         const float* grid, int ldg, int N, int Ho, int Wo, float* out, int ldo, int mode);
int mrfa_grid_sample_bwd(void* stream, const float* in, int ldi, long long in_bstride, int in_rep, int Hi, int Wi, int C,
                         const float* grid, int ldg, int N, int Ho, int Wo, const float* dout, int lddo, int mode,
                         float* din /*+= atomics, may be null*/, int lddi, long long din_bstride,
                         float* dgrid /*+= , may be null*/, int lddg);

/* v6: Transform.transform_frame (model.py:44-48): bilinear F.grid_sample(frame, grid, padding_mode="reflection") with align_corners=False on NCHW frames
 * (N,C,H,W) -> (N,C,Ho,Wo); grid (N,Ho,Wo,2) contiguous, normalised (x, y).  Forward only: nothing differentiates through the equivariance warp.   */
int mrfa_warp_frame_reflect(void* stream, const float* in_nchw, int N, int C, int H, int W, const float* grid, int Ho, int Wo, float* out_nchw);

/* K9: bilinear resize, align_corners=True (F.interpolate, raft.py:161-162,205-206,228,243,266-267,279-295,308)  */
int mrfa_resize_bilinear_fwd(void* stream, const float* in, int ldi, int N, int Hi, int Wi, int C,
                             float* out, int ldo, int Ho, int Wo, float scale_mul, int accumulate);
int mrfa_resize_bilinear_bwd(void* stream, const float* dout, int lddo, int N, int Hi, int Wi, int C,
                             float* din /*+=*/, int lddi, int Ho, int Wo, float scale_mul);

/* v8: many "dst (=|+=) sum_k mul_k resize(src_k)" records in ONE launch (table by value in the kernel arguments) -- the flow / occlusion re-composition between two
 * refinement levels and the running-flow updates (raft.py:258-262,276-295): ~16 copies / resizes of 1- and 2-channel maps per level, each a launch of its own on a
 * chain where nothing else runs.  One thread per dst element, the terms in table order: the arithmetic and the order of the mrfa_copy_view / mrfa_resize_bilinear_*
 * launches a record replaces.  mrfa_resize_sum_multi: dst (N, Hd, Wd, C) = an output, term k = an input (N, Hs, Ws, C) resized (align_corners=True) to Hd x Wd,
 * overwrite = 1: the first term overwrites dst.  mrfa_resize_sum_multi_bwd: dst = the gradient of an INPUT of size Hd x Wd, term k = the gradient of an output of size
 * Hs x Ws >= Hd x Wd it was resized to (the adjoint, gather form; same size: a copy); always accumulates.  Records of one call must not overlap in what they write. */
#define MRFA_RESIZE_SUM_MAX 48
#define MRFA_RESIZE_SUM_TERMS 4
typedef struct {
    float* dst; int ldd, N, Hd, Wd, C;
    int nterm, overwrite;
    struct { const float* src; int lds, Hs, Ws; float mul; } term[MRFA_RESIZE_SUM_TERMS];
} mrfa_resize_sum_desc;
int mrfa_resize_sum_multi(void* stream, const mrfa_resize_sum_desc* descs, int n);
int mrfa_resize_sum_multi_bwd(void* stream, const mrfa_resize_sum_desc* descs, int n);

/* K11/K13: correlation-pyramid window lookup (CorrBlock, raft.py:12-48).  vol0: (N*Q, Hs*Ws) rows, one source map
 * per query pixel; vol1: the 2x2 source-pooled level (N*Q, Hs/2*Ws/2).  coords (N,h1,w1,2) pixel (x,y) already
 * multiplied by the level scale.  out (N,h1,w1,2*(2r+1)^2) NHWC; channel = lvl*49 + a*7 + b at (x+a-r, y+b-r).    */
int mrfa_corr_lookup_fwd(void* stream, const float* vol0, const float* vol1, int Hs, int Ws, const float* coords,
                         int ldc, long long Q, int radius, float* out, int ldo);
int mrfa_corr_lookup_bwd(void* stream, const float* vol0, const float* vol1, int Hs, int Ws, const float* coords,
                         int ldc, long long Q, int radius, const float* dout, int lddo,
                         float* dvol0 /*+= atomics*/, float* dvol1, float* dcoords /*+=*/, int lddc);

/* ------------------------------------------------------------------------------------------------------------
 * small layout / elementwise helpers                                                                            */
int mrfa_nchw_to_nhwc(void* stream, const float* src, float* dst, int ldd, int N, int C, int H, int W, int accumulate);
int mrfa_nhwc_to_nchw(void* stream, const float* src, int lds, float* dst, int N, int C, int H, int W, int accumulate);
int mrfa_avgpool2_fwd(void* stream, const float* x, int ldx, int N, int H, int W, int C, float* y, int ldy);
int mrfa_sumpool2_acc(void* stream, const float* x, int ldx, int N, int Ho, int Wo, int C, float* y, int ldy, float mul);
int mrfa_unpool2_acc(void* stream, const float* dy, int lddy, int N, int Ho, int Wo, int C, float* dx, int lddx, float mul);
/* y = act(x (+ bias)) elementwise on a view; act 0 none, 1 relu, 2 sigmoid; stats optional (BN after split-K)    */
int mrfa_bias_act(void* stream, const float* x, int ldx, long long rows, int C, const float* bias, int act,
                  float* y, int ldy, double* stats);
/* dx += dy * act'(y) ; act 1: (y>0), 2: y*(1-y)                                                                  */
int mrfa_act_bwd(void* stream, const float* y, int ldy, const float* dy, int lddy, long long rows, int C, int act,
                 float* dx, int lddx, int accumulate);
int mrfa_copy_view(void* stream, const float* x, int ldx, long long rows, int C, float* y, int ldy, float mul, int accumulate);
/* y = a*occ + b*(1-occ) on NHWC views with a 1-channel occ  (generator.py:47,57,63)                              */
int mrfa_blend_fwd(void* stream, const float* a, int lda, const float* b, int ldb, const float* occ, int ldo,
                   long long rows, int C, float* y, int ldy);
int mrfa_blend_bwd(void* stream, const float* a, int lda, const float* b, int ldb, const float* occ, int ldo,
                   const float* dy, int lddy, long long rows, int C, float* da, int ldda, float* db, int lddb,
                   float* docc, int lddo);
/* K18: AntiAliasInterpolation2d (util.py:318-326) computing only the kept outputs: NCHW in -> NHWC out           */
int mrfa_antialias_down(void* stream, const float* x_nchw, int N, int C, int H, int W, const float* kern, int k,
                        int stride, float* y, int ldy);
int mrfa_colsum(void* stream, const float* x, int ldx, long long rows, int C, float* out /*+=*/);
/* measurement aid: a one-thread kernel that stores the device's constant-rate clock (100 MHz, s_memrealtime) into *dst when the stream reaches it.
 * Captured into a hipGraph like any other launch, so the phases of a REPLAYED step can be timed without a profiler (tools/step_phases.py);
 * nothing on the product path issues it                                                                                                     */
int mrfa_timestamp(void* stream, unsigned long long* dst);

/* ------------------------------------------------------------------------------------------------------------
 * K21: the MTIA prior, TokenPose_B (reference modules/transformer/pose_tokenpose_b.py:16-50,
 * hr_base.py:294-450, tokenpose_base.py:33-94,137-158,406-468).  Its convolutions, Linear layers (1x1 convolutions over
 * token rows) and BatchNorms run on K1-K8; these are the remaining ops.  All views NHWC / row-major fp32, C % 4 == 0 and
 * 16-byte aligned unless noted; every *_bwd ACCUMULATES (+=) into its gradient outputs.                          */
/* y[n,oy,ox,:] = x[n,oy*stride,ox*stride,:]: a stride-2 3x3 pad-1 convolution (hr_base.py:231,312,316,357) is the
 * stride-1 convolution kept at the even pixels                                                                   */
int mrfa_subsample_fwd(void* stream, const float* x, int ldx, int N, int H, int W, int C, int stride, float* y, int ldy);
int mrfa_subsample_bwd(void* stream, const float* dy, int lddy, int N, int H, int W, int C, int stride, float* dx /*+=*/, int lddx);
/* y = act(base + nearest_upsample(lo, factor)): the branch fusion of HighResolutionModule.forward (hr_base.py:278-289;
 * nn.Upsample(mode='nearest') at :218); factor 1 = plain add (+ReLU).  lo is (N,Hl,Wl,C), base / y (N,Hl*f,Wl*f,C)  */
int mrfa_upsample_add_act_fwd(void* stream, const float* lo, int ldl, int N, int Hl, int Wl, int C, int factor, const float* base, int ldb,
                              int relu, float* y, int ldy);
int mrfa_upsample_add_act_bwd(void* stream, const float* y, int ldy, const float* dy, int lddy, int N, int Hl, int Wl, int C, int factor,
                              int relu, float* dlo /*+=, may be null*/, int lddl, float* dbase /*+=, may be null*/, int lddb);
/* torch.nn.LayerNorm over the last dimension (tokenpose_base.py:33,38; any C <= 1024, no alignment needs); mean / rstd
 * [rows] are saved for the backward                                                                               */
int mrfa_layernorm_fwd(void* stream, const float* x, int ldx, long long rows, int C, const float* gamma, const float* beta, float eps,
                       float* y, int ldy, float* mean, float* rstd);
/* scratch (v8; may be NULL): MRFA_LN_SLOTS * 2 * C + 1 ZEROED floats, fresh per call -- the workgroups' parameter-gradient partials meet in MRFA_LN_SLOTS slotted
 * blocks instead of in dgamma / dbeta themselves (hundreds of workgroups adding into the same 2 C words serialise at the memory side) and the launch's last
 * workgroup (ticket: the word behind the slots) adds their sum to dgamma / dbeta -- nothing else may write those while the launch runs                       */
#define MRFA_LN_SLOTS 16
int mrfa_layernorm_bwd(void* stream, const float* x, int ldx, const float* dy, int lddy, long long rows, int C, const float* gamma,
                       const float* mean, const float* rstd, float* dx /*+=*/, int lddx, float* dgamma /*+=*/, float* dbeta /*+=*/, float* scratch);
/* exact (erf) GELU, nn.GELU() at tokenpose_base.py:51                                                             */
int mrfa_gelu_fwd(void* stream, const float* x, int ldx, long long rows, int C, float* y, int ldy);
int mrfa_gelu_bwd(void* stream, const float* x, int ldx, const float* dy, int lddy, long long rows, int C, float* dx /*+=*/, int lddx);
/* softmax(scale * q k^T) v per (sample, head), Attention.forward tokenpose_base.py:72-94 without mask.  qkv: (B*n) rows
 * of [q | k | v], each heads*d wide with head h at columns [h*d, (h+1)*d) ('b n (h d)'); out: (B*n) x (heads*d);
 * lse / delta: [B*heads*n] (log-sum-exp saved by the forward; scratch of the backward).  d in {16, 24, 32}.      */
int mrfa_attention_fwd(void* stream, const float* qkv, int ld, int B, int n, int heads, int d, float scale, float* out, int ldo, float* lse);
int mrfa_attention_bwd(void* stream, const float* qkv, int ld, const float* out, int ldo, const float* dout, int lddo, const float* lse,
                       float* delta, int B, int n, int heads, int d, float scale, float* dqkv /*+=*/, int lddq);

/* ------------------------------------------------------------------------------------------------------------
 * K22: the reference's training losses (modules/model.py:26-141,219-254): VGG19 perceptual
 * pyramid and the ImagePyramide of the generated image.  The VGG convolutions run on K1 (bias + ReLU in the epilogue).  */
/* nn.MaxPool2d(2, 2) of torchvision's vgg19.features (model.py:88-105).  Backward: dx[first maximum of the window in
 * (row, column) scan order] += dy, the index ATen's max_pool2d_with_indices records.  C % 4 == 0, 16-byte aligned views.  */
int mrfa_maxpool2_fwd(void* stream, const float* x, int ldx, int N, int H, int W, int C, float* y, int ldy);
int mrfa_maxpool2_bwd(void* stream, const float* x, int ldx, int N, int H, int W, int C, const float* dy, int lddy, float* dx /*+=*/, int lddx);
/* nn.MaxPool2d(3, stride 2, padding 1) of the resnet18 stem of BGMotionPredictor (bg_motion_predictor.py:12; torchvision
 * resnet.py maxpool): Ho = (H-1)/2+1, windows clipped to the image; backward scatters (atomically: windows overlap) to the first
 * maximum in scan order                                                                                            */
int mrfa_maxpool3s2_fwd(void* stream, const float* x, int ldx, int N, int H, int W, int C, float* y, int ldy);
int mrfa_maxpool3s2_bwd(void* stream, const float* x, int ldx, int N, int H, int W, int C, const float* dy, int lddy, float* dx /*+=*/, int lddx);
/* out_sum[0] += coef * sum |x - y| (fp64): with coef = weight / numel one perceptual term
 * weight * torch.abs(x_vgg[i] - y_vgg[i].detach()).mean() of model.py:226-227; backward dx += gscale[0] * coef * sign(x - y) (gscale: device scalar = upstream gradient, may be null = 1)  */
int mrfa_l1_diff_fwd(void* stream, const float* x, int ldx, const float* y, int ldy, long long rows, int C, double coef, double* out_sum);
int mrfa_l1_diff_bwd(void* stream, const float* x, int ldx, const float* y, int ldy, long long rows, int C, const float* gscale, float coef,
                     float* dx /*+=*/, int lddx);
/* gradient of mrfa_antialias_down (K18) with respect to its NCHW input image: dx += K^T dy                        */
int mrfa_antialias_down_bwd(void* stream, const float* dy, int lddy, int N, int C, int H, int W, const float* kern, int k, int stride,
                            float* dx_nchw /*+=*/);

/* ------------------------------------------------------------------------------------------------------------
 * K14-K17: the prior-motion stage's small dense ops, csrc/prior.hip.
 *
 * K15  kp2gaussian (reference modules/util.py:59-87) on the [-1,1]^2 grid of make_coordinate_grid (util.py:90-108), written into
 *      an NHWC view (channel k of out row (b,y,x)):  out = exp(-0.5 |g(y,x) - kp[b,k]|^2 / variance) [+ pos[k,y,x]]
 *      (pos = RaftFlow.pos_embedding (1,K,H,W), raft.py:177-178; may be null).  Backward: dkp (B,K,2) += , dpos (K,H,W) += .   */
int mrfa_kp_gaussian_fwd(void* stream, const float* kp, const float* pos, int B, int K, int H, int W, float variance, float* out, int ldo);
int mrfa_kp_gaussian_bwd(void* stream, const float* kp, int B, int K, int H, int W, float variance, const float* dout, int lddo,
                         float* dkp /*+=, may be null*/, float* dpos /*+=, may be null*/);

/* K15+K16+K10: DenseMotionNetwork.create_heatmap_representations / create_sparse_motions / create_deformed_source_image and the
 * channel interleave of its hourglass input (reference modules/dense_motion.py:36-46, 48-76, 78-85, 117-119) in ONE kernel.
 *   motion_0(g) = g, or from_homogeneous(bg[b] (g,1)) with a background affine (dense_motion.py:69-73);
 *   motion_k(g) = js[b,k] inv(jd[b,k]) (g - kd[b,k]) + ks[b,k]   (k = 1..K; closed-form 2x2 inverse; jd = js = null: identity Jacobians);
 *   heat_0 = 0, heat_k = exp(-|g - kd|^2 / 2 var) - exp(-|g - ks|^2 / 2 var);
 *   warp_k = grid_sample(src[b], motion_k), bilinear, zeros padding, align_corners=False (dense_motion.py:83).
 * Outputs: motions (B*(K+1), H, W, 2) rows with leading dimension ldm; inp (B,H,W) rows of ldi floats, channels
 * [k (C+1)] = heat_k, [k (C+1) + 1 + c] = warp_k[c]  (= the reference's (B, (K+1)(C+1), H, W) hourglass input in NHWC);
 * sparse (B, K+1, C, H, W) dense = the returned `sparse_deformed` (may be null).
 * Backward reads dinp (same geometry as inp, lddi), dmotions (gradient of `motions` from K17, same ldm; may be null), dsparse
 * (may be null) and adds into dkd, dks (B,K,2), djd, djs (B,K,4), dbg (B,9) (each may be null); the source image gets no
 * gradient (it is the network input).                                                                            */
typedef struct mrfa_prior_params {
    const float *kd, *ks, *jd, *js, *bg;
    const float* src; int lds;
    int B, K, H, W, C;
    float inv_var;                       /* 1 / kp_variance */
    float* motions; int ldm;
    float* inp; int ldi;
    float* sparse;
    /* backward only */
    const float* dinp; int lddi;
    const float* dmotions;
    const float* dsparse;
    float *dkd, *dks, *djd, *djs, *dbg;
} mrfa_prior_params;
int mrfa_prior_motion_fwd(void* stream, const mrfa_prior_params* p);
int mrfa_prior_motion_bwd(void* stream, const mrfa_prior_params* p);

/* K14+K17: mask = softmax over the K1 = K+1 motions of logit (B,H,W,K1 NHWC, ldl); deformation (B,H,W,2) = sum_k mask_k motion_k
 * (reference modules/dense_motion.py:129-136); also exports mask and the logits as NCHW (B,K1,H,W) (the returned `mask`,
 * `logit_mask`).  Backward: dlogit (NHWC view, lddl) += softmax backward of (ddeformation . motion_k + dmask_k) + dlogit_nchw,
 * dmotions (same geometry as motions) += mask_k ddeformation.  ddeformation / dmask / dlogit_nchw / dmotions may be null.   */
int mrfa_softmax_combine_fwd(void* stream, const float* logit, int ldl, const float* motions, int ldm, int B, int H, int W, int K1,
                             float* deformation, float* mask_nchw, float* logit_nchw);
int mrfa_softmax_combine_bwd(void* stream, const float* motions, int ldm, int B, int H, int W, int K1, const float* mask_nchw,
                             const float* ddeformation, const float* dmask_nchw, const float* dlogit_nchw, float* dlogit /*+=*/, int lddl,
                             float* dmotions /*+=*/);

/* K14: KPDetector head (reference modules/kp_detector.py:90-120): heat = softmax over the H*W positions of logits[b,:,:,k] / T,
 * kp[b,k] = sum heat * grid, jac[b,k,:] = sum heat * jm[b,:,:,0..3] (jm: the 4 Jacobian maps, NHWC, may be null).  stat (B,K,2)
 * receives (max, 1/sum) for the backward, which recomputes heat:  dlogits (NHWC view, lddl) +=, djm (lddj) += .           */
int mrfa_kp_head_fwd(void* stream, const float* logits, int ldl, const float* jm, int ldj, int B, int H, int W, int K, float temperature,
                     float* kp, float* jac, float* stat);
int mrfa_kp_head_bwd(void* stream, const float* logits, int ldl, const float* jm, int ldj, int B, int H, int W, int K, float temperature,
                     const float* kp, const float* jac, const float* stat, const float* dkp, const float* djac, float* dlogits /*+=*/,
                     int lddl, float* djm /*+=*/, int lddj);

/* ------------------------------------------------------------------------------------------------------------
 * K20: optimizer step of the data-parallel path on FLAT fp32 buffers (every parameter / gradient / Adam moment of a
 * parameter group in one allocation, 16-byte aligned slices).  Replaces torch.optim.Adam(betas=(0.5, 0.999)).step()
 * (reference train.py:21, 70) and nn.utils.clip_grad_norm_(.., norm_type=inf) (train.py:65-67); the 1/world of the
 * data-parallel gradient mean is folded in as `gscale`.
 * state: MRFA_ADAM_STATE_FLOATS device floats per parameter group:
 *   [0] t (steps taken)  [1] lr / (1 - beta1^t)  [2] 1 / sqrt(1 - beta2^t)  [3] lr (written by the host, read by
 *   mrfa_adam_prepare, so a hipGraph replay follows an LR scheduler)  [4..8) inf-norm slots of this step's gradients */
#define MRFA_ADAM_STATE_FLOATS 8
#define MRFA_ADAM_CLIP_SLOTS 4
/* for every group g < ngroups: t += 1, recompute [1], [2] from [3], zero the inf-norm slots                      */
int mrfa_adam_prepare(void* stream, float* state, int ngroups, double beta1, double beta2);
/* state[4 + clip_slot] = max(state[4 + clip_slot], max_i |g[i]|)   (n % 4 == 0, g 16-byte aligned)                */
int mrfa_grad_absmax(void* stream, const float* g, long long n, float* state, int clip_slot);
/* g' = gscale * g * (clip_slot < 0 ? 1 : min(1, max_norm / (gscale * state[4 + clip_slot] + 1e-6)));
 * m += (1 - beta1) (g' - m);  v = beta2 v + (1 - beta2) g'^2;  w -= state[1] * m / (sqrt(v) * state[2] + eps)     */
int mrfa_adam_flat(void* stream, float* w, const float* g, float* m, float* v, long long n, const float* state,
                   double beta1, double beta2, float eps, float gscale, int clip_slot, float max_norm);
/* (betas are doubles: torch.optim.Adam forms 1 - beta and beta^t in double precision before rounding to fp32)    */

#ifdef __cplusplus
}
#endif
#endif
