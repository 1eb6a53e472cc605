"""Kernel-level parity (-m gpu): every C-ABI entry point of libmrfa_hip.so on the MI355X against the executable
specification of the ABI (oracle/capi_emulator.py, plain torch CPU ops) on identical seeded inputs.  fp32 tolerance
2e-4 relative to the output scale (MFMA fp32 = exact fmaf chains; differences are summation order only)."""
import ctypes as C

import numpy as np
import pytest
import torch

from mrfa_amd import hip
from mrfa_amd.utils.prng import det_uniform
from oracle.capi_emulator import Emulator

pytestmark = pytest.mark.gpu


class Side:
    """one execution side: (library object, device, stream)"""

    def __init__(self, gpu: bool):
        self.gpu = gpu
        self.L = hip.lib() if gpu else Emulator()
        self.dev = torch.device("cuda:0") if gpu else torch.device("cpu")

    @property
    def s(self):
        return hip.stream_ptr() if self.gpu else 0

    def t(self, name, shape, lo=-1.0, hi=1.0):
        return det_uniform(name, shape, lo, hi).to(self.dev)

    def z(self, shape, dtype=torch.float32):
        return torch.zeros(shape, dtype=dtype, device=self.dev)

    def garbage(self, shape):
        return torch.full(shape, float("nan"), dtype=torch.float32, device=self.dev)

    def call(self, name, *args):
        rc = getattr(self.L, name)(self.s, *args)
        if rc:
            raise RuntimeError(f"{name} rc={rc}: {self.L.mrfa_last_error()}")

    def done(self, *ts):
        if self.gpu:
            torch.cuda.synchronize()
        return [x.detach().cpu().double() for x in ts]


def both(fn):
    a = fn(Side(False))
    b = fn(Side(True))
    return a, b


def assert_close(ref, got, tol=2e-4, what=""):
    for i, (r, g) in enumerate(zip(ref, got)):
        assert r.shape == g.shape
        assert torch.isfinite(g).all(), f"{what}[{i}] non-finite"
        scale = max(r.abs().max().item(), 1e-6)
        err = (r - g).abs().max().item()
        assert err <= tol * scale + 1e-6, f"{what}[{i}]: max err {err:.3e} vs scale {scale:.3e}"


def pack(side, w, mode):
    Co, Ci, R, S = w.shape
    T = R * S
    n = {0: T * ((Co + 127) // 128 * 128) * ((Ci + 31) // 32 * 32),
         1: ((Co + 127) // 128 * 128) * ((T * Ci + 31) // 32 * 32),
         2: T * ((Ci + 127) // 128 * 128) * ((Co + 31) // 32 * 32),
         3: ((Ci + 127) // 128 * 128) * ((T * Co + 31) // 32 * 32)}[mode]
    out = side.garbage((n,))
    side.call("mrfa_pack_conv_weight", w.contiguous().data_ptr(), out.data_ptr(), Co, Ci, R, S, mode)
    return out


def ktab(side, Cc, R, S, pad):
    kp = (R * S * Cc + 31) // 32 * 32
    buf = (C.c_int * kp)()
    assert side.L.mrfa_build_ktab(buf, Cc, R, S, pad, 0) == 0
    return torch.tensor(list(buf), dtype=torch.int32, device=side.dev)


def conv_case(side, *, N=2, H=12, W=10, Cin=64, Cout=96, R=3, pad=1, ups=0, pro=False, bias=True, relu=True, res=False,
              stats=False, acc=False, alpha=1.0, tile=0, splitk=1, oaff=False, ldx_extra=0, ldy_extra=4, wsplit=False, wphase=False, mask=False,
              stride=1, fin=False, bst=False, groups=1, fused=False, tag="c"):
    S = R
    x = side.t(f"{tag}/x", (N * H * W, Cin + ldx_extra))
    w = side.t(f"{tag}/w", (Cout, Cin, R, S), -0.2, 0.2)
    flat = (Cin % 32) != 0
    Hv, Wv = H << ups, W << ups
    Ho, Wo = (Hv + 2 * pad - R) // stride + 1, (Wv + 2 * pad - S) // stride + 1
    ldy = (Cout + 3) // 4 * 4 + ldy_extra
    y = side.t(f"{tag}/y0", (N * Ho * Wo, ldy)) if acc else (side.z((N * Ho * Wo, ldy)) if fused == "zero" else side.garbage((N * Ho * Wo, ldy)))
    p = hip.ConvParams()
    p.x, p.ldx, p.Hin, p.Win, p.ups, p.N, p.Cin = x.data_ptr(), x.shape[1], H, W, ups, N, Cin
    cop = (Cout + 127) // 128 * 128
    keep = []
    if flat:
        wp = pack(side, w, 1)
        kp = (R * S * Cin + 31) // 32 * 32
        kt = ktab(side, Cin, R, S, pad)
        keep.append(kt)
        p.w, p.w_ld, p.w_tap, p.kflat, p.ktab = wp.data_ptr(), kp, 0, R * S * Cin, kt.data_ptr()
    else:
        wp = pack(side, w, 0)
        p.w, p.w_ld, p.w_tap, p.kflat = wp.data_ptr(), Cin, cop * Cin, 0
        if wsplit:                  # weights pre-split into three bf16 pieces (pack mode 8), or ONE plane rounded to nearest even (mode 14)
            piece = R * S * cop * Cin
            rne = wsplit == "rne"
            wsb = torch.zeros((1 if rne else 3) * piece, dtype=torch.int16, device=side.dev)
            d = hip.PackDesc()
            d.src, d.Cout, d.Cin, d.R, d.S, d.ndst = w.data_ptr(), Cout, Cin, R, S, 1
            d.dst[0], d.mode[0] = wsb.data_ptr(), 14 if rne else 8
            side.call("mrfa_pack_conv_weights_multi", C.pointer(d), 1)
            keep.append(wsb)
            p.w_split, p.w_piece = wsb.data_ptr(), 0 if rne else piece
        if wphase:                  # the 16 phase-tap weights of nearest-x2 + 3x3 (pack mode 12)
            ppiece = 16 * cop * Cin
            wpb = torch.zeros(3 * ppiece, dtype=torch.int16, device=side.dev)
            d = hip.PackDesc()
            d.src, d.Cout, d.Cin, d.R, d.S, d.ndst = w.data_ptr(), Cout, Cin, R, S, 1
            d.dst[0], d.mode[0] = wpb.data_ptr(), 12
            side.call("mrfa_pack_conv_weights_multi", C.pointer(d), 1)
            keep.append(wpb)
            p.w_phase, p.w_phase_piece = wpb.data_ptr(), ppiece
    p.w_rows = cop
    p.y, p.ldy, p.Cout, p.Hout, p.Wout = y.data_ptr(), ldy, Cout, Ho, Wo
    p.R, p.S, p.pad = R, S, pad
    if pro:
        sc, sh = side.t(f"{tag}/sc", (Cin,), 0.5, 1.5), side.t(f"{tag}/sh", (Cin,), -0.3, 0.3)
        keep += [sc, sh]
        p.in_scale, p.in_shift, p.in_relu = sc.data_ptr(), sh.data_ptr(), 1
    if bias:
        bt = side.t(f"{tag}/b", (Cout,))
        keep.append(bt)
        p.bias = bt.data_ptr()
    if oaff:
        osc, osh = side.t(f"{tag}/osc", (Cout,), 0.5, 1.5), side.t(f"{tag}/osh", (Cout,), -0.3, 0.3)
        keep += [osc, osh]
        p.out_scale, p.out_shift = osc.data_ptr(), osh.data_ptr()
    p.relu = int(relu)
    if res:
        rt = side.t(f"{tag}/res", (N * Ho * Wo, Cout + 4))
        keep.append(rt)
        p.res, p.ldr = rt.data_ptr(), Cout + 4
    G = groups                      # v7, statistic groups: [G][MRFA_STATS_SLOTS][2C] statistics, [G][C] per-channel vectors
    stbuf = side.z((G * hip.STATS_SLOTS * 2 * Cout + hip.FIN_WORDS // 2,), torch.float64)  # [G][MRFA_STATS_SLOTS][2C], summed by the consumer (+ the finalize ticket words)
    st = stbuf[:G * hip.STATS_SLOTS * 2 * Cout].view(G, hip.STATS_SLOTS, 2 * Cout)
    if stats:
        p.stats, p.groups = st.data_ptr(), G
        if G > 1:
            assert side.L.mrfa_conv2d_groups_supported(C.byref(p)) == 1
    if bst:                         # v6: the launch writes d(act(bn(x))) and accumulates the first phase of that BatchNorm's backward (bst_*)
        assert stats and not fin
        bx = side.t(f"{tag}/bst_x", (N * Ho * Wo, Cout + 4))
        bsc, bsh = side.t(f"{tag}/bst_sc", (G * Cout,), 0.5, 1.5), side.t(f"{tag}/bst_sh", (G * Cout,), -0.3, 0.3)
        bme, biv = side.t(f"{tag}/bst_me", (G * Cout,), -0.2, 0.2), side.t(f"{tag}/bst_iv", (G * Cout,), 0.5, 2.0)
        keep += [bx, bsc, bsh, bme, biv]
        p.bst_x, p.bst_ldx, p.bst_relu = bx.data_ptr(), Cout + 4, int(bst != "linear")
        p.bst_scale, p.bst_shift, p.bst_mean, p.bst_invstd = bsc.data_ptr(), bsh.data_ptr(), bme.data_ptr(), biv.data_ptr()
        assert side.L.mrfa_conv2d_bwdstats_supported(C.byref(p)) == 1
    fin_out = []
    if fin:                         # v6: the BatchNorm that follows finished inside the call (scale, shift, mean, invstd, running statistics)
        assert stats
        gam, bet = side.t(f"{tag}/gamma", (Cout,), 0.5, 1.5), side.t(f"{tag}/beta", (Cout,), -0.3, 0.3)
        rm, rv = side.t(f"{tag}/rm", (Cout,), -0.2, 0.2), side.t(f"{tag}/rv", (Cout,), 0.5, 1.5)
        fin_out = [side.garbage((G * Cout,)) for _ in range(4)] + [rm, rv]
        keep += [gam, bet]
        p.fin_gamma, p.fin_beta, p.fin_rmean, p.fin_rvar = gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr()
        p.fin_momentum, p.fin_eps, p.fin_count = 0.1, 1e-5, N * Ho * Wo // G
        p.fin_scale, p.fin_shift, p.fin_mean, p.fin_invstd = (t.data_ptr() for t in fin_out[:4])
        p.fin_counter = stbuf.data_ptr() + 8 * G * hip.STATS_SLOTS * 2 * Cout
    if mask:                        # fused ReLU backward: the result is multiplied by (mask > 0) before the accumulation
        mk = side.t(f"{tag}/mask", (N * Ho * Wo, Cout + 8))
        keep.append(mk)
        p.mask, p.ldm = mk.data_ptr(), Cout + 8
    p.alpha, p.accumulate, p.nbatch, p.splitk, p.tile = alpha, int(acc), 1, splitk, tile
    if fused:                       # v8: a K split that finishes inside its launch ("zero": the output is handed over zero-filled, no init pass either)
        tk = side.z((-(-N * Ho * Wo // 32) * -(-Cout // 32),), torch.int32)
        keep.append(tk)
        p.sk_ticket, p.y_zero = tk.data_ptr(), int(fused == "zero")
        assert side.L.mrfa_conv2d_split_k(C.byref(p)) > 1, "the case is meant to split K"
    if stride > 1:
        p.stride = stride
        assert side.L.mrfa_conv2d_stride_supported(C.byref(p)) == 1
    if mask and side.gpu:
        assert side.L.mrfa_conv2d_mask_supported(C.byref(p)) == 1
    side.call("mrfa_conv2d_nhwc", C.byref(p))
    if fin and side.gpu and fin == "small":
        assert side.L.mrfa_conv2d_last_config() & 8, "not dispatched to the one-wave-per-tile kernel"
    return side.done(y[:, :Cout], st.sum(1).reshape(-1), *fin_out)


CONV_CASES = {
    "basic": dict(),
    "t128x128_oddM": dict(N=1, H=9, W=11, Cin=64, Cout=130, tile=(128 << 16) | 128),
    "t128x128_8wave": dict(N=1, H=9, W=11, Cin=64, Cout=130, tile=(128 << 16) | 0x8000 | 128, pro=True),
    "t128x64": dict(Cout=96, tile=(128 << 16) | 64),
    "t128x96": dict(Cout=192, tile=(128 << 16) | 96),
    "t128x96_auto": dict(N=4, H=32, W=32, Cin=64, Cout=192, R=1, pad=0),
    "t128x32": dict(Cout=24, tile=(128 << 16) | 32),
    "t64x128": dict(Cout=126, tile=(64 << 16) | 128),
    "t64x64": dict(Cout=64, tile=(64 << 16) | 64),
    "t32x128": dict(N=1, H=4, W=4, Cin=128, Cout=256, tile=(32 << 16) | 128),
    "ups": dict(ups=1, H=6, W=5),
    "prologue": dict(pro=True, relu=False),
    "residual_stats": dict(res=True, stats=True, relu=False, Cout=64),
    "accumulate_alpha": dict(acc=True, alpha=0.37, relu=False, bias=False),
    "out_affine": dict(oaff=True),
    "conv1x1": dict(R=1, pad=0, Cin=128, Cout=192),
    "conv7x7_c64_to3": dict(R=7, pad=3, Cin=64, Cout=3, relu=False),
    "flat_7x7_c3": dict(R=7, pad=3, Cin=3, Cout=64, ldx_extra=1),
    "flat_7x7_c2": dict(R=7, pad=3, Cin=2, Cout=128, ldx_extra=2),
    "flat_1x1_c98": dict(R=1, pad=0, Cin=98, Cout=128, ldx_extra=2),
    "flat_7x7_c35_pad0": dict(R=7, pad=0, Cin=35, Cout=10, relu=False, ldx_extra=1),
    "flat_3x3_c13_pro": dict(Cin=13, Cout=64, pro=True),
    "flat_ups_c44": dict(Cin=44, Cout=128, ups=1, H=5, W=7),
    "splitk4_relu_stats": dict(N=1, H=4, W=4, Cin=256, Cout=128, splitk=4, stats=True),
    "splitk_auto_small": dict(N=1, H=2, W=2, Cin=512, Cout=512, splitk=0, stats=True, relu=False),
    "splitk3_accumulate": dict(N=1, H=4, W=4, Cin=128, Cout=64, splitk=3, acc=True, relu=False, bias=False),
    # v8: the split finishes inside the launch (sk_ticket; "zero": zero-filled output, no init pass): every epilogue option, ragged tiles, statistic groups
    "fused_splitk4_relu_stats": dict(N=1, H=4, W=4, Cin=256, Cout=128, splitk=4, stats=True, fused="zero"),
    "fused_splitk_auto_res_stats": dict(N=1, H=2, W=2, Cin=512, Cout=512, splitk=0, stats=True, res=True, fused="zero"),
    "fused_splitk3_ticket_only_affine": dict(N=1, H=5, W=7, Cin=128, Cout=96, splitk=3, oaff=True, fused="ticket"),
    "fused_splitk_plain_nobias": dict(N=2, H=4, W=4, Cin=256, Cout=130, splitk=5, relu=False, bias=False, fused="zero"),
    "fused_splitk_fp32_tile": dict(N=1, H=6, W=6, Cin=128, Cout=64, splitk=4, tile=(64 << 16) | 64, stats=True, res=True, fused="zero"),
    "fused_splitk_fin": dict(N=2, H=8, W=8, Cin=512, Cout=256, splitk=0, stats=True, relu=False, fin=True, fused="zero"),
    "tile_fin_nosplit": dict(N=4, H=32, W=32, Cin=128, Cout=96, R=1, pad=0, tile=(128 << 16) | 96, stats=True, relu=False, fin=True),
    "fused_splitk_groups2_fin": dict(N=4, H=8, W=8, Cin=512, Cout=512, splitk=0, stats=True, relu=False, fin=True, groups=2, fused="zero"),
    # conv_small.hip (one wave per 16..32-row tile, no LDS): the MTIA prior's shapes, every epilogue option, ragged M / Cout
    "small_hr32_stats": dict(N=2, H=16, W=16, Cin=32, Cout=32, stats=True, relu=False, bias=False),
    "small_hr64_res": dict(N=2, H=8, W=8, Cin=64, Cout=64, res=True, relu=True),
    "small_hr128_acc": dict(N=1, H=16, W=16, Cin=128, Cout=128, acc=True, alpha=0.5, relu=False, bias=False),
    "small_linear_192_576": dict(N=2, H=1, W=276, Cin=192, Cout=576, R=1, pad=0, res=True, relu=False),
    "small_linear_576_192_affine": dict(N=1, H=1, W=276, Cin=576, Cout=192, R=1, pad=0, oaff=True),
    "small_ragged": dict(N=1, H=7, W=9, Cin=48, Cout=40, stats=True),
    "small_wide_m": dict(N=8, H=32, W=32, Cin=64, Cout=64, stats=True, relu=False),
    # HRNet's 3x3 stride-1 layers at the bench batch (whatever kernel the dispatch picks): every epilogue option, the pre-activation prologue, ragged channels
    "bench_hr32_at64": dict(N=8, H=64, W=64, Cin=32, Cout=32, stats=True, relu=False, bias=False),
    "bench_hr64_at32_res": dict(N=8, H=32, W=32, Cin=64, Cout=64, res=True, relu=True, stats=True),
    "bench_hr128_at16_acc": dict(N=8, H=16, W=16, Cin=128, Cout=128, acc=True, alpha=0.5, relu=False, bias=False),
    "bench_layer1_two_passes_pro": dict(N=4, H=64, W=64, Cin=64, Cout=64, pro=True, stats=True),
    "bench_ragged_c96_c40": dict(N=2, H=16, W=32, Cin=96, Cout=40, stats=True, oaff=True),
    "bench_row_segments": dict(N=1, H=8, W=64, Cin=32, Cout=128, res=True),
    # strided gather (HRNet's downsampling layers: hr_base.py:241,253,302,305,365), even and odd input sizes
    # v6: first phase of a BatchNorm backward in a data-gradient launch's epilogue (bst_*): ReLU and linear, accumulate, ragged tiles
    "bst_hr32": dict(N=4, H=32, W=32, Cin=32, Cout=32, stats=True, relu=False, bias=False, bst=True),
    "bst_hr64_acc": dict(N=2, H=16, W=16, Cin=64, Cout=64, stats=True, relu=False, bias=False, acc=True, bst=True),
    "bst_linear_ragged": dict(N=1, H=7, W=9, Cin=64, Cout=48, stats=True, relu=False, bias=False, bst="linear"),
    "bst_hr128": dict(N=2, H=16, W=16, Cin=128, Cout=128, stats=True, relu=False, bias=False, bst=True),
    "bst_hr64_m2048": dict(N=2, H=32, W=32, Cin=64, Cout=64, stats=True, relu=False, bias=False, bst=True, ldy_extra=0),
    "bst_1x1_256_64": dict(N=2, H=16, W=16, Cin=256, Cout=64, R=1, pad=0, stats=True, relu=False, bias=False, bst=True, ldy_extra=0),
    # v6: the BatchNorm finalize inside the call -- by the launch's last workgroup (one-wave-per-tile kernel), by a launch behind it (every other kernel)
    "fin_small_hr32": dict(N=8, H=64, W=64, Cin=32, Cout=32, stats=True, relu=False, bias=False, fin="small"),
    "fin_small_hr128": dict(N=2, H=16, W=16, Cin=128, Cout=128, stats=True, relu=False, bias=False, fin="small"),
    "fin_flat_ragged": dict(N=1, H=7, W=9, Cin=48, Cout=40, stats=True, relu=False, fin=True),
    "fin_small_stride2": dict(N=2, H=32, W=32, Cin=64, Cout=64, stride=2, stats=True, relu=False, bias=False, fin="small"),
    "fin_big_tile": dict(N=2, H=32, W=32, Cin=128, Cout=64, tile=(128 << 16) | 64, stats=True, relu=False, fin=True),
    "small_stride2_stem": dict(N=2, H=32, W=32, Cin=64, Cout=64, stride=2, stats=True, relu=False, bias=False),
    # the fp32 tile kernel's flat-K gather with stride 2 (HRNet's 3 -> 64 stem convolution, hr_base.py:302), statistic groups, finalize behind the launch
    "flat_stride2_stem_c3": dict(N=4, H=32, W=32, Cin=3, Cout=64, stride=2, stats=True, relu=False, bias=False, fin=True, groups=2, ldx_extra=1),
    "flat_stride2_odd_c13": dict(N=1, H=13, W=17, Cin=13, Cout=40, stride=2, relu=True),
    # v7: statistic groups (the source / driving / transformed-driving encoder calls as one batch): per-group statistics, finalize and bst_* on the
    # one-wave-per-tile kernel (all three wave tiles), the patch-tiled kernel, the row tiles and a strided layer
    "groups2_small_hr32_fin": dict(N=16, H=64, W=64, Cin=32, Cout=32, stats=True, relu=False, bias=False, fin="small", groups=2),
    "groups2_small_hr128_fin": dict(N=4, H=16, W=16, Cin=128, Cout=128, stats=True, relu=False, bias=False, fin="small", groups=2),
    "groups3_small_hr64_fin": dict(N=6, H=16, W=16, Cin=64, Cout=64, stats=True, relu=False, bias=False, fin="small", groups=3),
    "groups2_small_64rows": dict(N=2, H=8, W=8, Cin=64, Cout=64, stats=True, relu=False, bias=False, fin="small", groups=2),
    "groups2_small_stride2_fin": dict(N=4, H=32, W=32, Cin=64, Cout=64, stride=2, stats=True, relu=False, bias=False, fin="small", groups=2),
    "groups2_bst_hr32": dict(N=4, H=32, W=32, Cin=32, Cout=32, stats=True, relu=False, bias=False, bst=True, groups=2),
    "groups3_bst_hr64_acc": dict(N=6, H=16, W=16, Cin=64, Cout=64, stats=True, relu=False, bias=False, acc=True, bst=True, groups=3),
    "groups2_big_tile_fin": dict(N=4, H=32, W=32, Cin=128, Cout=64, tile=(128 << 16) | 64, stats=True, relu=False, fin=True, groups=2),
    "groups2_flat_stem": dict(N=4, H=16, W=16, Cin=3, Cout=64, stats=True, relu=False, bias=False, fin=True, groups=2),
    "groups2_1x1_64_256": dict(N=4, H=64, W=64, Cin=64, Cout=256, R=1, pad=0, stats=True, relu=False, bias=False, fin=True, groups=2),
    "groups2_auto_no_splitk": dict(N=4, H=8, W=8, Cin=512, Cout=512, splitk=0, stats=True, relu=False, fin=True, groups=2),
    "small_stride2_fuse_odd": dict(N=1, H=13, W=17, Cin=32, Cout=128, stride=2, stats=True, relu=False, bias=False),
}


@pytest.mark.parametrize("name", list(CONV_CASES))
def test_conv2d(name):
    ref, got = both(lambda s: conv_case(s, tag=f"conv/{name}", **CONV_CASES[name]))
    assert_close(ref, got, what=name)


def test_fused_finalize_sees_every_workgroups_statistics():
    """mrfa_conv_params.fin_* on the one-wave-per-tile kernel: the launch's LAST workgroup reduces the statistics slots without a release / acquire fence
    (device-scope atomics + device-scope loads, conv_small.hip).  A workgroup's sums missed -- or a stale L2 line read -- would show as a mean / variance
    that disagrees with the slots as they stand after the launch: 60 launches over three shapes (512 / 32 / 256 workgroups), each checked against
    its own slot sums in float64."""
    side = Side(True)
    shapes = [dict(N=8, H=64, W=64, Cin=32, Cout=32), dict(N=2, H=16, W=16, Cin=128, Cout=128), dict(N=8, H=32, W=32, Cin=64, Cout=64)]
    for it in range(60):
        kw = shapes[it % 3]
        cnt, Co = kw["N"] * kw["H"] * kw["W"], kw["Cout"]
        y, st, scale, shift, mean, invstd, rm, rv = conv_case(side, tag=f"fin/stress{it % 3}", stats=True, relu=False, bias=False, fin="small", **kw)
        m = st[:Co] / cnt
        var = (st[Co:] / cnt - m * m).clamp_min(0)
        assert torch.allclose(mean, m, rtol=1e-6, atol=1e-7), (it, float((mean - m).abs().max()))
        assert torch.allclose(invstd, 1.0 / torch.sqrt(var + 1e-5), rtol=2e-6), (it, float((invstd - 1.0 / torch.sqrt(var + 1e-5)).abs().max()))
        # and the statistics themselves are the output's (every workgroup's atomics arrived)
        assert torch.allclose(st[:Co], y.sum(0), rtol=1e-6, atol=1e-4)


SPLIT_CASES = {
    "oddM_c130": dict(N=1, H=9, W=11, Cin=64, Cout=130, tile=(128 << 16) | 128),
    "pro_relu": dict(N=2, H=16, W=16, Cin=96, Cout=128, pro=True, tile=(128 << 16) | 128),
    "ups_res_stats": dict(ups=1, H=6, W=5, Cin=128, Cout=256, res=True, stats=True, relu=False, tile=(128 << 16) | 128),
    "conv1x1_affine": dict(R=1, pad=0, Cin=256, Cout=128, oaff=True, N=4, H=16, W=16, tile=(128 << 16) | 128),
    "splitk4": dict(N=1, H=4, W=4, Cin=256, Cout=128, splitk=4, stats=True, tile=(128 << 16) | 128),
    "fused_splitk4": dict(N=1, H=4, W=4, Cin=256, Cout=128, splitk=4, stats=True, res=True, tile=(128 << 16) | 128, fused="zero"),
    "fused_splitk6_bn64_ticket": dict(N=2, H=8, W=8, Cin=256, Cout=50, splitk=6, oaff=True, tile=(128 << 16) | 64, fused="ticket"),
    "accumulate_alpha": dict(acc=True, alpha=0.37, relu=False, bias=False, Cin=64, Cout=128, tile=(128 << 16) | 128),
    "auto_256_to_128": dict(N=4, H=128, W=128, Cin=256, Cout=128),
    "bn64_c64": dict(N=2, H=32, W=32, Cin=128, Cout=64, tile=(128 << 16) | 64, res=True, stats=True),
    "bn64_c50_pro": dict(N=2, H=16, W=48, Cin=64, Cout=50, tile=(128 << 16) | 64, pro=True),
    "bn64_auto": dict(N=4, H=128, W=128, Cin=128, Cout=64, ups=0),
    "groups2_conv1x1_fin": dict(R=1, pad=0, Cin=64, Cout=256, N=4, H=32, W=32, stats=True, relu=False, bias=False, fin=True, groups=2, tile=(128 << 16) | 128),
    "groups2_bn64": dict(N=2, H=32, W=32, Cin=128, Cout=64, tile=(128 << 16) | 64, res=True, stats=True, groups=2),
}


@pytest.mark.parametrize("wsplit", [False, True])
@pytest.mark.parametrize("name", list(SPLIT_CASES))
def test_conv2d_split_operand_mode(name, wsplit):
    """mrfa_set_mfma_mode(1): fp32 operands split exactly into three bf16 pieces, six bf16 MFMA products, fp32 accumulate --
    must satisfy the SAME fp32 tolerance against the CPU specification as the native fp32 MFMA kernel"""
    L = hip.lib()
    ref = conv_case(Side(False), tag=f"split/{name}", **SPLIT_CASES[name])
    assert L.mrfa_set_mfma_mode(1) == 0
    try:
        got = conv_case(Side(True), tag=f"split/{name}", wsplit=wsplit, **SPLIT_CASES[name])
        assert L.mrfa_conv2d_last_config() & 4, "the split-operand kernel did not run"
    finally:
        L.mrfa_set_mfma_mode(0)
    assert_close(ref, got, what="split " + name)


HALO_CASES = {          # conv_halo.hip: 3x3 / pad 1 / stride 1, Wout % 32 == 0; every epilogue / prologue option, both patch heights, tails
    "pr8_c64_to_128": dict(N=2, H=16, W=32, Cin=64, Cout=128),
    "pr8_pro_res_stats": dict(N=1, H=24, W=64, Cin=96, Cout=256, pro=True, res=True, stats=True, relu=False),
    "pr4_tail_rows": dict(N=2, H=10, W=32, Cin=32, Cout=130, relu=True),              # H % 8 != 0 -> 4-row patches, last patch half empty
    "pr8_ups": dict(N=1, H=8, W=16, Cin=128, Cout=64, ups=1, stats=True),           # BN = 64, fused nearest x2
    "pr8_bn64_acc_alpha": dict(N=3, H=8, W=32, Cin=96, Cout=50, acc=True, alpha=0.37, relu=False, bias=False),
    "pr8_affine_c32": dict(N=1, H=8, W=32, Cin=32, Cout=96, oaff=True),             # two 16-channel chunks (chunked mode: Cin % 32 == 0)
    "pr8_wide_ld": dict(N=1, H=16, W=96, Cin=64, Cout=128, ldx_extra=8, ldy_extra=12),
    "bn256_c192": dict(N=1, H=8, W=64, Cin=64, Cout=192, res=True, bn192=0),        # 256-wide workgroup tile, 64 padding columns
    "bn256_c512_stats": dict(N=1, H=8, W=32, Cin=32, Cout=512, stats=True, relu=False, bias=False),
    "bn192_c160_acc": dict(N=1, H=16, W=64, Cin=128, Cout=160, acc=True, relu=False, bias=False, mask=True),     # 192-wide tile, 32 padding columns
    "bn192_c192_res": dict(N=2, H=8, W=32, Cin=64, Cout=192, res=True, stats=True),
    # too few 128-wide tiles for the chip, enough 64-wide ones (the rule is stated in workgroups: min_tiles scales it down to test size)
    "fill64_c256_pro": dict(N=1, H=16, W=64, Cin=64, Cout=256, pro=True, stats=True, min_tiles=5),
    "fill64_c126_acc": dict(N=1, H=16, W=64, Cin=96, Cout=126, acc=True, relu=False, min_tiles=3),
    "auto_256_to_128": dict(N=4, H=128, W=128, Cin=256, Cout=128, min_tiles=128),   # chosen by the default heuristic
    # data-gradient launches carrying the ReLU backward of the tensor they write (mrfa_conv_params.mask)
    "mask_acc_c160": dict(N=1, H=16, W=32, Cin=128, Cout=160, mask=True, acc=True, relu=False, bias=False),
    "mask_c192_pr4": dict(N=2, H=12, W=32, Cin=128, Cout=192, mask=True, relu=False, bias=False),
    "mask_ragged_c98": dict(N=1, H=8, W=64, Cin=32, Cout=98, mask=True, acc=True, relu=False, bias=False),
    # phase form of the fused upsample (four 2x2 convolutions on the low-resolution grid, pre-summed weights of pack mode 12)
    "phase_up_c64": dict(N=2, H=8, W=32, Cin=128, Cout=64, ups=1, stats=True, wphase=True),
    "phase_up_c256_pro_res": dict(N=1, H=16, W=32, Cin=64, Cout=256, ups=1, pro=True, res=True, relu=False, wphase=True),
    "phase_up_c130_acc": dict(N=1, H=8, W=64, Cin=32, Cout=130, ups=1, acc=True, alpha=0.5, bias=False, wphase=True),
    # v7: statistic groups (a patch lies inside one image): layer1's 64 -> 64 @64^2 of the keypoint encoder with the finalize behind the launch
    "groups2_layer1_fin": dict(N=4, H=32, W=32, Cin=64, Cout=64, stats=True, relu=False, bias=False, fin=True, groups=2),
    "decoder_down_fin": dict(N=2, H=32, W=32, Cin=64, Cout=128, stats=True, relu=False, fin=True),          # (the finalize by the launch's last workgroup)
    "decoder_ups_fin_c192": dict(N=1, H=16, W=16, Cin=128, Cout=192, ups=1, stats=True, relu=False, fin=True),
    "groups3_c128_res": dict(N=3, H=16, W=32, Cin=32, Cout=128, res=True, stats=True, groups=3),
}


@pytest.mark.parametrize("name", ["auto_256_to_128", "phase_up_c64", "mask_acc_c160", "pr4_tail_rows"])
def test_patch_tiled_kernel_is_run_to_run_identical(name):
    """no atomics on the output path: eight launches on the same operands must agree bit for bit (a missing barrier between the grouped
    weight-slab buffers / the halo double buffer of conv_halo.hip would show up here as a sporadic difference)"""
    L = hip.lib()
    kw = dict(HALO_CASES[name])
    kw.pop("min_tiles", None)
    kw.pop("bn192", None)
    kw["stats"] = False
    assert L.mrfa_set_mfma_mode(1) == 0
    prev = L.mrfa_set_tuning(b"conv_halo_min_tiles", 0)
    L.mrfa_set_tuning(b"conv_small", 0); L.mrfa_set_tuning(b"conv_lean", 0)
    try:
        outs = [conv_case(Side(True), tag=f"halo/{name}", wsplit=True, **kw)[0] for _ in range(8)]
        assert L.mrfa_conv2d_last_config() & (1 << 28), "the patch-tiled kernel did not run"
    finally:
        L.mrfa_set_mfma_mode(0)
        L.mrfa_set_tuning(b"conv_halo_min_tiles", prev)
        L.mrfa_set_tuning(b"conv_small", 1); L.mrfa_set_tuning(b"conv_lean", 1)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("name", list(HALO_CASES))
def test_conv2d_patch_tiled_kernel(name, mode):
    """conv_halo.hip (2-D output patches, the input halo split once per 16-channel chunk) against the CPU specification, in the
    six-product (fp32-accurate), the three-product and the plain-bf16 mode (one rounded weight plane, pack mode 14; phase forms: modes 1 / 2)"""
    L = hip.lib()
    kw = dict(HALO_CASES[name])
    if mode == 3 and kw.get("wphase"):
        pytest.skip("the phase form of the fused upsample exists for the split-operand modes")
    min_tiles = kw.pop("min_tiles", 0)              # 0: every eligible shape, whatever its workgroup count
    bn192 = kw.pop("bn192", 1)
    ref = conv_case(Side(False), tag=f"halo/{name}", **kw)
    assert L.mrfa_set_mfma_mode(mode) == 0
    prev = L.mrfa_set_tuning(b"conv_halo_min_tiles", min_tiles)
    L.mrfa_set_tuning(b"conv_small", 0); L.mrfa_set_tuning(b"conv_lean", 0)                     # (the small test shapes would otherwise go to conv_small.hip)
    L.mrfa_set_tuning(b"conv_halo_bn192", bn192)
    try:
        got = conv_case(Side(True), tag=f"halo/{name}", wsplit=("rne" if mode == 3 else True), **kw)
        assert L.mrfa_conv2d_last_config() & (1 << 28), "the patch-tiled kernel did not run"
    finally:
        L.mrfa_set_mfma_mode(0)
        L.mrfa_set_tuning(b"conv_halo_min_tiles", prev)
        L.mrfa_set_tuning(b"conv_small", 1); L.mrfa_set_tuning(b"conv_lean", 1)
        L.mrfa_set_tuning(b"conv_halo_bn192", 1)
    # mode 3: both operands rounded to 8 significand bits (2^-9 each), K = 288 .. 2304 products per output
    assert_close(ref, got, tol={1: 2e-4, 2: 2e-3, 3: 2e-2}[mode], what="halo " + name)


LEAN_CASES = {          # conv_lean.hip: the keypoint encoder's <= 128-channel 3x3 layers; every geometry (geo = row of its table), every epilogue / prologue option, tails
    # 32 channels: 8 x 32 patches with two pixel tiles per wave (geo 0), 4 x 32 patches (geo 3)
    "g0_hr32_stats_fin": dict(geo=0, N=2, H=16, W=64, Cin=32, Cout=32, stats=True, relu=False, bias=False, fin=True),
    "g0_pro_res_relu_tail": dict(geo=0, N=1, H=12, W=32, Cin=32, Cout=32, pro=True, res=True, relu=True, stats=True),     # H % 8 != 0: the last patch is half empty
    "g0_bst_groups2_c64": dict(geo=0, N=4, H=8, W=32, Cin=32, Cout=64, stats=True, relu=False, bias=False, bst=True, groups=2),
    "g3_hr32_stats_fin": dict(geo=3, N=2, H=16, W=64, Cin=32, Cout=32, stats=True, relu=False, bias=False, fin=True),
    "g3_hr32_pro_res_relu": dict(geo=3, N=1, H=8, W=32, Cin=32, Cout=32, pro=True, res=True, relu=True, stats=True),
    "g3_tail_rows_c64": dict(geo=3, N=2, H=10, W=32, Cin=32, Cout=64, relu=True),                  # H % 4 != 0; two channel tiles
    "g3_acc_alpha_affine": dict(geo=3, N=1, H=4, W=64, Cin=32, Cout=32, acc=True, alpha=0.37, relu=False, oaff=True),
    "g3_bst_groups2": dict(geo=3, N=4, H=8, W=32, Cin=32, Cout=32, stats=True, relu=False, bias=False, bst=True, groups=2),
    "g3_wide_ld": dict(geo=3, N=1, H=8, W=32, Cin=32, Cout=32, ldx_extra=8, ldy_extra=12, stats=True),
    # 64 channels: 2 x 32 patches x 64 output channels in two input-channel slices, two tiles per wave (geo 1); one tile per wave, no slices (geo 4);
    # 1 x 32 patches in two slices (geo 5)
    "g1_hr64_stats_fin_groups2": dict(geo=1, N=4, H=8, W=32, Cin=64, Cout=64, stats=True, relu=False, bias=False, fin=True, groups=2),
    "g1_pro_res_tail": dict(geo=1, N=2, H=5, W=32, Cin=64, Cout=64, pro=True, res=True, relu=True),
    "g1_bst_acc_c96": dict(geo=1, N=2, H=8, W=32, Cin=64, Cout=96, stats=True, relu=False, bias=False, acc=True, bst=True),
    "g4_hr64_stats_fin_groups2": dict(geo=4, N=4, H=8, W=32, Cin=64, Cout=64, stats=True, relu=False, bias=False, fin=True, groups=2),
    "g4_hr64_pro_res": dict(geo=4, N=2, H=6, W=32, Cin=64, Cout=64, pro=True, res=True, relu=True),
    "g4_bst_acc": dict(geo=4, N=2, H=8, W=32, Cin=64, Cout=64, stats=True, relu=False, bias=False, acc=True, bst=True),
    "g4_c96_out": dict(geo=4, N=1, H=8, W=32, Cin=64, Cout=96, stats=True),                          # three channel tiles over two 64-wide workgroup tiles
    "g5_hr64_slices": dict(geo=5, N=1, H=8, W=32, Cin=64, Cout=64, stats=True, relu=False, bias=False, fin=True),
    "g5_bst_linear_tail": dict(geo=5, N=1, H=5, W=32, Cin=64, Cout=64, stats=True, relu=False, bias=False, bst="linear"),
    # 128 channels on 16-wide patches: 4 x 16 x 32 output channels in four slices, two tiles per wave (geo 2); 2 x 16 x 64 in two slices (geo 6);
    # 2 x 16 x 32 in four slices (geo 7)
    "g2_hr128_stats_fin": dict(geo=2, N=2, H=16, W=16, Cin=128, Cout=128, stats=True, relu=False, bias=False, fin=True),
    "g2_pro_res_acc_tail": dict(geo=2, N=1, H=6, W=16, Cin=128, Cout=128, pro=True, res=True, acc=True, relu=False),
    "g2_bst_groups3_w32": dict(geo=2, N=3, H=4, W=32, Cin=128, Cout=64, stats=True, relu=False, bias=False, bst=True, groups=3),
    "g6_hr128_stats_fin": dict(geo=6, N=2, H=16, W=16, Cin=128, Cout=128, stats=True, relu=False, bias=False, fin=True),
    "g6_hr128_pro_res_acc": dict(geo=6, N=1, H=16, W=16, Cin=128, Cout=128, pro=True, res=True, acc=True, relu=False),
    "g6_bst_groups3": dict(geo=6, N=3, H=8, W=16, Cin=128, Cout=128, stats=True, relu=False, bias=False, bst=True, groups=3),
    "g6_tail_rows": dict(geo=6, N=2, H=7, W=16, Cin=128, Cout=128, stats=True, relu=True),           # H % 2 != 0
    "g7_hr128_slices": dict(geo=7, N=1, H=16, W=16, Cin=128, Cout=128, stats=True, relu=True),
    "g7_bst_c64": dict(geo=7, N=1, H=8, W=16, Cin=128, Cout=64, stats=True, relu=False, bias=False, bst=True),
}


def _lean_run(name, mode, **over):
    """one conv_lean.hip launch of LEAN_CASES[name] in matrix mode `mode`, the geometry forced"""
    L = hip.lib()
    kw = dict(LEAN_CASES[name])
    kw.update(over)
    geo = kw.pop("geo")
    assert L.mrfa_set_mfma_mode(mode) == 0
    L.mrfa_set_tuning(b"conv_lean_geo", geo)
    try:
        got = conv_case(Side(True), tag=f"lean/{name}", wsplit=("rne" if mode == 3 else True), **kw)
        assert L.mrfa_conv2d_last_config() & (1 << 27), "the lean patch kernel did not run"
    finally:
        L.mrfa_set_mfma_mode(0)
        L.mrfa_set_tuning(b"conv_lean_geo", -1)
    return got


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("name", list(LEAN_CASES))
def test_conv2d_lean_kernel(name, mode):
    """conv_lean.hip (four-wave patches, the halo of all input channels split once into LDS, weight fragments straight from the pre-split planes) against the
    CPU specification in the six-product (fp32-accurate), the three-product and the plain-bf16 mode"""
    kw = dict(LEAN_CASES[name])
    kw.pop("geo")
    ref = conv_case(Side(False), tag=f"lean/{name}", **kw)
    got = _lean_run(name, mode)
    assert_close(ref, got, tol={1: 2e-4, 2: 2e-3, 3: 2e-2}[mode], what="lean " + name)


GEMM_LEAN_CASES = {     # gemm_lean_kernel (conv_lean.hip): 1x1 convolutions / linears, K pipelined; both workgroup shapes, every epilogue option, ragged rows
    "linear_192_576_res": dict(N=16, H=1, W=276, Cin=192, Cout=576, R=1, pad=0, res=True, relu=False),                 # 64 x 128 tiles, 4.5 column tiles
    "linear_576_192_affine": dict(N=16, H=1, W=276, Cin=576, Cout=192, R=1, pad=0, oaff=True),                        # 64 x 64 tiles, two input-channel slices
    "linear_192_192_tail": dict(N=1, H=1, W=276, Cin=192, Cout=192, R=1, pad=0, relu=False),                          # M = 276: the last 64-row tile holds 20 rows
    "fuse_64_32_stats_fin_groups2": dict(N=4, H=32, W=32, Cin=64, Cout=32, R=1, pad=0, stats=True, relu=False, bias=False, fin=True, groups=2),
    "fuse_128_64_bst": dict(N=4, H=16, W=16, Cin=128, Cout=64, R=1, pad=0, stats=True, relu=False, bias=False, bst=True),
    "bottleneck_64_256_stats": dict(N=2, H=64, W=64, Cin=64, Cout=256, R=1, pad=0, stats=True, relu=False, bias=False, fin=True),
    "bottleneck_256_64_acc": dict(N=1, H=64, W=64, Cin=256, Cout=64, R=1, pad=0, acc=True, alpha=0.5, relu=False, bias=False),
    "k32_c96": dict(N=2, H=16, W=16, Cin=32, Cout=96, R=1, pad=0, stats=True),                                           # one stage only
}


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("name", list(GEMM_LEAN_CASES))
def test_conv2d_gemm_lean_kernel(name, mode):
    """gemm_lean_kernel (1x1 convolutions / linears of the keypoint encoder: 64-row workgroup tiles, the K axis pipelined through two LDS stage buffers,
    weight fragments straight from the pre-split planes) against the CPU specification in the three split / bf16 matrix modes"""
    L = hip.lib()
    kw = dict(GEMM_LEAN_CASES[name])
    ref = conv_case(Side(False), tag=f"glean/{name}", **kw)
    assert L.mrfa_set_mfma_mode(mode) == 0
    prev = L.mrfa_set_tuning(b"gemm_lean", 2)          # 2: wherever it can run (1, the default, keeps it to the sizes it is faster at)
    try:
        got = conv_case(Side(True), tag=f"glean/{name}", wsplit=("rne" if mode == 3 else True), **kw)
        assert L.mrfa_conv2d_last_config() & (1 << 26), "the lean 1x1 kernel did not run"
    finally:
        L.mrfa_set_tuning(b"gemm_lean", prev)
        L.mrfa_set_mfma_mode(0)
    assert_close(ref, got, tol={1: 2e-4, 2: 2e-3, 3: 2e-2}[mode], what="gemm lean " + name)


def test_lean_kernel_is_run_to_run_identical_and_equals_the_one_wave_kernel():
    """no atomics on the output path (the input-channel slices meet in LDS in a fixed order): eight launches agree bit for bit; and the result is the
    fp32-pipe kernel's (conv_small.hip, exact fmaf chains) to fp32 rounding of the sums"""
    L = hip.lib()
    for name in ["g0_pro_res_relu_tail", "g1_pro_res_tail", "g2_pro_res_acc_tail", "g5_hr64_slices", "g6_hr128_pro_res_acc", "g7_hr128_slices"]:
        outs = [_lean_run(name, 1, stats=False, fin=False)[0] for _ in range(8)]
        for o in outs[1:]:
            assert torch.equal(o, outs[0]), name
        if LEAN_CASES[name].get("pro"):
            continue                                     # (conv_small.hip has no prologue)
        kw = dict(LEAN_CASES[name])
        kw.pop("geo")
        kw.update(stats=False, fin=False)
        L.mrfa_set_tuning(b"conv_lean", 0)
        L.mrfa_set_tuning(b"conv_halo", 0)
        try:
            small = conv_case(Side(True), tag=f"lean/{name}", **kw)[0]
            assert L.mrfa_conv2d_last_config() & 8, "not the one-wave-per-tile kernel"
        finally:
            L.mrfa_set_tuning(b"conv_lean", 1)
            L.mrfa_set_tuning(b"conv_halo", 1)
        assert_close([small], [outs[0]], tol=2e-5, what="lean vs fp32 pipe " + name)


@pytest.mark.parametrize("cfg", [dict(N=2, H=8, W=32, Cin=64, Cout=128), dict(N=1, H=16, W=32, Cin=128, Cout=64, acc=True),
                                 dict(N=1, H=8, W=64, Cin=32, Cout=96)])
def test_phase_data_gradient_of_fused_upsample(cfg):
    """mrfa_conv2d_nhwc with ups = 2 (conv_halo.hip MODE 2): the data gradient of nearest-x2 + 3x3 as four transposed 2x2 convolutions of
    the phase images of dY, against the specification (3x3 data gradient on the 2H x 2W grid, 2x2 sum-pooled) and against autograd"""
    L = hip.lib()
    N, H, W, Cin, Cout, acc = cfg["N"], cfg["H"], cfg["W"], cfg["Cin"], cfg["Cout"], cfg.get("acc", False)
    tag = "phd/" + "_".join(f"{k}{v}" for k, v in cfg.items())

    def run(side):
        w = side.t(f"{tag}/w", (Cout, Cin, 3, 3), -0.2, 0.2)
        dy = side.t(f"{tag}/dy", (N * 2 * H * 2 * W, Cout + 4))
        dx = side.t(f"{tag}/dx0", (N * H * W, Cin + 4)) if acc else side.garbage((N * H * W, Cin + 4))
        cip = (Cin + 127) // 128 * 128
        wd = pack(side, w, 2)
        wpb = torch.zeros(3 * 16 * cip * Cout, dtype=torch.int16, device=side.dev)
        d = hip.PackDesc()
        d.src, d.Cout, d.Cin, d.R, d.S, d.ndst = w.data_ptr(), Cout, Cin, 3, 3, 1
        d.dst[0], d.mode[0] = wpb.data_ptr(), 13
        side.call("mrfa_pack_conv_weights_multi", C.pointer(d), 1)
        q = hip.ConvParams()
        q.x, q.ldx, q.Hin, q.Win, q.ups, q.N, q.Cin = dy.data_ptr(), Cout + 4, 2 * H, 2 * W, 2, N, Cout
        q.w, q.w_ld, q.w_tap, q.kflat, q.w_rows = wd.data_ptr(), Cout, cip * Cout, 0, cip
        q.w_phase, q.w_phase_piece = wpb.data_ptr(), 16 * cip * Cout
        q.y, q.ldy, q.Cout, q.Hout, q.Wout = dx.data_ptr(), Cin + 4, Cin, H, W
        q.R, q.S, q.pad, q.alpha, q.nbatch, q.accumulate = 3, 3, 1, 1.0, 1, int(acc)
        assert side.L.mrfa_conv2d_phase_dgrad_supported(C.byref(q)) == 1
        side.call("mrfa_conv2d_nhwc", C.byref(q))
        return side.done(dx[:, :Cin], dy[:, :Cout], w)
    ref = run(Side(False))
    assert L.mrfa_set_mfma_mode(1) == 0
    try:
        got = run(Side(True))
        assert L.mrfa_conv2d_last_config() & (1 << 28)
    finally:
        L.mrfa_set_mfma_mode(0)
    assert_close(ref[:1], got[:1], what=tag)
    if not acc:                                     # and the specification itself against autograd through the upsample
        x = torch.zeros(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
        y = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x, scale_factor=2), ref[2].double(), padding=1)
        (gx,) = torch.autograd.grad(y, x, ref[1].reshape(N, 2 * H, 2 * W, Cout).permute(0, 3, 1, 2).double())
        assert_close([gx.permute(0, 2, 3, 1).reshape(-1, Cin)], got[:1], what=tag + " vs autograd")


def test_patch_tiled_kernel_equals_the_row_tiled_kernel_and_fp64():
    """same arithmetic as conv_split.hip (exact three-way split, six products, fp32 accumulate): against fp64 the two kernels must be
    equally accurate on wide-dynamic-range operands"""
    L = hip.lib()
    N, H, W, Cin, Cout = 1, 32, 32, 256, 128
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(N, H, W, Cin, generator=g, dtype=torch.float64) * torch.exp(2.0 * torch.randn(N, H, W, Cin, generator=g, dtype=torch.float64))).float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64) * 0.05).float()
    exact = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    side = Side(True)
    xd, wd = x.reshape(-1, Cin).to(side.dev).contiguous(), w.to(side.dev)
    wp = pack(side, wd, 0)
    piece = 9 * 128 * Cin
    wsb = torch.zeros(3 * piece, dtype=torch.int16, device=side.dev)
    d = hip.PackDesc()
    d.src, d.Cout, d.Cin, d.R, d.S, d.ndst = wd.data_ptr(), Cout, Cin, 3, 3, 1
    d.dst[0], d.mode[0] = wsb.data_ptr(), 8
    side.call("mrfa_pack_conv_weights_multi", C.pointer(d), 1)
    errs = {}
    assert L.mrfa_set_mfma_mode(1) == 0
    prev = L.mrfa_set_tuning(b"conv_halo_min_tiles", 0)
    L.mrfa_set_tuning(b"conv_small", 0); L.mrfa_set_tuning(b"conv_lean", 0)
    try:
        for halo in (0, 1):
            L.mrfa_set_tuning(b"conv_halo", halo)
            y = side.garbage((N * H * W, Cout))
            p = hip.ConvParams()
            p.x, p.ldx, p.Hin, p.Win, p.ups, p.N, p.Cin = xd.data_ptr(), Cin, H, W, 0, N, Cin
            p.w, p.w_ld, p.w_tap, p.kflat, p.w_rows = wp.data_ptr(), Cin, 128 * Cin, 0, 128
            p.w_split, p.w_piece = wsb.data_ptr(), piece
            p.y, p.ldy, p.Cout, p.Hout, p.Wout = y.data_ptr(), Cout, Cout, H, W
            p.R, p.S, p.pad, p.alpha, p.nbatch, p.splitk = 3, 3, 1, 1.0, 1, 1
            side.call("mrfa_conv2d_nhwc", C.byref(p))
            assert bool(L.mrfa_conv2d_last_config() & (1 << 28)) == bool(halo)
            torch.cuda.synchronize()
            dlt = (y.double().cpu() - exact)
            errs[halo] = (float(dlt.abs().max()), float(dlt.pow(2).mean().sqrt()))
    finally:
        L.mrfa_set_mfma_mode(0)
        L.mrfa_set_tuning(b"conv_halo", 1)
        L.mrfa_set_tuning(b"conv_halo_min_tiles", prev)
        L.mrfa_set_tuning(b"conv_small", 1); L.mrfa_set_tuning(b"conv_lean", 1)
    scale = float(exact.abs().max())
    assert errs[1][0] <= max(2.0 * errs[0][0], 1e-6 * scale) and errs[1][1] <= max(2.0 * errs[0][1], 1e-7 * scale), (errs, scale)


def test_split_operand_mode_is_fp32_accurate():
    """error against an fp64 convolution: the bf16x6 kernel must be as accurate as the native fp32 MFMA kernel (K = 2304
    products per output, operands with a wide dynamic range so that all three bf16 pieces matter)"""
    L = hip.lib()
    N, H, W, Cin, Cout = 1, 24, 24, 256, 128
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(N, H, W, Cin, generator=g, dtype=torch.float64) * torch.exp(2.0 * torch.randn(N, H, W, Cin, generator=g, dtype=torch.float64))).float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g, dtype=torch.float64) * 0.05).float()
    exact = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    side = Side(True)
    xd, wd = x.reshape(-1, Cin).to(side.dev).contiguous(), w.to(side.dev)
    wp = pack(side, wd, 0)
    errs = {}
    for mode in (0, 1, 2, 3):
        y = side.garbage((N * H * W, Cout))
        p = hip.ConvParams()
        p.x, p.ldx, p.Hin, p.Win, p.ups, p.N, p.Cin = xd.data_ptr(), Cin, H, W, 0, N, Cin
        p.w, p.w_ld, p.w_tap, p.kflat, p.w_rows = wp.data_ptr(), Cin, 128 * Cin, 0, 128
        p.y, p.ldy, p.Cout, p.Hout, p.Wout = y.data_ptr(), Cout, Cout, H, W
        p.R, p.S, p.pad, p.alpha, p.nbatch, p.splitk, p.tile = 3, 3, 1, 1.0, 1, 1, (128 << 16) | 128
        assert L.mrfa_set_mfma_mode(mode) == 0
        try:
            side.call("mrfa_conv2d_nhwc", C.byref(p))
            assert bool(L.mrfa_conv2d_last_config() & 4) == bool(mode)
        finally:
            L.mrfa_set_mfma_mode(0)
        torch.cuda.synchronize()
        d = (y.double().cpu() - exact)
        errs[mode] = (float(d.abs().max()), float(d.pow(2).mean().sqrt()))
    scale = float(exact.abs().max())
    assert errs[1][0] <= max(2.0 * errs[0][0], 1e-6 * scale), (errs, scale)       # max error no worse than native fp32 (x2 slack)
    assert errs[1][1] <= max(2.0 * errs[0][1], 1e-7 * scale), (errs, scale)       # rms error likewise
    # mode 2 (bf16x3, opt-in): three dropped ~2^-16 cross terms per product -> 1e-5-class error relative to the output scale, i.e.
    # between fp32 (6e-8) and TF32 (5e-4); documented, not fp32-accurate
    assert errs[2][0] <= 1e-4 * scale and errs[2][1] <= 1e-5 * scale, (errs, scale)
    assert errs[2][1] >= 4.0 * errs[1][1], "bf16x3 unexpectedly as accurate as bf16x6: is it running the 6-product kernel?"
    # mode 3 (plain bf16 operands, round-to-nearest-even, one product): 2^-9 operand error, unbiased -> rms error ~ 2^-9 / sqrt(K)-ish
    # of the output scale; the bound also catches truncation instead of rounding (a one-sided 2^-8 error: ~3x this rms)
    assert errs[3][1] <= 2e-3 * scale and errs[3][0] <= 2e-2 * scale, (errs, scale)
    assert errs[3][1] >= 10.0 * errs[2][1]


def test_gemm_nt_batched():
    def run(side):
        B, M, Nn, K = 3, 100, 72, 64
        a = side.t("g/a", (B * M, K))
        bm = side.t("g/b", (B * Nn, K))
        c = side.garbage((B * M, Nn))
        p = hip.ConvParams()
        p.x, p.ldx, p.Hin, p.Win, p.N, p.Cin = a.data_ptr(), K, 1, M, 1, K
        p.w, p.w_ld, p.w_rows = bm.data_ptr(), K, Nn
        p.y, p.ldy, p.Cout, p.Hout, p.Wout = c.data_ptr(), Nn, Nn, 1, M
        p.R, p.S, p.pad, p.alpha = 1, 1, 0, 0.0625
        p.nbatch, p.x_bs, p.w_bs, p.y_bs, p.splitk = B, M * K, Nn * K, M * Nn, 1
        side.call("mrfa_conv2d_nhwc", C.byref(p))
        return side.done(c)
    ref, got = both(run)
    assert_close(ref, got, what="gemm_nt")
    a = det_uniform("g/a", (3, 100, 64)).double()
    b = det_uniform("g/b", (3, 72, 64)).double()
    assert_close([torch.einsum("bik,bjk->bij", a, b).reshape(300, 72) * 0.0625], got, what="gemm_nt vs einsum")


@pytest.mark.parametrize("mode", [0, 1])
def test_gemm_nt_batched_accumulating_split_k(mode):
    """the correlation volume's dq of the pooled levels (raft.py:185 backward: 64 query rows x 4 096 keys x 256 channels per sample): a batched NT GEMM
    that ACCUMULATES into the existing gradient and lets the library split its long K over workgroups (splitk = 0: atomics, no init / epilogue pass)"""
    def run(side):
        B, M, Nn, K = 4, 64, 256, 4096
        a = side.t("gs/a", (B * M, K))
        bm = side.t("gs/b", (B * Nn, K), -0.1, 0.1)
        c = side.t("gs/c0", (B * M, Nn))
        p = hip.ConvParams()
        p.x, p.ldx, p.Hin, p.Win, p.N, p.Cin = a.data_ptr(), K, 1, M, 1, K
        p.w, p.w_ld, p.w_rows = bm.data_ptr(), K, Nn
        p.y, p.ldy, p.Cout, p.Hout, p.Wout = c.data_ptr(), Nn, Nn, 1, M
        p.R, p.S, p.pad, p.alpha, p.accumulate = 1, 1, 0, 0.0625, 1
        p.nbatch, p.x_bs, p.w_bs, p.y_bs, p.splitk = B, M * K, Nn * K, M * Nn, 0
        side.call("mrfa_conv2d_nhwc", C.byref(p))
        if side.gpu:
            assert side.L.mrfa_conv2d_last_config() & 1, "the launch did not split K"
        return side.done(c)
    L = hip.lib()
    ref = run(Side(False))
    assert L.mrfa_set_mfma_mode(mode) == 0
    try:
        got = run(Side(True))
    finally:
        L.mrfa_set_mfma_mode(0)
    assert_close(ref, got, what="gemm_nt accumulate split-K")


def dgrad_case(side, *, N=2, H=8, W=9, Cin=64, Cout=96, R=3, pad=1, tag="d"):
    """data gradient = conv of dY with the flipped/transposed pack; checked against autograd in the test body"""
    w = side.t(f"{tag}/w", (Cout, Cin, R, R), -0.2, 0.2)
    Ho, Wo = H + 2 * pad - R + 1, W + 2 * pad - R + 1
    dy = side.t(f"{tag}/dy", (N * Ho * Wo, (Cout + 3) // 4 * 4))
    dx = side.garbage((N * H * W, (Cin + 3) // 4 * 4))
    flat = (Cout % 32) != 0
    p = hip.ConvParams()
    p.x, p.ldx, p.Hin, p.Win, p.N, p.Cin = dy.data_ptr(), dy.shape[1], Ho, Wo, N, Cout
    cip = (Cin + 127) // 128 * 128
    keep = []
    if flat:
        wp = pack(side, w, 3)
        kt = ktab(side, Cout, R, R, R - 1 - pad)
        keep.append(kt)
        p.w, p.w_ld, p.kflat, p.ktab = wp.data_ptr(), (R * R * Cout + 31) // 32 * 32, R * R * Cout, kt.data_ptr()
    else:
        wp = pack(side, w, 2)
        p.w, p.w_ld, p.w_tap = wp.data_ptr(), Cout, cip * Cout
    p.w_rows = cip
    p.y, p.ldy, p.Cout, p.Hout, p.Wout = dx.data_ptr(), dx.shape[1], Cin, H, W
    p.R, p.S, p.pad, p.alpha, p.nbatch, p.splitk = R, R, R - 1 - pad, 1.0, 1, 1
    side.call("mrfa_conv2d_nhwc", C.byref(p))
    return side.done(dx[:, :Cin])


@pytest.mark.parametrize("cfg", [dict(), dict(Cout=126), dict(Cout=3, Cin=64, R=7, pad=3), dict(Cout=10, Cin=35, R=7, pad=0),
                                 dict(Cin=3, Cout=64, R=7, pad=3), dict(R=1, pad=0, Cin=98, Cout=128),
                                 dict(N=4, H=32, W=32, Cin=64, Cout=64), dict(N=4, H=16, W=16, Cin=128, Cout=32)])
def test_dgrad_matches_autograd(cfg):
    tag = "dgrad/" + "_".join(f"{k}{v}" for k, v in cfg.items())
    (got,) = dgrad_case(Side(True), tag=tag, **cfg)
    c = dict(N=2, H=8, W=9, Cin=64, Cout=96, R=3, pad=1)
    c.update(cfg)
    w = det_uniform(f"{tag}/w", (c["Cout"], c["Cin"], c["R"], c["R"]), -0.2, 0.2).double()
    Ho, Wo = c["H"] + 2 * c["pad"] - c["R"] + 1, c["W"] + 2 * c["pad"] - c["R"] + 1
    dy = det_uniform(f"{tag}/dy", (c["N"] * Ho * Wo, (c["Cout"] + 3) // 4 * 4)).double()[:, :c["Cout"]]
    dy = dy.reshape(c["N"], Ho, Wo, c["Cout"]).permute(0, 3, 1, 2)
    x = torch.zeros(c["N"], c["Cin"], c["H"], c["W"], dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(x, w, padding=c["pad"])
    (gx,) = torch.autograd.grad(y, x, dy)
    assert_close([gx.permute(0, 2, 3, 1).reshape(-1, c["Cin"])], [got], what=tag)


def wgrad_case(side, *, N=2, H=10, W=9, Cin=64, Cout=96, R=3, pad=1, ups=0, pro=False, dbias=True, ksplit=0, dy_off=0, ws=False, stride=1,
               launch=True, groups=1, tag="w"):
    x = side.t(f"{tag}/x", (N * H * W, (Cin + 3) // 4 * 4))
    Hv, Wv = H << ups, W << ups
    Ho, Wo = (Hv + 2 * pad - R) // stride + 1, (Wv + 2 * pad - R) // stride + 1
    dy = side.t(f"{tag}/dy", (N * Ho * Wo, (Cout + dy_off + 3) // 4 * 4))
    dw = side.z((R * R * Cout * Cin,))
    db = side.z((Cout,))
    q = hip.WgradParams()
    q.x, q.ldx, q.Hin, q.Win, q.ups, q.N, q.Cin = x.data_ptr(), x.shape[1], H, W, ups, N, Cin
    keep = []
    if pro:                         # v9: one prologue vector pair per statistic group
        sc, sh = side.t(f"{tag}/sc", (groups * Cin,), 0.5, 1.5), side.t(f"{tag}/sh", (groups * Cin,), -0.3, 0.3)
        keep += [sc, sh]
        q.in_scale, q.in_shift, q.in_relu, q.groups = sc.data_ptr(), sh.data_ptr(), 1, groups
    q.dy, q.ldy, q.Cout, q.Hout, q.Wout = dy.data_ptr() + 4 * dy_off, dy.shape[1], Cout, Ho, Wo
    q.R, q.S, q.pad = R, R, pad
    q.dw = dw.data_ptr()
    q.dbias = db.data_ptr() if dbias else None
    q.alpha, q.nbatch, q.ksplit = 1.0, 1, ksplit
    if ws:
        wsb = side.garbage((4 << 20,))
        keep.append(wsb)
        q.ws, q.ws_bytes = wsb.data_ptr(), wsb.numel() * 4
    if Cin < 32:
        kt = ktab(side, Cin, R, R, pad)
        keep.append(kt)
        q.ktab, q.kflat = kt.data_ptr(), R * R * Cin
    if stride > 1:
        q.stride = stride
        assert side.L.mrfa_conv2d_wgrad_stride_supported(C.byref(q)) == 1
    if not launch:                  # the caller issues it (test_wgrad_multi): parameter block, outputs, everything that must stay alive
        return q, dw, db, keep + [x, dy]
    side.call("mrfa_conv2d_wgrad_nhwc", C.byref(q))
    return side.done(dw, db)


@pytest.mark.parametrize("cfg", [dict(), dict(Cout=126, Cin=160), dict(ups=1, H=5, W=6), dict(pro=True, Cin=128, Cout=64),
                                 dict(Cin=3, Cout=64, R=7, pad=3), dict(Cin=2, Cout=128, R=7, pad=3), dict(Cin=128, Cout=2),
                                 dict(Cin=64, Cout=3, R=7, pad=3), dict(Cin=35, Cout=10, R=7, pad=0), dict(Cin=13, Cout=64, pro=True),
                                 dict(R=1, pad=0, Cin=98, Cout=128), dict(N=4, H=32, W=32, Cin=64, Cout=64, ksplit=7),
                                 dict(Cin=128, Cout=1, dy_off=2), dict(Cin=128, Cout=64, dy_off=3),
                                 dict(N=2, H=32, W=32, Cin=192, Cout=128), dict(N=2, H=32, W=32, Cin=128, Cout=96),
                                 dict(N=2, H=32, W=32, Cin=160, Cout=126, pro=True), dict(N=1, H=32, W=64, Cin=64, Cout=128, ups=1),
                                 dict(N=4, H=64, W=64, Cin=98, Cout=128, R=1, pad=0, ksplit=40, ws=True),
                                 dict(N=2, H=64, W=64, Cin=2, Cout=128, R=7, pad=3, ksplit=32, ws=True),
                                 # strided layers (wgrad_small.hip's gather with stride 2)
                                 dict(N=2, H=32, W=32, Cin=64, Cout=64, stride=2, dbias=False), dict(N=1, H=13, W=17, Cin=32, Cout=128, stride=2, dbias=False),
                                 # wgrad_small.hip: one wave per 32 x 32 weight block of one tap (the MTIA prior's shapes)
                                 dict(N=2, H=16, W=16, Cin=32, Cout=32), dict(N=4, H=32, W=32, Cin=64, Cout=64),
                                 dict(N=2, H=16, W=16, Cin=128, Cout=128), dict(N=2, H=1, W=276, Cin=192, Cout=576, R=1, pad=0),
                                 dict(N=1, H=1, W=276, Cin=576, Cout=192, R=1, pad=0), dict(N=3, H=12, W=20, Cin=64, Cout=32),
                                 dict(N=1, H=5, W=7, Cin=32, Cout=96),
                                 # its ROWS forms (16-pixel chunks inside one image row: M % 16 == 0, power-of-two grids or same-size 1x1) with 32 x 32 and
                                 # 64 x 64 weight blocks, the strided gather among them
                                 dict(N=8, H=1, W=276, Cin=192, Cout=576, R=1, pad=0), dict(N=8, H=1, W=276, Cin=576, Cout=192, R=1, pad=0),
                                 dict(N=2, H=32, W=32, Cin=64, Cout=128), dict(N=2, H=16, W=32, Cin=32, Cout=64), dict(N=1, H=64, W=64, Cin=32, Cout=32),
                                 dict(N=2, H=32, W=32, Cin=64, Cout=128, stride=2, dbias=False), dict(N=1, H=16, W=16, Cin=192, Cout=64, dbias=False)])
def test_wgrad(cfg):
    tag = "wgrad/" + "_".join(f"{k}{v}" for k, v in cfg.items())
    ref, got = both(lambda s: wgrad_case(s, tag=tag, **cfg))
    assert_close(ref, got, tol=5e-4, what=tag)


def test_wgrad_multi():
    """mrfa_conv2d_wgrad_multi (v6): n weight gradients in as few launches as possible == n single calls.  The list mixes the three variants of the
    one-wave-per-block kernel (3x3 on power-of-two / other grids, same-size 1x1), a strided layer, two problems that SHARE their outputs (the two
    encoder passes of a training step add into the same dW / dbias), more problems than one launch holds (28), and two the small kernel does not
    take (ragged channels, a big layer): those must come out as if mrfa_conv2d_wgrad_nhwc had been called."""
    cfgs = [dict(N=2, H=16, W=16, Cin=32, Cout=32), dict(N=3, H=12, W=20, Cin=64, Cout=32), dict(N=2, H=1, W=276, Cin=192, Cout=576, R=1, pad=0),
            dict(N=2, H=32, W=32, Cin=64, Cout=64, stride=2, dbias=False), dict(Cout=126, Cin=160), dict(N=2, H=64, W=64, Cin=256, Cout=128)]
    cfgs += [dict(N=1, H=8, W=8, Cin=32 * (1 + i % 3), Cout=32 * (1 + i % 2)) for i in range(30)]

    def run(side):
        probs = [wgrad_case(side, tag=f"wmulti/{i}", launch=False, **c) for i, c in enumerate(cfgs)]
        # problem 1b: a second pass over other data into the outputs of problem 0
        q2, dw2, db2, keep2 = wgrad_case(side, tag="wmulti/0b", launch=False, **cfgs[0])
        q2.dw, q2.dbias = probs[0][0].dw, probs[0][0].dbias
        allq = [p[0] for p in probs] + [q2]
        arr = (hip.WgradParams * len(allq))()
        for i, q in enumerate(allq):
            C.memmove(C.byref(arr, i * C.sizeof(hip.WgradParams)), C.byref(q), C.sizeof(hip.WgradParams))
        side.call("mrfa_conv2d_wgrad_multi", arr, len(allq))
        return side.done(*[t for p in probs for t in (p[1], p[2])])
    ref, got = both(run)
    assert_close(ref, got, tol=5e-4, what="wgrad_multi")


WGRAD_LEAN_CASES = [       # wgrad_lean.hip: the residual blocks' 3x3 layers, all three geometries, prologue per statistic group, tails, shared outputs
    dict(N=2, H=16, W=64, Cin=32, Cout=32), dict(N=4, H=10, W=32, Cin=32, Cout=32, pro=True, groups=2),          # (H % 4 != 0: half-empty last patch)
    dict(N=2, H=8, W=32, Cin=64, Cout=64), dict(N=3, H=5, W=32, Cin=64, Cout=64, pro=True, groups=3),
    dict(N=2, H=16, W=16, Cin=128, Cout=128), dict(N=2, H=7, W=16, Cin=128, Cout=128, pro=True, groups=2),        # (H % 2 != 0)
    dict(N=1, H=8, W=32, Cin=64, Cout=128), dict(N=1, H=8, W=32, Cin=128, Cout=64, pro=True), dict(N=1, H=6, W=32, Cin=128, Cout=128),
]


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_wgrad_lean_multi(mode):
    """mrfa_conv2d_wgrad_multi: the residual blocks' 3x3 weight gradients on the all-taps kernel (wgrad_lean.hip) -- more problems than one launch holds (24),
    two that share their outputs, one the kernel does not take among them -- against the CPU specification; and a lone prologue problem through
    mrfa_conv2d_wgrad_nhwc"""
    L = hip.lib()
    cfgs = [dict(c, dbias=False) for c in WGRAD_LEAN_CASES]
    cfgs += [dict(N=1, H=8, W=32, Cin=32, Cout=32, dbias=False) for _ in range(26)] + [dict(N=2, H=16, W=16, Cin=32, Cout=32)]

    def run(side):
        probs = [wgrad_case(side, tag=f"wlean/{i}", launch=False, **c) for i, c in enumerate(cfgs)]
        q2, dw2, db2, keep2 = wgrad_case(side, tag="wlean/0b", launch=False, **cfgs[0])
        q2.dw = probs[0][0].dw                          # a second pass over other data into the output of problem 0
        allq = [p[0] for p in probs] + [q2]
        if side.gpu:
            took = [L.mrfa_conv2d_wgrad_lean_supported(C.byref(q)) for q in allq]
            assert all(took[:len(WGRAD_LEAN_CASES) + 26]) and not took[len(cfgs) - 1], took
        arr = (hip.WgradParams * len(allq))()
        for i, q in enumerate(allq):
            C.memmove(C.byref(arr, i * C.sizeof(hip.WgradParams)), C.byref(q), C.sizeof(hip.WgradParams))
        side.call("mrfa_conv2d_wgrad_multi", arr, len(allq))
        lone = wgrad_case(side, tag="wlean/lone", N=2, H=8, W=32, Cin=64, Cout=64, pro=True, groups=2, dbias=False)
        return side.done(*[p[1] for p in probs]) + lone[:1]
    ref = run(Side(False))
    assert L.mrfa_set_mfma_mode(mode) == 0
    try:
        got = run(Side(True))
    finally:
        L.mrfa_set_mfma_mode(0)
    assert_close(ref, got, tol={1: 5e-4, 2: 2e-3, 3: 2e-2}[mode], what="wgrad_lean_multi")


WGRAD_HALO_CASES = [       # wgrad_halo.hip: 3x3 / pad 1, Wout % 32 == 0, Cin % 32 == 0; both block shapes, prologue, upsample, ragged Cout, tails
    dict(N=2, H=16, W=32, Cin=64, Cout=128), dict(N=1, H=24, W=64, Cin=128, Cout=256, pro=True),
    dict(N=2, H=10, W=32, Cin=32, Cout=126), dict(N=1, H=8, W=16, Cin=128, Cout=64, ups=1),
    dict(N=3, H=8, W=32, Cin=160, Cout=50, dbias=False), dict(N=1, H=40, W=32, Cin=96, Cout=96, pro=True),
    dict(N=2, H=64, W=64, Cin=256, Cout=128, min_wgs=192),                      # chosen by the default heuristic, several segments per column
    # fused-upsample layers whose low-resolution grid is 32 pixels wide: phase form (four taps per phase workgroup, both block shapes)
    dict(N=1, H=8, W=32, Cin=128, Cout=64, ups=1), dict(N=2, H=16, W=32, Cin=64, Cout=130, ups=1), dict(N=1, H=24, W=64, Cin=32, Cout=128, ups=1, dbias=False),
]


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("cfg", WGRAD_HALO_CASES)
def test_wgrad_all_taps_kernel(cfg, mode):
    """wgrad_halo.hip (one staging of X / dY per row strip for all nine taps, transposing LDS reads) against the CPU specification"""
    L = hip.lib()
    cfg = dict(cfg)
    min_wgs = cfg.pop("min_wgs", 1)
    tag = "wgh/" + "_".join(f"{k}{v}" for k, v in cfg.items())
    ref = wgrad_case(Side(False), tag=tag, **cfg)
    assert L.mrfa_set_mfma_mode(mode) == 0
    prev = L.mrfa_set_tuning(b"wgrad_halo_min_wgs", min_wgs)
    L.mrfa_set_tuning(b"conv_small", 0); L.mrfa_set_tuning(b"conv_lean", 0)
    try:
        got = wgrad_case(Side(True), tag=tag, **cfg)
        plain = None
        if cfg.get("ups"):           # the phase form of a fused-upsample layer against the nine-tap form of the same kernel
            L.mrfa_set_tuning(b"wgrad_halo_phase", 0)
            plain = wgrad_case(Side(True), tag=tag, **cfg)
            L.mrfa_set_tuning(b"wgrad_halo_phase", 1)
        # the same call with the kernel off must give the same answer from the per-tap kernels (and proves the switch works)
        L.mrfa_set_tuning(b"wgrad_halo", 0)
        other = wgrad_case(Side(True), tag=tag, **cfg)
    finally:
        L.mrfa_set_mfma_mode(0)
        L.mrfa_set_tuning(b"wgrad_halo", 1)
        L.mrfa_set_tuning(b"wgrad_halo_phase", 1)
        L.mrfa_set_tuning(b"wgrad_halo_min_wgs", prev)
        L.mrfa_set_tuning(b"conv_small", 1); L.mrfa_set_tuning(b"conv_lean", 1)
    tol = {1: 5e-4, 2: 5e-3, 3: 3e-2}[mode]
    assert_close(ref, got, tol=tol, what=tag)
    assert_close(other, got, tol=tol, what=tag + " vs per-tap kernel")
    if plain is not None:
        assert_close(plain, got, tol=tol, what=tag + " phase form vs nine-tap form")
        if cfg["W"] % 32 == 0:
            assert any(float((g - o).abs().max()) > 0 for g, o in zip(got, plain)), "identical bits: did the phase form run at all?"
    assert any(float((g - o).abs().max()) > 0 for g, o in zip(got, other)), "identical bits: did the all-taps kernel run at all?"


@pytest.mark.parametrize("cfg", [dict(N=2, H=32, W=32, Cin=256, Cout=128), dict(N=2, H=32, W=64, Cin=128, Cout=256, pro=True),
                                 dict(N=1, H=32, W=64, Cin=128, Cout=126, ups=1), dict(N=2, H=32, W=32, Cin=128, Cout=128, R=1, pad=0, dbias=False),
                                 dict(N=4, H=64, W=64, Cin=128, Cout=128, ksplit=40, ws=True), dict(N=2, H=32, W=32, Cin=100, Cout=130),
                                 dict(N=2, H=32, W=64, Cin=64, Cout=128), dict(N=2, H=32, W=32, Cin=128, Cout=64, pro=True),
                                 dict(N=2, H=32, W=32, Cin=60, Cout=50), dict(N=1, H=32, W=64, Cin=128, Cout=64, ups=1),
                                 dict(N=8, H=16, W=16, Cin=512, Cout=512, pro=True), dict(N=8, H=8, W=8, Cin=256, Cout=128),
                                 dict(N=4, H=12, W=16, Cin=128, Cout=128, ksplit=3), dict(N=2, H=8, W=8, Cin=128, Cout=128, ups=1),
                                 # chunk-flat N axis (Cin not a multiple of 128, several taps): columns of one tile belong to different taps
                                 dict(N=2, H=32, W=32, Cin=192, Cout=128), dict(N=2, H=32, W=64, Cin=160, Cout=126, pro=True),
                                 dict(N=1, H=16, W=32, Cin=192, Cout=64, ups=1), dict(N=2, H=32, W=32, Cin=96, Cout=96, ksplit=5, ws=True),
                                 dict(N=2, H=16, W=16, Cin=36, Cout=64, R=5, pad=2)])
def test_wgrad_split_operand_mode(cfg):
    """mrfa_set_mfma_mode(1): the bf16x6 weight-gradient kernel (128 x 128 tiles, Wout % 32 == 0) against the CPU specification
    at the tolerance of the native fp32 MFMA kernel"""
    L = hip.lib()
    tag = "wsplit/" + "_".join(f"{k}{v}" for k, v in cfg.items())
    ref = wgrad_case(Side(False), tag=tag, **cfg)
    assert L.mrfa_set_mfma_mode(1) == 0
    try:
        got = wgrad_case(Side(True), tag=tag, **cfg)
    finally:
        L.mrfa_set_mfma_mode(0)
    assert_close(ref, got, tol=5e-4, what=tag)


def test_wgrad_split_operand_mode_is_fp32_accurate():
    """weight gradient against fp64 (K = 2 x 32 x 64 = 4096 pixels per output, wide dynamic range): the bf16x6 kernel is as accurate as
    the native fp32 MFMA kernel"""
    L = hip.lib()
    N, H, W, Cin, Cout = 2, 32, 64, 128, 128
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(N, H, W, Cin, generator=g, dtype=torch.float64) * torch.exp(2.0 * torch.randn(N, H, W, Cin, generator=g, dtype=torch.float64))).float()
    dy = (torch.randn(N, H, W, Cout, generator=g, dtype=torch.float64) * torch.exp(2.0 * torch.randn(N, H, W, Cout, generator=g, dtype=torch.float64))).float()
    xp = torch.nn.functional.pad(x.double().permute(0, 3, 1, 2), (1, 1, 1, 1))
    exact = torch.zeros(9, Cout, Cin, dtype=torch.float64)
    dyd = dy.double().reshape(-1, Cout)
    for r in range(3):
        for s_ in range(3):
            exact[r * 3 + s_] = dyd.t() @ xp[:, :, r:r + H, s_:s_ + W].permute(0, 2, 3, 1).reshape(-1, Cin)
    side = Side(True)
    xd, dyd32 = x.reshape(-1, Cin).to(side.dev).contiguous(), dy.reshape(-1, Cout).to(side.dev).contiguous()
    errs = {}
    for mode in (0, 1):
        dw = side.z((9 * Cout * Cin,))
        q = hip.WgradParams()
        q.x, q.ldx, q.Hin, q.Win, q.ups, q.N, q.Cin = xd.data_ptr(), Cin, H, W, 0, N, Cin
        q.dy, q.ldy, q.Cout, q.Hout, q.Wout = dyd32.data_ptr(), Cout, Cout, H, W
        q.R, q.S, q.pad, q.dw, q.alpha, q.nbatch, q.ksplit = 3, 3, 1, dw.data_ptr(), 1.0, 1, 1
        assert L.mrfa_set_mfma_mode(mode) == 0
        try:
            side.call("mrfa_conv2d_wgrad_nhwc", C.byref(q))
        finally:
            L.mrfa_set_mfma_mode(0)
        torch.cuda.synchronize()
        d = dw.double().cpu().view(9, Cout, Cin) - exact
        errs[mode] = (float(d.abs().max()), float(d.pow(2).mean().sqrt()))
    scale = float(exact.abs().max())
    assert errs[1][0] <= max(2.0 * errs[0][0], 1e-6 * scale), (errs, scale)
    assert errs[1][1] <= max(2.0 * errs[0][1], 1e-7 * scale), (errs, scale)


def test_gemm_tn_batched():
    def run(side):
        B, K, M, Nn = 2, 300, 96, 64
        a = side.t("tn/a", (B * K, M))
        bm = side.t("tn/b", (B * K, Nn))
        c = side.z((B * M * Nn,))
        q = hip.WgradParams()
        q.x, q.ldx, q.Hin, q.Win, q.N, q.Cin = bm.data_ptr(), Nn, 1, K, 1, Nn
        q.dy, q.ldy, q.Cout, q.Hout, q.Wout = a.data_ptr(), M, M, 1, K
        q.R, q.S, q.pad, q.dw, q.alpha = 1, 1, 0, c.data_ptr(), 0.5
        q.nbatch, q.x_bs, q.dy_bs, q.dw_bs = B, K * Nn, K * M, M * Nn
        side.call("mrfa_conv2d_wgrad_nhwc", C.byref(q))
        return side.done(c)
    ref, got = both(run)
    assert_close(ref, got, what="gemm_tn")
    a = det_uniform("tn/a", (2, 300, 96)).double()
    b = det_uniform("tn/b", (2, 300, 64)).double()
    assert_close([0.5 * torch.einsum("bkm,bkn->bmn", a, b).reshape(-1)], got, what="gemm_tn vs einsum")


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_pack_modes(mode):
    def run(side):
        Co, Ci, R = 70, 45, 3
        if mode == 4:
            g = side.t("pk/g", (R * R * Co * Ci,))
            d = side.t("pk/d", (Co, Ci, R, R))
            side.call("mrfa_pack_conv_weight", g.data_ptr(), d.data_ptr(), Co, Ci, R, R, 4)
            return side.done(d)
        return side.done(pack(side, side.t("pk/w", (Co, Ci, R, R)), mode))
    ref, got = both(run)
    assert_close(ref, got, tol=0, what=f"pack{mode}")


@pytest.mark.parametrize("Cc,ld,N", [(96, 100, 2), (37, 41, 2), (132, 132, 5)])
def test_bn_forward_backward(Cc, ld, N):
    """(96, 100) and (132, 132): float4 backward kernels (132 = 2 channel blocks + a 4-channel tail, 5*8*6 rows = ragged
    row blocks); (37, 41): scalar kernels"""
    def run(side):
        H, W = 8, 6
        x = side.t("bn/x", (N * H * W, ld), -2, 2)
        gamma, beta = side.t("bn/g", (Cc,), 0.5, 1.5), side.t("bn/b", (Cc,))
        rm, rv = side.t("bn/rm", (Cc,)), side.t("bn/rv", (Cc,), 0.5, 1.5)
        outs = []
        for train in (1, 0):
            st = side.z((hip.STATS_SLOTS * 2 * Cc,), torch.float64)
            side.call("mrfa_bn_stats", x.data_ptr(), ld, N * H * W, Cc, st.data_ptr())
            sc, sh, mean, inv = (side.z((Cc,)) for _ in range(4))
            side.call("mrfa_bn_finalize", st.data_ptr(), N * H * W, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                      0.1, 1e-5, Cc, train, sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), inv.data_ptr())
            for pool, blend in ((0, 0), (1, 0), (0, 1)):
                Ho, Wo = (H // 2, W // 2) if pool else (H, W)
                y = side.garbage((N * Ho * Wo, Cc + 4))
                a = side.t("bn/a", (N * Ho * Wo, Cc))
                occ = side.t("bn/occ", (N * Ho * Wo, 1), 0, 1)
                p = hip.BnActParams()
                p.x, p.ldx, p.N, p.H, p.W, p.C = x.data_ptr(), ld, N, H, W, Cc
                p.scale, p.shift, p.relu, p.pool = sc.data_ptr(), sh.data_ptr(), 1, pool
                if blend:
                    p.blend_a, p.lda, p.occ, p.ldo = a.data_ptr(), Cc, occ.data_ptr(), 1
                p.y, p.ldy = y.data_ptr(), Cc + 4
                side.call("mrfa_bn_act_fwd", C.byref(p))
                dy = side.t("bn/dy", (N * Ho * Wo, Cc))
                dx = side.t("bn/dx0", (N * H * W, Cc))
                da, docc = side.z((N * Ho * Wo, Cc)), side.z((N * Ho * Wo, 1))
                dg, dbt = side.z((Cc,)), side.z((Cc,))
                red = side.z((hip.STATS_SLOTS * 2 * Cc,), torch.float64)
                q = hip.BnBwdParams()
                q.x, q.ldx, q.N, q.H, q.W, q.C = x.data_ptr(), ld, N, H, W, Cc
                q.scale, q.shift, q.relu, q.pool = sc.data_ptr(), sh.data_ptr(), 1, pool
                q.mean, q.invstd, q.gamma = mean.data_ptr(), inv.data_ptr(), gamma.data_ptr()
                q.dy, q.lddy = dy.data_ptr(), Cc
                if blend:
                    q.blend_a, q.lda, q.occ, q.ldo = a.data_ptr(), Cc, occ.data_ptr(), 1
                    q.dblend_a, q.ldda, q.docc, q.lddo = da.data_ptr(), Cc, docc.data_ptr(), 1
                q.red, q.dx, q.lddx, q.dgamma, q.dbeta, q.train = red.data_ptr(), dx.data_ptr(), Cc, dg.data_ptr(), dbt.data_ptr(), train
                for ph in (1, 2):
                    q.phase = ph
                    side.call("mrfa_bn_act_bwd", C.byref(q))
                outs += [y[:, :Cc], dx, da, docc, dg, dbt]
                # sole-writer form: phase 2 overwrites an UNINITIALISED dx (no zero fill, no read)
                dx2 = side.garbage((N * H * W, Cc))
                q.dx, q.dx_overwrite, q.dgamma, q.dbeta, q.phase = dx2.data_ptr(), 1, None, None, 2
                side.call("mrfa_bn_act_bwd", C.byref(q))
                outs.append(dx2)
            outs += [rm.clone(), rv.clone(), sc, sh]
        return side.done(*outs)
    ref, got = both(run)
    assert_close(ref, got, tol=5e-4, what="bn")


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("Cc,in_rep", [(64, 1), (3, 1), (96, 1), (3, 11), (130, 1), (128, 1), (256, 2), (512, 1)])
def test_grid_sample(mode, Cc, in_rep):
    def run(side):
        Nin, Hi, Wi, Ho, Wo = 2, 9, 7, 6, 8
        N = Nin * in_rep
        ldi = (Cc + 3) // 4 * 4
        inp = side.t("gs/in", (Nin * Hi * Wi, ldi))
        if mode == 0:
            grid = side.t("gs/g0", (N * Ho * Wo, 2), -1.3, 1.3)
        else:
            grid = side.t("gs/g1", (N * Ho * Wo, 2), -5.0, 5.0)
        out = side.garbage((N * Ho * Wo, ldi))
        side.call("mrfa_grid_sample_fwd", inp.data_ptr(), ldi, Hi * Wi * ldi, in_rep, Hi, Wi, Cc, grid.data_ptr(), 2, N, Ho, Wo,
                  out.data_ptr(), ldi, mode)
        dout = side.t("gs/dout", (N * Ho * Wo, Cc))
        din = side.t("gs/din0", (Nin * Hi * Wi, ldi))
        dgrid = side.t("gs/dg0", (N * Ho * Wo, 2))
        side.call("mrfa_grid_sample_bwd", inp.data_ptr(), ldi, Hi * Wi * ldi, in_rep, Hi, Wi, Cc, grid.data_ptr(), 2, N, Ho, Wo,
                  dout.data_ptr(), Cc, mode, din.data_ptr(), ldi, Hi * Wi * ldi, dgrid.data_ptr(), 2)
        return side.done(out[:, :Cc], din[:, :Cc], dgrid)
    ref, got = both(run)
    assert_close(ref, got, tol=5e-4, what="grid_sample")


def test_timestamp_marks_are_ordered_on_a_stream():
    """mrfa_timestamp (the measurement aid behind tools/step_phases.py): three marks around two launches on one stream read back non-decreasing, in units of
    the device's 100 MHz clock, and the engine's Marks helper pairs them with their names"""
    from mrfa_amd import engine
    side = Side(True)
    buf = torch.zeros(3, dtype=torch.int64, device=side.dev)
    x = side.t("ts/x", (1 << 20, 4))
    y = side.garbage((1 << 20, 4))
    for k in range(3):
        side.call("mrfa_timestamp", buf.data_ptr() + 8 * k)
        if k < 2:
            side.call("mrfa_copy_view", x.data_ptr(), 4, 1 << 20, 4, y.data_ptr(), 4, 1.0, 0)
    torch.cuda.synchronize()
    t = buf.cpu().tolist()
    assert t[0] > 0 and t[0] <= t[1] <= t[2] and (t[2] - t[0]) < 100_000_000, t          # < 1 s between the first and the last
    engine.MARKS = engine.Marks(side.dev, 8)
    try:
        engine.MARKS.begin()
        engine.mark("a")
        engine.mark("b")
        torch.cuda.synchronize()
        r = engine.MARKS.read()
    finally:
        engine.MARKS = None
    assert [n for n, _ in r] == ["a", "b"] and r[0][1] == 0.0 and r[1][1] >= 0.0


def test_warp_frame_reflect():
    """mrfa_warp_frame_reflect == F.grid_sample(frame, grid, padding_mode='reflection', align_corners=False) (Transform.transform_frame, model.py:44-48): grids far outside
    [-1, 1] (several reflections), exactly on the edges, and a different output size"""
    def run(side):
        N, Cc, H, W, Ho, Wo = 2, 3, 20, 28, 16, 24
        x = side.t("wfr/x", (N, Cc, H, W))
        g = side.t("wfr/g", (N, Ho, Wo, 2), -3.7, 3.7)
        g[0, 0, :6, 0] = torch.tensor([-1.0, 1.0, -1.0 - 1.0 / W, 1.0 + 1.0 / W, 0.0, 3.0], device=side.dev)
        g[0, 0, :6, 1] = torch.tensor([1.0, -1.0, 1.0 + 1.0 / H, -1.0 - 1.0 / H, 0.0, -3.0], device=side.dev)
        out = side.garbage((N, Cc, Ho, Wo))
        side.call("mrfa_warp_frame_reflect", x.data_ptr(), N, Cc, H, W, g.data_ptr(), Ho, Wo, out.data_ptr())
        return side.done(out)
    ref, got = both(run)
    assert_close(ref, got, tol=1e-5, what="warp_frame_reflect")


@pytest.mark.parametrize("shape", [(8, 8, 16, 16, 2), (64, 64, 8, 8, 1), (8, 8, 13, 5, 98), (16, 16, 64, 64, 3)])
def test_resize(shape):
    Hi, Wi, Ho, Wo, Cc = shape

    def run(side):
        N = 2
        ld = (Cc + 3) // 4 * 4
        x = side.t("rs/x", (N * Hi * Wi, ld))
        y = side.t("rs/y0", (N * Ho * Wo, ld))
        side.call("mrfa_resize_bilinear_fwd", x.data_ptr(), ld, N, Hi, Wi, Cc, y.data_ptr(), ld, Ho, Wo, 2.0, 1)
        y2 = side.garbage((N * Ho * Wo, ld))
        side.call("mrfa_resize_bilinear_fwd", x.data_ptr(), ld, N, Hi, Wi, Cc, y2.data_ptr(), ld, Ho, Wo, 1.0, 0)
        dout = side.t("rs/do", (N * Ho * Wo, Cc))
        din = side.t("rs/di0", (N * Hi * Wi, ld))
        side.call("mrfa_resize_bilinear_bwd", dout.data_ptr(), Cc, N, Hi, Wi, Cc, din.data_ptr(), ld, Ho, Wo, 0.5)
        return side.done(y[:, :Cc], y2[:, :Cc], din[:, :Cc])
    ref, got = both(run)
    assert_close(ref, got, what="resize")


def test_resize_sum_multi():
    """v8: many "dst (=|+=) sum_k mul_k resize(src_k)" records in one launch (mrfa_resize_sum_multi / _bwd) against the ABI emulator AND against the separate
    mrfa_copy_view / mrfa_resize_bilinear_* launches they replace (same arithmetic, same order: equal to a few ulp -- the compiler contracts multiply-adds
    differently in the two forms): up- and down-sampling, same-size terms, three-term chains, overwriting and accumulating records, channel slices."""
    N = 2

    def build(side):
        def rec(dst, ldd, Hd, Wd, Cc, ow, terms):
            d = hip.ResizeSumDesc()
            d.dst, d.ldd, d.N, d.Hd, d.Wd, d.C, d.nterm, d.overwrite = dst, ldd, N, Hd, Wd, Cc, len(terms), ow
            for k, (src, lds, Hs, Ws, mul) in enumerate(terms):
                d.term[k].src, d.term[k].lds, d.term[k].Hs, d.term[k].Ws, d.term[k].mul = src, lds, Hs, Ws, mul
            return d
        a = side.t("rsm/a", (N * 8 * 8, 4))              # 3 channels of 4: slices [0:2] and [2:3]
        b = side.t("rsm/b", (N * 16 * 16, 2))
        c = side.t("rsm/c", (N * 8 * 8, 2))
        o1 = side.garbage((N * 16 * 16, 2))              # = 2 up(a[0:2]) + 0.5 b + 2 up(c)         (three-term chain, overwrite)
        o2 = side.t("rsm/o2", (N * 16 * 16, 4))          # [..., 1:2] += up(a[2:3])                  (accumulate into a channel slice)
        o3 = side.garbage((N * 4 * 4, 2))                # = down(b) / 8                             (down-sampling)
        o4 = side.garbage((N * 8 * 8, 2))                # = c + a[0:2]                              (same-size chain)
        fw = [rec(o1.data_ptr(), 2, 16, 16, 2, 1, [(a.data_ptr(), 4, 8, 8, 2.0), (b.data_ptr(), 2, 16, 16, 0.5), (c.data_ptr(), 2, 8, 8, 2.0)]),
              rec(o2.data_ptr() + 4, 4, 16, 16, 1, 0, [(a.data_ptr() + 8, 4, 8, 8, 1.0)]),
              rec(o3.data_ptr(), 2, 4, 4, 2, 1, [(b.data_ptr(), 2, 16, 16, 0.125)]),
              rec(o4.data_ptr(), 2, 8, 8, 2, 1, [(c.data_ptr(), 2, 8, 8, 1.0), (a.data_ptr(), 4, 8, 8, 1.0)])]
        return (a, b, c, o1, o2, o3, o4), fw, rec

    def run(side):
        (a, b, c, o1, o2, o3, o4), fw, rec = build(side)
        side.call("mrfa_resize_sum_multi", (hip.ResizeSumDesc * len(fw))(*fw), len(fw))
        # backward: gradients of a[0:2] (from o1 (x2, up-sampled) and o4 (same size)), a[2:3] (from o2), c (from o1 and o4)
        g1, g2, g4 = side.t("rsm/g1", (N * 16 * 16, 2)), side.t("rsm/g2", (N * 16 * 16, 4)), side.t("rsm/g4", (N * 8 * 8, 2))
        da, dc = side.t("rsm/da0", (N * 8 * 8, 4)), side.t("rsm/dc0", (N * 8 * 8, 2))
        bw = [rec(da.data_ptr(), 4, 8, 8, 2, 0, [(g4.data_ptr(), 2, 8, 8, 1.0), (g1.data_ptr(), 2, 16, 16, 2.0)]),
              rec(da.data_ptr() + 8, 4, 8, 8, 1, 0, [(g2.data_ptr() + 4, 4, 16, 16, 1.0)]),
              rec(dc.data_ptr(), 2, 8, 8, 2, 0, [(g4.data_ptr(), 2, 8, 8, 1.0), (g1.data_ptr(), 2, 16, 16, 2.0)])]
        side.call("mrfa_resize_sum_multi_bwd", (hip.ResizeSumDesc * len(bw))(*bw), len(bw))
        return side.done(o1, o2, o3, o4, da[:, :3], dc)

    def run_separately(side):
        (a, b, c, o1, o2, o3, o4), _, _ = build(side)
        side.call("mrfa_resize_bilinear_fwd", a.data_ptr(), 4, N, 8, 8, 2, o1.data_ptr(), 2, 16, 16, 2.0, 0)
        side.call("mrfa_copy_view", b.data_ptr(), 2, N * 256, 2, o1.data_ptr(), 2, 0.5, 1)
        side.call("mrfa_resize_bilinear_fwd", c.data_ptr(), 2, N, 8, 8, 2, o1.data_ptr(), 2, 16, 16, 2.0, 1)
        side.call("mrfa_resize_bilinear_fwd", a.data_ptr() + 8, 4, N, 8, 8, 1, o2.data_ptr() + 4, 4, 16, 16, 1.0, 1)
        side.call("mrfa_resize_bilinear_fwd", b.data_ptr(), 2, N, 16, 16, 2, o3.data_ptr(), 2, 4, 4, 0.125, 0)
        side.call("mrfa_copy_view", c.data_ptr(), 2, N * 64, 2, o4.data_ptr(), 2, 1.0, 0)
        side.call("mrfa_copy_view", a.data_ptr(), 4, N * 64, 2, o4.data_ptr(), 2, 1.0, 1)
        g1, g2, g4 = side.t("rsm/g1", (N * 16 * 16, 2)), side.t("rsm/g2", (N * 16 * 16, 4)), side.t("rsm/g4", (N * 8 * 8, 2))
        da, dc = side.t("rsm/da0", (N * 8 * 8, 4)), side.t("rsm/dc0", (N * 8 * 8, 2))
        side.call("mrfa_copy_view", g4.data_ptr(), 2, N * 64, 2, da.data_ptr(), 4, 1.0, 1)
        side.call("mrfa_resize_bilinear_bwd", g1.data_ptr(), 2, N, 8, 8, 2, da.data_ptr(), 4, 16, 16, 2.0)
        side.call("mrfa_resize_bilinear_bwd", g2.data_ptr() + 4, 4, N, 8, 8, 1, da.data_ptr() + 8, 4, 16, 16, 1.0)
        side.call("mrfa_copy_view", g4.data_ptr(), 2, N * 64, 2, dc.data_ptr(), 2, 1.0, 1)
        side.call("mrfa_resize_bilinear_bwd", g1.data_ptr(), 2, N, 8, 8, 2, dc.data_ptr(), 2, 16, 16, 2.0)
        return side.done(o1, o2, o3, o4, da[:, :3], dc)
    ref, got = both(run)
    assert_close(ref, got, tol=1e-5, what="resize_sum_multi")
    sep = run_separately(Side(True))
    for k, (x, y) in enumerate(zip(sep, got)):
        assert float((x - y).abs().max()) <= 2e-6 * max(1.0, float(x.abs().max())), (k, float((x - y).abs().max()))


def test_corr_lookup():
    def run(side):
        Q, Hs, Ws = 50, 16, 16
        v0 = side.t("cl/v0", (Q, Hs * Ws))
        v1 = side.t("cl/v1", (Q, Hs * Ws // 4))
        coords = side.t("cl/xy", (Q, 2), -4.0, 20.0)
        out = side.garbage((Q, 100))
        side.call("mrfa_corr_lookup_fwd", v0.data_ptr(), v1.data_ptr(), Hs, Ws, coords.data_ptr(), 2, Q, 3, out.data_ptr(), 100)
        dout = side.t("cl/do", (Q, 98))
        d0, d1, dc = side.z((Q, Hs * Ws)), side.z((Q, Hs * Ws // 4)), side.t("cl/dc0", (Q, 2))
        side.call("mrfa_corr_lookup_bwd", v0.data_ptr(), v1.data_ptr(), Hs, Ws, coords.data_ptr(), 2, Q, 3, dout.data_ptr(), 98,
                  d0.data_ptr(), d1.data_ptr(), dc.data_ptr(), 2)
        return side.done(out[:, :98], d0, d1, dc)
    ref, got = both(run)
    assert_close(ref, got, tol=5e-4, what="corr_lookup")


def test_layout_and_elementwise():
    def run(side):
        N, Cc, H, W = 2, 45, 6, 10
        outs = []
        src = side.t("el/nchw", (N, Cc, H, W))
        d = side.t("el/nhwc0", (N * H * W, 48))
        side.call("mrfa_nchw_to_nhwc", src.data_ptr(), d.data_ptr(), 48, N, Cc, H, W, 1)
        back = side.t("el/back0", (N, Cc, H, W))
        side.call("mrfa_nhwc_to_nchw", d.data_ptr(), 48, back.data_ptr(), N, Cc, H, W, 1)
        outs += [d[:, :Cc], back]
        y = side.garbage((N * (H // 2) * (W // 2), 48))
        side.call("mrfa_avgpool2_fwd", d.data_ptr(), 48, N, H, W, Cc, y.data_ptr(), 48)
        acc = side.t("el/acc", (N * (H // 2) * (W // 2), Cc))
        side.call("mrfa_sumpool2_acc", d.data_ptr(), 48, N, H // 2, W // 2, Cc, acc.data_ptr(), Cc, 0.5)
        up = side.t("el/up", (N * H * W, Cc))
        side.call("mrfa_unpool2_acc", y.data_ptr(), 48, N, H // 2, W // 2, Cc, up.data_ptr(), Cc, 0.25)
        outs += [y[:, :Cc], acc, up]
        rows = N * H * W
        bias = side.t("el/bias", (Cc,))
        for act in (0, 1, 2):
            o = side.garbage((rows, Cc))
            st = side.z((hip.STATS_SLOTS * 2 * Cc,), torch.float64)
            side.call("mrfa_bias_act", d.data_ptr(), 48, rows, Cc, bias.data_ptr(), act, o.data_ptr(), Cc, st.data_ptr())
            g = side.t("el/g", (rows, Cc))
            dx = side.t("el/dx0", (rows, Cc))
            side.call("mrfa_act_bwd", o.data_ptr(), Cc, g.data_ptr(), Cc, rows, Cc, act, dx.data_ptr(), Cc, 1)
            outs += [o, st.view(hip.STATS_SLOTS, 2 * Cc).sum(0), dx]
        a, b2, occ = side.t("el/a", (rows, Cc)), side.t("el/b2", (rows, Cc)), side.t("el/occ", (rows, 1), 0, 1)
        yb = side.garbage((rows, Cc))
        side.call("mrfa_blend_fwd", a.data_ptr(), Cc, b2.data_ptr(), Cc, occ.data_ptr(), 1, rows, Cc, yb.data_ptr(), Cc)
        yb2 = side.garbage((rows, Cc))
        side.call("mrfa_blend_fwd", a.data_ptr(), Cc, None, 0, occ.data_ptr(), 1, rows, Cc, yb2.data_ptr(), Cc)
        da, db, do = side.z((rows, Cc)), side.z((rows, Cc)), side.z((rows, 1))
        g = side.t("el/gb", (rows, Cc))
        side.call("mrfa_blend_bwd", a.data_ptr(), Cc, b2.data_ptr(), Cc, occ.data_ptr(), 1, g.data_ptr(), Cc, rows, Cc, da.data_ptr(), Cc,
                  db.data_ptr(), Cc, do.data_ptr(), 1)
        cs = side.z((Cc,))
        side.call("mrfa_colsum", a.data_ptr(), Cc, rows, Cc, cs.data_ptr())
        cp = side.t("el/cp0", (rows, Cc))
        side.call("mrfa_copy_view", a.data_ptr(), Cc, rows, Cc, cp.data_ptr(), Cc, 0.3, 1)
        outs += [yb, yb2, da, db, do, cs, cp]
        img = side.t("el/img", (2, 3, 32, 32), 0, 1)
        t = torch.arange(13, dtype=torch.float32)
        g1 = torch.exp(-((t - 6) ** 2) / (2 * 1.5 ** 2))
        ker = (g1[:, None] * g1[None, :] / (g1.sum() ** 2)).contiguous().to(side.dev)
        ya = side.garbage((2 * 8 * 8, 4))
        side.call("mrfa_antialias_down", img.data_ptr(), 2, 3, 32, 32, ker.data_ptr(), 13, 4, ya.data_ptr(), 4)
        outs.append(ya[:, :3])
        return side.done(*outs)
    ref, got = both(run)
    assert_close(ref, got, what="elementwise")


@pytest.mark.parametrize("cfg", [dict(Cin=64, Cout=3, R=7, pad=3), dict(Cin=128, Cout=2, R=3, pad=1), dict(Cin=128, Cout=1, R=3, pad=1),
                                 dict(Cin=108, Cout=1, R=7, pad=3, H=20, W=17), dict(Cin=128, Cout=2, R=7, pad=3, mode7=True),
                                 dict(Cin=36, Cout=4, R=7, pad=0, H=24, W=24), dict(Cin=128, Cout=2, R=3, pad=1, N=8, H=96, W=96),
                                 # conv_fewout3.hip (channels across the lanes): 16 / 32 / 64 lanes per pixel, ragged rows, tiny images
                                 dict(Cin=64, Cout=2, R=3, pad=1, H=9, W=21), dict(Cin=256, Cout=1, R=3, pad=1, N=1, H=8, W=8),
                                 dict(Cin=128, Cout=2, R=3, pad=1, N=3, H=5, W=4)])
def test_conv_fewout(cfg):
    """direct <=4-output-channel kernels (forward incl. accumulate, weight/bias gradient, pack modes 5/6/7)"""
    c = dict(N=2, H=33, W=40, mode7=False)
    c.update(cfg)
    tag = "fo/" + "_".join(f"{k}{v}" for k, v in cfg.items())

    def run(side):
        N, H, W, Cin, Cout, R, pad = c["N"], c["H"], c["W"], c["Cin"], c["Cout"], c["R"], c["pad"]
        T = R * R
        Ho, Wo = H + 2 * pad - R + 1, W + 2 * pad - R + 1
        x = side.t(f"{tag}/x", (N * H * W, Cin + 4))
        if c["mode7"]:      # weights of a conv with Cout(here)=its Cin: OIHW (Cin_here, Cout_here, R, R) packed flipped
            woihw = side.t(f"{tag}/w", (Cin, Cout, R, R), -0.2, 0.2)
            wp = side.garbage((Cout * T * Cin,))
            side.call("mrfa_pack_conv_weight", woihw.data_ptr(), wp.data_ptr(), Cin, Cout, R, R, 7)
        else:
            woihw = side.t(f"{tag}/w", (Cout, Cin, R, R), -0.2, 0.2)
            wp = side.garbage((Cout * T * Cin,))
            side.call("mrfa_pack_conv_weight", woihw.data_ptr(), wp.data_ptr(), Cout, Cin, R, R, 5)
        bias = side.t(f"{tag}/b", (Cout,))
        y = side.t(f"{tag}/y0", (N * Ho * Wo, 4))
        side.call("mrfa_conv_fewout_fwd", x.data_ptr(), Cin + 4, N, H, W, Cin, wp.data_ptr(), bias.data_ptr(), y.data_ptr(), 4, Cout, R, pad, 1)
        y2 = side.garbage((N * Ho * Wo, 4))
        side.call("mrfa_conv_fewout_fwd", x.data_ptr(), Cin + 4, N, H, W, Cin, wp.data_ptr(), None, y2.data_ptr(), 4, Cout, R, pad, 0)
        y3 = side.garbage((N * Ho * Wo, 4))               # bias without accumulation (the channel-split launch initialises y with it)
        side.call("mrfa_conv_fewout_fwd", x.data_ptr(), Cin + 4, N, H, W, Cin, wp.data_ptr(), bias.data_ptr(), y3.data_ptr(), 4, Cout, R, pad, 0)
        dy = side.t(f"{tag}/dy", (N * Ho * Wo, 4))
        dw = side.z((Cout * T * Cin,))
        db = side.z((Cout,))
        side.call("mrfa_conv_fewout_wgrad", x.data_ptr(), Cin + 4, N, H, W, Cin, dy.data_ptr(), 4, Cout, R, pad, dw.data_ptr(), db.data_ptr())
        g = side.t(f"{tag}/g0", (Cout, Cin, R, R))
        side.call("mrfa_pack_conv_weight", dw.data_ptr(), g.data_ptr(), Cout, Cin, R, R, 6)
        extra = []
        if not c["mode7"] and side.L.mrfa_conv_fewout_dgrad_supported(Cin, Cout, R, pad, W, Cin + 4):
            dx = side.garbage((N * H * W, Cin + 4))           # data gradient: overwrite, then accumulate on top
            side.call("mrfa_conv_fewout_dgrad", dy.data_ptr(), 4, N, H, W, Cout, wp.data_ptr(), dx.data_ptr(), Cin + 4, Cin, R, pad, 0, None, 0)
            dx2 = side.t(f"{tag}/dx0", (N * H * W, Cin + 4))
            side.call("mrfa_conv_fewout_dgrad", dy.data_ptr(), 4, N, H, W, Cout, wp.data_ptr(), dx2.data_ptr(), Cin + 4, Cin, R, pad, 1, None, 0)
            mk = side.t(f"{tag}/mask", (N * H * W, Cin + 8))  # ... and with the ReLU mask of the tensor the gradient belongs to
            dx3 = side.t(f"{tag}/dx0", (N * H * W, Cin + 4))
            side.call("mrfa_conv_fewout_dgrad", dy.data_ptr(), 4, N, H, W, Cout, wp.data_ptr(), dx3.data_ptr(), Cin + 4, Cin, R, pad, 1,
                      mk.data_ptr(), Cin + 8)
            extra = [dx[:, :Cin], dx2[:, :Cin], dx3[:, :Cin]]
        return side.done(y[:, :Cout], y2[:, :Cout], y3[:, :Cout], dw, db, g, wp, *extra)
    ref, got = both(run)
    assert_close(ref, got, tol=5e-4, what=tag)


@pytest.mark.parametrize("gmag,clip_slot", [(1.0, -1), (1.0, 0), (500.0, 0), (500.0, 2)])
def test_flat_clip_adam(gmag, clip_slot):
    """K20: mrfa_adam_prepare / mrfa_grad_absmax / mrfa_adam_flat over three steps on two parameter groups (odd sizes
    padded to 4), clipping inactive (|g|_inf < max_norm), active, and off; gscale = 1/8 as with 8 data-parallel ranks."""
    n0, n1 = 4 * 12345, 4 * 777

    def run(side):
        w = side.t("k20/w", (n0 + n1,))
        m, v = side.z((n0 + n1,)), side.z((n0 + n1,))
        state = side.z((2, 8))
        state[:, 3] = torch.tensor([2e-4, 1e-3], device=side.dev)
        outs = []
        for step in range(3):
            g = side.t(f"k20/g{step}", (n0 + n1,), -gmag, gmag)
            side.call("mrfa_adam_prepare", state.data_ptr(), 2, 0.5, 0.999)
            if clip_slot >= 0:
                side.call("mrfa_grad_absmax", g.data_ptr(), n0, state.data_ptr(), clip_slot)
            side.call("mrfa_adam_flat", w.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n0, state.data_ptr(), 0.5, 0.999, 1e-8,
                      0.125, clip_slot, 10.0)
            side.call("mrfa_adam_flat", w.data_ptr() + 4 * n0, g.data_ptr() + 4 * n0, m.data_ptr() + 4 * n0, v.data_ptr() + 4 * n0, n1,
                      state.data_ptr() + 32, 0.5, 0.999, 1e-8, 0.125, -1, 0.0)
            outs += [w.clone(), state.clone()]
        return side.done(*outs, m, v)
    ref, got = both(run)
    assert_close(ref, got, tol=2e-6, what="flat clip+adam")
    assert float(ref[1][0, 0]) == 1.0 and float(got[-3][0, 0]) == 3.0          # step counters
    if clip_slot >= 0:
        assert abs(float(got[1][0, 4 + clip_slot]) - float(ref[1][0, 4 + clip_slot])) == 0.0   # |g|_inf is exact


def test_pack_split_pieces():
    """pack modes 8 / 9: three bf16 pieces whose sum is the fp32 weight exactly, the mode 0 / 2 matrices in k16-chunk-major order"""
    shapes = [(130, 64, 3), (128, 100, 1), (96, 256, 3)]

    def run(side):
        outs = []
        for i, (Cout, Cin, R) in enumerate(shapes):
            T = R * R
            w = side.t(f"ps/w{i}", (Cout, Cin, R, R))
            n8 = T * ((Cout + 127) // 128 * 128) * ((Cin + 31) // 32 * 32)
            n9 = T * ((Cin + 127) // 128 * 128) * ((Cout + 31) // 32 * 32)
            b8 = torch.zeros(3 * n8, dtype=torch.int16, device=side.dev)
            b9 = torch.zeros(3 * n9, dtype=torch.int16, device=side.dev)
            d = hip.PackDesc()
            d.src, d.Cout, d.Cin, d.R, d.S, d.ndst = w.data_ptr(), Cout, Cin, R, R, 2
            d.dst[0], d.mode[0], d.dst[1], d.mode[1] = b8.data_ptr(), 8, b9.data_ptr(), 9
            side.call("mrfa_pack_conv_weights_multi", C.pointer(d), 1)
            for b, n, mode, rows, cols in ((b8, n8, 0, (Cout + 127) // 128 * 128, (Cin + 31) // 32 * 32),
                                           (b9, n9, 2, (Cin + 127) // 128 * 128, (Cout + 31) // 32 * 32)):
                pieces = (b.view(3, n).to(torch.int32) << 16).view(torch.float32)
                # the bf16 planes are k16-chunk-major: [tap][cols / 16][rows][16] (include/mrfa_hip.h, pack modes)
                w32 = pack(side, w, mode).view(T, rows, cols // 16, 16).permute(0, 2, 1, 3).reshape(-1)
                outs += [pieces[0], pieces[1], pieces[2], pieces.double().sum(0).float() - w32]
        return side.done(*outs)
    ref, got = both(run)
    for r, g in zip(ref, got):
        assert torch.equal(r, g)
    assert all(float(x.abs().max()) == 0.0 for x in got[3::4])          # piece sum == packed fp32 weight, exactly


def test_pack_and_unpack_multi():
    """batched (un)packing == the single-tensor entry points, layout by layout, on the path's awkward shapes
    (3-channel 7x7 stem, 98-channel 1x1, 126 outputs, 35-channel 7x7 head, 1024->512, few-output heads)."""
    shapes = [(64, 3, 7), (128, 98, 1), (126, 160, 3), (10, 35, 7), (512, 1024, 3), (2, 128, 3), (128, 2, 7), (96, 128, 3), (40, 36, 5)]

    def run(side):
        outs, keep = [], []
        descs = []
        for i, (Cout, Cin, R) in enumerate(shapes):
            T = R * R
            w = side.t(f"pm/w{i}", (Cout, Cin, R, R))
            sizes = {0: T * ((Cout + 127) // 128 * 128) * ((Cin + 31) // 32 * 32), 1: ((Cout + 127) // 128 * 128) * ((T * Cin + 31) // 32 * 32),
                     2: T * ((Cin + 127) // 128 * 128) * ((Cout + 31) // 32 * 32), 3: ((Cin + 127) // 128 * 128) * ((T * Cout + 31) // 32 * 32),
                     5: Cout * Cin * T, 7: Cout * Cin * T}
            for modes in ((0, 2, 5), (1, 3, 7)):
                d = hip.PackDesc()
                d.src, d.Cout, d.Cin, d.R, d.S, d.ndst = w.data_ptr(), Cout, Cin, R, R, 3
                for k, m in enumerate(modes):
                    buf = side.z((sizes[m],))
                    d.dst[k], d.mode[k] = buf.data_ptr(), m
                    outs.append(buf)
                descs.append(d)
            keep.append(w)
        table = (hip.PackDesc * len(descs))(*descs)
        side.call("mrfa_pack_conv_weights_multi", table, len(descs))
        # gradients back: accumulate twice into a non-zero OIHW buffer
        ud = []
        for i, (Cout, Cin, R) in enumerate(shapes):
            T = R * R
            for few in (0, 1):
                acc = side.t(f"pm/a{i}{few}", (T * Cout * Cin,))
                g = side.t(f"pm/g{i}{few}", (Cout, Cin, R, R))
                u = hip.UnpackDesc()
                u.src, u.dst, u.Cout, u.Cin, u.T, u.fewout = acc.data_ptr(), g.data_ptr(), Cout, Cin, T, few
                ud.append(u)
                outs.append(g)
                keep.append(acc)
        ut = (hip.UnpackDesc * len(ud))(*ud)
        side.call("mrfa_unpack_wgrads_multi", ut, len(ud))
        side.call("mrfa_unpack_wgrads_multi", ut, len(ud))
        return side.done(*outs)
    ref, got = both(run)
    assert_close(ref, got, tol=1e-6, what="pack/unpack multi")


@pytest.mark.parametrize("Cc,ld,N,G,res", [(32, 32, 4, 2, True), (64, 64, 6, 3, False), (37, 41, 4, 2, True), (132, 132, 2, 2, False)])
def test_bn_statistic_groups(Cc, ld, N, G, res):
    """v7, statistic groups: mrfa_bn_finalize_groups + mrfa_bn_act_fwd / mrfa_bn_act_bwd with groups = G on a batch of G x (N / G) samples
    (1) against the CPU specification and (2) against G UNGROUPED calls of the same entry points on the G sample ranges -- outputs, dx, the
    residual gradient, running statistics after G momentum updates in group order, gamma / beta gradients summed over the groups.
    (32 / 64 / 132 channels: float4 kernels; 37: scalar kernels.)"""
    H, W = 8, 8
    rows, n = N * H * W, N // G

    def tensors(side):
        x = side.t("bng/x", (rows, ld), -2, 2)
        for g in range(G):                                  # (different statistics per group)
            x[g * n * H * W:(g + 1) * n * H * W] *= 1.0 + 0.5 * g
        rs = side.t("bng/res", (rows, ld))
        gamma, beta = side.t("bng/g", (Cc,), 0.5, 1.5), side.t("bng/b", (Cc,))
        rm, rv = side.t("bng/rm", (Cc,)), side.t("bng/rv", (Cc,), 0.5, 1.5)
        dy = side.t("bng/dy", (rows, ld))
        return x, rs, gamma, beta, rm, rv, dy

    def one(side, x, rs, gamma, beta, rm, rv, dy, nn, groups, dx, dres, dg, dbt, y):
        """statistics, finalize, forward, backward of `nn` samples starting at the given tensors, as `groups` statistic groups"""
        r = nn * H * W
        st = side.z((groups * hip.STATS_SLOTS * 2 * Cc,), torch.float64)
        for g in range(groups):
            side.call("mrfa_bn_stats", x.data_ptr() + 4 * g * (r // groups) * ld, ld, r // groups, Cc, st.data_ptr() + 8 * g * hip.STATS_SLOTS * 2 * Cc)
        sc, sh, mean, inv = (side.z((groups * Cc,)) for _ in range(4))
        if groups > 1:
            side.call("mrfa_bn_finalize_groups", st.data_ptr(), r // groups, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, Cc, groups,
                      sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), inv.data_ptr())
        else:
            side.call("mrfa_bn_finalize", st.data_ptr(), r, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, Cc, 1,
                      sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), inv.data_ptr())
        p = hip.BnActParams()
        p.x, p.ldx, p.N, p.H, p.W, p.C = x.data_ptr(), ld, nn, H, W, Cc
        p.scale, p.shift, p.relu, p.pool, p.groups = sc.data_ptr(), sh.data_ptr(), 1, 0, groups
        if res:
            p.res, p.ldr = rs.data_ptr(), ld
        p.y, p.ldy = y.data_ptr(), ld
        side.call("mrfa_bn_act_fwd", C.byref(p))
        red = side.z((groups * hip.STATS_SLOTS * 2 * Cc,), torch.float64)
        q = hip.BnBwdParams()
        q.x, q.ldx, q.N, q.H, q.W, q.C = x.data_ptr(), ld, nn, H, W, Cc
        q.scale, q.shift, q.relu, q.pool = sc.data_ptr(), sh.data_ptr(), 1, 0
        q.mean, q.invstd, q.gamma = mean.data_ptr(), inv.data_ptr(), gamma.data_ptr()
        q.dy, q.lddy = dy.data_ptr(), ld
        if res:
            q.res, q.ldr, q.dres, q.lddr = rs.data_ptr(), ld, dres.data_ptr(), ld
        q.red, q.dx, q.lddx, q.dgamma, q.dbeta, q.train, q.groups = red.data_ptr(), dx.data_ptr(), ld, dg.data_ptr(), dbt.data_ptr(), 1, groups
        for ph in (1, 2):
            q.phase = ph
            side.call("mrfa_bn_act_bwd", C.byref(q))
        return sc, sh

    def run(side, grouped):
        x, rs, gamma, beta, rm, rv, dy = tensors(side)
        dx, dres = side.t("bng/dx0", (rows, ld)), side.t("bng/dr0", (rows, ld))
        dg, dbt = side.z((Cc,)), side.z((Cc,))
        y = side.garbage((rows, ld))
        if grouped:
            sc, sh = one(side, x, rs, gamma, beta, rm, rv, dy, N, G, dx, dres, dg, dbt, y)
        else:
            scs = []
            for g in range(G):
                sl = slice(g * n * H * W, (g + 1) * n * H * W)
                scs.append(one(side, x[sl], rs[sl], gamma, beta, rm, rv, dy[sl], n, 1, dx[sl], dres[sl], dg, dbt, y[sl]))
            sc, sh = torch.cat([a for a, _ in scs]), torch.cat([b for _, b in scs])
        return side.done(y[:, :Cc], dx[:, :Cc], dres[:, :Cc], dg, dbt, rm, rv, sc, sh)
    spec = run(Side(False), True)
    got = run(Side(True), True)
    sep = run(Side(True), False)
    assert_close(spec, got, tol=5e-4, what="bn groups vs specification")
    assert_close(sep, got, tol=5e-5, what="bn groups vs separate calls")


# ---------------------------------------------------------------------------------------------- K21 (MTIA prior)
@pytest.mark.parametrize("Cc,ld", [(64, 64), (37, 41)])
def test_bn_residual(Cc, ld):
    """y = relu(bn(x) + res) and its backward incl. the residual gradient (float4 and scalar kernels)"""
    def run(side):
        N, H, W = 2, 8, 6
        rows = N * H * W
        x = side.t("bnr/x", (rows, ld), -2, 2)
        res = side.t("bnr/res", (rows, ld))
        gamma, beta = side.t("bnr/g", (Cc,), 0.5, 1.5), side.t("bnr/b", (Cc,))
        rm, rv = side.t("bnr/rm", (Cc,)), side.t("bnr/rv", (Cc,), 0.5, 1.5)
        st = side.z((hip.STATS_SLOTS * 2 * Cc,), torch.float64)
        side.call("mrfa_bn_stats", x.data_ptr(), ld, rows, Cc, st.data_ptr())
        sc, sh, mean, inv = (side.z((Cc,)) for _ in range(4))
        side.call("mrfa_bn_finalize", st.data_ptr(), rows, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, Cc, 1,
                  sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), inv.data_ptr())
        outs = []
        for relu in (1, 0):
            y = side.garbage((rows, ld))
            p = hip.BnActParams()
            p.x, p.ldx, p.N, p.H, p.W, p.C = x.data_ptr(), ld, N, H, W, Cc
            p.scale, p.shift, p.relu, p.pool = sc.data_ptr(), sh.data_ptr(), relu, 0
            p.res, p.ldr = res.data_ptr(), ld
            p.y, p.ldy = y.data_ptr(), ld
            side.call("mrfa_bn_act_fwd", C.byref(p))
            dy = side.t("bnr/dy", (rows, ld))
            dx, dres = side.t("bnr/dx0", (rows, ld)), side.t("bnr/dr0", (rows, ld))
            dg, dbt = side.z((Cc,)), side.z((Cc,))
            red = side.z((hip.STATS_SLOTS * 2 * Cc,), torch.float64)
            q = hip.BnBwdParams()
            q.x, q.ldx, q.N, q.H, q.W, q.C = x.data_ptr(), ld, N, H, W, Cc
            q.scale, q.shift, q.relu, q.pool = sc.data_ptr(), sh.data_ptr(), relu, 0
            q.mean, q.invstd, q.gamma = mean.data_ptr(), inv.data_ptr(), gamma.data_ptr()
            q.dy, q.lddy = dy.data_ptr(), ld
            q.res, q.ldr, q.dres, q.lddr = res.data_ptr(), ld, dres.data_ptr(), ld
            q.red, q.dx, q.lddx, q.dgamma, q.dbeta, q.train = red.data_ptr(), dx.data_ptr(), ld, dg.data_ptr(), dbt.data_ptr(), 1
            for ph in (1, 2):
                q.phase = ph
                side.call("mrfa_bn_act_bwd", C.byref(q))
            outs += [y[:, :Cc], dx[:, :Cc], dres[:, :Cc], dg, dbt]
        return side.done(*outs)
    ref, got = both(run)
    assert_close(ref, got, tol=5e-4, what="bn_residual")


def test_subsample_and_upsample_add():
    def run(side):
        N, H, W, Cc, ld = 2, 8, 12, 32, 36
        x = side.t("ss/x", (N * H * W, ld))
        y = side.garbage((N * (H // 2) * (W // 2), Cc))
        side.call("mrfa_subsample_fwd", x.data_ptr(), ld, N, H, W, Cc, 2, y.data_ptr(), Cc)
        dy = side.t("ss/dy", (N * (H // 2) * (W // 2), Cc))
        dx = side.t("ss/dx0", (N * H * W, ld))
        side.call("mrfa_subsample_bwd", dy.data_ptr(), Cc, N, H, W, Cc, 2, dx.data_ptr(), ld)
        outs = [y, dx[:, :Cc]]
        for f, relu in ((2, 1), (4, 1), (1, 1), (2, 0)):
            Hl, Wl = 4, 3
            lo = side.t("ua/lo", (N * Hl * Wl, Cc))
            base = side.t("ua/base", (N * Hl * f * Wl * f, ld))
            out = side.garbage((N * Hl * f * Wl * f, Cc))
            side.call("mrfa_upsample_add_act_fwd", lo.data_ptr(), Cc, N, Hl, Wl, Cc, f, base.data_ptr(), ld, relu, out.data_ptr(), Cc)
            g = side.t("ua/dy", (N * Hl * f * Wl * f, Cc))
            dlo, dbase = side.t("ua/dlo0", (N * Hl * Wl, Cc)), side.t("ua/db0", (N * Hl * f * Wl * f, ld))
            side.call("mrfa_upsample_add_act_bwd", out.data_ptr(), Cc, g.data_ptr(), Cc, N, Hl, Wl, Cc, f, relu, dlo.data_ptr(), Cc,
                      dbase.data_ptr(), ld)
            outs += [out, dlo, dbase[:, :Cc]]
        return side.done(*outs)
    ref, got = both(run)
    assert_close(ref, got, tol=1e-6, what="subsample/upsample_add")


@pytest.mark.parametrize("slotted", [False, True])
@pytest.mark.parametrize("rows,Cc,ld", [(2208, 192, 192), (4416, 192, 192), (37, 70, 72), (5000, 512, 512), (1026, 256, 260)])
def test_layernorm(rows, Cc, ld, slotted):
    def run(side):
        x = side.t("ln/x", (rows, ld), -2, 2)
        gamma, beta = side.t("ln/g", (Cc,), 0.5, 1.5), side.t("ln/b", (Cc,))
        y = side.garbage((rows, ld))
        mean, rstd = side.z((rows,)), side.z((rows,))
        side.call("mrfa_layernorm_fwd", x.data_ptr(), ld, rows, Cc, gamma.data_ptr(), beta.data_ptr(), 1e-5, y.data_ptr(), ld, mean.data_ptr(),
                  rstd.data_ptr())
        dy = side.t("ln/dy", (rows, ld))
        dx = side.t("ln/dx0", (rows, ld))
        dg, db = side.t("ln/dg0", (Cc,)), side.t("ln/db0", (Cc,))             # (accumulated into)
        scr = side.z((hip.LN_SLOTS * 2 * Cc + 1,)) if slotted else None    # v8: slotted parameter-gradient partials, summed by the launch's last workgroup
        side.call("mrfa_layernorm_bwd", x.data_ptr(), ld, dy.data_ptr(), ld, rows, Cc, gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                  dx.data_ptr(), ld, dg.data_ptr(), db.data_ptr(), scr.data_ptr() if slotted else None)
        return side.done(y[:, :Cc], mean, rstd, dx[:, :Cc], dg, db)
    ref, got = both(run)
    assert_close(ref, got, tol=2e-4, what="layernorm")


def test_gelu():
    def run(side):
        rows, Cc = 300, 576
        x = side.t("gelu/x", (rows, Cc), -4, 4)
        y = side.garbage((rows, Cc))
        side.call("mrfa_gelu_fwd", x.data_ptr(), Cc, rows, Cc, y.data_ptr(), Cc)
        dy, dx = side.t("gelu/dy", (rows, Cc)), side.t("gelu/dx0", (rows, Cc))
        side.call("mrfa_gelu_bwd", x.data_ptr(), Cc, dy.data_ptr(), Cc, rows, Cc, dx.data_ptr(), Cc)
        return side.done(y, dx)
    ref, got = both(run)
    assert_close(ref, got, tol=1e-5, what="gelu")


@pytest.mark.parametrize("B,n,heads,d", [(2, 276, 8, 24), (1, 50, 2, 16), (1, 130, 3, 32)])
def test_attention(B, n, heads, d):
    """softmax(scale q k^T) v per (sample, head) and its backward against the einsum / softmax formulation of the reference
    (tokenpose_base.py:77-91)"""
    def run(side):
        inner = heads * d
        qkv = side.t("att/qkv", (B * n, 3 * inner), -2, 2)
        out = side.garbage((B * n, inner))
        lse = side.z((B * heads * n,))
        scale = d ** -0.5
        side.call("mrfa_attention_fwd", qkv.data_ptr(), 3 * inner, B, n, heads, d, scale, out.data_ptr(), inner, lse.data_ptr())
        dout = side.t("att/do", (B * n, inner))
        dqkv = side.t("att/dq0", (B * n, 3 * inner))
        delta = side.z((B * heads * n,))
        side.call("mrfa_attention_bwd", qkv.data_ptr(), 3 * inner, out.data_ptr(), inner, dout.data_ptr(), inner, lse.data_ptr(),
                  delta.data_ptr(), B, n, heads, d, scale, dqkv.data_ptr(), 3 * inner)
        return side.done(out, lse, dqkv)
    ref, got = both(run)
    assert_close(ref, got, tol=2e-4, what="attention")


# ---------------------------------------------------------------------------------------------- K22 (training losses)
def test_maxpool2_with_ties():
    """2x2 max-pool forward / backward; inputs are ReLU outputs quantised to 3 levels, so most windows tie and the gradient must
    go to the first maximum in scan order (ATen's index)"""
    def run(side):
        N, H, W, Cc, ld = 2, 6, 8, 32, 36
        x = torch.relu(torch.round(side.t("mp/x", (N * H * W, ld), -2, 2)))
        y = side.garbage((N * (H // 2) * (W // 2), Cc))
        side.call("mrfa_maxpool2_fwd", x.data_ptr(), ld, N, H, W, Cc, y.data_ptr(), Cc)
        dy = side.t("mp/dy", (N * (H // 2) * (W // 2), Cc))
        dx = side.t("mp/dx0", (N * H * W, ld))
        side.call("mrfa_maxpool2_bwd", x.data_ptr(), ld, N, H, W, Cc, dy.data_ptr(), Cc, dx.data_ptr(), ld)
        return side.done(y, dx[:, :Cc])
    ref, got = both(run)
    assert_close(ref, got, tol=0, what="maxpool2")


def test_l1_diff_and_antialias_backward():
    def run(side):
        rows, Cc = 5000, 64
        x, y = side.t("l1/x", (rows, Cc)), side.t("l1/y", (rows, Cc))
        y[::7] = x[::7]                                             # exact zeros of the difference: sign(0) = 0
        out = side.z((1,), torch.float64)
        side.call("mrfa_l1_diff_fwd", x.data_ptr(), Cc, y.data_ptr(), Cc, rows, Cc, 0.125, out.data_ptr())
        g = side.t("l1/g", (1,), 0.5, 1.5)
        dx = side.t("l1/dx0", (rows, Cc))
        side.call("mrfa_l1_diff_bwd", x.data_ptr(), Cc, y.data_ptr(), Cc, rows, Cc, g.data_ptr(), 0.25, dx.data_ptr(), Cc)
        outs = [out.float(), dx]
        for k, stride in ((5, 2), (13, 4), (29, 8)):
            N, Cn, H, W = 2, 3, 32, 48
            kern = side.t(f"aa/k{k}", (k, k), 0, 1)
            dy = side.t(f"aa/dy{k}", (N * (H // stride) * (W // stride), 4))
            dimg = side.t(f"aa/dx{k}", (N, Cn, H, W))
            side.call("mrfa_antialias_down_bwd", dy.data_ptr(), 4, N, Cn, H, W, kern.data_ptr(), k, stride, dimg.data_ptr())
            outs.append(dimg)
        return side.done(*outs)
    ref, got = both(run)
    assert_close(ref, got, tol=2e-5, what="l1_diff / antialias_bwd")


def test_maxpool3s2_with_ties():
    def run(side):
        N, H, W, Cc, ld = 2, 9, 12, 32, 36
        x = torch.relu(torch.round(side.t("mp3/x", (N * H * W, ld), -2, 2)))
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = side.garbage((N * Ho * Wo, Cc))
        side.call("mrfa_maxpool3s2_fwd", x.data_ptr(), ld, N, H, W, Cc, y.data_ptr(), Cc)
        dy = side.t("mp3/dy", (N * Ho * Wo, Cc))
        dx = side.t("mp3/dx0", (N * H * W, ld))
        side.call("mrfa_maxpool3s2_bwd", x.data_ptr(), ld, N, H, W, Cc, dy.data_ptr(), Cc, dx.data_ptr(), ld)
        return side.done(y, dx[:, :Cc])
    ref, got = both(run)
    assert_close(ref, got, tol=1e-6, what="maxpool3s2")


# ---------------------------------------------------------------------------------------------- K14-K17 (prior-motion stage)
@pytest.mark.parametrize("with_pos", [False, True])
def test_kp_gaussian(with_pos):
    B, K, H, W, ld, var = 3, 10, 12, 9, 16, 0.1

    def run(side):
        kp = side.t("pg/kp", (B, K, 2), -0.9, 0.9)
        pos = side.t("pg/pos", (1, K, H, W), -0.1, 0.1) if with_pos else None
        out = side.garbage((B * H * W, ld))
        side.call("mrfa_kp_gaussian_fwd", kp.data_ptr(), pos.data_ptr() if with_pos else None, B, K, H, W, var, out.data_ptr(), ld)
        dout = side.t("pg/dout", (B * H * W, ld))
        dkp, dpos = side.t("pg/dkp0", (B, K, 2)), side.t("pg/dpos0", (K, H, W))          # += semantics: non-zero start
        side.call("mrfa_kp_gaussian_bwd", kp.data_ptr(), B, K, H, W, var, dout.data_ptr(), ld, dkp.data_ptr(),
                  dpos.data_ptr() if with_pos else None)
        return side.done(out[:, :K], dkp, dpos)
    a, b = both(run)
    assert_close(a, b, what="kp_gaussian")


@pytest.mark.parametrize("jac,bg", [(True, False), (False, False), (True, True)])
def test_prior_motion(jac, bg):
    """dense_motion.py:36-85 in one kernel: motions, interleaved hourglass input, sparse_deformed; backward into kd / ks / jd / js / bg"""
    B, K, H, W, Cc, var = 2, 10, 16, 16, 3, 0.01
    K1 = K + 1
    lds, ldm, ldi = 4, 4, 64

    def run(side):
        P = hip.PriorParams()
        kd, ks = side.t("pm/kd", (B, K, 2), -0.8, 0.8), side.t("pm/ks", (B, K, 2), -0.8, 0.8)
        eye = torch.tensor([1.0, 0.0, 0.0, 1.0], device=side.dev)
        jd, js = eye + 0.2 * side.t("pm/jd", (B, K, 4)), eye + 0.2 * side.t("pm/js", (B, K, 4))
        bgm = torch.tensor([1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0], device=side.dev) + 0.1 * side.t("pm/bg", (B, 9))
        src = side.t("pm/src", (B * H * W, lds), 0, 1)
        motions, inp = side.garbage((B * K1 * H * W, ldm)), side.z((B * H * W, ldi))
        sparse = side.garbage((B, K1, Cc, H, W))
        P.kd, P.ks = kd.data_ptr(), ks.data_ptr()
        if jac:
            P.jd, P.js = jd.data_ptr(), js.data_ptr()
        if bg:
            P.bg = bgm.data_ptr()
        P.src, P.lds, P.B, P.K, P.H, P.W, P.C, P.inv_var = src.data_ptr(), lds, B, K, H, W, Cc, 1.0 / var
        P.motions, P.ldm, P.inp, P.ldi, P.sparse = motions.data_ptr(), ldm, inp.data_ptr(), ldi, sparse.data_ptr()
        side.call("mrfa_prior_motion_fwd", C.byref(P))
        dinp, dmot, dsp = side.t("pm/dinp", (B * H * W, ldi)), side.t("pm/dmot", (B * K1 * H * W, ldm)), side.t("pm/dsp", (B, K1, Cc, H, W))
        grads = [side.t(f"pm/g{i}", shp) for i, shp in enumerate([(B, K, 2), (B, K, 2), (B, K, 4), (B, K, 4), (B, 9)])]
        P.dinp, P.lddi, P.dmotions, P.dsparse = dinp.data_ptr(), ldi, dmot.data_ptr(), dsp.data_ptr()
        P.dkd, P.dks = grads[0].data_ptr(), grads[1].data_ptr()
        if jac:
            P.djd, P.djs = grads[2].data_ptr(), grads[3].data_ptr()
        if bg:
            P.dbg = grads[4].data_ptr()
        side.call("mrfa_prior_motion_bwd", C.byref(P))
        return side.done(motions[:, :2], inp[:, :K1 * (Cc + 1)], sparse, *grads)
    a, b = both(run)
    assert_close(a[:3], b[:3], tol=1e-5, what="prior_motion fwd")
    assert_close(a[3:], b[3:], tol=5e-4, what="prior_motion bwd")          # sums over 256 pixels of kinked bilinear gradients


def test_softmax_combine():
    B, H, W, K1, ldl, ldm = 2, 9, 7, 11, 12, 4

    def run(side):
        logit = side.t("sc/l", (B * H * W, ldl), -3, 3)
        mot = side.t("sc/m", (B * K1 * H * W, ldm))
        deform, mask, lg = side.garbage((B, H, W, 2)), side.garbage((B, K1, H, W)), side.garbage((B, K1, H, W))
        side.call("mrfa_softmax_combine_fwd", logit.data_ptr(), ldl, mot.data_ptr(), ldm, B, H, W, K1, deform.data_ptr(), mask.data_ptr(),
                  lg.data_ptr())
        dd, dmk, dlg = side.t("sc/dd", (B, H, W, 2)), side.t("sc/dmk", (B, K1, H, W)), side.t("sc/dlg", (B, K1, H, W))
        dlogit, dmot = side.t("sc/dl0", (B * H * W, ldl)), side.t("sc/dm0", (B * K1 * H * W, ldm))
        side.call("mrfa_softmax_combine_bwd", mot.data_ptr(), ldm, B, H, W, K1, mask.data_ptr(), dd.data_ptr(), dmk.data_ptr(), dlg.data_ptr(),
                  dlogit.data_ptr(), ldl, dmot.data_ptr())
        return side.done(deform, mask, lg, dlogit[:, :K1], dmot[:, :2])
    a, b = both(run)
    assert_close(a, b, tol=1e-5, what="softmax_combine")


@pytest.mark.parametrize("jac", [True, False])
def test_kp_head(jac):
    """kp_detector.py:90-120 at the reference's geometry: 58 x 58 logits, K = 10, temperature 0.1"""
    B, H, W, K, ldl, ldj, T = 2, 58, 58, 10, 12, 4, 0.1

    def run(side):
        lg = side.t("kh/l", (B * H * W, ldl), -1, 1)
        jm = side.t("kh/j", (B * H * W, ldj))
        kp, ja, st = side.garbage((B, K, 2)), side.garbage((B, K, 4)), side.garbage((B, K, 2))
        side.call("mrfa_kp_head_fwd", lg.data_ptr(), ldl, jm.data_ptr() if jac else None, ldj, B, H, W, K, T, kp.data_ptr(),
                  ja.data_ptr() if jac else None, st.data_ptr())
        dkp, dja = side.t("kh/dk", (B, K, 2)), side.t("kh/dj", (B, K, 4))
        dl, djm = side.t("kh/dl0", (B * H * W, ldl)), side.t("kh/djm0", (B * H * W, ldj))
        side.call("mrfa_kp_head_bwd", lg.data_ptr(), ldl, jm.data_ptr() if jac else None, ldj, B, H, W, K, T, kp.data_ptr(),
                  ja.data_ptr() if jac else None, st.data_ptr(), dkp.data_ptr(), dja.data_ptr() if jac else None, dl.data_ptr(), ldl,
                  djm.data_ptr() if jac else None, ldj)
        return side.done(kp, ja if jac else kp, dl[:, :K], djm if jac else dl[:, :K])
    a, b = both(run)
    assert_close(a, b, tol=2e-5, what="kp_head")
