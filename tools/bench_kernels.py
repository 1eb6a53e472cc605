"""Micro-benchmarks of the dominant kernels at the bench configuration's shapes (vox1, 256^2, B=8), hipEvent-timed.
    python tools/bench_kernels.py [--b 8]
Prints TFLOP/s against the 157.3 TF fp32 MFMA peak and GB/s against 8 TB/s HBM."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfa_amd import hip  # noqa: E402
from mrfa_amd.engine import Ctx, convw  # noqa: E402

PEAK_TF = 157.3


def time_it(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=8)
    ap.add_argument("--only", type=str, default="", help="substring filter on the layer name; skips the HBM kernels")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--mfma", type=int, default=0, help="mrfa_set_mfma_mode: 0 native fp32 MFMA, 1 bf16x6 split-operand kernel")
    ap.add_argument("--tile", type=lambda v: int(v, 0), default=0, help="force mrfa_conv_params.tile, e.g. 0x808080 = 128x128 8-wave")
    ap.add_argument("--ab-halo", action="store_true", help="3x3 layers: forward / dgrad with the patch-tiled kernel (conv_halo.hip) off and on")
    ap.add_argument("--tune", action="append", default=[], help="mrfa_set_tuning key=value (repeatable), e.g. --tune conv_small=0 --tune conv_halo_min_tiles=64")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    import mrfa_amd.engine as eng
    eng.FORCE_TILE = a.tile
    from mrfa_amd import hip as _hip
    _hip.check(_hip.lib().mrfa_set_mfma_mode(a.mfma), "set_mfma_mode")
    for kv in a.tune:
        k, v = kv.split("=")
        print(f"tuning {k} = {v} (was {_hip.lib().mrfa_set_tuning(k.encode(), int(v))})")
    torch.manual_seed(0)
    e = Ctx(dev, train=True, record=True)
    B = a.b
    shapes = [  # (name, Cin, Cout, k, res, ups)
        ("refine.conv1 256->128 3x3 @256", 256, 128, 3, 256, 0),
        ("refine.convc1 192->128 3x3 @256", 192, 128, 3, 256, 0),
        ("corr_enc.conv 160->126 3x3 @256", 160, 126, 3, 256, 0),
        ("corr_enc.convc2 128->96 3x3 @256", 128, 96, 3, 256, 0),
        ("gen.down0 64->128 3x3 @256", 64, 128, 3, 256, 0),
        ("gen.up4 128->64 3x3 ups @128->256", 128, 64, 3, 128, 1),
        ("gen.res 128->128 3x3 @128", 128, 128, 3, 128, 0),
        ("gen.chan 512->256 3x3 @64", 512, 256, 3, 64, 0),
        ("refine heads fused 256->256 3x3 @256", 256, 256, 3, 256, 0),
        ("refine heads fused 256->256 3x3 @128", 256, 256, 3, 128, 0),
        ("gen.res 256->256 3x3 @64", 256, 256, 3, 64, 0),
        ("gen.up 256->128 3x3 ups @64->128", 256, 128, 3, 64, 1),
        ("gen.down 128->256 3x3 @128", 128, 256, 3, 128, 0),
        ("gen.down 256->512 3x3 @64", 256, 512, 3, 64, 0),
        ("gen.chan 1024->512 3x3 @32", 1024, 512, 3, 32, 0),
        ("gen.chan 256->128 3x3 @128", 256, 128, 3, 128, 0),
        ("convf2 128->64 3x3 @256", 128, 64, 3, 256, 0),
        ("refine 256->128 3x3 @64", 256, 128, 3, 64, 0),
        ("refine 192->128 3x3 @64", 192, 128, 3, 64, 0),
        ("corr 160->126 3x3 @64", 160, 126, 3, 64, 0),
        ("refine 256->128 3x3 @32", 256, 128, 3, 32, 0),
        ("dm.up 512->128 3x3 ups @32->64", 512, 128, 3, 32, 1),
        ("gen 512->512 3x3 @32", 512, 512, 3, 32, 0),
        ("gen 512->512 3x3 @8", 512, 512, 3, 8, 0),
        ("hg 1024->1024 3x3 @4", 1024, 1024, 3, 4, 0),
        ("to_context 64->192 1x1 @256", 64, 192, 1, 256, 0),
        ("convc1 98->128 1x1 @256 (flat)", 98, 128, 1, 256, 0),
        ("first 3->64 7x7 @256 (flat)", 3, 64, 7, 256, 0),
        ("convf1 2->128 7x7 @256 (flat)", 2, 128, 7, 256, 0),
        ("final 64->3 7x7 @256", 64, 3, 7, 256, 0),
        ("conv2 128->2 3x3 @256", 128, 2, 3, 256, 0),
        ("convo2 128->1 3x3 @256", 128, 1, 3, 256, 0),
        ("conv2 128->2 3x3 @64", 128, 2, 3, 64, 0),
        # TokenPose_B's HRNet stem (MTIA prior): ~0.6 GF each, 32 of each per stage-3 pass
        ("hr 32->32 3x3 @64", 32, 32, 3, 64, 0),
        ("hr 64->64 3x3 @32", 64, 64, 3, 32, 0),
        ("hr 128->128 3x3 @16", 128, 128, 3, 16, 0),
        ("hr 64->64 3x3 @64 (layer1)", 64, 64, 3, 64, 0),
        ("hr 64->256 1x1 @64 (layer1)", 64, 256, 1, 64, 0),
        ("hr 256->64 1x1 @64 (layer1)", 256, 64, 1, 64, 0),
        ("hr 64->64 3x3 @128 (stem, then ::2)", 64, 64, 3, 128, 0),
        ("vit 192->576 1x1 tokens", 192, 576, 1, 0, 0),
    ]
    print(f"{'layer':42s} {'fwd ms':>8s} {'TF/s':>7s} {'%pk':>5s} | {'dgrad ms':>8s} {'TF/s':>7s} | {'wgrad ms':>8s} {'TF/s':>7s}")
    for name, ci, co, k, res, ups in shapes:
        if a.only and a.only not in name:
            continue
        conv = torch.nn.Conv2d(ci, co, k, padding=k // 2).to(dev)
        if res == 0:                                   # token rows: (B, 1, 276, C)
            x = e.new(B, 1, 276, ci)
            x.st.data.normal_()
            out = e.new(B, 1, 276, co)
            flops = 2.0 * B * 276 * co * ci
            e.record = False
            t_f = time_it(lambda: e.conv(x, conv, out=out), iters=a.iters)
            print(f"{name:42s} {t_f:8.3f} {flops / t_f / 1e9:7.1f}")
            continue
        x = e.new(B, res, res, ci)
        x.st.data.normal_()
        ro = res << ups
        out = e.new(B, ro, ro, co)
        cw = convw(conv)
        flops = 2.0 * B * ro * ro * co * ci * k * k
        e.record = False
        few = co <= 4                                  # few-output layers run the direct kernels (no ReLU in the model either)
        t_f = time_it(lambda: e.conv(x, conv, out=out, relu=not few, ups=bool(ups)), iters=a.iters)
        out.st.grad_buf().normal_()
        x.st.grad_buf()
        t_d = time_it(lambda: e._conv_dgrad(x, cw, out, bool(ups), None), iters=a.iters)
        if few:
            dwf, dbf = cw.grad_acc(e.pool32)
            t_w = time_it(lambda: e._chk(e.L.mrfa_conv_fewout_wgrad(e.s, x.ptr, x.ld, x.N, x.H, x.W, cw.Cin, out.gptr, out.ld, cw.Cout, cw.R, cw.pad,
                                                                      dwf.data_ptr(), dbf.data_ptr()), "fewout wgrad"), iters=a.iters)
        else:
            t_w = time_it(lambda: e._conv_wgrad(x, cw, out, bool(ups), None, True), iters=a.iters)
        cw.dw_acc = None
        tf = lambda t: flops / t / 1e9
        if a.ab_halo and k == 3:
            L = _hip.lib()
            res_ab = []
            for on in (0, 8, 9, 0, 8, 9):                      # off, 8-row patches with <= 128-wide tiles, 8-row patches with 256-wide tiles allowed
                L.mrfa_set_tuning(b"conv_halo", int(on > 0))
                L.mrfa_set_tuning(b"conv_halo_bn256", int(on == 9))
                tf_ = time_it(lambda: e.conv(x, conv, out=out, relu=True, ups=bool(ups)), iters=a.iters)
                used = bool(L.mrfa_conv2d_last_config() & (1 << 28))
                td_ = time_it(lambda: e._conv_dgrad(x, cw, out, bool(ups), None), iters=a.iters)
                used_d = bool(L.mrfa_conv2d_last_config() & (1 << 28))
                res_ab.append((f"{on}{'*' if used else ' '}{'*' if used_d else ' '}", used, tf_, td_))
            L.mrfa_set_tuning(b"conv_halo", 1)
            L.mrfa_set_tuning(b"conv_halo_bn256", 1)
            wres = []
            for on in (0, 1, 0, 1):
                L.mrfa_set_tuning(b"wgrad_halo", on)
                tw_ = time_it(lambda: e._conv_wgrad(x, cw, out, bool(ups), None, True), iters=a.iters)
                cw.dw_acc = None
                wres.append(f"wg{on} {tf(tw_):5.1f}")
            L.mrfa_set_tuning(b"wgrad_halo", 1)
            print(f"{name:42s} " + " ".join(f"h={on} f {tf(tf_):5.1f} d {tf(td_):5.1f} |"
                                             for on, used, tf_, td_ in res_ab) + "  " + " ".join(wres), flush=True)
            continue
        print(f"{name:42s} {t_f:8.3f} {tf(t_f):7.1f} {100*tf(t_f)/PEAK_TF:5.1f} | {t_d:8.3f} {tf(t_d):7.1f} | {t_w:8.3f} {tf(t_w):7.1f}", flush=True)
    if a.only:
        return
    # HBM-bound kernels
    print("\nHBM-bound kernels (GB/s of algorithmic bytes)")
    for cch, res in ((64, 256), (128, 128), (256, 64)):
        f = e.new(B, res, res, cch)
        f.st.data.normal_()
        flow = e.new(B, res, res, 2)
        flow.st.data.uniform_(-3, 3)
        o = e.new(B, res, res, cch)
        t = time_it(lambda: e.grid_sample(f, flow, 1, out=o))
        byt = 4.0 * (2 * B * res * res * cch + 2 * B * res * res)
        print(f"grid_sample C={cch} @{res}: {t:.3f} ms  {byt/t/1e6:.0f} GB/s")
        bn = torch.nn.BatchNorm2d(cch).to(dev)
        st = e.f64z(hip.STATS_SLOTS * 2 * cch)
        t = time_it(lambda: e.L.mrfa_bn_stats(e.s, f.ptr, f.ld, f.rows, cch, st.data_ptr()))
        print(f"bn_stats    C={cch} @{res}: {t:.3f} ms  {4.0*B*res*res*cch/t/1e6:.0f} GB/s")
        t = time_it(lambda: e.bn_act(f, bn, st, relu=True, out=o))
        print(f"bn_act      C={cch} @{res}: {t:.3f} ms  {8.0*B*res*res*cch/t/1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
