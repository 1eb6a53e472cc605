"""Split-K sweep of the row-tiled split-operand convolution on the low-resolution hourglass layers (small M, deep K):
   python tools/sweep_splitk.py      -> us per call (incl. the split-K init / epilogue launches) for splitk = auto, 1, 2, 4, ... per shape"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfa_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
L = hip.lib()
hip.set_mfma_mode("bf16x6")
SHAPES = [("512->512 3x3 @4", 8, 4, 4, 512, 512, 0), ("1024->1024 3x3 @4", 8, 4, 4, 1024, 1024, 0), ("512->512 3x3 @8", 8, 8, 8, 512, 512, 0),
          ("1024->512 3x3 @8", 8, 8, 8, 1024, 512, 0), ("512->1024 3x3 @8", 8, 8, 8, 512, 1024, 0), ("2048->512 3x3 @8 ups", 8, 4, 4, 2048, 512, 1),
          ("1024->512 3x3 @16", 8, 16, 16, 1024, 512, 0), ("256->512 3x3 @8", 8, 8, 8, 256, 512, 0), ("1024->256 3x3 @8 ups", 8, 4, 4, 1024, 256, 1), ("512->128 3x3 @16 ups", 8, 8, 8, 512, 128, 1),
          ("512->512 3x3 @16", 8, 16, 16, 512, 512, 0), ("256->64 3x3 @32 ups", 8, 16, 16, 256, 64, 1), ("128->256 3x3 @16", 8, 16, 16, 128, 256, 0),
          ("128->32 3x3 @64 ups", 8, 32, 32, 128, 32, 1), ("64->128 3x3 @32", 8, 32, 32, 64, 128, 0)]
for name, N, H, W, Cin, Cout, ups in SHAPES:
    x = torch.randn(N * H * W, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    cop = (Cout + 127) // 128 * 128
    wp = torch.zeros(9 * cop * Cin, device=dev)
    hip.check(L.mrfa_pack_conv_weight(hip.stream_ptr(), w.data_ptr(), wp.data_ptr(), Cout, Cin, 3, 3, 0), "pack")
    piece = 9 * cop * Cin
    wsb = torch.zeros(3 * piece, dtype=torch.int16, device=dev)
    d = hip.PackDesc()
    d.src, d.Cout, d.Cin, d.R, d.S, d.ndst = w.data_ptr(), Cout, Cin, 3, 3, 1
    d.dst[0], d.mode[0] = wsb.data_ptr(), 8
    hip.check(L.mrfa_pack_conv_weights_multi(hip.stream_ptr(), C.pointer(d), 1), "pack8")
    Ho, Wo = H << ups, W << ups
    y = torch.zeros(N * Ho * Wo, Cout, device=dev)
    stats = torch.zeros(hip.STATS_SLOTS * 2 * Cout + 1, dtype=torch.float64, device=dev)
    p = hip.ConvParams()
    p.x, p.ldx, p.Hin, p.Win, p.ups, p.N, p.Cin = x.data_ptr(), Cin, H, W, ups, N, Cin
    p.w, p.w_ld, p.w_tap, p.kflat, p.w_rows = wp.data_ptr(), Cin, cop * Cin, 0, cop
    p.w_split, p.w_piece = wsb.data_ptr(), piece
    if os.environ.get("NOWS", "0") == "1":           # the fp32 weight layout, split inside the kernel (4 instead of 6 bytes per weight from HBM)
        p.w_split, p.w_piece = None, 0
    p.y, p.ldy, p.Cout, p.Hout, p.Wout = y.data_ptr(), Cout, Cout, Ho, Wo
    p.R, p.S, p.pad, p.alpha, p.nbatch = 3, 3, 1, 1.0, 1
    p.stats = stats.data_ptr()
    fl = 2.0 * N * Ho * Wo * Cout * Cin * 9
    res = []
    # COLD=1: every call reads another copy of the weight planes (600 MB of copies: more than the 256 MB Infinity Cache) -- the training step reads each
    # layer's planes once per pass, straight from HBM; without it the 30 timed calls re-read one copy from the caches
    cold = os.environ.get("COLD", "0") == "1"
    copies = [wsb] + ([wsb.clone() for _ in range(min(63, (600 << 20) // (wsb.numel() * 2)))] if cold else [])
    # FUSED=1: the form the training step launches (engine.Ctx._conv_out): the K split finishes inside the launch (sk_ticket), the output is a fresh
    # zero-filled buffer (y_zero) -- 33 outputs + ticket blocks zeroed outside the timed loop
    fused = os.environ.get("FUSED", "0") == "1"
    if fused:
        nt = ((N * Ho * Wo + 31) // 32) * ((Cout + 31) // 32)
        ys = [torch.zeros(N * Ho * Wo, Cout, device=dev) for _ in range(33)]
        tks = [torch.zeros(nt * 64, dtype=torch.int32, device=dev) for _ in range(33)]
        bias = torch.randn(Cout, device=dev)
        p.bias, p.relu, p.y_zero = bias.data_ptr(), 1, 1
    tile128 = os.environ.get("TILE128", "0") == "1"      # forced splits keep the 128 x 128 (or 128 x 64) split-operand tile instead of shrinking the tile
    for sk in ((0, 1, 8, 16, 32, 64) if cold else (0, 1, 2, 4, 8, 16, 32, 64)):
        p.splitk = sk
        p.tile = ((128 << 16) | (64 if Cout <= 64 else 128)) if (tile128 and sk > 0) else 0
        if fused:
            for t in ys + tks:
                t.zero_()
        for i in range(3):
            if fused:
                p.y, p.sk_ticket = ys[30 + i].data_ptr(), tks[30 + i].data_ptr()
            hip.check(L.mrfa_conv2d_nhwc(hip.stream_ptr(), C.byref(p)), "conv")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30):
            p.w_split = copies[i % len(copies)].data_ptr()
            if fused:
                p.y, p.sk_ticket = ys[i].data_ptr(), tks[i].data_ptr()
            hip.check(L.mrfa_conv2d_nhwc(hip.stream_ptr(), C.byref(p)), "conv")
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        res.append(f"{'auto' if sk == 0 else sk}:{us:6.1f}")
    print(f"{name:24s} {fl / 1e9:5.2f} GF  cfg {L.mrfa_conv2d_last_config():#x}  " + "  ".join(res), flush=True)
