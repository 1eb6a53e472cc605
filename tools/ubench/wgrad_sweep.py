"""ksplit sweep of mrfa_conv2d_wgrad_nhwc for one layer shape (tuning aid)."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mrfa_amd import hip
L = hip.lib()
dev = torch.device("cuda:0")
def run(Cin, Cout, R, res, B=8, ldx=None):
    ldx = ldx or (Cin + 3) // 4 * 4
    x = torch.randn(B * res * res, ldx, device=dev)
    dy = torch.randn(B * res * res, (Cout + 3) // 4 * 4, device=dev)
    dw = torch.zeros(R * R * Cout * Cin, device=dev)
    ws = torch.empty(16 << 20, device=dev)
    for use_ws in (0, 1):
        for ks in (0, 8, 16, 32, 64, 128, 256, 512):
            q = hip.WgradParams()
            q.x, q.ldx, q.Hin, q.Win, q.N, q.Cin = x.data_ptr(), ldx, res, res, B, Cin
            q.dy, q.ldy, q.Cout, q.Hout, q.Wout = dy.data_ptr(), dy.shape[1], Cout, res, res
            q.R, q.S, q.pad, q.dw, q.alpha, q.nbatch, q.ksplit = R, R, R // 2, dw.data_ptr(), 1.0, 1, ks
            if use_ws:
                q.ws, q.ws_bytes = ws.data_ptr(), ws.numel() * 4
            f = lambda: L.mrfa_conv2d_wgrad_nhwc(hip.stream_ptr(), C.byref(q))
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(f"{Cin}->{Cout} {R}x{R} @{res} ws={use_ws} ksplit={ks:4d}: {ms:.3f} ms  {2.0*B*res*res*Cout*Cin*R*R/ms/1e9:.1f} TF/s", flush=True)
run(98, 128, 1, 128, ldx=128)
run(64, 192, 1, 256)
