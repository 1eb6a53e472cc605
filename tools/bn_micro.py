import os, sys, ctypes as C, torch
sys.path.insert(0, "/root/repo")
from mrfa_amd import hip
L = hip.lib(); dev = "cuda:0"
def t(fn, it=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it * 1e3
for (N, H, W, Cc) in ((8, 64, 64, 32), (8, 32, 32, 64), (8, 16, 16, 128), (8, 64, 64, 64), (8, 64, 64, 256), (8, 256, 256, 64)):
    rows = N * H * W
    x = torch.randn(rows, Cc, device=dev); dy = torch.randn(rows, Cc, device=dev); dx = torch.zeros(rows, Cc, device=dev)
    sc, sh, mean, inv, gamma = (torch.rand(Cc, device=dev) + 0.5 for _ in range(5))
    red = torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=dev); dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    st = torch.zeros(32 * 2 * Cc, dtype=torch.float64, device=dev)      # MRFA_STATS_SLOTS blocks
    q = hip.BnBwdParams()
    q.x, q.ldx, q.N, q.H, q.W, q.C = x.data_ptr(), Cc, N, H, W, Cc
    q.scale, q.shift, q.relu, q.pool = sc.data_ptr(), sh.data_ptr(), 1, 0
    q.mean, q.invstd, q.gamma = mean.data_ptr(), inv.data_ptr(), gamma.data_ptr()
    q.dy, q.lddy = dy.data_ptr(), Cc
    q.red, q.dx, q.lddx, q.dgamma, q.dbeta, q.train = red.data_ptr(), dx.data_ptr(), Cc, dg.data_ptr(), db.data_ptr(), 1
    s = hip.stream_ptr()
    def p1(): q.phase = 1; L.mrfa_bn_act_bwd(s, C.byref(q))
    def p2(): q.phase = 2; L.mrfa_bn_act_bwd(s, C.byref(q))
    def stt(): L.mrfa_bn_stats(s, x.data_ptr(), Cc, rows, Cc, st.data_ptr())
    print(f"MRFA_BN_WGS={os.environ.get('MRFA_BN_WGS','2048'):5s} ({N},{H},{W},{Cc}): stats {t(stt):6.1f} us  bwd1 {t(p1):6.1f} us  bwd2 {t(p2):6.1f} us   ({rows*Cc*4/1e6:.1f} MB)")
