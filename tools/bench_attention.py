"""Per-launch time of mrfa_attention_fwd / mrfa_attention_bwd at the MTIA prior's shape (B=8, 276 tokens, 8 heads x 24), VALU kernels
(tokenpose.hip) vs the matrix-pipe kernels (attention_mfma.hip): 100 launches captured into a hipGraph each (no host launch overhead).
    python tools/bench_attention.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfa_amd import hip  # noqa: E402

L = hip.lib()
dev = torch.device("cuda:0")
B, n, heads, d = 8, 276, 8, 24
inner = heads * d
qkv = torch.randn(B * n, 3 * inner, device=dev)
out = torch.empty(B * n, inner, device=dev)
lse = torch.empty(B * heads * n, device=dev)
dout = torch.randn(B * n, inner, device=dev)
dqkv = torch.zeros(B * n, 3 * inner, device=dev)
delta = torch.empty(B * heads * n, device=dev)
scale = d ** -0.5


def fwd():
    hip.check(L.mrfa_attention_fwd(hip.stream_ptr(), qkv.data_ptr(), 3 * inner, B, n, heads, d, scale, out.data_ptr(), inner, lse.data_ptr()), "fwd")


def bwd():
    hip.check(L.mrfa_attention_bwd(hip.stream_ptr(), qkv.data_ptr(), 3 * inner, out.data_ptr(), inner, dout.data_ptr(), inner, lse.data_ptr(),
                                   delta.data_ptr(), B, n, heads, d, scale, dqkv.data_ptr(), 3 * inner), "bwd")


def graph_time(fn, reps=100):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (5 * reps)


for mode in (0, 1, 0, 1):
    L.mrfa_set_tuning(b"attention_mfma", mode)
    print(f"attention_mfma={mode}: fwd {graph_time(fwd):6.1f} us   bwd (q + kv) {graph_time(bwd):6.1f} us", flush=True)
