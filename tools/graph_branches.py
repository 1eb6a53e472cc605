"""Do two independent chains of small kernels, captured on two streams into ONE hipGraph, overlap on replay?
(decides whether the two TokenPose_B encoder passes of a training step are worth forking onto a side stream)"""
import os
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

dev = torch.device("cuda", 0)
n_ops = 300
a = [torch.randn(2048, 288, device=dev) for _ in range(2)]
w = [torch.randn(288, 64, device=dev) * 0.05 for _ in range(2)]
w2 = [torch.randn(64, 288, device=dev) * 0.05 for _ in range(2)]


def chain(k):
    x = a[k]
    for _ in range(n_ops // 2):
        x = torch.relu(x @ w[k]) @ w2[k]
    return x


def capture(parallel):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    side = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain(0); chain(1)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            if parallel:
                side.wait_stream(s)
                with torch.cuda.stream(side):
                    y1 = chain(1)
                y0 = chain(0)
                s.wait_stream(side)
            else:
                y0 = chain(0)
                y1 = chain(1)
            out = y0.sum() + y1.sum()
    return g, out


for parallel in (False, True):
    g, out = capture(parallel)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print(f"parallel={parallel}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per replay ({2 * n_ops * 3 // 2} kernels), out {float(out):.4f}")
