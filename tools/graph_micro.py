"""Micro reproducers for hipGraph replay anomalies (python tools/graph_micro.py)."""
import torch

dev = torch.device("cuda", 0)
a = torch.rand(2, 3, 256, 256, device=dev)
b = torch.rand(2, 3, 256, 256, device=dev)
big = torch.zeros(116_000_000, device=dev)
ref = float((a - b).abs().mean())


def trial(name, body, side_stream, replay_on_side, n=30, busy=False):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    if side_stream:
        with torch.cuda.graph(g, stream=s):
            out = body()
    else:
        with torch.cuda.graph(g):
            out = body()
    bad = 0
    vals = set()
    for i in range(n):
        if busy:
            big.add_(1.0)
        if replay_on_side:
            with torch.cuda.stream(s):
                g.replay()
        else:
            g.replay()
        torch.cuda.synchronize()
        v = float(out)
        if abs(v - ref) > 1e-6:
            bad += 1
            vals.add(round(v, 6))
    print(f"{name}: bad {bad}/{n} {sorted(vals)[:5]}", flush=True)


def mean_only():
    return (a - b).abs().mean()


def zero_then_mean():
    big.zero_()
    return (a - b).abs().mean()


def many_then_mean():
    x = a
    for _ in range(200):
        x = x * 1.0001
    big.zero_()
    return (x - b).abs().mean() * 0 + (a - b).abs().mean()


for side in (False, True):
    for rs in (False, True):
        for busy in (False, True):
            trial(f"mean_only side={side} replay_on_side={rs} busy={busy}", mean_only, side, rs, busy=busy)
            trial(f"zero_then_mean side={side} replay_on_side={rs} busy={busy}", zero_then_mean, side, rs, busy=busy)
trial("many_then_mean", many_then_mean, True, False)
