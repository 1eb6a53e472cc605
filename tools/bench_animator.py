"""Per-frame time of the streaming animation loop (source-side work cached) vs the full per-pair forward, hipGraph replays."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfa_amd.graph import GraphedForward  # noqa: E402
from mrfa_amd.infer import Animator  # noqa: E402
from mrfa_amd.train import VOX1, HotPath  # noqa: E402
from mrfa_amd.utils.prng import det_uniform, fill_state_dict  # noqa: E402

dev = torch.device("cuda", 0)
model = HotPath(VOX1)
for pfx, mod in (("encoder.", model.encoder), ("dense_motion.", model.dense_motion), ("decoder.", model.decoder)):
    mod.load_state_dict(fill_state_dict(mod.state_dict(), tag=pfx))
model.to(dev).eval()
for B in (1, 8):
    src = det_uniform("ba/src", (B, 3, 256, 256), 0, 1).to(dev)
    drv = det_uniform("ba/drv", (B, 3, 256, 256), 0, 1).to(dev)
    gf = GraphedForward(model, src, drv)
    an = Animator(model, graph=True)
    an.set_source(src)
    res = {}
    for name, fn in (("full forward", lambda: gf(src, drv)), ("animator frame", lambda: an(drv))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t) / 20 * 1e3
    print(f"B={B}: full forward {res['full forward']:.2f} ms, animator frame {res['animator frame']:.2f} ms "
          f"({B / res['animator frame'] * 1e3:.0f} frames/s)")
