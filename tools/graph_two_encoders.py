"""Upper bound of running the two TokenPose_B passes of a step concurrently: two INDEPENDENT encoders, forward+backward,
captured into one hipGraph sequentially vs on two streams (timing experiment only: shared scratch makes values meaningless)."""
import copy
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfa_amd.modules.transformer import get_pose_net  # noqa: E402
from mrfa_amd.modules.util import convert_dict_to_attrit_dict  # noqa: E402
from mrfa_amd.train import VOX1  # noqa: E402
from mrfa_amd.utils.prng import det_uniform, fill_tokenpose_state_dict  # noqa: E402

dev = torch.device("cuda", 0)
nets = []
for k in range(2):
    n = get_pose_net(convert_dict_to_attrit_dict(copy.deepcopy(VOX1["mtia_kp_detector"])), is_train=True)
    n.load_state_dict(fill_tokenpose_state_dict(n.state_dict(), "encoder."))
    nets.append(n.to(dev).train(True))
xs = [det_uniform(f"x{k}", (8, 3, 256, 256), 0, 1).to(dev) for k in range(2)]


def fb(k):
    o = nets[k](xs[k])
    (o["kp"].sum() + o["jacobian"].sum()).backward()


for parallel in (False, True):
    s, side = torch.cuda.Stream(), torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for n in nets:
            n.zero_grad(set_to_none=True)
        if parallel:
            side.wait_stream(s)
            with torch.cuda.stream(side):
                fb(1)
            fb(0)
            s.wait_stream(side)
        else:
            fb(0); fb(1)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            if parallel:
                side.wait_stream(s)
                with torch.cuda.stream(side):
                    fb(1)
                fb(0)
                s.wait_stream(side)
            else:
                fb(0); fb(1)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    print(f"parallel={parallel}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms for two encoder forward+backward passes (B=8 each)")
