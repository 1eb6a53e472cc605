"""What one launch costs on the keypoint encoder's forward chain (timing experiment only: values are meaningless when launches are skipped).
The TokenPose_B forward of B=8 frames is captured into a hipGraph alone and as two passes on two streams; then again with selected C-ABI
entry points replaced by no-ops (MRFA_PROBE_SKIP=mrfa_bn_finalize,mrfa_bn_act_fwd ...), which gives the wall time the chain would have
WITHOUT those launches -- the upper bound of fusing them into their neighbours.

    python tools/enc_chain_probe.py [skip,list ...]
"""
import copy
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfa_amd import hip  # noqa: E402
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
keep = []


def fwd(k):
    keep.append(nets[k](xs[k]))


def measure(parallel, label):
    s, side = torch.cuda.Stream(), torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())

    def body():
        if parallel:
            side.wait_stream(s)
            with torch.cuda.stream(side):
                fwd(1)
            fwd(0)
            s.wait_stream(side)
        else:
            fwd(0)
    with torch.cuda.stream(s):
        body()
        torch.cuda.synchronize()
        keep.clear()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{label:44s} {'two passes, two streams' if parallel else 'one pass':24s} {ms:7.3f} ms", flush=True)
    keep.clear()
    return ms


base = [measure(False, "all launches"), measure(True, "all launches")]
for spec in sys.argv[1:] or ["mrfa_bn_finalize", "mrfa_bn_finalize,mrfa_bn_act_fwd"]:
    L = hip.lib()
    saved = {}
    counts = {}
    for name in spec.split(","):
        saved[name] = getattr(L, name)
        counts[name] = 0

        def noop(*a, _n=name):
            counts[_n] += 1
            return 0
        setattr(L, name, noop)
    r = [measure(False, "without " + spec), measure(True, "without " + spec)]
    for name, f in saved.items():
        setattr(L, name, f)
    n1 = sum(counts.values())
    print(f"    skipped launches (all captures + warm passes): {counts};  one pass: {base[0] - r[0]:.3f} ms less, two passes: {base[1] - r[1]:.3f} ms less")
