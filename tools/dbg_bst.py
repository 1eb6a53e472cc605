import os, sys, copy, torch
sys.path.insert(0, "/root/repo")
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
from mrfa_amd import engine
from mrfa_amd.modules.transformer import get_pose_net
from mrfa_amd.modules.util import convert_dict_to_attrit_dict
from mrfa_amd.train import VOX1
from mrfa_amd.utils.prng import det_uniform, fill_tokenpose_state_dict
dev = torch.device("cuda", 0)
n = get_pose_net(convert_dict_to_attrit_dict(copy.deepcopy(VOX1["mtia_kp_detector"])), is_train=True)
n.load_state_dict(fill_tokenpose_state_dict(n.state_dict(), "encoder."))
n = n.to(dev).train(True)
x = det_uniform("x", (2, 3, 256, 256), 0, 1).to(dev)
def run(flag):
    engine.BN_BWD_IN_DGRAD = flag
    n.zero_grad(set_to_none=True)
    o = n(x)
    (o["kp"].sum() + o["jacobian"].sum()).backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().double().clone() for k, p in n.named_parameters() if p.grad is not None}
a, b = run(False), run(True)
a2 = run(False)
rows = []
for k in a:
    d = float((a[k] - b[k]).norm() / (a[k].norm() + 1e-30)); nz = float((a[k] - a2[k]).norm() / (a[k].norm() + 1e-30))
    rows.append((d, nz, k))
rows.sort(reverse=True)
for d, nz, k in rows[:25]:
    print(f"{d:9.2e} (noise {nz:8.1e}) {k}  {tuple(a[k].shape)}")

