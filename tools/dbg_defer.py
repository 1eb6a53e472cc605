"""Gradient of one eager step with the encoder weight gradients in line (fan-out 0) and dealt onto side streams (4): per-parameter relative
distance, largest first.   python tools/dbg_defer.py [2|3 encoder passes]"""
import os, sys, torch
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/mrfa_amd') else os.getcwd())
import bench
from mrfa_amd import engine
from mrfa_amd.train import VOX1, HotPath, make_optimizer, train_step, l1_loss
from mrfa_amd.utils.prng import det_uniform
dev = torch.device("cuda:0")
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 2
def run(fanout):
    torch.manual_seed(0)
    m = HotPath(VOX1, prior="mtia"); bench.init_weights(m); m.to(dev).train(True)
    m.concurrent_encoder = True; m.defer_decoder_wgrads = True
    m._wdefer_enc.fanout = fanout
    opt = make_optimizer(m, fused=True)
    src, drv = det_uniform("d/s", (2, 3, 256, 256), 0, 1).to(dev), det_uniform("d/d", (2, 3, 256, 256), 0, 1).to(dev)
    third = det_uniform("d/t", (2, 3, 256, 256), 0, 1).to(dev)
    outs = []
    for rep in range(2):
        opt.zero_grad(set_to_none=False) if hasattr(opt, 'zero_grad') else None
        for p in m.parameters():
            if p.grad is not None: p.grad.zero_()
        with engine.direct_param_grads():
            if nfr == 2:
                kp_s, kp_d = m.encode_pair(src, drv)
                extra = 0
            else:
                kp_s, kp_d, kp_t = m.encode_many([src, drv, third])
                extra = sum(v.float().pow(2).mean() for v in kp_t.values())
            gen = m.decode(src, kp_s, kp_d)
            loss = l1_loss(gen, drv) + extra
            loss.backward()
        m.join()
        torch.cuda.synchronize()
        outs.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    return outs
a = run(0); a2 = run(0); b = run(4)
for rep in range(2):
    rows = []
    for n in a[rep]:
        ga, gb, gc = a[rep][n], b[rep][n], a2[rep][n]
        if float(ga.norm()) < 1e-6 * ga.numel() ** 0.5:
            continue                      # (conv biases in front of a BatchNorm: their gradient is rounding noise around zero)
        d = float((ga - gb).norm() / (ga.norm() + 1e-20))
        d0 = float((ga - gc).norm() / (ga.norm() + 1e-20))
        rows.append((d / max(d0, 1e-7), d, d0, n))
    rows.sort(reverse=True)
    print("rep", rep, ": distance fan-out 4 vs 0, next to the distance of two fan-out 0 runs (run-to-run noise); largest ratio first")
    for r in rows[:14]: print("   ratio %8.2f   on/off %.3e   noise %.3e   %s" % r)
    enc = [r for r in rows if r[3].startswith("encoder.")]
    print("   encoder parameters: median on/off %.3e, median noise %.3e" % (sorted(r[1] for r in enc)[len(enc) // 2], sorted(r[2] for r in enc)[len(enc) // 2]))
