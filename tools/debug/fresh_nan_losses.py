import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mrfa_amd import engine
from tests import cases
from tests.test_losses import TRAIN_PARAMS, _vgg, _transform, _g
from mrfa_amd.losses import GeneratorFullLoss
from mrfa_amd.train import VOX1, HotPath
gd = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
g = _g(gd)
dev = "cuda:0"

def run(fresh, third=True):
    engine.FRESH_MIN_ELEMS = 0 if fresh else (4 << 20) // 4
    engine.FRESH_NAN = fresh and os.environ.get('NONAN') != '1'
    model = HotPath(VOX1, prior="fomm")
    for pfx, mod in (("encoder.", model.encoder), ("dense_motion.", model.dense_motion), ("decoder.", model.decoder)):
        mod.load_state_dict(cases.weights_for(mod.state_dict(), pfx))
    model.to(dev).train(True)
    src, drv = cases.images("g7/src", 1, 256).to(dev), cases.images("g7/drv", 1, 256).to(dev)
    full = GeneratorFullLoss(TRAIN_PARAMS, _vgg(dev)).to(dev)
    kp_s, kp_d = model.encoder(src), model.encoder(drv)
    dm = model.dense_motion(src, kp_d, kp_s)
    gen, _, _ = model.decoder(kp_s["kp"], kp_d["kp"], dm, img=model.down(src), img_full=src)
    lv = full(model.encoder, drv, gen, kp_d, transform=_transform(g, dev))
    sum(v.mean() for v in lv.values()).backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

c = run(True); a, b = run(False), run(False)
names = json.load(open(os.path.join(gd, "losses_param_names.json")))
ref = dict(zip(names, g["param_grad_norms"]))
big = max(ref.values())
rows = []
for n in a:
    rn = ref.get(n)
    if rn is None: continue
    sc = max(rn, 1e-3 * big)
    rows.append((abs(c[n].norm().item() - rn) / sc, abs(a[n].norm().item() - rn) / sc, abs(b[n].norm().item() - rn) / sc, (a[n]-c[n]).norm().item()/max(a[n].norm().item(),1e-3*big), n))
rows.sort(reverse=True)
print("err(fresh-nan)  err(default)  err(default rerun)  |default-fresh|/|default|  name")
for r in rows[:25]:
    print("  %.3e  %.3e  %.3e  %.3e  %s" % r)
