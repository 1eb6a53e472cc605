python -m pytest tests/test_kernels_gpu.py -q -x -k "layernorm" 2>&1 | grep -E "passed|failed|Error" | tail -3
python - <<'PY'
import torch, ctypes as C, sys
sys.path.insert(0, ".")
from mrfa_amd import hip
L = hip.lib(); dev = torch.device("cuda:0")
rows, Cc = 4416, 192
x = torch.randn(rows, Cc, device=dev); dy = torch.randn(rows, Cc, device=dev); dx = torch.zeros(rows, Cc, device=dev)
g = torch.ones(Cc, device=dev); mean = torch.zeros(rows, device=dev); rstd = torch.ones(rows, device=dev)
dg = torch.zeros(Cc, device=dev); db = torch.zeros(Cc, device=dev)
scr = torch.zeros(64, hip.LN_SLOTS * 2 * Cc + 4, device=dev)
s = hip.stream_ptr()
for slotted in (False, True):
    def f(i):
        L.mrfa_layernorm_bwd(s, x.data_ptr(), Cc, dy.data_ptr(), Cc, rows, Cc, g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), Cc, dg.data_ptr(), db.data_ptr(),
                             scr[i % 64].data_ptr() if slotted else None)
    for i in range(5): f(i)
    scr.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(50): f(i)
    e1.record(); torch.cuda.synchronize()
    print("layernorm_bwd 4416x192 slotted" if slotted else "layernorm_bwd 4416x192 direct", f"{e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
PY
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward --no-roofline 2>/dev/null | tail -1 | cut -c1-200; done
