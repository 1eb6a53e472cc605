python -m pytest tests/test_kernels_gpu.py -q -x -k "grid_sample" 2>&1 | grep -E "passed|failed|Error" | tail -3
python - <<'PY'
import torch, sys
sys.path.insert(0, ".")
from mrfa_amd import hip
L = hip.lib(); dev = torch.device("cuda:0"); s = hip.stream_ptr()
B = 4
for Cc, res in ((64, 512), (128, 256), (256, 128), (512, 64), (64, 256), (128, 128)):
    inp = torch.randn(B * res * res, Cc, device=dev)
    out = torch.empty(B * res * res, Cc, device=dev)
    # smooth flow: identity grid + a few pixels of displacement (mode 1: pixel offsets)
    flow = (torch.randn(B, 2, res // 16, res // 16, device=dev) * 3.0)
    flow = torch.nn.functional.interpolate(flow, size=(res, res), mode="bilinear").permute(0, 2, 3, 1).reshape(-1, 2).contiguous()
    for tiled in (0, 1, 0, 1):
        L.mrfa_set_tuning(b"grid_sample_tiled", tiled)
        f = lambda: L.mrfa_grid_sample_fwd(s, inp.data_ptr(), Cc, res * res * Cc, 1, res, res, Cc, flow.data_ptr(), 2, B, res, res, out.data_ptr(), Cc, 1)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20
        byt = 4.0 * (2 * B * res * res * Cc + 2 * B * res * res)
        print(f"grid_sample fwd C={Cc} @{res} B={B} tiled={tiled}: {t*1e3:.1f} us  {byt/t/1e6:.0f} GB/s")
PY
for t in 0 1; do python bench.py --size 512 --batch 4 --inference --steps 20 --warmup 3 --tune grid_sample_tiled=$t 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config5 tiled=$t', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"; done
