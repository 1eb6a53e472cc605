python -m pytest tests/test_kernels_gpu.py -q -x -k "conv2d or fused or finalize" 2>&1 | grep -E "passed|failed|Error" | tail -3
python -m pytest tests/test_graph_gpu.py tests/test_parity_gpu.py tests/test_headline.py -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2; do
MRFA_BN_FIN_FUSED=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward --no-roofline 2>/dev/null | tail -1 | cut -c1-200
MRFA_BN_FIN_FUSED=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-forward --no-roofline 2>/dev/null | tail -1 | cut -c1-200
done
