python -m pytest tests/test_headline.py -q -x -s -k "bench_batch" 2>&1 | grep -vE "^\s*$|amdgpu.ids" | tail -45
