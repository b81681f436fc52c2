#!/bin/bash
O=gpurun_out/r04_s6; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_attn_hd64_gpu.py -x -q 2>&1 | tail -15 ) > $O/pytest_hd64.txt
( timeout 900 python -m pytest tests/test_attn_plain_gpu.py tests/test_kernels_gpu.py -x -q -k "attention or engine_with or kv_ or scatter or gather or fused_step" 2>&1 | tail -8 ) > $O/pytest_attn_all.txt
for f in pytest_hd64 pytest_attn_all; do echo "== $f"; cut -c1-300 $O/$f.txt; done
