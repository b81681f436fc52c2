#!/bin/bash
O=gpurun_out/r04_s3; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "prefill_attention" 2>&1 | tail -8 ) > $O/pytest_attn.txt
( timeout 200 python tools/attn_timeline.py 2>&1 | grep -v Warn ) > $O/attn_timeline_ctx1100.txt
( CTX=1024 timeout 200 python tools/attn_timeline.py 2>&1 | grep -v Warn ) > $O/attn_timeline_ctx1024.txt
for f in pytest_attn attn_timeline_ctx1100 attn_timeline_ctx1024; do echo "== $f"; cut -c1-300 $O/$f.txt; done
