#!/bin/bash
O=gpurun_out/r04_s30; mkdir -p $O
export PYTHONUNBUFFERED=1
( GEMM_QUICK=1 timeout 600 python tools/gemm_bench.py 2>&1 | grep '^{' ) > $O/gemm_bench.txt
python - <<PY
import json
for l in open("$O/gemm_bench.txt"):
    d=json.loads(l); print(d['label'], {k[:-3]:v for k,v in d.items() if k.endswith('_us')})
PY
