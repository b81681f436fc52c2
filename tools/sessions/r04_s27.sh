#!/bin/bash
O=gpurun_out/r04_s27; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 900 python tools/gemm_bench.py $O/gemm_bench_full.json 2>&1 | grep '^{' ) > $O/gemm_bench_full.txt
python - <<PY
import json
for l in open("$O/gemm_bench_full.txt"):
    d=json.loads(l); print(d['label'], d['shape'], {k[:-3]:v for k,v in d.items() if k.endswith('_us')})
PY
