#!/bin/bash
O=gpurun_out/r04_s14; mkdir -p $O
export PYTHONUNBUFFERED=1
( GEMM_QUICK=1 timeout 600 python tools/gemm_bench.py $O/gemm_ablate.json 2>&1 | grep -v Warn | cut -c1-1500 ) > $O/gemm_ablate.txt
python - <<PY
import json
for l in open("$O/gemm_ablate.txt"):
    if l.startswith('{'):
        d=json.loads(l); print(d['label'], {k[:-3]:v for k,v in d.items() if k.endswith('_us')})
    else: print(l[:300])
PY
