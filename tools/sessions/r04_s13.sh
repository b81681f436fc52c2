#!/bin/bash
O=gpurun_out/r04_s13; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gemm_tile_gpu.py -x -q -m "gpu or gpu_slow" 2>&1 | tail -5 ) > $O/pytest_gemm.txt
cat $O/pytest_gemm.txt
( GEMM_QUICK=1 timeout 600 python tools/gemm_bench.py $O/gemm_bench.json 2>&1 | grep -v Warn | cut -c1-1500 ) > $O/gemm_bench.txt
python - <<PY
import json
for l in open("$O/gemm_bench.txt"):
    if l.startswith('{'):
        d=json.loads(l); print(d['label'], {k[:-3]:v for k,v in d.items() if k.endswith('_us')}, d['max_abs_diff_vs_library'], d.get('direct_stores_max_abs_diff'))
    else: print(l[:300])
PY
