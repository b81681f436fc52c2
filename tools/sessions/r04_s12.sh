#!/bin/bash
O=gpurun_out/r04_s12; mkdir -p $O
export PYTHONUNBUFFERED=1
( GEMM_QUICK=1 GEMM_ABLATE=1 timeout 600 python tools/gemm_bench.py $O/gemm_ablate.json 2>&1 | grep -v Warn | cut -c1-1500 ) > $O/gemm_ablate.txt
python - <<PY
import json
for l in open("$O/gemm_ablate.txt"):
    if l.startswith('{'):
        d=json.loads(l); print(d['label'], {k[:-3]:v for k,v in d.items() if k.endswith('_us')})
PY
( timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep '^{"metric"' ) > $O/bench.json
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","ttft_p50_ms")}, d["prefill_roofline"], d["roofline"]["frac"])
PY
