#!/bin/bash
O=gpurun_out/r04_s7; mkdir -p $O
export PYTHONUNBUFFERED=1
( KBENCH_ONLY=plain timeout 600 python tools/kbench.py 2>&1 | grep -v Warn ) > $O/kbench_small_geometries.txt
( timeout 900 python tools/small_models_run.py 2>&1 | grep '^{' ) > $O/small_models.txt
( timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep '^{"metric"' ) > $O/bench.json
cat $O/kbench_small_geometries.txt; cut -c1-330 $O/small_models.txt
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","ttft_p50_ms")}, d["prefill_roofline"]["ms_per_step"], d["roofline"]["frac"], d["chain_roofline"]["us_per_layer"])
PY
