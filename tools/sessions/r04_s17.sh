#!/bin/bash
O=gpurun_out/r04_s17; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep '^{"metric"' ) > $O/bench.json
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","ttft_p50_ms")}, d["prefill_roofline"]["frac"], d["prefill_roofline"]["ms_per_step"], d["roofline"]["frac"])
PY
( timeout 600 python tools/prefill_host_timeline.py 2>&1 | grep -v Warn ) > $O/prefill_host_timeline.txt
cat $O/prefill_host_timeline.txt
