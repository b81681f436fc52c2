#!/bin/bash
O=gpurun_out/r04_s20; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 1200 python -m pytest tests/test_parity_full_shape_gpu.py tests/test_engine_gpu.py -x -q -m gpu -k "not tp_ranks and not bench_ranks" 2>&1 | tail -8 ) > $O/pytest_engine.txt
cat $O/pytest_engine.txt
( timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep '^{"metric"' ) > $O/bench.json
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","ttft_p50_ms")}, d["prefill_roofline"]["frac"], d["prefill_roofline"]["ms_per_step"], d["roofline"]["frac"])
PY
( MI355_QKV_STORE=0 timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep '^{"metric"' ) > $O/bench_two_launches.json
python - <<PY
import json
d=json.load(open("$O/bench_two_launches.json"))
print("two launches:", {k:d[k] for k in ("value","ms_per_step","ttft_p50_ms")}, d["prefill_roofline"]["frac"], d["prefill_roofline"]["ms_per_step"])
PY
