#!/bin/bash
O=gpurun_out/r04_s32; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_engine_gpu.py -q -x -k "prefill_steps_queued or lookahead or config0 or abort" 2>&1 | tail -8 ) > $O/pytest.txt
cat $O/pytest.txt
for i in 1 2; do ( timeout 600 python bench.py 2>/dev/null | tail -1 ) > $O/bench_$i.json; python - <<PY
import json
d=json.load(open("$O/bench_$i.json"))
print({k:d[k] for k in ("value","ms_per_step","ttft_p50_ms","ttft_max_ms","prefill_steps")}, d["prefill_roofline"]["frac"], d["prefill_roofline"]["ms_per_step"], d["roofline"]["frac"])
PY
done
( MI355_LOOKAHEAD=0 timeout 600 python bench.py 2>/dev/null | tail -1 ) > $O/bench_sync.json; python - <<PY
import json
d=json.load(open("$O/bench_sync.json"))
print("sync", {k:d[k] for k in ("value","ms_per_step","ttft_p50_ms","ttft_max_ms","prefill_steps")}, d["prefill_roofline"]["frac"], d["prefill_roofline"]["ms_per_step"])
PY
