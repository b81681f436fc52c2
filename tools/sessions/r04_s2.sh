#!/bin/bash
O=gpurun_out/r04_s2; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "prefill_attention" 2>&1 | tail -8 ) > $O/pytest_attn.txt
( timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -k "micro_batches" 2>&1 | tail -12 ) > $O/pytest_split.txt
( timeout 200 python tools/attn_timeline.py 2>&1 | grep -v Warn ) > $O/attn_timeline_ctx1100.txt
( timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep '^{"metric"' ) > $O/bench.json
( MI355_PREFILL_SPLIT=0 timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep '^{"metric"' ) > $O/bench_nosplit.json
for f in pytest_attn pytest_split attn_timeline_ctx1100; do echo "== $f"; cut -c1-300 $O/$f.txt; done
for f in bench bench_nosplit; do echo "== $f"; python - <<PY
import json
d=json.load(open("$O/$f.json"))
print({k:d[k] for k in ("value","ms_per_step","ttft_p50_ms","ttft_max_ms")}, d["prefill_roofline"]["ms_per_step"], d["roofline"]["frac"])
PY
done
