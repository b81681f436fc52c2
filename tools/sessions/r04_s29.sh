#!/bin/bash
O=gpurun_out/r04_s29; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 1200 python -m pytest tests/test_gemm_tile_gpu.py tests/test_gemm_qkv_store_gpu.py tests/test_parity_full_shape_gpu.py tests/test_engine_gpu.py -x -q -m gpu -k "not tp_ranks and not bench_ranks" 2>&1 | tail -5 ) > $O/pytest.txt
cat $O/pytest.txt
( timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep '^{"metric"' ) > $O/bench.json
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","ttft_p50_ms")}, d["prefill_roofline"]["frac"], d["prefill_roofline"]["ms_per_step"], d["roofline"]["frac"])
PY
( GEMM_QUICK=1 timeout 600 python tools/gemm_bench.py 2>&1 | grep '^{' ) > $O/gemm_bench.txt
python - <<PY
import json
for l in open("$O/gemm_bench.txt"):
    d=json.loads(l); print(d['label'], {k[:-3]:v for k,v in d.items() if k.endswith('_us')})
PY
