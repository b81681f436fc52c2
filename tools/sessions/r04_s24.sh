#!/bin/bash
O=gpurun_out/r04_s24; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gemm_qkv_store_gpu.py tests/test_gemm_tile_gpu.py "tests/test_parity_full_shape_gpu.py::test_bench_workload_bs32_paging_invariance_and_decode_equals_reprefill" -x -q 2>&1 | tail -5 ) > $O/pytest.txt
cat $O/pytest.txt
( GEMM_QUICK=1 timeout 600 python tools/gemm_bench.py 2>&1 | grep '^{' ) > $O/bench.txt
python - <<PY
import json
for l in open("$O/bench.txt"):
    d=json.loads(l); print(d['label'], d['library_us'], d['tile_us'], d['eight_wave_r03_us'], d.get('swiglu_fused_us',''), end=" | ")
print()
PY
( timeout 300 python tools/qkv_store_bench.py 2>&1 | grep 'qkv GEMM' ) | tee $O/qkv_store_bench.txt
