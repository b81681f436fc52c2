#!/bin/bash
# round 4, GPU session 1: validate the prefill-attention changes, measure its variants, the decode attention timeline,
# the Infinity-Cache read ceiling, the two-stream prefill experiment, and a bench line.
O=gpurun_out/r04_s1; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "prefill_attention or decode_attention or fused_step or qknorm" 2>&1 | tail -8 ) > $O/pytest_attn.txt
( ROUNDS=3 KBENCH_ONLY=prefill_order timeout 300 python tools/kbench.py 2>&1 | grep -v Warn ) > $O/kbench_prefill_variants.txt
( timeout 200 python tools/attn_timeline.py 2>&1 | grep -v Warn ) > $O/attn_timeline_ctx1100.txt
( CTX=1024 timeout 200 python tools/attn_timeline.py 2>&1 | grep -v Warn ) > $O/attn_timeline_ctx1024.txt
( timeout 120 tools/ubench/hbm_peak 2>&1 ) > $O/hbm_peak.txt
( timeout 300 python tools/prefill_overlap_exp.py 2>&1 | grep -v Warn ) > $O/prefill_overlap.txt
( timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep '^{"metric"' ) > $O/bench.json
for f in pytest_attn kbench_prefill_variants attn_timeline_ctx1100 hbm_peak prefill_overlap; do echo "== $f"; cat $O/$f.txt; done
echo "== bench"; head -c 1500 $O/bench.json
