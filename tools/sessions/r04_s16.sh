#!/bin/bash
O=gpurun_out/r04_s16; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests -q -m gpu --durations=15 2>&1 | tail -30 ) > $O/pytest_gpu.txt
tail -22 $O/pytest_gpu.txt
( timeout 600 python bench.py 2>&1 | grep '^{"metric"' ) > $O/bench.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $R/bench.py --no-cpu-baseline > /tmp/log_kt 2>&1
cd $R
grep '^{"metric"' /tmp/log_kt > $O/bench_under_kernel_trace.json
db=$(find /tmp/prof_kt -name "*.db" | head -1)
python tools/prof_db.py $db 40 > $O/kernel_trace.txt
python tools/prof_db.py $db --last paged_attn_decode_kernel 560 >> $O/kernel_trace.txt
python tools/prof_db.py $db --window paged_attn_prefill_kernel 28 "prefill step, 16 x 1024 tokens" > $O/prefill_step_breakdown.txt
cat $O/prefill_step_breakdown.txt
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","ttft_p50_ms")}, d["prefill_roofline"]["frac"], d["prefill_roofline"]["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"])
PY
