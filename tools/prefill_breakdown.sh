#!/bin/bash
# Device timeline of the bench's last prefill step (16 x 1024 tokens) from a rocprofv3 kernel trace:
#   tools/prefill_breakdown.sh <outfile>
R=$PWD; O=$R/$1; mkdir -p $(dirname $O)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_pf
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_pf -- python $R/bench.py --no-cpu-baseline --steps 8 --warmup 2 > /tmp/log_pf 2>&1
db=$(find /tmp/prof_pf -name "*.db" | head -1)
python $R/tools/prof_db.py $db --window paged_attn_prefill_kernel 28 "prefill step, 16 x 1024 tokens" > $O
grep '^{"metric"' /tmp/log_pf | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('under the tracer:', r['value'], 'tok/s, ttft p50', r['ttft_p50_ms'], 'ms, prefill step', r['prefill_roofline']['ms_per_step'], 'ms')" >> $O
cat $O
