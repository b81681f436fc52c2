#!/bin/bash
O=$PWD/gpurun_out/r04_s33; mkdir -p $O; R=$PWD
export PYTHONUNBUFFERED=1
( timeout 300 python tools/prefill_lookahead_ab.py 8 2>&1 | grep -v "amdgpu.ids\|torch_dtype" ) > $O/prefill_lookahead_ab.txt
cat $O/prefill_lookahead_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $R/bench.py --no-cpu-baseline > /tmp/log_kt 2>&1
db=$(find /tmp/prof_kt -name "*.db" | head -1)
python $R/tools/prof_db.py $db --edges paged_attn_prefill_kernel 28 > $O/prefill_step_edges.txt
head -20 $O/prefill_step_edges.txt | cut -c1-150
