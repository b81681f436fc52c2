#!/bin/bash
O=gpurun_out/r04_s18; mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $R/bench.py --no-cpu-baseline > /tmp/log_kt 2>&1
cd $R
db=$(find /tmp/prof_kt -name "*.db" | head -1)
python tools/prof_db.py $db --edges paged_attn_prefill_kernel 28 > $O/prefill_step_edges.txt
cat $O/prefill_step_edges.txt
