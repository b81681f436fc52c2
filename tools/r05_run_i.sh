#!/bin/bash
R=$PWD; mkdir -p gpurun_out/i
cd /tmp && export TMPDIR=/tmp
for b in 512x1 1024x1; do
  rm -rf /tmp/prof_b
  BUCKETS=$b timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -- python3 $R/tools/prefill_bucket_times.py > /tmp/log_b 2>&1
  db=$(find /tmp/prof_b -name "*.db" | head -1)
  echo "== bucket $b (21 replays + capture + warm-up runs of ALL buckets are in the trace: read the kernels with ~22 x 28 calls)" >> $R/gpurun_out/i/bucket_kernels.txt
  python3 $R/tools/prof_db.py $db 45 >> $R/gpurun_out/i/bucket_kernels.txt
done
cut -c1-200 $R/gpurun_out/i/bucket_kernels.txt
