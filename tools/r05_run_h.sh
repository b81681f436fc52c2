#!/bin/bash
mkdir -p gpurun_out/h
for m in 256 128 64; do
  echo "== MI355_PREFILL_STREAM_MAX=$m" >> gpurun_out/h/prefill_bucket_times.txt
  MI355_PREFILL_STREAM_MAX=$m timeout 600 python3 tools/prefill_bucket_times.py 2>&1 | grep -v "^\[\|amdgpu.ids" | tr -d '\n' >> gpurun_out/h/prefill_bucket_times.txt
  echo >> gpurun_out/h/prefill_bucket_times.txt
done
cat gpurun_out/h/prefill_bucket_times.txt
