#!/bin/bash
mkdir -p gpurun_out/g
timeout 900 python3 -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "bucketed or queued_behind or lookahead or golden" > gpurun_out/g/pytest.txt 2>&1; echo "rc=$?"; tail -4 gpurun_out/g/pytest.txt | cut -c1-300
timeout 600 python3 tools/prefill_bucket_times.py > gpurun_out/g/prefill_bucket_times.txt 2>&1; tail -40 gpurun_out/g/prefill_bucket_times.txt
