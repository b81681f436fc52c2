#!/bin/bash
O=gpurun_out/r04_s31; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 1200 python -m pytest tests -q -m gpu_slow --durations=8 2>&1 | tail -16 ) > $O/pytest_gpu_slow.txt
cat $O/pytest_gpu_slow.txt
