#!/bin/bash
O=gpurun_out/r04_s4; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "rccl_code_paths or self_test_failure" 2>&1 | tail -40 ) > $O/pytest_rccl.txt
cut -c1-400 $O/pytest_rccl.txt
