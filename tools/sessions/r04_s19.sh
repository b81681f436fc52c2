#!/bin/bash
O=gpurun_out/r04_s19; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gemm_qkv_store_gpu.py -x -q 2>&1 | tail -25 ) > $O/pytest_qkv.txt
cat $O/pytest_qkv.txt
