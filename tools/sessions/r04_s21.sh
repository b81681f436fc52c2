#!/bin/bash
O=gpurun_out/r04_s21; mkdir -p $O
( timeout 600 python -m pytest tests/test_gemm_qkv_store_gpu.py -x -q 2>&1 | tail -4 ) > $O/pytest_qkv.txt
cat $O/pytest_qkv.txt
( timeout 300 python tools/qkv_store_bench.py 2>&1 | grep -v Warn | tail -3 ) > $O/qkv_store_bench.txt
cat $O/qkv_store_bench.txt
