#!/bin/bash
O=gpurun_out/r04_s11; mkdir -p $O
export PYTHONUNBUFFERED=1
( GEMM_QUICK=1 timeout 600 python tools/gemm_bench.py $O/gemm_bench.json 2>&1 | grep -v Warn | cut -c1-1200 ) > $O/gemm_bench.txt
cat $O/gemm_bench.txt
