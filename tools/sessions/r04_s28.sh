#!/bin/bash
O=gpurun_out/r04_s28; mkdir -p $O
( timeout 300 python tools/gemm_clock.py 2>&1 | grep -v Warn | tail -12 ) | tee $O/gemm_clock.txt
