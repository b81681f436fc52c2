#!/bin/bash
O=gpurun_out/r04_s9; mkdir -p $O
timeout 120 tools/ubench/gemm_feed > $O/gemm_feed.txt 2>&1
cat $O/gemm_feed.txt
