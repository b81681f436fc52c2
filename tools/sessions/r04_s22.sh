#!/bin/bash
O=gpurun_out/r04_s22; mkdir -p $O
for abl in 0 1 2 3 0; do
  if [ $abl = 0 ]; then unset MI355_NANOVLLM_LIB; else export MI355_NANOVLLM_LIB=$PWD/tools/ubench/libmi355_qkv_abl$abl.so; fi
  echo "ablate=$abl (1: K heads skipped, 2: V heads skipped, 3: both): $(timeout 300 python tools/qkv_store_bench.py 2>&1 | grep 'qkv GEMM')" | tee -a $O/qkv_store_ablation.txt
done
