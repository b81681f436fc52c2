#!/bin/bash
# round 5, GPU call B: split-K geometry A/B of the decode chain, the double-buffered K loop on Qwen3-32B TP-8 shapes
mkdir -p gpurun_out/b
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "double_buffered or two_row_tiles or instrumented or qwen3_32b or test_gemm_packed" 2>&1 | tail -4
timeout 300 python3 tools/chain_ab.py 5 > gpurun_out/b/chain_ab.txt 2>&1; cat gpurun_out/b/chain_ab.txt
for pipe in 1 0 1 0; do
  echo "== MI355_GEMM_PIPE=$pipe" >> gpurun_out/b/kbench_32b.txt
  MI355_GEMM_PIPE=$pipe KBENCH_ONLY=32b_gemms timeout 300 python3 tools/kbench.py >> gpurun_out/b/kbench_32b.txt 2>&1
  MI355_GEMM_PIPE=$pipe KBENCH_ONLY=32b timeout 300 python3 tools/kbench.py >> gpurun_out/b/kbench_32b.txt 2>&1
done
grep -v "^$" gpurun_out/b/kbench_32b.txt | grep -E "==|tp8|us\"|frac\"" | head -80
