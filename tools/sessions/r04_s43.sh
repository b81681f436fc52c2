#!/bin/bash
O=gpurun_out/r04_s43; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 55 python -m pytest tests/test_engine_gpu.py -q -m gpu -k "rccl_code_paths or prefill_steps_queued" 2>&1 | tail -3 > $O/a.txt; cat $O/a.txt
timeout 40 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "moe_route or moe_block_golden or moe_block_qwen3" 2>&1 | tail -3 > $O/b.txt; cat $O/b.txt
timeout 75 python -m pytest tests/test_gemm_tile_gpu.py tests/test_gemm_qkv_store_gpu.py tests/test_xgmi_comm_gpu.py -q -m gpu 2>&1 | tail -3 > $O/c.txt; cat $O/c.txt
