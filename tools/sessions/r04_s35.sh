#!/bin/bash
O=gpurun_out/r04_s35; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_engine_gpu.py -v -x -m gpu -k "rccl_code_paths or prefill_steps_queued or xgmi_self_test" > $O/pytest.txt 2>&1
echo "rc=$?"; grep -n "PASSED\|FAILED\|Fatal\|passed\|failed" $O/pytest.txt | head; grep -A12 "Fatal Python" $O/pytest.txt | head -30
