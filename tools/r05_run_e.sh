#!/bin/bash
mkdir -p gpurun_out/e
timeout 1500 python3 -m pytest tests/test_engine_gpu.py tests/test_parity_full_shape_gpu.py -x -q -m gpu > gpurun_out/e/pytest_engine.txt 2>&1; echo "rc=$?"; grep -n "Fatal\|fault\|passed\|failed\|Error" gpurun_out/e/pytest_engine.txt | head -20; tail -5 gpurun_out/e/pytest_engine.txt | cut -c1-300
