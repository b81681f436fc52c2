#!/bin/bash
mkdir -p gpurun_out/f
timeout 1500 python3 -m pytest tests/test_engine_gpu.py -x -q -m gpu > gpurun_out/f/pytest_engine.txt 2>&1; echo "rc=$?"; grep -n "Fatal\|fault\|passed\|failed" gpurun_out/f/pytest_engine.txt | head; tail -4 gpurun_out/f/pytest_engine.txt | cut -c1-300
bash tools/serving_round.sh f > gpurun_out/f/serving_summary.txt 2>&1; cat gpurun_out/f/serving_summary.txt
