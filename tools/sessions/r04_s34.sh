#!/bin/bash
O=gpurun_out/r04_s34; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -v -x -m gpu --durations=15 > $O/pytest_full.txt 2>&1
echo "rc=$?"
grep -n "Fatal\|Error\|error\|FAILED\|passed\|failed" $O/pytest_full.txt | head -40
grep -n -B3 -A45 "Fatal Python error" $O/pytest_full.txt | head -120
tail -5 $O/pytest_full.txt
