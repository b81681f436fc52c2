#!/bin/bash
O=gpurun_out/r04_full; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests -q -m gpu --durations=45 2>&1 | tail -80 ) > $O/pytest_gpu.txt
tail -70 $O/pytest_gpu.txt | cut -c1-200
