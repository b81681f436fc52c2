#!/bin/bash
O=gpurun_out/r04_s44; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 95 python -m pytest tests/test_parity_full_shape_gpu.py -q -m gpu 2>&1 | tail -3 > $O/a.txt; cat $O/a.txt
