#!/bin/bash
O=gpurun_out/r04_s26; mkdir -p $O
( timeout 600 python tools/prefill_host_timeline.py 2>&1 | grep -v Warn ) > $O/prefill_host_timeline.txt
cat $O/prefill_host_timeline.txt
