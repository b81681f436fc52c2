#!/bin/bash
# A/B on one box: previous commit's library vs the working tree
O=gpurun_out/r04_s23; mkdir -p $O
export PYTHONUNBUFFERED=1
for rep in 1 2; do
for which in prev cur; do
  if [ $which = prev ]; then export MI355_NANOVLLM_LIB=$PWD/tools/ubench/libmi355_prev.so; else unset MI355_NANOVLLM_LIB; fi
  ( GEMM_QUICK=1 timeout 600 python tools/gemm_bench.py 2>&1 | grep '^{' ) > $O/bench_${which}_$rep.txt
  python - <<PY
import json
print("$which $rep", end=": ")
for l in open("$O/bench_${which}_$rep.txt"):
    d=json.loads(l); print(d['label'], d['library_us'], d['tile_us'], d['eight_wave_r03_us'], d.get('swiglu_fused_us',''), end=" | ")
print()
PY
done; done
