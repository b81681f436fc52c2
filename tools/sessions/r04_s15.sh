#!/bin/bash
# A/B in one session on one box: XOR-swizzled image (commit 9677538) vs padded pieces, alternating
O=gpurun_out/r04_s15; mkdir -p $O
export PYTHONUNBUFFERED=1
for rep in 1 2; do
for which in xor pad; do
  if [ $which = xor ]; then export MI355_NANOVLLM_LIB=$PWD/tools/ubench/libmi355_xor_layout.so; else unset MI355_NANOVLLM_LIB; fi
  ( GEMM_QUICK=1 timeout 600 python tools/gemm_bench.py 2>&1 | grep '^{' ) > $O/bench_${which}_$rep.txt
  python - <<PY
import json
print("$which $rep", end=": ")
for l in open("$O/bench_${which}_$rep.txt"):
    d=json.loads(l); print(d['label'], d['library_us'], d['tile_us'], d['eight_wave_r03_us'], d.get('swiglu_fused_us',''), end=" | ")
print()
PY
done; done
