#!/bin/sh
# builds the experiment libraries next to their sources (git-ignored; they travel with gpurun)
set -e
cd "$(dirname "$0")"
for f in gemm_exp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o lib$f.so $f.hip
done
