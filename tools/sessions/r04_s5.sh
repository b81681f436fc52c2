#!/bin/bash
O=gpurun_out/r04_s5; mkdir -p $O
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_tile_gpu.py -x -q -k "head_gemm_pick or head_kernel or lm_head or linear_forward_takes" 2>&1 | tail -8 ) > $O/pytest_head.txt
( MI355_PREFILL_P_SPLIT=1 timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -k "30b_a3b_widths" 2>&1 | tail -5 ) > $O/pytest_moe_split.txt
( timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -k "30b_a3b_widths" 2>&1 | tail -5 ) > $O/pytest_moe_default.txt
python - <<'PY' > $O/model_dir.txt
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "nano-vllm-ascend_amd")
from model_configs import QWEN3_0_6B, make_model_dir
print(make_model_dir(QWEN3_0_6B))
PY
D=$(tail -1 $O/model_dir.txt)
( timeout 900 python nano-vllm-ascend_amd/bench/throughput_bench.py --model $D --max-num-seqs 256 2>&1 | grep '^{' ) > $O/throughput_256.json
( timeout 900 python nano-vllm-ascend_amd/bench/throughput_bench.py --model $D --max-num-seqs 32 2>&1 | grep '^{' ) > $O/throughput_32.json
KBENCH_ONLY=head timeout 300 python tools/kbench.py 2>&1 | grep -v Warn > $O/kbench_head.txt
for f in pytest_head pytest_moe_split pytest_moe_default kbench_head; do echo "== $f"; cut -c1-250 $O/$f.txt; done
cat $O/throughput_256.json $O/throughput_32.json | cut -c1-400
