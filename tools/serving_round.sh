#!/bin/bash
# Open-loop serving runs of a round (the reference's bench/serving_bench.py workload on Qwen3-0.6B-shaped synthetic
# weights): Poisson arrivals at 16 and 64 requests/s with max_num_seqs 64 (the round-2 settings), and 256 requests
# arriving at once with max_num_seqs 256 (decode batches up to 256 rows).   usage: tools/serving_round.sh r03
TAG=${1:-r03}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
MODEL=$(python - <<PY
import sys; sys.path.insert(0, "$R/tests")
from model_configs import QWEN3_0_6B, make_model_dir
print(make_model_dir(QWEN3_0_6B))
PY
)
export PYTHONPATH=$R/nano-vllm-ascend_amd:$PYTHONPATH
S=$R/nano-vllm-ascend_amd/bench/serving_bench.py
for rate in 16 64; do
  timeout 300 python $S --model $MODEL --num-requests 192 --request-rate $rate --max-output-len 256 --max-num-seqs 64 \
    2>/dev/null | grep '^{' > $O/serving_qwen3_0p6b_rate$rate.json
done
timeout 400 python $S --model $MODEL --num-requests 256 --request-rate 0 --max-output-len 256 --max-num-seqs 256 \
  2>/dev/null | grep '^{' > $O/serving_qwen3_0p6b_256_at_once.json
timeout 400 python $S --model $MODEL --num-requests 512 --request-rate 256 --max-output-len 256 --max-num-seqs 256 \
  2>/dev/null | grep '^{' > $O/serving_qwen3_0p6b_rate256_seqs256.json
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/serving_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f.split("/")[-1], "tok/s", round(d["throughput_tok_s"]), "ttft p50", round(d["ttft_ms"]["p50"], 2),
          "tpot p50", round(d["tpot_ms"]["p50"], 3), "steps", d["engine_steps"], "time", round(d["total_time_s"], 2))
PY
