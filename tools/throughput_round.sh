#!/bin/bash
# The reference's own throughput benchmark workload (bench/bench.py:16-40 -> nano-vllm-ascend_amd/bench/throughput_bench.py) at
# max_num_seqs 32 and 256, and the 8-rank dry run of bench.py on one GPU.   usage: tools/throughput_round.sh <tag>
TAG=${1:-r05}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
MODEL=$(python3 - <<PY
import sys; sys.path.insert(0, "$R/tests")
from model_configs import QWEN3_0_6B, make_model_dir
print(make_model_dir(QWEN3_0_6B))
PY
)
export PYTHONPATH=$R/nano-vllm-ascend_amd:$PYTHONPATH
for n in 32 256; do
  timeout 600 python3 $R/nano-vllm-ascend_amd/bench/throughput_bench.py --model $MODEL --max-num-seqs $n 2>/dev/null | grep '^{' > $O/throughput_max_num_seqs_$n.json
  python3 -c "import json;d=json.load(open('$O/throughput_max_num_seqs_$n.json'));print('max_num_seqs',$n,round(d['throughput_tok_s']),'tok/s',round(d['seconds'],2),'s')"
done
timeout 900 python3 bench.py --gpus 8 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_gpus8_dry_run.json 2> $O/bench_gpus8_dry_run.err; echo "dry run rc=$?"; tail -c 600 $O/bench_gpus8_dry_run.json
