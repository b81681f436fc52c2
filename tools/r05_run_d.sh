#!/bin/bash
# round 5, GPU call D: engine tests with the captured prefill steps, the open-loop serving runs, the 8-rank dry run
mkdir -p gpurun_out/d
timeout 1200 python3 -m pytest tests/test_engine_gpu.py tests/test_parity_full_shape_gpu.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/d/pytest_engine.txt; cat gpurun_out/d/pytest_engine.txt
bash tools/serving_round.sh d > gpurun_out/d/serving_summary.txt 2>&1; cat gpurun_out/d/serving_summary.txt
MI355_PREFILL_GRAPHS=0 bash tools/serving_round.sh d_eager_prefill > gpurun_out/d/serving_summary_eager_prefill.txt 2>&1; cat gpurun_out/d/serving_summary_eager_prefill.txt
timeout 900 python3 bench.py --gpus 8 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/d/bench_gpus8_dry_run.json 2> gpurun_out/d/bench_gpus8_dry_run.err; echo "dry run rc=$?"; tail -c 1500 gpurun_out/d/bench_gpus8_dry_run.json
