#!/bin/bash
# round 5, GPU call A: chain timeline, the driver's bench command x3 on the reordered start-up, the tests of what changed
mkdir -p gpurun_out/a
timeout 300 python3 tools/chain_timeline.py 20 > gpurun_out/a/chain_timeline.txt 2> gpurun_out/a/chain_timeline.err; echo "timeline rc=$?"
for i in 1 2 3; do
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/a/bench_$i.json 2> gpurun_out/a/bench_$i.err; echo "bench_$i rc=$?"
done
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/a/bench_*.json")):
    r = json.loads(open(f).read().strip().splitlines()[-1]); s = r["prefill_steps_ms"]
    print(f, f"ms/step={r['ms_per_step']:.4f} ttft_p50={r['ttft_p50_ms']:.2f} frac={r['prefill_roofline']['frac']:.3f} "
          f"launch={s[0]['host_launch_ms']:.2f}/{s[1]['host_launch_ms']:.2f} dev={s[0]['device_ms']:.2f}/{s[1]['device_ms']:.2f} "
          f"min_tokens={r['prefill_lookahead_min_tokens']} gc={r['gc']['in_prefill']['collections']}/{r['gc']['in_timed_region']['collections']}")
PY
timeout 900 python3 -m pytest tests/test_gemm_tile_gpu.py tests/test_engine_gpu.py -x -q -m gpu -k "row_pieces or queued_behind or lookahead or graph or golden" 2>&1 | tail -5
cat gpurun_out/a/chain_timeline.txt
