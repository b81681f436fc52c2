#!/bin/bash
mkdir -p gpurun_out/j
timeout 1500 python3 -m pytest tests/test_engine_gpu.py -x -q -m gpu > gpurun_out/j/pytest_engine.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/j/pytest_engine.txt | cut -c1-300
for i in 1 2 3 4 5; do
  extra="--no-cpu-baseline"; [ "$i" = 1 ] && extra=""
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 $extra 2>/dev/null | grep '^{"metric"' > gpurun_out/j/bench_driver_cmd_$i.json
done
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/j/bench_driver_cmd_*.json")):
    r = json.loads(open(f).read().strip().splitlines()[-1]); s = r["prefill_steps_ms"]
    print(f, f"value={r['value']:.0f} ms/step={r['ms_per_step']:.4f} ttft_p50={r['ttft_p50_ms']:.2f} max={r['ttft_max_ms']:.2f} frac={r['prefill_roofline']['frac']:.3f} "
          f"launch={s[0]['host_launch_ms']:.2f}/{s[1]['host_launch_ms']:.2f} dev={s[0]['device_ms']:.2f}/{s[1]['device_ms']:.2f} gc={r['gc']['in_prefill']['collections']}/{r['gc']['in_timed_region']['full']}")
PY
bash tools/prefill_attn_pmc.sh gpurun_out/j/prefill_attention_pmc.txt > /dev/null 2>&1; cat gpurun_out/j/prefill_attention_pmc.txt | cut -c1-200
