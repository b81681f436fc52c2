#!/bin/bash
# 2 x 2: collector control (engine/host_gc.py) x full-house warm-up, the driver's bench command, alternating, 2 runs each
out=${1:-gpurun_out/ttft_2x2}; mkdir -p "$out"
for i in 1 2; do for gc in 1 0; do for fh in 1 0; do
  MI355_GC_CONTROL=$gc MI355_WARMUP_FULL_HOUSE=$fh timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 \
    --no-cpu-baseline > "$out/gc${gc}_fh${fh}_$i.json" 2> "$out/gc${gc}_fh${fh}_$i.err"
done; done; done
python3 - "$out" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    r = json.loads(open(f).read().strip().splitlines()[-1])
    s = r["prefill_steps_ms"]
    print(os.path.basename(f), f"ms/step={r['ms_per_step']:.4f} ttft_p50={r['ttft_p50_ms']:.2f} frac={r['prefill_roofline']['frac']:.3f} "
          f"start={s[0]['launch_start_ms']:.2f} launch={s[0]['host_launch_ms']:.2f}/{s[1]['host_launch_ms']:.2f} "
          f"dev={s[0]['device_ms']:.2f}/{s[1]['device_ms']:.2f} stamp={s[0]['stamp_ms']:.2f}/{s[1]['stamp_ms']:.2f}")
PY
