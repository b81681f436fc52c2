#!/bin/bash
# First thing on a FRESH box (cold page cache, nothing of the library touched yet): the driver's exact bench command with
# last round's host behaviour (no collector control, no full-house warm-up), then with this round's.  One JSON line each.
out=${1:-gpurun_out/ttft_cold}
mkdir -p "$out"
MI355_GC_CONTROL=0 MI355_WARMUP_FULL_HOUSE=0 timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline \
  > "$out/cold_without_1.json" 2> "$out/cold_without_1.err"; echo "cold_without_1 rc=$?"
MI355_GC_CONTROL=0 MI355_WARMUP_FULL_HOUSE=0 timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline \
  > "$out/warm_without_2.json" 2> "$out/warm_without_2.err"; echo "warm_without_2 rc=$?"
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline \
  > "$out/warm_with_3.json" 2> "$out/warm_with_3.err"; echo "warm_with_3 rc=$?"
python3 - "$out" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json")), key=os.path.getmtime):
    r = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), f"value={r['value']:.0f} ms/step={r['ms_per_step']:.4f} ttft_p50={r['ttft_p50_ms']:.2f} "
          f"ttft_max={r['ttft_max_ms']:.2f} prefill_frac={r['prefill_roofline']['frac']:.3f} gc_prefill={r['gc']['in_prefill']}")
    for s in r.get("prefill_steps_ms", []):
        print("    ", s)
PY
