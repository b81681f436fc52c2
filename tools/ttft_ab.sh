#!/bin/bash
# The driver's exact bench command, several times in fresh processes, with and without this round's host-side changes
# (VERDICT r04 item 1): garbage-collector control (engine/host_gc.py) and the full-house warm-up.  One JSON line per run.
#   tools/ttft_ab.sh <out-dir> [runs-with] [runs-without]
out=${1:-gpurun_out/ttft_ab}; with=${2:-3}; without=${3:-2}
mkdir -p "$out"
for i in $(seq 1 "$with"); do
  extra="--no-cpu-baseline"; [ "$i" = 1 ] && extra=""
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 $extra > "$out/with_$i.json" 2> "$out/with_$i.err"
  echo "with_$i rc=$?"
done
for i in $(seq 1 "$without"); do
  MI355_GC_CONTROL=0 MI355_WARMUP_FULL_HOUSE=0 timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 \
    --no-cpu-baseline > "$out/without_$i.json" 2> "$out/without_$i.err"
  echo "without_$i rc=$?"
done
python3 - "$out" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e); continue
    print(os.path.basename(f), f"value={r['value']:.0f} ms/step={r['ms_per_step']:.4f} ttft_p50={r['ttft_p50_ms']:.2f} "
          f"ttft_max={r['ttft_max_ms']:.2f} prefill_frac={r['prefill_roofline']['frac']:.3f} "
          f"gc_prefill={r['gc']['in_prefill']} gc_timed={r['gc']['in_timed_region']}")
    for s in r.get("prefill_steps_ms", []):
        print("    ", s)
PY
