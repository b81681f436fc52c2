for v in 4096 4099 4223 5001; do
  BENCH_NBLK=$v timeout 300 python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>&1 | grep -v Warning | tail -1 > /tmp/o.json
  python -c "import json; d=json.load(open('/tmp/o.json')); print($v, round(d['ms_per_step'],4), round(d['roofline']['frac'],4), round(d['roofline']['avg_launch_us'],2))"
done
