# Round-end validation on the GPU box (run from the repo root): the driver's own test selection with its durations,
# smoke(), the default bench line, a rocprofv3 kernel trace of the same command (per-kernel averages, the prefill step's
# timeline and edges), and the FETCH_SIZE / WRITE_SIZE passes.  SLOW=1 adds the gpu_slow twins to the test run.
set -x
R=$PWD
mkdir -p gpurun_out/final
SEL="gpu"; [ -n "$SLOW" ] && SEL="gpu or gpu_slow"
timeout 1500 python -m pytest tests -q -m "$SEL" --durations=15 2>&1 | tail -24 > gpurun_out/final/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > gpurun_out/final/smoke.txt
timeout 600 python bench.py 2>&1 | grep '^{"metric"' > gpurun_out/final/bench.json
for i in 1 2 3 4 5; do  # the driver's exact command, five fresh processes (the first with the CPU baseline, as the driver runs it)
  extra="--no-cpu-baseline"; [ "$i" = 1 ] && extra=""
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 $extra 2>/dev/null | grep '^{"metric"' > gpurun_out/final/bench_driver_cmd_$i.json
done
bash tools/serving_round.sh final > gpurun_out/final/serving_summary.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $R/bench.py --no-cpu-baseline > /tmp/log_kt 2>&1
grep '^{"metric"' /tmp/log_kt > $R/gpurun_out/final/bench_under_kernel_trace.json
db=$(find /tmp/prof_kt -name "*.db" | head -1)
python $R/tools/prof_db.py $db 40 > $R/gpurun_out/final/kernel_trace.txt
python $R/tools/prof_db.py $db --last paged_attn_decode_kernel 560 >> $R/gpurun_out/final/kernel_trace.txt
python $R/tools/prof_db.py $db --window paged_attn_prefill_kernel 28 "prefill step, 16 x 1024 tokens" > $R/gpurun_out/final/prefill_step_breakdown.txt
python $R/tools/prof_db.py $db --edges paged_attn_prefill_kernel 28 > $R/gpurun_out/final/prefill_step_edges.txt
python $R/tools/prof_db.py $db --layers paged_attn_prefill_kernel 28 2 > $R/gpurun_out/final/prefill_steps_layer_by_layer.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_$c -- python $R/bench.py --no-cpu-baseline > /tmp/log_$c 2>&1
  db=$(find /tmp/prof_$c -name "*.db" | head -1)
  python $R/tools/prof_pmc.py $db 12 > $R/gpurun_out/final/pmc_$c.txt
  python $R/tools/prof_pmc.py $db --last paged_attn_decode_kernel 560 >> $R/gpurun_out/final/pmc_$c.txt
done
cd $R
python tools/attn_traffic_json.py gpurun_out/final/pmc_FETCH_SIZE.txt gpurun_out/final/pmc_WRITE_SIZE.txt \
  gpurun_out/final/bench_under_kernel_trace.json gpurun_out/final/attn_traffic.json > /dev/null
ls -la gpurun_out/final; cat gpurun_out/final/pytest_gpu.txt gpurun_out/final/smoke.txt; head -c 700 gpurun_out/final/bench.json; cat gpurun_out/final/prefill_step_breakdown.txt; tail -2 gpurun_out/final/pmc_FETCH_SIZE.txt; tail -1 gpurun_out/final/pmc_WRITE_SIZE.txt
