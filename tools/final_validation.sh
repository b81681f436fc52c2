set -x
R=$PWD
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -x -q -m "gpu or gpu_slow" 2>&1 | tail -4 > gpurun_out/final/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > gpurun_out/final/smoke.txt
timeout 600 python bench.py 2>&1 | grep -v Warning | tail -1 > gpurun_out/final/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $R/bench.py --no-cpu-baseline > /tmp/log_kt 2>&1
tail -1 /tmp/log_kt | grep -v Warning > $R/gpurun_out/final/bench_under_kernel_trace.json
python $R/tools/prof_db.py $(find /tmp/prof_kt -name "*.db" | head -1) 40 > $R/gpurun_out/final/kernel_trace.txt
find /tmp/prof_kt -name "*kernel_stats*" | head -1 | xargs -r -I{} cp {} $R/gpurun_out/final/rocprofv3_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_$c -- python $R/bench.py --no-cpu-baseline > /tmp/log_$c 2>&1
  db=$(find /tmp/prof_$c -name "*.db" | head -1)
  python $R/tools/prof_pmc.py $db 12 > $R/gpurun_out/final/pmc_$c.txt
  python $R/tools/prof_pmc.py $db --last paged_attn_decode_kernel 560 >> $R/gpurun_out/final/pmc_$c.txt
done
cd $R; ls -la gpurun_out/final; cat gpurun_out/final/pytest_gpu.txt gpurun_out/final/smoke.txt; head -c 600 gpurun_out/final/bench.json
