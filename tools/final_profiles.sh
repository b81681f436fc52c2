R=$PWD
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $R/bench.py --no-cpu-baseline > /tmp/log_kt 2>&1
grep '^{"metric"' /tmp/log_kt > $R/gpurun_out/final/bench_under_kernel_trace.json
db=$(find /tmp/prof_kt -name "*.db" | head -1)
python $R/tools/prof_db.py $db 40 > $R/gpurun_out/final/kernel_trace.txt
python $R/tools/prof_db.py $db --last paged_attn_decode_kernel 560 >> $R/gpurun_out/final/kernel_trace.txt
find /tmp/prof_kt -type f | head -20 > $R/gpurun_out/final/prof_files.txt
f=$(find /tmp/prof_kt -name "*kernel_stats*" | head -1); [ -n "$f" ] && head -40 "$f" > $R/gpurun_out/final/rocprofv3_kernel_stats.csv
cd $R
python tools/kbench.py 2>&1 | grep -v Warn > gpurun_out/final/kbench.txt
KBENCH_ONLY=prefill python tools/kbench.py 2>&1 | grep -v Warn | tail -1 >> gpurun_out/final/kbench.txt
timeout 200 python tools/comm_bench.py 2>&1 | grep "us per call" > gpurun_out/final/comm_bench.txt
tail -3 gpurun_out/final/kernel_trace.txt; cat gpurun_out/final/prof_files.txt; cat gpurun_out/final/comm_bench.txt; tail -5 gpurun_out/final/kbench.txt
