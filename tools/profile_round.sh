#!/bin/bash
# Round profile set (run on the GPU box from the repo root): bench line, rocprofv3 kernel trace of the same
# command, HBM traffic of the attention kernel (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) and the SQ
# counters of the decode attention kernel.  Results under gpurun_out/<tag>/; copy what is to be judged to
# profiles/.   usage: tools/profile_round.sh r02 [bench args]
TAG=${1:-r02}; shift
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout 900 python bench.py "$@" 2>&1 | grep '^{"metric"' > $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $R/bench.py --no-cpu-baseline "$@" > /tmp/log_kt 2>&1
grep '^{"metric"' /tmp/log_kt > $O/bench_under_kernel_trace.json
db=$(find /tmp/prof_kt -name "*.db" | head -1)
python $R/tools/prof_db.py $db 40 > $O/kernel_trace.txt
python $R/tools/prof_db.py $db --last paged_attn_decode_kernel 560 >> $O/kernel_trace.txt
# the bench's last prefill step (16 x 1024 tokens): its 28 attention launches bracket 27 whole layers
python $R/tools/prof_db.py $db --window paged_attn_prefill_kernel 28 "prefill step, 16 x 1024 tokens" > $O/prefill_step_breakdown.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_$c -- python $R/bench.py --no-cpu-baseline "$@" > /tmp/log_$c 2>&1
  db=$(find /tmp/prof_$c -name "*.db" | head -1)
  python $R/tools/prof_pmc.py $db 12 > $O/pmc_$c.txt
  python $R/tools/prof_pmc.py $db --last paged_attn_decode_kernel 560 >> $O/pmc_$c.txt
done
# SQ counters of the decode attention kernel (8 SQ slots per pass)
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM --kernel-trace -d /tmp/prof_sq -- python $R/bench.py --no-cpu-baseline "$@" > /tmp/log_sq 2>&1
db=$(find /tmp/prof_sq -name "*.db" | head -1)
python $R/tools/prof_pmc.py $db --last paged_attn_decode_kernel 560 > $O/pmc_sq_attention.txt
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace -d /tmp/prof_sq2 -- python $R/bench.py --no-cpu-baseline "$@" > /tmp/log_sq2 2>&1
db=$(find /tmp/prof_sq2 -name "*.db" | head -1)
python $R/tools/prof_pmc.py $db --last paged_attn_decode_kernel 560 >> $O/pmc_sq_attention.txt
cd $R
ls -la $O; cat $O/bench.json; tail -4 $O/kernel_trace.txt; cat $O/pmc_sq_attention.txt; tail -3 $O/pmc_FETCH_SIZE.txt; tail -3 $O/pmc_WRITE_SIZE.txt
