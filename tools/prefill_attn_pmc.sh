#!/bin/bash
# SQ / LDS counters of the prefill attention kernel (kbench's 16 x 1024-token case): tools/prefill_attn_pmc.sh <outfile>
R=$PWD; O=$R/$1; mkdir -p $(dirname $O)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pa1 /tmp/pa2
KBENCH_ONLY=prefill timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace -d /tmp/pa1 -- python $R/tools/kbench.py > /tmp/pa1.log 2>&1
python $R/tools/prof_pmc.py $(find /tmp/pa1 -name "*.db" | head -1) --last paged_attn_prefill_kernel 28 > $O
KBENCH_ONLY=prefill timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pa2 -- python $R/tools/kbench.py > /tmp/pa2.log 2>&1
python $R/tools/prof_pmc.py $(find /tmp/pa2 -name "*.db" | head -1) --last paged_attn_prefill_kernel 28 >> $O
grep paged_attn_prefill /tmp/pa2.log >> $O
cat $O
