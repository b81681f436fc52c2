#!/bin/bash
# SQ / LDS counters of the tile GEMM on one shape: tools/gemm_pmc.sh <outdir> M N K [variant]
O=$PWD/$1; shift
R=$PWD
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
tag="$1x$2x$3_v${4:-0}"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace -d /tmp/p1_$tag -- python $R/tools/gemm_prof.py "$@" > /tmp/log1_$tag 2>&1
db=$(find /tmp/p1_$tag -name "*.db" | head -1)
python $R/tools/prof_pmc.py $db --last gemm_tile_kernel 8 > $O/pmc_$tag.txt
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace -d /tmp/p2_$tag -- python $R/tools/gemm_prof.py "$@" > /tmp/log2_$tag 2>&1
db=$(find /tmp/p2_$tag -name "*.db" | head -1)
python $R/tools/prof_pmc.py $db --last gemm_tile_kernel 8 >> $O/pmc_$tag.txt
tail -3 /tmp/log2_$tag >> $O/pmc_$tag.txt
cat $O/pmc_$tag.txt
