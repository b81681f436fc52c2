#!/bin/bash
# kernel trace of a TP=2 engine run (both ranks on this GPU, gloo) with the forked seam branch on;
# tools/seam_overlap_trace.py then counts the warm launches that ran concurrently with a seam launch.
# usage: tools/seam_overlap_trace.sh r03
TAG=${1:-r03}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_seam
MI355_SEAM_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_seam -- python $R/tools/tp_on_one_gpu.py > $O/seam_overlap_run.txt 2>&1
tail -4 $O/seam_overlap_run.txt
python $R/tools/seam_overlap_trace.py /tmp/prof_seam | tee $O/seam_overlap.txt
