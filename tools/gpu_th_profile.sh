#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the threshold-fusion evaluation at BASELINE sizes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_th
rm -rf $OUT && mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o th -- python3 tools/gpu_th.py > $R/gpurun_out/prof_th.log 2>&1
echo rocprof_rc=$?
for f in $(find $OUT -name "*kernel_stats.csv"); do head -12 $f; done
