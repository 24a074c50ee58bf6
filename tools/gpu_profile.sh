#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats for the default bench command.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_stats
rm -rf $OUT && mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
echo rocprof_rc=$?
find $OUT -name "*stats*" | head; 
for f in $(find $OUT -name "*kernel_stats.csv"); do head -5 $f; done
