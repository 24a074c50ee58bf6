#!/bin/bash
# On the GPU box: kernel-trace stats of tools/gpu_th_diag.py (plan / apply split per dataset)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in "$@"; do
  OUT=$R/gpurun_out/prof_thdiag_$w
  rm -rf $OUT && mkdir -p $OUT
  (cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o th -- python3 tools/gpu_th_diag.py $w > $R/gpurun_out/thdiag_$w.log 2>&1)
  echo "== $w rc=$?"
  grep -v "^W2\|rocprof" $R/gpurun_out/thdiag_$w.log | tail -8
  for f in $(find $OUT -name "*kernel_stats.csv"); do cut -d, -f1-7 $f | head -8; done
done
