#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
export EXTRACK_TH_OVERSUB=4
for pl in 4 30; do for pt in 512 1024; do for bs in 1 4; do
  echo "== pair_lanes $pl plan threads $pt, batch x$bs"
  EXTRACK_TH_PAIR_LANES=$pl EXTRACK_TH_PLAN_THREADS=$pt EXTRACK_TH_PLAN_BS=$bs python3 tools/gpu_th_diag.py c3 c2 c1 2>&1 | grep "^C"
done; done; done
echo "== phase profile pair lanes 30"
EXTRACK_TH_PAIR_LANES=30 EXTRACK_HIP_LIB=$R/build/libextrack_hip_prof.so python3 tools/gpu_th_diag.py c3 2>&1 | grep -v "^\[th\]" | grep "plan phases" | sort -t' ' -k8 -n | tail -1
