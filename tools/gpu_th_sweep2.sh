#!/bin/bash
# On the GPU box: plan-kernel phase profile (build/libextrack_hip_prof.so, -DXT_TH_PROFILE) and knob sweeps for C3 threshold fusion
R=$GRAFT_REPO_ROOT; cd $R
echo "== phase profile (block 0 = longest chunk)"
EXTRACK_HIP_LIB=$R/build/libextrack_hip_prof.so python3 tools/gpu_th_diag.py c3 2>&1 | grep -v "^\[th\]" | tail -4
for pt in 512 1024; do for bs in 1 2 4 16; do
  echo "== plan threads $pt, batch x$bs"
  EXTRACK_TH_PLAN_THREADS=$pt EXTRACK_TH_PLAN_BS=$bs python3 tools/gpu_th_diag.py c3 2>&1 | grep "^C3"
done; done
for ov in 4 8 16; do
  echo "== apply oversub $ov"
  EXTRACK_TH_OVERSUB=$ov python3 tools/gpu_th_diag.py c3 2>&1 | grep "^C3\|apply" | tail -2
done
