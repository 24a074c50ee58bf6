#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python3 -m pytest tests/test_hip_th_parity.py -x -q 2>&1 | tail -2
for bs in 8 16 32 64 138; do
  echo "== shared rows, batch $bs"
  EXTRACK_TH_PLAN_BS=$bs python3 tools/gpu_th_diag.py c3 2>&1 | grep "^C"
done
echo "== defaults"
python3 tools/gpu_th_diag.py c3 c2 c1 2>&1 | grep "^C"
echo "== phase profile C3 (longest chunk), C1"
EXTRACK_HIP_LIB=$R/build/libextrack_hip_prof.so python3 tools/gpu_th_diag.py c3 2>&1 | grep "plan phases" | sort -t' ' -k13 -n | tail -1
EXTRACK_HIP_LIB=$R/build/libextrack_hip_prof.so python3 tools/gpu_th_diag.py c1 2>&1 | grep "plan phases" | tail -2
