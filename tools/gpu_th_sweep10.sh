#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for bs in 0 1; do
echo "== plan_bs $bs"
EXTRACK_TH_PLAN_BS=$bs python3 tools/gpu_th_diag.py c3 c2 c1 2>&1 | grep "^C" 
done
echo "== phase profile C3, C1 (all chunks of block 0)"
EXTRACK_HIP_LIB=$R/build/libextrack_hip_prof.so python3 tools/gpu_th_diag.py c3 2>&1 | grep "plan phases" > gpurun_out/phases_c3.txt
EXTRACK_HIP_LIB=$R/build/libextrack_hip_prof.so python3 tools/gpu_th_diag.py c1 2>&1 | grep "plan phases" > gpurun_out/phases_c1.txt
EXTRACK_TH_PLAN_BS=1 EXTRACK_HIP_LIB=$R/build/libextrack_hip_prof.so python3 tools/gpu_th_diag.py c3 2>&1 | grep "plan phases" > gpurun_out/phases_c3_bs1.txt
tail -3 gpurun_out/phases_c3.txt; tail -3 gpurun_out/phases_c3_bs1.txt; tail -2 gpurun_out/phases_c1.txt
