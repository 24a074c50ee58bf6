#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python3 -m pytest tests/test_hip_th_parity.py -x -q 2>&1 | tail -3
for bs in 0 1 2; do
echo "== plan_bs $bs"
EXTRACK_TH_PLAN_BS=$bs python3 tools/gpu_th_diag.py c3 c2 c1 2>&1 | grep "^C" 
done
echo "== plan threads 512 forced, bs 0"
EXTRACK_TH_PLAN_THREADS=512 python3 tools/gpu_th_diag.py c3 2>&1 | grep "^C"
echo "== phase profile C3, C1 (all chunks of block 0)"
EXTRACK_HIP_LIB=$R/build/libextrack_hip_prof.so python3 tools/gpu_th_diag.py c3 2>&1 | grep "plan phases" > gpurun_out/phases_c3.txt
EXTRACK_HIP_LIB=$R/build/libextrack_hip_prof.so python3 tools/gpu_th_diag.py c1 2>&1 | grep "plan phases" > gpurun_out/phases_c1.txt
tail -3 gpurun_out/phases_c3.txt; tail -2 gpurun_out/phases_c1.txt
