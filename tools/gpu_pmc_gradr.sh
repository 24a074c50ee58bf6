#!/bin/bash
# PMC passes for a gradient kernel on C3 (quarter size); usage: tools/gpu_pmc_gradr.sh <frame_len> [kernel-name substring: gradr | rev]  (EXTRACK_GRAD_PATH selects the path)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_gradr
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS"; do
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_gradr/$(echo $set | cut -c1-8) -o pmc -- python3 tools/gpu_grad_c3.py 0.25 $1 > /dev/null 2>&1
done
python3 - "${2:-gradr}" <<PY
import csv,glob,collections,sys
agg=collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_gradr/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"] and "project" not in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k:sum(v)/len(v) for k,v in agg.items()}
for k,v in sorted(m.items()): print("%-24s %.4g" % (k, v))
cyc=m["GRBM_GUI_ACTIVE"]/8
print("VALU busy %.3f  LDS conflict share %.3f  HBM MB/launch %.0f  wave-cycles per kernel cycle per SIMD %.2f  wait_any share %.2f" % (m["SQ_ACTIVE_INST_VALU"]*4/1024/cyc, m["SQ_LDS_BANK_CONFLICT"]/m["SQ_LDS_IDX_ACTIVE"], (2*m["FETCH_SIZE"]+m["WRITE_SIZE"])*1024/1e6, m["SQ_WAVE_CYCLES"]*4/1024/cyc/4, m["SQ_WAIT_ANY"]/m["SQ_WAVE_CYCLES"] if "SQ_WAIT_ANY" in m else -1))
PY
