#!/bin/bash
# PMC counters of one script under two library builds: tools/gpu_pmc_ab.sh script.py libA.so libB.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
S=$1; shift
for lib in "$@"; do
  export EXTRACK_HIP_LIB=$R/extrack_amd/$lib
  OUT=$R/gpurun_out/pmcab_$lib; rm -rf $OUT; mkdir -p $OUT
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
             "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_FLAT"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o pmc -- python3 $S > $OUT/p$i.log 2>&1
  done
  python3 - "$OUT" "$lib" <<'PY'
import csv, glob, collections, sys
root, lib = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "xt_th_apply" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(lib, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(agg.items())))
PY
done
