#!/bin/bash
# usage: tools/gpu_pmc_cmd.sh <outdir-name> <python script> [args...]   (runs on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NAME=$1; shift
OUT=$R/gpurun_out/$NAME
rm -rf $OUT && mkdir -p $OUT
cd $R
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o pmc -- python3 "$@" > $OUT/p$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(root + "/summary.txt", "w") as out:
    for k, d in agg.items():
        if "xt_" not in k or "reduce" in k: continue
        for c, v in sorted(d.items()):
            line = "%-50s %-24s n=%d mean=%.6g" % (k, c, len(v), sum(v) / len(v))
            print(line); out.write(line + "\n")
PY
