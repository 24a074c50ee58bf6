#!/bin/bash
# Runs on the GPU box (via gpurun): PMC passes (each its own run, no tracing domains) for the bench command.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
rm -rf $OUT && mkdir -p $OUT
cd $R
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(root + "/summary.txt", "w") as out:
    for k, d in agg.items():
        for c, v in sorted(d.items()):
            line = "%-42s %-24s n=%d mean=%.6g" % (k, c, len(v), sum(v) / len(v))
            print(line); out.write(line + "\n")
PY
