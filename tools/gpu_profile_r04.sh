#!/bin/bash
# Runs on the GPU box (via gpurun): round-4 profiles of the default bench command (headline kernel + the `extra` block: configs[2],
# configs[4], gradient, threshold fusion).  1) kernel-trace + stats; 2) PMC passes, one counter set per run (no tracing domains).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04prof
rm -rf $OUT && mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fits > $OUT/bench_trace.log 2>&1
echo "trace rc=$?"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fits > $OUT/p$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys, os
root = sys.argv[1]
# ---- PMC summary per kernel
agg = collections.defaultdict(lambda: collections.defaultdict(list))
rows = []
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "xt_" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]:
            rows.append((r["Kernel_Name"].split("(")[0][:64], int(r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"])))
# the bench also runs small-dataset fits (hundreds of tiny launches of the same kernels): per kernel keep the launches whose grid is at
# least half the largest one seen, i.e. the full-size BASELINE configurations
gmax = collections.defaultdict(int)
for k, g, c, v in rows:
    gmax[k] = max(gmax[k], g)
for k, g, c, v in rows:
    if 2 * g >= gmax[k]:
        agg[k][c].append(v)
with open(root + "/pmc_summary.txt", "w") as out:
    for k, d in sorted(agg.items()):
        for c, v in sorted(d.items()):
            out.write("%-66s %-24s n=%d mean=%.6g\n" % (k, c, len(v), sum(v) / len(v)))
        g = lambda c: (sum(d[c]) / len(d[c])) if c in d and d[c] else None
        if g("SQ_ACTIVE_INST_VALU") and g("GRBM_GUI_ACTIVE"):
            cyc = g("GRBM_GUI_ACTIVE") / 8
            out.write("%-66s %-24s %.3f   (SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / kernel cycles)\n" % (k, "=> VALU busy", g("SQ_ACTIVE_INST_VALU") * 4 / 1024 / cyc))
        if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
            out.write("%-66s %-24s %.3f\n" % (k, "=> LDS conflict share", g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")))
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            out.write("%-66s %-24s %.1f MB   (2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes; gfx950 correction)\n" % (k, "=> HBM traffic/launch", (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / 1e6))
# ---- the headline workload alone: xt_ll_r2_kernel<6,2,1> also serves the 1e7-track c4 run of `scaling_runs` and the full-size fits; the
# 4 launches that open each pass (1e6 tracks x 30: the bench's clock-settle launches of the headline workload) are that workload alone
with open(root + "/pmc_summary.txt", "a") as out:
    per = {}
    for f in sorted(glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("void xt_ll_r2_kernel<6, 2, 1>")]
        for c in set(r["Counter_Name"] for r in rows):
            rr = sorted([r for r in rows if r["Counter_Name"] == c], key=lambda r: int(r["Dispatch_Id"]))[:4]
            per[c] = sum(float(r["Counter_Value"]) for r in rr) / len(rr)
    out.write("\nHEADLINE WORKLOAD ONLY (first 4 launches of every pass: 1e6 tracks x 30, the workload of the bench's timed region)\n")
    for c, v in sorted(per.items()):
        out.write("%-66s %-24s n=4 mean=%.6g\n" % ("xt_ll_r2_kernel<6, 2, 1> [1e6 x 30]", c, v))
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        out.write("%-66s %-24s %.1f MB   (2 x FETCH_SIZE + WRITE_SIZE; algorithmic: 480.0 MB)\n" % ("xt_ll_r2_kernel<6, 2, 1> [1e6 x 30]", "=> HBM traffic/launch", (2 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024 / 1e6))
    if "SQ_ACTIVE_INST_VALU" in per and "GRBM_GUI_ACTIVE" in per:
        out.write("%-66s %-24s %.3f\n" % ("xt_ll_r2_kernel<6, 2, 1> [1e6 x 30]", "=> VALU busy", per["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (per["GRBM_GUI_ACTIVE"] / 8)))
# ---- the headline kernel's timed launches only (warm-up launches dropped)
for f in glob.glob(root + "/trace/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("void xt_ll_r2_kernel<6, 2, 1>")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows]
    big = [d for d in dur if d > 1e6]          # the 1e6-track launches (the gradient section launches small ones too)
    timed = big[27:47]                         # bench.py: 24 clock-settle launches + 3 warm-up steps, then the 20 timed launches
    with open(root + "/headline_timed_launches.txt", "w") as out:
        out.write("xt_ll_r2_kernel<6,2,1>: %d launches of the 1e6-track bucket in the trace; the 20 timed ones (after 24 clock-settle launches + 3 warm-up steps):\n" % len(big))
        out.write("  mean %.1f us  min %.1f us  max %.1f us\n" % (sum(timed) / len(timed) / 1e3, min(timed) / 1e3, max(timed) / 1e3))
        out.write("  all (us): " + " ".join("%.0f" % (d / 1e3) for d in big) + "\n")
PY
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
grep "^{" $OUT/bench_trace.log | tail -1 > $OUT/bench_line.json
cat $OUT/headline_timed_launches.txt; head -14 $OUT/kernel_stats.csv
