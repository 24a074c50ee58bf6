#!/usr/bin/env python3
"""Instruction mix of the headline kernel's steady-state step, taken from the ISA the CURRENT sources compile to (gfx950): bench.py
reads the result (profiles/isa_mix.json) for its fp64-issue figures instead of hand-copied constants.

    python tools/isa_mix.py            # compiles xt_ll_r2_kernel<6,2,1> into /tmp, writes profiles/isa_mix.json

Method: the likelihood-only register-resident kernel (csrc/xt_reg2.h) unrolls its step loop over the F - 1 exchange phases in three
variants; the basic blocks with >= 50 fp64 vector instructions and the table reads are the steps, and the 2 (F - 1) of them with the fewest
instructions are the steady-state ones (well-scaled model: zero-free, lazily re-normalised).  Counts are per wave-step (one wavefront = 64 / 2^(F-1) tracks,
one position)."""
import json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, D, K = 6, 2, 1
src = '''#include "xt_host.h"
#include "xt_reg2.h"
__global__ void __launch_bounds__(64 * XT_F2_WAVES) xt_ll_r2_kernel_mix(XtKernelArgs a)
{
    DevCtx cx;
    XtGradArgs ga;
    ga.dblob = nullptr;
    ga.gpartials = nullptr;
    xt_r2_body<%d, %d, %d, 0>(a, ga, cx);
}
''' % (F, D, K)
tmp = tempfile.mkdtemp(prefix="xt_isa_")
open(os.path.join(tmp, "mix.hip"), "w").write(src)
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "extrack_amd", "csrc"), "-c", "mix.hip",
                       "-o", "mix.o", "--save-temps"], cwd=tmp, stderr=subprocess.DEVNULL)
asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
blocks, cur = [], None
for line in open(os.path.join(tmp, asm)):
    if line.startswith(".LBB") or line.startswith("_Z"):
        cur = dict(name=line.split(":")[0], f64=0, fma=0, valu32=0, lds=0, salu=0, vmem=0, trans=0)
        blocks.append(cur)
        continue
    if cur is None or not line.startswith("\t") or line.startswith("\t.") or line.startswith("\t;"):
        continue
    op = line.split()[0]
    if op.startswith("v_"):
        if "_f64" in op or op in ("v_lshl_add_u64",):
            cur["f64"] += 1
            if "fma" in op or "fmac" in op:
                cur["fma"] += 1
            if op.startswith("v_rcp") or op.startswith("v_sqrt") or op.startswith("v_rsq"):
                cur["trans"] += 1
        else:
            cur["valu32"] += 1
    elif op.startswith("ds_"):
        cur["lds"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
    elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        cur["vmem"] += 1
steps = sorted([b for b in blocks if b["f64"] >= 50 and b["lds"] >= 4], key=lambda b: b["f64"] + b["valu32"])
ss = steps[:2 * (F - 1)]  # the step loop is inlined twice (before / after the stay-in-FOV factor sets in): both copies of the F - 1 phases
n = float(len(ss))
mix = {k: sum(b[k] for b in ss) / n for k in ("f64", "fma", "valu32", "lds", "salu", "vmem", "trans")}
tpw = 64 >> (F - 1)
out = {"kernel": "xt_ll_r2_kernel<%d,%d,%d> (steady-state step, mean over the %d exchange phases)" % (F, D, K, F - 1),
       "tracks_per_wave": tpw, "fp64_valu_per_wave_step": mix["f64"], "fp64_fma_per_wave_step": mix["fma"], "fp64_transcendental_per_wave_step": mix["trans"],
       "valu32_per_wave_step": mix["valu32"], "lds_per_wave_step": mix["lds"], "salu_per_wave_step": mix["salu"], "vmem_per_wave_step": mix["vmem"],
       "flop_per_wave_step": 64 * (mix["f64"] + mix["fma"]),
       "issue_cycles_per_wave_step": 4 * mix["f64"] + 2 * mix["valu32"],
       "method": "tools/isa_mix.py: hipcc --save-temps of the current csrc/xt_reg2.h; an fp64 vector instruction issues in 4 cycles per wavefront, "
                 "a 32-bit one (incl. the DPP moves of the lane exchange) in 2 (profiles/r01_valu_rates.txt); a v_permlane swap is counted as 32-bit",
       "step_blocks": [(b["name"], b["f64"], b["valu32"], b["lds"]) for b in steps]}
path = os.path.join(ROOT, "profiles", "isa_mix.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "step_blocks"}, indent=1))
print("step blocks (name, fp64, 32-bit valu, lds):", out["step_blocks"])
