#!/bin/bash
# Register / scratch / occupancy table of the kernels (compiler remarks); compiles into /tmp, the in-tree library is not touched.
#   tools/build_report.sh [regex on the mangled kernel name]      e.g.  tools/build_report.sh 'xt_th_apply'
cd "$(dirname "$0")/../extrack_amd/csrc" || exit 1
for u in extrack_hip extrack_grad extrack_hist extrack_rev extrack_gradr extrack_reg2_f6 extrack_thgrad; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -o /tmp/xt_report_$u.o $u.hip -Rpass-analysis=kernel-resource-usage 2> /tmp/xt_report_$u.log &
done
wait
python3 - "$@" <<'PY'
import re, sys
pat = sys.argv[1] if len(sys.argv) > 1 else r'xt_ll_r2_kernelILi6ELi2ELi1|xt_ll_s2_kernelILi6ELi2ELi1|xt_grad_r2_kernelILi6ELi2ELi1ELi[67]|xt_rev_kernelILi[34]ELi2ELi1|xt_gradr_kernelILi3ELi2ELi1|xt_track_kernelILi[234]ELi2ELi1ELb[01]ELi256|xt_th_(plan|apply)_kernelILi2ELi1|xt_grad_kernelILi3ELi2ELi1ELi256|xt_hist_kernelILi2ELi1ELi256|xt_entry_kernelILi64ELi2ELi1ELi256|xt_refine_combineILi2|xt_thg_kernelILi2ELi1|xt_thg2_kernelILi2ELi1|xt_big_kernelILi2ELi1|xt_refine_componentsILi2'
for u in ("extrack_hip", "extrack_grad", "extrack_hist", "extrack_rev", "extrack_gradr", "extrack_reg2_f6", "extrack_thgrad"):
    txt = open('/tmp/xt_report_%s.log' % u).read()
    for b in txt.split('Function Name:')[1:]:
        name = b.split()[0]
        if not re.search(pat, name):
            continue
        g = lambda k: re.search(k + r':\s*(\d+)', b).group(1)
        print(name[:72].ljust(72), 'VGPR', g('VGPRs'), 'AGPR', g('AGPRs'), 'SGPR spill', g('SGPRs Spill'), 'scratch', g(r'ScratchSize \[bytes/lane\]'),
              'occ', g(r'Occupancy \[waves/SIMD\]'))
PY
