#!/bin/bash
# Rebuild libextrack_hip.so with the resource-usage remarks and print a table of the fast-path kernels.
cd /root/repo/extrack_amd/csrc || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Xclang -target-feature -Xclang -load-store-opt -o ../libextrack_hip.so extrack_hip.hip -Rpass-analysis=kernel-resource-usage 2> /tmp/build.log
rc=$?
grep -E " error" /tmp/build.log | head -10
python3 - "$@" <<'PY'
import re, sys
pat = sys.argv[1] if len(sys.argv) > 1 else r'_Z15xt_ll_s2_kernelILi(\d)ELi2ELi(\d)EE'
txt = open('/tmp/build.log').read()
for b in txt.split('Function Name:')[1:]:
    name = b.split()[0]
    if not re.match(pat, name): continue
    g = lambda k: re.search(k + r':\s*(\d+)', b).group(1)
    print(name[:60].ljust(60), 'VGPR', g('VGPRs'), 'SGPR', g('SGPRs'), 'scratch', g(r'ScratchSize \[bytes/lane\]'), 'occ', g(r'Occupancy \[waves/SIMD\]'))
PY
exit $rc
