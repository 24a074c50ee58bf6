#!/usr/bin/env python3
"""configs[4] likelihood (5e5 tracks x 60, 4 states, nb_substeps 3, frame_len 4: the entry-parallel kernel) and nb_substeps 2: kernel times.
usage: gpu_c5_ll.py [scale]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import synth, tracking as T
from extrack_amd.lmfit_compat import Parameters
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
N5, L5 = int(500000 * scale), 60
Tm = np.full((4, 4), 0.05 / 3)
Tm[np.arange(4), np.arange(4)] = 0.95
Cs = synth.brownian_tracks(N5, L5, [0.0, 0.02, 0.1, 0.5], Tm, [0.25] * 4, seed=2)
vals = dict(D0=1e-4, D1=0.02, D2=0.1, D3=0.5, LocErr=0.02, F0=.25, F1=.25, F2=.25, F3=.25, pBL=0.1)
for i in range(4):
    for j in range(4):
        if i != j:
            vals["p%d%d" % (i, j)] = 0.05 / 3
p = Parameters()
for k, v in vals.items():
    p.add(k, value=v)
ts = T.TrackSet([Cs])
for ns, F in ((3, 4), (2, 4), (2, 3), (3, 5)):
    try:
        m = T._objective_model(p, ts, 0.02, [1.0], None, 4, ns, F, 1)
        for _ in range(3):
            ll = ts.loglik(m)
        print("4 states ns=%d F=%d: kernel %.2f ms  LL %.6f  %s" % (ns, F, ts.ctx.last_kernel_ms(), ll, ts.ctx.last_launch_info()), flush=True)
    except Exception as e:
        print("ns=%d F=%d: %s" % (ns, F, str(e)[:100]))
ts.close()
