#!/usr/bin/env python3
"""configs[4] posteriors (5e5 tracks x 60, 4 states, frame_len 5) and the 4-state likelihood through the general kernel (frame_len 4, 5, 6):
kernel times, two rounds.  (Used in round 3 for the waves-per-SIMD A/B of the posterior kernels and for the shift-register addressing
experiment)  usage: gpu_pred_c5.py [scale]"""
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
res = {}
for rnd in range(2):
    for sh in ("-",):
        ts = T.TrackSet([Cs])
        m5 = T._objective_model(p, ts, 0.02, [1.0], None, 4, 1, 5, 1)
        for _ in range(2):
            pr = ts.predict(m5)[0]
        kp, info = ts.ctx.last_kernel_ms(), ts.ctx.last_launch_info()
        out = {"pred": pr[:2000].copy()}
        line = "predict F=5: kernel %.1f ms (lds %d, blocks/CU %d)" % (kp, info["lds_bytes"], info["blocks_per_cu"])
        for F in (4, 5, 6):
            m = T._objective_model(p, ts, 0.02, [1.0], None, 4, 1, F, 1)
            for _ in range(3):
                ll = ts.loglik(m)
            out["ll%d" % F] = ll
            line += " | LL F=%d %.2f ms" % (F, ts.ctx.last_kernel_ms())
        print(line, flush=True)
        ts.close()
        res[sh] = out
