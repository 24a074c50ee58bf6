#!/usr/bin/env python3
"""Times the general likelihood + gradient kernel on C3 (1e6 tracks, 3 states, lengths 5-50, 13 free parameters).
usage: gpu_grad_c3.py [scale] [frame_lens, e.g. 4,6] [n_dirs]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import gradient, synth, tracking as T
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
Fs_ = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "4,6").split(",")]
sizes = synth.bucket_sizes_geometric(int(1e6 * scale), list(range(5, 51)), 0.9)
Tm = np.array([[0.9, 0.07, 0.03], [0.05, 0.9, 0.05], [0.03, 0.07, 0.9]])
tracks = {str(L): synth.brownian_tracks(n, L, [0.0, 0.04, 0.25], Tm, [0.3, 0.3, 0.4], seed=L) for L, n in sizes.items() if n > 0}
p = T.generate_params(nb_states=3, LocErr_type=1, estimated_Ds=[0.0001, 0.04, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.3, 0.3], estimated_transition_rates=0.06)
names = gradient.free_names(p)
nd = int(sys.argv[3]) if len(sys.argv) > 3 else len(names)
_, lst, _ = T.engine.sort_buckets(tracks)
ts = T.TrackSet(lst)
for F in Fs_:
    for _ in range(2):
        v, g = gradient.objective_and_gradient(p, ts, 0.02, [1], 3, 1, F, names=names[:nd])
    gms = ts.ctx.last_grad_ms()
    for _ in range(3):
        ll = ts.loglik(T._objective_model(p, ts, 0.02, [1], None, 3, 1, F, 1))
    print("C3 F=%d: grad kernel %.2f ms, %d dirs, LL kernel %.2f ms (fd gradient = %.1f ms), launch %s" % (
        F, gms, nd, ts.ctx.last_kernel_ms(), (nd + 1) * ts.ctx.last_kernel_ms(), ts.ctx.last_launch_info()), flush=True)
ts.close()
