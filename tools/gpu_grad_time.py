#!/usr/bin/env python3
"""Times the likelihood + gradient kernel on the C2 dataset (1e6 x 30, 2 states, 7 free parameters) and on C3 (3 states, 13)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import gradient, synth, tracking as T
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
Cs = synth.brownian_tracks(int(1e6 * scale), 30, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=0)
p = T.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[0.001, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.6], estimated_transition_rates=0.1)
names = gradient.free_names(p)
ts = T.TrackSet([Cs])
for F in (6, 4):
    for _ in range(2):
        v, g = gradient.objective_and_gradient(p, ts, 0.02, [1], 2, 1, F, names=names)
    t0 = time.perf_counter(); v, g = gradient.objective_and_gradient(p, ts, 0.02, [1], 2, 1, F, names=names); dt = time.perf_counter() - t0
    ll = ts.loglik(T._objective_model(p, ts, 0.02, [1], None, 2, 1, F, 1))
    print("C2 F=%d: grad kernel %.2f ms (wall %.2f ms), %d dirs, LL kernel %.2f ms, launch %s" % (F, ts.ctx.last_grad_ms(), dt * 1e3, len(names), ts.ctx.last_kernel_ms(), ts.ctx.last_launch_info()))
ts.close()
sizes = synth.bucket_sizes_geometric(int(1e6 * scale), list(range(5, 51)), 0.9)
Tm = np.array([[0.9, 0.07, 0.03], [0.05, 0.9, 0.05], [0.03, 0.07, 0.9]])
tracks = {str(L): synth.brownian_tracks(n, L, [0.0, 0.04, 0.25], Tm, [0.3, 0.3, 0.4], seed=L) for L, n in sizes.items() if n > 0}
p = T.generate_params(nb_states=3, LocErr_type=1, estimated_Ds=[0.0001, 0.04, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.3, 0.3], estimated_transition_rates=0.06)
names = gradient.free_names(p)
_, lst, _ = T.engine.sort_buckets(tracks)
ts = T.TrackSet(lst)
for F in (4, 6):
    for _ in range(2):
        v, g = gradient.objective_and_gradient(p, ts, 0.02, [1], 3, 1, F, names=names)
    ll = ts.loglik(T._objective_model(p, ts, 0.02, [1], None, 3, 1, F, 1))
    print("C3 F=%d: grad kernel %.2f ms, %d dirs, LL kernel %.2f ms, launch %s" % (F, ts.ctx.last_grad_ms(), len(names), ts.ctx.last_kernel_ms(), ts.ctx.last_launch_info()))
ts.close()
