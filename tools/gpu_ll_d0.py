#!/usr/bin/env python3
"""LL kernel time on C2 data as a function of D0 (0 vs 0.001) and of what ran before."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import gradient, synth, tracking as T
Cs = synth.brownian_tracks(1000000, 30, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=0)
ts = T.TrackSet([Cs])
def P(d0):
    return T.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[d0, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.6], estimated_transition_rates=0.1)
for d0 in (0.0, 1e-3, 0.0, 1e-5, 1e-2):
    p = P(d0)
    m = T._objective_model(p, ts, 0.02, [1], None, 2, 1, 6, 1)
    for _ in range(3):
        ll = ts.loglik(m)
    print("D0=%g  LL %.6f  kernel %.3f ms" % (d0, ll, ts.ctx.last_kernel_ms()))
p = P(1e-3)
v, g = gradient.objective_and_gradient(p, ts, 0.02, [1], 2, 1, 6)
m = T._objective_model(P(0.0), ts, 0.02, [1], None, 2, 1, 6, 1)
ll = ts.loglik(m)
print("after grad: D0=0 kernel %.3f ms" % ts.ctx.last_kernel_ms())
ll = ts.loglik(m)
print("again: D0=0 kernel %.3f ms" % ts.ctx.last_kernel_ms())
