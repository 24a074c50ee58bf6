import os, sys
import numpy as np
sys.path.insert(0, ".")
from extrack_amd import gradient, synth, tracking as T
Ds, Tm, Fs = [0.0, 0.25], np.array([[0.9, 0.1], [0.1, 0.9]]), [0.6, 0.4]
pg = T.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[1e-3, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.6], estimated_transition_rates=0.1)
lst = [synth.brownian_tracks(40 + (L % 7) * 30, L, Ds, Tm, Fs, seed=300 + L) for L in range(3, 81)]
names = gradient.free_names(pg)
tf = (0.2, 120, 2000)
ts = T.TrackSet(lst)
v, g = gradient.objective_and_gradient(pg, ts, 0.02, [1], 2, 1, 6, names=names, threshold_fusion=tf)
v0 = -ts.loglik_th(T._objective_model(pg, ts, 0.02, [1], None, 2, 1, 6, 1), *tf)
ts.close()
parts = []
for half in (lst[:40], lst[40:]):
    ts = T.TrackSet(half, min_len=3, max_len=80)
    parts.append(gradient.objective_and_gradient(pg, ts, 0.02, [1], 2, 1, 6, names=names, threshold_fusion=tf))
    ts.close()
vs, gs = parts[0][0] + parts[1][0], parts[0][1] + parts[1][1]
print("th: value vs loglik_th", abs(v - v0) / abs(v0), "halves", abs(vs - v) / abs(v), np.abs(gs - g).max() / np.abs(g).max())
