import os, sys
import numpy as np
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/extrack_amd") else ".")
from extrack_amd import gradient, synth, tracking as T
def stats(name, lst, pg, S, F):
    ts = T.TrackSet(lst)
    model = T._objective_model(pg, ts, 0.02, [1], None, S, 1, F, 1)
    ts.loglik_th(model, 0.2, 120, 2000)
    allg, mx = [], []
    for b, (N, L, D, KS) in enumerate(ts.ctx.buckets):
        nch = (N + 1999) // 2000
        for c in range(0, nch, max(1, nch // 3)):
            gs = []
            for t in range(2, L - 1):
                nE, groups = ts.ctx.th_plan_step(b, c, t)
                gs.append(len(groups))
            if gs:
                allg += gs; mx.append(max(gs))
    allg = np.array(allg); mx = np.array(mx)
    print(name, "steps sampled", len(allg), "nG mean %.1f median %d p90 %d max %d | per-chunk max: mean %.1f p50 %d p90 %d max %d" % (allg.mean(), np.median(allg), np.percentile(allg, 90), allg.max(), mx.mean(), np.median(mx), np.percentile(mx, 90), mx.max()))
    ts.close()
Cs = synth.brownian_tracks(100000, 30, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=0)
pg = T.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[1e-3, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.6], estimated_transition_rates=0.1)
stats("C2", [Cs], pg, 2, 6)
sizes = synth.bucket_sizes_geometric(int(2e5), list(range(5, 51)), 0.9)
Tm = np.array([[0.9, 0.07, 0.03], [0.05, 0.9, 0.05], [0.03, 0.07, 0.9]])
tracks = {str(L): synth.brownian_tracks(n, L, [0.0, 0.04, 0.25], Tm, [0.3, 0.3, 0.4], seed=L) for L, n in sizes.items() if n > 0}
p = T.generate_params(nb_states=3, LocErr_type=1, estimated_Ds=[0.0001, 0.04, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.3, 0.3], estimated_transition_rates=0.06)
_, lst, _ = T.engine.sort_buckets(tracks)
stats("C3", lst, p, 3, 6)
