"""Time of the frozen-plan gradient of the threshold-fusion objective (extrack_loglik_th_grad) against the evaluation it extends and the
finite differences it replaces: C2 (1e6 x 30, 2 states, 7 free parameters) and C3 (1e6 tracks, 3 states, 46 buckets, 13 free parameters).
python tools/gpu_thgrad_time.py [c2|c3|both] [scale]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import gradient, synth, tracking as T

which = sys.argv[1] if len(sys.argv) > 1 else "both"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0


def run(name, lst, pg, S, F):
    names = gradient.free_names(pg)
    ts = T.TrackSet(lst)
    tf = (0.2, 120, 2000)
    model = T._objective_model(pg, ts, 0.02, [1], None, S, 1, F, 1)
    for _ in range(3):
        v0 = ts.loglik_th(model, *tf)
    t0 = time.perf_counter()
    for _ in range(5):
        v0 = ts.loglik_th(model, *tf)
    t_ll = (time.perf_counter() - t0) / 5
    k_ll = ts.ctx.last_kernel_ms()
    for _ in range(2):
        v, g = gradient.objective_and_gradient(pg, ts, 0.02, [1], S, 1, F, names=names, threshold_fusion=tf)
    t0 = time.perf_counter()
    for _ in range(5):
        v, g = gradient.objective_and_gradient(pg, ts, 0.02, [1], S, 1, F, names=names, threshold_fusion=tf)
    t_g = (time.perf_counter() - t0) / 5
    print("%s: objective %.3f ms (kernels %.3f) | objective + gradient (%d parameters) %.3f ms (kernels %.3f ms) = %.1f evaluations; finite differences: %.1f ms | value diff %.2e | launch %s"
          % (name, t_ll * 1e3, k_ll, len(names), t_g * 1e3, ts.ctx.last_grad_ms(), t_g / t_ll, (len(names) + 1) * t_ll * 1e3, abs(v + v0) / abs(v0), ts.ctx.last_launch_info()), flush=True)
    ts.close()


if which in ("c2", "both"):
    Cs = synth.brownian_tracks(int(1e6 * scale), 30, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=0)
    pg = T.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[1e-3, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.6], estimated_transition_rates=0.1)
    run("C2 %g x 30" % (1e6 * scale), [Cs], pg, 2, 6)
    del Cs
if which in ("c3", "both"):
    Ds, Tm = [0.0, 0.04, 0.25], np.array([[0.9, 0.07, 0.03], [0.05, 0.9, 0.05], [0.03, 0.07, 0.9]])
    sizes = synth.bucket_sizes_geometric(int(1e6 * scale), list(range(5, 51)), 0.9)
    lst = [synth.brownian_tracks(n, L, Ds, Tm, [0.3, 0.3, 0.4], seed=1000 + L) for L, n in sizes.items() if n > 0]
    pg = T.generate_params(nb_states=3, LocErr_type=1, estimated_Ds=[1e-4, 0.04, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.3, 0.3], estimated_transition_rates=0.06)
    run("C3 %g tracks, 46 buckets" % (1e6 * scale), lst, pg, 3, 6)
