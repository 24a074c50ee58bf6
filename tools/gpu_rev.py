#!/usr/bin/env python3
"""Reverse-mode gradient kernels (xt_rev.h) against the forward-mode ones and finite differences: kernel times on
C3 (1e6 tracks, 3 states, lengths 5-50), C2 (1e6 x 30, 2 states) and a 4-state set (5e5 x 60).
usage: gpu_rev.py [scale] [cases, e.g. c3f4,c3f6,c2f6,c4f4] [paths, e.g. rev,reg2]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import gradient, synth, tracking as T
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
cases = (sys.argv[2] if len(sys.argv) > 2 else "c3f4,c3f6,c2f6,c4f4").split(",")
paths = (sys.argv[3] if len(sys.argv) > 3 else "rev,reg2").split(",")


def data(case):
    if case.startswith("c3"):
        sizes = synth.bucket_sizes_geometric(int(1e6 * scale), list(range(5, 51)), 0.9)
        Tm = np.array([[0.9, 0.07, 0.03], [0.05, 0.9, 0.05], [0.03, 0.07, 0.9]])
        tr = {str(L): synth.brownian_tracks(n, L, [0.0, 0.04, 0.25], Tm, [0.3, 0.3, 0.4], seed=L) for L, n in sizes.items() if n > 0}
        p = T.generate_params(nb_states=3, LocErr_type=1, estimated_Ds=[0.0001, 0.04, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.3, 0.3], estimated_transition_rates=0.06)
        return tr, p, 3
    if case.startswith("c2"):
        tr = {"30": synth.brownian_tracks(int(1e6 * scale), 30, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=0)}
        p = T.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[0.001, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.6], estimated_transition_rates=0.1)
        return tr, p, 2
    Tm = np.full((4, 4), 0.03) + np.eye(4) * 0.88
    tr = {"60": synth.brownian_tracks(int(5e5 * scale), 60, [0.0, 0.02, 0.1, 0.4], Tm, [0.25] * 4, seed=1)}
    p = T.generate_params(nb_states=4, LocErr_type=1, estimated_Ds=[0.0001, 0.02, 0.1, 0.4], estimated_LocErr=[0.02], estimated_Fs=[0.25, 0.25, 0.25], estimated_transition_rates=0.03)
    return tr, p, 4


cache = {}
for case in cases:
    key = case[:2]
    if key not in cache:
        cache = {key: data(case)}
    tr, p, S = cache[key]
    F = int(case[3:])
    names = gradient.free_names(p)
    _, lst, _ = T.engine.sort_buckets(tr)
    ref = None
    for path in paths:
        os.environ["EXTRACK_GRAD_PATH"] = path
        ts = T.TrackSet(lst)
        for _ in range(3):
            v, g = gradient.objective_and_gradient(p, ts, 0.02, [1], S, 1, F, names=names)
        gms = ts.ctx.last_grad_ms()
        info = ts.ctx.last_launch_info()
        for _ in range(3):
            ts.loglik(T._objective_model(p, ts, 0.02, [1], None, S, 1, F, 1))
        lms = ts.ctx.last_kernel_ms()
        ts.close()
        if ref is None:
            ref = (v, g)
        err = np.abs(g - ref[1]).max() / np.abs(ref[1]).max()
        print("%s %-5s grad %8.2f ms  (%d dirs; LL %.2f ms, fd = %.1f ms)  blocks %d lds %d tpb %d occ %d  | vs first path: obj %.1e grad %.1e" % (
            case, path, gms, len(names), lms, (len(names) + 1) * lms, info["blocks"], info["lds_bytes"], info["tracks_per_block"], info["blocks_per_cu"],
            abs(v - ref[0]) / abs(ref[0]), err), flush=True)
