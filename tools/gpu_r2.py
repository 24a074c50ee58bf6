#!/usr/bin/env python3
"""Register-resident 2-state kernels (xt_reg2.h) on the GPU: parity against the LDS-resident kernels and the oracle, and timings of
both on the C2 dataset (1e6 x 30).  usage: gpu_r2.py [scale]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import gradient, synth, tracking as T
from oracle import oracle_np as O
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0


def ctx_env(**kw):
    for k in ("EXTRACK_LL_PATH", "EXTRACK_GRAD_PATH", "EXTRACK_R2_EXP"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in kw.items()})


# ---- parity on small data, all frame lengths / dims
worst = 0.0
for F in (4, 5, 6, 7):
    for D in (1, 2, 3):
        Ds = [0.001, 0.25]
        for L in (2, 3, 7, 40):
            Cs = synth.brownian_tracks(301, L, Ds, [[.9, .1], [.2, .8]], [.6, .4], seed=L + F, dims=D)
            p = T.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=Ds, estimated_LocErr=[0.02], estimated_Fs=[0.6], estimated_transition_rates=0.1)
            names = gradient.free_names(p)
            out = {}
            for path in ("lds", "reg2"):
                ctx_env(EXTRACK_LL_PATH=path, EXTRACK_GRAD_PATH=path)
                ts = T.TrackSet([Cs])
                model = T._objective_model(p, ts, 0.02, [1], None, 2, 1, F, 1)
                ll, per = ts.loglik(model, per_track=True)
                v, g = gradient.objective_and_gradient(p, ts, 0.02, [1], 2, 1, F, names=names)
                out[path] = (ll, per, v, g)
                ts.close()
            e1 = np.abs(out["lds"][1] - out["reg2"][1]).max()
            e2 = abs(out["lds"][2] - out["reg2"][2]) / abs(out["lds"][2])
            e3 = np.abs(out["lds"][3] - out["reg2"][3]).max() / np.abs(out["lds"][3]).max()
            e4 = abs(out["reg2"][0] + out["reg2"][2]) / abs(out["reg2"][0])
            worst = max(worst, e1, e2, e3, e4)
            assert e1 < 1e-10 and e2 < 1e-12 and e3 < 1e-9 and e4 < 1e-12, (F, D, L, e1, e2, e3, e4)
print("parity reg2 vs lds kernels (LL per track, objective, gradient): worst %.3e" % worst)

# ---- timings on C2
Cs = synth.brownian_tracks(int(1e6 * scale), 30, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=0)
p = T.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[0.001, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.6], estimated_transition_rates=0.1)
names = gradient.free_names(p)
res = {}
for rnd in range(2):
    for label, env in (("lds", dict(EXTRACK_LL_PATH="lds", EXTRACK_GRAD_PATH="lds")), ("reg2", dict(EXTRACK_LL_PATH="reg2", EXTRACK_GRAD_PATH="reg2")),
                       ("reg2-exp1", dict(EXTRACK_LL_PATH="reg2", EXTRACK_GRAD_PATH="reg2", EXTRACK_R2_EXP=1)),
                       ("reg2-exp2", dict(EXTRACK_LL_PATH="reg2", EXTRACK_GRAD_PATH="reg2", EXTRACK_R2_EXP=2)),
                       ("reg2-exp3", dict(EXTRACK_LL_PATH="reg2", EXTRACK_GRAD_PATH="reg2", EXTRACK_R2_EXP=3))):
        ctx_env(**env)
        ts = T.TrackSet([Cs])
        for F in (6,):
            model = T._objective_model(p, ts, 0.02, [1], None, 2, 1, F, 1)
            for _ in range(12):
                ll = ts.loglik(model)
            kms = []
            for _ in range(10):
                ll = ts.loglik(model)
                kms.append(ts.ctx.last_kernel_ms())
            for nd in (len(names), 4):
                for _ in range(3):
                    v, g = gradient.objective_and_gradient(p, ts, 0.02, [1], 2, 1, F, names=names[:nd])
                gms = []
                for _ in range(4):
                    v, g = gradient.objective_and_gradient(p, ts, 0.02, [1], 2, 1, F, names=names[:nd])
                    gms.append(ts.ctx.last_grad_ms())
                print("round %d %-9s F=%d: LL kernel %.3f ms (min %.3f)   grad %d dirs %.2f ms (min %.2f)  launch %s  LL %.6f  g0 %.6f" % (
                    rnd, label, F, np.mean(kms), np.min(kms), nd, np.mean(gms), np.min(gms), ts.ctx.last_launch_info(), ll, g[0]), flush=True)
        ts.close()
