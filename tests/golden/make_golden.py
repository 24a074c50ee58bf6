#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the REFERENCE itself.

Run in the build container only (needs /root/reference; see oracle/ref_loader.py for
the import shims).  The produced ``*.npz`` / ``*.json`` files are data: inputs and the
reference's outputs.  Nothing of the reference's source travels.

    python tests/golden/make_golden.py

Reference entry points exercised (relative to /root/reference/):
  extrack/tracking.py:109      P_Cs_inter_bound_stats   (== tracking_0.py:96)
  extrack/tracking_0.py:440    Proba_Cs
  extrack/tracking_0.py:637    cum_Proba_Cs
  extrack/tracking_0.py:463    predict_Bs
  extrack/tracking.py:913      extract_params
  extrack/tracking.py:1214     generate_params
  extrack/tracking.py:1090     get_params
  extrack/simulate_tracks.py:123 sim_FOV  (input generator)
"""
import contextlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_loader as R  # noqa: E402

T = R.load("tracking")
T0 = R.load("tracking_0")
SIM = R.load("simulate_tracks")


def quiet(f, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return f(*a, **k)


def rand_model(rng, S):
    ds = np.sort(rng.uniform(0.004, 0.2, S))
    Fs = rng.dirichlet(np.ones(S) * 2)
    Tm = rng.uniform(0.02, 0.9 / max(S - 1, 1) if S > 3 else 0.3, (S, S))
    Tm[np.arange(S), np.arange(S)] = 0
    Tm[np.arange(S), np.arange(S)] = 1 - Tm.sum(1)
    return ds, Fs, Tm


def kernel_cases():
    """Direct calls of the recursion kernel on tiny batches."""
    rng = np.random.default_rng(20251003)
    out = {}
    meta = []
    cid = 0
    configs = [(2, 1, 2), (2, 1, 4), (2, 1, 6), (3, 1, 3), (3, 1, 4), (4, 1, 3), (4, 1, 5), (5, 1, 3),
               (2, 2, 3), (2, 2, 4), (3, 2, 3), (4, 2, 3), (2, 3, 4), (3, 3, 4), (4, 3, 4)]
    for S, ns, F in configs:
        Ls = sorted(set([2, 3, 5, F, F + 1, F + 2, 12, 30]))
        for L in Ls:
            if L < 2:
                continue
            for D, le in [(2, "scalar"), (2, "dim"), (2, "peak"), (3, "scalar"), (3, "peak"), (1, "scalar")]:
                for isBL in (0, 1):
                    nfinal = S ** (min(L - 1, F) * 1 + ns) if ns == 1 else None
                    big = S ** (F + 2 * ns) > 70000
                    if big and (L not in (2, F + 2, 30) or le == "dim" or D != 2):
                        continue
                    if L == 30 and (D, le) not in ((2, "scalar"), (2, "peak")):
                        continue
                    N = 2 if big else 3
                    min_len = int(rng.choice([2, 3, 5]))
                    ds, Fs, Tm = rand_model(rng, S)
                    pBL = float(rng.uniform(0.02, 0.2))
                    cell = [float(rng.uniform(0.5, 2.0))] if rng.random() < 0.6 else [0.6, 2.5]
                    step = ds[rng.integers(0, S, (N, L, 1))]
                    Cs = np.cumsum(rng.normal(0, 1, (N, L, D)) * step, 1) + rng.normal(0, 0.02, (N, L, D)) + rng.uniform(0, 5, (N, 1, D))
                    if le == "scalar":
                        LE = np.array([[[0.02]]])
                    elif le == "dim":
                        LE = rng.uniform(0.01, 0.04, (1, 1, D))
                    else:
                        LE = rng.uniform(0.01, 0.04, (N, L, D))
                    do_preds = 1 if ns == 1 else 0
                    LP, _, preds = T.P_Cs_inter_bound_stats(Cs, LE, ds, Fs, Tm, pBL, isBL, cell, ns, F, do_preds, min_len)
                    LPC = T0.Proba_Cs(Cs, LE, ds, Fs, Tm, pBL, isBL, cell, ns, F, min_len)
                    pre = "k%04d_" % cid
                    out[pre + "Cs"], out[pre + "LE"] = Cs, LE
                    out[pre + "ds"], out[pre + "Fs"], out[pre + "T"] = ds, Fs, Tm
                    out[pre + "LPC"] = LPC
                    if LP.shape[1] <= 256:
                        out[pre + "LP"] = LP
                    if do_preds:
                        out[pre + "preds"] = np.asarray(preds)
                    meta.append(dict(id=cid, S=S, ns=ns, F=F, L=L, D=D, le=le, isBL=isBL, min_len=min_len, pBL=pBL,
                                     cell_dims=cell, nB=int(LP.shape[1])))
                    cid += 1
    np.savez_compressed(os.path.join(HERE, "kernel_cases.npz"), **out)
    with open(os.path.join(HERE, "kernel_cases.json"), "w") as f:
        json.dump(meta, f, indent=0)
    print("kernel cases:", cid)


def appendix_b():
    """RNG-free known answers."""
    c = np.array([[[0, 0], [0.05, -0.02], [0.07, 0.01], [0.20, 0.15], [0.21, 0.16], [0.18, 0.17]]], float)
    ds, Fs, Tm = np.array([0.01, 0.1]), np.array([0.4, 0.6]), np.array([[0.9, 0.1], [0.2, 0.8]])
    LE = np.array([[[0.02]]])
    rows = []
    for isBL, F in [(1, 2), (1, 4), (0, 2), (0, 4)]:
        LP, _, preds = T.P_Cs_inter_bound_stats(c, LE, ds, Fs, Tm, 0.1, isBL, [1.0], 1, F, 1, 3)
        LPC = T0.Proba_Cs(c, LE, ds, Fs, Tm, 0.1, isBL, [1.0], 1, F, 3)
        rows.append(dict(isBL=isBL, F=F, ns=1, nB=int(LP.shape[1]), LP_C=float(LPC[0]), preds0=[float(x) for x in preds[0, :, 0]]))
    LPC = T0.Proba_Cs(c, LE, ds, Fs, Tm, 0.1, 1, [1.0], 2, 3, 3)
    rows.append(dict(isBL=1, F=3, ns=2, LP_C=float(LPC[0])))
    with open(os.path.join(HERE, "appendix_b.json"), "w") as f:
        json.dump(dict(track=c[0].tolist(), ds=ds.tolist(), Fs=Fs.tolist(), TrMat=Tm.tolist(), LocErr=0.02, pBL=0.1,
                       cell_dims=[1.0], min_len=3, rows=rows,
                       get_all_Bs_3_2=T.get_all_Bs(3, 2).tolist()), f, indent=1)


def params_of(p):
    return {k: dict(value=(None if v.value is None else float(v.value)), vary=bool(v.vary), min=float(v.min), max=float(v.max),
                    expr=v.expr) for k, v in p.items()}


def params_plumbing():
    """extract_params / generate_params / get_params fixtures."""
    res = {"extract": [], "generate": [], "get": []}
    sets = [
        (dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1), 0.02, 1, 1),
        (dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1), 0.02, 2, 1),
        (dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1), 0.02, 3, 0),
        (dict(D0=0, D1=0.04, D2=0.25, LocErr0=0.02, LocErr1=0.025, F0=0.3, F1=0.3, F2=0.4, p01=0.1, p02=0.05, p10=0.07,
              p12=0.2, p20=0.03, p21=0.11, pBL=0.05), 0.06, 1, 1),
        (dict(D0=0, D1=0.02, D2=0.1, D3=0.5, LocErr=0.03, F0=0.1, F1=0.2, F2=0.3, F3=0.4, p01=.05, p02=.05, p03=.05, p10=.05,
              p12=.05, p13=.05, p20=.05, p21=.05, p23=.05, p30=.05, p31=.05, p32=.05, pBL=0.1), 0.02, 3, 1),
        (dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1), 0.02, 1, 2),
        (dict(D0=0, D1=0.04, D2=0.25, LocErr=0.02, F0=0.3, F1=0.3, F2=0.4, p01=0.1, p02=0.05, p10=0.07,
              p12=0.2, p20=0.03, p21=0.11, pBL=0.05), 0.06, 2, 3),
        (dict(D0=0, D1=0.04, D2=0.25, LocErr=0.02, F0=0.3, F1=0.3, F2=0.4, p01=0.1, p02=0.05, p10=0.07,
              p12=0.2, p20=0.03, p21=0.11, pBL=0.05), 0.06, 1, 4),
    ]
    for vals, dt, ns, mt in sets:
        p = R.make_params(**vals)
        LocErr, ds, Fs, Tm, pBL = T.extract_params(p, dt, len(ds_names(vals)), ns, None, mt)
        res["extract"].append(dict(values=vals, dt=dt, nb_substeps=ns, Matrix_type=mt, LocErr=np.asarray(LocErr[0]).tolist(),
                                   ds=ds.tolist(), Fs=Fs.tolist(), TrMat=Tm.tolist(), pBL=pBL))
    gen_calls = [
        dict(nb_states=2), dict(nb_states=3), dict(nb_states=4),
        dict(nb_states=2, LocErr_type=1, LocErr_bounds=[0.005, 0.1], D_max=3, Fractions_bounds=[0.001, 0.99], estimated_transition_rates=0.1),
        dict(nb_states=3, LocErr_type=2, nb_dims=2), dict(nb_states=2, LocErr_type=3, nb_dims=3),
        dict(nb_states=2, LocErr_type=4, slope_offsets_estimates=[1.0, 0.0]),
        dict(nb_states=2, LocErr_type=None),
        dict(nb_states=3, estimated_LocErr=[0.022], estimated_Ds=[0.0001, 0.03, 0.25], estimated_Fs=[0.3, 0.3, 0.4],
             estimated_transition_rates=[0.1, 0.05, 0.03, 0.07, 0.2, 0.2]),
    ]
    for kw in gen_calls:
        res["generate"].append(dict(kwargs=kw, params=params_of(T.generate_params(**kw))))
    get_calls = [
        dict(),
        dict(nb_states=3,
             vary_params={'LocErr': True, 'D0': False, 'D1': True, 'D2': True, 'F0': True, 'F1': True, 'p01': True, 'p02': True,
                          'p10': True, 'p12': True, 'p20': True, 'p21': True, 'pBL': True},
             estimated_vals={'LocErr': 0.023, 'D0': 1e-20, 'D1': 0.02, 'D2': 0.1, 'F0': 0.33, 'F1': 0.33, 'p01': 0.1, 'p02': 0.1,
                             'p10': 0.1, 'p12': 0.1, 'p20': 0.1, 'p21': 0.1, 'pBL': 0.1},
             min_values={'LocErr': 0.007, 'D0': 1e-20, 'D1': 0.0000001, 'D2': 0.000001, 'F0': 0.001, 'F1': 0.001, 'p01': 0.001,
                         'p02': 0.001, 'p10': 0.001, 'p12': 0.001, 'p20': 0.001, 'p21': 0.001, 'pBL': 0.001},
             max_values={'LocErr': 0.6, 'D0': 1e-20, 'D1': 1, 'D2': 10, 'F0': 0.999, 'F1': 0.999, 'p01': 1, 'p02': 1, 'p10': 1,
                         'p12': 1, 'p20': 1, 'p21': 1, 'pBL': 0.99}),
        dict(nb_states=2,
             vary_params={'LocErr': [True, True], 'D0': True, 'D1': True, 'F0': True, 'p01': True, 'p10': True, 'pBL': True},
             estimated_vals={'LocErr': [0.025, 0.03], 'D0': 1e-20, 'D1': 0.05, 'F0': 0.45, 'p01': 0.05, 'p10': 0.05, 'pBL': 0.1},
             min_values={'LocErr': [0.007, 0.007], 'D0': 1e-12, 'D1': 0.00001, 'F0': 0.001, 'p01': 0.001, 'p10': 0.001, 'pBL': 0.001},
             max_values={'LocErr': [0.6, 0.6], 'D0': 1, 'D1': 10, 'F0': 0.999, 'p01': 1., 'p10': 1., 'pBL': 0.99}),
    ]
    for kw in get_calls:
        res["get"].append(dict(kwargs=kw, params=params_of(T.get_params(**kw))))
    # p_stay known values (scipy.stats.norm.cdf path, tracking.py:186-191) via a 1-step call is implicit in the
    # kernel cases; record two explicit values quoted in SURVEY.md section 8c as well.
    with open(os.path.join(HERE, "params_plumbing.json"), "w") as f:
        json.dump(res, f, indent=0)


def ds_names(vals):
    return [k for k in vals if k.startswith("D") and len(k) < 3]


def end_to_end():
    """Whole-dataset objective and posteriors on sim_FOV data (seeded)."""
    out = {}
    info = {}
    # --- E1: the SURVEY/BASELINE anchor (2 states, 500 tracks, len 3..20) ---
    np.random.seed(42)
    tracks, states, sigs = quiet(SIM.sim_FOV, nb_tracks=500, max_track_len=20, min_track_len=3, LocErr=0.02, Ds=np.array([0, 0.25]),
                                 initial_fractions=np.array([0.6, 0.4]), TrMat=np.array([[.9, .1], [.1, .9]]), dt=0.02, pBL=0.1,
                                 cell_dims=[1, None, None])
    vals = dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1)
    p = R.make_params(**vals)
    keys = np.sort(np.array(list(tracks.keys())).astype(int)).astype(str)
    lst = [tracks[k] for k in keys]
    for k in keys:
        out["e1_tr_" + k] = tracks[k]
        out["e1_st_" + k] = states[k]
    e1 = {"values": vals, "dt": 0.02, "cell_dims": [1], "keys": list(keys)}
    e1["cum"] = {}
    for F, ns in [(6, 1), (4, 1), (3, 2)]:
        e1["cum"]["F%d_ns%d" % (F, ns)] = float(quiet(T0.cum_Proba_Cs, p, lst, 0.02, [1], None, 2, ns, F, 0, 1, 1))
    pr = quiet(T0.predict_Bs, tracks, 0.02, p, [1], 2, 6, 1, None)
    for k in keys:
        out["e1_pred_F6_" + k] = pr[k]
    # per-track LL for F=6 (same chunking as cum_Proba_Cs is irrelevant: the kernel is per-track pure)
    LocErr, ds, Fs, Tm, pBL = T0.extract_params(p, 0.02, 2, 1)
    for k in keys:
        isBL = 0 if int(k) == int(keys[-1]) else 1
        out["e1_lpc_F6_" + k] = T0.Proba_Cs(tracks[k], LocErr[0], ds, Fs, Tm, pBL, isBL, [1], 1, 6, int(keys[0]))
    info["e1"] = e1

    # --- E2: 3 states, per-peak localisation error input, 2 cell dims ---
    np.random.seed(7)
    Tm3 = np.array([[0.85, 0.1, 0.05], [0.08, 0.82, 0.1], [0.05, 0.1, 0.85]])
    tracks, states, sigs = quiet(SIM.sim_FOV, nb_tracks=250, max_track_len=12, min_track_len=2, LocErr=0.025, Ds=np.array([0, 0.04, 0.3]),
                                 initial_fractions=np.array([0.3, 0.3, 0.4]), TrMat=Tm3, LocErr_std=0.3, dt=0.03, pBL=0.07,
                                 cell_dims=[0.8, 2.0, None])
    vals = dict(D0=1e-4, D1=0.04, D2=0.3, F0=0.3, F1=0.3, F2=0.4, p01=0.1, p02=0.05, p10=0.08, p12=0.1, p20=0.05, p21=0.1, pBL=0.07,
                slope_LocErr=1.1, offset_LocErr=0.002)
    p = R.make_params(**vals)
    keys = np.sort(np.array(list(tracks.keys())).astype(int)).astype(str)
    lst = [tracks[k] for k in keys]
    lsig = [sigs[k] for k in keys]
    for k in keys:
        out["e2_tr_" + k] = tracks[k]
        out["e2_sig_" + k] = sigs[k]
    e2 = {"values": vals, "dt": 0.03, "cell_dims": [0.8, 2.0], "keys": list(keys)}
    e2["cum_F4_ns1_affine"] = float(quiet(T0.cum_Proba_Cs, p, lst, 0.03, [0.8, 2.0], lsig, 3, 1, 4, 0, 1, 1))
    vals_raw = {k: v for k, v in vals.items() if "LocErr" not in k}
    p_raw = R.make_params(**vals_raw)
    e2["values_raw"] = vals_raw
    e2["cum_F4_ns1_raw"] = float(quiet(T0.cum_Proba_Cs, p_raw, lst, 0.03, [0.8, 2.0], lsig, 3, 1, 4, 0, 1, 1))
    e2["cum_F3_ns2_raw"] = float(quiet(T0.cum_Proba_Cs, p_raw, lst, 0.03, [0.8, 2.0], lsig, 3, 2, 3, 0, 1, 1))
    pr = quiet(T0.predict_Bs, tracks, 0.03, p_raw, [0.8, 2.0], 3, 4, 1, sigs)
    for k in keys:
        out["e2_pred_F4_raw_" + k] = pr[k]
    info["e2"] = e2

    # --- E3: invalid parameters -> +inf (tracking_0.py:657,708-710) ---
    bad = R.make_params(D0=0.25, D1=1e-3, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1)  # ds decreasing
    info["e3_inf"] = float(quiet(T0.cum_Proba_Cs, bad, [out["e1_tr_3"]], 0.02, [1], None, 2, 1, 4, 0, 1, 1))
    np.savez_compressed(os.path.join(HERE, "end_to_end.npz"), **out)
    with open(os.path.join(HERE, "end_to_end.json"), "w") as f:
        json.dump(info, f, indent=1)
    print("e1 cum", e1["cum"], "e2", e2["cum_F4_ns1_affine"], e2["cum_F4_ns1_raw"], e2["cum_F3_ns2_raw"])


def c1_config():
    """BASELINE.json configs[0]: sim_FOV 10k tracks, 2 states, len <= 20 (CPU plumbing case).
    Stores the tracks (float32-exact rounding is NOT applied: full float64) and the reference objective."""
    np.random.seed(1)
    tracks, states, sigs = quiet(SIM.sim_FOV, nb_tracks=10000, max_track_len=20, min_track_len=5, LocErr=0.02, Ds=np.array([0, 0.25]),
                                 initial_fractions=np.array([0.6, 0.4]), TrMat=np.array([[.9, .1], [.1, .9]]), dt=0.02, pBL=0.1,
                                 cell_dims=[1, None, None])
    vals = dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1)
    p = R.make_params(**vals)
    keys = np.sort(np.array(list(tracks.keys())).astype(int)).astype(str)
    lst = [tracks[k] for k in keys]
    val = float(quiet(T0.cum_Proba_Cs, p, lst, 0.02, [1], None, 2, 1, 6, 0, 1, 1))
    out = {"tr_" + k: tracks[k] for k in keys}
    np.savez_compressed(os.path.join(HERE, "c1_simfov_10k.npz"), **out)
    with open(os.path.join(HERE, "c1_simfov_10k.json"), "w") as f:
        json.dump(dict(values=vals, dt=0.02, cell_dims=[1], frame_len=6, nb_substeps=1, keys=list(keys), n_tracks=int(sum(len(x) for x in lst)),
                       cum_Proba_Cs=val), f, indent=1)
    print("c1:", sum(len(x) for x in lst), "tracks, -LL", val)


if __name__ == "__main__":
    assert R.available(), "reference not mounted"
    appendix_b()
    params_plumbing()
    kernel_cases()
    end_to_end()
    c1_config()
