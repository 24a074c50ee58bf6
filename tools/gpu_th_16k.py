#!/usr/bin/env python3
"""Threshold fusion with more than 8192 expanded sequences per step (4 states x 3 substeps: 16 384 at the second position) on the GPU:
against the oracle on a few tracks, then timing at C5 size.   usage: gpu_th_16k.py [n_tracks_big]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from extrack_amd import synth
from extrack_amd.engine import TrackSet
from oracle import oracle_th as OT

S, ns, F = 4, 3, 4
ds = np.array([0.003, 0.02, 0.06, 0.15])
Fs = np.array([0.25, 0.3, 0.2, 0.25])
T = np.full((S, S), 0.02) + np.diag([0.01, 0.0, 0.015, 0.005])
T[np.arange(S), np.arange(S)] = 0
T[np.arange(S), np.arange(S)] = 1 - T.sum(1)
rng = np.random.default_rng(12)
for N, L, thr in ((3, 4, 0.3), (40, 7, 0.2)):
    Cs = np.cumsum(rng.normal(0, 1, (N, L, 2)) * ds[rng.integers(0, S, (N, L, 1))], 1) + rng.normal(0, 0.02, (N, L, 2))
    LE = np.full((1, 1, 1), 0.02)
    t0 = time.time()
    ref = OT.proba_cs_th(Cs, LE, ds, Fs, T, 0.1, 1, [1.0], ns, F, 3, thr, 120)
    t1 = time.time()
    ts = TrackSet([Cs], min_len=3, max_len=L + 1)
    model = ts.make_model(LE, ds, Fs, T, 0.1, [1.0], ns, F)
    tot, ll = ts.loglik_th(model, thr, 120, N, per_track=True)
    print("N %d L %d: oracle %.1f s, max |dLL| %.2e, total %.10f vs %.10f, launch %s" % (N, L, t1 - t0, np.abs(ll - ref).max(), tot, ref.sum(), ts.ctx.last_launch_info()), flush=True)
    ts.close()
Nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
Ds = [0.0, 0.02, 0.1, 0.5]
Tm = np.array([[.85, .05, .05, .05], [.05, .85, .05, .05], [.05, .05, .85, .05], [.05, .05, .05, .85]])
Tsub = 1 - np.exp(-(Tm - np.diag(np.diag(Tm))) / ns)
Tsub[np.arange(4), np.arange(4)] = 0
Tsub[np.arange(4), np.arange(4)] = 1 - Tsub.sum(1)
dsb = np.sqrt(2 * np.maximum(np.array(Ds), 1e-3 * 0.25) * 0.02)
X = synth.brownian_tracks(Nb, 60, Ds, Tm, [.25] * 4, seed=2)
ts = TrackSet([X])
model = ts.make_model(np.array([[[0.02]]]), dsb, np.array([.25] * 4), Tsub, 0.1, (1.0,), ns, 4)
for rep in range(3):
    t0 = time.perf_counter()
    tot = ts.loglik_th(model, 0.2, 120, 2000)
    print("C5 model, %d x 60, threshold fusion: total %.4f wall %.1f ms kernels %.1f ms launch %s" % (Nb, tot, (time.perf_counter() - t0) * 1e3, ts.ctx.last_kernel_ms(), ts.ctx.last_launch_info()), flush=True)
ts.th_freeze_plan(True)
t0 = time.perf_counter()
tot = ts.loglik_th(model, 0.2, 120, 2000)
print("  frozen plan: total %.4f wall %.1f ms" % (tot, (time.perf_counter() - t0) * 1e3), flush=True)
ts.close()
