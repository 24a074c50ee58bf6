"""GPU tests of the state-duration histograms (extrack_segment_len_hist) through the C ABI: every
reference-generated P_segment_len fixture (216 chunks: 2-4 states, lengths 2-20, pruning from 4 to 200 sequences, per-peak and
per-dimension localisation errors, 1-3 dims) and the end-to-end len_hist fixtures, to 1e-9; larger datasets against the oracle
and through size-independent properties."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _params(vals):
    from extrack_amd.lmfit_compat import Parameters
    p = Parameters()
    for k, v in vals.items():
        p.add(k, value=v)
    return p


@pytest.fixture(scope="module")
def hist_cases():
    return json.load(open(os.path.join(GOLDEN, "hist_cases.json"))), np.load(os.path.join(GOLDEN, "hist_cases.npz"))


def test_p_segment_len_golden(hist_cases):
    from extrack_amd import histograms as H
    info, data = hist_cases
    worst = 0.0
    for row in info["cases"]:
        pre = "h%04d_" % row["id"]
        g = lambda k: data[pre + k]
        _, _, h = H.P_segment_len(g("Cs"), g("LE"), g("ds"), g("Fs"), g("T"), row["min_l"], row["pBL"], row["isBL"], row["cell_dims"], 1, row["K"])
        ref = g("hist")
        assert h.shape == ref.shape
        d = np.abs(h - ref).max()
        assert d < 1e-9 * max(1.0, row["N"]), (row, d)
        worst = max(worst, d)
    print("cases", len(info["cases"]), "worst |d hist|", worst)


def test_len_hist_end_to_end_golden(hist_cases, capsys):
    from extrack_amd import histograms as H
    info, data = hist_cases
    e = info["e2e"]
    tracks = {k: data["e_tr_" + k] for k in e["keys"]}
    p = _params(e["values"])
    for name, ref in e["len_hist"].items():
        h = H.len_hist(tracks, p, e["dt"], cell_dims=e["cell_dims"], nb_states=2, max_nb_states=int(name[1:]))
        capsys.readouterr()
        ref = np.array(ref)
        assert h.shape == ref.shape and np.abs(h - ref).max() < 1e-8, (name, np.abs(h - ref).max())
    # every track contributes exactly its number of runs that are shorter than the track: totals are bounded by the positions
    n_tracks = sum(len(v) for v in tracks.values())
    assert 0 < h.sum() < sum(len(v) * int(k) for k, v in tracks.items()) and h.sum() > 0.5 * n_tracks


def test_hist_larger_dataset_vs_oracle_and_shard_additivity(capsys):
    """3 states, lengths 4-40 (histories of up to 80 bits: two words), 2000 tracks, default max_nb_states 500 (the sort works on 2048
    candidates): against the numpy oracle, and additive over two row shards (what the multi-GPU path relies on)."""
    from extrack_amd import histograms as H, synth
    from oracle import oracle_hist as OH
    Tm = np.array([[0.9, 0.07, 0.03], [0.05, 0.9, 0.05], [0.03, 0.07, 0.9]])
    lens = {4: 300, 9: 500, 17: 600, 28: 400, 40: 200}
    # asymmetric rates: a model with symmetric rates (p01 == p21, p10 == p12, ...) produces EXACT ties in the ranking, whose order is
    # implementation-defined in the reference (unstable argsort) and, through its LL quirk, visible in the result
    tracks = {str(L): synth.brownian_tracks(n, L, [0.0, 0.04, 0.25], Tm, [0.3, 0.3, 0.4], seed=40 + L) for L, n in lens.items()}
    vals = dict(D0=1e-4, D1=0.04, D2=0.25, LocErr=0.02, F0=0.3, F1=0.33, F2=0.37, p01=0.071, p02=0.033, p10=0.052, p12=0.047, p20=0.029, p21=0.068,
                pBL=0.1)
    p = _params(vals)
    h = H.len_hist(tracks, p, 0.02, cell_dims=[1.0, None], nb_states=3, max_nb_states=500)
    capsys.readouterr()
    sub = {k: v[:40] for k, v in tracks.items()}
    hs = H.len_hist(sub, p, 0.02, cell_dims=[1.0, None], nb_states=3, max_nb_states=500)
    capsys.readouterr()
    ref = OH.len_hist(vals, sub, 0.02, [1.0, None], max_nb_states=500)
    assert np.abs(hs - ref).max() < 1e-8, np.abs(hs - ref).max()
    parts = np.zeros_like(h)
    for half in (0, 1):
        sh = {k: (v[:len(v) // 2] if half == 0 else v[len(v) // 2:]) for k, v in tracks.items()}
        parts += H.len_hist(sh, p, 0.02, cell_dims=[1.0, None], nb_states=3, max_nb_states=500)
        capsys.readouterr()
    assert np.abs(parts - h).max() < 1e-9 * h.max()
    assert h.shape == (40, 3) and np.all(h >= 0) and h[-1].sum() == 0.0


def test_hist_long_tracks_many_history_words():
    """Histories longer than the 256 bits of rounds 2 - 3 (round 4: a run-time number of 64-bit words): 3 states x 200 positions (7 words),
    2 states x 400 (7 words), 5 states x 90 (3 bits per state, 5 words) against the numpy oracle."""
    from extrack_amd import histograms as H, synth
    from oracle import oracle_hist as OH
    rng = np.random.default_rng(5)
    for S, L, N, K in ((3, 200, 3, 40), (2, 400, 3, 30), (5, 90, 2, 60)):
        Tm = rng.uniform(0.02, 0.12, (S, S))
        Tm[np.arange(S), np.arange(S)] = 0
        Tm[np.arange(S), np.arange(S)] = 1 - Tm.sum(1)
        ds = np.sort(rng.uniform(0.005, 0.2, S))
        Fs = rng.dirichlet(np.ones(S) * 3)
        Cs = synth.brownian_tracks(N, L, list(ds ** 2 / (2 * 0.02)), Tm, list(Fs), seed=100 + S)
        LE = np.array([[[0.02]]])
        _, _, h = H.P_segment_len(Cs, LE, ds, Fs, Tm, min_l=3, pBL=0.05, isBL=1, cell_dims=[1.0], max_nb_states=K)
        ref = OH.p_segment_len(Cs, LE, ds, Fs, Tm, min_l=3, pBL=0.05, isBL=1, cell_dims=[1.0], max_nb_states=K)
        assert h.shape == ref.shape and np.abs(h - ref).max() < 1e-9 * N, (S, L, np.abs(h - ref).max())


def test_hist_argument_errors_and_nan():
    from extrack_amd import _lib, histograms as H, synth
    Cs = synth.brownian_tracks(8, 6, [0.0, 0.25], [[.9, .1], [.1, .9]], [.6, .4], seed=1)
    LE = np.array([[[0.02]]])
    with pytest.raises(NotImplementedError):
        H.P_segment_len(Cs, LE, [0.01, 0.1], [.5, .5], np.array([[.9, .1], [.1, .9]]), nb_substeps=2)
    with pytest.raises(ValueError):
        H.P_segment_len(Cs[:, :1], LE, [0.01, 0.1], [.5, .5], np.array([[.9, .1], [.1, .9]]))
    with pytest.raises(_lib.ExtrackError):  # 2100 positions x 2 bits do not fit the 4096-bit histories
        H.P_segment_len(synth.brownian_tracks(2, 2100, [0.0, 0.1, 0.2], np.full((3, 3), 1 / 3), [.3, .3, .4], seed=2), LE, [0.01, 0.05, 0.1],
                        [.3, .3, .4], np.full((3, 3), 1 / 3), max_nb_states=20)
    Cs[3, 2, 0] = np.nan
    _, _, h = H.P_segment_len(Cs, LE, [0.01, 0.1], [.5, .5], np.array([[.9, .1], [.1, .9]]), min_l=3)
    assert np.all(np.isnan(h))
