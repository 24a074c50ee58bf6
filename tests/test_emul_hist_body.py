"""The state-duration histogram kernel body (extrack_amd/csrc/xt_hist.h) run on CPU threads (tests/emul) against the golden vectors the
reference produced (extrack/histograms.py P_segment_len): sort-based top-K pruning, the LL re-ordering quirk, the streamed final step,
run-length decoding of the bit-packed histories, LDS and global-workspace parent buffers."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul"))


def test_emulated_histogram_body_vs_reference_fixtures():
    import run_emul as E
    from oracle import oracle_np as O
    info = json.load(open(os.path.join(GOLDEN, "hist_cases.json")))
    data = np.load(os.path.join(GOLDEN, "hist_cases.npz"))
    worst, n = 0.0, 0
    for row in info["cases"]:
        if (row["id"] % 9 != 0 and not (row["L"] == 2 and row["id"] % 4 == 1)) or row["K"] > 50:
            continue  # a subset (incl. minimal-length cases; the 200-sequence sorts are left to the GPU test) keeps the CPU suite short
        pre = "h%04d_" % row["id"]
        g = lambda k: data[pre + k]
        ps = O.p_stay_table(g("ds"), row["S"], 1, row["cell_dims"])
        h = E.run_hist(g("Cs"), g("LE"), g("ds"), g("Fs"), g("T"), row["pBL"], row["isBL"], ps, row["min_l"], row["K"], nblocks=1 + row["id"] % 2,
                       threads=64, par_lds=row["id"] % 4 != 3)
        d = np.abs(h - g("hist")).max()
        assert d < 1e-9 * max(1.0, row["N"]), (row, d)
        worst = max(worst, d)
        n += 1
    assert n > 15
    print("cases", n, "worst |d hist|", worst)


def test_emulated_histogram_long_tracks_many_history_words():
    """Histories of more than 256 bits (3 states x 150 positions = 5 words of 64 bits; 2 states x 300 = 5 words) against the numpy oracle:
    the shift with carry across words and the run-length decoding across word boundaries."""
    import run_emul as E
    from oracle import oracle_hist as OH, oracle_np as O
    rng = np.random.default_rng(9)
    for S, L, N, K, par_lds in ((3, 140, 1, 12, 1), (2, 270, 1, 10, 0)):
        Tm = rng.uniform(0.03, 0.12, (S, S))
        Tm[np.arange(S), np.arange(S)] = 0
        Tm[np.arange(S), np.arange(S)] = 1 - Tm.sum(1)
        ds = np.sort(rng.uniform(0.01, 0.15, S))
        Fs = rng.dirichlet(np.ones(S) * 3)
        st = rng.integers(0, S, (N, L, 1))
        Cs = np.cumsum(rng.normal(0, 1, (N, L, 2)) * ds[st], 1) + rng.normal(0, 0.02, (N, L, 2))
        LE = np.array([[[0.02]]])
        ps = O.p_stay_table(ds, S, 1, [1.0])
        h = E.run_hist(Cs, LE, ds, Fs, Tm, 0.05, 1, ps, 3, K, nblocks=2, threads=64, par_lds=par_lds)
        ref = OH.p_segment_len(Cs, LE, ds, Fs, Tm, min_l=3, pBL=0.05, isBL=1, cell_dims=[1.0], max_nb_states=K)
        assert np.abs(h - ref).max() < 1e-9 * N, (S, L, np.abs(h - ref).max())
