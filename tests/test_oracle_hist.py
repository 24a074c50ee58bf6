"""The state-duration histogram oracle (oracle/oracle_hist.py) against golden vectors produced by the reference itself
(tests/golden/make_golden_hist.py: extrack/histograms.py P_segment_len on 216 chunks, len_hist end to end)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def hist_cases():
    info = json.load(open(os.path.join(GOLDEN, "hist_cases.json")))
    data = np.load(os.path.join(GOLDEN, "hist_cases.npz"))
    return info, data


def test_p_segment_len_matches_reference(hist_cases):
    from oracle import oracle_hist as OH
    info, data = hist_cases
    worst = 0.0
    pruned = 0
    for row in info["cases"]:
        pre = "h%04d_" % row["id"]
        g = lambda k: data[pre + k]
        h = OH.p_segment_len(g("Cs"), g("LE"), g("ds"), g("Fs"), g("T"), row["min_l"], row["pBL"], row["isBL"], row["cell_dims"], 1, row["K"])
        ref = g("hist")
        assert h.shape == ref.shape == (row["L"] - 1, row["S"])
        d = np.abs(h - ref).max()
        assert d < 1e-9 * max(1.0, row["N"]), (row, d)
        worst = max(worst, d)
        pruned += row["S"] ** row["L"] > row["K"] * row["S"]
    assert pruned > 100  # most cases exercise the top-K pruning (and with it the reference's LL re-ordering quirk)
    print("worst |d hist|", worst)


def test_len_hist_end_to_end_matches_reference(hist_cases):
    from oracle import oracle_hist as OH
    info, data = hist_cases
    e = info["e2e"]
    tracks = {k: data["e_tr_" + k] for k in e["keys"]}
    for name, ref in e["len_hist"].items():
        h = OH.len_hist(e["values"], tracks, e["dt"], e["cell_dims"], max_nb_states=int(name[1:]))
        ref = np.array(ref)
        assert h.shape == ref.shape
        assert np.abs(h - ref).max() < 1e-8, (name, np.abs(h - ref).max())
        assert abs(h.sum() - ref.sum()) < 1e-8
