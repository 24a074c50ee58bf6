"""CPU-side checks: the C-ABI library loads and exports every symbol include/extrack_hip.h declares (no compute calls
without a GPU), host logic mirrors the reference's parameter plumbing, the product fails loudly without a device."""
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "extrack_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(extrack_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from extrack_amd import _lib
    lib = _lib.load()
    declared = _declared_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_lib.EXPORTS) == declared
    assert lib.extrack_abi_version() == 6


def test_no_silent_cpu_fallback():
    """Without a gfx950 device every compute entry point must raise - there is no CPU path in the product."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from extrack_amd import _lib, tracking
    with pytest.raises(_lib.ExtrackError):
        _lib.Context(0)
    with pytest.raises(_lib.ExtrackError):
        tracking.Proba_Cs(np.zeros((2, 5, 2)), np.array([[[0.02]]]), [0.01, 0.1], [0.5, 0.5], [[.9, .1], [.1, .9]], 0.1, 1, [1.0], 1, 4, 3)
    import extrack_amd
    src = "".join(open(os.path.join(ROOT, "extrack_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "extrack_amd")) if f.endswith(".py"))
    assert "oracle" not in src.replace("the oracle", ""), "the product package must not reference the oracle"


def test_p_stay_host_helper_matches_scipy_quadrature():
    from extrack_amd import _lib, engine
    ds = np.array([0.0063, 0.05, 0.1])
    for ns, cells in [(1, [1.0]), (2, [0.8, 2.0]), (3, [0.5])]:
        a = _lib.p_stay_table_c(ds, ns, cells)
        b = engine.p_stay_table(ds, 3, ns, cells)
        np.testing.assert_allclose(a, b, rtol=1e-13, atol=0)
    # values quoted in SURVEY.md section 8c (scipy.stats.norm.cdf path of the reference)
    v = engine.p_stay_table(np.array([0.01, 0.1]), 2, 1, [1.0])
    np.testing.assert_allclose(v, [0.992024479881394, 0.9202118763725834], rtol=1e-13)


def _mk(vals):
    from extrack_amd.lmfit_compat import Parameters
    p = Parameters()
    for k, v in vals.items():
        p.add(k, value=v)
    return p


def test_extract_params_matches_reference(params_plumbing):
    from extrack_amd import tracking as T
    for row in params_plumbing["extract"]:
        LocErr, ds, Fs, TrMat, pBL = T.extract_params(_mk(row["values"]), row["dt"], len(row["ds"]), row["nb_substeps"], None,
                                                      row["Matrix_type"])
        np.testing.assert_allclose(LocErr[0], row["LocErr"], rtol=0, atol=0)
        np.testing.assert_allclose(ds, row["ds"], rtol=1e-15)
        np.testing.assert_allclose(Fs, row["Fs"], rtol=0, atol=0)
        np.testing.assert_allclose(TrMat, row["TrMat"], rtol=1e-14, atol=1e-17)
        assert pBL == row["pBL"]


def _check_params(p, ref):
    assert list(p.keys()) == list(ref.keys())
    for k, r in ref.items():
        q = p[k]
        assert q.expr == r["expr"], k
        if r["expr"] is None:
            assert q.vary == r["vary"], k
            assert q.min == r["min"] and q.max == r["max"], k
        if r["value"] is not None:
            # lmfit clips an initial value into [min, max] (the fixture came from a non-clipping stub)
            want = float(np.clip(r["value"], r["min"], r["max"])) if r["expr"] is None else r["value"]
            tol = 1e-15 * max(1.0, abs(want)) if r["expr"] is None else 1e-11  # exprs see the clipped inputs
            assert abs(q.value - want) <= tol, k


def test_generate_and_get_params_match_reference(params_plumbing):
    from extrack_amd import tracking as T
    for row in params_plumbing["generate"]:
        _check_params(T.generate_params(**row["kwargs"]), row["params"])
    for row in params_plumbing["get"]:
        _check_params(T.get_params(**row["kwargs"]), row["params"])


def test_lmfit_compat_bounds_and_expr():
    from extrack_amd import lmfit_compat as L
    if L.HAVE_LMFIT:
        pytest.skip("real lmfit present")
    p = L.Parameters()
    p.add("a", value=0.3, min=0.0, max=1.0)
    p.add("b", value=2.0, min=1.0)
    p.add("c", value=-1.0, max=0.0)
    p.add("d", value=5.0)
    p.add("e", expr="1 - a")
    p.add("f", expr="a + b_minus" if False else "a + b")
    assert abs(p["e"].value - 0.7) < 1e-15 and abs(p["f"].value - 2.3) < 1e-15
    for k in "abcd":
        x = L._to_internal(p[k])
        assert abs(L._from_internal(p[k], x) - p[k].value) < 1e-12
    with pytest.raises(ValueError):
        p.add("bad", expr="__import__('os').system('true')")

    # scalar objective, bounded minimum at the boundary-respecting optimum
    def obj(pp, shift):
        return (pp["a"].value - 0.9) ** 2 + (pp["b"].value - shift) ** 2 + (pp["e"].value - 0.1) ** 2
    for method in ("bfgs", "powell", "nelder"):
        r = L.minimize(obj, p, args=(0.5,), method=method)
        assert abs(r.params["a"].value - 0.9) < 1e-3, method
        assert abs(r.params["b"].value - 1.0) < 1e-3, method  # lower bound active
        assert abs(r.params["e"].value - (1 - r.params["a"].value)) < 1e-12
        assert r.residual.shape == (1,) and r.nfev > 3
    assert p["a"].value == 0.3  # the caller's Parameters are not modified


def test_predict_bs_type_error_and_sorting():
    from extrack_amd import engine, tracking as T
    tr = {"10": np.zeros((2, 10, 2)), "3": np.zeros((0, 3, 2)), "7": np.zeros((1, 7, 2))}
    keys, lst, sig = engine.sort_buckets(tr)
    assert keys == ["3", "7", "10"] and [a.shape[1] for a in lst] == [7, 10] and sig is None
    with pytest.raises(TypeError):
        T.predict_Bs(tr, 0.02, {"D0": 0.1})
    with pytest.raises(ValueError):
        T.param_fitting({"5": np.zeros((0, 5, 2))}, 0.02)
    # per-track time steps: dt as a list of arrays [n, len] (bucket order) gives one ds array [n, len, S] per bucket (tracking.py:979-982)
    dts = [np.full((2, 3), 0.02), np.array([[0.01, 0.02, 0.03, 0.04]])]
    _, ds, _, _, _ = T.extract_params(_mk(dict(D0=0, D1=1, F0=.5, F1=.5, p01=.1, p10=.1, pBL=.1, LocErr=.02)), dts, 2, 1)
    assert [d.shape for d in ds] == [(2, 3, 2), (1, 4, 2)] and np.allclose(ds[1][0, :, 1], np.sqrt(2 * np.array([0.01, 0.02, 0.03, 0.04])))
    with pytest.raises(TypeError):
        T.extract_params(_mk(dict(D0=0, D1=1, F0=.5, F1=.5, p01=.1, p10=.1, pBL=.1, LocErr=.02)), {"3": dts[0]}, 2, 1)


def test_invalid_parameters_short_circuit_to_inf(capsys):
    """The validity guard (tracking.py:1017,1078-1080) is host logic: no device work is attempted."""
    from extrack_amd import tracking as T

    class FakeTS:
        has_sigma = False
        has_dt = False
    bad = _mk(dict(D0=0.25, D1=1e-3, LocErr=0.02, F0=0.6, F1=0.4, p01=0.1, p10=0.1, pBL=0.1))
    assert T._objective_model(bad, FakeTS(), 0.02, [1], None, 2, 1, 6, 1) is None
    neg = _mk(dict(D0=1e-3, D1=0.25, LocErr=0.02, F0=1.2, F1=-0.2, p01=0.1, p10=0.1, pBL=0.1))
    assert T._objective_model(neg, FakeTS(), 0.02, [1], None, 2, 1, 6, 1) is None
