"""CPU checks of the gradient path: the kernel body of extrack_amd/csrc/xt_grad.h run on CPU threads
(tests/emul) against Richardson-extrapolated central differences of the pinned numpy oracle along every model direction; the host
chain rule (expr constraints -> extract_params -> p_stay) against finite differences; the optimiser's use of an analytic gradient."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul"))


def _model(S, seed):
    rng = np.random.default_rng(seed)
    Ds = np.sort(rng.uniform(0.01, 0.3, S))
    Ds[0] = 0.001
    T = np.full((S, S), 0.05) + rng.uniform(0, 0.03, (S, S))
    T[np.arange(S), np.arange(S)] = 0
    T[np.arange(S), np.arange(S)] = 1 - T.sum(1)
    Fs = rng.uniform(0.2, 1, S)
    return Ds, T, Fs / Fs.sum()


def _richardson(f, h):
    d1 = (f(h) - f(-h)) / (2 * h)
    d2 = (f(h / 2) - f(-h / 2)) / h
    return (4 * d2 - d1) / 3


def model_directions(S, K, ns, ds2, T, le, cell):
    """Unit directions of every model field + the induced p_stay tangent: [(name, tangent dict, perturb(x) -> kwargs of total())]."""
    from oracle import oracle_np as O
    ps = lambda d2: O.p_stay_table(np.sqrt(d2), S, ns, cell)
    out = []
    for s in range(S):
        e = np.zeros(S)
        e[s] = 1
        dps = _richardson(lambda x: ps(ds2 + e * x), 1e-4 * ds2[s])
        out.append(("ds2_%d" % s, dict(ds2=e, p_stay=dps), dict(ds2=e), 1e-3 * ds2[s]))
    for s in range(S):
        e = np.zeros(S)
        e[s] = 1
        out.append(("F%d" % s, dict(Fs=e), dict(Fs=e), 1e-3))
    for i in range(S):
        for j in range(S):
            e = np.zeros((S, S))
            e[i, j] = 1
            out.append(("T%d%d" % (i, j), dict(TrMat=e), dict(T=e), 1e-3 * T[i, j]))
    for k in range(K):
        e = np.zeros(K)
        e[k] = 1
        out.append(("le%d" % k, dict(locerr=e), dict(le=e), 1e-5))
    out.append(("pBL", dict(pBL=1.0), dict(pBL=1.0), 1e-4))
    return out


def oracle_fd_gradient(Cs, le, ds2, Fs, T, pBL, isBL, cell, ns, F, min_len, dirs):
    from oracle import oracle_np as O

    def total(x, d):
        return O.proba_cs(Cs, (le + x * d.get("le", 0.0))[None, None], np.sqrt(ds2 + x * d.get("ds2", 0.0)), Fs + x * d.get("Fs", 0.0),
                          T + x * d.get("T", 0.0), pBL + x * d.get("pBL", 0.0), isBL, cell, ns, F, min_len).sum()

    return np.array([_richardson(lambda x: total(x, d), h) for _, _, d, h in dirs])


@pytest.mark.parametrize("S,ns,F,L,N,D,K,isBL,kw", [(2, 1, 4, 9, 10, 2, 1, 1, {}), (2, 1, 6, 12, 8, 2, 1, 0, dict(PJ=4)), (3, 1, 3, 8, 6, 2, 2, 1, dict(PJ=8)),
                                                     (2, 2, 3, 7, 6, 1, 1, 1, dict(PJ=2)), (3, 1, 4, 9, 4, 3, 3, 1, dict(tan_lds=0, PJ=4)),
                                                     (2, 1, 4, 9, 10, 2, 1, 1, dict(generic_g=1, PJ=2))])
def test_emulated_gradient_body_vs_oracle_differences(S, ns, F, L, N, D, K, isBL, kw):
    import run_emul as E
    from extrack_amd import synth
    from oracle import oracle_np as O
    Ds, T, Fs = _model(S, S * 10 + F)
    Cs = synth.brownian_tracks(N, L, Ds, T, Fs, seed=S + F, dims=D)
    ds2, cell, pBL, min_len = 2 * Ds * 0.02, [1.0], 0.1, 3
    le = np.array([0.02, 0.025, 0.03][:K])
    dirs = model_directions(S, K, ns, ds2, T, le, cell)
    ll, tot, g = E.run_grad(Cs, le[None, None], np.sqrt(ds2), Fs, T, pBL, isBL, O.p_stay_table(np.sqrt(ds2), S, ns, cell), ns, F, min_len,
                            [d[1] for d in dirs], **kw)
    ref = O.proba_cs(Cs, le[None, None], np.sqrt(ds2), Fs, T, pBL, isBL, cell, ns, F, min_len)
    assert np.abs(ll - ref).max() < 1e-10 and abs(tot - ref.sum()) < 1e-12 * abs(tot)
    fd = oracle_fd_gradient(Cs, le, ds2, Fs, T, pBL, isBL, cell, ns, F, min_len, dirs)
    rel = np.abs(g - fd) / np.maximum(np.abs(fd), 1e-3 * np.abs(fd).max())
    assert rel.max() < 1e-6, [(d[0], a, b) for d, a, b, r in zip(dirs, g, fd, rel) if r > 1e-6]


@pytest.mark.parametrize("F,L,N,D,K,isBL", [(4, 9, 10, 2, 1, 1), (6, 12, 8, 2, 1, 0), (5, 7, 5, 2, 2, 1), (7, 10, 3, 3, 1, 1), (4, 2, 5, 1, 1, 1), (6, 3, 4, 3, 3, 0),
                                            (6, 36, 3, 2, 1, 1)])
def test_emulated_register_resident_gradient_body(F, L, N, D, K, isBL):
    """The register-resident 2-state body (xt_reg2.h: state and tangents in VGPRs, lane exchanges; passes of <= 8 directions) on CPU
    threads, against Richardson differences of the pinned oracle along every model direction (10 - 12 directions: two passes)."""
    import run_emul as E
    from extrack_amd import synth
    from oracle import oracle_np as O
    S, ns = 2, 1
    Ds, T, Fs = _model(S, S * 10 + F)
    Cs = synth.brownian_tracks(N, L, Ds, T, Fs, seed=S + F, dims=D)
    ds2, cell, pBL, min_len = 2 * Ds * 0.02, [1.0], 0.1, 3
    le = np.array([0.02, 0.025, 0.03][:K])
    dirs = model_directions(S, K, ns, ds2, T, le, cell)
    ll, tot, g = E.run_grad(Cs, le[None, None], np.sqrt(ds2), Fs, T, pBL, isBL, O.p_stay_table(np.sqrt(ds2), S, ns, cell), ns, F, min_len,
                            [d[1] for d in dirs], generic_g=2)
    ref = O.proba_cs(Cs, le[None, None], np.sqrt(ds2), Fs, T, pBL, isBL, cell, ns, F, min_len)
    assert np.abs(ll - ref).max() < 1e-10 and abs(tot - ref.sum()) < 1e-12 * abs(tot)
    fd = oracle_fd_gradient(Cs, le, ds2, Fs, T, pBL, isBL, cell, ns, F, min_len, dirs)
    rel = np.abs(g - fd) / np.maximum(np.abs(fd), 1e-3 * np.abs(fd).max())
    assert rel.max() < 1e-6, [(d[0], a, b) for d, a, b, r in zip(dirs, g, fd, rel) if r > 1e-6]


@pytest.mark.parametrize("S,ns,F,L,N,D,K,isBL,gg", [(3, 1, 3, 8, 6, 2, 1, 1, 3), (3, 1, 4, 9, 5, 2, 1, 0, 4), (2, 2, 3, 7, 6, 1, 1, 1, 4), (3, 1, 3, 8, 4, 2, 2, 1, 4),
                                                      (4, 1, 3, 7, 3, 3, 1, 1, 3), (3, 1, 3, 2, 4, 2, 1, 1, 3)])
def test_emulated_register_resident_general_gradient_body(S, ns, F, L, N, D, K, isBL, gg):
    """xt_gradr.h (2 - 4 members per group: state and tangents in registers, LDS exchange rounds; gg = 3 / 4 directions per pass,
    17 - 26 directions: several passes) on CPU threads against Richardson differences of the pinned oracle."""
    import run_emul as E
    from extrack_amd import synth
    from oracle import oracle_np as O
    Ds, T, Fs = _model(S, S * 10 + F)
    Cs = synth.brownian_tracks(N, L, Ds, T, Fs, seed=S + F, dims=D)
    ds2, cell, pBL, min_len = 2 * Ds * 0.02, [1.0], 0.1, 3
    le = np.array([0.02, 0.025, 0.03][:K])
    dirs = model_directions(S, K, ns, ds2, T, le, cell)
    ll, tot, g = E.run_grad(Cs, le[None, None], np.sqrt(ds2), Fs, T, pBL, isBL, O.p_stay_table(np.sqrt(ds2), S, ns, cell), ns, F, min_len,
                            [d[1] for d in dirs], generic_g=gg)
    ref = O.proba_cs(Cs, le[None, None], np.sqrt(ds2), Fs, T, pBL, isBL, cell, ns, F, min_len)
    assert np.abs(ll - ref).max() < 1e-10 and abs(tot - ref.sum()) < 1e-12 * abs(tot)
    fd = oracle_fd_gradient(Cs, le, ds2, Fs, T, pBL, isBL, cell, ns, F, min_len, dirs)
    rel = np.abs(g - fd) / np.maximum(np.abs(fd), 1e-3 * np.abs(fd).max())
    assert rel.max() < 1e-6, [(d[0], a, b) for d, a, b, r in zip(dirs, g, fd, rel) if r > 1e-6]


_REV_CASES = [(3, 1, 3, 8, 6, 2, 1, 1), (3, 1, 4, 9, 5, 2, 1, 0), (2, 2, 4, 7, 6, 1, 1, 1), (3, 1, 3, 8, 4, 2, 2, 1), (4, 1, 3, 7, 3, 3, 1, 1),
              (3, 1, 3, 2, 4, 2, 1, 1), (3, 1, 3, 3, 4, 2, 1, 0), (2, 1, 5, 36, 3, 3, 3, 1), (3, 1, 5, 12, 7, 2, 1, 1)]


@pytest.mark.parametrize("S,ns,F,L,N,D,K,isBL,nbuf", [c + (2,) for c in _REV_CASES] + [c + (1,) for c in _REV_CASES if (c[0], c[2], c[3]) in
                                                                                        ((3, 3, 8), (4, 3, 7), (2, 5, 36), (3, 3, 2))])
def test_emulated_reverse_mode_gradient_body(S, ns, F, L, N, D, K, isBL, nbuf, monkeypatch):
    """xt_rev.h (reverse mode: forward sweep logging the merged state of every group, backward sweep of adjoints, adjoint of the model
    blob contracted with the tangent blocks) on CPU threads against Richardson differences of the pinned oracle along every model
    direction; N is not a multiple of the tracks per workgroup (partial last batch), 2- and 3-position tracks, a track longer than
    the staging block of 32 positions.  nbuf: exchange buffers per track (1 = the variant with two barriers per step the launcher takes
    where a second buffer would cost the second workgroup per CU)."""
    monkeypatch.setenv("XT_EMUL_REV_NBUF", str(nbuf))
    import run_emul as E
    from extrack_amd import synth
    from oracle import oracle_np as O
    Ds, T, Fs = _model(S, S * 10 + F)
    Cs = synth.brownian_tracks(N, L, Ds, T, Fs, seed=S + F, dims=D)
    ds2, cell, pBL, min_len = 2 * Ds * 0.02, [1.0], 0.1, 3
    le = np.array([0.02, 0.025, 0.03][:K])
    dirs = model_directions(S, K, ns, ds2, T, le, cell)
    ll, tot, g = E.run_grad(Cs, le[None, None], np.sqrt(ds2), Fs, T, pBL, isBL, O.p_stay_table(np.sqrt(ds2), S, ns, cell), ns, F, min_len,
                            [d[1] for d in dirs], generic_g=5)
    ref = O.proba_cs(Cs, le[None, None], np.sqrt(ds2), Fs, T, pBL, isBL, cell, ns, F, min_len)
    assert np.abs(ll - ref).max() < 1e-10 and abs(tot - ref.sum()) < 1e-12 * abs(tot)
    fd = oracle_fd_gradient(Cs, le, ds2, Fs, T, pBL, isBL, cell, ns, F, min_len, dirs)
    rel = np.abs(g - fd) / np.maximum(np.abs(fd), 1e-3 * np.abs(fd).max())
    assert rel.max() < 1e-6, [(d[0], a, b) for d, a, b, r in zip(dirs, g, fd, rel) if r > 1e-6]


@pytest.mark.parametrize("S,F,L,N,D,KS,affine,gg", [(2, 4, 9, 7, 2, 1, True, 0), (2, 4, 9, 7, 2, 1, True, 4), (2, 4, 9, 7, 2, 1, True, 5), (3, 3, 8, 5, 2, 2, True, 5),
                                                     (3, 3, 8, 5, 2, 2, False, 5), (3, 3, 8, 5, 2, 2, True, 3), (2, 5, 35, 3, 1, 1, True, 5)])
def test_emulated_gradient_bodies_with_per_peak_localisation_errors(S, F, L, N, D, KS, affine, gg):
    """Per-peak localisation errors (input_LocErr; affine = LocErr_type 4: clip(sigma * slope + offset, 1e-6), extrack/tracking.py:
    946-955): the gradient bodies - gg 0: xt_grad.h, 3 / 4: xt_gradr.h, 5: reverse mode xt_rev.h - against Richardson differences of the
    pinned oracle along every model direction plus slope and offset (some peaks sit below the clip, where the derivative is 0)."""
    import run_emul as E
    from extrack_amd import synth
    from oracle import oracle_np as O
    Ds, T, Fs = _model(S, S * 10 + F)
    Cs = synth.brownian_tracks(N, L, Ds, T, Fs, seed=S + F, dims=D)
    rng = np.random.default_rng(5)
    sig = rng.uniform(0.01, 0.04, (N, L, KS))
    sig[0, 1] = 1e-9  # below the clip of the affine mode
    slope, offset = (1.2, -0.003) if affine else (None, None)
    ds2, cell, pBL, min_len, ns, isBL = 2 * Ds * 0.02, [1.0], 0.1, 3, 1, 1

    def le_of(sl, of):
        return np.maximum(sig * sl + of, 1e-6) if affine else sig

    dirs = [d for d in model_directions(S, 1, ns, ds2, T, np.array([0.02]), cell) if not d[0].startswith("le")]
    tang = [d[1] for d in dirs] + ([dict(slope=1.0), dict(offset=1.0)] if affine else [])
    ll, tot, g = E.run_grad(Cs, sig, np.sqrt(ds2), Fs, T, pBL, isBL, O.p_stay_table(np.sqrt(ds2), S, ns, cell), ns, F, min_len, tang, generic_g=gg,
                            slope=slope, offset=offset)
    ref = O.proba_cs(Cs, le_of(slope or 1.0, offset or 0.0), np.sqrt(ds2), Fs, T, pBL, isBL, cell, ns, F, min_len)
    assert np.abs(ll - ref).max() < 1e-10 and abs(tot - ref.sum()) < 1e-12 * abs(tot)

    def total(x, d, dsl=0.0, dof=0.0):
        return O.proba_cs(Cs, le_of((slope or 1.0) + x * dsl, (offset or 0.0) + x * dof), np.sqrt(ds2 + x * d.get("ds2", 0.0)), Fs + x * d.get("Fs", 0.0),
                          T + x * d.get("T", 0.0), pBL + x * d.get("pBL", 0.0), isBL, cell, ns, F, min_len).sum()

    fd = [_richardson(lambda x: total(x, d), h) for _, _, d, h in dirs]
    if affine:
        fd += [_richardson(lambda x: total(x, {}, dsl=1.0), 1e-4), _richardson(lambda x: total(x, {}, dof=1.0), 1e-6)]
    fd = np.array(fd)
    rel = np.abs(g - fd) / np.maximum(np.abs(fd), 1e-3 * np.abs(fd).max())
    assert rel.max() < 1e-6, (g, fd)


def test_host_chain_rule_matches_finite_differences():
    """params (with expr constraints and bounds) -> model arrays: the complex-step tangents against central differences, for every
    Matrix_type and a D0 sitting exactly at 0 (where d ds / d D is infinite but d ds^2 / d D is not)."""
    from extrack_amd import engine, gradient, tracking as T
    for Matrix_type in (0, 1, 2, 3, 4):
        p = T.generate_params(nb_states=3, LocErr_type=2, nb_dims=2, estimated_Ds=[0.0, 0.05, 0.3], estimated_LocErr=[0.02, 0.03],
                              estimated_Fs=[0.3, 0.45], estimated_transition_rates=[0.05, 0.1, 0.15, 0.2, 0.07, 0.12])
        names = gradient.free_names(p)
        assert "F2" not in names and len(names) == 3 + 2 + 2 + 6 + 1
        tang = gradient.tangent_rows(gradient.model_tangents(p, 0.02, 2, Matrix_type, [1.0, 0.7], names))

        def arrays(q):
            le, Ds, Fs, Tm, pBL, so = T._extract_arrays(q, 0.02, 2, Matrix_type)
            ds2 = 2 * Ds * 0.02
            return dict(locerr=le, ds2=ds2, Fs=Fs, TrMat=Tm, pBL=np.array(pBL), p_stay=engine._p_stay_table(np.sqrt(ds2), 3, 2, [1.0, 0.7]))

        for n, t in zip(names, tang):
            h = 1e-6 * max(abs(p[n].value), 1e-2)
            lo, hi = p.copy(), p.copy()
            hi[n].value = p[n].value + h
            lo[n].value = p[n].value - (0.0 if n == "D0" else h)
            hi.update_constraints()
            lo.update_constraints()
            a, b = arrays(hi), arrays(lo)
            for k in a:
                fd = (a[k] - b[k]) / (h if n == "D0" else 2 * h)
                assert np.allclose(np.asarray(t[k], float), fd, rtol=2e-5, atol=2e-6 * max(1.0, np.abs(fd).max())), (Matrix_type, n, k)


def test_minimiser_uses_the_analytic_gradient():
    """Bounded Rosenbrock-type objective: with fcn_grad the lmfit-compatible minimiser reaches the same optimum with several times
    fewer objective calls (the chain rule through the bounds transform is applied inside)."""
    from extrack_amd.lmfit_compat import Parameters, _own_minimize
    p = Parameters()
    p.add("a", value=-1.0, min=-3, max=3)
    p.add("b", value=2.0, min=0)
    p.add("c", value=0.5)
    p.add("d", expr="a + c")

    def f(q):
        a, b, c = q["a"].value, q["b"].value, q["c"].value
        return (1 - a) ** 2 + 100 * (b - a * a) ** 2 + (q["d"].value - 1.5) ** 2

    def fg(q, names):
        a, b, c = q["a"].value, q["b"].value, q["c"].value
        g = dict(a=-2 * (1 - a) - 400 * a * (b - a * a) + 2 * (a + c - 1.5), b=200 * (b - a * a), c=2 * (a + c - 1.5))
        return f(q), np.array([g[n] for n in names])

    r_fd = _own_minimize(f, p, method="bfgs")
    r_an = _own_minimize(f, p, method="bfgs", fcn_grad=fg)
    for k in ("a", "b", "c"):
        assert abs(r_an.params[k].value - r_fd.params[k].value) < 1e-4
    assert abs(r_an.params["a"].value - 1) < 1e-5 and abs(r_an.params["d"].value - 1.5) < 1e-5
    assert r_an.nfev * 2.5 < r_fd.nfev and r_an.ngev > 0
    r_pw = _own_minimize(f, p, method="powell", fcn_grad=fg)  # derivative-free methods ignore it
    assert r_pw.ngev == 0


def test_non_analytic_expression_is_detected():
    """advisor r2: an ``expr`` using abs / min / max / a comparison cannot be complex-differentiated; ``analytic_support`` reports it and
    ``param_fitting(gradient=None)`` then differences the objective (``gradient='analytic'`` raises)."""
    from extrack_amd import gradient, tracking as T
    p = T.generate_params(nb_states=2, LocErr_type=1, estimated_Ds=[1e-3, 0.25], estimated_LocErr=[0.02], estimated_Fs=[0.6], estimated_transition_rates=0.1)
    names = gradient.free_names(p)
    assert gradient.analytic_support(p, names) is None
    p.add("D1", expr="max(D0, 0.1) + 0.1")
    assert gradient.analytic_support(p, gradient.free_names(p)) is not None
    p.add("D1", expr="D0 + 0.1 if D0 > 0.05 else 0.2")
    assert gradient.analytic_support(p, gradient.free_names(p)) is not None
    p.add("D1", expr="D0 + exp(-D0) * 0.2")
    assert gradient.analytic_support(p, gradient.free_names(p)) is None
    t = gradient.tangent_rows(gradient.model_tangents(p, 0.02, 1, 1, [1.0], gradient.free_names(p)))
    d0 = p["D0"].value
    assert abs(t[0]["ds2"][1] - 2 * 0.02 * (1 - 0.2 * np.exp(-d0))) < 1e-12  # d(2 D1 dt)/d D0 through the expression


def test_gradient_fit_on_foreign_parameters_object():
    """``minimize_with_gradient`` runs the built-in driver on a converted copy of any lmfit-style Parameters mapping (what happens when
    real lmfit is installed) and returns the caller's type with the fitted values."""
    import copy
    from extrack_amd.lmfit_compat import minimize_with_gradient

    class FPar:
        def __init__(self, name, value=None, vary=True, min=-np.inf, max=np.inf, expr=None):
            self.name, self.value, self.vary, self.min, self.max, self.expr = name, value, vary, min, max, expr

    class FPars(dict):
        def update_constraints(self):
            self["d"].value = self["a"].value + self["c"].value

    fp = FPars(a=FPar("a", -1.0, min=-3, max=3), b=FPar("b", 2.0, min=0), c=FPar("c", 0.5), d=FPar("d", 0.0, vary=False, expr="a + c"))

    def f(q):
        a, b = q["a"].value, q["b"].value
        return (1 - a) ** 2 + 100 * (b - a * a) ** 2 + (q["d"].value - 1.5) ** 2

    def fg(q, names):
        a, b, c = q["a"].value, q["b"].value, q["c"].value
        g = dict(a=-2 * (1 - a) - 400 * a * (b - a * a) + 2 * (a + c - 1.5), b=200 * (b - a * a), c=2 * (a + c - 1.5))
        return f(q), np.array([g[n] for n in names])

    r = minimize_with_gradient(f, fp, method="bfgs", fcn_grad=fg)
    assert isinstance(r.params, FPars) and r.params is not fp and fp["a"].value == -1.0
    assert abs(r.params["a"].value - 1) < 1e-5 and abs(r.params["b"].value - 1) < 1e-4 and abs(r.params["d"].value - 1.5) < 1e-5
    assert r.ngev > 0 and abs(r.residual[0]) < 1e-8
