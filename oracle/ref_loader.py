"""Load the *reference* ExTrack modules by path (test infrastructure only).

This file is part of the parity oracle tooling: it is used ONLY by
``tests/golden/make_golden.py`` (run in the build container, where
``/root/reference`` is mounted) to generate golden vectors.  It is never
imported by the product package and it cannot work on the GPU box (the
reference does not travel).

Shims needed to import the unmodified reference in this image
(SURVEY.md section 8c):
  * ``lmfit`` is not installed -> a stub module exposing ``Parameters`` /
    ``minimize`` is put in ``sys.modules`` (reference imports it at
    ``extrack/tracking.py:31`` and ``extrack/tracking_0.py:30``).
  * numpy 2.x removed ``np.product`` (used by ``fuse_tracks_general``,
    ``extrack/tracking.py:417-421``) -> alias to ``np.prod``.
  * ``import extrack`` fails on ``xmltodict`` (``extrack/__init__.py:6``) ->
    modules are loaded with ``importlib`` from their file paths.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get("EXTRACK_REFERENCE", "/root/reference")


class _Param:
    def __init__(self, name, value=None, vary=True, min=-np.inf, max=np.inf, expr=None, brute_step=None):
        self.name, self.value, self.vary = name, value, vary
        self.min, self.max, self.expr, self.brute_step = min, max, expr, brute_step


class StubParameters(dict):
    """Just enough of lmfit.Parameters for the reference's extract_params."""

    def add(self, name, value=None, vary=True, min=-np.inf, max=np.inf, expr=None, brute_step=None):
        if expr is not None:
            env = {k: v.value for k, v in self.items()}
            value = eval(expr, {"__builtins__": {}}, env)
        self[name] = _Param(name, value, vary, min, max, expr, brute_step)


def install_shims():
    if "lmfit" not in sys.modules:
        stub = types.ModuleType("lmfit")
        stub.Parameters = StubParameters

        def _minimize(*a, **k):
            raise RuntimeError("lmfit stub: minimize is not available")

        stub.minimize = _minimize
        sys.modules["lmfit"] = stub
    if not hasattr(np, "product"):
        np.product = np.prod


def available():
    return os.path.isfile(os.path.join(REF_ROOT, "extrack", "tracking.py"))


def load(name):
    """name in {'tracking', 'tracking_0', 'simulate_tracks'}."""
    install_shims()
    path = os.path.join(REF_ROOT, "extrack", name + ".py")
    spec = importlib.util.spec_from_file_location("extrack_ref_" + name, path)
    mod = importlib.util.module_from_spec(spec)
    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):
        spec.loader.exec_module(mod)
    return mod


def make_params(**vals):
    install_shims()
    p = sys.modules["lmfit"].Parameters()
    for k, v in vals.items():
        p.add(k, value=v)
    return p


def load_with_package(name):
    """Modules of the reference that import their siblings through the package (``from extrack.tracking import ...``:
    extrack/histograms.py:24, extrack/refined_localization.py:27-29): a stand-in ``extrack`` package whose submodules are the
    path-loaded reference modules is put in ``sys.modules`` first (``import extrack`` itself fails on ``xmltodict``)."""
    install_shims()
    if "extrack" not in sys.modules or not getattr(sys.modules["extrack"], "_ref_stub", False):
        pkg = types.ModuleType("extrack")
        pkg._ref_stub = True
        pkg.__path__ = []
        sys.modules["extrack"] = pkg
        for sub in ("tracking", "tracking_0"):
            m = load(sub)
            sys.modules["extrack." + sub] = m
            setattr(pkg, sub, m)
    return load(name)
