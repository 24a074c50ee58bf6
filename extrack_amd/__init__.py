"""extrack_amd - MI355X (gfx950) implementation of ExTrack's track-likelihood hot path.

Public surface mirrors ``extrack.tracking`` for this path only:
``param_fitting``, ``predict_Bs``, ``cum_Proba_Cs``, ``extract_params``, ``generate_params``, ``get_params``
(and ``extrack.histograms.len_hist`` in ``extrack_amd.histograms``).
The recursion runs in hand-written HIP kernels behind the C ABI of ``include/extrack_hip.h``;
importing this package does not touch the GPU, calling into it without the built library or without
a gfx950 device raises.
"""
from . import histograms, tracking  # noqa: F401
from .lmfit_compat import Parameters, minimize  # noqa: F401
from .tracking import (P_Cs_inter_bound_stats, Proba_Cs, TrackSet, cum_Proba_Cs, extract_params, generate_params, get_params,  # noqa: F401
                       param_fitting, predict_Bs)

__version__ = "0.1.0"
