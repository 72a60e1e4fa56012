"""etol_amd -- MI355X-native collocation backend for ETOL (eMI355X).

Python side of the package: a ctypes mirror of the C ABI (include/emi355x.h)
and helpers for the synthetic workloads.  The product is the C/C++/HIP code in
etol_amd/csrc (kernels + C ABI) and etol_amd/host (ETOL::eMI355X).
"""
from ._lib import (EVAL_ALL, EVAL_DEFECT, EVAL_NODES, EVAL_NOJAC, MODEL_FIXEDWING12,  # noqa: F401
                   MODEL_POINTMASS2D, MODEL_QUADROTOR2D, MODEL_SOURCE, PATH_DISC, PATH_ELLIPSE, PATH_REC,
                   PATH_TRACK, EmiError, load)
from .evaluator import Evaluator, edge_ellipse, lgl, model_dims, track_centres  # noqa: F401
