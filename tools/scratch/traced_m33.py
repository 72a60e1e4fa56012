import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import etol_amd as E
from etol_amd import _lib as L, workloads as W
lib = C.CDLL(os.path.join(ROOT, "tests", "harness", "libetol_harness.so"))
lib.harness_traced_model_source.restype = C.c_char_p
src = lib.harness_traced_model_source(2).decode()
for M in (33, 34, 256):
    B = 2
    X, U, _ = W.quadrotor_batch(5, B, M, 0)
    disc = np.zeros(L.PATH_REC); disc[:4] = [L.PATH_DISC, 4.0, 3.2, 0.64]
    ell = E.edge_ellipse(3.2, 2.5, 3.4, 2.6)
    tr = E.Evaluator(0); tr.set_mesh(M, 0.0, W.TF); tr.set_model_source("TracedModel", src, 6, 2, npath=2); tr.set_batch(B)
    tr.set_path(np.zeros((0, L.PATH_REC)), 0, 1)
    bi = E.Evaluator(0); bi.set_mesh(M, 0.0, W.TF); bi.set_model(E.MODEL_QUADROTOR2D, W.QUAD_PARAMS); bi.set_batch(B)
    bi.set_path(np.array([disc, ell]), 0, 1)
    b = bi.eval_host(X, U)
    for rep in range(3):
        a = tr.eval_host(X, U)
        print(M, rep, [float(np.abs(p - q).max()) for p, q in zip(a, b)], float(np.abs(a[0]).max()))
