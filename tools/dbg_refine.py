import ctypes as C, numpy as np, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = C.CDLL(os.path.join(ROOT, "tests", "harness", "libetol_harness.so"))
D = C.POINTER(C.c_double)
H.harness_solve_quadrotor.argtypes = [C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, D,
                                      C.POINTER(C.c_int), D, D, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), D]
X, U = np.zeros(6 * 160), np.zeros(2 * 160)
cost, M, it, mit, oerr = C.c_double(), C.c_int(), C.c_int(), C.c_int(), C.c_double()
rc = H.harness_solve_quadrotor(12, 4.0 / 12, 1, 1e-8, 5, 1, 1e-4, C.byref(cost), C.byref(M), X.ctypes.data_as(D), U.ctypes.data_as(D), 160, C.byref(it), C.byref(mit), C.byref(oerr))
m = M.value
print(rc, cost.value, m, mit.value, oerr.value)
U = U[:2*m].reshape(2, m); X = X[:6*m].reshape(6, m)
np.set_printoptions(precision=3, linewidth=200, suppress=True)
print("T", U[0]); print("tau", U[1]); print("theta", X[2])
