"""Iteration logs of the quadrotor solve with the host and the device Newton step (diagnostics)."""
import ctypes as C, os, sys
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = C.CDLL(os.path.join(ROOT, "tests", "harness", "libetol_harness.so"))
D = C.POINTER(C.c_double)
H.harness_solve_quadrotor.argtypes = [C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, D,
                                      C.POINTER(C.c_int), D, D, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), D]
H.harness_set_linear_solver.argtypes = [C.c_char_p]
H.harness_last_message.restype = C.c_char_p
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
H.harness_set_linear_solver(sys.argv[1].encode())
cap = nsteps + 10
X, U = np.zeros(6 * cap), np.zeros(2 * cap)
cost, M, it, mit, oerr = C.c_double(), C.c_int(), C.c_int(), C.c_int(), C.c_double()
ndiscs = int(sys.argv[4]) if len(sys.argv) > 4 else 2
import time
t0 = time.time()
rc = H.harness_solve_quadrotor(nsteps, 4.0 / nsteps, ndiscs, 1e-8, int(sys.argv[3]) if len(sys.argv) > 3 else 5, 0, 1e-4, C.byref(cost), C.byref(M), X.ctypes.data_as(D),
                               U.ctypes.data_as(D), cap, C.byref(it), C.byref(mit), C.byref(oerr))
sys.stdout.flush()
print("wall %.2f s" % (time.time() - t0))
print("rc", rc, H.harness_last_message().decode(), "cost", cost.value, "iters", it.value)
