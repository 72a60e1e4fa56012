# fixed-wing lateral-offset problem through ETOL::eMI355X on the GPU (harness_solve_fixedwing)
import ctypes as C, sys, time, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = C.CDLL(ROOT + "/tests/harness/libetol_harness.so")
D = C.POINTER(C.c_double)
H.harness_solve_fixedwing.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, D, C.POINTER(C.c_int), D, D, C.c_int, C.POINTER(C.c_int)]
H.harness_last_message.restype = C.c_char_p
H.harness_set_linear_solver.argtypes = [C.c_char_p]
for solver in sys.argv[1:] or ["host", "device"]:
    H.harness_set_linear_solver(solver.encode())
    for n, tf, lat in ((24, 8.0, 10.0), (48, 8.0, 10.0), (64, 12.0, 20.0)):
        cap = n + 5
        X, U = np.zeros(12 * cap), np.zeros(4 * cap)
        cost, M, it = C.c_double(), C.c_int(), C.c_int()
        t0 = time.time()
        rc = H.harness_solve_fixedwing(n, tf, lat, 1e-7, 0, C.byref(cost), C.byref(M), X.ctypes.data_as(D), U.ctypes.data_as(D), cap, C.byref(it))
        print(solver, n, tf, lat, "rc", rc, H.harness_last_message().decode(), cost.value, it.value, "%.2fs" % (time.time() - t0), flush=True)
