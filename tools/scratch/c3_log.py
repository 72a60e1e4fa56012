# iteration log of the config-3-sized single solve (1024 nodes, 20 keep-outs) -> stdout
import ctypes as C, os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = C.CDLL(ROOT + "/tests/harness/libetol_harness.so")
D = C.POINTER(C.c_double)
H.harness_solve_quadrotor.argtypes = [C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, D,
                                      C.POINTER(C.c_int), D, D, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), D]
H.harness_last_message.restype = C.c_char_p
nsteps, nd = int(sys.argv[1]), int(sys.argv[2]); pl = int(sys.argv[3])
cap = nsteps + 80
for rep in range(2):
    X, U = np.zeros(6 * cap), np.zeros(2 * cap)
    cost, M, it, mit, oerr = C.c_double(), C.c_int(), C.c_int(), C.c_int(), C.c_double()
    t0 = time.time()
    rc = H.harness_solve_quadrotor(nsteps, 4.0 / nsteps, nd, 1e-8, pl if rep else 0, 0, 1e-4, C.byref(cost), C.byref(M), X.ctypes.data_as(D),
                                   U.ctypes.data_as(D), cap, C.byref(it), C.byref(mit), C.byref(oerr))
    sys.stdout.flush()
    print("rc", rc, H.harness_last_message().decode(), cost.value, it.value, "%.2fs" % (time.time() - t0), flush=True)
