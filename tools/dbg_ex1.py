"""Iteration log of the shipped 2-D example through eMI355X (diagnostics)."""
import ctypes as C, os, sys
import numpy as np
import torch  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_xml_fixtures as G
import tempfile
d = tempfile.mkdtemp()
paths = G.write_all(d)
H = C.CDLL(os.path.join(ROOT, "tests", "harness", "libetol_harness.so"))
D = C.POINTER(C.c_double)
H.harness_solve_example1.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_int, D, C.POINTER(C.c_int), D, D, D, C.c_int, C.POINTER(C.c_int)]
H.harness_last_message.restype = C.c_char_p
H.harness_set_traced.argtypes = [C.c_int]
H.harness_set_traced(int(os.environ.get("EMI_TRACED", "0")))
cap = 600
X, U, T = np.zeros((2, cap)), np.zeros((2, cap)), np.zeros(cap)
cost, M, iters = C.c_double(), C.c_int(), C.c_int()
xml = paths["ocp_2d_ex1.xml"] if isinstance(paths, dict) else os.path.join(d, "ocp_2d_ex1.xml")
for rep in range(int(os.environ.get("EMI_REPS", "1"))):
  rc = H.harness_solve_example1(xml.encode(), 1, 1e-9, int(sys.argv[1]) if len(sys.argv) > 1 else 5, C.byref(cost), C.byref(M), X.ctypes.data_as(D),
                              U.ctypes.data_as(D), T.ctypes.data_as(D), cap, C.byref(iters))
  sys.stdout.flush()
  print("rc", rc, H.harness_last_message().decode(), cost.value, M.value, iters.value)
