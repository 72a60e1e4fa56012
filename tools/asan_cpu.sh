#!/bin/bash
# AddressSanitizer run of the host C++ (loader, trace/codegen, NLP iteration with both Newton-step backends) on the
# CPU: builds instrumented copies of libetol_mi355x.so / the test harness under /tmp/asan and drives the oracle-backed
# solves through them.  GPU sanitizers are not available on the pool; the device side is covered by the parity tests.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/asan
mkdir -p $OUT
cd $ROOT
FLAGS="-O1 -g -fsanitize=address -fno-omit-frame-pointer -std=c++17 -fPIC -Iinclude -Ietol_amd/host -fopenmp-simd"
g++ $FLAGS -I/usr/include/libxml2 -shared -o $OUT/libetol_mi355x.so etol_amd/host/TrajectoryOptimizer.cpp etol_amd/host/eMI355X.cpp \
    etol_amd/host/emi_nlp.cpp etol_amd/host/emi_trace.cpp -Letol_amd/lib -lemi355x -lxml2 -Wl,-rpath,$ROOT/etol_amd/lib
g++ $FLAGS -shared -o $OUT/libetol_harness.so tests/harness/etol_harness.cpp -L$OUT -letol_mi355x -Letol_amd/lib -lemi355x -ldl \
    -Wl,-rpath,$OUT -Wl,-rpath,$ROOT/etol_amd/lib
cat > $OUT/run.py <<PY
import ctypes as C, os, sys, tempfile
import numpy as np
ROOT = "$ROOT"
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_xml_fixtures as G
H = C.CDLL("$OUT/libetol_harness.so")
D = C.POINTER(C.c_double)
H.harness_last_message.restype = C.c_char_p
H.harness_traced_model_source.restype = C.c_char_p
for w in (0, 1, 2, 3):
    print("traced source", w, len(H.harness_traced_model_source(w).decode()))
xml = G.write_all(tempfile.mkdtemp())["ocp_2d_ex1.xml"]
H.harness_solve_example1_oracle.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_int, D, C.POINTER(C.c_int), D, D, C.c_int, C.POINTER(C.c_int)]
H.harness_solve_quadrotor_oracle.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, D, C.POINTER(C.c_int), D, D, C.c_int, C.POINTER(C.c_int)]
H.harness_set_linear_solver.argtypes = [C.c_char_p]
orc = (ROOT + "/oracle/liboracle.so").encode()
H.harness_set_scaling.argtypes = [C.c_int]
for sc, ls in ((-1, b"host"), (-1, b"device"), (1, b"host"), (1, b"device")):      # (1: Alg::scaling = "automatic", the ScaledEvaluator wrapper)
    H.harness_set_scaling(sc)
    H.harness_set_linear_solver(ls)
    for wo in (0, 1):
        X, U = np.zeros(128), np.zeros(128); cost, M, it = C.c_double(), C.c_int(), C.c_int()
        rc = H.harness_solve_example1_oracle(xml.encode(), orc, wo, 1e-9, 0, 300, C.byref(cost), C.byref(M), X.ctypes.data_as(D), U.ctypes.data_as(D), 64, C.byref(it))
        print("example1", ls, wo, rc, cost.value, it.value)
    for nd in (0, 2, 5):
        X, U = np.zeros(6 * 64), np.zeros(2 * 64); cost, M, it = C.c_double(), C.c_int(), C.c_int()
        rc = H.harness_solve_quadrotor_oracle(orc, 24, 0.16, nd, 1e-8, 0, C.byref(cost), C.byref(M), X.ctypes.data_as(D), U.ctypes.data_as(D), 64, C.byref(it))
        print("quadrotor", ls, nd, rc, cost.value, it.value, H.harness_last_message().decode())
H.harness_set_scaling(-1)
H.harness_solve_fixedwing_oracle.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, D, C.POINTER(C.c_int), D, D, C.c_int, C.POINTER(C.c_int)]
H.harness_set_linear_solver(b"device")
X, U = np.zeros(12 * 32), np.zeros(4 * 32); cost, M, it = C.c_double(), C.c_int(), C.c_int()
rc = H.harness_solve_fixedwing_oracle(orc, 16, 6.0, 6.0, 1e-7, 0, C.byref(cost), C.byref(M), X.ctypes.data_as(D), U.ctypes.data_as(D), 32, C.byref(it))
print("fixedwing (inertia search)", rc, cost.value, it.value, H.harness_last_message().decode())
# delayed states / controls solved through coupling rows (NlpLink), with and without bound scaling; Jacobian-based defect row weights
H.harness_solve_delay_demo_oracle.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, D, D, C.POINTER(C.c_int)]
for sc in (0, 1):
    for r in (0.0, 0.9):
        Z = np.zeros(10 * 25); cost, it = C.c_double(), C.c_int()
        rc = H.harness_solve_delay_demo_oracle(orc, 24, 0.25, r, 1e-10, 0, sc, C.byref(cost), Z.ctypes.data_as(D), C.byref(it))
        print("delayed", sc, r, rc, cost.value, it.value, H.harness_last_message().decode())
H.harness_set_defect_scaling.argtypes = [C.c_int]
H.harness_set_defect_scaling(1)
H.harness_set_linear_solver(b"host")
for nd in (0, 5):
    X, U = np.zeros(6 * 64), np.zeros(2 * 64); cost, M, it = C.c_double(), C.c_int(), C.c_int()
    rc = H.harness_solve_quadrotor_oracle(orc, 24, 0.16, nd, 1e-8, 0, C.byref(cost), C.byref(M), X.ctypes.data_as(D), U.ctypes.data_as(D), 64, C.byref(it))
    print("quadrotor, jacobian-based defect scaling", nd, rc, cost.value, it.value, H.harness_last_message().decode())
H.harness_set_defect_scaling(0)
# the planned cold-start route (grid Dijkstra) on a layout with a wall and a gap, and on a walled-in start
H.harness_planned_path.argtypes = [C.c_int, D, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, D, D]
discs = np.array([1.10, 2.09, 0.44, 1.96, 2.52, 0.58, 1.75, 1.64, 0.41, 1.23, 2.72, 0.33, 4.04, 3.13, 0.30, 7.21, 5.93, 0.21])
xs, ys = np.zeros(129), np.zeros(129)
print("planned route", H.harness_planned_path(6, discs.ctypes.data_as(D), 1.0, 1.0, 8.0, 6.0, 0.0, 10.0, 129, xs.ctypes.data_as(D), ys.ctypes.data_as(D)), xs[:3], ys[:3])
ring = np.array([v for a in np.linspace(0, 2 * np.pi, 16, endpoint=False) for v in (1 + 0.9 * np.cos(a), 1 + 0.9 * np.sin(a), 0.5)])
print("walled in", H.harness_planned_path(16, ring.ctypes.data_as(D), 1.0, 1.0, 8.0, 6.0, -5.0, 10.0, 129, xs.ctypes.data_as(D), ys.ctypes.data_as(D)))
print("asan run complete")
PY
cd $OUT && ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python run.py
