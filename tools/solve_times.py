"""End-to-end solve times of ETOL::eMI355X on this machine -> gpurun_out/solve_times.json (copy to profiles/).

Problems: the shipped 2-D example (33 nodes, 11 keep-out rows) and the 6-state quadrotor VGP at 41 / 256 /
1024 nodes with 2 / 2 / 20 disc keep-outs.  Every solve goes through setup()/solve() of the C++ class; the
Newton step runs on the host (dense LDL^T) or on the device as `linear_solver` = "auto" decides."""
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_xml_fixtures as G  # noqa: E402

H = C.CDLL(os.path.join(ROOT, "tests", "harness", "libetol_harness.so"))
D = C.POINTER(C.c_double)
H.harness_solve_example1.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_int, D, C.POINTER(C.c_int), D, D, D, C.c_int,
                                     C.POINTER(C.c_int)]
H.harness_solve_quadrotor.argtypes = [C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, D,
                                      C.POINTER(C.c_int), D, D, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), D]
H.harness_last_message.restype = C.c_char_p
H.harness_last_linear_solver.restype = C.c_char_p
H.harness_set_linear_solver.argtypes = [C.c_char_p]


def example1():
    d = tempfile.mkdtemp()
    xml = G.write_all(d)["ocp_2d_ex1.xml"]
    cap = 600
    X, U, T = np.zeros((2, cap)), np.zeros((2, cap)), np.zeros(cap)
    cost, M, it = C.c_double(), C.c_int(), C.c_int()
    t0 = time.time()
    rc = H.harness_solve_example1(xml.encode(), 1, 1e-9, 0, C.byref(cost), C.byref(M), X.ctypes.data_as(D), U.ctypes.data_as(D),
                                  T.ctypes.data_as(D), cap, C.byref(it))
    return dict(problem="shipped ocp_2d_ex1.xml, 11 keep-out rows, mesh refinement automatic", rc=rc, seconds=time.time() - t0,
                cost=cost.value, nodes=M.value, last_solve_iterations=it.value,
                newton_step=H.harness_last_linear_solver().decode())


def quadrotor(nsteps, ndiscs):
    cap = nsteps + 80
    X, U = np.zeros(6 * cap), np.zeros(2 * cap)
    cost, M, it, mit, oerr = C.c_double(), C.c_int(), C.c_int(), C.c_int(), C.c_double()
    t0 = time.time()
    rc = H.harness_solve_quadrotor(nsteps, 4.0 / nsteps, ndiscs, 1e-8, 0, 0, 1e-4, C.byref(cost), C.byref(M), X.ctypes.data_as(D),
                                   U.ctypes.data_as(D), cap, C.byref(it), C.byref(mit), C.byref(oerr))
    return dict(problem=f"6-state quadrotor VGP, {nsteps + 1} LGL nodes, {ndiscs} disc keep-outs, fixed mesh", rc=rc,
                seconds=time.time() - t0, cost=cost.value, nodes=M.value, meshes=mit.value, last_solve_iterations=it.value,
                newton_step=H.harness_last_linear_solver().decode(), message=H.harness_last_message().decode())


H.harness_solve_fixedwing.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, D, C.POINTER(C.c_int), D, D, C.c_int,
                                      C.POINTER(C.c_int)]


def fixedwing(nsteps, tf, lateral):
    cap = nsteps + 1
    X, U = np.zeros(12 * cap), np.zeros(4 * cap)
    cost, M, it = C.c_double(), C.c_int(), C.c_int()
    t0 = time.time()
    rc = H.harness_solve_fixedwing(nsteps, tf, lateral, 1e-7, 0, C.byref(cost), C.byref(M), X.ctypes.data_as(D), U.ctypes.data_as(D),
                                   cap, C.byref(it))
    return dict(problem=f"12-state fixed wing, {nsteps + 1} LGL nodes, {tf:g} s, {lateral:g} m lateral offset, fixed mesh", rc=rc,
                seconds=time.time() - t0, cost=cost.value, nodes=M.value, last_solve_iterations=it.value,
                newton_step=H.harness_last_linear_solver().decode(), message=H.harness_last_message().decode())


def main():
    out = []
    example1()                      # first call pays context creation / library load
    out.append(example1())
    for nsteps, nd in ((40, 2), (255, 2), (1023, 20)):
        out.append(quadrotor(nsteps, nd))
    # the same 41-node problem with the Newton step forced onto the other backend
    H.harness_set_linear_solver(b"device")
    r = quadrotor(40, 2)
    r["problem"] += " (linear_solver=device)"
    out.append(r)
    for nsteps, tf, lat in ((48, 8.0, 10.0), (128, 12.0, 20.0)):
        out.append(fixedwing(nsteps, tf, lat))
    H.harness_set_linear_solver(b"auto")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "solve_times.json"), "w") as f:
        json.dump(out, f, indent=1)
    for r in out:
        print(f"{r['seconds']:8.2f} s  rc={r['rc']}  cost={r['cost']:.6f}  {r['problem']}  [{r['newton_step']}]")


if __name__ == "__main__":
    main()
