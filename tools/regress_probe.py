"""Which switch took the config-3-sized solve from 16 to 56 last-mesh iterations and the 129-node fixed wing from 11 to 103
(profiles/r02_solve_times.json vs r03_solve_times.json)?  Runs the two problems of tools/solve_times.py under the existing
switches -- Alg::scaling none / automatic, kkt_primal_levels 0 / 1 -- in one process on one box.
-> gpurun_out/regress_probe.json"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import etol_amd as E  # noqa: E402

H = C.CDLL(os.path.join(ROOT, "tests", "harness", "libetol_harness.so"))
D = C.POINTER(C.c_double)
H.harness_solve_quadrotor.argtypes = [C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, D,
                                      C.POINTER(C.c_int), D, D, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), D]
H.harness_solve_fixedwing.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, D, C.POINTER(C.c_int), D, D, C.c_int,
                                      C.POINTER(C.c_int)]
H.harness_last_message.restype = C.c_char_p
H.harness_set_scaling.argtypes = [C.c_int]


def quadrotor(nsteps, nd):
    cap = nsteps + 80
    X, U = np.zeros(6 * cap), np.zeros(2 * cap)
    cost, M, it, mit, oerr = C.c_double(), C.c_int(), C.c_int(), C.c_int(), C.c_double()
    t0 = time.time()
    rc = H.harness_solve_quadrotor(nsteps, 4.0 / nsteps, nd, 1e-8, 0, 0, 1e-4, C.byref(cost), C.byref(M), X.ctypes.data_as(D),
                                   U.ctypes.data_as(D), cap, C.byref(it), C.byref(mit), C.byref(oerr))
    return dict(problem=f"quadrotor {nsteps + 1} nodes {nd} keep-outs", rc=rc, seconds=time.time() - t0, cost=cost.value,
                last_solve_iterations=it.value, meshes=mit.value, message=H.harness_last_message().decode())


def fixedwing(nsteps, tf, lat):
    cap = nsteps + 1
    X, U = np.zeros(12 * cap), np.zeros(4 * cap)
    cost, M, it = C.c_double(), C.c_int(), C.c_int()
    t0 = time.time()
    rc = H.harness_solve_fixedwing(nsteps, tf, lat, 1e-7, 0, C.byref(cost), C.byref(M), X.ctypes.data_as(D), U.ctypes.data_as(D),
                                   cap, C.byref(it))
    return dict(problem=f"fixed wing {nsteps + 1} nodes", rc=rc, seconds=time.time() - t0, cost=cost.value,
                last_solve_iterations=it.value, message=H.harness_last_message().decode())


def main():
    ev = E.Evaluator(0)                      # any context: the kkt_* switches are process-wide
    quadrotor(40, 2)                         # library load, first-call costs
    out = []
    for scaling in (1, 0):
        for primal in (1, 0):
            H.harness_set_scaling(scaling)
            ev.set_option("kkt_primal_levels", primal)
            for r in (quadrotor(1023, 20), fixedwing(128, 12.0, 20.0), quadrotor(255, 2), fixedwing(48, 8.0, 10.0)):
                r.update(scaling="automatic" if scaling else "none", kkt_primal_levels=primal)
                out.append(r)
                print(f"scaling {r['scaling']:9s} primal {primal}  {r['seconds']:6.2f} s  it {r['last_solve_iterations']:4d}  cost {r['cost']:.6f}  "
                      f"{r['problem']}  rc={r['rc']} {r['message']}", flush=True)
    H.harness_set_scaling(-1)
    ev.set_option("kkt_primal_levels", 1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "regress_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
