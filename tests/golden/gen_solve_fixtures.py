"""Solve-level golden fixtures from an INDEPENDENT optimiser (build container, CPU only).

The NLP that ePSOPT::setup / addBounds define (reference src/ePSOPT/ePSOPT.cpp:40-81,125-155) is restated here on
the CPU oracle's functions (oracle/emi_oracle.c through tests/oracle_lib.py: values, complex-step first derivatives,
long-double D.X) and solved with scipy's SLSQP, then polished to a KKT point by Newton's method on the active set
(exact first derivatives; Lagrangian Hessian from orc_hess).  Nothing of the product (etol_amd/) takes part: no
interior point, no structured factorisation, no HIP.  The GPU tests (tests/test_gpu_solve.py) assert that
ETOL::eMI355X::solve() lands within 1e-6 (north_star's tolerance) of these trajectories.

    variables    z = [X (ns x M), U (nc x M)], node index fastest
    minimise     h sum_k w_k L(x_k, u_k)                                   (integrand_cost, :186-216)
    subject to   D.X - h f(X, U) = 0                                       (dae :252-260 + PSOPT's LGL defects)
                 -1000 <= c_j(x_k, t_k) <= 0                               (path rows :261-270, bounds :147-150)
                 x(t0) = x0,  xf - tol <= x(tf) <= xf + tol                (events :281-291, :137-141)
                 state / control boxes at every node                       (:134-135, :143-146)

Extra starts: tests/golden/solve_starts.json holds trajectories rounded to three decimals (written by
tests/test_gpu_solve.py on a GPU box when a solve ends at an optimum not stored yet); they are STARTS for the same
SLSQP + polish, nothing more: what is stored is what the independent optimiser converges to and verifies.

Problems: (1) the shipped resource/configs/ocp_2d_ex1.xml with all 11 rows, 33 nodes;  (2) the 6-state quadrotor of
tests/harness/etol_harness.cpp (configure_quadrotor), 41 nodes, 2 disc keep-outs.  Several starts per problem: every
distinct local optimum found is stored (keep-outs make the problem non-convex; which side a trajectory passes is a
property of the start, not of the optimiser).

    python tests/golden/gen_solve_fixtures.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402


from indep_nlp import least_violation, mip_problem, quad_problem, shipped_problem, solve_all  # noqa: E402


def extra_starts(name, P):
    path = os.path.join(HERE, "solve_starts.json")
    if not os.path.exists(path):
        return []
    return [np.concatenate([np.ravel(e["X"]), np.ravel(e["U"])[: P.nc * P.M]]) for e in json.load(open(path)).get(name, [])]


def main():
    out = {}
    P, node_t = shipped_problem()
    sols = solve_all(P, [0.0, 0.6, -0.6, 1.5, -1.5], "ocp_2d_ex1", extra_starts("ocp_2d_ex1", P))
    out["ocp_2d_ex1"] = dict(M=P.M, tf=P.tf, node_t=node_t.tolist(),
                             optima=[dict(cost=s["cost"], X=P.split(s["z"])[0][0].tolist(), U=P.split(s["z"])[1][0].tolist(),
                                          kkt=s["kkt"]) for s in sols])
    Q = quad_problem()
    sols = solve_all(Q, [0.0, 1.0, -1.0, 2.5, -2.5], "quadrotor_41", extra_starts("quadrotor_41", Q))
    out["quadrotor_41"] = dict(M=Q.M, tf=Q.tf, optima=[dict(cost=s["cost"], X=Q.split(s["z"])[0][0].tolist(),
                                                           U=Q.split(s["z"])[1][0].tolist(), kkt=s["kkt"]) for s in sols])
    Pm = mip_problem()
    v = least_violation(Pm, [0.0, 0.8, -0.8, 2.0, -2.0])
    print(f"  mip_2d_ex1 (17 nodes, tf = 8): smallest path-row violation reachable = {v:.3e}")
    sols = solve_all(Pm, [0.0, 0.8, -0.8, 2.0, -2.0], "mip_2d_ex1", extra_starts("mip_2d_ex1", Pm)) if v < 1e-9 else []
    out["mip_2d_ex1"] = dict(M=Pm.M, tf=Pm.tf, least_path_violation=v, feasible=bool(v < 1e-9),
                             optima=[dict(cost=s["cost"], X=Pm.split(s["z"])[0][0].tolist(), U=Pm.split(s["z"])[1][0].tolist(),
                                          kkt=s["kkt"]) for s in sols])
    json.dump(dict(source="scipy SLSQP + active-set Newton polish on oracle/emi_oracle.c functions (gen_solve_fixtures.py)",
                   problems=out), open(os.path.join(HERE, "solve_optima.json"), "w"))
    print("wrote solve_optima.json:", {k: len(v["optima"]) for k, v in out.items()})


if __name__ == "__main__":
    main()
