# Do concurrent contexts disturb each other?  T threads, each its own Evaluator: evaluation, Hessian and the
# KKT factor / low-rank / solve path repeated; every repeat must be bit-identical to the thread's first result.
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch  # noqa
import etol_amd as E
from etol_amd import workloads as W

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
M = int(sys.argv[3]) if len(sys.argv) > 3 else 65
MODE = sys.argv[4] if len(sys.argv) > 4 else "full"      # full | nolr | lu | solve2
bad = {}

def work(tid):
    rng = np.random.default_rng(7)          # same data in every thread
    ns, nc, _ = E.model_dims(1)
    nv, nh = ns + nc, (ns + nc) * (ns + nc + 1) // 2
    ev = E.Evaluator(0)
    ev.set_mesh(M, 0.0, 4.0); ev.set_model(1, W.QUAD_PARAMS); ev.set_batch(1)
    recs = np.array([[1, 4.0, 3.2, 0.64, 0, 0, 0, 0], [1, 6.3, 4.4, 0.49, 0, 0, 0, 0]], dtype=float)
    ev.set_path(recs)
    X = rng.standard_normal((1, ns, M)); U = rng.standard_normal((1, nc, M))
    lamF = rng.standard_normal((1, ns, M)); lamC = rng.standard_normal((1, 2, M))
    Qblk = np.zeros((nh, M))
    for k in range(M):
        A = rng.standard_normal((nv, nv)); Qk = A @ A.T + nv * np.eye(nv)
        for v in range(nv):
            for q in range(v + 1): Qblk[v * (v + 1) // 2 + q, k] = Qk[v, q]
    Jblk = rng.standard_normal((ns * nv, M))
    for i in range(ns): Jblk[i * nv + i] += np.diag(ev.D)
    fixed = np.zeros(nv * M, dtype=np.uint8); fixed[np.arange(ns) * M] = 1
    r = 12
    node = rng.integers(0, M, r).astype(np.int32); vec = rng.standard_normal((r, nv)); delta = np.abs(rng.standard_normal(r)) * 0.1 + 0.01
    rhs = rng.standard_normal((3, (nv + ns) * M))
    ref = None
    for rep in range(REPS):
        out = {}
        RES, VALS, COST = ev.eval_host(X, U)
        out["RES"], out["VALS"], out["COST"] = RES.copy(), VALS.copy(), COST.copy()
        out["H"] = np.array(ev.hess_host(X, U, lamF, lamC, 1.0)).copy()
        if MODE == "lu": ev.set_option("kkt_method", 0)
        assert ev.kkt_factor(Qblk, Jblk, fixed, 1e-9) == 0
        if MODE != "nolr": out["exact"] = np.array([ev.kkt_lowrank(node, vec, delta)])
        out["sol"] = np.array(ev.kkt_solve(rhs)).copy()
        if MODE == "solve2":
            for again in range(3):
                s2 = np.array(ev.kkt_solve(rhs))
                if not np.array_equal(s2, out["sol"]): bad.setdefault((tid, "same-factor"), []).append((rep, float(np.abs(s2 - out["sol"]).max())))
        if ref is None: ref = out
        else:
            for k in out:
                if not np.array_equal(out[k], ref[k]):
                    bad.setdefault((tid, k), []).append((rep, float(np.abs(out[k] - ref[k]).max())))
    ev.close()

t0 = time.time()
ths = [threading.Thread(target=work, args=(i,)) for i in range(T)]
[t.start() for t in ths]; [t.join() for t in ths]
print(f"{MODE} threads {T} reps {REPS} M {M}: {time.time()-t0:.1f} s; mismatching (thread, output): {len(bad)}")
for k, v in sorted(bad.items())[:20]: print(k, len(v), v[:3])
