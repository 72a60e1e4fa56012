#!/usr/bin/env python3
"""Wall time of the pieces of a Newton step on the device, per scenario: factorisation, low-rank correction (r columns), plain
solve, refined solve -- through the single entry points and through the batched ones (n contexts), one mesh size per call.
   python tools/kkt_times.py --nodes 1024 --batch 1,4,16,32 --lowrank 770   -> gpurun_out/kkt_times.jsonl"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401

import etol_amd as E
from etol_amd import _lib as L
from etol_amd import workloads as W


def problem(ev, M, ns, nv, rng):
    nh = nv * (nv + 1) // 2
    A = rng.standard_normal((M, nv, nv))
    Q = A @ A.transpose(0, 2, 1) + nv * np.eye(nv)
    Qblk = np.zeros((nh, M))
    for v in range(nv):
        for q in range(v + 1):
            Qblk[v * (v + 1) // 2 + q] = Q[:, v, q]
    Jblk = rng.standard_normal((ns * nv, M))
    for i in range(ns):
        Jblk[i * nv + i] += np.diag(ev.D)
    fixed = np.zeros(nv * M, dtype=np.uint8)
    fixed[np.arange(ns) * M] = 1
    return Qblk, Jblk, fixed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1024)
    ap.add_argument("--batch", default="1,4,16,32")
    ap.add_argument("--lowrank", type=int, default=770)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--opt", default="", help="process-wide kkt_* options, name=value,...")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "kkt_times.jsonl"))
    a = ap.parse_args()
    lib = L.load()
    M, ns, nv = a.nodes, 6, 8
    N = (2 * ns + 2) * M
    rng = np.random.default_rng(5)
    D = C.POINTER(C.c_double)
    dp = lambda x: x.ctypes.data_as(D)
    ip = lambda x: x.ctypes.data_as(C.POINTER(C.c_int))
    nmax = max(int(b) for b in a.batch.split(","))
    evs = []
    for b in range(nmax):
        ev = E.Evaluator(0)
        ev.set_mesh(M, 0.0, 4.0)
        ev.set_model(1, W.QUAD_PARAMS)
        ev.set_batch(1)
        evs.append(ev)
    for kv in [x for x in a.opt.split(",") if x]:
        evs[0].set_option(kv.split("=")[0], int(kv.split("=")[1]))
    probs = [problem(evs[0], M, ns, nv, rng) for _ in range(min(nmax, 4))]
    probs = [probs[b % len(probs)] for b in range(nmax)]
    r = min(a.lowrank, M)
    node = np.sort(rng.choice(M, size=r, replace=False)).astype(np.int32)
    vec = rng.standard_normal((r, nv)) * 0.2
    delta = np.full(r, 0.5)
    rhs = [rng.standard_normal(N) for _ in range(nmax)]
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    # single entry points (context 0), warm
    ev = evs[0]
    ev.kkt_factor(*probs[0], dc=1e-9)
    t = {}
    for name, fn in (("factor", lambda: ev.kkt_factor(*probs[0], dc=1e-9)), ("lowrank", lambda: ev.kkt_lowrank(node, vec, delta)),
                     ("solve", lambda: ev.kkt_solve(rhs[0]))):
        fn()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        t[name] = 1e3 * (time.perf_counter() - t0) / a.reps
    w = rhs[0].copy()
    rel, nsv, rev, stat = C.c_double(), C.c_int(), C.c_int(), C.c_int()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        w[:] = rhs[0]
        lib.emi_kkt_solve_refined(ev.ctx, dp(w), 1e-9, 8, C.byref(rel), C.byref(nsv), C.byref(rev), C.byref(stat))
    t["refined"] = 1e3 * (time.perf_counter() - t0) / a.reps
    rec = dict(nodes=M, form="single", n=1, lowrank_columns=r, ms=t, refined_solves=nsv.value, refined_rel=rel.value)
    print(json.dumps(rec), flush=True)
    open(a.out, "a").write(json.dumps(rec) + "\n")
    for n in [int(b) for b in a.batch.split(",")]:
        ctxs = (C.c_void_p * n)(*[e.ctx for e in evs[:n]])
        Qp = (D * n)(*[dp(p[0]) for p in probs[:n]])
        Jp = (D * n)(*[dp(p[1]) for p in probs[:n]])
        Fp = (C.POINTER(C.c_ubyte) * n)(*[p[2].ctypes.data_as(C.POINTER(C.c_ubyte)) for p in probs[:n]])
        dc = np.full(n, 1e-9)
        info = np.zeros(n, dtype=np.int32)
        t = {}
        lib.emi_kkt_factor_batch(n, ctxs, Qp, Jp, Fp, dp(dc), ip(info))
        t0 = time.perf_counter()
        for _ in range(a.reps):
            st = lib.emi_kkt_factor_batch(n, ctxs, Qp, Jp, Fp, dp(dc), ip(info))
        t["factor"] = 1e3 * (time.perf_counter() - t0) / a.reps / n
        assert st == 0 and np.all(info == 0), (st, info)
        t0 = time.perf_counter()
        for e in evs[:n]:
            e.kkt_lowrank(node, vec, delta)
        t["lowrank"] = 1e3 * (time.perf_counter() - t0) / n
        work = [x.copy() for x in rhs[:n]]
        Rp = (D * n)(*[dp(x) for x in work])
        lib.emi_kkt_solve_batch(n, ctxs, Rp)
        t0 = time.perf_counter()
        for _ in range(a.reps):
            lib.emi_kkt_solve_batch(n, ctxs, Rp)
        t["solve"] = 1e3 * (time.perf_counter() - t0) / a.reps / n
        relv = np.zeros(n)
        nsv, rev, stat = (np.zeros(n, dtype=np.int32) for _ in range(3))
        t0 = time.perf_counter()
        for _ in range(a.reps):
            for x, b0 in zip(work, rhs):
                x[:] = b0
            lib.emi_kkt_solve_refined_batch(n, ctxs, Rp, dp(dc), 8, dp(relv), ip(nsv), ip(rev), ip(stat))
        t["refined"] = 1e3 * (time.perf_counter() - t0) / a.reps / n
        rec = dict(nodes=M, form="batched", opt=a.opt, n=n, lowrank_columns=r, ms_per_scenario=t, refined_solves=float(nsv.mean()), refined_rel=float(relv.max()))
        print(json.dumps(rec), flush=True)
        open(a.out, "a").write(json.dumps(rec) + "\n")
    for e in evs:
        e.close()


if __name__ == "__main__":
    main()
