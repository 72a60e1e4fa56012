#!/usr/bin/env python3
"""Generates tests/golden/*.json from first principles (run in the dev container; needs sympy/mpmath).

Nothing here imports the product or the oracle: these fixtures are the independent pin both
are checked against (the reference itself ships no golden vectors: SURVEY.md section 8c).

  lgl.json     LGL nodes / weights / differentiation matrix at 50 digits, rounded to double:
               full matrices for M in {3,4,5,9,33}, probe rows for M = 256
  models.json  symbolic (sympy) f, df/dz, L, dL/dz and the Hessian of cL*L + sum cf_i f_i for the
               three node models, and value/gradient/Hessian of the keep-out rows, evaluated
               in 40-digit arithmetic at seeded points
  ocp2d.json   the analytic optimum of the obstacle-free shipped problem (a convex QP)
"""
import json
import random

import mpmath as mp
import sympy as sp

mp.mp.dps = 50


def lgl(M):
    N = M - 1
    x = [mp.mpf(-1)] + [None] * (N - 1) + [mp.mpf(1)]
    dP = lambda t: mp.diff(lambda s: mp.legendre(N, s), t)
    for k in range(1, N):
        g = -mp.cos(mp.pi * k / N)
        # Newton on (1-t^2) P'_N(t) = N (P_{N-1} - t P_N)
        for _ in range(100):
            pn, pn1 = mp.legendre(N, g), mp.legendre(N - 1, g)
            step = (g * pn - pn1) / ((N + 1) * pn)
            g -= step
            if abs(step) < mp.mpf(10) ** (-45):
                break
        x[k] = g
    pN = [mp.legendre(N, t) for t in x]
    w = [2 / (N * (N + 1) * p * p) for p in pN]
    return x, w, pN


def dmat_row(x, pN, i):
    N = len(x) - 1
    row = []
    for j in range(N + 1):
        if i != j:
            row.append(pN[i] / (pN[j] * (x[i] - x[j])))
        elif i == 0:
            row.append(-mp.mpf(N * (N + 1)) / 4)
        elif i == N:
            row.append(mp.mpf(N * (N + 1)) / 4)
        else:
            row.append(mp.mpf(0))
    return row


def gen_lgl():
    out = {}
    for M in (3, 4, 5, 9, 33):
        x, w, pN = lgl(M)
        out[str(M)] = dict(tau=[float(t) for t in x], w=[float(t) for t in w],
                           D=[[float(v) for v in dmat_row(x, pN, i)] for i in range(M)])
    M = 256
    x, w, pN = lgl(M)
    rows = [0, 1, 2, 100, 127, 128, 254, 255]
    out[str(M)] = dict(tau=[float(t) for t in x], w=[float(t) for t in w], rows=rows,
                       D_rows=[[float(v) for v in dmat_row(x, pN, i)] for i in rows])
    # closed forms (exact): M=3 and M=4
    out["closed"] = {"3": dict(tau=[-1, 0, 1], w=[1 / 3, 4 / 3, 1 / 3], D=[[-1.5, 2, -0.5], [-0.5, 0, 0.5], [0.5, -2, 1.5]]),
                     "4": dict(tau=[-1, -float(mp.sqrt(mp.mpf(1) / 5)), float(mp.sqrt(mp.mpf(1) / 5)), 1],
                               w=[1 / 6, 5 / 6, 5 / 6, 1 / 6])}
    json.dump(out, open("lgl.json", "w"))


# ---- models (restated from DESIGN.md "Node models") ---------------------------------------------
def model_syms(model):
    if model == 0:
        z = sp.symbols("x y u0 u1")
        p = []
        f = [z[2], z[3]]
        L = z[2] ** 2 + z[3] ** 2
    elif model == 1:
        z = sp.symbols("px pz th vx vz om T tq")
        p = sp.symbols("m I g wT wq")
        f = [z[3], z[4], z[5], -(z[6] / p[0]) * sp.sin(z[2]), (z[6] / p[0]) * sp.cos(z[2]) - p[2], z[7] / p[1]]
        L = p[3] * z[6] ** 2 + p[4] * z[7] ** 2
    else:
        z = sp.symbols("pn pe pd ph th ps u v w p q r thr da de dr")
        p = sp.symbols("m Ixx Iyy Izz g qS CL0 CLa CD0 CDk Clda Cmde Cndr V damp wc")
        (pn, pe, pd, ph, th, ps, u, v, w, pr, qr, rr, thr, da, de, dr) = z
        (m, Ixx, Iyy, Izz, g, qS, CL0, CLa, CD0, CDk, Clda, Cmde, Cndr, V, damp, wc) = p
        s, c = sp.sin, sp.cos
        CL = CL0 + CLa * w / V
        CD = CD0 + CDk * CL ** 2
        f = [c(th) * c(ps) * u + (s(ph) * s(th) * c(ps) - c(ph) * s(ps)) * v + (c(ph) * s(th) * c(ps) + s(ph) * s(ps)) * w,
             c(th) * s(ps) * u + (s(ph) * s(th) * s(ps) + c(ph) * c(ps)) * v + (c(ph) * s(th) * s(ps) - s(ph) * c(ps)) * w,
             -s(th) * u + s(ph) * c(th) * v + c(ph) * c(th) * w,
             pr + sp.tan(th) * (s(ph) * qr + c(ph) * rr),
             c(ph) * qr - s(ph) * rr,
             (s(ph) * qr + c(ph) * rr) / c(th),
             rr * v - qr * w - g * s(th) + (thr - qS * CD) / m,
             pr * w - rr * u + g * s(ph) * c(th) - damp * v / m,
             qr * u - pr * v + g * c(ph) * c(th) - qS * CL / m,
             ((Iyy - Izz) * qr * rr + qS * Clda * da - damp * pr) / Ixx,
             ((Izz - Ixx) * pr * rr + qS * Cmde * de - damp * qr) / Iyy,
             ((Ixx - Iyy) * pr * qr + qS * Cndr * dr - damp * rr) / Izz]
        L = wc * (thr ** 2 + da ** 2 + de ** 2 + dr ** 2)
    return list(z), list(p), f, L


PARAMS = {0: [], 1: [1.0, 0.01, 9.81, 1.0, 1.0],
          2: [10.0, 0.8, 1.1, 1.8, 9.81, 120.0, 0.3, 4.5, 0.03, 0.05, 0.08, -0.6, 0.06, 25.0, 0.9, 1.0]}
RANGES = {0: [(0, 7)] * 2 + [(-0.5, 0.5)] * 2,
          1: [(0, 10), (0, 10), (-0.8, 0.8), (-2, 2), (-2, 2), (-1, 1), (5, 15), (-0.1, 0.1)],
          2: [(-100, 100)] * 2 + [(-150, -50), (-0.4, 0.4), (-0.3, 0.3), (-3, 3), (20, 30), (-2, 2), (-2, 2)] +
             [(-0.5, 0.5)] * 3 + [(10, 50)] + [(-0.3, 0.3)] * 3}


def gen_models():
    rnd = random.Random(0xE701)
    out = {}
    for model in (0, 1, 2):
        z, p, f, L = model_syms(model)
        nv, ns = len(z), len(f)
        cL = sp.Symbol("cL")
        cf = sp.symbols(f"cf0:{ns}")
        lagr = cL * L + sum(a * b for a, b in zip(cf, f))
        J = [[sp.diff(fi, zv) for zv in z] for fi in f]
        gL = [sp.diff(L, zv) for zv in z]
        H = [[sp.diff(lagr, z[a], z[b]) for b in range(a + 1)] for a in range(nv)]
        pts = []
        for _ in range(6):
            zv = [rnd.uniform(*r) for r in RANGES[model]]
            cfv = [rnd.uniform(-2, 2) for _ in range(ns)]
            cLv = rnd.uniform(0.1, 2)
            sub = dict(zip(z, zv))
            sub.update(dict(zip(p, PARAMS[model])))
            sub.update(dict(zip(cf, cfv)))
            sub[cL] = cLv
            ev = lambda e: float(sp.N(e.subs(sub), 40)) if not isinstance(e, (int, float)) else float(e)
            pts.append(dict(z=zv, cf=cfv, cL=cLv, f=[ev(e) for e in f], J=[[ev(e) for e in r] for r in J], L=ev(L),
                            gL=[ev(e) for e in gL], H=[ev(H[a][b]) for a in range(nv) for b in range(a + 1)]))
        out[str(model)] = dict(params=PARAMS[model], points=pts)
    # keep-out rows: value, gradient, Hessian (exact)
    x, y, xc, yc, ct, st, asq, bsq, rsq = sp.symbols("x y xc yc ct st asq bsq rsq")
    dx, dy = x - xc, y - yc
    delx, dely = ct * dx - st * dy, st * dx + ct * dy
    ell = asq * bsq - (bsq * delx ** 2 + asq * dely ** 2)
    disc = rsq - (dx ** 2 + dy ** 2)
    rows = []
    for _ in range(6):
        tt = rnd.uniform(-3, 3)
        a2 = rnd.uniform(0.01, 0.5)
        sub = {x: rnd.uniform(0, 7), y: rnd.uniform(0, 7), xc: rnd.uniform(1, 5), yc: rnd.uniform(1, 5),
               ct: sp.cos(tt), st: sp.sin(tt), asq: a2, bsq: 0.2 * a2, rsq: rnd.uniform(0.04, 0.4)}
        ev = lambda e: float(sp.N(e.subs(sub), 40))
        rows.append(dict(x=sub[x], y=sub[y], xc=sub[xc], yc=sub[yc], ct=float(sub[ct]), st=float(sub[st]), asq=a2,
                         bsq=0.2 * a2, rsq=sub[rsq],
                         ell=[ev(ell), ev(sp.diff(ell, x)), ev(sp.diff(ell, y)), ev(sp.diff(ell, x, 2)),
                              ev(sp.diff(ell, x, y)), ev(sp.diff(ell, y, 2))],
                         disc=[ev(disc), ev(sp.diff(disc, x)), ev(sp.diff(disc, y)), -2.0, 0.0, -2.0]))
    out["keepout"] = rows
    json.dump(out, open("models.json", "w"))


def gen_ocp2d():
    # obstacle-free resource/configs/ocp_2d_ex1.xml: minimise int u^2 with xdot=u, x(0)=(1,2),
    # x(16) in [5-0.01,5+0.01] x [4-0.01,4+0.01], |u| <= 0.5: constant control to the nearest
    # point of the terminal box.
    tf = 16.0
    ux, uy = (5 - 0.01 - 1) / tf, (4 - 0.01 - 2) / tf
    json.dump(dict(tf=tf, u=[ux, uy], cost=(ux * ux + uy * uy) * tf, x0=[1.0, 2.0], xf=[1 + ux * tf, 2 + uy * tf]),
              open("ocp2d.json", "w"))


if __name__ == "__main__":
    gen_lgl()
    gen_models()
    gen_ocp2d()
    print("golden fixtures written")
