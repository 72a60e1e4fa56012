"""Reference-EXECUTED golden vectors (build container only: needs /root/reference and g++).

Builds oracle/_ref/ref_vectors (oracle/Makefile target `ref`: the reference's own
src/Examples/Dymos/etol_dymos_example1.cpp and include/ETOL/TrajectoryOptimizer.hpp compiled where
they lie, driven by oracle/ref_driver.cpp), runs it on the DATA of the shipped
resource/configs/ocp_2d_ex1.xml plus seeded node points, and writes

    tests/golden/ref_dymos_ex1.json   node callbacks of the example (values and analytic partials)
    tests/golden/ref_interp.json      linear_interpolation, header template and the example's own
    tests/golden/ref_traj.json        extractTraj / scaleTraj / offsetTraj

Inputs AND the reference's outputs are stored (json floats round-trip exactly).  What the example
wraps around the constraint values -- exp(g) - 1 on values, exp(exp(g)) on the ellipse partials,
exp(g) on the moving-disc partials (etol_dymos_example1.cpp:219,244-246,275,297-298) -- is left
as the reference printed it; the tests apply the same wrapper to the build's numbers.

    python tests/golden/gen_ref_vectors.py
"""
import json
import os
import subprocess
import sys
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("ETOL_REFERENCE", "/root/reference")


def hexf(v):
    return float(v).hex()


def run_ref(text):
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_vectors")
    out = subprocess.run([exe], input=text, capture_output=True, text=True, check=True).stdout
    rows = []
    for line in out.splitlines():
        tag, *vals = line.split()
        rows.append((tag, [float.fromhex(v) for v in vals]))
    return rows


def shipped_problem():
    """polygons and tracks of resource/configs/ocp_2d_ex1.xml (data)"""
    root = ET.parse(os.path.join(REF, "resource", "configs", "ocp_2d_ex1.xml")).getroot()
    polys = [[(float(c.get("x")), float(c.get("y"))) for c in b.findall("corner")]
             for b in root.find("exzones").findall("border")]
    tracks = []
    for tr in root.find("mexzones").findall("track"):
        wps = tr.findall("waypoint")
        tracks.append(dict(radius=float(tr.get("radius")), t=[float(w.get("t")) for w in wps],
                           x=[float(w.findall("datum")[0].text) for w in wps],
                           y=[float(w.findall("datum")[1].text) for w in wps]))
    return dict(nsteps=int(root.get("nsteps")), dt=float(root.get("dt")), polys=polys, tracks=tracks)


def edge_tuple(xa, ya, xb, yb):
    """the tuple setExz stores per polygon edge (etol_dymos_example1.cpp:316-326); setExz itself
    needs a TrajectoryOptimizer object and cannot be run here (oracle/ref_driver.cpp)"""
    xc = (xb + xa) / 2.0
    m = (yb - ya) / (xb - xa)
    yc = ya + m * (xc - xa)
    radsq = (xc - xa) ** 2 + (yc - ya) ** 2
    tt = -1.0 * np.arctan2(yc - ya, xc - xa)
    return [float(xc), float(yc), float(radsq), float(tt)]


def dymos_case(name, polys, tracks, node_t, B, seed):
    rng = np.random.default_rng(seed)
    edges, exz = [], []
    for poly in polys:
        for i in range(len(poly)):
            (xa, ya), (xb, yb) = poly[i], poly[(i + 1) % len(poly)]
            edges.append([xa, ya, xb, yb])
            exz.append(edge_tuple(xa, ya, xb, yb))
    M = len(node_t)
    # node points: a band around the obstacles and the moving zones, so that the wrapped values
    # exp(g) - 1 keep their digits (g of order 1, not -40)
    X = np.empty((B, 2, M))
    X[:, 0] = rng.uniform(0.8, 3.8, (B, M))
    X[:, 1] = rng.uniform(1.8, 4.2, (B, M))
    U = rng.uniform(-0.5, 0.5, (B, 2, M))
    lines = [f"exz {len(exz)}"] + [" ".join(hexf(v) for v in e) for e in exz]
    lines.append(f"mexz {len(tracks)}")
    for tr in tracks:
        lines.append(f"{hexf(tr['radius'])} {len(tr['t'])}")
        lines += [f"{hexf(t)} {hexf(x)} {hexf(y)}" for t, x, y in zip(tr["t"], tr["x"], tr["y"])]
    lines.append(f"points {B * M}")
    for b in range(B):
        for k in range(M):
            lines.append(" ".join(hexf(v) for v in (X[b, 0, k], X[b, 1, k], node_t[k], U[b, 0, k], U[b, 1, k])))
    rows = run_ref("\n".join(lines) + "\n")
    ne, nt = len(exz), len(tracks)
    per = 6 + 2 * ne + 2 * nt + 2 * nt
    assert len(rows) == per * B * M
    out = dict(L=np.empty((B, M)), L_p=np.empty((B, 4, M)), F=np.empty((B, 2, M)), F_p=np.empty((B, 2, 4, M)),
               obs=np.empty((B, ne, M)), obs_p=np.empty((B, ne, 2, M)), saa=np.empty((B, nt, M)),
               saa_p=np.empty((B, nt, 2, M)), centres_hdr=np.empty((nt, 2, M)), centres_ex=np.empty((nt, 2, M)))
    it = iter(rows)

    def nxt(tag):
        t, v = next(it)
        assert t == tag, (t, tag)
        return v

    for b in range(B):
        for k in range(M):
            out["L"][b, k] = nxt("obj")[0]
            out["L_p"][b, :, k] = nxt("obj_p")
            out["F"][b, 0, k] = nxt("dx")[0]
            out["F_p"][b, 0, :, k] = nxt("dx_p")
            out["F"][b, 1, k] = nxt("dy")[0]
            out["F_p"][b, 1, :, k] = nxt("dy_p")
            for e in range(ne):
                out["obs"][b, e, k] = nxt("obs")[0]
            for e in range(ne):
                out["obs_p"][b, e, :, k] = nxt("obs_p")[:2]
            for t in range(nt):
                out["saa"][b, t, k] = nxt("saa")[0]
            for t in range(nt):
                out["saa_p"][b, t, :, k] = nxt("saa_p")[:2]
            for t in range(nt):
                out["centres_hdr"][t, :, k] = nxt("interp_hdr")
                out["centres_ex"][t, :, k] = nxt("interp_ex")
    return dict(name=name, M=M, B=B, node_t=list(map(float, node_t)), edges=edges, exz=exz, tracks=tracks,
                X=X.tolist(), U=U.tolist(), ref={k: v.tolist() for k, v in out.items()})


def interp_cases(seed):
    """bracket rule of linear_interpolation: below the first knot, above the last, on knots, between"""
    rng = np.random.default_rng(seed)
    cases = []
    for nway in (2, 3, 7, 24):
        t = np.cumsum(rng.uniform(0.05, 3.0, nway)) - 1.0
        x = rng.uniform(-5, 5, nway)
        y = rng.uniform(-5, 5, nway)
        q = np.concatenate([t, [t[0] - 2.5, t[0] - 1e-9, t[-1] + 1e-9, t[-1] + 7.0], rng.uniform(t[0] - 1, t[-1] + 1, 40),
                            np.nextafter(t, np.inf), np.nextafter(t, -np.inf)])
        lines = ["exz 0", "mexz 1", f"{hexf(0.5)} {nway}"] + [f"{hexf(a)} {hexf(b)} {hexf(c)}" for a, b, c in zip(t, x, y)]
        lines.append(f"points {len(q)}")
        lines += [f"0 0 {hexf(v)} 0 0" for v in q]
        rows = run_ref("\n".join(lines) + "\n")
        hdr = [v for tag, v in rows if tag == "interp_hdr"]
        ex = [v for tag, v in rows if tag == "interp_ex"]
        assert len(hdr) == len(q) == len(ex)
        cases.append(dict(t=t.tolist(), x=x.tolist(), y=y.tolist(), query=q.tolist(), header=hdr, example=ex))
    return cases


def traj_case(seed):
    rng = np.random.default_rng(seed)
    R, C = 6, 4
    tr = np.concatenate([np.linspace(0, 2.5, R)[:, None], rng.uniform(-3, 3, (R, C))], axis=1)
    idxs, scale, offset = [2, 0, 4], [2.0, -0.5, 3.0], [1.0, -2.0]          # fewer factors than columns on purpose
    lines = [f"traj {R} {C}"] + [" ".join(hexf(v) for v in row) for row in tr]
    lines += [f"idxs {len(idxs)} " + " ".join(map(str, idxs)), f"scale {len(scale)} " + " ".join(hexf(v) for v in scale),
              f"offset {len(offset)} " + " ".join(hexf(v) for v in offset)]
    rows = run_ref("\n".join(lines) + "\n")
    pick = lambda tag: [v for t, v in rows if t == tag]
    return dict(traj=tr.tolist(), idxs=idxs, scale=scale, offset=offset, extract=pick("extract"), scaled=pick("scale"),
                offsetted=pick("offset"))


def main():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, capture_output=True)
    lgl = json.load(open(os.path.join(HERE, "lgl.json")))
    sp = shipped_problem()
    tf = sp["nsteps"] * sp["dt"]
    M = sp["nsteps"] + 1
    node_t = [0.5 * tf * (tau + 1.0) for tau in lgl[str(M)]["tau"]]           # ePSOPT.cpp:44-45,151-154
    shipped = dymos_case("ocp_2d_ex1", sp["polys"], sp["tracks"], node_t, B=4, seed=0xE701)
    # a second table set: more waypoints, times outside the tables, other polygons
    tracks2 = [dict(radius=0.8, t=[-1.0, 2.0, 2.5, 9.0, 20.0], x=[1.0, 2.0, 2.2, 3.5, 1.5], y=[2.0, 2.5, 3.5, 3.0, 2.2]),
               dict(radius=0.3, t=[3.0, 11.0], x=[3.0, 1.2], y=[4.0, 2.1])]
    polys2 = [[(1.0, 2.0), (2.5, 1.9), (3.0, 3.1), (1.7, 3.9)], [(3.0, 3.5), (3.6, 2.2), (2.9, 1.9)]]
    node_t2 = [0.5 * 12.0 * (tau + 1.0) + 0.25 for tau in lgl["9"]["tau"]]
    synth = dymos_case("synthetic", polys2, tracks2, node_t2, B=8, seed=0xE702)
    json.dump(dict(source="oracle/_ref/ref_vectors <- reference src/Examples/Dymos/etol_dymos_example1.cpp",
                   cases=[shipped, synth]), open(os.path.join(HERE, "ref_dymos_ex1.json"), "w"))
    json.dump(dict(source="reference include/ETOL/TrajectoryOptimizer.hpp:239-258 and etol_dymos_example1.cpp:362-379",
                   cases=interp_cases(0xE703)), open(os.path.join(HERE, "ref_interp.json"), "w"))
    json.dump(dict(source="reference include/ETOL/TrajectoryOptimizer.hpp:268-324", **traj_case(0xE704)),
              open(os.path.join(HERE, "ref_traj.json"), "w"))
    print("wrote ref_dymos_ex1.json, ref_interp.json, ref_traj.json")


if __name__ == "__main__":
    sys.exit(main())
