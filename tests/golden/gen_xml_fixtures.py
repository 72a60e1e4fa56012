#!/usr/bin/env python3
"""Writes the problem-definition XML fixtures (into a directory the tests choose: tests/conftest.py
generates them per session; nothing is committed as .xml).

ocp_2d_ex1.xml / mip_2d_ex1.xml carry the DATA of the two configurations the reference ships under
resource/configs (numbers as listed in SURVEY.md section 6 and tests/cases.py); the edge_* files
exercise the loader's child caps (nstates/ncontrols/nzones/ncorners/nwaypoints/ndatums), a
<mexzones> without a count, and unknown elements."""


def xml(nsteps, dt, states, xr, controls, ur, zones, tracks, nstates=None, ncontrols=None, nzones=None,
        ncorners=None, nmex="auto", nway=None, ndat=None, extra=""):
    o = ['<?xml version="1.0" encoding="UTF-8"?>', f'<etol nsteps="{nsteps}" dt="{dt:.2f}">']
    o.append(f'\t<states nstates="{nstates if nstates is not None else len(states)}" rhorizon="{xr}">')
    for i, s in enumerate(states):
        o.append('\t\t<state name="x%d" vartype="%s" lower="%.2f" upper="%.2f" initial="%.2f" terminal="%.2f" '
                 'tolerance="%.2f"/>' % ((i,) + s))
    o.append('\t</states>')
    o.append(f'\t<controls ncontrols="{ncontrols if ncontrols is not None else len(controls)}" rhorizon="{ur}">')
    for i, c in enumerate(controls):
        o.append('\t\t<control name="u%d" vartype="%s" lower="%.2f" upper="%.2f"/>' % ((i,) + c))
    o.append('\t</controls>')
    o.append(f'\t<exzones nzones="{nzones if nzones is not None else len(zones)}">')
    for i, z in enumerate(zones):
        o.append(f'\t\t<border name="exz{i}" ncorners="{ncorners if ncorners is not None else len(z)}">')
        for (x, y) in z:
            o.append(f'\t\t\t<corner x="{x:.2f}" y="{y:.2f}" z="0.00"/>')
        o.append('\t\t</border>')
    o.append('\t</exzones>')
    if nmex == "auto":
        o.append(f'\t<mexzones nzones="{len(tracks)}">')
    elif nmex is None:
        o.append('\t<mexzones>')
    else:
        o.append(f'\t<mexzones nzones="{nmex}">')
    for i, t in enumerate(tracks):
        o.append(f'\t\t<track name="mexz{i}" radius="{t["radius"]:.2f}" '
                 f'nwaypoints="{nway if nway is not None else len(t["wp"])}">')
        for j, (tt, vals) in enumerate(t["wp"]):
            o.append(f'\t\t\t<waypoint name="pt{j}" t="{tt:.2f}" ndatums="{ndat if ndat is not None else len(vals)}">')
            for v in vals:
                o.append(f'\t\t\t\t<datum>{v:.2f}</datum>')
            o.append('\t\t\t</waypoint>')
        o.append('\t\t</track>')
    o.append('\t</mexzones>')
    if extra:
        o.append(extra)
    o.append('</etol>')
    return "\n".join(o) + "\n"


S = [("C", 0, 7, 1, 5, 0.01), ("C", 0, 7, 2, 4, 0.01)]
C2 = [("C", -0.5, 0.5)] * 2
C4 = [("C", -0.5, 0.5)] * 4
Z = [[(3.2, 2.5), (3.4, 2.6), (3.5, 3.4), (3.3, 3.0), (3.1, 3.5)], [(2.2, 2.5), (2.4, 2.6), (2.5, 3.4), (2.1, 3.5)]]
T_OCP = [dict(radius=0.5, wp=[(0, [1.51, 2.0]), (32, [2.0, 2.0])]),
         dict(radius=0.5, wp=[(0, [1.0, 4.0]), (32, [1.0, 3.0])])]
T_MIP = [dict(radius=0.5, wp=[(0, [2.0, 2.0]), (32, [2.5, 2.0])]),
         dict(radius=0.5, wp=[(0, [1.0, 4.0]), (32, [1.0, 3.0])])]

def write_all(dirpath):
    """-> dict name -> path of the XML files written into dirpath"""
    import os
    S3 = S + [("I", -1, 1, 0, 0, 0.5)]
    files = {
        "ocp_2d_ex1.xml": xml(32, 0.5, S, 0, C2, 0, Z, T_OCP),
        "mip_2d_ex1.xml": xml(16, 0.5, S, 1, C4, 0, Z, T_MIP),
        "edge_caps.xml": xml(8, 0.25, S3, 2, C4, 3, Z, T_OCP, nstates=2, ncontrols=3, nzones=1, ncorners=2, nmex=1,
                             nway=1, ndat=1, extra='\t<unknown foo="1"><state vartype="C"/></unknown>'),
        "edge_no_mex_count.xml": xml(4, 1.0, S, 0, C2, 0, [], T_OCP, nmex=None),
    }
    out = {}
    for name, text in files.items():
        path = os.path.join(dirpath, name)
        with open(path, "w") as f:
            f.write(text)
        out[name] = path
    return out


if __name__ == "__main__":
    import sys
    print(write_all(sys.argv[1] if len(sys.argv) > 1 else "."))
