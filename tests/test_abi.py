"""The C-ABI library loads and exports every symbol include/emi355x.h declares; host-only entry
points (mesh construction, keep-out constants) agree with the golden fixtures.  No GPU needed."""
import ctypes as C
import json
import os
import re

import numpy as np

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_every_declared_symbol_is_exported(built):
    import etol_amd._lib as L
    hdr = open(os.path.join(ROOT, "include", "emi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(emi_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    lib = L.load()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.emi_abi_version() == 2
    assert lib.emi_status_string(0) == b"ok"
    assert b"argument" in lib.emi_status_string(1)


def test_no_device_is_an_error_not_a_fallback(built):
    """On a box without a GPU, creating a context must fail loudly (there is no CPU path)."""
    import torch
    import etol_amd as E
    if torch.cuda.is_available():
        return
    try:
        E.Evaluator(0)
    except E.EmiError as e:
        assert "NO_DEVICE" in str(e)
    else:
        raise AssertionError("Evaluator() succeeded without a GPU")


def test_lgl_against_golden(built):
    import etol_amd as E
    g = json.load(open(os.path.join(GOLD, "lgl.json")))
    for M in (3, 4, 5, 9, 33):
        tau, w, D = E.lgl(M)
        ref = g[str(M)]
        assert np.abs(tau - ref["tau"]).max() < 2e-16
        assert np.abs(w - ref["w"]).max() < 1e-15
        Dr = np.array(ref["D"])
        # off-diagonal entries to rounding; the diagonal is the negative row sum (~1e-13 at M=33)
        off = ~np.eye(M, dtype=bool)
        assert (np.abs(D - Dr)[off] / np.abs(Dr)[off]).max() < 1e-14
        assert np.abs(np.diag(D) - np.diag(Dr)).max() < 1e-12 * M * M
    tau, w, D = E.lgl(256)
    ref = g["256"]
    assert np.abs(tau - ref["tau"]).max() < 2e-16 and np.abs(w - ref["w"]).max() < 1e-15
    for r, row in zip(ref["rows"], ref["D_rows"]):
        row = np.array(row)
        m = np.arange(256) != r
        assert (np.abs(D[r] - row)[m] / np.abs(row)[m]).max() < 1e-13
    c3 = g["closed"]["3"]
    tau, w, D = E.lgl(3)
    assert np.allclose(tau, c3["tau"], atol=0) and np.allclose(w, c3["w"], rtol=1e-16) and np.allclose(D, c3["D"], atol=1e-16)
    assert E.lgl(2)[2].tolist() == [[-0.5, 0.5], [-0.5, 0.5]]


def test_lgl_invariants(built):
    import etol_amd as E
    for M in (2, 3, 8, 33, 256, 1024):
        tau, w, D = E.lgl(M)
        N = M - 1
        assert abs(w.sum() - 2.0) < 1e-14
        assert np.all(np.diff(tau) > 0) and tau[0] == -1 and tau[-1] == 1
        assert np.array_equal(tau, -tau[::-1])                     # symmetric node set
        assert np.abs(D.sum(1)).max() < 1e-9 * max(1, N * N / 1e3)  # D annihilates constants
        assert np.abs(D + D[::-1, ::-1]).max() < 1e-9 * N * N      # centro-antisymmetry
        for p in range(1, min(N, 6) + 1):                          # exact on low-degree polynomials
            assert np.abs(D @ tau ** p - p * tau ** (p - 1)).max() < 2e-9 * max(1.0, N * N / 1e4)
        for p in range(0, min(2 * N - 1, 9) + 1):                  # Lobatto quadrature exact to degree 2N-1
            exact = 0.0 if p % 2 else 2.0 / (p + 1)
            assert abs(w @ tau ** p - exact) < 1e-14
    # the product mesh and the oracle's independent construction agree
    for M in (5, 64, 1024):
        a, b = E.lgl(M), O.lgl(M)
        assert np.abs(a[0] - b[0]).max() < 3e-16 and np.abs(a[1] - b[1]).max() < 1e-15
        assert np.abs(a[2] - b[2]).max() < 1e-9


def test_model_dims_and_errors(built):
    import etol_amd as E
    assert E.model_dims(E.MODEL_POINTMASS2D) == (2, 2, 0)
    assert E.model_dims(E.MODEL_QUADROTOR2D) == (6, 2, 5)
    assert E.model_dims(E.MODEL_FIXEDWING12) == (12, 4, 16)
    lib = E.load()
    assert lib.emi_model_dims(77, None, None, None) == 1
    assert lib.emi_lgl(1, None, None, None) == 1


def test_edge_ellipse_matches_oracle_and_reference_partials(built):
    import etol_amd as E
    import cases
    for poly in cases.OCP2D["exz"]:
        n = len(poly)
        for i in range(n):
            (xa, ya), (xb, yb) = poly[i], poly[(i + 1) % n]
            rec, ref = E.edge_ellipse(xa, ya, xb, yb), O.edge_ellipse(xa, ya, xb, yb)
            assert np.array_equal(rec, ref)
            # centre is the edge midpoint, a^2 the squared half length, b^2 = a^2/5
            assert abs(rec[1] - (xa + xb) / 2) < 1e-15 and abs(rec[2] - (ya + yb) / 2) < 1e-13  # slope form, as the reference
            assert abs(rec[5] - ((xb - xa) ** 2 + (yb - ya) ** 2) / 4) < 1e-13 and rec[6] == 0.2 * rec[5]
            assert abs(rec[3] ** 2 + rec[4] ** 2 - 1) < 1e-15
    # a vertical edge divides by zero exactly like the reference (etol_psopt_example1.cpp:169)
    assert np.isnan(E.edge_ellipse(1.0, 0.0, 1.0, 2.0)[2])


def test_track_centres(built):
    import etol_amd as E
    t, x, y = [0.0, 4.0, 10.0], [1.0, 3.0, 3.0], [0.0, -1.0, 2.0]
    q = np.array([-1.0, 0.0, 2.0, 4.0, 7.0, 10.0, 12.0])
    xc, yc = E.track_centres(t, x, y, q)
    oxc, oyc = O.track_centres(t, x, y, q)
    assert np.array_equal(xc, oxc) and np.array_equal(yc, oyc)
    # inside the table it is plain linear interpolation; outside, the end segments extrapolate
    assert np.allclose(xc[1:6], np.interp(q[1:6], t, x)) and np.allclose(yc[1:6], np.interp(q[1:6], t, y))
    assert np.isclose(xc[0], 0.5) and np.isclose(yc[0], 0.25) and np.isclose(yc[6], 3.0)


def test_every_tile_order_of_the_mfma_defect_kernel_is_a_permutation():
    """emi_debug_tile_order (no device needed): for meshes of 1 .. 32 column tiles, batch sizes around the 16-instance tile,
    2 / 6 / 12 states, every state split and every sym_cpart, the workgroup -> tile mapping the kernel uses
    (ring_tile_of in csrc/emi_args.hpp, one function for host and device) visits each tile exactly once.  (A 1280-node mesh
    once had partitions of 5 columns walked in blocks of 3: tiles left out, found on the GPU by tools/gpu_stress.py.)"""
    import ctypes as C
    from etol_amd import _lib as L
    lib = L.load()
    checked = partitioned = 0
    for M in (128, 256, 384, 512, 640, 768, 896, 1024, 1152, 1280, 1536, 1792, 2048, 2560, 4096):
        for B in (1, 15, 16, 17, 40, 128, 200, 256, 384, 512, 640, 768, 1000, 1024, 2048):
            for ns in (2, 6, 12):
                for ct in (0, 5, 6, 7, 8):
                    for cpart in (-1, 0, 1, 2, 4, 8):
                        tot, cp, cx = C.c_int(), C.c_int(), C.c_int()
                        cap = ((M // 128) * ((B + 15) // 16) * ns)
                        out = (C.c_int * cap)()
                        st = lib.emi_debug_tile_order(ns, B, M, ct, cpart, out, cap, C.byref(tot), C.byref(cp), C.byref(cx))
                        if st != 0:
                            continue
                        assert tot.value <= cap
                        seen = sorted(out[:tot.value])
                        assert seen == list(range(tot.value)), (M, B, ns, ct, cpart, cp.value, cx.value)
                        if cp.value > 0:
                            ncol = (M // 128) // cp.value
                            assert cx.value >= 1 and ncol % cx.value == 0
                            partitioned += 1
                        checked += 1
    assert checked > 5000 and partitioned > 500, (checked, partitioned)
    # the grouped order ("sym_gblk" instance groups per super-block, "sym_cx" column tiles per block)
    grouped = 0
    for M in (128, 256, 512, 768, 1024, 1280, 2048):
        for B in (128, 256, 384, 512, 640, 1024, 1536, 2048, 4096):
            for ns in (2, 6, 12):
                for ct in (5, 6, 7, 8):
                    for gblk in (1, 2, 3, 4, 8):
                        for cxo in (0, 1, 2, 3, 4, 8):
                            tot, cp, cx = C.c_int(), C.c_int(), C.c_int()
                            cap = ((M // 128) * ((B + 15) // 16) * ns)
                            out = (C.c_int * cap)()
                            st = lib.emi_debug_tile_order2(ns, B, M, ct, 0, gblk, cxo, out, cap, C.byref(tot), C.byref(cp), C.byref(cx))
                            if st != 0:
                                continue
                            assert sorted(out[:tot.value]) == list(range(tot.value)), (M, B, ns, ct, gblk, cxo, cp.value, cx.value)
                            if cp.value < 0:
                                assert cp.value == -gblk and (M // 128) % cx.value == 0
                                # an XCD's tiles stay inside its own contiguous range of instance groups
                                nt, per = M // 128, tot.value // 8
                                nsg = tot.value // nt // ((B + 15) // 16)
                                for x in range(8):
                                    mts = {out[t] // nt // nsg for t in range(x * per, (x + 1) * per)}
                                    assert mts == set(range(x * len(mts), (x + 1) * len(mts))), (M, B, ns, ct, gblk, x)
                                grouped += 1
    assert grouped > 500, grouped


def test_every_role_map_of_the_one_launch_pass_is_a_bijection():
    """emi_debug_pass_roles (no device needed): for many (MFMA blocks, node blocks) per XCD and every "pass_order" -- interleaved,
    MFMA first, front-loaded densities up to far beyond the clamp -- each MFMA block index and each node block index is dealt
    exactly once (pass_role_of in csrc/emi_args.hpp, one function for host and device).  A first version of the front-loaded
    order was not a bijection above one MFMA block per block and sent node workgroups out of range: a GPU memory fault."""
    import ctypes as C
    from etol_amd import _lib as L
    lib = L.load()
    n = 0
    for nm in (0, 1, 6, 8, 48, 96, 192, 384, 768):
        for nn in (1, 2, 16, 32, 64, 256, 512):
            for order in (0, 1, 100, 105, 110, 125, 150, 200, 400, 1000, 100000):
                out = (C.c_int * (nm + nn))()
                assert lib.emi_debug_pass_roles(nm, nn, order, out, nm + nn) == 0
                roles = list(out)
                assert sorted(r for r in roles if r >= 0) == list(range(nm)), (nm, nn, order)
                assert sorted(-1 - r for r in roles if r < 0) == list(range(nn)), (nm, nn, order)
                if order == 1:
                    assert all(r >= 0 for r in roles[:nm])
                n += 1
    assert n > 500
