"""Traced callbacks (mi355x::Var): trace + symbolic derivatives + generated model code, on the CPU.

The generated struct is compiled with g++ behind a three-line host prelude and evaluated against
(a) the sympy golden values of the hand-written quadrotor (tests/golden/models.json) and
(b) sympy derivatives of a model that uses every traced operation.
ePSOPT gets these derivatives from ADOL-C (reference src/ePSOPT/ePSOPT.cpp:64-65)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
MODELS = json.load(open(os.path.join(HERE, "golden", "models.json")))

PRELUDE = r"""
#include <cmath>
#define EMI_DEV inline
#define EMI_MAX_PARAMS 16
template <typename T> struct ModelParams { T p[EMI_MAX_PARAMS]; };
inline double emi_sin(double a) { return std::sin(a); }
inline double emi_cos(double a) { return std::cos(a); }
inline double emi_tan(double a) { return std::tan(a); }
inline double emi_exp(double a) { return std::exp(a); }
inline double emi_log(double a) { return std::log(a); }
inline double emi_sqrt(double a) { return std::sqrt(a); }
inline double emi_pow(double a, double c) { return std::pow(a, c); }
"""
POSTLUDE = r"""
typedef TracedModel<double> TM;
extern "C" void traced_eval(const double* z, double t, double cL, const double* cf,
                            double* f, double* J, double* L, double* g, double* H) {
    ModelParams<double> P = {};
    TM::f(P, z, t, f);
    double Jm[TM::NS][TM::NV];
    TM::jac(P, z, t, Jm);
    for (int i = 0; i < TM::NS; ++i) for (int v = 0; v < TM::NV; ++v) J[i * TM::NV + v] = Jm[i][v];
    *L = TM::cost(P, z, t);
    TM::grad(P, z, t, g);
    for (int k = 0; k < TM::NV * (TM::NV + 1) / 2; ++k) H[k] = 0;
    TM::hess(P, z, t, cL, cf, H);
}
"""


def _compile_traced(built, tmp_path, which):
    import torch  # noqa: F401  (one libamdhip64 for the harness's dependency on libemi355x)
    lib = C.CDLL(os.path.join(HERE, "harness", "libetol_harness.so"))
    lib.harness_traced_model_source.restype = C.c_char_p
    src = lib.harness_traced_model_source(which).decode()
    cpp = tmp_path / f"traced{which}.cpp"
    cpp.write_text(PRELUDE + src + POSTLUDE)
    so = tmp_path / f"traced{which}.so"
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-o", str(so), str(cpp)])
    t = C.CDLL(str(so))
    dp = C.POINTER(C.c_double)
    t.traced_eval.argtypes = [dp, C.c_double, C.c_double, dp, dp, dp, dp, dp, dp]
    return t, src


def _eval(t, ns, nc, z, tk, cL, cf):
    nv = ns + nc
    z = np.ascontiguousarray(z, dtype=np.float64)
    cf = np.ascontiguousarray(cf, dtype=np.float64)
    f = np.zeros(ns); J = np.zeros((ns, nv)); L = np.zeros(1); g = np.zeros(nv); H = np.zeros(nv * (nv + 1) // 2)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    t.traced_eval(p(z), tk, cL, p(cf), p(f), p(J), p(L), p(g), p(H))
    return f, J, L[0], g, H


def test_traced_quadrotor_matches_sympy_golden(built, tmp_path):
    t, src = _compile_traced(built, tmp_path, 0)
    assert "struct TracedModel" in src and "NS = 6, NC = 2" in src
    for pt in MODELS["1"]["points"]:
        f, J, L, g, H = _eval(t, 6, 2, pt["z"], 0.0, pt["cL"], pt["cf"])
        tol = lambda ref: 1e-13 * (np.abs(ref).max() + 1)
        assert np.abs(f - pt["f"]).max() < tol(np.array(pt["f"]))
        assert np.abs(J - np.array(pt["J"])).max() < tol(np.array(pt["J"]))
        assert abs(L - pt["L"]) < tol(np.array(pt["L"]))
        assert np.abs(g - np.array(pt["gL"])).max() < tol(np.array(pt["gL"]))
        assert np.abs(H - np.array(pt["H"])).max() < tol(np.array(pt["H"]))


def test_traced_every_operation_against_sympy(built, tmp_path):
    sp = pytest.importorskip("sympy")
    t, _ = _compile_traced(built, tmp_path, 1)
    a, b, c, tk = sp.symbols("a b c t")
    f0 = sp.exp(a * sp.Rational(3, 10)) / (1 + b * b) + sp.tan(c) * sp.sqrt(2 + a * a) - tk
    f1 = sp.log(3 + a * b + c * c) * (2 + b) ** sp.Rational(3, 2) - sp.cos(a - c) / sp.sin(1 + b / 10)
    Lc = c * c * sp.exp(-a) + b * sp.sin(a * c)
    zs = [a, b, c]
    rng = np.random.default_rng(11)
    for _ in range(5):
        z = rng.uniform(0.2, 1.2, 3)
        tv = float(rng.uniform(0, 2)); cL = float(rng.uniform(0.5, 2)); cf = rng.uniform(-1, 1, 2)
        sub = {a: z[0], b: z[1], c: z[2], tk: tv}
        ev = lambda e: float(e.evalf(30, subs=sub))
        f, J, L, g, H = _eval(t, 2, 1, z, tv, cL, cf)
        ref_f = np.array([ev(f0), ev(f1)])
        ref_J = np.array([[ev(sp.diff(fi, v)) for v in zs] for fi in (f0, f1)])
        ref_g = np.array([ev(sp.diff(Lc, v)) for v in zs])
        lag = cL * Lc + cf[0] * f0 + cf[1] * f1
        ref_H = np.array([ev(sp.diff(lag, zs[v], zs[q])) for v in range(3) for q in range(v + 1)])
        for got, ref in ((f, ref_f), (J, ref_J), (g, ref_g), (H, ref_H)):
            assert np.abs(got - ref).max() < 1e-12 * (np.abs(ref).max() + 1)
        assert abs(L - ev(Lc)) < 1e-13 * (abs(ev(Lc)) + 1)


def _check_source(src, ns, nc, f32=0, name="TracedModel", npath=0, pw=2):
    import torch  # noqa: F401
    from etol_amd import _lib
    lib = _lib.load()
    log = C.create_string_buffer(1 << 16)
    st = lib.emi_check_model_source(name.encode(), src.encode(), ns, nc, npath, pw, f32, log, len(log))
    return st, log.value.decode(errors="replace")


@pytest.mark.parametrize("which,ns,nc,npath,pw", [(0, 6, 2, 0, 0), (1, 2, 1, 0, 0), (2, 6, 2, 2, 2), (3, 6, 2, 4, 6)])
def test_generated_model_compiles_against_the_kernel_templates(built, which, ns, nc, npath, pw):
    """hiprtc cross-compiles for gfx950 without a GPU: the generated struct must instantiate the node,
    Hessian and even/odd MFMA defect kernels the library itself is built from."""
    import torch  # noqa: F401
    lib = C.CDLL(os.path.join(HERE, "harness", "libetol_harness.so"))
    lib.harness_traced_model_source.restype = C.c_char_p
    src = lib.harness_traced_model_source(which).decode()
    for f32 in (0, 1):
        st, log = _check_source(src, ns, nc, f32, npath=npath, pw=pw)
        assert st == 0, log
    if npath:       # the number of traced rows and of the variables they depend on are part of the contract
        st, log = _check_source(src, ns, nc, 0, npath=npath + 1, pw=pw)
        assert st == 1 and "dimensions differ" in log
        st, log = _check_source(src, ns, nc, 0, npath=npath, pw=pw + 1)
        assert st == 1 and "another number of variables" in log


def test_model_source_errors_are_reported(built):
    st, log = _check_source("template <typename T> struct TracedModel { static constexpr int NS = 2; };", 2, 1)
    assert st == 1 and "error" in log
    # dimension mismatch between the text and the call
    lib = C.CDLL(os.path.join(HERE, "harness", "libetol_harness.so"))
    lib.harness_traced_model_source.restype = C.c_char_p
    src = lib.harness_traced_model_source(1).decode()
    st, log = _check_source(src, 3, 1)
    assert st == 1 and "dimensions differ" in log
    st, log = _check_source(src, 2, 1, name="not a name")
    assert st == 1


PATH_POSTLUDE = r"""
typedef TracedModel<double> TM;
extern "C" int traced_pw(void) { return TM::PW; }
extern "C" int traced_pvar(int q) { return TM::pvar(q); }
// cd[NPATH][PW], H[NV (NV + 1) / 2] (packed lower triangle, accumulated into zeros)
extern "C" void traced_path_full(const double* z, double t, const double* mu, double* c, double* cd, double* H) {
    ModelParams<double> P = {};
    TM::path(P, z, t, c, cd);
    for (int q = 0; q < TM::NV * (TM::NV + 1) / 2; ++q) H[q] = 0;
    TM::path_hess(P, z, t, mu, H);
}
// rows on the first two variables: cx, cy and the three second derivatives (xx, xy, yy)
extern "C" void traced_path(const double* z, double t, const double* mu, double* c, double* cx, double* cy, double* h) {
    double cd[TM::NPATH * TM::PW], H[TM::NV * (TM::NV + 1) / 2];
    traced_path_full(z, t, mu, c, cd, H);
    for (int j = 0; j < TM::NPATH; ++j) { cx[j] = cd[j * TM::PW]; cy[j] = cd[j * TM::PW + 1]; }
    h[0] = H[0]; h[1] = H[1]; h[2] = H[2];
}
"""


def test_traced_path_rows_against_the_reference_formulas(built, tmp_path):
    """Constraint rows written with Var arithmetic, the formulas of the reference's obstacle callbacks
    (src/Examples/PSOPT/etol_psopt_example1.cpp:163-182 ellipse per edge, :243-247 disc), restated here in
    numpy: values, the two partials per row and the multiplier-weighted second derivatives of the generated code."""
    import torch  # noqa: F401
    lib = C.CDLL(os.path.join(HERE, "harness", "libetol_harness.so"))
    lib.harness_traced_model_source.restype = C.c_char_p
    src = lib.harness_traced_model_source(2).decode()
    assert "NPATH = 2" in src and not src.startswith("ERROR")
    cpp = tmp_path / "path.cpp"
    cpp.write_text(PRELUDE + src + PATH_POSTLUDE)
    so = tmp_path / "path.so"
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-o", str(so), str(cpp)])
    t = C.CDLL(str(so))
    dp = C.POINTER(C.c_double)
    t.traced_path.argtypes = [dp, C.c_double, dp, dp, dp, dp, dp]
    p = lambda a: a.ctypes.data_as(dp)

    def ref(x, y):
        disc = 0.8 ** 2 - ((x - 4.0) ** 2 + (y - 3.2) ** 2)
        xa, ya, xb, yb = 3.2, 2.5, 3.4, 2.6
        xc = (xb + xa) / 2
        m = (yb - ya) / (xb - xa)
        yc = ya + m * (xc - xa)
        radsq = (xc - xa) ** 2 + (yc - ya) ** 2
        tt = -np.arctan2(yc - ya, xc - xa)
        dx, dy = x - xc, y - yc
        delx, dely = np.cos(tt) * dx - np.sin(tt) * dy, np.sin(tt) * dx + np.cos(tt) * dy
        asq, bsq = radsq, .2 * radsq
        return np.array([disc, asq * bsq - (bsq * delx ** 2 + asq * dely ** 2)])

    rng = np.random.default_rng(2)
    for _ in range(5):
        z = rng.uniform(0, 6, 8)
        mu = rng.standard_normal(2)
        c, cx, cy, h = np.zeros(2), np.zeros(2), np.zeros(2), np.zeros(3)
        t.traced_path(p(z), 0.3, p(mu), p(c), p(cx), p(cy), p(h))
        assert np.abs(c - ref(z[0], z[1])).max() < 1e-14 * (np.abs(c).max() + 1)
        e = 1e-6
        fx = (ref(z[0] + e, z[1]) - ref(z[0] - e, z[1])) / (2 * e)
        fy = (ref(z[0], z[1] + e) - ref(z[0], z[1] - e)) / (2 * e)
        assert np.abs(cx - fx).max() < 1e-8 and np.abs(cy - fy).max() < 1e-8
        # both rows are quadratics: second derivatives are constants, the weighted sum is exact by differences of cx, cy
        c2, cx2, cy2, h2 = np.zeros(2), np.zeros(2), np.zeros(2), np.zeros(3)
        zp = z.copy(); zp[0] += 1.0
        t.traced_path(p(zp), 0.3, p(mu), p(c2), p(cx2), p(cy2), p(h2))
        zq = z.copy(); zq[1] += 1.0
        c3, cx3, cy3, h3 = np.zeros(2), np.zeros(2), np.zeros(2), np.zeros(3)
        t.traced_path(p(zq), 0.3, p(mu), p(c3), p(cx3), p(cy3), p(h3))
        assert abs(h[0] - mu @ (cx2 - cx)) < 1e-12 and abs(h[1] - mu @ (cy2 - cy)) < 1e-12 and abs(h[2] - mu @ (cy3 - cy)) < 1e-12


def test_traced_rows_on_more_than_two_variables(built, tmp_path):
    """Traced rows may depend on any states and controls of their node (and on time): a disc keep-out on (x, z), a
    speed limit on (vx, vz), a thrust-tilt coupling on (theta, thrust), a row of time and one state.  The generated
    struct lists the union of the variables (PW = 6: 0 1 2 3 4 6); values, every partial and the multiplier-weighted
    second derivatives against numpy / finite differences."""
    import torch  # noqa: F401
    lib = C.CDLL(os.path.join(HERE, "harness", "libetol_harness.so"))
    lib.harness_traced_model_source.restype = C.c_char_p
    src = lib.harness_traced_model_source(3).decode()
    assert "NPATH = 4, PW = 6" in src and not src.startswith("ERROR"), src[:300]
    cpp = tmp_path / "path3.cpp"
    cpp.write_text(PRELUDE + src + PATH_POSTLUDE)
    so = tmp_path / "path3.so"
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-o", str(so), str(cpp)])
    t = C.CDLL(str(so))
    dp = C.POINTER(C.c_double)
    t.traced_path_full.argtypes = [dp, C.c_double, dp, dp, dp, dp]
    p = lambda a: a.ctypes.data_as(dp)
    pv = [t.traced_pvar(q) for q in range(t.traced_pw())]
    assert pv == [0, 1, 2, 3, 4, 6]

    def rows(z, tk):
        return np.array([0.64 - ((z[0] - 4.0) ** 2 + (z[1] - 3.2) ** 2), z[3] ** 2 + z[4] ** 2 - 9.0,
                         z[6] * np.sin(z[2]) - 6.0, z[1] * np.cos(0.3 * tk) - 9.5])

    rng = np.random.default_rng(4)
    for _ in range(4):
        z, mu, tk = rng.uniform(-2, 6, 8), rng.standard_normal(4), 1.7
        c, cd, H = np.zeros(4), np.zeros(4 * 6), np.zeros(36)
        t.traced_path_full(p(z), tk, p(mu), p(c), p(cd), p(H))
        assert np.abs(c - rows(z, tk)).max() < 1e-14 * (np.abs(c).max() + 1)
        cd = cd.reshape(4, 6)
        e = 1e-6
        for q, v in enumerate(pv):
            zp, zm = z.copy(), z.copy()
            zp[v] += e; zm[v] -= e
            assert np.abs(cd[:, q] - (rows(zp, tk) - rows(zm, tk)) / (2 * e)).max() < 1e-8
        # weighted second derivatives, packed lower triangle of the 8 x 8 node block
        Href = np.zeros((8, 8))
        Href[0, 0] = Href[1, 1] = -2 * mu[0]
        Href[3, 3] = Href[4, 4] = 2 * mu[1]
        Href[2, 2] = -mu[2] * z[6] * np.sin(z[2])
        Href[6, 2] = Href[2, 6] = mu[2] * np.cos(z[2])
        for a in range(8):
            for b in range(a + 1):
                assert abs(H[a * (a + 1) // 2 + b] - Href[a, b]) < 1e-12, (a, b)


def test_traced_interp1_rows_against_numpy(built, tmp_path):
    """mx::interp1 (piecewise linear through max / min) inside traced rows: values and partials of the generated
    code against numpy.interp, at times inside, at the knots of and outside the waypoint tables."""
    import torch  # noqa: F401
    lib = C.CDLL(os.path.join(HERE, "harness", "libetol_harness.so"))
    lib.harness_traced_model_source.restype = C.c_char_p
    src = lib.harness_traced_model_source(4).decode()
    assert "NPATH = 3" in src
    cpp = tmp_path / "interp.cpp"
    cpp.write_text(PRELUDE + src + PATH_POSTLUDE)
    so = tmp_path / "interp.so"
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-o", str(so), str(cpp)])
    t = C.CDLL(str(so))
    dp = C.POINTER(C.c_double)
    t.traced_path.argtypes = [dp, C.c_double, dp, dp, dp, dp, dp]
    p = lambda a: a.ctypes.data_as(dp)
    tracks = [(0.5, [0.0, 32.0], [1.51, 2.00], [2.00, 2.00]), (0.5, [0.0, 32.0], [1.00, 1.00], [4.00, 3.00]),
              (0.4, [0.0, 4.0, 9.0, 16.0], [3.0, 3.5, 2.5, 4.0], [1.0, 2.5, 3.0, 2.0])]
    rng = np.random.default_rng(4)
    for tk in (-1.0, 0.0, 1.7, 4.0, 6.5, 9.0, 12.25, 16.0, 20.0, 40.0):
        z = rng.uniform(0, 5, 4)
        mu = rng.standard_normal(3)
        c, cx, cy, h = np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(3)
        t.traced_path(p(z), tk, p(mu), p(c), p(cx), p(cy), p(h))
        for j, (r, tw, xw, yw) in enumerate(tracks):
            xc, yc = np.interp(tk, tw, xw), np.interp(tk, tw, yw)      # constant outside the table, like interp1
            assert abs(c[j] - (r * r - ((z[0] - xc) ** 2 + (z[1] - yc) ** 2))) < 1e-13
            assert abs(cx[j] + 2 * (z[0] - xc)) < 1e-13 and abs(cy[j] + 2 * (z[1] - yc)) < 1e-13
        assert abs(h[0] + 2 * mu.sum()) < 1e-13 and abs(h[1]) < 1e-13 and abs(h[2] + 2 * mu.sum()) < 1e-13


def _fixedwing_struct_text():
    src = open(os.path.join(os.path.dirname(HERE), "etol_amd", "csrc", "emi_models.hpp")).read()
    a = src.index("template <typename T> struct FixedWing12 {")
    b = src.index("};", src.index("// ---- generated body end ----")) + 2
    return src[a:b]


def test_fixedwing_hessian_is_the_generated_one_and_matches_sympy(built, tmp_path):
    """FixedWing12::hess in emi_models.hpp is machine-generated (the model written once more on the expression
    trace, differentiated twice).  The checked-in text must be what the generator emits today, and the struct,
    compiled for the host, must reproduce the sympy Hessian of tests/golden/models.json."""
    import torch  # noqa: F401
    lib = C.CDLL(os.path.join(HERE, "harness", "libetol_harness.so"))
    lib.harness_fixedwing_hess_body.restype = C.c_char_p
    body = lib.harness_fixedwing_hess_body().decode()
    text = _fixedwing_struct_text()
    checked_in = text[text.index("// ---- generated body begin ----\n") + len("// ---- generated body begin ----\n"):
                      text.index("        // ---- generated body end ----")]
    assert checked_in == body, "regenerate FixedWing12::hess (harness_fixedwing_hess_body) and paste it into emi_models.hpp"
    prelude = PRELUDE + "inline void emi_sincos(double a, double* s, double* c) { *s = std::sin(a); *c = std::cos(a); }\n"
    post = r"""
typedef FixedWing12<double> FW;
extern "C" void fw_eval(const double* p, const double* z, double cL, const double* cf, double* f, double* J, double* g, double* H) {
    ModelParams<double> P = {};
    for (int i = 0; i < 16; ++i) P.p[i] = p[i];
    FW::f(P, z, 0.0, f);
    double Jm[FW::NS][FW::NV];
    FW::jac(P, z, 0.0, Jm);
    for (int i = 0; i < FW::NS; ++i) for (int v = 0; v < FW::NV; ++v) J[i * FW::NV + v] = Jm[i][v];
    FW::grad(P, z, 0.0, g);
    for (int k = 0; k < 136; ++k) H[k] = 0;
    FW::hess(P, z, 0.0, cL, cf, H);
}
"""
    cpp = tmp_path / "fw.cpp"
    cpp.write_text(prelude + text + post)
    so = tmp_path / "fw.so"
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-o", str(so), str(cpp)])
    t = C.CDLL(str(so))
    dp = C.POINTER(C.c_double)
    t.fw_eval.argtypes = [dp, dp, C.c_double, dp, dp, dp, dp, dp]
    p = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)
    params = np.array(MODELS["2"]["params"], dtype=np.float64)
    for pt in MODELS["2"]["points"]:
        z, cf = np.array(pt["z"]), np.array(pt["cf"])
        f, J, g, H = np.zeros(12), np.zeros(12 * 16), np.zeros(16), np.zeros(136)
        t.fw_eval(p(params), p(z), pt["cL"], p(cf), p(f), p(J), p(g), p(H))
        assert np.abs(f - pt["f"]).max() < 1e-12 * (np.abs(pt["f"]).max() + 1)
        assert np.abs(J.reshape(12, 16) - np.array(pt["J"])).max() < 1e-12 * (np.abs(pt["J"]).max() + 1)
        assert np.abs(H - np.array(pt["H"])).max() < 1e-11 * (np.abs(pt["H"]).max() + 1)
