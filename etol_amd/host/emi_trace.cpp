// emi_trace.cpp -- see emi_trace.hpp.
#include "emi_trace.hpp"

#include <algorithm>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <sstream>

namespace ETOL {
namespace mi355x {

Trace& Trace::active() {
    static thread_local Trace t;     // one trace per thread: solvers may be set up concurrently
    return t;
}

void Trace::clear() {
    _nodes.clear();
    _cse.clear();
}

int Trace::intern(const Node& n) {
    const auto key = std::make_tuple((int)n.op, n.a, n.b, n.c);
    auto it = _cse.find(key);
    if (it != _cse.end()) return it->second;
    _nodes.push_back(n);
    const int id = (int)_nodes.size() - 1;
    _cse[key] = id;
    return id;
}

bool Trace::is_const(int n, double* v) const {
    if (n < 0 || _nodes[n].op != CONST) return false;
    if (v) *v = _nodes[n].c;
    return true;
}

int Trace::constant(double c) { return intern(Node{CONST, -1, -1, c}); }
int Trace::input(Op kind, int index) { return intern(Node{kind, index, -1, 0.0}); }

int Trace::unary(Op op, int a, double c) {
    double va;
    if (is_const(a, &va)) {
        switch (op) {
            case NEG: return constant(-va);
            case SIN: return constant(std::sin(va));
            case COS: return constant(std::cos(va));
            case TAN: return constant(std::tan(va));
            case EXP: return constant(std::exp(va));
            case LOG: return constant(std::log(va));
            case SQRT: return constant(std::sqrt(va));
            case POWC: return constant(std::pow(va, c));
            case ABS: return constant(std::fabs(va));
            case STEP: return constant(va > 0.0 ? 1.0 : 0.0);
            default: break;
        }
    }
    if (op == NEG && _nodes[a].op == NEG) return _nodes[a].a;
    if (op == POWC) {
        if (c == 0.0) return constant(1.0);
        if (c == 1.0) return a;
        if (c == 2.0) return binary(MUL, a, a);
    }
    return intern(Node{op, a, -1, c});
}

int Trace::binary(Op op, int a, int b) {
    double va = 0, vb = 0;
    const bool ca = is_const(a, &va), cb = is_const(b, &vb);
    if (ca && cb) {
        switch (op) {
            case ADD: return constant(va + vb);
            case SUB: return constant(va - vb);
            case MUL: return constant(va * vb);
            case DIV: return constant(va / vb);
            case MAX: return constant(va > vb ? va : vb);
            case MIN: return constant(va < vb ? va : vb);
            default: break;
        }
    }
    switch (op) {
        case MAX:
        case MIN:
            if (a == b) return a;
            if (a > b) std::swap(a, b);
            break;
        case ADD:
            if (ca && va == 0.0) return b;
            if (cb && vb == 0.0) return a;
            if (a > b) std::swap(a, b);       // commutative: one canonical form for the CSE table
            break;
        case SUB:
            if (cb && vb == 0.0) return a;
            if (ca && va == 0.0) return unary(NEG, b);
            if (a == b) return constant(0.0);
            break;
        case MUL:
            if ((ca && va == 0.0) || (cb && vb == 0.0)) return constant(0.0);
            if (ca && va == 1.0) return b;
            if (cb && vb == 1.0) return a;
            if (ca && va == -1.0) return unary(NEG, b);
            if (cb && vb == -1.0) return unary(NEG, a);
            if (a > b) std::swap(a, b);
            break;
        case DIV:
            if (ca && va == 0.0) return constant(0.0);
            if (cb && vb == 1.0) return a;
            if (cb) return binary(MUL, a, constant(1.0 / vb));
            break;
        default: break;
    }
    return intern(Node{op, a, b, 0.0});
}

// Reverse sweep over the nodes below `out` (operands always have smaller ids than their users).
std::vector<int> Trace::adjoints(int out) {
    std::vector<int> adj(out + 1, -1);
    adj[out] = constant(1.0);
    auto add = [&](int n, int contribution) {
        adj[n] = adj[n] < 0 ? contribution : binary(ADD, adj[n], contribution);
    };
    for (int n = out; n >= 0; --n) {
        if (adj[n] < 0) continue;
        const Node nd = _nodes[n];          // copy: _nodes grows below
        const int g = adj[n];
        switch (nd.op) {
            case ADD: add(nd.a, g); add(nd.b, g); break;
            case SUB: add(nd.a, g); add(nd.b, unary(NEG, g)); break;
            case MUL: add(nd.a, binary(MUL, g, nd.b)); add(nd.b, binary(MUL, g, nd.a)); break;
            case DIV:
                add(nd.a, binary(DIV, g, nd.b));
                add(nd.b, unary(NEG, binary(DIV, binary(MUL, g, n), nd.b)));
                break;
            case NEG: add(nd.a, unary(NEG, g)); break;
            case SIN: add(nd.a, binary(MUL, g, unary(COS, nd.a))); break;
            case COS: add(nd.a, unary(NEG, binary(MUL, g, unary(SIN, nd.a)))); break;
            case TAN: add(nd.a, binary(MUL, g, binary(ADD, constant(1.0), binary(MUL, n, n)))); break;
            case EXP: add(nd.a, binary(MUL, g, n)); break;
            case LOG: add(nd.a, binary(DIV, g, nd.a)); break;
            case SQRT: add(nd.a, binary(DIV, g, binary(MUL, constant(2.0), n))); break;
            case POWC: add(nd.a, binary(MUL, g, binary(MUL, constant(nd.c), unary(POWC, nd.a, nd.c - 1.0)))); break;
            // piecewise-linear operations: the derivative of the active branch (a tie takes the second operand)
            case MAX: {
                const int sa = unary(STEP, binary(SUB, nd.a, nd.b));
                add(nd.a, binary(MUL, g, sa));
                add(nd.b, binary(MUL, g, binary(SUB, constant(1.0), sa)));
                break;
            }
            case MIN: {
                const int sb = unary(STEP, binary(SUB, nd.b, nd.a));
                add(nd.a, binary(MUL, g, sb));
                add(nd.b, binary(MUL, g, binary(SUB, constant(1.0), sb)));
                break;
            }
            case ABS: add(nd.a, binary(MUL, g, binary(SUB, binary(MUL, constant(2.0), unary(STEP, nd.a)), constant(1.0)))); break;
            case STEP: break;   // piecewise constant
            default: break;   // CONST and inputs: leaves
        }
    }
    return adj;
}

double Trace::eval(int node, const std::vector<double>& x, const std::vector<double>& u, double t,
                   const std::vector<double>& coef) const {
    std::vector<double> v(node + 1, 0.0);
    for (int n = 0; n <= node; ++n) {
        const Node& nd = _nodes[n];
        switch (nd.op) {
            case CONST: v[n] = nd.c; break;
            case IN_STATE: v[n] = x.at(nd.a); break;
            case IN_CONTROL: v[n] = u.at(nd.a); break;
            case IN_TIME: v[n] = t; break;
            case IN_COEF: v[n] = coef.at(nd.a); break;
            case IN_PARAM: v[n] = param_values.at(nd.a); break;
            case ADD: v[n] = v[nd.a] + v[nd.b]; break;
            case SUB: v[n] = v[nd.a] - v[nd.b]; break;
            case MUL: v[n] = v[nd.a] * v[nd.b]; break;
            case DIV: v[n] = v[nd.a] / v[nd.b]; break;
            case NEG: v[n] = -v[nd.a]; break;
            case SIN: v[n] = std::sin(v[nd.a]); break;
            case COS: v[n] = std::cos(v[nd.a]); break;
            case TAN: v[n] = std::tan(v[nd.a]); break;
            case EXP: v[n] = std::exp(v[nd.a]); break;
            case LOG: v[n] = std::log(v[nd.a]); break;
            case SQRT: v[n] = std::sqrt(v[nd.a]); break;
            case POWC: v[n] = std::pow(v[nd.a], nd.c); break;
            case MAX: v[n] = v[nd.a] > v[nd.b] ? v[nd.a] : v[nd.b]; break;
            case MIN: v[n] = v[nd.a] < v[nd.b] ? v[nd.a] : v[nd.b]; break;
            case ABS: v[n] = std::fabs(v[nd.a]); break;
            case STEP: v[n] = v[nd.a] > 0.0 ? 1.0 : 0.0; break;
        }
    }
    return v[node];
}

// Straight-line code for the nodes the outputs depend on, then the assignments to `targets`.
std::string Trace::emit(const std::vector<int>& outs, const std::vector<std::string>& targets, bool accumulate,
                        const std::string& ind) const {
    std::set<int> need;
    std::vector<int> stack(outs.begin(), outs.end());
    while (!stack.empty()) {
        const int n = stack.back();
        stack.pop_back();
        if (n < 0 || need.count(n)) continue;
        need.insert(n);
        const Node& nd = _nodes[n];
        if (nd.op >= ADD) {
            stack.push_back(nd.a);
            if (nd.b >= 0) stack.push_back(nd.b);
        }
    }
    std::ostringstream o;
    o.precision(17);
    auto ref = [&](int n) -> std::string {
        const Node& nd = _nodes[n];
        std::ostringstream r;
        r.precision(17);
        switch (nd.op) {
            case CONST: r << "T(" << std::scientific << nd.c << ")"; return r.str();
            case IN_STATE: r << "z[" << nd.a << "]"; return r.str();
            case IN_CONTROL: r << "z[NS + " << nd.a << "]"; return r.str();
            case IN_TIME: return "tk";
            case IN_COEF: r << "cc[" << nd.a << "]"; return r.str();
            case IN_PARAM: r << "P.p[" << nd.a << "]"; return r.str();
            default: r << "v" << n; return r.str();
        }
    };
    for (int n : need) {
        const Node& nd = _nodes[n];
        if (nd.op < ADD) continue;
        o << ind << "const T v" << n << " = ";
        switch (nd.op) {
            case ADD: o << ref(nd.a) << " + " << ref(nd.b); break;
            case SUB: o << ref(nd.a) << " - " << ref(nd.b); break;
            case MUL: o << ref(nd.a) << " * " << ref(nd.b); break;
            case DIV: o << ref(nd.a) << " / " << ref(nd.b); break;
            case NEG: o << "-" << ref(nd.a); break;
            case SIN: o << "emi_sin(" << ref(nd.a) << ")"; break;
            case COS: o << "emi_cos(" << ref(nd.a) << ")"; break;
            case TAN: o << "emi_tan(" << ref(nd.a) << ")"; break;
            case EXP: o << "emi_exp(" << ref(nd.a) << ")"; break;
            case LOG: o << "emi_log(" << ref(nd.a) << ")"; break;
            case SQRT: o << "emi_sqrt(" << ref(nd.a) << ")"; break;
            case POWC: o << "emi_pow(" << ref(nd.a) << ", T(" << std::scientific << nd.c << "))"; break;
            case MAX: o << "(" << ref(nd.a) << " > " << ref(nd.b) << " ? " << ref(nd.a) << " : " << ref(nd.b) << ")"; break;
            case MIN: o << "(" << ref(nd.a) << " < " << ref(nd.b) << " ? " << ref(nd.a) << " : " << ref(nd.b) << ")"; break;
            case ABS: o << "(" << ref(nd.a) << " < T(0) ? -" << ref(nd.a) << " : " << ref(nd.a) << ")"; break;
            case STEP: o << "(" << ref(nd.a) << " > T(0) ? T(1) : T(0))"; break;
            default: break;
        }
        o << ";\n";
    }
    for (size_t i = 0; i < outs.size(); ++i) {
        if (outs[i] < 0) continue;
        o << ind << targets[i] << (accumulate ? " += " : " = ") << ref(outs[i]) << ";\n";
    }
    return o.str();
}

std::vector<int> Trace::dependencies(int out, int ns, int nc) {
    std::vector<int> deps;
    std::set<int> seen;
    std::vector<int> stack = {out};
    while (!stack.empty()) {
        const int n = stack.back();
        stack.pop_back();
        if (n < 0 || seen.count(n)) continue;
        seen.insert(n);
        const Node& nd = _nodes[n];
        if (nd.op == IN_STATE && nd.a < ns) deps.push_back(nd.a);
        else if (nd.op == IN_CONTROL && nd.a < nc) deps.push_back(ns + nd.a);
        else if (nd.op >= ADD) {
            stack.push_back(nd.a);
            if (nd.b >= 0) stack.push_back(nd.b);
        }
    }
    std::sort(deps.begin(), deps.end());
    return deps;
}

std::string Trace::generate_model(const std::string& name, int ns, int nc, const std::vector<int>& f, int L,
                                  const std::vector<int>& paths, std::vector<int>* path_vars, std::string* err) {
    const int nv = ns + nc;
    (void)err;
    auto in_node = [&](int v) { return v < ns ? input(IN_STATE, v) : input(IN_CONTROL, v - ns); };
    // path rows: first derivatives w.r.t. the union of the variables the rows depend on, second derivatives of
    // psi = sum_j mu_j c_j w.r.t. the same set
    const int npath = (int)paths.size();
    std::vector<int> pv;                           // union of dependencies, ascending
    for (int j = 0; j < npath; ++j)
        for (int v : dependencies(paths[j], ns, nc))
            if (std::find(pv.begin(), pv.end(), v) == pv.end()) pv.push_back(v);
    std::sort(pv.begin(), pv.end());
    if (npath > 0 && pv.empty()) pv.push_back(0);  // rows of time only: one (zero) column keeps the layout regular
    const int pw = (int)pv.size();
    if (path_vars) *path_vars = pv;
    std::vector<std::vector<int>> cd(npath, std::vector<int>(pw, -1));
    std::vector<int> PH;                           // Hessian entries of psi (packed positions of the full triangle)
    std::vector<std::string> PHt;
    if (npath > 0) {
        for (int j = 0; j < npath; ++j) {
            const std::vector<int> adj = adjoints(paths[j]);
            for (int q = 0; q < pw; ++q) {
                const int n = in_node(pv[q]);
                cd[j][q] = n < (int)adj.size() ? adj[n] : -1;
            }
        }
        int psi = -1;
        for (int j = 0; j < npath; ++j) {
            const int term = binary(MUL, input(IN_COEF, j), paths[j]);
            psi = psi < 0 ? term : binary(ADD, psi, term);
        }
        const std::vector<int> a1 = adjoints(psi);
        for (int qa = 0; qa < pw; ++qa) {
            const int na = in_node(pv[qa]);
            const int ga = na < (int)a1.size() ? a1[na] : -1;
            if (ga < 0) continue;
            const std::vector<int> a2 = adjoints(ga);
            for (int qb = 0; qb <= qa; ++qb) {
                const int nb = in_node(pv[qb]);
                const int h = nb < (int)a2.size() ? a2[nb] : -1;
                double c;
                if (h < 0 || (is_const(h, &c) && c == 0.0)) continue;
                PH.push_back(h);
                PHt.push_back("H[" + std::to_string(pv[qa] * (pv[qa] + 1) / 2 + pv[qb]) + "]");     // pv ascending: qa >= qb
            }
        }
    }
    // first derivatives
    std::vector<std::vector<int>> J(ns, std::vector<int>(nv, -1));
    for (int i = 0; i < ns; ++i) {
        const std::vector<int> adj = adjoints(f[i]);
        for (int v = 0; v < nv; ++v) {
            const int n = in_node(v);
            J[i][v] = n < (int)adj.size() ? adj[n] : -1;
        }
    }
    std::vector<int> gL(nv, -1);
    {
        const std::vector<int> adj = adjoints(L);
        for (int v = 0; v < nv; ++v) {
            const int n = in_node(v);
            gL[v] = n < (int)adj.size() ? adj[n] : -1;
        }
    }
    // Lagrangian Hessian: phi = cc[0] * L + sum_i cc[1+i] * f_i ; H = d/dz (d phi/dz)
    int phi = binary(MUL, input(IN_COEF, 0), L);
    for (int i = 0; i < ns; ++i) phi = binary(ADD, phi, binary(MUL, input(IN_COEF, 1 + i), f[i]));
    std::vector<int> g(nv, -1);
    {
        const std::vector<int> adj = adjoints(phi);
        for (int v = 0; v < nv; ++v) {
            const int n = in_node(v);
            g[v] = n < (int)adj.size() ? adj[n] : -1;
        }
    }
    std::vector<int> H;
    std::vector<std::string> Ht;
    for (int v = 0; v < nv; ++v) {
        std::vector<int> adj;
        if (g[v] >= 0) adj = adjoints(g[v]);
        for (int q = 0; q <= v; ++q) {
            const int n = in_node(q);
            int h = (g[v] >= 0 && n < (int)adj.size()) ? adj[n] : -1;
            double c;
            if (h >= 0 && is_const(h, &c) && c == 0.0) h = -1;
            H.push_back(h);
            Ht.push_back("H[" + std::to_string(v * (v + 1) / 2 + q) + "]");
        }
    }

    std::ostringstream o;
    o << "template <typename T> struct " << name << " {\n";
    o << "    static constexpr int NS = " << ns << ", NC = " << nc << ", NV = " << nv << ", NPARAM = 0, NPATH = " << npath
      << ", PW = " << pw << ";\n";
    if (npath > 0) {
        // pvar(q): the q-th node variable the traced rows depend on
        o << "    EMI_DEV static constexpr int pvar(int q) { return ";
        for (int q = 0; q + 1 < pw; ++q) o << "q == " << q << " ? " << pv[q] << " : ";
        o << pv[pw - 1] << "; }\n";
    }
    // f
    o << "    EMI_DEV static void f(const ModelParams<T>&, const T* z, T tk, T* fo) {\n";
    {
        std::vector<std::string> t;
        for (int i = 0; i < ns; ++i) t.push_back("fo[" + std::to_string(i) + "]");
        o << emit(f, t, false, "        ");
    }
    o << "        (void)tk;\n    }\n";
    // jac
    o << "    EMI_DEV static void jac(const ModelParams<T>&, const T* z, T tk, T (*J)[NV]) {\n";
    o << "        for (int i = 0; i < NS; ++i)\n            for (int v = 0; v < NV; ++v) J[i][v] = T(0);\n";
    {
        std::vector<int> outs;
        std::vector<std::string> t;
        for (int i = 0; i < ns; ++i)
            for (int v = 0; v < nv; ++v) {
                double c;
                if (J[i][v] < 0 || (is_const(J[i][v], &c) && c == 0.0)) continue;
                outs.push_back(J[i][v]);
                t.push_back("J[" + std::to_string(i) + "][" + std::to_string(v) + "]");
            }
        o << emit(outs, t, false, "        ");
    }
    o << "        (void)tk; (void)z;\n    }\n";
    // cost
    o << "    EMI_DEV static T cost(const ModelParams<T>&, const T* z, T tk) {\n        T Lv;\n";
    o << emit({L}, {"Lv"}, false, "        ");
    o << "        (void)tk;\n        return Lv;\n    }\n";
    // grad
    o << "    EMI_DEV static void grad(const ModelParams<T>&, const T* z, T tk, T* g) {\n";
    o << "        for (int v = 0; v < NV; ++v) g[v] = T(0);\n";
    {
        std::vector<int> outs;
        std::vector<std::string> t;
        for (int v = 0; v < nv; ++v) {
            if (gL[v] < 0) continue;
            outs.push_back(gL[v]);
            t.push_back("g[" + std::to_string(v) + "]");
        }
        o << emit(outs, t, false, "        ");
    }
    o << "        (void)tk; (void)z;\n    }\n";
    // hess: cc[0] = cL, cc[1+i] = cf[i]
    o << "    EMI_DEV static void hess(const ModelParams<T>&, const T* z, T tk, T cL, const T* cf, T* H) {\n";
    o << "        T cc[NS + 1];\n        cc[0] = cL;\n        for (int i = 0; i < NS; ++i) cc[1 + i] = cf[i];\n";
    o << emit(H, Ht, true, "        ");
    o << "        (void)tk; (void)z; (void)cc;\n    }\n";
    if (npath > 0) {
        // path rows: values and partials w.r.t. the PW variables of pvar()
        o << "    EMI_DEV static void path(const ModelParams<T>&, const T* z, T tk, T* c, T* cd) {\n";
        o << "        for (int j = 0; j < NPATH * PW; ++j) cd[j] = T(0);\n";
        {
            std::vector<int> outs;
            std::vector<std::string> t;
            for (int j = 0; j < npath; ++j) {
                outs.push_back(paths[j]);
                t.push_back("c[" + std::to_string(j) + "]");
                for (int q = 0; q < pw; ++q) {
                    double cc;
                    if (cd[j][q] < 0 || (is_const(cd[j][q], &cc) && cc == 0.0)) continue;
                    outs.push_back(cd[j][q]);
                    t.push_back("cd[" + std::to_string(j * pw + q) + "]");
                }
            }
            o << emit(outs, t, false, "        ");
        }
        o << "        (void)tk; (void)z;\n    }\n";
        // adds sum_j cc[j] c_j,zz into the packed lower triangle H of the node block
        o << "    EMI_DEV static void path_hess(const ModelParams<T>&, const T* z, T tk, const T* cc, T* H) {\n";
        o << emit(PH, PHt, true, "        ");
        o << "        (void)tk; (void)z; (void)cc; (void)H;\n    }\n";
    }
    o << "};\n";
    return o.str();
}

std::string Trace::generate_hess_body(int ns, int nc, const std::vector<int>& f, int L) {
    const int nv = ns + nc;
    auto in_node = [&](int v) { return v < ns ? input(IN_STATE, v) : input(IN_CONTROL, v - ns); };
    int phi = binary(MUL, input(IN_COEF, 0), L);
    for (int i = 0; i < ns; ++i) phi = binary(ADD, phi, binary(MUL, input(IN_COEF, 1 + i), f[i]));
    std::vector<int> g(nv, -1);
    {
        const std::vector<int> adj = adjoints(phi);
        for (int v = 0; v < nv; ++v) {
            const int n = in_node(v);
            g[v] = n < (int)adj.size() ? adj[n] : -1;
        }
    }
    std::vector<int> H;
    std::vector<std::string> Ht;
    for (int v = 0; v < nv; ++v) {
        std::vector<int> adj;
        if (g[v] >= 0) adj = adjoints(g[v]);
        for (int q = 0; q <= v; ++q) {
            const int n = in_node(q);
            int h = (g[v] >= 0 && n < (int)adj.size()) ? adj[n] : -1;
            double c;
            if (h >= 0 && is_const(h, &c) && c == 0.0) h = -1;
            if (h < 0) continue;
            H.push_back(h);
            Ht.push_back("H[" + std::to_string(v * (v + 1) / 2 + q) + "]");
        }
    }
    return emit(H, Ht, true, "        ");
}

// ---- Var arithmetic (declared in include/ETOL/eMI355X_Types.hpp) -------------------------------
namespace {
int node_of(const Var& v) {
    if (v.node < 0) {
        fprintf(stderr, "mi355x::Var used outside a callback trace\n");
        exit(EXIT_FAILURE);
    }
    return v.node;
}
Var wrap(int n) {
    Var r;
    r.kind = Var::EXPR;
    r.node = n;
    return r;
}
}  // namespace

Var::Var(double constant) : kind(EXPR), index(0), node(Trace::active().constant(constant)) {}
Var::Var(Kind k, size_t i) : kind(k), index(i), node(-1) {
    Trace& t = Trace::active();
    node = t.input(k == STATE ? Trace::IN_STATE : (k == CONTROL ? Trace::IN_CONTROL : Trace::IN_TIME), (int)i);
}
Var operator+(const Var& a, const Var& b) { return wrap(Trace::active().binary(Trace::ADD, node_of(a), node_of(b))); }
Var operator-(const Var& a, const Var& b) { return wrap(Trace::active().binary(Trace::SUB, node_of(a), node_of(b))); }
Var operator*(const Var& a, const Var& b) { return wrap(Trace::active().binary(Trace::MUL, node_of(a), node_of(b))); }
Var operator/(const Var& a, const Var& b) { return wrap(Trace::active().binary(Trace::DIV, node_of(a), node_of(b))); }
Var operator-(const Var& a) { return wrap(Trace::active().unary(Trace::NEG, node_of(a))); }
Var sin(const Var& a) { return wrap(Trace::active().unary(Trace::SIN, node_of(a))); }
Var cos(const Var& a) { return wrap(Trace::active().unary(Trace::COS, node_of(a))); }
Var tan(const Var& a) { return wrap(Trace::active().unary(Trace::TAN, node_of(a))); }
Var exp(const Var& a) { return wrap(Trace::active().unary(Trace::EXP, node_of(a))); }
Var log(const Var& a) { return wrap(Trace::active().unary(Trace::LOG, node_of(a))); }
Var sqrt(const Var& a) { return wrap(Trace::active().unary(Trace::SQRT, node_of(a))); }
Var max(const Var& a, const Var& b) { return wrap(Trace::active().binary(Trace::MAX, node_of(a), node_of(b))); }
Var min(const Var& a, const Var& b) { return wrap(Trace::active().binary(Trace::MIN, node_of(a), node_of(b))); }
Var abs(const Var& a) { return wrap(Trace::active().unary(Trace::ABS, node_of(a))); }
// piecewise-linear interpolation of the table (tw, vw) at t, constant outside it:
//   v_0 + sum_i slope_i * (clamp(t, tw_i, tw_{i+1}) - tw_i)
Var interp1(const std::vector<double>& tw, const std::vector<double>& vw, const Var& t) {
    if (tw.size() != vw.size() || tw.empty()) {
        fprintf(stderr, "mi355x::interp1: table sizes differ or are empty\n");
        exit(EXIT_FAILURE);
    }
    Var out(vw[0]);
    for (size_t i = 0; i + 1 < tw.size(); ++i) {
        if (!(tw[i + 1] > tw[i])) continue;
        const double slope = (vw[i + 1] - vw[i]) / (tw[i + 1] - tw[i]);
        out = out + slope * (min(max(t, Var(tw[i])), Var(tw[i + 1])) - tw[i]);
    }
    return out;
}
Var pow(const Var& a, double c) { return wrap(Trace::active().unary(Trace::POWC, node_of(a), c)); }

}  // namespace mi355x
}  // namespace ETOL
