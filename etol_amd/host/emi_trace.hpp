// emi_trace.hpp -- expression trace of user callbacks, symbolic derivatives, model code generation.
//
// ePSOPT gets derivatives of the user's callbacks from ADOL-C: the callbacks run on `adouble`, the
// operations are recorded on a tape, and the tape is interpreted for every Jacobian / Hessian
// evaluation (reference src/ePSOPT/ePSOPT.cpp:64-65 "automatic", "exact").  eMI355X records the
// same kind of trace ONCE at setup() (mi355x::Var handles), differentiates it symbolically on the
// host and emits a model struct (f, jac, cost, grad, hess) in the form the hand-written kernel
// templates of etol_amd/csrc expect; the device evaluates straight-line code, never a tape.
#ifndef ETOL_MI355X_EMI_TRACE_HPP_
#define ETOL_MI355X_EMI_TRACE_HPP_

#include <map>
#include <string>
#include <tuple>
#include <vector>

#include <ETOL/eMI355X_Types.hpp>

namespace ETOL {
namespace mi355x {

class Trace {
 public:
    enum Op { CONST, IN_STATE, IN_CONTROL, IN_TIME, IN_COEF, IN_PARAM, ADD, SUB, MUL, DIV, MAX, MIN, NEG, SIN, COS, TAN, EXP, LOG, SQRT, POWC,
              ABS, STEP };    // MAX/MIN binary; STEP(a) = a > 0 ? 1 : 0 (the derivative of max / min / abs);
                              // IN_PARAM a: entry a of the model's parameter block (P.p[a] in generated code)
    struct Node {
        Op op;
        int a, b;        // operand nodes (or input index in a for IN_*)
        double c;        // constant / exponent
    };

    void clear();
    int constant(double c);
    int input(Op kind, int index);
    int unary(Op op, int a, double c = 0.0);
    int binary(Op op, int a, int b);
    const std::vector<Node>& nodes() const { return _nodes; }

    // adjoint sweep: d out / d every node that out depends on, as new nodes; result[n] = -1 if zero
    std::vector<int> adjoints(int out);

    // C++ source of  `template <typename T> struct <name> {...}`  with the Model interface of
    // etol_amd/csrc/emi_models.hpp for dynamics f[0..ns), integrand cost L (already sign-free)
    // `paths`: traced path rows c_j; they may depend on any of the node's states and controls (and on time).  The
    // struct gets PW = the number of distinct variables any row depends on, pvar(q) = the q-th of them (ascending),
    // path(P, z, t, c, cd) with cd[j * PW + q] = d c_j / d z_pvar(q), and path_hess(P, z, t, mu, H) adding
    // sum_j mu_j c_j,zz into the packed lower triangle H.  *path_vars receives the variable list.
    std::string generate_model(const std::string& name, int ns, int nc, const std::vector<int>& f, int L,
                               const std::vector<int>& paths = {}, std::vector<int>* path_vars = nullptr,
                               std::string* err = nullptr);
    // the state / control inputs node `out` depends on (indices v < ns: states, else controls)
    std::vector<int> dependencies(int out, int ns, int nc);

    // numeric evaluation on the host (unit tests of the trace itself)
    double eval(int node, const std::vector<double>& x, const std::vector<double>& u, double t,
                const std::vector<double>& coef = {}) const;

    // C++ statements adding the Lagrangian Hessian  cL*L_zz + sum_i cf[i]*f_i,zz  into H (packed lower triangle): the body of
    // a model's hess(); used to generate the second derivatives of hand-written models (parameters as IN_PARAM inputs)
    std::string generate_hess_body(int ns, int nc, const std::vector<int>& f, int L);
    std::vector<double> param_values;    // values of the IN_PARAM inputs for eval()

    static Trace& active();

 private:
    std::vector<Node> _nodes;
    std::map<std::tuple<int, int, int, double>, int> _cse;
    int intern(const Node& n);
    bool is_const(int n, double* v = nullptr) const;
    std::string emit(const std::vector<int>& outs, const std::vector<std::string>& targets, bool accumulate,
                     const std::string& indent) const;
};

}  // namespace mi355x
}  // namespace ETOL
#endif
